"""ABI 17: IWVI_GP_REUSE_FACTOR / IWVI_GP_FACTOR_ONLY (include/iwvi_hip.h) -- the two halves of a precompute, for callers that know which
inputs moved (a training step between its two ops).  Checked on the state buffer itself: bit for bit against full precomputes."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _layer_and_state(gpu_device, M, R, seed):
    from dgps_with_iwvi_amd import synthetic, settings
    settings.set_seed(seed)
    spec = synthetic.make_spec(L=2, M=M, B=32, K=2, R=R, with_lv=False, seed=seed)
    model = synthetic.build_model(spec, gpu_device)
    return model, model.layers[0]


def _run(descs):
    from dgps_with_iwvi_amd.temp_workaround import precompute_states
    precompute_states(descs)
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,R", [(128, 5), (64, 1), (48, 3), (256, 2)])
def test_reuse_factor_rewrites_exactly_the_q_images(gpu_device, M, R):
    """full precompute with q0 -> q moves -> IWVI_GP_REUSE_FACTOR  ==  a full precompute with the moved q, byte for byte; and the bytes it
    changed are a strict subset of the buffer (the factorisation's images are not rewritten: poisoned bytes there would survive)."""
    from dgps_with_iwvi_amd import _abi
    model, l = _layer_and_state(gpu_device, M, R, 3)
    _run([l.state_desc()])
    before = l.state().buf.clone()
    gen = torch.Generator(device="cpu").manual_seed(5)
    l.q_mu = (l.q_mu + 0.3 * torch.randn(l.q_mu.shape, generator=gen).to(gpu_device)).contiguous()
    l.q_sqrt = (l.q_sqrt + 0.05 * torch.tril(torch.randn(l.q_sqrt.shape, generator=gen)).to(gpu_device)).contiguous()
    d = l.state_desc()
    d.flags |= _abi.GP_REUSE_FACTOR
    _run([d])
    reused = l.state().buf.clone()
    _run([l.state_desc()])                                        # the reference: everything again
    full = l.state().buf.clone()
    assert torch.equal(reused, full)
    changed = (reused != before)
    assert changed.any() and not changed.all()
    # the factor part is left alone: poison a byte range the q roles never write (the forward-substitution stream) and repeat
    offs = (ctypes.c_size_t * 8)()
    assert _abi.lib().iwvi_gp_state_offsets(M, R, offs) == 0
    lo = int(offs[2])                                             # {Lm, Linv, LsP, ...}: the forward-substitution stream, written by the factorisation only
    l.state().buf[lo:lo + 64] = 0x5A
    d = l.state_desc()
    d.flags |= _abi.GP_REUSE_FACTOR
    _run([d])
    assert bool((l.state().buf[lo:lo + 64] == 0x5A).all())


@pytest.mark.parametrize("M,R", [(128, 5), (64, 1)])
def test_factor_only_leaves_the_q_images_alone(gpu_device, M, R):
    """IWVI_GP_FACTOR_ONLY after a full precompute with other kernel parameters: the buffer equals a full precompute with the new kernel
    parameters wherever the factorisation writes, and the q images (identical q in both) are untouched -- so the whole buffer is equal; and
    with q poisoned meanwhile the call neither reads nor writes anything of q (no NaN reaches the buffer)."""
    from dgps_with_iwvi_amd import _abi
    model, l = _layer_and_state(gpu_device, M, R, 7)
    _run([l.state_desc()])
    k = l._base_kern()
    k.lengthscales = (k.lengthscales * 1.3).contiguous()
    good_q = l.q_sqrt.clone()
    l.q_sqrt = torch.full_like(l.q_sqrt, float("nan"))            # q is "being written by another stream"
    d = l.state_desc()
    d.flags |= _abi.GP_FACTOR_ONLY
    _run([d])
    half = l.state().buf.clone()
    l.q_sqrt = good_q
    _run([l.state_desc()])
    assert torch.equal(half, l.state().buf)


def test_both_flags_are_refused(gpu_device):
    from dgps_with_iwvi_amd import _abi
    from dgps_with_iwvi_amd.temp_workaround import precompute_states
    model, l = _layer_and_state(gpu_device, 32, 1, 1)
    d = l.state_desc()
    d.flags |= _abi.GP_REUSE_FACTOR | _abi.GP_FACTOR_ONLY
    with pytest.raises(_abi.IwviError):
        precompute_states([d])


def test_q_moved_needs_the_side_stream_preparation(gpu_device):
    """backward.iw_elbo_and_gradients(q_moved=...) is the second half of a step: not with wrt='final_q', not without overlap."""
    from dgps_with_iwvi_amd import backward
    model, _ = _layer_and_state(gpu_device, 32, 1, 2)
    with pytest.raises(ValueError):
        backward.iw_elbo_and_gradients(model, wrt="final_q", q_moved={1})
    with pytest.raises(ValueError):
        backward.iw_elbo_and_gradients(model, overlap=False, q_moved={1})
    with pytest.raises(ValueError):
        model.precompute(q_moved=set())
