"""The variant of ``k_dgp_forward`` that bench.py times (LEAN, csrc/dgp_forward.hip): the bound's own evaluation of an all-RBF M = 128 stack
at 80 samples per workgroup -- compile-time shapes, no per-layer outputs, the half-wave-per-point tail.  It cannot return the draws it made,
so it is pinned to the general variant: the SAME evaluation (same seed, same device-resident step) through both, and the general variant is
what the oracle suites cover (tests/test_gpu_device_noise.py reads its draws back, tests/test_gpu_parity.py injects them).  The two differ
in the order of one sum (the logsumexp over k is a tree in LEAN), nothing else.  The last test closes the loop at the bench shape itself:
the LEAN evaluation of BASELINE.json configs[2] at full size against the float64 oracle, on the draws read back from the same noise step."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LEAN_BIT, S16_BIT, SHAPES_BIT = 1 << 10, 1 << 8, 1 << 11      # (bit 11: the variant with the same shapes compiled in that keeps outputs and the general tail)


def _last_variant():
    from dgps_with_iwvi_amd import _abi
    f = _abi.lib().iwvi_debug_last_forward_variant
    f.restype, f.argtypes = ctypes.c_int, []
    return int(f())


def _evaluate(model, spec, force_general):
    """One IW-ELBO evaluation from step 0 of the model's noise stream -> (elbo, logp [B], log-weights [B, K], variant bits)."""
    from dgps_with_iwvi_amd import _abi
    B, K = spec["B"], spec["K"]
    model._words().zero_()                                       # the device-resident counter: both evaluations draw the same numbers
    model.precompute(with_encoders=True)
    _abi.set_debug_option("IWVI_FW_NO_LEAN", 1 if force_general else 0)
    try:
        logw, _, red = model._fused_forward(B * K, K, B, (B, K), zs=None, sampled_kl=True,
                                            elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
        torch.cuda.synchronize()
        variant = _last_variant()
    finally:
        _abi.set_debug_option("IWVI_FW_NO_LEAN", 0)
    return float(red[0]), red[1].double().cpu().numpy(), logw.double().cpu().numpy().reshape(B, K), variant


# (K must divide 80 and lie in 5..32 for the variant's tail; T = B K a multiple of 80 and large enough for 80-sample workgroups)
@pytest.mark.parametrize("K,B,with_lv", [(20, 1024, True),       # BASELINE.json configs[2]: the bench line
                                         (10, 2048, True),
                                         (16, 1280, False),
                                         (5, 4096, True)])
def test_lean_variant_equals_the_general_variant_on_the_same_draws(gpu_device, K, B, with_lv):
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=B, K=K, with_lv=with_lv, seed=K, n_data=65536)
    settings.set_seed(99)
    model = synthetic.build_model(spec, gpu_device)
    e_lean, lp_lean, lw_lean, v_lean = _evaluate(model, spec, force_general=False)
    e_gen, lp_gen, lw_gen, v_gen = _evaluate(model, spec, force_general=True)
    assert v_lean & LEAN_BIT and (v_lean & 0xff) == 5, "the bench shape must take the LEAN variant (%#x)" % v_lean
    assert not (v_gen & (LEAN_BIT | SHAPES_BIT)) and v_gen & S16_BIT       # (IWVI_FW_NO_LEAN: the fully general variant)
    assert np.isfinite(lw_lean).all()
    # per-sample log-weights: the same arithmetic in both -- identical bits
    np.testing.assert_array_equal(lw_lean, lw_gen)
    # per-point logsumexp: tree against serial sum of K float32 terms
    np.testing.assert_allclose(lp_lean, lp_gen, rtol=0, atol=4e-6 * max(1.0, np.abs(lp_gen).max()))
    assert abs(e_lean - e_gen) <= 2e-7 * abs(e_gen), (e_lean, e_gen)
    # and the draws move on: the next evaluation is another sample of the bound
    logw2, _, red2 = model._fused_forward(B * K, K, B, (B, K), zs=None, sampled_kl=True,
                                          elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
    assert float(red2[0]) != e_lean


def test_fp32_route_has_its_compiled_in_shapes_variant_too(gpu_device):
    """``settings.fw_f32_stage2`` (IWVI_LAYER_F32_STAGE2: fp32 MFMAs in stage 2) at the bench shape: since round 5 it takes a LEAN
    instantiation of its own (it ran the general variant: 77.7 us per evaluation against 66.4 when fp32 was the default)
    -- same log-weights, bit for bit, as the general fp32 variant on the same draws."""
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=1024, K=20, with_lv=True, seed=20, n_data=65536)
    settings.set_seed(77)
    old, settings.fw_f32_stage2 = settings.fw_f32_stage2, True
    try:
        model = synthetic.build_model(spec, gpu_device)
        e_lean, lp_lean, lw_lean, v_lean = _evaluate(model, spec, force_general=False)
        e_gen, lp_gen, lw_gen, v_gen = _evaluate(model, spec, force_general=True)
    finally:
        settings.fw_f32_stage2 = old
    assert v_lean & LEAN_BIT and not v_lean & S16_BIT and (v_lean & 0xff) == 5, hex(v_lean)
    assert not v_gen & (LEAN_BIT | SHAPES_BIT | S16_BIT), hex(v_gen)
    np.testing.assert_array_equal(lw_lean, lw_gen)
    np.testing.assert_allclose(lp_lean, lp_gen, rtol=0, atol=4e-6 * max(1.0, np.abs(lp_gen).max()))
    assert abs(e_lean - e_gen) <= 2e-7 * abs(e_gen), (e_lean, e_gen)


@pytest.mark.parametrize("why,kw", [("M = 64", dict(L=2, M=64, B=1024, K=20, with_lv=True)),
                                    ("K = 40 (tail needs K <= 32)", dict(L=2, M=128, B=512, K=40, with_lv=True)),
                                    ("a ragged last workgroup", dict(L=2, M=128, B=1023, K=20, with_lv=True))])
def test_shapes_outside_the_variant_take_the_general_one(gpu_device, why, kw):
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(seed=3, n_data=65536, **kw)
    settings.set_seed(5)
    model = synthetic.build_model(spec, gpu_device)
    e, lp, lw, v = _evaluate(model, spec, force_general=False)
    assert not (v & LEAN_BIT), why
    assert np.isfinite(e) and np.isfinite(lw).all()


def test_requested_layer_outputs_take_the_variant_that_keeps_them(gpu_device):
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=1024, K=20, with_lv=True, seed=0, n_data=65536)
    settings.set_seed(5)
    model = synthetic.build_model(spec, gpu_device)
    B, K = spec["B"], spec["K"]
    model.precompute(with_encoders=True)
    model._fused_forward(B * K, K, B, (B, K), zs=None, sampled_kl=True, want_layers=True,
                         elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
    torch.cuda.synchronize()
    assert not (_last_variant() & LEAN_BIT) and (_last_variant() & SHAPES_BIT)


def test_lean_variant_against_the_oracle_at_the_bench_shape(gpu_device):
    """BASELINE.json configs[2] at full size: the LEAN evaluation's bound and per-point terms against the float64 oracle, fed the draws the
    general variant reads back from the SAME noise step (the two variants' log-weights are identical bits, see above).  Tolerances of the
    device-noise suite: ELBO relative 1e-4, per-point logsumexp rtol 2e-4 + atol 2e-2."""
    from dgps_with_iwvi_amd import settings, synthetic
    from oracle.from_spec import build_oracle, oracle_noise
    spec = synthetic.make_spec(L=2, M=128, B=1024, K=20, with_lv=True, seed=20, n_data=65536)
    B, K = spec["B"], spec["K"]
    settings.set_seed(4321)
    model = synthetic.build_model(spec, gpu_device)
    e_lean, lp_lean, lw_lean, v_lean = _evaluate(model, spec, force_general=False)
    assert v_lean & LEAN_BIT
    model._words().zero_()                                       # the same noise step once more, through the variant that returns its draws
    model.precompute(with_encoders=True)
    lw_gen, outs, red = model._fused_forward(B * K, K, B, (B, K), zs=None, sampled_kl=True, want_layers=True, want_saved=True,
                                             elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
    torch.cuda.synchronize()
    assert not (_last_variant() & LEAN_BIT)
    np.testing.assert_array_equal(lw_lean, lw_gen.double().cpu().numpy().reshape(B, K))
    zs = [o["noise_out"].double().cpu().numpy().reshape(B, K, -1) for o in outs]
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    assert abs(e_lean - ref) <= 1e-4 * abs(ref), (e_lean, ref)
    L_NK = om.log_weights(oracle_noise(spec, zs))[0]
    m_o = L_NK.max(1)
    np.testing.assert_allclose(lp_lean, m_o + np.log(np.exp(L_NK - m_o[:, None]).sum(1)) - np.log(K), rtol=2e-4, atol=2e-2)
    np.testing.assert_allclose(lw_lean, L_NK, rtol=2e-4, atol=2e-2)


def test_the_variant_that_keeps_outputs_equals_the_general_variant(gpu_device):
    """Mode 2 (shapes compiled in, per-layer outputs and the general tail kept: the forward of a value + gradient evaluation) against the fully
    general variant on the same draws: every per-layer output, every saved operand of the adjoint and the log-weights, bit for bit."""
    from dgps_with_iwvi_amd import _abi, settings, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=1024, K=20, with_lv=True, seed=7, n_data=65536)
    B, K = spec["B"], spec["K"]
    settings.set_seed(11)
    model = synthetic.build_model(spec, gpu_device)
    got = []
    for force_general in (False, True):
        model._words().zero_()
        model.precompute(with_encoders=True)
        _abi.set_debug_option("IWVI_FW_NO_LEAN", 1 if force_general else 0)
        try:
            lw, outs, red = model._fused_forward(B * K, K, B, (B, K), zs=None, sampled_kl=True, want_layers=True, want_saved=True,
                                                 elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
            torch.cuda.synchronize()
            v = _last_variant()
        finally:
            _abi.set_debug_option("IWVI_FW_NO_LEAN", 0)
        assert bool(v & SHAPES_BIT) == (not force_general) and not (v & LEAN_BIT)
        got.append((lw.clone(), [{k: t.clone() for k, t in o.items()} for o in outs], float(red[0])))
    (lw_a, outs_a, e_a), (lw_b, outs_b, e_b) = got
    assert torch.equal(lw_a, lw_b) and e_a == e_b
    for oa, ob in zip(outs_a, outs_b):
        assert oa.keys() == ob.keys()
        for k in oa:
            assert torch.equal(oa[k], ob[k]), k
