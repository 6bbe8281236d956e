"""The merged launch (``iwvi_dgp_forward_fused``: the inducing-set factorisations run in workgroups of the layer stack's own
launch) against the two launches it replaces (``iwvi_model_precompute`` + ``iwvi_dgp_forward``): the SAME arithmetic in the same
order, so every result must agree BIT FOR BIT -- ELBO, per-point terms, per-layer outputs -- for injected and for device-drawn
noise, over repeated calls (the sync words are generation-counted, never reset), under hipGraph replay, for stacks with more pack
jobs than chunks, and for stacks the merged launch hands back to the two launches (M > 128).  Reference lines replaced:
temp_workaround.py:39,48 (once per layer per evaluation) + models.py:112-150."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _model(spec, dev, cls=None):
    from dgps_with_iwvi_amd import synthetic
    return synthetic.build_model(spec, dev, cls=cls)


def _both(fn):
    """fn() under merged_launch False then True, each from the same seed / noise counters."""
    from dgps_with_iwvi_amd import settings
    out = []
    old = settings.merged_launch
    try:
        for merged in (False, True):
            settings.merged_launch = merged
            settings.set_seed(7)
            out.append(fn())
    finally:
        settings.merged_launch = old
    return out


CASES = [
    # L, M, B, K, with_lv, R
    (2, 128, 64, 20, True, 5),        # the headline family, 16 chunks of 80
    (2, 128, 1024, 20, True, 5),      # configs[2]: 256 chunks + 2 factorising workgroups on 256 CUs
    (2, 128, 1024, 5, False, 5),      # configs[1]
    (1, 64, 16, 1, False, 1),         # one chunk, one pack job
    (3, 96, 8, 2, True, 4),           # one chunk, nine pack jobs: a chunk packs more than once
    (2, 100, 48, 4, True, 3),         # odd block count: the fp32 stage 2, M not a multiple of 16
    (3, 32, 200, 7, False, 2),        # K does not divide the chunk: the last workgroup reduces from the log-weights
]


@pytest.mark.parametrize("L,M,B,K,with_lv,R", CASES)
def test_merged_launch_equals_two_launches_bitwise(gpu_device, L, M, B, K, with_lv, R):
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, R=R, with_lv=with_lv, seed=11)
    zs = [torch.as_tensor(z, dtype=torch.float32, device=gpu_device) for z in synthetic.make_noise(spec, seed=12)]

    def run():
        m = _model(spec, gpu_device)
        e_inj, logp_inj, _ = m._elbo_parts(zs)
        e1, logp1, _ = m._elbo_parts(None)                       # device noise, evaluation 0 of the stream
        e2, logp2, _ = m._elbo_parts(None)                       # evaluation 1: a second generation of the sync words
        return [t.clone() for t in (e_inj, logp_inj, e1, logp1, e2, logp2)], [l.state().kl_parts.clone() for l in m.layers if hasattr(l, "q_sqrt")]

    (a, kla), (b, klb) = _both(run)
    for x, y in zip(a, b):
        assert torch.equal(x, y), (x.flatten()[:4], y.flatten()[:4])
    for x, y in zip(kla, klb):
        assert torch.equal(x, y)
    assert torch.isfinite(a[0])
    if L > 1 or with_lv:
        assert not torch.equal(a[2], a[4])                         # fresh noise per evaluation (a lone final layer's draw is never used)


def test_merged_launch_matches_the_oracle(gpu_device):
    """... and the merged path itself against the float64 oracle (not only against the other launch sequence)."""
    from dgps_with_iwvi_amd import synthetic, settings
    from oracle.from_spec import build_oracle, oracle_noise
    spec = synthetic.make_spec(L=2, M=128, B=32, K=20, with_lv=True, seed=3)
    zs = synthetic.make_noise(spec, seed=4)
    old, settings.merged_launch = settings.merged_launch, True
    try:
        m = _model(spec, gpu_device)
        elbo = m.compute_log_likelihood([torch.as_tensor(z, dtype=torch.float32, device=gpu_device) for z in zs])
    finally:
        settings.merged_launch = old
    ref = build_oracle(spec).build_likelihood(oracle_noise(spec, zs))
    assert abs(elbo - ref) <= 1e-4 * abs(ref), (elbo, ref)


def test_merged_launch_vi_bound(gpu_device):
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.models import DGP_VI
    spec = synthetic.make_spec(L=2, M=64, B=96, K=3, with_lv=True, seed=5)

    def run():
        m = _model(spec, gpu_device, cls=DGP_VI)
        return [m._build_likelihood(None).clone() for _ in range(3)]

    a, b = _both(run)
    for x, y in zip(a, b):
        assert torch.equal(x, y) and torch.isfinite(x)


def test_merged_launch_hands_large_layers_to_the_two_launches(gpu_device):
    """M = 256 factorises in an L2 workspace, not in LDS: the merged entry point runs the two launches itself (same call, same result)."""
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=2, M=256, B=32, K=4, with_lv=False, seed=8)

    def run():
        m = _model(spec, gpu_device)
        return [m._elbo_parts(None)[0].clone() for _ in range(2)]

    a, b = _both(run)
    for x, y in zip(a, b):
        assert torch.equal(x, y) and torch.isfinite(x)


def test_merged_launch_replays_from_a_graph(gpu_device):
    """25 evaluations captured into one hipGraph (what bench.py times): every replayed evaluation draws fresh noise and equals the
    eager evaluation with the same counter; the generation word advances once per evaluation."""
    from dgps_with_iwvi_amd import synthetic, settings
    spec = synthetic.make_spec(L=2, M=128, B=256, K=20, with_lv=True, seed=9)
    n = 6
    settings.merged_launch, old_merged = True, settings.merged_launch
    settings.set_seed(21)
    m = _model(spec, gpu_device)
    eager = [m._elbo_parts(None)[0].clone() for _ in range(2 * n)]
    settings.set_seed(21)
    m2 = _model(spec, gpu_device)
    outs = [torch.zeros(1, dtype=torch.float64, device=gpu_device) for _ in range(n)]
    side = torch.cuda.Stream(device=gpu_device)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m2._elbo_parts(None)                                      # warm-up (allocations): advance, then rewind the counters
        m2._words().zero_()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for o in outs:
                m2._build_likelihood(None, out=o)
    torch.cuda.current_stream().wait_stream(side)
    got = []
    for _ in range(2):
        g.replay()
        torch.cuda.synchronize()
        got += [o.clone() for o in outs]
    for x, y in zip(eager, got):
        assert torch.equal(x.reshape(-1), y.reshape(-1)), (x, y)
    settings.merged_launch = old_merged
    words = m2._fz_ws().view(torch.int32)
    assert int(words[0]) == 2 * n + 1 and int(words[5]) == 0       # gen = launches completed; no wait ever gave up


def test_merged_launch_sees_parameter_updates_between_evaluations(gpu_device):
    """Every evaluation rewrites the factorisation state in place, and workgroups of the SAME launch read it on other CUs (other XCDs):
    a hand-off that served last evaluation's lines from a cache would go unnoticed while the parameters stand still.  Here Z, the
    lengthscales, q_mu, q_sqrt and the kernel variance move between evaluations (as under training); every evaluation must equal the two
    launches' bit for bit -- also at the full configs[2] grid, where the factorising workgroups resume chunks from snapshots."""
    from dgps_with_iwvi_amd import synthetic
    for B in (64, 1024):
        spec = synthetic.make_spec(L=2, M=128, B=B, K=20, R=5, with_lv=True, seed=17)
        zs = [torch.as_tensor(z, dtype=torch.float32, device=gpu_device) for z in synthetic.make_noise(spec, seed=18)]

        def run():
            m = _model(spec, gpu_device)
            gen = torch.Generator(device="cpu").manual_seed(5)
            out = []
            for it in range(6):
                out.append(m._elbo_parts(zs)[0].clone())
                out.append(m._elbo_parts(None)[1].clone())
                for l in m.layers:                                   # an in-place parameter step, the same in both runs
                    if not hasattr(l, "q_sqrt"):
                        for w in l.encoder.Ws:
                            w.mul_(1.0 + 0.01 * float(torch.randn((), generator=gen)))
                        continue
                    l.q_mu.add_(0.05 * torch.randn(l.q_mu.shape, generator=gen).to(gpu_device))
                    l.q_sqrt.mul_(1.0 + 0.02 * float(torch.randn((), generator=gen)))
                    l._Z().add_(0.02 * torch.randn(l._Z().shape, generator=gen).to(gpu_device))
                    k = l._base_kern()
                    k.lengthscales.mul_(1.0 + 0.02 * float(torch.randn((), generator=gen)))
                    k.variance = k.variance * (1.0 + 0.03 * float(torch.randn((), generator=gen)))
            return out

        a, b = _both(run)
        for i, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), (B, i, x.flatten()[:3], y.flatten()[:3])
        assert not torch.equal(a[0], a[2])                           # the parameters did move
