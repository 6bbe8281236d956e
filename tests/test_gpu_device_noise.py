"""The noise mode bench.py times: N(0,1) draws made INSIDE ``iwvi_dgp_forward`` (Philox4x32-10 counters
(sample, layer * 256 + component group, step), Box-Muller on the hardware log2 / sqrt / sin / cos) instead of injected arrays.

(a) parity: the draws a launch used are read back (``noise_out`` of every layer) and fed to the float64 oracle -- the same ELBO /
    per-point / per-layer tolerances as the injected-noise suite (tests/test_gpu_parity.py), so a mis-keyed stream cannot hide behind
    determinism tests;
(b) the draws themselves at the full configs[2] size: moments per layer, independence between latent GPs, between layers, between
    consecutive evaluations and between two ranks' keys (bench.py: seed + 7919 * rank);
(c) the statistical intent of the reference's disabled tests (tests/test_latent_var_layer.py:166-241, tf.random_normal in-graph):
    E[IW_10] > E[VI_10]; for K = 1 the two estimators share their expectation and the VI one (analytic KL) has the smaller spread."""
import numpy as np
import pytest
import torch

from oracle.from_spec import build_oracle, oracle_noise

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().double().cpu().numpy()


def _forward_with_readback(model, spec):
    """One IW-ELBO evaluation on device noise -> (elbo, logp [B], per-layer outputs incl. the draws used)."""
    B, K = spec["B"], spec["K"]
    model.precompute(with_encoders=True)
    _, outs, red = model._fused_forward(B * K, K, B, (B, K), zs=None, sampled_kl=True, want_layers=True, want_saved=True,
                                        elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
    torch.cuda.synchronize()
    return red[0], red[1], outs


# (the prologue of the launch splits its duties by how many waves the noise items fill -- csrc/dgp_forward.hip, n_early: 5 of 8 waves at
#  configs[2] -> three waves issue the big copies first thing; R = 2 without an LV layer: 2 waves draw, the cap of four early waves; R = 13 in two
#  inner layers: more items than threads, every wave draws and copies, the second round of draws)
@pytest.mark.parametrize("L,M,K,B,lv,R", [(2, 128, 20, 48, True, 5),     # configs[2]'s stack at a batch the oracle finishes in seconds
                                          (2, 128, 5, 64, False, 5),     # configs[1]'s
                                          (3, 64, 7, 33, True, 5),       # ragged chunk, three GP layers
                                          (2, 32, 10, 24, False, 2),
                                          (3, 32, 5, 32, True, 13)])
def test_device_drawn_noise_read_back_matches_oracle(gpu_device, L, M, K, B, lv, R):
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, with_lv=lv, R=R, seed=L * 10 + K)
    settings.set_seed(1234)
    model = synthetic.build_model(spec, gpu_device)
    elbo, logp, outs = _forward_with_readback(model, spec)
    zs = [_np(o["noise_out"]).reshape(B, K, -1) for o in outs]
    for z in zs[:-1]:                                             # every consumed stream looks like N(0,1) (the last layer's is never used)
        assert abs(z.mean()) < 5.0 / np.sqrt(z.size) and abs(z.std() - 1.0) < 5.0 / np.sqrt(2 * z.size)
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    assert abs(float(elbo) - ref) <= 1e-4 * abs(ref), (float(elbo), ref)                      # ELBO relative 1e-4
    L_NK, _, means_o, covs_o, samples_o = om.log_weights(oracle_noise(spec, zs))
    m_o = L_NK.max(1)
    np.testing.assert_allclose(_np(logp), m_o + np.log(np.exp(L_NK - m_o[:, None]).sum(1)) - np.log(K), rtol=2e-4, atol=2e-2)
    for i in range(len(spec["layers"]) - 1):                                                    # per-layer mean rtol 2e-3 + atol 1e-3
        np.testing.assert_allclose(_np(outs[i]["mean"]), means_o[i], rtol=2e-3, atol=1e-3)
        np.testing.assert_allclose(_np(outs[i]["sample"]), samples_o[i], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(_np(outs[-1]["mean"]), means_o[-1], rtol=2e-3, atol=2e-3)
    vo = np.diagonal(covs_o[-1], axis1=-2, axis2=-1).transpose(0, 2, 1)
    np.testing.assert_allclose(_np(outs[-1]["var"]), vo, rtol=5e-3, atol=2e-4)
    # the next evaluation draws other numbers (the device-resident step counter moved on)
    _, _, outs2 = _forward_with_readback(model, spec)
    assert float((outs2[0]["noise_out"] - outs[0]["noise_out"]).abs().max()) > 0.1


def _streams(gpu_device, seed, n_eval=2):
    """configs[2] at full size: the draws of `n_eval` consecutive evaluations, as [evaluation][layer] -> [T, dims] arrays."""
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=1024, K=20, with_lv=True, seed=0, n_data=65536)
    settings.set_seed(seed)
    model = synthetic.build_model(spec, gpu_device)
    return [[_np(o["noise_out"]) for o in _forward_with_readback(model, spec)[2]] for _ in range(n_eval)]


def test_in_kernel_draws_moments_and_independence_at_full_size(gpu_device):
    from dgps_with_iwvi_amd import settings
    base = settings.seed
    try:
        ev = _streams(gpu_device, 77)                             # rank 0's key
        other = _streams(gpu_device, 77 + 7919, n_eval=1)         # rank 1's key (bench.py: seed + 7919 * rank)
    finally:
        settings.set_seed(base)
    n = ev[0][0].shape[0]                                         # T = 20480 samples
    tol = 5.0 / np.sqrt(n)

    def corr(a, b):
        return abs(float(np.corrcoef(a, b)[0, 1]))

    cols = []                                                     # (evaluation, layer, component) -> the column
    for e, layers in enumerate(ev):
        for li, z in enumerate(layers):
            for r in range(z.shape[1]):
                c = z[:, r]
                cols.append((e, li, r, c))
                assert abs(c.mean()) < tol, (e, li, r, c.mean())
                assert abs(c.var() - 1.0) < 5.0 * np.sqrt(2.0 / n), (e, li, r, c.var())
                kurt = float(((c - c.mean()) ** 4).mean() / c.var() ** 2)
                assert abs(kurt - 3.0) < 5.0 * np.sqrt(24.0 / n), (e, li, r, kurt)
                assert abs(float((c ** 3).mean())) < 5.0 * np.sqrt(15.0 / n)
    # every pair of columns -- latent GPs of one layer, different layers, consecutive evaluations -- is uncorrelated ...
    for i in range(len(cols)):
        for j in range(i + 1, len(cols)):
            assert corr(cols[i][3], cols[j][3]) < tol, (cols[i][:3], cols[j][:3])
    # ... also against the other rank's key, and shifted by one sample (a counter off by one would show here)
    for (e, li, r, c) in cols[:7]:
        assert corr(c, other[0][li][:, r]) < tol
        assert corr(c[1:], c[:-1]) < tol
        assert not np.array_equal(c, other[0][li][:, r])
    # no two columns are the same stream re-used
    sig = {tuple(np.round(c[:8], 6)) for (_, _, _, c) in cols}
    assert len(sig) == len(cols)


def _lv_gp_models(gpu_device, K, seed=5):
    """[LatentVariableLayer, GPLayer] as DGP_VI and DGP_IWVI with the same parameters (the reference's test_IW_vs_VI models)."""
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.models import DGP_IWVI, DGP_VI
    spec = synthetic.make_spec(L=1, M=32, B=64, K=K, Dx=3, with_lv=True, seed=seed, n_data=64)
    return synthetic.build_model(spec, gpu_device, cls=DGP_VI), synthetic.build_model(spec, gpu_device, cls=DGP_IWVI)


def _estimates(model, n):
    out = torch.empty(n, dtype=torch.float64, device=model.X.device)
    for i in range(n):
        out[i] = model._build_likelihood(None)
    return _np(out)


def test_iw_bound_above_vi_bound_on_device_noise(gpu_device):
    """reference tests/test_latent_var_layer.py:166-200 (disabled there): for K > 1 the IW estimate is greater than the VI estimate on
    average -- here over 300 evaluations each, on noise drawn inside the kernels."""
    from dgps_with_iwvi_amd import settings
    settings.set_seed(11)
    m_vi, m_iw = _lv_gp_models(gpu_device, 10)
    L_vi, L_iw = _estimates(m_vi, 300), _estimates(m_iw, 300)
    se = L_vi.std() / np.sqrt(len(L_vi)) + L_iw.std() / np.sqrt(len(L_iw))
    assert L_iw.mean() > L_vi.mean() + 3 * se, (L_iw.mean(), L_vi.mean(), se)


def test_single_sample_iw_and_vi_agree_in_expectation_on_device_noise(gpu_device):
    """reference :203-241 (disabled there): K = 1 -- the IW estimator is the VI one with the sampled log q / p in place of the analytic
    KL: same expectation (within 4 standard errors over 1500 evaluations), and the VI estimator has the smaller spread."""
    from dgps_with_iwvi_amd import settings
    settings.set_seed(12)
    m_vi, m_iw = _lv_gp_models(gpu_device, 1)
    L_vi, L_iw = _estimates(m_vi, 1500), _estimates(m_iw, 1500)
    se = L_vi.std() / np.sqrt(len(L_vi)) + L_iw.std() / np.sqrt(len(L_iw))
    assert abs(L_iw.mean() - L_vi.mean()) < 4 * se, (L_iw.mean(), L_vi.mean(), se)
    assert L_vi.std() < L_iw.std()
