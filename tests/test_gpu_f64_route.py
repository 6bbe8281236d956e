"""The float64 stage-1 route (include/iwvi_hip.h: IWVI_LAYER_F64_STAGE1 / IWVI_GP_F64_STAGE1).

The reference computes everything in ``settings.float_type`` = float64 (temp_workaround.py:39,89; layers.py:61-62) and its own tests and
demo are 1-D.  With many inducing points in a 1-3-dimensional box K_uu is ill-conditioned (cond(Lm) ~ 1e4): a float32 k = K_uf alone moves
the conditional mean by ~5e-4, the float32 substitution a = Lm^-1 k by 1e-2 .. 1e-1.  A flagged layer forms K_uf, a and
sigma^2 - |a|^2 in float64 (v_mfma_f64_16x16x4_f64 against the dense float64 Lm^-1) and rounds a to float32 behind the solve.
Checked here against the float64 oracle at the tolerance written at each assert, for every solve form the float32 path has
(M <= 128 unrolled, 128 < M <= 240 column at a time, M > 240 super-blocks), padded M, Matern52, the full-covariance entry, and that
the 8-dimensional BASELINE stacks never take the route."""
import numpy as np
import pytest
import torch

from oracle import iwvi_oracle as O
from oracle.from_spec import build_oracle, oracle_noise

pytestmark = pytest.mark.gpu
F64_BIT = 1 << 12


def _t(a, dev):
    return torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)


def _np(t):
    return t.detach().double().cpu().numpy()


def _variant():
    from dgps_with_iwvi_amd import _abi
    return int(_abi.lib().iwvi_debug_last_forward_variant())


def _errors(spec, dev):
    from dgps_with_iwvi_amd import synthetic
    zs = synthetic.make_noise(spec, seed=1)
    zd = [_t(z, dev) for z in zs]
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    _, _, means_o, covs_o, _ = om.log_weights(oracle_noise(spec, zs))
    m = synthetic.build_model(spec, dev)
    elbo = float(m.compute_log_likelihood(zd))
    var_bits = _variant()
    fmean, fvar, _, _, _, means, covs = m._forward_iw(zd)
    dm = max(float(np.abs(_np(mm) - mo).max()) for mm, mo in zip(means[:-1], means_o[:-1])) if len(means) > 1 else 0.0
    dm = max(dm, float(np.abs(_np(fmean) - means_o[-1]).max()))
    vo = np.diagonal(covs_o[-1], axis1=-2, axis2=-1).transpose(0, 2, 1)
    dv = float(np.abs(_np(fvar) - vo).max())
    return dm, dv, abs(elbo - ref) / abs(ref), var_bits, m


@pytest.mark.parametrize("M,Dx", [(100, 1), (128, 1), (160, 1), (224, 1), (250, 1), (256, 1), (384, 1), (512, 1), (128, 2), (256, 3)])
def test_ill_conditioned_stack_matches_the_float64_oracle(gpu_device, M, Dx):
    """L = 2, K = 10, B = 16 on 1-3-dimensional inputs: per-layer mean 2e-5, final variance 1e-5, ELBO relative 2e-6 -- where the
    float32 route measures 1e-2 .. 8e-2 / 1e-3 / 3e-3 (profiles/r05_f64_route_error.txt).  'auto' picks the route by the input dimension."""
    from dgps_with_iwvi_amd import settings, synthetic
    assert settings.f64_stage1 == "auto"
    spec = synthetic.make_spec(seed=M, parity=True, n_data=4096, L=2, M=M, K=10, B=16, Dx=Dx, with_lv=False)
    dm, dv, de, bits, model = _errors(spec, gpu_device)
    assert bits & F64_BIT and not bits & (1 << 8), hex(bits)      # the F64 variants run stage 2 on fp32 MFMAs
    assert all(l.uses_f64_stage1() for l in model.layers)
    assert dm <= 2e-5 and dv <= 1e-5 and de <= 2e-6, (dm, dv, de)


def test_the_route_is_what_makes_the_difference(gpu_device):
    """The same stack with the route forced off: the float32 error it removes (so the tight asserts above are not vacuous)."""
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(seed=224, parity=True, n_data=4096, L=2, M=224, K=10, B=16, Dx=1, with_lv=False)
    with settings.temp_settings(f64_stage1="off"):
        dm32, _, de32, bits32, _ = _errors(spec, gpu_device)
    dm64, _, de64, bits64, _ = _errors(spec, gpu_device)
    assert not bits32 & F64_BIT and bits64 & F64_BIT
    assert dm32 > 1e-3 and dm64 < 2e-5 and dm32 > 100 * dm64, (dm32, dm64)
    assert de64 < de32


def test_baseline_stacks_do_not_take_the_route(gpu_device):
    """The headline (D = 8 / 9, well-conditioned) keeps its compiled-in-shapes split-f16 variant; a per-layer override flags ONE layer."""
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(seed=0, parity=True, n_data=4096, L=2, M=128, K=20, B=1024, with_lv=True)     # (the bench shape: T = 20480 = 256 chunks of 80)
    model = synthetic.build_model(spec, gpu_device)
    assert not any(getattr(l, "uses_f64_stage1", lambda: False)() for l in model.layers)
    a = model.compute_log_likelihood()
    bits = _variant()
    assert not bits & F64_BIT and bits & (1 << 8) and bits & (1 << 10), hex(bits)     # split-f16, LEAN
    zs = [_t(z, gpu_device) for z in synthetic.make_noise(spec, seed=5)]
    ref = float(model.compute_log_likelihood(zs))
    gp = [l for l in model.layers if hasattr(l, "uses_f64_stage1")]
    gp[0].f64_stage1 = True                                      # the inner layer alone in float64: same bound to float32 accuracy
    got = float(model.compute_log_likelihood(zs))
    assert _variant() & F64_BIT
    assert abs(got - ref) <= 2e-5 * abs(ref), (got, ref)
    gp[0].f64_stage1 = None
    assert np.isfinite(a)


@pytest.mark.parametrize("kern_name", ["RBF", "Matern52"])
@pytest.mark.parametrize("full_cov", [False, True])
def test_conditional_function_in_float64(gpu_device, kern_name, full_cov):
    """``multisample_sample_conditional`` (temp_workaround.py:118-161) on 1-D inputs, M = 100 on [0, 1] with lengthscale 0.1: mean, marginal
    variance / full [S, R, N, N] covariance against the oracle at 1e-5 of their scale (float32: 1e-3 .. 1e-2)."""
    from dgps_with_iwvi_amd import features, kernels
    from dgps_with_iwvi_amd.temp_workaround import multisample_sample_conditional
    rng = np.random.default_rng(3)
    M, R, S, N = 100, 2, 3, 17
    Z = np.linspace(0, 1, M).reshape(-1, 1).astype(np.float32)
    X = rng.uniform(0, 1, (S, N, 1)).astype(np.float32)
    q_mu = rng.standard_normal((M, R)).astype(np.float32)
    q_sqrt = np.tril(rng.standard_normal((R, M, M)) * 0.2).astype(np.float32)
    z = None if full_cov else np.zeros((S, N, R), np.float32)     # (full_cov: the joint sample is not fetched -- want_sample=False)
    ls = float(np.float32(0.1))
    ko = getattr(O, kern_name)(1, variance=1.3, lengthscales=ls)
    so, mo, vo = O.multisample_sample_conditional(X.astype(np.float64), Z.astype(np.float64), ko, q_mu.astype(np.float64),
                                                  full_cov=full_cov, q_sqrt=q_sqrt.astype(np.float64), white=True, z=z)
    kern = getattr(kernels, kern_name)(1, variance=1.3, lengthscales=0.1).to(gpu_device)
    feat = features.InducingPoints(Z).to(gpu_device)
    for mode, tol in (("auto", 1e-5), ("off", None)):
        s, m, v = multisample_sample_conditional(_t(X, gpu_device), feat, kern, _t(q_mu, gpu_device), full_cov=full_cov, q_sqrt=_t(q_sqrt, gpu_device),
                                                 white=True, z=None if z is None else _t(z, gpu_device), want_sample=not full_cov,
                                                 f64_stage1=None if mode == "auto" else False)
        em = np.abs(_np(m) - mo).max() / np.abs(mo).max()
        ev = np.abs(_np(v) - vo).max() / np.abs(vo).max()
        if tol is not None:
            assert bool(_variant() & F64_BIT)
            assert em <= tol and ev <= tol, (mode, em, ev)
        else:
            assert not _variant() & F64_BIT
            e32 = (em, ev)
    assert e32[0] > 1e-5 or e32[1] > 1e-5, e32                     # (what float32 loses on this case)


def test_autotune_reads_the_factor(gpu_device):
    """``DGP_VI.autotune_f64``: per-layer overrides from the measured max / min of diag(Lm) -- a well-conditioned 8-D layer stays float32,
    a 1-D layer with 200 inducing points is flagged, whatever the static rule says."""
    from dgps_with_iwvi_amd import settings, synthetic
    spec = synthetic.make_spec(seed=7, parity=True, n_data=4096, L=2, M=128, K=4, B=16, with_lv=False)
    model = synthetic.build_model(spec, gpu_device)
    rep = model.autotune_f64()
    assert [r["f64_stage1"] for r in rep] == [False, False] and all(r["diag_ratio"] < 100 for r in rep), rep
    spec1 = synthetic.make_spec(seed=8, parity=True, n_data=4096, L=2, M=200, K=4, B=16, Dx=1, with_lv=False)
    with settings.temp_settings(f64_stage1="off"):
        model1 = synthetic.build_model(spec1, gpu_device)
        rep1 = model1.autotune_f64()
        assert all(r["f64_stage1"] for r in rep1) and all(r["diag_ratio"] >= 300 for r in rep1), rep1
        model1.compute_log_likelihood()
        assert _variant() & F64_BIT                              # the per-layer override wins over settings "off"


def _model_errors(model, spec, dev, zs):
    """(max |d mean| over layers, |d ELBO| / |ELBO|) of ``model`` against the float64 oracle on the noise ``zs``."""
    zd = [_t(z, dev) for z in zs]
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    _, _, means_o, _, _ = om.log_weights(oracle_noise(spec, zs))
    elbo = float(model.compute_log_likelihood(zd))
    fmean, _, _, _, _, means, _ = model._forward_iw(zd)
    dm = max([float(np.abs(_np(mm) - mo).max()) for mm, mo in zip(means[:-1], means_o[:-1])] + [float(np.abs(_np(fmean) - means_o[-1]).max())])
    return dm, abs(elbo - ref) / abs(ref)


def test_trainer_measures_the_route_of_a_clustered_8d_stack(gpu_device):
    """the static rule keys on the input dimension alone.  An 8-D stack whose inducing inputs are CLUSTERED (16 centres,
    8 near-copies each -- what k-means initialisation or training can produce) has an ill-conditioned K_uu: the rule says float32, the
    measured diag(Lm) ratio says float64.  ``training.Trainer`` (and ``build_models.build_model``) run ``autotune_f64`` at construction and at
    every staircase epoch: the layers move to the float64 stage-1 route and the stack holds the stated tolerance (mean rtol 2e-3 + atol 1e-3
    -> checked here as max |d mean| <= 1e-3 on |mean| ~ 1; ELBO 1e-4 relative) which the float32 solve does not."""
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(seed=11, parity=True, n_data=4096, L=2, M=128, K=10, B=16, with_lv=False)
    rng = np.random.default_rng(5)
    for l in spec["layers"]:
        Z = l["Z"]
        l["Z"] = (np.repeat(Z[:16], 8, axis=0) + 1e-3 * rng.standard_normal(Z.shape)).astype(np.float32).astype(np.float64)
    zs = synthetic.make_noise(spec, seed=1)
    model = synthetic.build_model(spec, gpu_device)
    assert not any(l.uses_f64_stage1() for l in model.layers)            # the static rule: 8-D -> float32
    dm32, de32 = _model_errors(model, spec, gpu_device, zs)
    assert not _variant() & F64_BIT
    tr = Trainer(model, use_graph=True)
    (step0, rep), = tr.route_reports
    assert step0 == 0 and all(r["f64_stage1"] and r["diag_ratio"] >= 300 for r in rep), rep
    assert all(l.uses_f64_stage1() for l in model.layers)
    dm64, de64 = _model_errors(model, spec, gpu_device, zs)
    assert _variant() & F64_BIT
    assert dm64 <= 1e-3 and de64 <= 1e-4, (dm64, de64)
    assert dm32 > 3 * dm64, (dm32, dm64)                                  # what the measured route buys on this stack
    # an explicit per-layer choice still wins over the measurement; a moved route drops the captured graphs
    e0 = tr.step()
    key0 = model.route_key()
    assert tr._graphs and all(k[0][1] == key0 for k in [(v[0],) for v in tr._graphs.values()])
    model.layers[0].f64_stage1 = False
    assert model.route_key() != key0
    e1 = tr.step()                                                       # re-captured with the new route, not replayed with the old flags
    assert all(v[0][1] == model.route_key() for v in tr._graphs.values())
    assert np.isfinite(float(e0)) and np.isfinite(float(e1))
