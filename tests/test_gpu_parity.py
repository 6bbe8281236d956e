"""GPU parity: every entry point of the C-ABI (through dgps_with_iwvi_amd) against the fp64 oracle
on identical seeded inputs and injected noise.

Stated float32 tolerances (BASELINE.md section 5 / SURVEY.md section 7.2, D=8 family):
per-layer conditional mean rtol 2e-3 + atol 1e-3, variance atol 1e-4 (+ rtol 2e-3),
ELBO relative 1e-4 (the ELBO is O(1e4) here, so this is the tightest check).
"""
import numpy as np
import pytest
import torch

from oracle import iwvi_oracle as O
from oracle import svgp_closed_form as C
from oracle.from_spec import build_oracle, oracle_noise

pytestmark = pytest.mark.gpu

MEAN_TOL = dict(rtol=2e-3, atol=1e-3)
VAR_TOL = dict(rtol=2e-3, atol=1e-4)
ELBO_RTOL = 1e-4


def _t(a, dev, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a), dtype=dtype, device=dev)


def _np(t):
    return t.detach().double().cpu().numpy()


# ------------------------------------------------------------------------------------------
# K1 / K2 / precompute
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,D,R,kern", [(10, 1, 2, "matern"), (64, 8, 1, "rbf"), (100, 3, 2, "rbf"),
                                        (128, 9, 5, "rbf"), (200, 8, 2, "rbf"), (256, 8, 3, "rbf"),
                                        (128, 1, 1, "rbf"),      # dense 1-D cloud: K_uu is numerically rank-deficient, jitter decides
                                        (512, 8, 1, "rbf")])     # the largest supported inducing set
def test_precompute_factorisation(gpu_device, M, D, R, kern):
    """Kuu + tf.cholesky (temp_workaround.py:39,48), Lm^-1 and gauss_kl (:186-188) vs NumPy fp64."""
    from dgps_with_iwvi_amd import kernels, settings
    from dgps_with_iwvi_amd.temp_workaround import GpState, precompute_states
    rng = np.random.default_rng(M + D)
    Z = rng.standard_normal((M, D)).astype(np.float32)
    ls = ((0.7 + rng.random(D)) * np.sqrt(D)).astype(np.float32)
    q_mu = rng.standard_normal((M, R)).astype(np.float32)
    q_sqrt = (rng.standard_normal((R, M, M)) * 0.2 + np.eye(M)).astype(np.float32)
    kcls, ocls = (kernels.Matern52, O.Matern52) if kern == "matern" else (kernels.RBF, O.RBF)
    k = kcls(D, variance=1.3, lengthscales=ls).to(gpu_device)
    st = GpState(M, R, gpu_device)
    precompute_states([st.desc(_t(Z, gpu_device), k, _t(q_mu, gpu_device), _t(q_sqrt, gpu_device),
                               settings.jitter_level)])
    torch.cuda.synchronize()
    # the device factorises the Gram of the float32-rounded, centred scaled inputs -- the values K_uf also
    # sees (DESIGN.md "Precision"): Zs = fl32(Z / ls), zc = fl32(mean_m Zs), Zs <- fl32(Zs - zc)
    Zs32 = (Z.astype(np.float64) / ls.astype(np.float64)).astype(np.float32)
    zc = Zs32.astype(np.float64).mean(0).astype(np.float32)
    Zs = (Zs32 - zc).astype(np.float32).astype(np.float64)
    ok = ocls(D, variance=float(np.float32(1.3)), lengthscales=1.0)   # the ABI takes a float32 variance
    Kuu = ok.K(Zs) + 1e-6 * np.eye(M)
    # cond(Kuu) reaches 1e6..1e8 here, so two float64 factorisations agree entry-wise only to
    # ~cond * 1e-16; the backward-stable checks are the residuals
    Lm, Linv = _np(st.Lm), _np(st.Linv)
    assert np.all(np.triu(Lm, 1) == 0) and np.all(np.triu(Linv, 1) == 0)
    np.testing.assert_allclose(Lm @ Lm.T, Kuu, rtol=0, atol=1e-12)
    np.testing.assert_allclose(Lm, np.linalg.cholesky(Kuu), rtol=0, atol=1e-6)
    np.testing.assert_allclose(Linv @ Lm, np.eye(M), atol=1e-7)
    np.testing.assert_allclose(float(st.kl.item()), O.gauss_kl(q_mu, q_sqrt), rtol=1e-6)


def test_gram_and_cholesky_entry_points(gpu_device):
    from dgps_with_iwvi_amd import _abi, kernels
    rng = np.random.default_rng(0)
    M, D = 100, 5
    Z = rng.standard_normal((M, D)).astype(np.float32)
    ls = np.full(D, np.sqrt(D), np.float32)
    k = kernels.RBF(D, variance=0.9, lengthscales=ls).to(gpu_device)
    K = _np(k.K(_t(Z, gpu_device)))
    Zs = (Z.astype(np.float64) / ls).astype(np.float32).astype(np.float64)
    np.testing.assert_allclose(K, O.RBF(D, float(np.float32(0.9)), 1.0).K(Zs), rtol=1e-12, atol=1e-14)
    A = K + 1e-6 * np.eye(M)
    Ad = _t(A, gpu_device, torch.float64)
    L = torch.empty_like(Ad)
    ws = torch.empty(_abi.lib().iwvi_chol_ws_bytes(M), dtype=torch.uint8, device=gpu_device)
    _abi.check(_abi.lib().iwvi_chol_factor(_abi.ptr(Ad), _abi.ptr(L), M, _abi.ptr(ws), _abi.stream_ptr()))
    Ln = _np(L)
    np.testing.assert_allclose(Ln @ Ln.T, A, rtol=0, atol=1e-12)
    np.testing.assert_allclose(Ln, np.linalg.cholesky(A), rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------
# GP layer forward (temp_workaround.py:12-98, 118-161; layers.py:35-50)
# ------------------------------------------------------------------------------------------
def _layer_case(seed, M, D, R, P, mixing, mf, S, N):
    rng = np.random.default_rng(seed)
    Z = rng.standard_normal((M, D)).astype(np.float32)
    ls = ((0.8 + 0.4 * rng.random(D)) * np.sqrt(D)).astype(np.float32)
    q_mu = rng.standard_normal((M, R)).astype(np.float32)
    q_sqrt = (np.tril(rng.standard_normal((R, M, M))) * 0.1 / np.sqrt(M) + 0.5 * np.eye(M)).astype(np.float32)
    W = rng.standard_normal((P, R)).astype(np.float32) if mixing else None
    A = rng.standard_normal((D, P)).astype(np.float32) if mf == "linear" else None
    b = rng.standard_normal(P).astype(np.float32) if mf == "linear" else None
    X = rng.standard_normal((S, N, D)).astype(np.float32)
    z = rng.standard_normal((S, N, R)).astype(np.float32)
    return dict(Z=Z, ls=ls, q_mu=q_mu, q_sqrt=q_sqrt, W=W, A=A, b=b, X=X, z=z)


def _run_layer(c, dev, D, R, mixing, mf, full_cov=False, z=None):
    from dgps_with_iwvi_amd import features, kernels, mean_functions
    from dgps_with_iwvi_amd.layers import GPLayer
    from dgps_with_iwvi_amd.temp_workaround import SharedMixedMok
    kern = kernels.RBF(D, variance=c.get("variance", 1.1), lengthscales=c["ls"])
    feat = features.InducingPoints(c["Z"])
    mfo = {"linear": lambda: mean_functions.Linear(c["A"], c["b"]), "identity": mean_functions.Identity,
           "zero": lambda: None}[mf]()
    if mixing:
        layer = GPLayer(SharedMixedMok(kern, c["W"]), features.MixedKernelSharedMof(feat), R, mfo)
    else:
        layer = GPLayer(kern, feat, R, mfo)
    layer.to(dev)
    layer.q_mu, layer.q_sqrt = _t(c["q_mu"], dev), _t(c["q_sqrt"], dev)
    return layer.propagate(_t(c["X"], dev), full_cov=full_cov, z=z)


def _oracle_layer(c, D, R, mixing, mf, full_cov=False, z=None):
    kern = O.RBF(D, variance=c.get("variance", 1.1), lengthscales=c["ls"])
    mfo = {"linear": lambda: O.Linear(c["A"], c["b"]), "identity": O.Identity, "zero": lambda: None}[mf]()
    layer = O.GPLayer(O.SharedMixedMok(kern, c["W"]) if mixing else kern, c["Z"], R, mfo)
    layer.q_mu, layer.q_sqrt = c["q_mu"], c["q_sqrt"]
    return layer.propagate(c["X"], full_cov=full_cov, z=z)


@pytest.mark.parametrize("M,D,R,P,mixing,mf,S,N", [
    (128, 9, 5, 8, True, "linear", 7, 20),      # headline inner layer (L1_G5): D_in 9 -> 8
    (128, 8, 1, 1, False, "zero", 11, 20),      # headline final layer
    (64, 1, 1, 1, False, "linear", 1, 33),      # config-1-like, ragged tile
    (100, 3, 2, 2, False, "zero", 3, 5),        # M not a multiple of 32
    (32, 4, 3, 4, True, "identity", 2, 1),      # single-sample rows, identity mean function
    (256, 8, 5, 8, True, "linear", 4, 50),      # config 4 layer
    (512, 8, 2, 8, True, "linear", 2, 40),      # config 5 width (one workgroup per CU)
    (40, 17, 2, 2, False, "zero", 3, 7),        # D > 16 instantiation
    (250, 8, 2, 2, False, "zero", 3, 40),       # super-block solve (M > 240) with padded inducing rows, split-f16 dense part
    (270, 5, 1, 1, False, "linear", 2, 30),     # 17 block rows: odd, so the fp32 variant; a one-row last super-block
    (384, 8, 3, 8, True, "linear", 2, 24),      # three super-blocks
])
def test_gp_layer_forward(gpu_device, M, D, R, P, mixing, mf, S, N):
    c = _layer_case(M + D + R, M, D, R, P, mixing, mf, S, N)
    s, m, v, kl = _run_layer(c, gpu_device, D, R, mixing, mf, z=_t(c["z"], gpu_device))
    so, mo, vo, klo = _oracle_layer(c, D, R, mixing, mf, z=c["z"])
    assert s.shape == so.shape and m.shape == mo.shape and v.shape == vo.shape
    # D = 1 makes cond(Kuu) ~ 1e8: such layers take the float64 stage-1 route by themselves (settings.f64_stage1 = "auto": input
    # dimension <= 3; until round 5 this row needed 10 x the stated tolerance), so every row holds the stated float32 tolerance
    np.testing.assert_allclose(_np(m), mo, rtol=MEAN_TOL["rtol"], atol=MEAN_TOL["atol"])
    np.testing.assert_allclose(_np(v), vo, rtol=VAR_TOL["rtol"], atol=VAR_TOL["atol"])
    np.testing.assert_allclose(_np(s), so, rtol=2e-3, atol=2e-3)
    if D <= 3:                                                   # ... and the float64 route much more than that
        np.testing.assert_allclose(_np(m), mo, rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(_np(v), vo, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(float(kl.item()), klo, rtol=1e-6)
    # noise-injected sample identity: sample == mean + W (z * sqrt(var_g)) is implied by the above;
    # with z = 0 the sample is exactly the mean
    s0, m0, _, _ = _run_layer(c, gpu_device, D, R, mixing, mf, z=torch.zeros(S, N, R, device=gpu_device))
    assert torch.equal(s0, m0)


@pytest.mark.parametrize("variance", [1e4, 1e-4, 37.0])
def test_super_block_solve_at_extreme_kernel_variances(gpu_device, variance):
    """M > 240, split-f16 operands: the scales of the packed inverse blocks and of the published right-hand sides are powers of two fixed
    by the kernel variance; their entries are bounded by sigma / sqrt(jitter), so beyond variance / jitter = 2^30 (here: 1e4 at jitter 1e-6)
    the layer API asks for the fp32 variant (settings.split16_variance_ok).  The tolerances of test_gp_layer_forward hold relative to the
    layer's scale at variances far from 1, on whichever variant runs."""
    M, D, R, S, N = 256, 8, 2, 3, 40
    c = dict(_layer_case(77, M, D, R, R, False, "zero", S, N), variance=variance)
    s, m, v, kl = _run_layer(c, gpu_device, D, R, False, "zero", z=_t(c["z"], gpu_device))
    so, mo, vo, klo = _oracle_layer(c, D, R, False, "zero", z=c["z"])
    sd = float(np.sqrt(variance))
    np.testing.assert_allclose(_np(m), mo, rtol=MEAN_TOL["rtol"], atol=MEAN_TOL["atol"] * sd)
    np.testing.assert_allclose(_np(v), vo, rtol=VAR_TOL["rtol"], atol=VAR_TOL["atol"] * variance)
    np.testing.assert_allclose(_np(s), so, rtol=2e-3, atol=2e-3 * sd)
    assert bool(torch.isfinite(m).all()) and bool(torch.isfinite(v).all())
    np.testing.assert_allclose(float(kl.item()), klo, rtol=1e-6)
    from dgps_with_iwvi_amd import _abi
    took_split16 = bool(_abi.lib().iwvi_debug_last_forward_variant() >> 8 & 1)
    assert took_split16 == (variance < 1e3)


def test_super_block_operands_from_a_dense_state(gpu_device):
    """M > 240: the split-f16 slabs of the solve's dense part are packed from the factorisation's workspace -- or, for a state
    precomputed with IWVI_GP_WANT_DENSE (whose workspace then holds the full inverse), from its dense Lm: the same numbers."""
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=2, M=256, B=40, K=6, with_lv=False, seed=5)
    zs = [_t(z, gpu_device).reshape(40 * 6, -1) for z in synthetic.make_noise(spec, seed=6)]
    model = synthetic.build_model(spec, gpu_device)
    out = []
    for dense in (False, True):
        model.precompute(with_encoders=True, dense=dense)
        logw, _, _ = model._fused_forward(40 * 6, 6, 40, (40, 6), zs=zs)
        out.append(logw.clone())
    assert torch.equal(out[0], out[1])


def test_gp_layer_2d_equals_3d_flat(gpu_device):
    """[S,N,D] path == the 2-D path on the reshaped input."""
    c = _layer_case(5, 64, 4, 2, 2, False, "zero", 6, 9)
    z = _t(c["z"], gpu_device)
    s3, m3, v3, _ = _run_layer(c, gpu_device, 4, 2, False, "zero", z=z)
    c2 = dict(c, X=c["X"].reshape(54, 4))
    s2, m2, v2, _ = _run_layer(c2, gpu_device, 4, 2, False, "zero", z=z.reshape(54, 2))
    assert torch.equal(s3.reshape(54, 2), s2) and torch.equal(m3.reshape(54, 2), m2)
    assert torch.equal(v3.reshape(54, 2), v2)


def test_gp_layer_bit_reproducible(gpu_device):
    c = _layer_case(9, 128, 8, 5, 8, True, "linear", 40, 20)
    z = _t(c["z"], gpu_device)
    a = _run_layer(c, gpu_device, 8, 5, True, "linear", z=z)
    b = _run_layer(c, gpu_device, 8, 5, True, "linear", z=z)
    for x, y in zip(a[:3], b[:3]):
        assert torch.equal(x, y)


@pytest.mark.parametrize("M,D,R,S,N", [(64, 3, 2, 4, 7), (128, 8, 1, 5, 20)])
def test_gp_layer_full_cov(gpu_device, M, D, R, S, N):
    """full covariance over the K axis (temp_workaround.py:45,56,83) and its diagonal."""
    c = _layer_case(M + 1, M, D, R, R, False, "zero", S, N)
    zf = np.random.default_rng(1).standard_normal((S, R, N, 1)).astype(np.float32)
    s, m, cov, _ = _run_layer(c, gpu_device, D, R, False, "zero", full_cov=True, z=_t(zf, gpu_device))
    so, mo, covo, _ = _oracle_layer(c, D, R, False, "zero", full_cov=True, z=zf)
    assert cov.shape == (S, R, N, N)
    np.testing.assert_allclose(_np(m), mo, **MEAN_TOL)
    np.testing.assert_allclose(_np(cov), covo, rtol=2e-3, atol=1e-4)
    np.testing.assert_allclose(_np(s), so, rtol=5e-3, atol=5e-3)
    _, _, vd, _ = _run_layer(c, gpu_device, D, R, False, "zero", z=torch.zeros(S, N, R, device=gpu_device))
    np.testing.assert_allclose(_np(torch.diagonal(cov, dim1=-2, dim2=-1).transpose(1, 2)), _np(vd),
                               rtol=1e-4, atol=2e-5)


def test_conditional_argument_errors(gpu_device):
    """error behaviour of the reference: NotImplementedError for full_output_cov (:36-37),
    ValueError for a bad q_sqrt rank (:80-81)."""
    from dgps_with_iwvi_amd import features, kernels
    from dgps_with_iwvi_amd.temp_workaround import independent_multisample_sample_conditional as cond
    k = kernels.RBF(2).to(gpu_device)
    feat = features.InducingPoints(np.zeros((4, 2), np.float32) + np.arange(4)[:, None]).to(gpu_device)
    X = torch.zeros(2, 3, 2, device=gpu_device)
    f = torch.zeros(4, 1, device=gpu_device)
    with pytest.raises(NotImplementedError):
        cond(X, feat, k, f, full_output_cov=True, white=True)
    with pytest.raises(ValueError):
        cond(X, feat, k, f, q_sqrt=torch.zeros(1, 1, 4, 4, device=gpu_device), white=True)
    with pytest.raises(ValueError):                              # a precomputed state holds whitened operands
        cond(X, feat, k, f, white=False, precomputed=True)
    cond(X, feat, k, f, white=False)                             # white=False itself is implemented (:63-65)
    # diagonal q_sqrt [M, R] (:72-73) and q_sqrt None are accepted
    s, m, v = cond(X, feat, k, f, q_sqrt=torch.ones(4, 1, device=gpu_device), white=True,
                   z=torch.zeros(2, 3, 1, device=gpu_device))
    assert m.shape == (2, 3, 1) and torch.isfinite(v).all()
    cond(X, feat, k, f, q_sqrt=None, white=True)
    # empty batch
    s, m, v = cond(torch.zeros(0, 3, 2, device=gpu_device), feat, k, f, white=True)
    assert s.shape == (0, 3, 1)


# ------------------------------------------------------------------------------------------
# LatentVariableLayer + Encoder (layers.py:72-152)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sampled", [True, False])
def test_lv_layer(gpu_device, sampled):
    from dgps_with_iwvi_amd.layers import Encoder, LatentVariableLayer
    rng = np.random.default_rng(2)
    B, K, D, XYd, Lw = 5, 7, 8, 9, 2
    enc_o = O.Encoder(Lw, XYd, [20, 20], rng)
    enc_o.bs = [rng.standard_normal(b.shape) * 0.2 for b in enc_o.bs]
    enc = Encoder(Lw, XYd, [20, 20]).to(gpu_device)
    enc.Ws = [_t(w, gpu_device) for w in enc_o.Ws]
    enc.bs = [_t(b, gpu_device) for b in enc_o.bs]
    F = rng.standard_normal((B, K, D)).astype(np.float32)
    XY = rng.standard_normal((B, K, XYd)).astype(np.float32)
    z = rng.standard_normal((B, K, Lw)).astype(np.float32)
    lv = LatentVariableLayer(Lw, encoder=enc)
    s, m, c, kl = lv.propagate(_t(F, gpu_device), _t(XY, gpu_device), sampled, z=_t(z, gpu_device))
    so, mo, co, klo = O.LatentVariableLayer(Lw, encoder=enc_o).propagate(F, XY, sampled, z=z)
    for a, b_ in ((s, so), (m, mo), (c, co), (kl, klo)):
        np.testing.assert_allclose(_np(a), b_, rtol=1e-4, atol=1e-5)
    qm, qs = enc(_t(XY, gpu_device))
    qmo, qso = enc_o(XY)
    np.testing.assert_allclose(_np(qm), qmo, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(_np(qs), qso, rtol=1e-4, atol=1e-6)
    # prior mode (layers.py:73-81): W = z, log q/p = 0
    sp, mp, cp, klp = lv.propagate(_t(F, gpu_device), None, True, z=_t(z, gpu_device))
    np.testing.assert_allclose(_np(sp[..., D:]), z, rtol=1e-6)
    assert float(klp.abs().max()) < 1e-6 and float((cp[..., D:] - 1).abs().max()) == 0.0


# ------------------------------------------------------------------------------------------
# models: IW-ELBO (models.py:112-150), VI ELBO (:49-86)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("L,M,K,B,lv", [(1, 64, 1, 40, False), (2, 128, 5, 64, False), (2, 128, 20, 48, True),
                                        (3, 64, 7, 33, True), (2, 32, 70, 5, True)])
def test_iw_elbo_matches_oracle(gpu_device, L, M, K, B, lv):
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, with_lv=lv, seed=L * 100 + K)
    zs = synthetic.make_noise(spec, seed=5)
    model = synthetic.build_model(spec, gpu_device)
    zd = [_t(z, gpu_device) for z in zs]
    elbo = model.compute_log_likelihood(zd)
    om = build_oracle(spec)
    ref = om.build_likelihood(oracle_noise(spec, zs))
    assert abs(elbo - ref) <= ELBO_RTOL * abs(ref), (elbo, ref)
    # per-point log-weights and per-layer outputs
    L_NK, _, means_o, covs_o, samples_o = om.log_weights(oracle_noise(spec, zs))
    m_o = L_NK.max(1)
    logp_o = m_o + np.log(np.exp(L_NK - m_o[:, None]).sum(1)) - np.log(K)
    np.testing.assert_allclose(_np(model.E_log_p_Y(zd)), logp_o, rtol=2e-4, atol=2e-2)
    fmean, fvar, _, _, samples, means, covs = model._forward_iw(zd)
    for i in range(len(spec["layers"]) - 1):
        np.testing.assert_allclose(_np(means[i]), means_o[i], **MEAN_TOL)
        np.testing.assert_allclose(_np(samples[i]), samples_o[i], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(_np(fmean), means_o[-1], rtol=2e-3, atol=2e-3)
    vo = np.diagonal(covs_o[-1], axis1=-2, axis2=-1).transpose(0, 2, 1)
    np.testing.assert_allclose(_np(fvar), vo, rtol=5e-3, atol=2e-4)


@pytest.mark.parametrize("L,M,K,B", [(2, 128, 20, 48), (3, 64, 7, 33), (2, 32, 70, 5), (2, 128, 20, 1024)])
def test_lv_layer_in_the_precompute_launch(gpu_device, L, M, K, B):
    """``lv_in_precompute``: the leading latent-variable layer (layers.py:83-103) evaluated by the precompute
    launch and handed to the layer kernel as [T, Dx+Lw] rows + per-sample regulariser.  (1) against the oracle on
    the draws it exported; (2) same Philox streams as the in-kernel layer -> the same estimate from both routes."""
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, with_lv=True, seed=L * 10 + K)
    zs = synthetic.make_noise(spec, seed=9)
    model = synthetic.build_model(spec, gpu_device)
    model.lv_in_precompute, model.keep_lv_noise = True, True
    zd = [None] + [_t(z, gpu_device) for z in zs[1:]]
    logp = model.E_log_p_Y(zd)
    z0 = _np(model.layers[0]._smp_z).reshape(B, K, -1)
    assert abs(z0.mean()) < 5.0 / np.sqrt(z0.size) and abs(z0.std() - 1) < 5.0 / np.sqrt(2 * z0.size)       # five sigma of each statistic
    elbo = model.compute_log_likelihood(zd)                      # the step counter moved on: new LV draws
    z1 = _np(model.layers[0]._smp_z).reshape(B, K, -1)
    assert np.abs(z1 - z0).max() > 0.1
    if B * K <= 2000:
        om = build_oracle(spec)
        L_NK = om.log_weights(oracle_noise(spec, [z0] + zs[1:]))[0]
        m_o = L_NK.max(1)
        logp_o = m_o + np.log(np.exp(L_NK - m_o[:, None]).sum(1)) - np.log(K)
        np.testing.assert_allclose(_np(logp), logp_o, rtol=2e-4, atol=2e-2)
        ref = om.build_likelihood(oracle_noise(spec, [z1] + zs[1:]))
        assert abs(elbo - ref) <= ELBO_RTOL * abs(ref), (elbo, ref)
    # route 2: the layer inside the layer kernel, fed the exported draws
    model.lv_in_precompute = False
    e2 = model.compute_log_likelihood([_t(z1, gpu_device)] + zd[1:])
    assert abs(elbo - e2) <= 2e-6 * abs(e2), (elbo, e2)
    # in-kernel draws everywhere: both routes key the streams by (seed, step, model layer index, sample, quad)
    step = int(model._words()[1])
    a = model.compute_log_likelihood(None)
    model._words()[1] = step
    model.lv_in_precompute = True
    b = model.compute_log_likelihood(None)
    assert abs(a - b) <= 2e-6 * abs(a), (a, b)
    assert int(model._words()[1]) == step + 1


def test_full_cov_over_samples_flag_matches_diag(gpu_device):
    """following the reference literally ([B,Dy,K,K] covariance then matrix_diag_part, models.py:133)
    gives the same ELBO as asking the final layer for marginals."""
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(L=2, M=64, B=12, K=6, with_lv=True, seed=8)
    zs = [_t(z, gpu_device) for z in synthetic.make_noise(spec, seed=6)]
    model = synthetic.build_model(spec, gpu_device)
    a = model.compute_log_likelihood(zs)
    model.full_cov_over_samples = True
    zs_f = zs[:-1] + [torch.zeros(12, 1, 6, 1, device=gpu_device)]
    b = model.compute_log_likelihood(zs_f)
    assert abs(a - b) <= 2e-5 * abs(a)


@pytest.mark.parametrize("K", [1, 4, 9])
def test_one_layer_iwvi_equals_closed_form_svgp(gpu_device, K):
    """mirrors reference tests/test_gp_layer.py:15-54 on the IW path: any K gives the SVGP bound."""
    from dgps_with_iwvi_amd import features, kernels, likelihoods, mean_functions
    from dgps_with_iwvi_amd.layers import GPLayer
    from dgps_with_iwvi_amd.models import DGP_IWVI, DGP_VI
    rng = np.random.default_rng(0)
    N, M = 300, 100
    X = np.linspace(0, 1, N).reshape(-1, 1)
    Z = np.linspace(0, 1, M).reshape(-1, 1)
    Y = np.sin(10 * X)
    A = rng.standard_normal((1, 1))
    q_mu = rng.standard_normal((M, 1)).astype(np.float32)
    q_sqrt = (rng.standard_normal((1, M, M)) * 0.1).astype(np.float32)       # non-triangular on purpose
    q_sqrt[0][np.diag_indices(M)] = np.abs(q_sqrt[0][np.diag_indices(M)]) + 0.5
    ko = O.Matern52(1, lengthscales=0.1)
    ref = C.svgp_elbo(X, Y, Z, ko, q_mu.astype(np.float64), q_sqrt.astype(np.float64), 0.1,
                      O.Linear(A.astype(np.float32).astype(np.float64)))
    for cls in (DGP_IWVI, DGP_VI):
        layer = GPLayer(kernels.Matern52(1, lengthscales=0.1), features.InducingPoints(Z), 1,
                        mean_functions.Linear(A))
        layer.q_mu, layer.q_sqrt = _t(q_mu, gpu_device), _t(q_sqrt, gpu_device)
        m = cls(X, Y, [layer], likelihoods.Gaussian(0.1), num_samples=K).to(gpu_device)
        got = m.compute_log_likelihood()
        # Matern52 with lengthscale 0.1 and 100 inducing points on [0,1] is ill-conditioned (cond(Kuu) ~ 1e7): a 1-D layer takes the
        # float64 stage-1 route (5e-3 with float32 arithmetic until round 5)
        assert abs(got - ref) <= 1e-5 * abs(ref), (cls.__name__, got, ref)


def test_vi_elbo_and_predict_match_oracle(gpu_device):
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.models import DGP_VI
    spec = synthetic.make_spec(L=2, M=64, B=24, K=3, with_lv=True, seed=11)
    S, B = 3, 24
    rng = np.random.default_rng(12)
    zs = [rng.standard_normal((S * B, l["latent_dim"] if l["type"] == "lv" else l["q_mu"].shape[1])).astype(np.float32)
          for l in spec["layers"]]
    model = synthetic.build_model(spec, gpu_device, cls=DGP_VI, num_samples=S)
    got = model.compute_log_likelihood([_t(z, gpu_device) for z in zs])
    ref = build_oracle(spec, iw=False, num_samples=S).build_likelihood(zs)
    assert abs(got - ref) <= ELBO_RTOL * abs(ref), (got, ref)


def test_fill_normal(gpu_device):
    from dgps_with_iwvi_amd import _abi
    n = 1 << 20
    a = torch.empty(n, device=gpu_device)
    b = torch.empty(n, device=gpu_device)
    _abi.check(_abi.lib().iwvi_fill_normal(_abi.ptr(a), n, 7, 0, _abi.stream_ptr()))
    _abi.check(_abi.lib().iwvi_fill_normal(_abi.ptr(b), n, 7, 0, _abi.stream_ptr()))
    assert torch.equal(a, b)
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.var()) - 1) < 1e-2
    assert abs(float((a ** 4).mean()) - 3) < 0.1
    _abi.check(_abi.lib().iwvi_fill_normal(_abi.ptr(b), n - 4, 7, 1, _abi.stream_ptr()))
    assert torch.equal(a[4:], b[:n - 4])                       # counter offset = 4 normals
    _abi.check(_abi.lib().iwvi_fill_normal(_abi.ptr(b), n, 8, 0, _abi.stream_ptr()))
    assert not torch.equal(a, b)


# ------------------------------------------------------------------------------------------
# committed golden vectors (tests/golden/*.npz; generator tests/golden/make_golden.py)
# ------------------------------------------------------------------------------------------
def _golden_paths():
    import glob
    import os
    return sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz"))
                  if not os.path.basename(p).startswith("grad_"))      # grad_*: backward-pass targets (row F1), not forward cases


@pytest.mark.parametrize("path", _golden_paths(), ids=lambda p: p.split("/")[-1][:-4])
def test_golden_fixture(gpu_device, path):
    """HIP path vs the committed fp64 vectors: per-layer mean / var / sample, log-weights, ELBO."""
    from dgps_with_iwvi_amd import synthetic
    from test_golden import load
    spec, zs, out = load(path)
    model = synthetic.build_model(spec, gpu_device)
    zd = [_t(z, gpu_device) for z in zs]
    elbo = model.compute_log_likelihood(zd)
    assert abs(elbo - float(out["elbo"])) <= ELBO_RTOL * abs(float(out["elbo"])), (elbo, float(out["elbo"]))
    np.testing.assert_allclose(_np(model.E_log_p_Y(zd)), out["logp"], rtol=2e-4, atol=2e-2)
    fmean, fvar, _, _, samples, means, covs = model._forward_iw(zd)
    n = len(spec["layers"])
    for i in range(n - 1):
        np.testing.assert_allclose(_np(means[i]), out["mean%d" % i], **MEAN_TOL)
        np.testing.assert_allclose(_np(covs[i]), out["var%d" % i], **VAR_TOL)
        np.testing.assert_allclose(_np(samples[i]), out["sample%d" % i], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(_np(fmean), out["mean%d" % (n - 1)], rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(_np(fvar), out["var%d" % (n - 1)], rtol=5e-3, atol=2e-4)


def test_k_shard_merge_on_one_gpu(gpu_device):
    """K-sharding: two 'ranks' with 4 + 3 importance samples on this one GPU; the
    gathered (max, sumexp) pairs merged by iwvi_lse_merge give the unsharded IW-ELBO and the torch merge."""
    from dgps_with_iwvi_amd import sharding, synthetic
    spec = synthetic.make_spec(L=2, M=32, B=40, K=7, with_lv=True, seed=31, n_data=512)
    zs = synthetic.make_noise(spec, seed=32)
    full = synthetic.build_model(spec, gpu_device)
    ref = full.compute_log_likelihood([_t(z, gpu_device) for z in zs])
    parts, k0 = [], 0
    for Kr in sharding.split_samples(7, 2):
        m = synthetic.build_model(spec, gpu_device, num_samples=Kr)
        ms, glob = m.lse_partials([_t(z[:, k0:k0 + Kr], gpu_device) for z in zs], K_total=7)
        parts.append(ms)
        k0 += Kr
    gathered = torch.stack(parts)
    logp, elbo = sharding.merge_lse(gathered, 7, glob, spec["n_data"] / 40)
    assert abs(float(elbo.item()) - ref) <= 1e-6 * abs(ref), (float(elbo.item()), ref)
    np.testing.assert_allclose(_np(logp), _np(sharding.merge_lse_reference(gathered.double(), 7)), rtol=1e-5, atol=1e-4)


@pytest.mark.gpu
def test_batched_merge_matches_per_evaluation_merge():
    """iwvi_lse_merge_steps on [G, steps, B, 2] == iwvi_lse_merge on each evaluation's [G, B, 2] slice."""
    import ctypes
    from dgps_with_iwvi_amd import _abi
    from dgps_with_iwvi_amd.sharding import merge_lse
    dev = torch.device("cuda:0")
    G, S, B, K_total, scale = 3, 4, 257, 12, 7.5
    gen = torch.Generator().manual_seed(5)
    ms = torch.empty(G, S, B, 2)
    ms[..., 0] = torch.randn(G, S, B, generator=gen) * 3 - 50
    ms[..., 1] = torch.rand(G, S, B, generator=gen) * 3 + 1
    ms = ms.to(dev)
    kl = [torch.tensor([1.25, 0.5], dtype=torch.float64, device=dev), torch.tensor([2.0], dtype=torch.float64, device=dev)]
    logp = torch.empty(S, B, dtype=torch.float32, device=dev)
    elbo = torch.empty(S, dtype=torch.float64, device=dev)
    glob_n = (ctypes.c_int32 * 2)(2, 1)
    _abi.check(_abi.lib().iwvi_lse_merge_steps(_abi.ptr(ms), G, S, B, K_total, _abi.ptr_array(kl), glob_n, 2, scale,
                                               _abi.ptr(logp), _abi.ptr(elbo), _abi.stream_ptr()))
    for e in range(S):
        lp, el = merge_lse(ms[:, e].contiguous(), K_total, kl, scale)
        assert torch.equal(lp, logp[e])
        assert float(el) == float(elbo[e])


# ------------------------------------------------------------------------------------------
# BASELINE.json's full sizes: the fp64 oracle takes minutes there, so the HIP path is checked through
# size-independent properties of the estimator (and against itself across decompositions)
# ------------------------------------------------------------------------------------------
FULL = dict(L=2, M=128, B=1024, K=20, with_lv=True, seed=0, parity=True, n_data=65536)    # BASELINE.json configs[2]
FULL_CFG = {"configs2": {},                                   # 2-layer DGP + LatentVariableLayer, M=128, K=20, batch=1024
            "configs1": dict(K=5, with_lv=False),             # 2-layer DGP, RBF, M=128, K=5, batch=1024
            "configs3": dict(L=3, M=256, K=50, B=4096, with_lv=False),    # 3-layer DGP, M=256, K=50, batch=4096 (K-sharded across 8)
            "configs4": dict(L=5, M=512, K=100, B=8192, with_lv=False)}   # 5-layer DGP, M=512, K=100, batch=8192
# (configs[3] / [4] are 8-GPU jobs in BASELINE.json: one GPU evaluates the whole job here -- 1.4 ms / 30 ms per evaluation)
both_full_configs = pytest.mark.parametrize("cfg", ["configs2", "configs1", "configs3", "configs4"])
# float32 rounding of a per-point estimate under a different summation order grows with depth and M (deeper stacks, longer solves)
PERM_TOL = {"configs2": dict(rtol=5e-5, atol=5e-3), "configs1": dict(rtol=5e-5, atol=5e-3),
            "configs3": dict(rtol=2e-4, atol=2e-2), "configs4": dict(rtol=5e-4, atol=5e-2)}


def _full_model(gpu_device, cfg="configs2", **over):
    from dgps_with_iwvi_amd import synthetic
    spec = synthetic.make_spec(**dict(FULL, **dict(FULL_CFG[cfg], **over)))
    return spec, synthetic.build_model(spec, gpu_device)


@both_full_configs
def test_full_size_reproducible_and_jensen_ordered(gpu_device, cfg):
    """Same injected noise -> bit-identical ELBO; logsumexp_k(L) - log K >= mean_k(L) for every point (Jensen),
    and the fused tail's per-point values equal those recomputed from the exported log-weights."""
    from dgps_with_iwvi_amd import synthetic
    spec, model = _full_model(gpu_device, cfg)
    zs = [_t(z, gpu_device) for z in synthetic.make_noise(spec, seed=3)]
    a, b = model.compute_log_likelihood(zs), model.compute_log_likelihood(zs)
    assert a == b
    B, K = spec["B"], spec["K"]
    logw = model._logw(zs).double().reshape(B, K)
    logp = model.E_log_p_Y(zs).double()
    ref = torch.logsumexp(logw, 1) - np.log(K)
    assert torch.all(ref >= logw.mean(1) - 1e-9)
    np.testing.assert_allclose(_np(logp), _np(ref), rtol=2e-6, atol=2e-4)
    elbo_ref = float(ref.sum()) * spec["n_data"] / B - float(sum(g.double().sum() for g in model._global_kls()))
    assert abs(a - elbo_ref) <= 2e-6 * abs(elbo_ref)


@both_full_configs
def test_full_size_invariant_under_sample_permutation(gpu_device, cfg):
    """Permuting the K importance samples of every point (their noise) leaves the per-point estimate unchanged up to
    float32 rounding (a sample's sub-tile decides the summation order of its solve; log-weights are O(100) with
    sensitivity 1 / lik_variance)."""
    from dgps_with_iwvi_amd import synthetic
    spec, model = _full_model(gpu_device, cfg)
    zs = synthetic.make_noise(spec, seed=4)
    perm = np.random.default_rng(0).permutation(spec["K"])
    lp0 = _np(model.E_log_p_Y([_t(z, gpu_device) for z in zs]))
    lp1 = _np(model.E_log_p_Y([_t(z[:, perm], gpu_device) for z in zs]))
    np.testing.assert_allclose(lp0, lp1, **PERM_TOL[cfg])


@both_full_configs
def test_full_size_k_shards_merge_to_the_unsharded_value(gpu_device, cfg):
    """K = 20 as 8 uneven shards (3,3,3,3,2,2,2,2: the 8-GPU split; K = 5 as 4 shards; configs[3]: K = 50 as 7,7,6,6,6,6,6,6 -- the split
    BASELINE.json names; configs[4]: K = 100 as 13,13,13,13,12,12,12,12) merged with iwvi_lse_merge == one evaluation."""
    from dgps_with_iwvi_amd import sharding, synthetic
    spec, model = _full_model(gpu_device, cfg)
    zs = synthetic.make_noise(spec, seed=5)
    ref = model.compute_log_likelihood([_t(z, gpu_device) for z in zs])
    parts, k0, glob = [], 0, None
    for Kr in sharding.split_samples(spec["K"], 8 if spec["K"] >= 8 else 4):
        m = synthetic.build_model(spec, gpu_device, num_samples=Kr)
        ms, glob = m.lse_partials([_t(z[:, k0:k0 + Kr], gpu_device) for z in zs], K_total=spec["K"])
        parts.append(ms.clone())
        k0 += Kr
    _, elbo = sharding.merge_lse(torch.stack(parts), spec["K"], glob, spec["n_data"] / spec["B"])
    assert abs(float(elbo.item()) - ref) <= 2e-6 * abs(ref), (float(elbo.item()), ref)


@both_full_configs
def test_full_size_scale_is_linear_in_num_data(gpu_device, cfg):
    """ELBO(n) = (n / B) sum_b logp_b - KL: two values of num_data differ by exactly the scaled data term."""
    from dgps_with_iwvi_amd import synthetic
    spec, model = _full_model(gpu_device, cfg)
    zs = [_t(z, gpu_device) for z in synthetic.make_noise(spec, seed=6)]
    e1 = model.compute_log_likelihood(zs)
    s = float(model.E_log_p_Y(zs).double().sum())
    model.num_data = 2 * spec["n_data"]
    e2 = model.compute_log_likelihood(zs)
    assert abs((e2 - e1) - s * spec["n_data"] / spec["B"]) <= 1e-6 * abs(e1)


@pytest.mark.parametrize("over", [dict(L=3, M=256, B=512, K=10, with_lv=False),     # M > 128: the generic solve path
                                  dict(L=2, M=120, B=300, K=7, with_lv=True)])      # padded inducing rows, ragged chunks
def test_large_shapes_layer_identities(gpu_device, over):
    """sample = mean + z sqrt(var) at every GP layer (temp_workaround.py:89-91), variances >= 0, finite ELBO,
    reproducible bits -- on shapes the oracle does not reach in test time."""
    from dgps_with_iwvi_amd import synthetic
    spec, model = _full_model(gpu_device, **over)
    zs = synthetic.make_noise(spec, seed=7)
    zd = [_t(z, gpu_device) for z in zs]
    a, b = model.compute_log_likelihood(zd), model.compute_log_likelihood(zd)
    assert a == b and np.isfinite(a)
    fmean, fvar, _, _, samples, means, covs = model._forward_iw(zd)
    assert torch.all(fvar >= 0)
    for i, l in enumerate(spec["layers"][:-1]):
        if l["type"] != "gp" or l["W"] is not None:
            continue                                   # mixed layers: the identity holds per latent GP, not per output
        np.testing.assert_allclose(_np(samples[i]), _np(means[i] + zd[i] * covs[i].sqrt()), rtol=1e-5, atol=1e-5)
