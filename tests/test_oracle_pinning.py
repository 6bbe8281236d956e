"""Pin the fp64 oracle with closed-form identities (list 1-6).

The reference's own tests are differential tests against gpflow.models.SVGP
(reference tests/test_gp_layer.py:15-96) with no stored numbers; GPflow is not
installable here, so the SVGP side is oracle/svgp_closed_form.py, an
independent unwhitened derivation.
"""
import numpy as np
import pytest

from oracle import iwvi_oracle as O
from oracle import svgp_closed_form as C


def _toy(N=40, M=12, Dy=1, seed=0):
    rng = np.random.default_rng(seed)
    X = np.linspace(0, 1, N).reshape(-1, 1)
    Z = np.linspace(0, 1, M).reshape(-1, 1)
    Y = np.concatenate([np.sin(10 * X), np.cos(10 * X)], 1)[:, :Dy]
    q_mu = rng.standard_normal((M, Dy))
    q_sqrt = rng.standard_normal((Dy, M, M))       # non-triangular on purpose (ref :36)
    A = rng.standard_normal((1, Dy))
    return X, Y, Z, q_mu, q_sqrt, A


@pytest.mark.parametrize("kern_cls", [O.Matern52, O.RBF])
def test_one_layer_dgp_vi_equals_svgp(kern_cls):
    """mirrors reference tests/test_gp_layer.py:15-54 (test_gp_layer)."""
    X, Y, Z, q_mu, q_sqrt, A = _toy()
    kern = kern_cls(1, lengthscales=0.1)
    mf = O.Linear(A)
    layer = O.GPLayer(kern, Z, 1, mf)
    layer.q_mu, layer.q_sqrt = q_mu, q_sqrt
    m = O.DGP_VI(X, Y, [layer], O.Gaussian(1e-1), num_samples=1)
    L2 = m.build_likelihood()
    L1 = C.svgp_elbo(X, Y, Z, kern, q_mu, q_sqrt, 1e-1, mf)
    np.testing.assert_allclose(L1, L2, rtol=1e-7)
    Xs = np.linspace(0, 1, 39).reshape(-1, 1)
    m2, v2 = m.build_predict(Xs, full_cov=True)
    m1, v1 = C.svgp_predict(Xs, Z, kern, q_mu, q_sqrt, mf)
    np.testing.assert_allclose(m1, m2, rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(v1, v2, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("K", [1, 3, 7])
def test_one_layer_iwvi_equals_svgp_any_K(K):
    """1-layer DGP_IWVI has identical log-weights over k, so logsumexp - log K is the
    identity and the IW-ELBO equals the SVGP bound (exercises models.py:112-150 incl.
    the [N,Dy,K,K] diag extraction)."""
    X, Y, Z, q_mu, q_sqrt, A = _toy(N=17, M=9)
    kern = O.RBF(1, lengthscales=0.3, variance=1.3)
    mf = O.Linear(A)
    layer = O.GPLayer(kern, Z, 1, mf)
    layer.q_mu, layer.q_sqrt = q_mu, q_sqrt
    m = O.DGP_IWVI(X, Y, [layer], O.Gaussian(0.2), num_samples=K, num_data=170)
    L = m.build_likelihood()
    Lref = C.svgp_elbo(X, Y, Z, kern, q_mu, q_sqrt, 0.2, mf, num_data=170)
    np.testing.assert_allclose(L, Lref, rtol=1e-7)


def test_zero_inner_layer_collapse():
    """mirrors reference tests/test_gp_layer.py:57-96 (test_dgp_zero_inner_layers)."""
    rng = np.random.default_rng(1)
    N, Dy = 10, 2
    X = np.linspace(0, 1, N).reshape(-1, 1)
    Xs = np.linspace(0, 1, N - 1).reshape(-1, 1)
    Y = np.concatenate([np.sin(10 * X), np.cos(10 * X)], 1)
    kern = O.Matern52(1, lengthscales=0.1)
    mf = O.Linear(rng.standard_normal((1, 2)))
    q_mu = rng.standard_normal((N, Dy))
    q_sqrt = rng.standard_normal((Dy, N, N))
    m1, v1 = C.svgp_predict(Xs, X, kern, q_mu, q_sqrt, mf)
    inner = O.GPLayer(O.RBF(1, variance=1e-6), X, 1, O.Identity(), jitter=1e-18)
    inner.q_sqrt = inner.q_sqrt * 1e-12
    outer = O.GPLayer(kern, X, Dy, mf, jitter=1e-18)
    outer.q_mu, outer.q_sqrt = q_mu, q_sqrt
    m = O.DGP_VI(X, Y, [inner, outer], O.Gaussian(1e-1))
    # z = 0: the inner layer's sample is its mean (its variance is ~1e-6 anyway)
    m2, v2 = m.build_predict(Xs, full_cov=True)
    # the closed form was built with the default jitter; rebuild it at 1e-18 for a fair diff
    m1, v1 = C.svgp_predict(Xs, X, kern, q_mu, q_sqrt, mf, jitter=1e-18)
    np.testing.assert_allclose(m1, m2, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(v1, v2, atol=1e-5, rtol=1e-5)


def test_multisample_equals_flat_2d():
    """temp_workaround.py:12-98 with [S,N,D] input == 2-D conditional on [S*N,D]."""
    rng = np.random.default_rng(2)
    S, N, D, M, R = 5, 4, 3, 8, 2
    Xn = rng.standard_normal((S, N, D))
    Z = rng.standard_normal((M, D))
    kern = O.RBF(D, lengthscales=np.array([0.7, 1.1, 1.9]), variance=0.8)
    f = rng.standard_normal((M, R))
    q_sqrt = rng.standard_normal((R, M, M)) * 0.3
    z = rng.standard_normal((S, N, R))
    s3, m3, v3 = O.independent_multisample_sample_conditional(
        Xn, Z, kern, f, q_sqrt=q_sqrt, white=True, z=z)
    s2, m2, v2 = O.sample_conditional(Xn.reshape(S * N, D), Z, kern, f, q_sqrt=q_sqrt,
                                      white=True, z=z.reshape(S * N, R))
    np.testing.assert_allclose(m3.reshape(S * N, R), m2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(v3.reshape(S * N, R), v2, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(s3.reshape(S * N, R), s2, rtol=1e-10, atol=1e-12)
    # full covariance over N: its diagonal is the marginal variance
    _, mf_, vf = O.independent_multisample_sample_conditional(
        Xn, Z, kern, f, q_sqrt=q_sqrt, white=True, full_cov=True)
    np.testing.assert_allclose(np.diagonal(vf, axis1=-2, axis2=-1).transpose(0, 2, 1), v3,
                               rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(mf_, m3, rtol=1e-12)
    # and each [N,N] block equals the 2-D full-cov conditional on that row of S
    for s in range(S):
        _, _, v2f = O.sample_conditional(Xn[s], Z, kern, f, q_sqrt=q_sqrt, white=True,
                                         full_cov=True)
        np.testing.assert_allclose(vf[s], v2f, rtol=1e-9, atol=1e-12)


def test_conditional_against_unwhitened_closed_form():
    """mean / marginal variance of the whitened batched conditional == dense unwhitened algebra."""
    rng = np.random.default_rng(3)
    S, N, D, M, R = 3, 6, 2, 10, 3
    Xn = rng.standard_normal((S, N, D))
    Z = rng.standard_normal((M, D))
    kern = O.RBF(D, lengthscales=1.3, variance=1.7)
    f = rng.standard_normal((M, R))
    q_sqrt = rng.standard_normal((R, M, M)) * 0.5
    _, m3, v3 = O.independent_multisample_sample_conditional(
        Xn, Z, kern, f, q_sqrt=q_sqrt, white=True)
    mc, vc = C.svgp_predict(Xn.reshape(S * N, D), Z, kern, f, q_sqrt)
    np.testing.assert_allclose(m3.reshape(S * N, R), mc, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(v3.reshape(S * N, R),
                               np.stack([np.diag(c) for c in vc], 1), rtol=1e-6, atol=1e-8)


def test_gauss_kl_against_definition():
    rng = np.random.default_rng(4)
    M, R = 7, 3
    q_mu = rng.standard_normal((M, R))
    q_sqrt = rng.standard_normal((R, M, M))
    kl = O.gauss_kl(q_mu, q_sqrt)
    ref = 0.0
    for r in range(R):
        L = np.tril(q_sqrt[r])
        S_ = L @ L.T
        ref += 0.5 * (np.trace(S_) + q_mu[:, r] @ q_mu[:, r] - M - np.linalg.slogdet(S_)[1])
    np.testing.assert_allclose(kl, ref, rtol=1e-10)


def test_shared_mixed_mok_mixing():
    """temp_workaround.py:123-147: forced marginal conditional then W mixing."""
    rng = np.random.default_rng(5)
    S, N, D, M, R, P = 3, 4, 3, 6, 2, 3
    Xn = rng.standard_normal((S, N, D))
    Z = rng.standard_normal((M, D))
    base = O.RBF(D, lengthscales=np.sqrt(D))
    W = rng.standard_normal((P, R))
    f = rng.standard_normal((M, R))
    q_sqrt = rng.standard_normal((R, M, M)) * 0.2
    z = rng.standard_normal((S, N, R))
    fs, fm, fv = O.multisample_sample_conditional(Xn, Z, O.SharedMixedMok(base, W), f,
                                                  full_cov=True, q_sqrt=q_sqrt, white=True, z=z)
    gs, gm, gv = O.independent_multisample_sample_conditional(Xn, Z, base, f, q_sqrt=q_sqrt,
                                                              white=True, z=z)
    assert fs.shape == (S, N, P) and fv.shape == (S, N, P)          # full_cov ignored
    np.testing.assert_allclose(fs, gs @ W.T)
    np.testing.assert_allclose(fm, gm @ W.T)
    np.testing.assert_allclose(fv, gv @ (W ** 2).T)


def test_lv_layer_terms():
    """layers.py:72-105: sampled log q/p averages to the analytic KL; prior mode is N(0,1)."""
    rng = np.random.default_rng(6)
    lv = O.LatentVariableLayer(2, XY_dim=4, encoder=O.Encoder(2, 4, [20, 20], rng))
    F = rng.standard_normal((3, 5, 3))
    XY = rng.standard_normal((3, 5, 4))
    z = rng.standard_normal((3, 5, 2))
    s, m, c, kl = lv.propagate(F, XY, True, z=z)
    assert s.shape == (3, 5, 5) and kl.shape == (3, 5, 2)
    np.testing.assert_allclose(s[..., :3], F)
    np.testing.assert_allclose(c[..., :3], 0)
    q_mu, q_sqrt = lv.encoder(XY)
    np.testing.assert_allclose(s[..., 3:], q_mu + z * q_sqrt)
    _, _, _, kl_a = lv.propagate(F, XY, False, z=z)
    zz = rng.standard_normal((20000,) + q_mu.shape)
    Wm = q_mu + zz * q_sqrt
    mc = np.mean(-0.5 * zz ** 2 - np.log(q_sqrt) + 0.5 * Wm ** 2, 0)
    np.testing.assert_allclose(mc, kl_a, atol=0.05)
    sp, mp, cp, klp = lv.propagate(F, None, True, z=z)              # prior mode
    np.testing.assert_allclose(sp[..., 3:], z)
    np.testing.assert_allclose(klp, 0, atol=1e-12)


def _two_layer_lv_model(K, seed=7, B=6, M=8, Dx=3, R=2):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((B, Dx))
    Y = np.sin(X.sum(1, keepdims=True))
    W = rng.standard_normal((Dx, R))
    A = np.eye(Dx + 1, Dx)
    lv = O.LatentVariableLayer(1, XY_dim=Dx + 1, encoder=O.Encoder(1, Dx + 1, [20, 20], rng))
    inner = O.GPLayer(O.SharedMixedMok(O.RBF(Dx + 1, lengthscales=np.sqrt(Dx + 1)), W),
                      rng.standard_normal((M, Dx + 1)), R, O.Linear(A))
    inner.q_mu = rng.standard_normal((M, R))
    inner.q_sqrt = np.tril(rng.standard_normal((R, M, M))) * 0.1 + np.eye(M) * 0.3
    final = O.GPLayer(O.RBF(Dx, lengthscales=np.sqrt(Dx)), rng.standard_normal((M, Dx)), 1)
    final.q_mu = rng.standard_normal((M, 1))
    final.q_sqrt = np.tril(rng.standard_normal((1, M, M))) * 0.1 + np.eye(M)
    cls = O.DGP_IWVI if K else O.DGP_VI
    return cls(X, Y, [lv, inner, final], O.Gaussian(0.05), num_samples=max(K, 1)), rng


def test_iw_bound_monotone_in_K_and_above_vi():
    """disabled reference tests' intent (tests/test_latent_var_layer.py:137-241):
    E[IW_K] is non-decreasing in K and E[IW_K] >= E[VI] (Jensen / Burda et al.)."""
    vals = {}
    for K in (1, 5, 25):
        m, rng = _two_layer_lv_model(K)
        B = m.X.shape[0]
        ests = []
        for _ in range(300):
            zs = [rng.standard_normal((B, K, 1)), rng.standard_normal((B, K, 2)),
                  rng.standard_normal((B, 1, K, 1))]
            ests.append(m.build_likelihood(zs))
        vals[K] = (np.mean(ests), np.std(ests) / np.sqrt(len(ests)))
    assert vals[5][0] > vals[1][0] - 3 * (vals[5][1] + vals[1][1])
    assert vals[25][0] > vals[5][0] - 3 * (vals[25][1] + vals[5][1])
    assert vals[25][0] > vals[1][0]


def test_iwvi_K1_matches_vi_single_sample_in_expectation():
    """K=1: the IW estimator is the single-sample VI estimator with the sampled log q/p
    instead of the analytic KL -- same expectation (reference test intent, :180-241)."""
    m_iw, rng = _two_layer_lv_model(1)
    m_vi, _ = _two_layer_lv_model(0)
    B = m_iw.X.shape[0]
    iw, vi = [], []
    for _ in range(1500):
        z1, z2 = rng.standard_normal((B, 1, 1)), rng.standard_normal((B, 1, 2))
        iw.append(m_iw.build_likelihood([z1, z2, np.zeros((B, 1, 1, 1))]))
        vi.append(m_vi.build_likelihood([z1[:, 0], z2[:, 0], None]))
    se = np.std(iw) / np.sqrt(len(iw)) + np.std(vi) / np.sqrt(len(vi))
    assert abs(np.mean(iw) - np.mean(vi)) < 4 * se


def test_reference_full_cov_sample_bug_documented():
    """temp_workaround.py:95 adds fmean [S,N,R] to a [S,R,N,1] tensor: for R=1 the literal
    restatement broadcasts to [S,1,N,S] garbage unless S==1; the intended form is [S,N,R]."""
    rng = np.random.default_rng(8)
    S, N, D, M = 1, 4, 2, 5
    Xn = rng.standard_normal((S, N, D))
    Z = rng.standard_normal((M, D))
    kern = O.RBF(D)
    f = rng.standard_normal((M, 1))
    z = rng.standard_normal((S, 1, N, 1))
    s_int, _, _ = O.independent_multisample_sample_conditional(
        Xn, Z, kern, f, full_cov=True, white=True, z=z, intended_full_cov_sample=True)
    assert s_int.shape == (S, N, 1)
    with pytest.raises(Exception):
        O.independent_multisample_sample_conditional(
            rng.standard_normal((3, N, D)), Z, kern, rng.standard_normal((M, 2)),
            full_cov=True, white=True, z=rng.standard_normal((3, 2, N, 1)),
            intended_full_cov_sample=False)
