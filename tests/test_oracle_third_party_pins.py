"""Pin the oracle's restatement of the THIRD-PARTY arithmetic on the path (GPflow-1.x
kernels, Gaussian likelihood, TF distributions, reduce_logsumexp) against independent implementations importable
here: scikit-learn's GP kernels, torch.distributions, scipy.  (GPflow / TensorFlow themselves are not installable;
this does not lift "parity unpinned", it removes the risk that a formula was recalled wrongly.)"""
import numpy as np
import pytest
import torch

from oracle import iwvi_oracle as O


@pytest.mark.parametrize("D,ard", [(1, False), (3, True), (8, True)])
def test_rbf_and_matern52_match_sklearn(D, ard):
    """gpflow.kernels.RBF / Matern52 (call sites temp_workaround.py:39,44,45) == sigma^2 * sklearn RBF / Matern(nu=2.5)
    with the same (ARD) lengthscales."""
    from sklearn.gaussian_process.kernels import RBF, Matern
    rng = np.random.default_rng(D)
    X, X2 = rng.standard_normal((13, D)), rng.standard_normal((7, D))
    ls = (0.5 + rng.random(D)) if ard else np.full(D, 0.7)
    var = 1.7
    np.testing.assert_allclose(O.RBF(D, var, ls).K(X, X2), var * RBF(length_scale=ls)(X, X2), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(O.RBF(D, var, ls).K(X), var * RBF(length_scale=ls)(X), rtol=1e-12, atol=1e-12)
    # the restatement keeps GPflow's sqrt(r^2 + 1e-12): a 1e-6-sized difference in r, 1e-11 in k
    np.testing.assert_allclose(O.Matern52(D, var, ls).K(X, X2), var * Matern(length_scale=ls, nu=2.5)(X, X2), rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(O.RBF(D, var, ls).Kdiag(X), np.diag(var * RBF(length_scale=ls)(X)), rtol=1e-14)
    # batched [S, N, D] input (kern.K(Xnew) at :45 broadcasts over S)
    Xb = rng.standard_normal((4, 5, D))
    Kb = O.RBF(D, var, ls).K(Xb)
    for s in range(4):
        np.testing.assert_allclose(Kb[s], var * RBF(length_scale=ls)(Xb[s]), rtol=1e-12, atol=1e-12)


def test_latent_variable_regularisers_match_torch_distributions():
    """layers.py:93-103: log q(W) - log p(W) and the closed-form KL(N(mu, s) || N(0, 1)) vs torch.distributions."""
    rng = np.random.default_rng(0)
    F = rng.standard_normal((6, 4, 3))
    XY = rng.standard_normal((6, 4, 5))
    z = rng.standard_normal((6, 4, 2))
    lv = O.LatentVariableLayer(2, XY_dim=5)
    lv.encoder.bs = [rng.standard_normal(b.shape) * 0.3 for b in lv.encoder.bs]
    q_mu, q_sqrt = lv.encoder(XY)
    q = torch.distributions.Normal(torch.tensor(q_mu), torch.tensor(q_sqrt))
    p = torch.distributions.Normal(torch.zeros(()).double(), torch.ones(()).double())
    s, _, _, kl_s = lv.propagate(F, XY, True, z=z)
    W = torch.tensor(s[..., 3:])
    np.testing.assert_allclose(kl_s, (q.log_prob(W) - p.log_prob(W)).numpy(), rtol=1e-12, atol=1e-12)
    _, _, _, kl_a = lv.propagate(F, XY, False, z=z)
    np.testing.assert_allclose(kl_a, torch.distributions.kl_divergence(q, p).numpy(), rtol=1e-12, atol=1e-13)
    # softplus(raw - 3) of the encoder head (layers.py:150)
    raw = rng.standard_normal(50) * 4
    np.testing.assert_allclose(np.logaddexp(0.0, raw - 3.0), torch.nn.functional.softplus(torch.tensor(raw) - 3.0).numpy(), rtol=1e-13)


def test_gaussian_likelihood_matches_torch_normal():
    """gpflow Gaussian.variational_expectations = E_{f ~ N(mu, v)} log N(y; f, s2) (closed form vs Gauss-Hermite of
    torch's Normal.log_prob) and predict_mean_and_var."""
    rng = np.random.default_rng(1)
    mu, v, y = rng.standard_normal(9), rng.random(9) + 0.05, rng.standard_normal(9)
    s2 = 0.37
    got = O.Gaussian(s2).variational_expectations(mu, v, y)
    x, w = np.polynomial.hermite_e.hermegauss(40)
    f = mu[:, None] + np.sqrt(v)[:, None] * x[None]
    lp = torch.distributions.Normal(torch.tensor(f), np.sqrt(s2)).log_prob(torch.tensor(y)[:, None]).numpy()
    np.testing.assert_allclose(got, (lp * w).sum(1) / np.sqrt(2 * np.pi), rtol=1e-10)
    m2, v2 = O.Gaussian(s2).predict_mean_and_var(mu, v)
    np.testing.assert_allclose(v2, v + s2)


def test_iw_reduction_matches_scipy_logsumexp():
    """models.py:148: tf.reduce_logsumexp(L_NK, 1) - log K, as restated in DGP_IWVI.build_likelihood."""
    from scipy.special import logsumexp
    from dgps_with_iwvi_amd import synthetic
    from oracle.from_spec import build_oracle, oracle_noise
    spec = synthetic.make_spec(L=2, M=10, B=7, K=6, Dx=3, R=2, with_lv=True, seed=4, n_data=77)
    zs = synthetic.make_noise(spec, seed=5)
    om = build_oracle(spec)
    L_NK, glob, _, _, _ = om.log_weights(oracle_noise(spec, zs))
    ref = (logsumexp(L_NK, axis=1) - np.log(6)).sum() * 77 / 7 - np.sum(glob)
    np.testing.assert_allclose(om.build_likelihood(oracle_noise(spec, zs)), ref, rtol=1e-13)


def test_whitened_gauss_kl_matches_torch_mvn_kl():
    """gauss_kl (temp_workaround.py:186-188, white) = sum_r KL(N(q_mu_r, L_r L_r^T) || N(0, I)) vs torch's MVN KL."""
    rng = np.random.default_rng(2)
    M, R = 9, 3
    q_mu = rng.standard_normal((M, R))
    q_sqrt = np.tril(rng.standard_normal((R, M, M))) * 0.3 + np.eye(M)
    ref = 0.0
    for r in range(R):
        q = torch.distributions.MultivariateNormal(torch.tensor(q_mu[:, r]), scale_tril=torch.tensor(np.tril(q_sqrt[r])))
        p = torch.distributions.MultivariateNormal(torch.zeros(M).double(), torch.eye(M).double())
        ref += float(torch.distributions.kl_divergence(q, p))
    # scale_tril needs a positive diagonal; the KL only sees L L^T and log(diag^2), so flip signs for torch only
    sgn = np.sign(np.diagonal(q_sqrt, axis1=-2, axis2=-1))
    assert (sgn > 0).all()
    np.testing.assert_allclose(O.gauss_kl(q_mu, q_sqrt), ref, rtol=1e-12)
