"""Independent closed-form SVGP used to PIN the oracle (test infrastructure only).

The reference's own tests (tests/test_gp_layer.py:15-96) check a one-layer
DGP_VI against ``gpflow.models.SVGP``.  GPflow is not available, so this file
derives the same quantities by a different algebraic route than
oracle/iwvi_oracle.py: *unwhitened* q(u) = N(m_u, S_u) with dense solves,
``inv`` and ``slogdet`` -- no Cholesky-whitened projection, no ``A = L^-1 Kuf``.

    m_u = Lm q_mu,  S_u = Lm Lq Lq^T Lm^T          (whitened -> unwhitened)
    mean(x) = K_fu Kuu^-1 m_u + mf(x)
    cov     = K_ff - K_fu Kuu^-1 K_uf + K_fu Kuu^-1 S_u Kuu^-1 K_uf
    KL      = 1/2 [ tr(Kuu^-1 S_u) + m_u^T Kuu^-1 m_u - M + logdet Kuu - logdet S_u ]
"""
import numpy as np


def svgp_predict(Xs, Z, kern, q_mu, q_sqrt, mean_function=None, jitter=1e-6, full_cov=True):
    """-> mean [N, R], cov [R, N, N] (full covariance) or var [R, N] (full_cov=False: the diagonal only, for large N)."""
    Z = np.asarray(Z, np.float64)
    Kuu = kern.K(Z) + jitter * np.eye(len(Z))
    Lm = np.linalg.cholesky(Kuu)
    Kuu_inv = np.linalg.inv(Kuu)
    Kuf = kern.K(Z, Xs)
    Kff = kern.K(Xs) if full_cov else kern.Kdiag(Xs)
    R = q_mu.shape[1]
    P = Kuu_inv @ Kuf                                   # M x N
    mean = np.zeros((len(Xs), R))
    cov = np.zeros((R, len(Xs), len(Xs))) if full_cov else np.zeros((R, len(Xs)))
    for r in range(R):
        Lq = np.tril(q_sqrt[r])
        m_u = Lm @ q_mu[:, r]
        S_u = Lm @ Lq @ Lq.T @ Lm.T
        mean[:, r] = P.T @ m_u
        if full_cov:
            cov[r] = Kff - Kuf.T @ P + P.T @ S_u @ P
        else:
            cov[r] = Kff - np.sum(Kuf * P, 0) + np.sum(P * (S_u @ P), 0)
    if mean_function is not None:
        mean = mean + mean_function(Xs)
    return mean, cov


def svgp_kl(Z, kern, q_mu, q_sqrt, jitter=1e-6):
    Z = np.asarray(Z, np.float64)
    M = len(Z)
    Kuu = kern.K(Z) + jitter * np.eye(M)
    Lm = np.linalg.cholesky(Kuu)
    Kuu_inv = np.linalg.inv(Kuu)
    _, logdet_K = np.linalg.slogdet(Kuu)
    kl = 0.0
    for r in range(q_mu.shape[1]):
        Lq = np.tril(q_sqrt[r])
        m_u = Lm @ q_mu[:, r]
        S_u = Lm @ Lq @ Lq.T @ Lm.T
        _, logdet_S = np.linalg.slogdet(S_u)
        kl += 0.5 * (np.trace(Kuu_inv @ S_u) + m_u @ Kuu_inv @ m_u - M + logdet_K - logdet_S)
    return kl


def svgp_elbo(X, Y, Z, kern, q_mu, q_sqrt, lik_variance, mean_function=None,
              jitter=1e-6, num_data=None):
    """Gaussian-likelihood SVGP bound: sum_n E_q[log p(y_n | f_n)] * scale - KL."""
    mean, cov = svgp_predict(X, Z, kern, q_mu, q_sqrt, mean_function, jitter, full_cov=False)
    var = cov.T                                         # N x R
    ve = (-0.5 * np.log(2 * np.pi) - 0.5 * np.log(lik_variance)
          - 0.5 * ((Y - mean) ** 2 + var) / lik_variance)
    scale = 1.0 if num_data is None else num_data / len(X)
    return np.sum(ve) * scale - svgp_kl(Z, kern, q_mu, q_sqrt, jitter)
