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

LD = np.longdouble        # x86 extended precision (eps 1.1e-19): Kuu of the reference's own test (100 Matern52 points on [0, 1], lengthscale
                          # 0.1) has cond ~ 1e8 and THIS route multiplies by the explicit inverse.  (Round 5: the device on its float64 stage-1
                          # route agreed with the whitened oracle to 1e-7 and with this file only to 4e-5 -- this file's KL was the one off.)


def _inv_logdet(K):
    """Gauss-Jordan inverse and log-determinant of a symmetric positive definite matrix in extended precision (no LAPACK, no Cholesky:
    this file stays an algebraic route of its own).  Returns (K^-1, log det K) as longdouble."""
    A = np.array(K, dtype=LD)
    n = len(A)
    aug = np.concatenate([A, np.eye(n, dtype=LD)], axis=1)
    logdet = LD(0)
    for j in range(n):
        p = j + int(np.argmax(np.abs(aug[j:, j])))
        if p != j:
            aug[[j, p]] = aug[[p, j]]                            # (never taken for an SPD matrix in exact arithmetic; kept for safety)
        piv = aug[j, j]
        logdet += np.log(np.abs(piv))
        aug[j] = aug[j] / piv
        col = aug[:, j].copy()
        col[j] = 0
        aug -= col[:, None] * aug[j][None, :]
    return aug[:, n:], logdet



def svgp_predict(Xs, Z, kern, q_mu, q_sqrt, mean_function=None, jitter=1e-6, full_cov=True):
    """-> mean [N, R], cov [R, N, N] (full covariance) or var [R, N] (full_cov=False: the diagonal only, for large N)."""
    Z = np.asarray(Z, np.float64)
    Kuu = kern.K(Z) + jitter * np.eye(len(Z))
    Lm = np.linalg.cholesky(Kuu)
    Kuu_inv, _ = _inv_logdet(Kuu)
    Kuf = kern.K(Z, Xs)
    Kff = kern.K(Xs) if full_cov else kern.Kdiag(Xs)
    R = q_mu.shape[1]
    # K_fu Kuu^-1 in extended precision (the cancellations of this unwhitened form live in it), rounded to float64 once
    P = np.asarray(Kuu_inv @ Kuf.astype(LD)) if len(Xs) <= 4096 else np.concatenate(
        [np.asarray(Kuu_inv @ Kuf[:, i:i + 4096].astype(LD)) for i in range(0, len(Xs), 4096)], axis=1)   # M x N
    mean = np.zeros((len(Xs), R))
    cov = np.zeros((R, len(Xs), len(Xs))) if full_cov else np.zeros((R, len(Xs)))
    for r in range(R):
        Lq = np.tril(q_sqrt[r])
        m_u = (Lm @ q_mu[:, r]).astype(LD)
        S_u = (Lm @ Lq @ Lq.T @ Lm.T).astype(LD)
        mean[:, r] = np.asarray(P.T @ m_u, np.float64)
        if full_cov:
            P64 = np.asarray(P, np.float64)
            cov[r] = Kff - np.asarray(Kuf.T.astype(LD) @ P, np.float64) + P64.T @ (np.asarray(S_u @ P, np.float64))
        else:
            cov[r] = np.asarray(Kff.astype(LD) - np.sum(Kuf.astype(LD) * P, 0) + np.sum(P * (S_u @ P), 0), np.float64)
    if mean_function is not None:
        mean = mean + mean_function(Xs)
    return mean, cov


def svgp_kl(Z, kern, q_mu, q_sqrt, jitter=1e-6):
    Z = np.asarray(Z, np.float64)
    M = len(Z)
    Kuu = kern.K(Z) + jitter * np.eye(M)
    Lm = np.linalg.cholesky(Kuu)
    Kuu_inv, logdet_K = _inv_logdet(Kuu)
    kl = LD(0)
    for r in range(q_mu.shape[1]):
        Lq = np.tril(q_sqrt[r])
        m_u = (Lm @ q_mu[:, r]).astype(LD)
        S_u = (Lm.astype(LD) @ Lq.astype(LD)) @ (Lq.T.astype(LD) @ Lm.T.astype(LD))
        # log det S_u from its factors, det(Lm)^2 det(Lq)^2: S_u itself has cond(Lm)^2 cond(Lq)^2 -- for the reference test's random
        # lower-triangular q_sqrt (cond(Lq) grows like 2^M) far beyond any floating-point format, and an elimination on the formed
        # matrix returned a KL that was off by 60 .. 74 of 2491 in float64 AND in extended precision
        logdet_S = logdet_K + 2 * np.sum(np.log(np.abs(np.diag(Lq).astype(LD))))
        kl += LD(0.5) * (np.trace(Kuu_inv @ S_u) + m_u @ (Kuu_inv @ m_u) - M + logdet_K - logdet_S)
    return float(kl)


def svgp_elbo(X, Y, Z, kern, q_mu, q_sqrt, lik_variance, mean_function=None,
              jitter=1e-6, num_data=None):
    """Gaussian-likelihood SVGP bound: sum_n E_q[log p(y_n | f_n)] * scale - KL."""
    mean, cov = svgp_predict(X, Z, kern, q_mu, q_sqrt, mean_function, jitter, full_cov=False)
    var = cov.T                                         # N x R
    ve = (-0.5 * np.log(2 * np.pi) - 0.5 * np.log(lik_variance)
          - 0.5 * ((Y - mean) ** 2 + var) / lik_variance)
    scale = 1.0 if num_data is None else num_data / len(X)
    return np.sum(ve) * scale - svgp_kl(Z, kern, q_mu, q_sqrt, jitter)
