"""Oracle of the evaluation loop (TEST INFRASTRUCTURE ONLY): experiments/run_conditional_density_estimation.py:148-165
restated in NumPy float64 -- Gaussian KDE with Silverman's bandwidth, log density at y, squared error of the sample
mean.  Pinned against ``sklearn.neighbors.KernelDensity`` itself (the reference's dependency, importable here) in
tests/test_kde_oracle.py."""
import numpy as np


def kde_loglik(samples, y):
    """samples [S, N], y [N] -> (logp [N], sqerr [N])."""
    samples, y = np.asarray(samples, dtype=np.float64), np.asarray(y, dtype=np.float64).reshape(-1)
    S = samples.shape[0]
    bw = 1.06 * samples.std(0) * S ** (-1.0 / 5)                       # :158 (np.std: population)
    e = -0.5 * ((y[None, :] - samples) / bw[None, :]) ** 2
    m = e.max(0)
    logp = m + np.log(np.exp(e - m).sum(0)) - np.log(S * bw) - 0.5 * np.log(2 * np.pi)
    return logp, (samples.mean(0) - y) ** 2
