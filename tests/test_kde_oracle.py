"""The evaluation-loop oracle against the reference's own dependency: sklearn's KernelDensity, used exactly as
experiments/run_conditional_density_estimation.py:158-162 uses it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.kde_oracle import kde_loglik   # noqa: E402


def test_kde_oracle_equals_sklearn_kernel_density():
    from sklearn.neighbors import KernelDensity
    rng = np.random.default_rng(0)
    S, N = 300, 7
    samples = rng.standard_normal((S, N)) * rng.uniform(0.2, 3.0, N) + rng.standard_normal(N)
    y = rng.standard_normal(N)
    logp, sq = kde_loglik(samples, y)
    for i in range(N):
        Ss = samples[:, i:i + 1]
        bandwidth = 1.06 * np.std(Ss) * len(Ss) ** (-1. / 5)
        ref = KernelDensity(bandwidth=float(bandwidth)).fit(Ss).score(y[i].reshape(-1, 1))
        assert abs(logp[i] - ref) <= 1e-9 * max(1.0, abs(ref)), (i, logp[i], ref)
        assert abs(sq[i] - (np.average(Ss) - y[i]) ** 2) <= 1e-12
