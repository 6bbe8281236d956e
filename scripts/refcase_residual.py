"""Residual of the reference's test_dgp_zero_inner_layers (tests/test_gp_layer.py:57-96) with the inner layer's random draw left in, now that
a 1-D inner layer takes the float64 stage-1 route: max |d mean|, max |d cov| / |cov|max over several draws, and the inner layer's variance."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from oracle import iwvi_oracle as O
from oracle import svgp_closed_form as C
from dgps_with_iwvi.layers import GPLayer
from dgps_with_iwvi.models import DGP_VI
from dgps_with_iwvi import kernels, likelihoods, mean_functions, settings
dev = torch.device("cuda:0")
_f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
_t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)
N, Dy = 10, 2
rng = np.random.RandomState(1)
X = np.linspace(0, 1, N).reshape(-1, 1); Xs = np.linspace(0, 1, N - 1).reshape(-1, 1)
Y = np.concatenate([np.sin(10 * X), np.cos(10 * X)], 1)
A = _f32(rng.randn(1, 2)); q_mu = _f32(rng.randn(N, Dy)); q_sqrt = _f32(rng.randn(Dy, N, N))
ko = O.Matern52(1, lengthscales=float(np.float32(0.1)))
m1, v1 = C.svgp_predict(_f32(Xs), _f32(X), ko, q_mu, q_sqrt, O.Linear(A), jitter=1e-18)
for mode in ("auto", "off"):
    with settings.temp_settings(jitter=1e-18, f64_stage1=mode):
        m = DGP_VI(X, Y, [GPLayer(kernels.RBF(1, variance=1e-6), X, 1, mean_functions.Identity()),
                          GPLayer(kernels.Matern52(1, lengthscales=0.1), X, Dy, mean_functions.Linear(A))],
                   likelihoods.Gaussian(variance=1e-1)).to(dev)
        m.layers[-1].q_mu = _t(q_mu); m.layers[-1].q_sqrt = _t(q_sqrt); m.layers[0].q_sqrt = m.layers[0].q_sqrt * 1e-12
        dm, dv = [], []
        for rep in range(40):
            m2, v2 = m.predict_f_full_cov(Xs)
            dm.append(np.abs(m2.double().cpu().numpy() - m1).max()); dv.append(np.abs(v2.double().cpu().numpy() - v1).max() / np.abs(v1).max())
        s0, mean0, cov0, _ = m.layers[0].propagate(_t(Xs), full_cov=False)
        print("f64_stage1=%s uses=%s  max|dmean| %.3e (median %.3e)  max|dcov|/|cov| %.3e   inner var max %.3e  |sample - x| max %.3e   |m1|max %.2f" % (
            mode, m.layers[0].uses_f64_stage1(), max(dm), np.median(dm), max(dv), float(cov0.max()), float((s0 - _t(Xs)).abs().max()), np.abs(m1).max()))
