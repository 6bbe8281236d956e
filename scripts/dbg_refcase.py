import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import iwvi_oracle as O
from oracle import svgp_closed_form as C
from dgps_with_iwvi.layers import GPLayer
from dgps_with_iwvi.models import DGP_VI
from dgps_with_iwvi import kernels, likelihoods, mean_functions
dev = torch.device("cuda:0")
_f32 = lambda a: np.asarray(a, np.float32).astype(np.float64)
_t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32, device=dev)
N, M, Dy = 10001, 100, 1
np.random.seed(0)
X = np.linspace(0, 1, N).reshape(-1, 1); Z = np.linspace(0, 1, M).reshape(-1, 1)
Y = np.concatenate([np.sin(10 * X), np.cos(10 * X)], 1)[:, 0:1]
A = _f32(np.random.randn(1, Dy)); q_mu = _f32(np.random.randn(M, Dy)); q_sqrt = _f32(np.random.randn(Dy, M, M))
X32, Z32, Y32 = _f32(X), _f32(Z), _f32(Y)
ko = O.Matern52(1, lengthscales=float(np.float32(0.1))); mfo = O.Linear(A)
L1 = C.svgp_elbo(X32, Y32, Z32, ko, q_mu, q_sqrt, float(np.float32(1e-1)), mfo)
layer = GPLayer(kernels.Matern52(1, lengthscales=0.1), Z, Dy, mean_functions.Linear(A))
m_dgp = DGP_VI(X, Y, [layer], likelihoods.Gaussian(variance=1e-1), num_samples=1).to(dev)
m_dgp.layers[0].q_mu = _t(q_mu); m_dgp.layers[0].q_sqrt = _t(q_sqrt)
L2 = m_dgp.compute_log_likelihood()
print("bound", L1, L2, (L2 - L1), (L2 - L1) / abs(L1))
m1, v1 = C.svgp_predict(X32, Z32, ko, q_mu, q_sqrt, mfo, full_cov=False)
m2, v2 = m_dgp.predict_f(_t(X32))
m2 = m2.double().cpu().numpy().reshape(m1.shape); v2 = v2.double().cpu().numpy().reshape(-1); v1 = np.asarray(v1).reshape(-1); print(m1.shape, v1.shape, Y32.shape)
s2 = float(np.float32(0.1))
print("mean: max|d| %.3e  |m|max %.2f   var: max|d| %.3e mean d %.3e |v|max %.2f" % (np.abs(m2 - m1).max(), np.abs(m1).max(), np.abs(v2 - v1).max(), (v2 - v1).mean(), np.abs(v1).max()))
t_mean = (-0.5 * ((Y32 - m2) ** 2 - (Y32 - m1) ** 2) / s2).sum()
t_var = (-0.5 * (v2 - v1) / s2).sum(); m1 = m1.reshape(Y32.shape); m2 = m2.reshape(Y32.shape); t_mean = (-0.5 * ((Y32 - m2) ** 2 - (Y32 - m1) ** 2) / s2).sum()
print("bound difference from the mean term %.4f  from the variance term %.4f" % (t_mean, t_var))
kl_dev = float(m_dgp.layers[0].kl.item()); print("KL dev %.6f oracle %.6f" % (kl_dev, O.gauss_kl(q_mu, np.tril(q_sqrt))))
d = (v2 - v1).ravel()
bad = np.argsort(-np.abs(d))[:12]
print("worst points", [(int(i), float("%.4g" % d[i]), float("%.4g" % v1.ravel()[i])) for i in bad])
print("n |d|>1e-3:", int((np.abs(d) > 1e-3).sum()), " positions:", np.where(np.abs(d) > 1e-3)[0][:40])
S, Nn = 1, m_dgp.X.shape[0]
m_dgp.precompute(with_encoders=True)
_, _, red = m_dgp._fused_forward(S * Nn, 1, Nn, (S * Nn,), zs=None, sampled_kl=False, elbo=dict(B=Nn, K=S, stride_b=1, stride_k=Nn, mode_vi=True))
logp = red[1].double().cpu().numpy().ravel()
ref_lp = (-0.5 * np.log(2 * np.pi) - 0.5 * np.log(s2) - 0.5 * ((Y32 - m1) ** 2 + v1.reshape(-1, 1)) / s2).ravel()
dd = logp - ref_lp
print("per-point logp: sum d %.4f  max|d| %.3e  mean d %.3e ; first 5 d" % (dd.sum(), np.abs(dd).max(), dd.mean()), dd[:5], " last 5", dd[-5:])
print("elbo from red", float(red[0].item()), " sum logp", logp.sum(), " ref sum", ref_lp.sum(), "KL", kl_dev, " ref total", ref_lp.sum() - kl_dev, "L1", L1)
