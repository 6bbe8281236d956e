"""Seeded synthetic DGP of the benchmark family (BASELINE.json configs).

The layer stack and initial values follow the reference's model factory
(experiments/build_models.py:176-268,276-278): Dx-dimensional inputs, optional
``LatentVariableLayer(1)`` first, ``L-1`` inner ``G<R>`` layers (R latent GPs, SharedMixedMok mixing
W from the right singular vectors of X, identity-padded Linear mean function, RBF-ARD with
lengthscale sqrt(D_in), variance 1), then a plain-RBF final layer with Dy outputs.

``make_spec`` returns plain NumPy arrays (float32-representable values stored as float64), so the
same numbers can be handed to the device model (``build_model``) and to the fp64 oracle
(oracle/from_spec.py) -- nothing here imports the oracle.
"""
import numpy as np


def _f32(a):
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def make_spec(L=2, M=128, B=1024, K=20, Dx=8, Dy=1, R=5, with_lv=False, seed=0, n_data=None,
              parity=True, latent_dim=1):
    """L = number of GP layers (L-1 inner ``G<R>`` + final).  parity=True randomises q_mu / q_sqrt
    (q_mu ~ N(0,1), q_sqrt = 0.1 tril(N(0,1)) + I scaled) so that every term of the conditional is
    exercised; parity=False uses the reference's initial values (q_mu = 0, inner q_sqrt = 1e-5 I)."""
    rng = np.random.default_rng(seed)
    n_data = max(B, M) if n_data is None else n_data
    X = _f32(rng.standard_normal((n_data, Dx)))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n_data, 1))
    Y = np.tile(Y, [1, Dy]) if Dy > 1 else Y
    Y = _f32((Y - Y.mean(0)) / Y.std(0))
    Z0 = X[:M].copy()                                              # build_models.py:179-183 (no k-means)
    P = np.linalg.svd(X, full_matrices=False)[2]                   # :186
    layers = []
    D_in, D_out = Dx, Dx
    if with_lv:
        XYd = Dx + Dy                                              # :236 (DX+1 in the reference, Dy=1)
        dims = [XYd, 20, 20, 2 * latent_dim]
        enc_W, enc_b = [], []
        for din, dout in zip(dims[:-1], dims[1:]):
            enc_W.append(_f32(rng.standard_normal((din, dout)) * (2.0 / (din + dout)) ** 0.5))
            enc_b.append(_f32(rng.standard_normal(dout) * 0.1 if parity else np.zeros(dout)))
        layers.append(dict(type="lv", latent_dim=latent_dim, dims=dims, enc_W=enc_W, enc_b=enc_b))
        D_in += latent_dim                                         # :234

    def gp_params(D_in_, R_, scale_sqrt):
        ZZ = rng.standard_normal((M, D_in_))                        # :218-219, :239-240
        ZZ[:, :min(D_in_, Dx)] = Z0[:, :min(D_in_, Dx)]
        if parity:
            q_mu = rng.standard_normal((M, R_))
            q_sqrt = (np.tril(rng.standard_normal((R_, M, M))) * 0.1 / np.sqrt(M) + np.eye(M)) * scale_sqrt
        else:
            q_mu = np.zeros((M, R_))                                # layers.py:21
            q_sqrt = np.tile(np.eye(M)[None], [R_, 1, 1]) * scale_sqrt   # layers.py:24, build_models.py:276-278
        return _f32(ZZ), _f32(q_mu), _f32(q_sqrt)

    for _ in range(L - 1):
        A = np.zeros((D_in, D_out))
        A[:min(D_in, D_out), :min(D_in, D_out)] = np.eye(min(D_in, D_out))   # :204-207
        W = np.zeros((D_out, R))
        W[:, :min(R, Dx)] = P[:, :min(R, Dx)]                       # :216-217
        ZZ, q_mu, q_sqrt = gp_params(D_in, R, 0.3 if parity else 1e-5)
        layers.append(dict(type="gp", Z=ZZ, ls=_f32(np.full(D_in, float(D_in) ** 0.5)), var=1.0,
                           q_mu=q_mu, q_sqrt=q_sqrt, W=_f32(W), mf=("linear", _f32(A), np.zeros(D_out))))
        D_in = D_out
    ZZ, q_mu, q_sqrt = gp_params(D_in, Dy, 1.0)
    layers.append(dict(type="gp", Z=ZZ, ls=_f32(np.full(D_in, float(D_in) ** 0.5)), var=1.0,
                       q_mu=q_mu, q_sqrt=q_sqrt, W=None, mf=("zero",)))      # :238-241
    return dict(X=X, Y=Y, B=B, K=K, lik_var=0.01, layers=layers, n_data=n_data,
                name="L%d_M%d_K%d_B%d%s" % (L, M, K, B, "_LV" if with_lv else ""))


def make_noise(spec, seed=1, K=None, B=None):
    """One N(0,1) array per layer in the IW layout ([B,K,latent] for LV, [B,K,R] for GP layers)."""
    rng = np.random.default_rng(seed)
    B = spec["B"] if B is None else B
    K = spec["K"] if K is None else K
    zs = []
    for l in spec["layers"]:
        n = l["latent_dim"] if l["type"] == "lv" else l["q_mu"].shape[1]
        zs.append(_f32(rng.standard_normal((B, K, n))))
    return zs


def build_model(spec, device=None, cls=None, num_samples=None, minibatch=True):
    """Device model (this package's layers/models) from a spec.  X, Y = the first B rows."""
    import torch
    from . import settings
    from .features import InducingPoints, MixedKernelSharedMof
    from .kernels import RBF
    from .layers import Encoder, GPLayer, LatentVariableLayer
    from .likelihoods import Gaussian
    from .mean_functions import Linear
    from .models import DGP_IWVI
    from .temp_workaround import SharedMixedMok

    device = device or settings.default_device()
    t = lambda a: torch.as_tensor(np.asarray(a, np.float32), device=device)
    layers = []
    for l in spec["layers"]:
        if l["type"] == "lv":
            enc = Encoder(l["latent_dim"], l["dims"][0], l["dims"][1:-1])
            enc.Ws = [t(w) for w in l["enc_W"]]
            enc.bs = [t(b) for b in l["enc_b"]]
            layers.append(LatentVariableLayer(l["latent_dim"], encoder=enc))
            continue
        D_in = l["Z"].shape[1]
        kern = RBF(D_in, variance=l["var"], lengthscales=l["ls"], ARD=True)
        feat = InducingPoints(l["Z"])
        mf = Linear(l["mf"][1], l["mf"][2]) if l["mf"][0] == "linear" else None
        if l["W"] is not None:
            layer = GPLayer(SharedMixedMok(kern, l["W"]), MixedKernelSharedMof(feat), l["q_mu"].shape[1], mf)
        else:
            layer = GPLayer(kern, feat, l["q_mu"].shape[1], mf)
        layer.q_mu, layer.q_sqrt = t(l["q_mu"]), t(l["q_sqrt"])
        layers.append(layer.to(device))
    B = spec["B"]
    cls = cls or DGP_IWVI
    m = cls(spec["X"][:B], spec["Y"][:B], layers, Gaussian(spec["lik_var"]),
            num_samples=spec["K"] if num_samples is None else num_samples)
    m.num_data = spec["n_data"]
    return m.to(device)
