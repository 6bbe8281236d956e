"""Build the fp64 oracle model from a ``dgps_with_iwvi_amd.synthetic.make_spec`` spec (test infrastructure)."""
import numpy as np

from . import iwvi_oracle as O


def build_oracle(spec, iw=True, num_samples=None, B=None):
    layers = []
    for l in spec["layers"]:
        if l["type"] == "lv":
            enc = O.Encoder(l["latent_dim"], l["dims"][0], l["dims"][1:-1])
            enc.Ws = [np.asarray(w, np.float64) for w in l["enc_W"]]
            enc.bs = [np.asarray(b, np.float64) for b in l["enc_b"]]
            layers.append(O.LatentVariableLayer(l["latent_dim"], encoder=enc))
            continue
        D_in = l["Z"].shape[1]
        kern = O.RBF(D_in, variance=l["var"], lengthscales=l["ls"])
        mf = O.Linear(l["mf"][1], l["mf"][2]) if l["mf"][0] == "linear" else None
        k = O.SharedMixedMok(kern, l["W"]) if l["W"] is not None else kern
        layer = O.GPLayer(k, l["Z"], l["q_mu"].shape[1], mf)
        layer.q_mu, layer.q_sqrt = l["q_mu"], l["q_sqrt"]
        layers.append(layer)
    B = spec["B"] if B is None else B
    cls = O.DGP_IWVI if iw else O.DGP_VI
    return cls(spec["X"][:B], spec["Y"][:B], layers, O.Gaussian(spec["lik_var"]),
               num_samples=spec["K"] if num_samples is None else num_samples, num_data=spec["n_data"])


def oracle_noise(spec, zs):
    """IW-layout noise -> what the oracle's layers expect (the final plain layer runs full_cov=True and
    its sample is never consumed: z=None)."""
    out = []
    for l, z in zip(spec["layers"], zs):
        if l["type"] == "gp" and l["W"] is None:
            out.append(None)
        else:
            out.append(z)
    return out
