#!/usr/bin/env python3
"""Generate the golden input/output vectors under tests/golden/ (run from the repo root).

The reference (TensorFlow 1.x + GPflow 1.x) cannot be imported in this container,
so these vectors are produced by the fp64 oracle (oracle/iwvi_oracle.py), which is pinned by the
closed-form identities of tests/test_oracle_pinning.py.  They serve two purposes:
  * `-m "not gpu"`: guard the oracle against drift (tests/test_golden.py::test_oracle_reproduces_golden);
  * `-m gpu`: fixed known-answer cases for the HIP path (tests/test_gpu_parity.py::test_golden_*).
A fixture is data only: the model parameters, the minibatch, the injected N(0,1) noise, and the oracle's
per-layer means / variances / samples, per-sample log-weights L_NK, per-point logp and the IW-ELBO.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from dgps_with_iwvi_amd import synthetic            # noqa: E402  (spec builder: plain NumPy)
from oracle.from_spec import build_oracle, oracle_noise   # noqa: E402

CASES = {
    # name: make_spec kwargs
    "tiny_L2_lv": dict(L=2, M=8, B=4, K=3, Dx=2, R=2, with_lv=True, seed=11),
    "mid_L2_lv": dict(L=2, M=32, B=16, K=4, Dx=8, R=5, with_lv=True, seed=12),
    "ragged_L3": dict(L=3, M=40, B=7, K=5, Dx=5, R=3, with_lv=False, seed=13),
    "k1_L2": dict(L=2, M=16, B=33, K=1, Dx=3, R=2, with_lv=False, seed=14),
}


GRAD_CASES = ("tiny_L2_lv", "mid_L2_lv")


def spec_to_arrays(spec):
    out = dict(X=spec["X"][:spec["B"]], Y=spec["Y"][:spec["B"]], B=spec["B"], K=spec["K"],
               lik_var=spec["lik_var"], n_data=spec["n_data"], n_layers=len(spec["layers"]))
    for i, l in enumerate(spec["layers"]):
        p = "l%d_" % i
        out[p + "type"] = l["type"]
        if l["type"] == "lv":
            out[p + "latent_dim"] = l["latent_dim"]
            out[p + "dims"] = np.asarray(l["dims"])
            for j, (w, b) in enumerate(zip(l["enc_W"], l["enc_b"])):
                out[p + "encW%d" % j], out[p + "encb%d" % j] = w, b
        else:
            for k in ("Z", "ls", "q_mu", "q_sqrt"):
                out[p + k] = l[k]
            out[p + "var"] = l["var"]
            out[p + "has_W"] = l["W"] is not None
            if l["W"] is not None:
                out[p + "W"] = l["W"]
            out[p + "mf"] = l["mf"][0]
            if l["mf"][0] == "linear":
                out[p + "mfA"], out[p + "mfb"] = l["mf"][1], l["mf"][2]
    return out


def arrays_to_spec(a):
    """Inverse of spec_to_arrays (used by the tests)."""
    layers = []
    for i in range(int(a["n_layers"])):
        p = "l%d_" % i
        if str(a[p + "type"]) == "lv":
            dims = [int(d) for d in a[p + "dims"]]
            n = len(dims) - 1
            layers.append(dict(type="lv", latent_dim=int(a[p + "latent_dim"]), dims=dims,
                               enc_W=[a[p + "encW%d" % j] for j in range(n)],
                               enc_b=[a[p + "encb%d" % j] for j in range(n)]))
        else:
            mf = ("linear", a[p + "mfA"], a[p + "mfb"]) if str(a[p + "mf"]) == "linear" else ("zero",)
            layers.append(dict(type="gp", Z=a[p + "Z"], ls=a[p + "ls"], var=float(a[p + "var"]),
                               q_mu=a[p + "q_mu"], q_sqrt=a[p + "q_sqrt"],
                               W=a[p + "W"] if bool(a[p + "has_W"]) else None, mf=mf))
    return dict(X=a["X"], Y=a["Y"], B=int(a["B"]), K=int(a["K"]), lik_var=float(a["lik_var"]),
                n_data=int(a["n_data"]), layers=layers, name="golden")


def oracle_outputs(spec, zs):
    m = build_oracle(spec)
    L_NK, global_kls, means, covs, samples = m.log_weights(oracle_noise(spec, zs))
    K = spec["K"]
    mx = L_NK.max(1, keepdims=True)
    logp = mx[:, 0] + np.log(np.exp(L_NK - mx).sum(1)) - np.log(K)
    out = dict(L_NK=L_NK, logp=logp, elbo=m.build_likelihood(oracle_noise(spec, zs)),
               global_kls=np.asarray(global_kls, np.float64))
    for i, (mu, cv, s) in enumerate(zip(means, covs, samples)):
        out["mean%d" % i] = mu
        # the final plain-kernel layer returns [B, Dy, K, K]; store its diagonal [B, K, Dy] (models.py:133)
        out["var%d" % i] = np.diagonal(cv, axis1=-2, axis2=-1).transpose(0, 2, 1) if cv.ndim == 4 else cv
        if i < len(samples) - 1:
            out["sample%d" % i] = s
    return out


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    for name, kw in CASES.items():
        spec = synthetic.make_spec(parity=True, **kw)
        zs = synthetic.make_noise(spec, seed=kw["seed"] + 100)
        arrays = spec_to_arrays(spec)
        for i, z in enumerate(zs):
            arrays["z%d" % i] = z
        for k, v in oracle_outputs(spec, zs).items():
            arrays["out_" + k] = v
        path = os.path.join(here, name + ".npz")
        np.savez_compressed(path, **arrays)
        print("%-12s %6.1f KB  elbo=%.6f" % (name, os.path.getsize(path) / 1024, arrays["out_elbo"]))
        if name in GRAD_CASES:
            # target vectors for the backward pass: d ELBO / d parameters from the
            # gradient oracle (torch autodiff of the fp64 restatement, pinned by finite differences)
            from oracle.grad_oracle import iw_elbo_and_gradients
            val, grads = iw_elbo_and_gradients(spec, zs)
            gpath = os.path.join(here, "grad_" + name + ".npz")
            np.savez_compressed(gpath, elbo=val, **{k.replace(".", "_"): v for k, v in grads.items()})
            print("%-12s %6.1f KB  (gradients, %d arrays)" % ("grad_" + name, os.path.getsize(gpath) / 1024, len(grads)))


if __name__ == "__main__":
    main()
