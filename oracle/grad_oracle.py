"""Gradient oracle for SURVEY.md section 8 row F1 (the backward pass of the IW-ELBO; not built yet on the HIP side).
TEST INFRASTRUCTURE ONLY.

d IW-ELBO / d theta by reverse-mode autodiff (torch, CPU, float64) through the op-for-op restatement of the
reference's graph (oracle/ref_torch_cpu.py; the reference obtains its gradients the same way, from TensorFlow's
autodiff of that graph, experiments/build_models.py:284-304).  The noise of every layer is an explicit input, so
the reparameterised gradient is deterministic.  Pinned by central finite differences of the independent NumPy
restatement (tests/test_grad_oracle.py).

Parameters differentiated, per layer:  GP: Z [M, D], lengthscales [D], kernel variance, q_mu [M, R],
q_sqrt [R, M, M] (lower triangle), mixing matrix W [P, R] and linear mean function A [D, P] where present;  LV: encoder weights and biases;  plus the likelihood variance."""
import numpy as np
import torch

from .ref_torch_cpu import CpuDGP


def iw_elbo_and_gradients(spec, zs, mode_vi=False):
    """Returns (elbo, {name: ndarray}); names: 'l<i>.Z', 'l<i>.ls', 'l<i>.var', 'l<i>.q_mu', 'l<i>.q_sqrt', 'l<i>.W', 'l<i>.mfA',
    'l<i>.encW<j>', 'l<i>.encb<j>', 'lik_var'."""
    m = CpuDGP(spec, torch.float64)
    params = {}

    def leaf(x):
        t = torch.as_tensor(np.asarray(x, dtype=np.float64)).clone().requires_grad_(True)
        return t

    for i, L in enumerate(m.layers):
        if L["type"] == "lv":
            L["W"] = [leaf(w.detach().numpy()) for w in L["W"]]
            L["b"] = [leaf(b.detach().numpy()) for b in L["b"]]
            for j, (w, b) in enumerate(zip(L["W"], L["b"])):
                params["l%d.encW%d" % (i, j)], params["l%d.encb%d" % (i, j)] = w, b
        else:
            for k in ("Z", "ls", "q_mu"):
                L[k] = leaf(L[k].detach().numpy())
                params["l%d.%s" % (i, k)] = L[k]
            raw = leaf(L["q_sqrt"].detach().numpy())
            L["q_sqrt"] = torch.tril(raw)                      # gradient lands on the lower triangle only
            params["l%d.q_sqrt" % i] = raw
            L["var"] = leaf(L["var"])
            params["l%d.var" % i] = L["var"]
            if L["W"] is not None:                             # trainable when the reference runs with fix_linear=False
                L["W"] = leaf(L["W"].detach().numpy())
                params["l%d.W" % i] = L["W"]
            if L["A"] is not None:
                L["A"] = leaf(L["A"].detach().numpy())
                params["l%d.mfA" % i] = L["A"]
    m.lik_var = leaf(m.lik_var)
    params["lik_var"] = m.lik_var
    val = m.elbo_tensor(zs, mode_vi=mode_vi)      # mode_vi: the DGP_VI bound (models.py:49-86), same noise layout
    val.backward()
    grads = {k: (np.zeros(tuple(v.shape)) if v.grad is None else v.grad.detach().numpy().copy()) for k, v in params.items()}
    return float(val.detach()), grads
