"""Backward pass of the IW-ELBO path: host side of ``csrc/backward.hip``.

The reference obtains its gradients from TensorFlow's autodiff of the graph of ``models.py:112-150``
(``experiments/build_models.py:284-304``); here every layer has a hand-written adjoint behind the C-ABI
(``iwvi_gp_layer_backward`` ...), driven layer by layer in reverse.  First version: correct and
deterministic, intermediates in HBM; not fused like the forward yet."""
import ctypes

import torch

from . import _abi, settings
from .layers import GPLayer, SharedMixedMok
from .temp_workaround import precompute_states


class GpSaved:
    """What one GP layer's forward leaves for its adjoint."""
    __slots__ = ("F", "noise", "A", "U", "sample", "mean", "var", "T")


def _words(device):
    return torch.zeros(4, dtype=torch.int64, device=device)


def gp_forward_saved(layer, F, z=None, words=None):
    """``GPLayer.propagate`` on per-sample rows F [T, D] (marginal variances) that also keeps a = Lm^-1 k,
    u_r = L_r^T a and the draws.  Factorises with IWVI_GP_WANT_DENSE: the adjoint reads the dense Lm, Lm^-1."""
    if not isinstance(layer, GPLayer):
        raise TypeError("gp_forward_saved needs a GPLayer")
    F = _abi.dev_tensor(F.contiguous(), "F")
    T, D = F.shape
    dev = F.device
    d = layer.state_desc()
    d.flags = _abi.GP_WANT_DENSE
    precompute_states([d])
    R, Mp = layer.num_outputs, layer.state().Mp
    P = layer.kern.W.shape[0] if isinstance(layer.kern, SharedMixedMok) else R
    s = GpSaved()
    s.F, s.T = F, T
    s.A = torch.empty(T, Mp, dtype=settings.float_type, device=dev)
    s.U = torch.empty(R, T, Mp, dtype=settings.float_type, device=dev)
    s.noise = torch.empty(T, R, dtype=settings.float_type, device=dev)
    s.sample, s.mean, s.var = (torch.empty(T, P, dtype=settings.float_type, device=dev) for _ in range(3))
    z2 = None if z is None else _abi.dev_tensor(z.reshape(T, R).contiguous(), "z")
    outs = dict(sample=s.sample, mean=s.mean, var=s.var, a_out=s.A, u_out=s.U, noise_out=s.noise)
    ld, keep = layer.fused_desc(z2, outs)
    descs = (_abi.LayerDesc * 1)(ld)
    words = _words(dev) if words is None else words
    _abi.check(_abi.lib().iwvi_dgp_forward(
        descs, 1, _abi.ptr(F), D, None, 0, None, 0, T, 1, T, 1.0,
        settings.seed, ctypes.c_void_p(words.data_ptr() + 8), None, None, _abi.stream_ptr()))
    return s


def gp_backward(layer, saved, d_sample=None, d_mean=None, d_var=None, kl_weight=1.0, want_dF=True):
    """``iwvi_gp_layer_backward``: upstream gradients [T, P] -> dict(dF [T, D], dZ, dls, dvariance, dq_mu, dq_sqrt)."""
    dev = saved.F.device
    T, D = saved.F.shape
    M, R = layer.num_inducing, layer.num_outputs
    kern = layer._base_kern()
    W = _abi.dev_tensor(layer.kern.W, "W") if isinstance(layer.kern, SharedMixedMok) else None
    P = W.shape[0] if W is not None else R
    ft = settings.float_type
    out = dict(dZ=torch.empty(M, D, dtype=ft, device=dev), dls=torch.empty(D, dtype=ft, device=dev),
               dvariance=torch.empty(1, dtype=ft, device=dev), dq_mu=torch.empty(M, R, dtype=ft, device=dev),
               dq_sqrt=torch.empty(R, M, M, dtype=ft, device=dev))
    if want_dF:
        out["dF"] = torch.empty(T, D, dtype=ft, device=dev)
    b = _abi.GpBwdDesc()
    Z, q_mu, q_sqrt = (_abi.dev_tensor(t.contiguous(), n) for t, n in ((layer._Z(), "Z"), (layer.q_mu, "q_mu"), (layer.q_sqrt, "q_sqrt")))
    b.state, b.Z, b.lengthscales = layer.state().buf.data_ptr(), Z.data_ptr(), kern.lengthscales.data_ptr()
    b.q_mu, b.q_sqrt, b.variance = q_mu.data_ptr(), q_sqrt.data_ptr(), kern.variance
    b.M, b.D, b.R, b.P, b.kern_type = M, D, R, P, kern.kern_type
    mf = layer.mean_function
    b.mf_type = mf.mf_type
    keep = [Z, q_mu, q_sqrt, W]
    if W is not None:
        b.W = W.data_ptr()
    if mf.mf_type == _abi.MF_LINEAR:
        b.mf_A = _abi.dev_tensor(mf.A, "mean_function.A").data_ptr()
    b.F, b.noise, b.A, b.U = saved.F.data_ptr(), saved.noise.data_ptr(), saved.A.data_ptr(), saved.U.data_ptr()
    for name, t in (("d_sample", d_sample), ("d_mean", d_mean), ("d_var", d_var)):
        if t is not None:
            t = _abi.dev_tensor(t.reshape(T, P).contiguous(), name)
            keep.append(t)
            setattr(b, name, t.data_ptr())
    b.kl_weight = float(kl_weight)
    for k, t in out.items():
        setattr(b, k, t.data_ptr())
    ws = torch.empty(_abi.lib().iwvi_gp_layer_backward_ws_bytes(T, M, D, R), dtype=torch.uint8, device=dev)
    _abi.check(_abi.lib().iwvi_gp_layer_backward(ctypes.byref(b), T, ws.data_ptr(), _abi.stream_ptr()))
    return out
