"""Stationary kernels the path supports (GPflow-1.x ``gpflow.kernels`` counterparts).

Only parameter holders: every evaluation happens inside the HIP kernels (K_uu in float64 by
``iwvi_gp_precompute``, K_uf in float32 fused into ``iwvi_gp_layer_forward``).
Reference call sites: temp_workaround.py:39,44,45; experiments/build_models.py:211-214,238.
"""
import numpy as np
import torch

from . import _abi, settings


def _as_param(x, shape=None, device=None):
    t = torch.as_tensor(np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x)
    t = t.to(dtype=settings.float_type, device=device or settings.default_device())
    if shape is not None:
        t = t.expand(shape)
    return t.contiguous().clone()


class DeviceScalarVariance:
    """``variance`` as a host float that may be BOUND to a 1-element device tensor (``bind_device_variance``): a trainer then
    updates it on the device, every kernel launch reads it there (``*_desc.variance_dev``), and the host copy is refreshed
    lazily, only when somebody reads ``.variance`` -- no device-to-host copy per training step, and a captured hipGraph of a
    step stays valid while the value changes."""
    _var_dev = None
    _var_stale = False

    @property
    def variance(self):
        if self._var_dev is not None and self._var_stale:
            self._var_host = float(self._var_dev.item())
            self._var_stale = False
        return self._var_host

    @variance.setter
    def variance(self, v):
        self._var_host = float(v)
        if self._var_dev is not None:
            self._var_dev.fill_(self._var_host)
        self._var_stale = False

    def bind_device_variance(self, t):
        """``t``: 1-element float32 device tensor holding the current value; updated in place by the owner."""
        self._var_dev = t
        self._var_stale = False

    def mark_device_variance_changed(self):
        self._var_stale = self._var_dev is not None

    def desc_variance(self):
        """(by-value float for a descriptor -- possibly stale when bound --, device pointer or None)."""
        return self._var_host, (None if self._var_dev is None else self._var_dev.data_ptr())


class Stationary(DeviceScalarVariance):
    kern_type = None

    def __init__(self, input_dim, variance=1.0, lengthscales=1.0, ARD=False, active_dims=None, name=None):
        if active_dims is not None:
            raise NotImplementedError("active_dims is not on the reference's hot path")
        self.input_dim = int(input_dim)
        if self.input_dim > _abi.MAX_D:
            raise ValueError("input_dim %d > %d" % (self.input_dim, _abi.MAX_D))
        self.variance = float(variance)      # host scalar (fixed in the reference's experiments)
        self.ARD = ARD
        self.lengthscales = _as_param(lengthscales, (self.input_dim,))
        self.name = name

    def to(self, device):
        self.lengthscales = self.lengthscales.to(device)
        return self

    def K(self, X, X2=None):
        """K(Z, Z) in float64 through ``iwvi_rbf_gram_sym`` (jitter-free); X2 must be None."""
        if X2 is not None:
            raise NotImplementedError("cross-Grams are fused into iwvi_gp_layer_forward and never materialised")
        X = _abi.dev_tensor(torch.as_tensor(X, dtype=settings.float_type).contiguous(), "X")
        M, D = X.shape
        out = torch.empty(M, M, dtype=torch.float64, device=X.device)
        _abi.check(_abi.lib().iwvi_rbf_gram_sym(_abi.ptr(X), _abi.ptr(self.lengthscales), self.variance, 0.0,
                                                self.kern_type, M, D, _abi.ptr(out), _abi.stream_ptr()))
        return out

    def Kdiag(self, X):
        X = torch.as_tensor(X)
        return torch.full(X.shape[:-1], self.variance, dtype=settings.float_type, device=X.device)


class RBF(Stationary):
    kern_type = _abi.KERN_RBF


SquaredExponential = RBF


class Matern52(Stationary):
    kern_type = _abi.KERN_MATERN52
