"""Gaussian likelihood (reference: experiments/build_models.py:198-199; models.py:66,105,134).
On the hot path ``variational_expectations`` is fused into the tail of ``iwvi_dgp_forward``; the method here is the
reference's callable form on explicit moments (``iwvi_gaussian_var_exp``)."""
import torch

from . import _abi, settings
from .kernels import DeviceScalarVariance


class Gaussian(DeviceScalarVariance):
    def __init__(self, variance=1.0, name=None):
        self.variance = float(variance)
        self.name = name

    def variational_expectations(self, Fmu, Fvar, Y):
        """E_{N(f; Fmu, Fvar)} log N(Y; f, variance), elementwise (gpflow 1.x Gaussian; called at models.py:66,134)."""
        Fmu = _abi.dev_tensor(Fmu.contiguous(), "Fmu")
        Fvar = _abi.dev_tensor(Fvar.expand_as(Fmu).contiguous(), "Fvar")
        Y = _abi.dev_tensor(torch.as_tensor(Y, dtype=settings.float_type, device=Fmu.device).expand_as(Fmu).contiguous(), "Y")
        out = torch.empty_like(Fmu)
        Dy = Fmu.shape[-1] if Fmu.dim() else 1
        T = Fmu.numel() // max(Dy, 1)
        _abi.check(_abi.lib().iwvi_gaussian_var_exp(_abi.ptr(Fmu), _abi.ptr(Fvar), _abi.ptr(Y), self.variance,
                                                    T, Dy, 1, max(T, 1), _abi.ptr(out), _abi.stream_ptr()))
        return out

    def predict_mean_and_var(self, Fmu, Fvar):
        return Fmu, Fvar + self.variance
