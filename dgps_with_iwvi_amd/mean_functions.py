"""Mean functions GPLayer.propagate adds to samples and mean (reference layers.py:46-48).
On the hot path they are evaluated inside ``iwvi_gp_layer_forward``'s epilogue and these classes only carry parameters; ``__call__``
(GPflow 1.x ``MeanFunction.__call__``: what the reference's ``self.mean_function(F)`` at layers.py:46 invokes) exists for user code
that calls a mean function itself -- e.g. a custom layer following the layer protocol -- and runs as torch ops on the tensor's device."""
import numpy as np
import torch

from . import _abi, settings


class MeanFunction:
    mf_type = _abi.MF_ZERO
    A = None
    b = None

    def to(self, device):
        return self

    def __call__(self, X):
        raise NotImplementedError


class Zero(MeanFunction):
    def __init__(self, output_dim=1):
        self.output_dim = output_dim

    def __call__(self, X):
        """GPflow 1.x ``Zero``: zeros of shape ``X.shape[:-1] + [output_dim]`` (default 1: it broadcasts against any P)."""
        return torch.zeros(*X.shape[:-1], self.output_dim, dtype=X.dtype, device=X.device)


class Identity(MeanFunction):
    mf_type = _abi.MF_IDENTITY

    def __init__(self, input_dim=None):
        self.input_dim = input_dim

    def __call__(self, X):
        return X


class Linear(MeanFunction):
    """X A + b on the last axis; A [D_in, D_out], b [D_out] (zeros by default)."""
    mf_type = _abi.MF_LINEAR

    def __init__(self, A=None, b=None):
        A = np.ones((1, 1)) if A is None else A
        A = torch.as_tensor(np.asarray(A, dtype=np.float32) if not isinstance(A, torch.Tensor) else A)
        self.A = A.to(dtype=settings.float_type, device=settings.default_device()).contiguous().clone()
        if b is None:
            b = torch.zeros(self.A.shape[1])
        b = torch.as_tensor(np.asarray(b, dtype=np.float32) if not isinstance(b, torch.Tensor) else b)
        self.b = b.to(dtype=settings.float_type, device=self.A.device).reshape(-1).contiguous().clone()

    def to(self, device):
        self.A, self.b = self.A.to(device), self.b.to(device)
        return self

    def __call__(self, X):
        """``X A + b`` on the last axis, any number of leading axes (GPflow 1.x ``Linear``)."""
        return torch.matmul(X, self.A.to(device=X.device, dtype=X.dtype)) + self.b.to(device=X.device, dtype=X.dtype)
