"""Mean functions GPLayer.propagate adds to samples and mean (reference layers.py:46-48).
They are evaluated inside ``iwvi_gp_layer_forward``'s epilogue; these classes only carry parameters."""
import numpy as np
import torch

from . import _abi, settings


class MeanFunction:
    mf_type = _abi.MF_ZERO
    A = None
    b = None

    def to(self, device):
        return self


class Zero(MeanFunction):
    def __init__(self, output_dim=1):
        self.output_dim = output_dim


class Identity(MeanFunction):
    mf_type = _abi.MF_IDENTITY

    def __init__(self, input_dim=None):
        self.input_dim = input_dim


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
