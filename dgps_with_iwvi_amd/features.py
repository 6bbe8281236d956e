"""Inducing features (GPflow-1.x ``gpflow.features`` / ``gpflow.multioutput`` counterparts).
Reference call sites: layers.py:28; experiments/build_models.py:222,241."""
import numpy as np
import torch

from . import settings


class InducingFeature:
    pass


class InducingPoints(InducingFeature):
    def __init__(self, Z, name=None):
        Z = torch.as_tensor(np.asarray(Z, dtype=np.float32) if not isinstance(Z, torch.Tensor) else Z)
        self.Z = Z.to(dtype=settings.float_type, device=settings.default_device()).contiguous().clone()
        self.name = name

    def __len__(self):
        return self.Z.shape[0]

    def to(self, device):
        self.Z = self.Z.to(device)
        return self


class MixedKernelSharedMof(InducingFeature):
    """Shared inducing points for linearly mixed latent GPs (wraps one InducingPoints)."""

    def __init__(self, feat, name=None):
        self.feat = feat
        self.name = name

    def __len__(self):
        return len(self.feat)

    def to(self, device):
        self.feat.to(device)
        return self
