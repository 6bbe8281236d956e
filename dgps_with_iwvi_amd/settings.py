"""The slice of ``gpflow.settings`` the hot path reads (reference: temp_workaround.py:39,89; layers.py:61-62).

float_type is fixed: per-sample arithmetic is float32 on the MFMA pipe, the inducing-set
factorisation (Gram, Cholesky, inverse) is float64 ("Precision").
"""
import contextlib
import os

import torch

float_type = torch.float32
jitter_level = 1e-6          # gpflow.settings.numerics.jitter_level default
seed = 0                     # Philox key of the on-device N(0,1) stream (iwvi_fill_normal)
_offset = 0                  # Philox counter; advanced by every draw
# Arithmetic of the R * M^2 contraction (stage 2 of the layer kernel) and of the adjoint chain's S_r products: split-f16 operands on
# v_mfma_f32_16x16x32_f16 (default; 22 operand mantissa bits) or fp32 MFMAs.  Passed PER CALL in the descriptors
# (iwvi_layer_desc.flags / iwvi_gp_bwd_desc.flags); the environment variables only set these defaults.
fw_f32_stage2 = bool(os.environ.get("IWVI_FW_F32_STAGE2"))
bw_f32_chain = bool(os.environ.get("IWVI_BW_F32_CHAIN"))


def default_device():
    if not torch.cuda.is_available():
        return torch.device("cpu")
    return torch.device("cuda", torch.cuda.current_device())


def next_noise_offset(n):
    """Reserve ``n`` normals of the Philox stream; returns the counter offset (in units of 4 normals)."""
    global _offset
    off = _offset
    _offset += (n + 3) // 4
    return off


def set_seed(s):
    global seed, _offset
    seed, _offset = int(s), 0


@contextlib.contextmanager
def temp_settings(jitter=None):
    """Counterpart of ``gpflow.settings.temp_settings`` (reference tests/test_gp_layer.py:81-83)."""
    global jitter_level
    old = jitter_level
    if jitter is not None:
        jitter_level = float(jitter)
    try:
        yield
    finally:
        jitter_level = old
