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
# float64 stage-1 route of a GP layer (iwvi_layer_desc.flags & IWVI_LAYER_F64_STAGE1 / iwvi_gp_desc.flags & IWVI_GP_F64_STAGE1): K_uf,
# a = Lm^-1 k and sigma^2 - |a|^2 in float64 -- what the reference's float_type = float64 gives (temp_workaround.py:39-59) -- for layers whose
# K_uu is ill-conditioned.  "auto" (default): a layer takes it when its INPUT dimension is <= f64_auto_max_dim (inducing points crowd a
# 1-3-dimensional box: cond(Lm) ~ 1e4, float32 loses 1e-2 .. 1e-1 of the mean there; the reference's own tests and demo are 1-D) --
# a static rule, so no launch ever waits for a condition estimate; "on" / "off" force it for every layer; GPLayer.f64_stage1 = True / False
# overrides per layer, and models.DGP_VI.autotune_f64() records a MEASURED choice per layer from the diag(Lm) ratio of the current parameters
# (it ranks below the layer's explicit True / False and above the global setting; training.Trainer and build_models.build_model call it --
# at construction and at every staircase epoch -- while this setting is "auto").
# The 8-dimensional BASELINE stacks never take it (tests/test_gpu_f64_route.py asserts the variant bits).
f64_stage1 = os.environ.get("IWVI_F64_STAGE1", "auto")
f64_auto_max_dim = 3


def use_f64_stage1(input_dim, override=None, measured=None):
    """The rule above for one GP layer of input dimension ``input_dim`` (``override``: the layer's own True / False / None;
    ``measured``: what ``autotune_f64`` found for the layer's current parameters, or None)."""
    if override is not None:
        return bool(override)
    if measured is not None:
        return bool(measured)
    if f64_stage1 == "on":
        return True
    if f64_stage1 == "off":
        return False
    if f64_stage1 != "auto":
        raise ValueError("settings.f64_stage1 must be 'auto', 'on' or 'off', got %r" % (f64_stage1,))
    return int(input_dim) <= f64_auto_max_dim
bw_f32_chain = bool(os.environ.get("IWVI_BW_F32_CHAIN"))


def split16_variance_ok(num_inducing, variance):
    """False for a GP layer whose launch must take the fp32-MFMA variant (``IWVI_LAYER_F32_STAGE2``) whatever ``fw_f32_stage2`` says: M > 240
    and kernel variance / jitter >= 2^30.  The split-f16 image of the super-block inverses holds 2^ceil(log2 sigma) (L_II)^-1 as f16 pairs;
    its entries are bounded by sigma / sqrt(jitter) up to a factor 2, which leaves the f16 range beyond that ratio (the packer would saturate).
    ``variance``: the host copy (possibly a step stale for a trained, device-resident variance -- the bound has a factor 2 of slack)."""
    return int(num_inducing) <= 240 or float(variance) < 2.0 ** 30 * float(jitter_level)


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
def temp_settings(jitter=None, f64_stage1=None):
    """Counterpart of ``gpflow.settings.temp_settings`` (reference tests/test_gp_layer.py:81-83)."""
    global jitter_level
    old, old64 = jitter_level, globals()["f64_stage1"]
    if jitter is not None:
        jitter_level = float(jitter)
    if f64_stage1 is not None:
        globals()["f64_stage1"] = f64_stage1
    try:
        yield
    finally:
        jitter_level = old
        globals()["f64_stage1"] = old64
