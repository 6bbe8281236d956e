"""ctypes binding of libiwvi_hip.so (the C-ABI declared in include/iwvi_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C dgps_with_iwvi_amd/csrc``.
There is deliberately NO fallback: if the shared object is missing, or a tensor is not on a
ROCm device, every compute entry point raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libiwvi_hip.so")

KERN_RBF, KERN_MATERN52 = 0, 1
LAYER_GP, LAYER_LV = 0, 1
ABI_VERSION = 17
GP_WANT_DENSE = 1
GP_WANT_LM = 2
GP_REUSE_FACTOR = 8
GP_FACTOR_ONLY = 16
LAYER_F32_STAGE2 = 1        # iwvi_layer_desc.flags
LAYER_F64_STAGE1 = 2        # iwvi_layer_desc.flags: K_uf, Lm^-1 k, sigma^2 - |a|^2 of the layer in float64
GP_F64_STAGE1 = 4           # iwvi_gp_desc.flags: prepare the state for it (dense float64 Lm^-1, plain z~)
BW_F32_CHAIN = 1            # iwvi_gp_bwd_desc.flags
BW_OWN_QSCALE = 4
ADAM_GRAD_F64 = 16
MAX_STACK = 8
MF_ZERO, MF_IDENTITY, MF_LINEAR = 0, 1, 2
ACT_TANH, ACT_RELU, ACT_SIGMOID, ACT_SOFTPLUS, ACT_IDENTITY = 0, 1, 2, 3, 4
MAX_LAYERS, MAX_R, MAX_P, MAX_D, MAX_M, MAX_KL, MAX_ENC = 8, 32, 32, 32, 512, 4, 8

c_void_p, c_int, c_int64, c_float, c_double, c_size_t = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_size_t)


ERR_ARG, ERR_LAUNCH, ERR_UNSUPPORTED = -1, -2, -3      # enum iwvi_status (include/iwvi_hip.h)


class IwviError(RuntimeError):
    """A non-zero status of the library; ``rc`` is the IWVI_ERR_* code (None when raised by the host side itself)."""

    def __init__(self, msg, rc=None):
        super().__init__(msg)
        self.rc = rc


class GpDesc(ctypes.Structure):
    """struct iwvi_gp_desc (include/iwvi_hip.h)."""
    _fields_ = [("Z", c_void_p), ("lengthscales", c_void_p), ("q_mu", c_void_p),
                ("q_sqrt", c_void_p), ("state", c_void_p), ("variance", c_float),
                ("jitter", c_double), ("M", ctypes.c_int32), ("D", ctypes.c_int32),
                ("R", ctypes.c_int32), ("kern_type", ctypes.c_int32), ("flags", ctypes.c_int32),
                ("variance_dev", c_void_p)]


class EncDesc(ctypes.Structure):
    """struct iwvi_enc_desc (include/iwvi_hip.h): an encoder evaluated inside the precompute launch."""
    _fields_ = [("XY", c_void_p), ("rows", c_int64), ("enc_W", ctypes.POINTER(c_void_p)),
                ("enc_b", ctypes.POINTER(c_void_p)), ("dims", ctypes.POINTER(ctypes.c_int32)),
                ("n_enc", ctypes.c_int32), ("latent_dim", ctypes.c_int32), ("out", c_void_p),
                ("X", c_void_p), ("Dx", ctypes.c_int32), ("K", ctypes.c_int32), ("sampled_kl", ctypes.c_int32),
                ("layer_index", ctypes.c_int32), ("seed", ctypes.c_uint64), ("rng_state", c_void_p),
                ("sample_X", c_void_p), ("sample_kl", c_void_p), ("sample_z", c_void_p), ("act", ctypes.c_int32)]


class LayerDesc(ctypes.Structure):
    """struct iwvi_layer_desc (include/iwvi_hip.h): one layer of the fused forward."""
    _fields_ = [("type", ctypes.c_int32), ("state", c_void_p),
                ("M", ctypes.c_int32), ("D", ctypes.c_int32), ("R", ctypes.c_int32), ("P", ctypes.c_int32),
                ("kern_type", ctypes.c_int32), ("mf_type", ctypes.c_int32), ("variance", c_float),
                ("W", c_void_p), ("mf_A", c_void_p), ("mf_b", c_void_p),
                ("enc_W", ctypes.POINTER(c_void_p)), ("enc_b", ctypes.POINTER(c_void_p)),
                ("enc_dims", ctypes.POINTER(ctypes.c_int32)),
                ("n_enc", ctypes.c_int32), ("latent_dim", ctypes.c_int32), ("sampled_kl", ctypes.c_int32),
                ("enc_out", c_void_p),
                ("noise", c_void_p), ("zero_noise", ctypes.c_int32), ("noise_out", c_void_p),
                ("sample", c_void_p), ("mean", c_void_p), ("var", c_void_p), ("kl_local", c_void_p),
                ("a_out", c_void_p), ("u_out", c_void_p), ("gmv_out", c_void_p), ("variance_dev", c_void_p),
                ("enc_act", ctypes.c_int32), ("flags", ctypes.c_int32)]


class ElboDesc(ctypes.Structure):
    """struct iwvi_elbo_desc (include/iwvi_hip.h): the reduction fused into the tail of the forward launch."""
    _fields_ = [("B", c_int64), ("K", ctypes.c_int32), ("stride_b", c_int64), ("stride_k", c_int64),
                ("kl_global", ctypes.POINTER(c_void_p)), ("kl_global_counts", ctypes.POINTER(ctypes.c_int32)),
                ("n_glob", ctypes.c_int32), ("scale", c_double), ("K_total", ctypes.c_int32),
                ("mode_vi", ctypes.c_int32), ("out_lse_ms", c_void_p), ("out_logp", c_void_p),
                ("out_elbo", c_void_p), ("ws", c_void_p), ("lw_init", c_void_p), ("noise_layer_base", ctypes.c_int32),
                ("x_per_sample", ctypes.c_int32), ("lik_variance_dev", c_void_p),
                ("adj_w", c_void_p), ("adj_dmean", c_void_p), ("adj_dvar", c_void_p), ("adj_sums", c_void_p)]


class GpBwdDesc(ctypes.Structure):
    """struct iwvi_gp_bwd_desc (include/iwvi_hip.h): adjoint of one GP layer."""
    _fields_ = [("state", c_void_p), ("Z", c_void_p), ("lengthscales", c_void_p), ("q_mu", c_void_p),
                ("q_sqrt", c_void_p), ("variance", c_float),
                ("M", ctypes.c_int32), ("D", ctypes.c_int32), ("R", ctypes.c_int32), ("P", ctypes.c_int32),
                ("kern_type", ctypes.c_int32), ("W", c_void_p), ("mf_type", ctypes.c_int32), ("mf_A", c_void_p),
                ("F", c_void_p), ("noise", c_void_p), ("A", c_void_p), ("U", c_void_p), ("GMV", c_void_p),
                ("d_sample", c_void_p), ("d_mean", c_void_p), ("d_var", c_void_p), ("kl_weight", c_double),
                ("dF", c_void_p), ("dZ", c_void_p), ("dls", c_void_p), ("dvariance", c_void_p),
                ("dq_mu", c_void_p), ("dq_sqrt", c_void_p), ("dW", c_void_p), ("dmf_A", c_void_p),
                ("side_stream", c_void_p), ("side_stream2", c_void_p), ("prepared", ctypes.c_int32),
                ("variance_dev", c_void_p), ("phase", ctypes.c_int32), ("flags", ctypes.c_int32)]


class AdamTensor(ctypes.Structure):
    """struct iwvi_adam_tensor (include/iwvi_hip.h)."""
    _fields_ = [("param", c_void_p), ("grad", c_void_p), ("x", c_void_p), ("m", c_void_p), ("v", c_void_p),
                ("n", c_int64), ("transform", ctypes.c_int32)]


# name -> (restype, argtypes); every symbol include/iwvi_hip.h declares
PROTOTYPES = {
    "iwvi_version": (c_int, []),
    "iwvi_debug_set_option": (c_int, [ctypes.c_char_p, c_int]),
    "iwvi_debug_last_forward_variant": (c_int, []),
    "iwvi_debug_set_stamps": (None, [c_void_p, c_int64]),
    "iwvi_debug_set_pre_stamps": (None, [c_void_p]),
    "iwvi_debug_set_exit": (None, [c_int]),
    "iwvi_last_error": (ctypes.c_char_p, []),
    "iwvi_gp_state_bytes": (c_size_t, [c_int, c_int]),
    "iwvi_gp_state_offsets": (c_int, [c_int, c_int, ctypes.POINTER(c_size_t)]),
    "iwvi_gp_precompute": (c_int, [ctypes.POINTER(GpDesc), c_int, c_void_p]),
    "iwvi_gp_dense_inverse": (c_int, [ctypes.POINTER(GpDesc), c_int, c_void_p]),
    "iwvi_model_precompute": (c_int, [ctypes.POINTER(GpDesc), c_int, ctypes.POINTER(EncDesc), c_int, c_void_p]),
    "iwvi_rbf_gram_sym": (c_int, [c_void_p, c_void_p, c_float, c_double, c_int, c_int, c_int,
                                  c_void_p, c_void_p]),
    "iwvi_chol_ws_bytes": (c_size_t, [c_int]),
    "iwvi_chol_factor": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "iwvi_gp_layer_backward_ws_bytes": (c_size_t, [c_int64, c_int, c_int, c_int]),
    "iwvi_gp_layer_backward": (c_int, [ctypes.POINTER(GpBwdDesc), c_int64, c_void_p, c_void_p]),
    "iwvi_gp_layer_backward_needs_u": (c_int, [c_int64, c_int, c_int, c_int, c_int]),
    "iwvi_gp_layer_backward_prepare": (c_int, [ctypes.POINTER(GpBwdDesc), c_int64, c_void_p, c_void_p]),
    "iwvi_gp_layers_backward_prepare": (c_int, [ctypes.POINTER(GpBwdDesc), c_int, c_int64, ctypes.POINTER(c_void_p), c_void_p]),
    "iwvi_iw_elbo_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.POINTER(c_void_p),
                                      ctypes.POINTER(ctypes.c_int32), c_int, c_int64, c_int, c_float, c_double, c_int,
                                      c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int,
                                      c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "iwvi_iw_elbo_backward_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.POINTER(c_void_p),
                                          ctypes.POINTER(ctypes.c_int32), c_int, c_int64, c_int, c_float, c_void_p, c_double, c_int,
                                          c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int,
                                          c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "iwvi_lv_layer_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                       c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "iwvi_encoder_backward_ws_bytes": (c_size_t, [c_int64, ctypes.POINTER(ctypes.c_int32), c_int]),
    "iwvi_encoder_backward_act": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                          ctypes.POINTER(ctypes.c_int32), c_int, c_int, c_void_p,
                                          ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "iwvi_lv_encoder_backward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                         c_int, c_int64, c_int, c_int,
                                         c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                         ctypes.POINTER(ctypes.c_int32), c_int, c_int,
                                         ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "iwvi_encoder_backward": (c_int, [c_void_p, c_int64, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                      ctypes.POINTER(ctypes.c_int32), c_int, c_void_p,
                                      ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "iwvi_kde_loglik": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iwvi_natgrad_ws_bytes": (c_size_t, [c_int]),
    "iwvi_natgrad_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_void_p]),
    "iwvi_natgrad_ws_bytes_ex": (c_size_t, [c_int, c_int]),
    "iwvi_natgrad_step_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_double, c_void_p, c_size_t, c_void_p]),
    "iwvi_debug_last_natgrad_route": (c_int, []),
    "iwvi_adam_step": (c_int, [ctypes.POINTER(AdamTensor), c_int, c_double, c_double, c_double, c_double, c_int64,
                               c_int, c_int, c_void_p]),
    "iwvi_adam_step_dev": (c_int, [ctypes.POINTER(AdamTensor), c_int, c_double, c_double, c_double, c_double, c_void_p, c_int, c_void_p]),
    "iwvi_gp_layer_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                      c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "iwvi_gp_layer_forward_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                         c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "iwvi_gp_layer_fullcov_ex": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                         c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "iwvi_gp_fullcov_ws_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "iwvi_gp_layer_fullcov": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                      c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iwvi_mvn_sample_ws_bytes": (c_size_t, [c_int64, c_int, c_int]),
    "iwvi_mvn_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, c_void_p, c_void_p]),
    "iwvi_lv_layer_forward_act": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p),
                                          ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int, c_int,
                                          c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_int64, c_void_p]),
    "iwvi_lv_layer_forward": (c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_void_p),
                                      ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int,
                                      c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int64, c_void_p]),
    "iwvi_dgp_forward": (c_int, [ctypes.POINTER(LayerDesc), c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int,
                                 c_int64, c_int64, c_int64, c_float, ctypes.c_uint64, c_void_p, c_void_p,
                                 ctypes.POINTER(ElboDesc), c_void_p]),
    "iwvi_logw_reduce": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int64, ctypes.POINTER(c_void_p),
                                 ctypes.POINTER(ctypes.c_int32), c_int, c_double, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iwvi_iw_elbo_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int,
                                    c_int64, c_int64,
                                    ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int,
                                    ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int, c_double,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iwvi_iw_elbo_reduce_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int64, c_int, c_int,
                                        c_int64, c_int64,
                                        ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int,
                                        ctypes.POINTER(c_void_p), ctypes.POINTER(ctypes.c_int32), c_int, c_double,
                                        c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iwvi_lse_merge": (c_int, [c_void_p, c_int, c_int64, c_int, ctypes.POINTER(c_void_p),
                               ctypes.POINTER(ctypes.c_int32), c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "iwvi_lse_merge_steps": (c_int, [c_void_p, c_int, c_int, c_int64, c_int, ctypes.POINTER(c_void_p),
                                     ctypes.POINTER(ctypes.c_int32), c_int, c_double, c_void_p, c_void_p, c_void_p]),
    "iwvi_gauss_kl": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "iwvi_gaussian_var_exp": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int64, c_int, c_int64, c_int64,
                                      c_void_p, c_void_p]),
    "iwvi_unwhiten": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "iwvi_fill_normal": (c_int, [c_void_p, c_int64, ctypes.c_uint64, ctypes.c_uint64, c_void_p]),
    "iwvi_fill_normal_dev": (c_int, [c_void_p, c_int64, ctypes.c_uint64, c_void_p, c_void_p]),
}

DEBUG_OPTIONS = ("IWVI_BW_FUSED", "IWVI_CHAIN_EXIT", "IWVI_FW_SLOW_TAIL", "IWVI_FW_MAX_NS", "IWVI_NATGRAD_UNFUSED", "IWVI_NG_ONE_WG", "IWVI_NG_STOP", "IWVI_DEBUG_STOP", "IWVI_PRE_STAMP_P",
                 "IWVI_FW_NO_LEAN", "IWVI_PRE_SB_INLINE", "IWVI_BW_P5_F32")


def set_debug_option(name, value):
    """``iwvi_debug_set_option``: a development route switch of the library (process-wide; 0 = default route)."""
    check(lib().iwvi_debug_set_option(name.encode(), int(value)))


_lib = None


def lib():
    """Load libiwvi_hip.so once; raise loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IwviError(
                "HIP extension not built: %s is missing. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C dgps_with_iwvi_amd/csrc` (needs hipcc, gfx950)." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        if handle.iwvi_version() != ABI_VERSION:
            raise IwviError("libiwvi_hip.so ABI version %d != %d (rebuild: make -C dgps_with_iwvi_amd/csrc)" % (handle.iwvi_version(), ABI_VERSION))
        # development route switches: the C library never reads the environment; IWVI_* variables seen HERE become its options
        for name in DEBUG_OPTIONS:
            if os.environ.get(name):
                try:
                    handle.iwvi_debug_set_option(name.encode(), int(os.environ[name]))
                except ValueError:
                    handle.iwvi_debug_set_option(name.encode(), 1)
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        raise IwviError("libiwvi_hip: %s (code %d)" % (lib().iwvi_last_error().decode(), rc), rc)


def dev_tensor(t, name="tensor", dtype=torch.float32):
    """Validate a tensor that is about to cross the C-ABI: ROCm device, dtype, contiguous."""
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise IwviError(
            "%s is on %s: the IW-ELBO hot path only exists as gfx950 HIP kernels "
            "(there is no CPU fallback); move the model to a ROCm device" % (name, t.device))
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def ptr(t):
    return None if t is None else c_void_p(t.data_ptr())


def stream_ptr():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr_array(tensors):
    arr = (c_void_p * max(len(tensors), 1))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr
