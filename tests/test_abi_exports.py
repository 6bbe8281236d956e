"""The C-ABI library loads on a GPU-less host and exports every function include/iwvi_hip.h declares, with the
ctypes prototypes of dgps_with_iwvi_amd/_abi.py covering all of them (no compute call is made here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "iwvi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(iwvi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("iwvi_gp_precompute", "iwvi_dgp_forward", "iwvi_gp_layer_forward", "iwvi_lv_layer_forward",
                 "iwvi_iw_elbo_reduce", "iwvi_logw_reduce", "iwvi_lse_merge", "iwvi_gauss_kl", "iwvi_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from dgps_with_iwvi_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        pytest.skip("libiwvi_hip.so not built (run __graft_entry__.build())")
    lib = _abi.lib()
    names = _declared()
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    unbound = [n for n in names if n not in _abi.PROTOTYPES]
    assert not unbound, "no ctypes prototype for %s" % unbound
    assert lib.iwvi_version() == _abi.ABI_VERSION


def test_product_path_has_no_cpu_fallback():
    """A CPU tensor must be refused loudly (the oracle is test infrastructure, never a fallback)."""
    import torch
    from dgps_with_iwvi_amd import _abi
    with pytest.raises(_abi.IwviError):
        _abi.dev_tensor(torch.zeros(3), "x")
    src = "".join(open(os.path.join(ROOT, "dgps_with_iwvi_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "dgps_with_iwvi_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_training_entry_points_refuse_bad_arguments_without_touching_the_gpu():
    """Argument validation happens before any HIP call: error code + text, on a GPU-less host too."""
    import ctypes
    from dgps_with_iwvi_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        pytest.skip("libiwvi_hip.so not built (run __graft_entry__.build())")
    lib = _abi.lib()
    assert lib.iwvi_gp_layer_backward_ws_bytes(0, 128, 8, 5) == 0
    assert lib.iwvi_gp_layer_backward_ws_bytes(20480, 128, 9, 5) > 20480 * 128 * 4 * 2
    assert lib.iwvi_natgrad_ws_bytes(0) == 0 and lib.iwvi_natgrad_ws_bytes(128) > 8 * 128 * 128 * 8
    d = _abi.GpBwdDesc()
    assert lib.iwvi_gp_layer_backward(ctypes.byref(d), 16, None, None) == -1          # IWVI_ERR_ARG: no workspace
    assert b"iwvi_gp_layer_backward" in lib.iwvi_last_error()
    assert lib.iwvi_iw_elbo_backward(None, None, None, 1, None, None, 0, 4, 2, 0.1, 1.0, 0, None, None, None, None, None, 0, None, 0, None, None, None) == -1
    assert lib.iwvi_lv_layer_backward(None, None, 1, 0, None, None, 0, 0, None, 1, 4, 2, 1, None, None) == -1
    assert lib.iwvi_natgrad_step(None, None, None, None, 8, 1, 0.1, None, None) == -1
    arr = (_abi.AdamTensor * 1)()
    assert lib.iwvi_adam_step(arr, 0, 1e-3, 0.9, 0.999, 1e-8, 1, 1, 0, None) == -1
    assert lib.iwvi_adam_step(arr, 1, 1e-3, 0.9, 0.999, 1e-8, 1, 1, 0, None) == -1     # null tensor pointers
    assert b"iwvi_adam_step" in lib.iwvi_last_error()
    dims = (ctypes.c_int32 * 3)(9, 20, 2)
    assert lib.iwvi_encoder_backward_ws_bytes(0, dims, 2) == 0 and lib.iwvi_encoder_backward_ws_bytes(64, dims, 2) > 0
    assert lib.iwvi_encoder_backward(None, 64, None, None, dims, 2, None, None, None, None, None) == -1


def test_backward_sizing_entries_answer_for_both_arithmetic_modes():
    """``iwvi_gp_layer_backward_needs_u`` / ``_ws_bytes`` take no descriptor, the arithmetic mode is chosen per call (desc.flags &
    IWVI_BW_F32_CHAIN): they must cover whichever mode the later call asks for.  M = 512 at a small batch: the split-f16 chain runs from
    a_out alone, the fp32 chain does not fit and the GEMM path reads u -- the forward must be told to keep it.  Host logic only."""
    from dgps_with_iwvi_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        pytest.skip("libiwvi_hip.so not built (run __graft_entry__.build())")
    lib = _abi.lib()
    assert lib.iwvi_gp_layer_backward_needs_u(1024, 512, 8, 5, 5) == 1
    assert lib.iwvi_gp_layer_backward_needs_u(20480, 128, 9, 5, 8) == 0            # configs[2]: the chain in either mode
    assert lib.iwvi_gp_layer_backward_needs_u(20480, 120, 9, 5, 8) == 1            # M not a multiple of 16: GEMM path
    assert lib.iwvi_gp_layer_backward_ws_bytes(1024, 512, 8, 5) >= lib.iwvi_gp_layer_backward_ws_bytes(1024, 256, 8, 5) > 0
