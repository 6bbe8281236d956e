"""the register / scratch budget of the hot kernels is checked from the code objects' metadata on a GPU-less host
(a stage-2 edit once put the configs[3] variant of k_dgp_forward 216 B into scratch for four commits: +12 % unnoticed)."""
import os

import pytest


def _rows():
    from dgps_with_iwvi_amd import kernel_resources as kr
    if not os.path.exists(kr.LIB_PATH):
        pytest.skip("libiwvi_hip.so not built (run __graft_entry__.build())")
    if not os.path.exists(os.path.join(kr.LLVM_BIN, "llvm-readelf")):
        pytest.skip("no llvm-readelf here")
    return kr, kr.kernel_table()


def test_hot_kernels_stay_within_their_register_budget():
    kr, rows = _rows()
    kr.check(rows)
    names = {r["demangled"] for r in rows}
    # the variants the BASELINE configs take: configs[1]/[2] (bound-only and with outputs), configs[3] (NS = 5, large M), configs[4] (NS = 3)
    for must in ("k_dgp_forward<5,true,false,1,false>", "k_dgp_forward<5,true,false,2,false>", "k_dgp_forward<5,true,true,0,false>", "k_dgp_forward<3,true,true,0,false>",
                 "k_precompute", "k_bw_chain<5,8>"):
        assert must in names, must
    by = {r["demangled"]: r for r in rows}
    assert by["k_dgp_forward<5,true,false,1,false>"]["private_segment_fixed_size"] == 0      # the headline variant: no scratch, ever
    assert by["k_dgp_forward<3,true,true,0,false>"]["private_segment_fixed_size"] <= 52      # configs[4]: the solve's operand rings + the Gram's prefetch (round 5)
    assert all(r["vgpr_count"] <= 256 for r in rows if r["demangled"].startswith("k_dgp_forward"))


def test_the_guard_fires_on_a_variant_that_spills():
    kr, rows = _rows()
    worse = [dict(r) for r in rows]
    for r in worse:
        if r["demangled"] == "k_dgp_forward<5,true,false,1,false>":
            r["private_segment_fixed_size"] = 216
            r["vgpr_spill_count"] = 54
    with pytest.raises(AssertionError, match="216 B of scratch"):
        kr.check(worse)
    assert kr._readable("_ZN4iwvi13k_dgp_forwardILi5ELb1ELb0ELi1ELb0EEEvNS_6FwArgsE") == "k_dgp_forward<5,true,false,1,false>"
