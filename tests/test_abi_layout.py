"""The ctypes mirrors of dgps_with_iwvi_amd/_abi.py against include/iwvi_hip.h as a C compiler lays the structs out (gcc, no GPU):
same size, same offset for every field.  A silent mismatch here would hand the kernels shifted descriptors."""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STRUCTS = [("iwvi_gp_desc", "GpDesc"), ("iwvi_enc_desc", "EncDesc"), ("iwvi_layer_desc", "LayerDesc"), ("iwvi_elbo_desc", "ElboDesc"),
           ("iwvi_gp_bwd_desc", "GpBwdDesc"), ("iwvi_adam_tensor", "AdamTensor")]


def test_ctypes_structs_match_the_header_layout(tmp_path):
    from dgps_with_iwvi_amd import _abi
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "iwvi_hip.h"', "int main(void) {"]
    for cname, pyname in STRUCTS:
        cls = getattr(_abi, pyname)
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    r = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]                       # (a field the header does not have fails right here)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    got = {}
    for ln in out.splitlines():
        s, f, v = ln.split()
        got[(s, f)] = int(v)
    for cname, pyname in STRUCTS:
        cls = getattr(_abi, pyname)
        assert ctypes.sizeof(cls) == got[(cname, "sizeof")], (cname, ctypes.sizeof(cls), got[(cname, "sizeof")])
        for fname, _ in cls._fields_:
            assert getattr(cls, fname).offset == got[(cname, fname)], (cname, fname, getattr(cls, fname).offset, got[(cname, fname)])


def test_abi_version_constant_matches_the_header():
    from dgps_with_iwvi_amd import _abi
    with open(os.path.join(ROOT, "include", "iwvi_hip.h")) as f:
        ver = [ln for ln in f if ln.startswith("#define IWVI_ABI_VERSION")][0].split()[2]
    assert int(ver) == _abi.ABI_VERSION


def test_flag_constants_match_the_header():
    """Every IWVI_* flag / enum value _abi.py mirrors by hand equals the header's #define (a drifted constant is a silent wrong route)."""
    import re
    from dgps_with_iwvi_amd import _abi
    defs = {}
    with open(os.path.join(ROOT, "include", "iwvi_hip.h")) as f:
        for ln in f:
            m = re.match(r"#define\s+(IWVI_[A-Z0-9_]+)\s+(-?\d+)\b", ln)
            if m:
                defs[m.group(1)] = int(m.group(2))
    pairs = {"GP_WANT_DENSE": "IWVI_GP_WANT_DENSE", "GP_WANT_LM": "IWVI_GP_WANT_LM", "GP_F64_STAGE1": "IWVI_GP_F64_STAGE1",
             "GP_REUSE_FACTOR": "IWVI_GP_REUSE_FACTOR", "GP_FACTOR_ONLY": "IWVI_GP_FACTOR_ONLY",
             "BW_F32_CHAIN": "IWVI_BW_F32_CHAIN", "BW_OWN_QSCALE": "IWVI_BW_OWN_QSCALE",
             "MAX_STACK": "IWVI_MAX_STACK", "ERR_UNSUPPORTED": "IWVI_ERR_UNSUPPORTED"}
    checked = 0
    for py, c in pairs.items():
        if hasattr(_abi, py) and c in defs:
            assert getattr(_abi, py) == defs[c], (py, getattr(_abi, py), c, defs[c])
            checked += 1
    assert checked >= 6, (checked, sorted(defs)[:10])
