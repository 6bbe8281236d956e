"""Register / scratch budget of every kernel in libiwvi_hip.so, read from the code objects' metadata (no GPU needed).

The layer kernel's hot variants sit at the 256-VGPR limit: a source change anywhere in the layer loop can push one of them into
scratch, and a variant that only the large configs take then regresses unnoticed (round 4: the five-sub-tile large-M variant went
216 B into scratch, configs[3] 1.27 -> 1.43 ms, for four commits).  ``check()`` is run by ``__graft_entry__.build()`` and by
tests/test_kernel_resources.py: it fails when a ``k_dgp_forward`` or ``k_bw_chain`` instantiation uses scratch memory
(``.private_segment_fixed_size`` > 0) beyond what ``ALLOWED_SCRATCH`` lists, or when a listed spill count grows.

  python -m dgps_with_iwvi_amd.kernel_resources [--write profiles/<tag>_kernel_resources.txt]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libiwvi_hip.so")
LLVM_BIN = os.environ.get("IWVI_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")

# kernels whose instantiations must not touch scratch memory at all ...
NO_SCRATCH = ("k_dgp_forward", "k_bw_chain")
# ... except these (demangled-name substring -> bytes of scratch it is known to use; lower it when a kernel improves)
ALLOWED_SCRATCH = {
    # round 6 (profiles/r06_kernel_resources.txt): every BIG variant (some layer with M > 128: configs[3] / [4]) and every float64-route
    # variant is at ZERO scratch and 178-228 VGPRs since the thread index is made opaque again at every phase boundary (FW_REBASE in
    # csrc/dgp_forward.hip: the lane-dependent tile addresses are formed in the phase that uses them instead of living -- ~80 registers --
    # across the whole layer loop).  Left: a few spilled VGPRs in two narrow M <= 128 variants no BASELINE config takes and in the value +
    # gradient variant of the headline stack (the same rebase there costs the headline 1 %: measured, LABNOTES.md); none may grow.
    # (template arguments: NS, S16, BIG, LEAN_MODE, F64)
    "k_dgp_forward<2,true,false,0,false>": 16, "k_dgp_forward<4,true,false,0,false>": 56,
    "k_dgp_forward<5,true,false,2,false>": 12,
}
# spill counts that may not GROW (demangled-name substring -> (max vgpr spills, max sgpr spills)).  Round 6: the factorisation's column loop
# exists once per role (csrc/precompute_dev.h: chol_blocks), the kernel compiles to 118 VGPRs without a spilled VGPR or a byte of scratch
# (128 VGPRs, 35 spilled, 104 B until then); the SGPR spills (v_writelane into a VGPR, no memory) are what is left of its 1024-thread budget
MAX_SPILLS = {
    "k_precompute": (0, 760),
}


def csrc_hash():
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/iwvi_hip.h; names and contents, sorted): what a committed PMC
    profile is OF -- comparable on the GPU box, where there is no git history (profiles/traffic_latest.json: `csrc_sha256`)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".h")))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "iwvi_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def _readable(name):
    """`_ZN4iwvi13k_dgp_forwardILi5ELb1ELb0ELi1EEEvNS_6FwArgsE` -> `k_dgp_forward<5,true,false,1>` (no c++filt in the image: the kernel
    name and its integral / bool template arguments are all the guard needs)."""
    m = re.match(r"_ZN4iwvi(\d+)", name)
    if not m:
        m2 = re.match(r"_Z(\d+)", name)
        if not m2:
            return name
        n = int(m2.group(1))
        return name[m2.end():m2.end() + n]
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    if not rest.startswith("I"):
        return base
    args = []
    for kind, neg, val in re.findall(r"L([ib])(n?)(\d+)E", rest[:rest.index("EE") + 1] if "EE" in rest else rest):
        args.append(("true" if val == "1" else "false") if kind == "b" else ("-" if neg else "") + val)
    return "%s<%s>" % (base, ",".join(args))


def kernel_table(lib_path=LIB_PATH):
    """[{name, demangled, vgpr_count, ...}] for every kernel of every gfx950 code object bundled in ``lib_path``."""
    objdump, readelf = os.path.join(LLVM_BIN, "llvm-objdump"), os.path.join(LLVM_BIN, "llvm-readelf")
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        raise RuntimeError("llvm-objdump / llvm-readelf not found under %s (set IWVI_LLVM_BIN)" % LLVM_BIN)
    tmp = tempfile.mkdtemp(prefix="iwvi_cobj_")
    try:
        so = os.path.join(tmp, "lib.so")                          # (the bundles are extracted next to the input file)
        shutil.copy(lib_path, so)
        subprocess.run([objdump, "--offloading", so], capture_output=True, check=True, cwd=tmp)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([readelf, "--notes", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            if "---" not in notes:
                continue
            doc = notes[notes.index("---") + 3:]
            doc = doc[:doc.rindex("...")] if "..." in doc else doc
            meta = yaml.safe_load(doc) or {}
            for k in meta.get("amdhsa.kernels", []):
                row = {"symbol": k[".symbol"]}
                for fld in FIELDS:
                    row[fld] = int(k.get("." + fld, 0))
                rows.append(row)
        rows = [r for r in rows if "symbol" in r]
        for r in rows:
            r["name"] = r["symbol"][:-3] if r["symbol"].endswith(".kd") else r["symbol"]
        for r in rows:
            r["demangled"] = _readable(r["name"])
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def check(rows=None):
    """Raise AssertionError naming every kernel over its budget; returns the table."""
    rows = kernel_table() if rows is None else rows
    bad = []
    seen = {k: 0 for k in NO_SCRATCH}
    for r in rows:
        d = r["demangled"]
        for k in NO_SCRATCH:
            if d == k or d.startswith(k + "<"):
                seen[k] += 1
                allowed = ALLOWED_SCRATCH.get(d, 0)
                if r.get("private_segment_fixed_size", 0) > allowed:
                    bad.append("%s: %d B of scratch per lane (allowed %d; %d VGPR / %d SGPR spills)" % (
                        d, r["private_segment_fixed_size"], allowed, r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0)))
        for s, (mv, ms) in MAX_SPILLS.items():
            if s == d and (r.get("vgpr_spill_count", 0) > mv or r.get("sgpr_spill_count", 0) > ms):
                bad.append("%s: %d VGPR / %d SGPR spills (budget %d / %d)" % (d, r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0), mv, ms))
    for k, n in seen.items():
        if n == 0:
            bad.append("no instantiation of %s found in the library (metadata not read?)" % k)
    assert not bad, "kernel register budget:\n  " + "\n  ".join(bad)
    return rows


def format_table(rows):
    def short(d):
        return d
    out = ["%-64s %5s %5s %5s %6s %6s %8s %8s" % ("kernel", "vgpr", "agpr", "sgpr", "v.spl", "s.spl", "scratchB", "ldsB")]
    for r in sorted(rows, key=lambda r: r["demangled"]):
        out.append("%-64s %5d %5d %5d %6d %6d %8d %8d" % (short(r["demangled"])[:64], r.get("vgpr_count", 0), r.get("agpr_count", 0),
                                                          r.get("sgpr_count", 0), r.get("vgpr_spill_count", 0), r.get("sgpr_spill_count", 0),
                                                          r.get("private_segment_fixed_size", 0), r.get("group_segment_fixed_size", 0)))
    return "\n".join(out) + "\n"


if __name__ == "__main__":
    rows = kernel_table()
    text = format_table(rows)
    if len(sys.argv) > 2 and sys.argv[1] == "--write":
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)
    check(rows)
