#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into one small JSON + CSV for profiles/.

Per kernel: calls, average duration (kernel trace), and per-dispatch averages of every collected counter.
Derived for the forward kernel: MFMA busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES per SE...)) is
reported raw -- the gfx950 derived-metric XML is missing in ROCm 7.2, so the raw counters are kept and the
formula used is stated next to each derived number.  HBM traffic: FETCH_SIZE is doubled (gfx950 reports half
of a wide coalesced read, MI355X_MICROARCH.md section HBM), WRITE_SIZE is taken as is; both are in KiB."""
import csv, glob, json, os, sys
from collections import defaultdict

tag = sys.argv[1]
root = os.path.join("gpurun_out", "prof_" + tag)
out = {"tag": tag, "kernels": {}}

def short(name):
    return name.split("(")[0].replace("void ", "").strip()

# kernel trace: durations
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        acc[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in acc.items():
        if k.startswith("iwvi::"):
            out["kernels"].setdefault(k, {})["calls"] = len(v)
            out["kernels"][k]["avg_ns"] = sum(v) / len(v)
            v2 = sorted(v); out["kernels"][k]["median_ns"] = v2[len(v2) // 2]
# counters
for p in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob(os.path.join(root, p, "**", "*counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = short(row["Kernel_Name"])
            if k.startswith("iwvi::"):
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            d = out["kernels"].setdefault(k, {}).setdefault("counters_per_dispatch", {})
            for c, v in cs.items():
                d[c] = sum(v) / len(v)
for k, d in out["kernels"].items():
    c = d.get("counters_per_dispatch", {})
    der = {}
    if "FETCH_SIZE" in c: der["hbm_read_bytes(FETCH_SIZE*1024*2)"] = c["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in c: der["hbm_write_bytes(WRITE_SIZE*1024)"] = c["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
        # busy cycles are summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        der["mfma_busy_frac(SQ_VALU_MFMA_BUSY_CYCLES/(1024*GRBM_GUI_ACTIVE/8))"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in c and "avg_ns" in d:
        der["mfma_f32_flops_per_s(MOPS*512/avg_ns)"] = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512 / (d["avg_ns"] * 1e-9)
    if "SQ_INSTS_VALU_MFMA_MOPS_F16" in c and "avg_ns" in d:
        der["mfma_f16_flops_per_s(MOPS*512/avg_ns)"] = c["SQ_INSTS_VALU_MFMA_MOPS_F16"] * 512 / (d["avg_ns"] * 1e-9)
    if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c and "avg_ns" in d:
        der["mfma_f64_flops_per_s(MOPS*512/avg_ns)"] = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / (d["avg_ns"] * 1e-9)
    if "mfma_f32_flops_per_s(MOPS*512/avg_ns)" in der and "mfma_f16_flops_per_s(MOPS*512/avg_ns)" in der:
        # what the matrix pipe was asked to do, each instruction class against its own dense peak (MI355X_MICROARCH.md): must come out
        # near the busy fraction the SQ counter reports
        der["mfma_pipe_frac(f32/157.3T+f16/2.5P+f64/78.6T)"] = (der["mfma_f32_flops_per_s(MOPS*512/avg_ns)"] / 157.3e12 +
                                                                  der["mfma_f16_flops_per_s(MOPS*512/avg_ns)"] / 2.5e15 +
                                                                  der.get("mfma_f64_flops_per_s(MOPS*512/avg_ns)", 0.0) / 78.6e12)
    d["derived"] = der
os.makedirs("profiles", exist_ok=True)
# what bench.py quotes next to its live timing (counters need their own passes): the dominant kernel's entry of THIS profile
if len(sys.argv) > 2 and sys.argv[2] == "--latest":
    ks = [k for k in out["kernels"] if "k_dgp_forward" in k and "counters_per_dispatch" in out["kernels"][k]]
    if ks:
        k = max(ks, key=lambda n: out["kernels"][n].get("calls", 0))
        d, der = out["kernels"][k], out["kernels"][k]["derived"]
        import subprocess
        try:
            commit = subprocess.check_output(["git", "log", "-1", "--format=%h", "--", "dgps_with_iwvi_amd/csrc"], stderr=subprocess.DEVNULL).decode().strip()   # the kernels' last commit (tests/test_profiles_fresh.py)
        except Exception:
            commit = None
        lat = {"kernel": k.replace("iwvi::", ""), "config": "BASELINE.json configs[2]",
               "hbm_read_bytes": der.get("hbm_read_bytes(FETCH_SIZE*1024*2)"), "hbm_write_bytes": der.get("hbm_write_bytes(WRITE_SIZE*1024)"),
               "source": "profiles/%s_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM)" % tag,
               "mfma_busy_frac": der.get("mfma_busy_frac(SQ_VALU_MFMA_BUSY_CYCLES/(1024*GRBM_GUI_ACTIVE/8))"),
               "mfma_issued_f32_tflops": der.get("mfma_f32_flops_per_s(MOPS*512/avg_ns)", 0.0) / 1e12,
               "mfma_issued_f16_tflops": der.get("mfma_f16_flops_per_s(MOPS*512/avg_ns)", 0.0) / 1e12,
               "mfma_pipe_frac": der.get("mfma_pipe_frac(f32/157.3T+f16/2.5P+f64/78.6T)"),
               "kernel_avg_us_rocprof": d.get("avg_ns", 0.0) / 1e3, "pmc_profile_of_commit": commit}
        try:
            sys.path.insert(0, os.getcwd())
            from dgps_with_iwvi_amd.kernel_resources import csrc_hash
            lat["csrc_sha256"] = csrc_hash()                      # the sources the profiled library was built from (git-free freshness check)
        except Exception:
            lat["csrc_sha256"] = None
        lat["hbm_bytes"] = (lat["hbm_read_bytes"] or 0.0) + (lat["hbm_write_bytes"] or 0.0)
        json.dump(lat, open(os.path.join("profiles", "traffic_latest.json"), "w"), indent=1)
# the value + gradient evaluation (scripts/profile_backward.sh): its dominant kernels' entries, quoted by bench.py's training_step.roofline
if len(sys.argv) > 2 and sys.argv[2] == "--backward":
    import subprocess
    try:
        commit = subprocess.check_output(["git", "log", "-1", "--format=%h", "--", "dgps_with_iwvi_amd/csrc"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        commit = None
    tot = sum(d.get("avg_ns", 0.0) * d.get("calls", 0) for d in out["kernels"].values())
    rows = []
    for k, d in sorted(out["kernels"].items(), key=lambda kv: -kv[1].get("avg_ns", 0.0) * kv[1].get("calls", 0)):
        der = d.get("derived", {})
        rows.append({"kernel": k.replace("iwvi::", ""), "calls": d.get("calls"), "avg_us": d.get("avg_ns", 0.0) / 1e3,
                     "share_of_kernel_time": d.get("avg_ns", 0.0) * d.get("calls", 0) / max(tot, 1.0),
                     "mfma_busy_frac": der.get("mfma_busy_frac(SQ_VALU_MFMA_BUSY_CYCLES/(1024*GRBM_GUI_ACTIVE/8))"),
                     "mfma_pipe_frac": der.get("mfma_pipe_frac(f32/157.3T+f16/2.5P+f64/78.6T)"),
                     "hbm_read_bytes": der.get("hbm_read_bytes(FETCH_SIZE*1024*2)"), "hbm_write_bytes": der.get("hbm_write_bytes(WRITE_SIZE*1024)")})
    json.dump({"config": "BASELINE.json configs[2], one value + gradient evaluation (scripts/time_backward.py --only-gradient)",
               "source": "profiles/%s_summary.json" % tag, "pmc_profile_of_commit": commit, "kernels": rows[:12]},
              open(os.path.join("profiles", "backward_latest.json"), "w"), indent=1)
with open(os.path.join("profiles", tag + "_summary.json"), "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = [r for r in csv.reader(open(f))]
    with open(os.path.join("profiles", tag + "_kernel_stats.csv"), "w") as fh:
        w = csv.writer(fh)
        for r in rows:
            if r and (r[0] == "Name" or "iwvi::" in r[0]): w.writerow(r)
print(json.dumps(out, indent=1, sort_keys=True))
