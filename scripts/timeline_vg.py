"""Kernel timeline of the LAST value + gradient evaluation in a rocprofv3 kernel trace of scripts/vg_graph_once.py (hipGraph replays):
from the evaluation's first k_precompute to its last kernel.  usage: python scripts/timeline_vg.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("iwvi::", "")
fw = [i for i, r in enumerate(rows) if "k_dgp_forward" in r["Kernel_Name"]]
last_fw, prev_fw = fw[-1], fw[-2]
# the evaluation's first kernel: the first k_precompute after the previous evaluation's forward
a = min(i for i in range(prev_fw + 1, last_fw) if "k_precompute" in rows[i]["Kernel_Name"] and int(rows[i]["Start_Timestamp"]) > int(rows[prev_fw]["End_Timestamp"]) + 100000) if any(
    "k_precompute" in rows[i]["Kernel_Name"] and int(rows[i]["Start_Timestamp"]) > int(rows[prev_fw]["End_Timestamp"]) + 100000 for i in range(prev_fw + 1, last_fw)) else prev_fw + 1
t0 = int(rows[a]["Start_Timestamp"])
end = 0
for r in rows[a:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    end = max(end, en)
    print("%-44s q%-3s start %8.1f dur %7.1f end %8.1f" % (name(r)[:44], r.get("Queue_Id", "?"), (st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3))
print("span %.1f us" % ((end - t0) / 1e3))
