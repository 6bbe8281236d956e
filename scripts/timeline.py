"""Print the kernel timeline of the last graph replay in a rocprofv3 kernel trace (development aid).
usage: python scripts/timeline.py <kernel_trace.csv> [end-marker kernel substring, default k_adam]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
mark = sys.argv[2] if len(sys.argv) > 2 else "k_adam"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("iwvi::", "")[:40]
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-40s q%-3s start %8.1f dur %7.1f end %8.1f" % (n, r.get("Queue_Id", "?"), (st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3))
