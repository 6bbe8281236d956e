import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "").replace("iwvi::", "")
# last occurrence of k_ng_vec = end of a NatGrad op; print the 40 kernels before the last k_adam
idx = [i for i, r in enumerate(rows) if "k_adam" in r["Kernel_Name"]]
end = idx[-1]; start = idx[-2] + 1
t0 = int(rows[start]["Start_Timestamp"])
for r in rows[start:end + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-44s q%-3s start %8.1f dur %7.1f end %8.1f" % (name(r)[:44], r.get("Queue_Id", "?"), (st - t0) / 1e3, (en - st) / 1e3, (en - t0) / 1e3))
