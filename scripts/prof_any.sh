#!/bin/bash
# kernel-trace summary of any python script on the GPU box.  usage: scripts/prof_any.sh <tag> <script> [args...]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1
python3 - <<PY
import csv, glob
from collections import defaultdict
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
acc = defaultdict(list)
for row in csv.DictReader(open(f)):
    acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
tot = sum(sum(v) for v in acc.values())
lines = ["%-60s %6d %10.1f %8.2f %5.1f%%" % (k[:60], len(v), sum(v) / 1e3, sum(v) / len(v) / 1e3, 100.0 * sum(v) / tot) for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))]
open("$OUT/kernel_stats.txt", "w").write("kernel calls total_us avg_us pct\n" + "\n".join(lines) + "\n")
print("\n".join(lines[:40]))
PY
