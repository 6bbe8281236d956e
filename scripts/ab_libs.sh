#!/bin/bash
# development aid: the same bench line from two builds of the library on ONE box, interleaved (box-to-box spread is larger than most kernel changes).
#   usage: scripts/ab_libs.sh <a.so> <b.so> [rounds]      (paths relative to dgps_with_iwvi_amd/csrc)
D=dgps_with_iwvi_amd/csrc
cp $D/libiwvi_hip.so /tmp/lib_keep.so
for r in $(seq 1 ${3:-3}); do
  for v in $1 $2; do
    cp $D/$v $D/libiwvi_hip.so
    python bench.py --no-cpu-baseline --no-train-leg 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v  ms_per_step %.5f  median %.5f  fwd launch_ms %.5f' % (r['ms_per_step'], r.get('ms_per_step_median',0), r['roofline']['launch_ms']))"
  done
done
cp /tmp/lib_keep.so $D/libiwvi_hip.so
