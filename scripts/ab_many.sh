#!/bin/bash
# development aid: the same bench line from several builds of the library on ONE box, interleaved.  usage: scripts/ab_many.sh <rounds> a.so b.so ...
D=dgps_with_iwvi_amd/csrc
R=$1; shift
cp $D/libiwvi_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp $D/$v $D/libiwvi_hip.so
    python bench.py --no-cpu-baseline --no-train-leg --median-iters 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s ms_per_step %.5f  fwd launch_ms %.5f' % ('$v', r['ms_per_step'], r['roofline']['launch_ms']))"
  done
done
cp /tmp/lib_keep.so $D/libiwvi_hip.so
