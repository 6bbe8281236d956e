#!/bin/bash
# development aid: value + gradient / training-step time of several builds on ONE box, interleaved.   usage: scripts/ab_bw.sh <rounds> <a.so> <b.so> ...
D=dgps_with_iwvi_amd/csrc
R=$1; shift
cp $D/libiwvi_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp $D/$v $D/libiwvi_hip.so
    python scripts/time_backward.py --config 2 2>/dev/null | grep -E "graph replay|config 2" | sed "s/^/$v  /" | cut -c1-220
  done
done
cp /tmp/lib_keep.so $D/libiwvi_hip.so
