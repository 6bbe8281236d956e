#!/bin/bash
# development aid: value + gradient (hipGraph replay) from several builds of the library on ONE box, interleaved.
#   usage: scripts/ab_bw_libs.sh <rounds> <a.so> <b.so> ...      (paths relative to dgps_with_iwvi_amd/csrc)
D=dgps_with_iwvi_amd/csrc
R=$1; shift
cp $D/libiwvi_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp $D/$v $D/libiwvi_hip.so
    echo "$v  $(python scripts/vg_graph_once.py 2>/dev/null | grep median)"
  done
done
cp /tmp/lib_keep.so $D/libiwvi_hip.so
