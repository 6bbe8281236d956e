#!/bin/bash
# development aid: in-kernel total of the factorisation launch for several builds (and debug switches) on ONE box, interleaved
D=dgps_with_iwvi_amd/csrc
R=$1; shift
cp $D/libiwvi_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    lib=${v%%:*}; dbg=${v##*:}; [ "$dbg" = "$v" ] && dbg=0
    cp $D/$lib $D/libiwvi_hip.so
    T=$(IWVI_DEBUG_STOP=$dbg python scripts/stamp_precompute.py --p 1 2>/dev/null | grep "^total\|Gram block" | awk '{printf "%s ", $(NF-1)}')
    echo "$v  total / up-front: $T"
  done
done
cp /tmp/lib_keep.so $D/libiwvi_hip.so
