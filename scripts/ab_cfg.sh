#!/bin/bash
# development aid: bench lines of several builds at one config on ONE box.   usage: scripts/ab_cfg.sh <config> <rounds> <a.so> <b.so> ...
D=dgps_with_iwvi_amd/csrc
C=$1; R=$2; shift; shift
cp $D/libiwvi_hip.so /tmp/lib_keep.so
for r in $(seq 1 $R); do
  for v in "$@"; do
    cp $D/$v $D/libiwvi_hip.so
    python bench.py --config $C --steps 20 --warmup 5 --no-cpu-baseline --no-train-leg --median-iters 0 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
g=r['roofline'].get('gemm_phase_mfma_util') or {}
print('$v  config $C  ms_per_step %.5f  fwd launch_ms %.5f  gemm util %s  elbo %.6e' % (r['ms_per_step'], r['roofline']['launch_ms'], g.get('value'), r['elbo']))"
  done
done
cp /tmp/lib_keep.so $D/libiwvi_hip.so
