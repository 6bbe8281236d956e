#!/bin/bash
# Profile of the value + gradient evaluation (row F1) on the GPU box: kernel trace + three PMC passes, each in its own run with
# --kernel-trace only (as the pool requires).  Writes gpurun_out/prof_$1/ ; summarise with scripts/summarise_profile.py <tag> --backward
#   usage: scripts/profile_backward.sh <tag> [config]
set -u
TAG=${1:-rXXbw}; CFG=${2:-2}
ARGS="scripts/time_backward.py --config $CFG --iters 20 --only-gradient"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- python3 $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc2 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -- python3 $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 -- python3 $ARGS > $OUT/pmc3.log 2>&1
tail -2 $OUT/trace.log
