#!/bin/bash
# Round profile on the GPU box: kernel trace + three PMC passes of the same bench command (each collected in its own
# run, with --kernel-trace only, as the pool requires).  Writes under gpurun_out/prof_$1/ ; summarise with
# scripts/summarise_profile.py and copy the result into profiles/.
#   usage: scripts/profile_round.sh <tag> [bench args...]
set -u
TAG=${1:-rXX}; shift || true
ARGS=${@:---steps 40 --warmup 5 --no-cpu-baseline --no-train-leg --no-graph}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc1 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -- python3 bench.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc2 --pmc FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -- python3 bench.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/pmc3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F64 -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
find $OUT -name "*.csv" | head -20
