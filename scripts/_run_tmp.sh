timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_backward.py -q -m gpu 2>&1 | grep -E "passed|failed|Error|max err|FAILED" | head -30
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bwprof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/bwprof -o bw -- python3 scripts/time_backward.py --config 2 --iters 20 --only-gradient > gpurun_out/bwprof/log.txt 2>&1
for c in 1 2 3; do timeout 300 python scripts/time_backward.py --config $c 2>&1 | tail -1; done
