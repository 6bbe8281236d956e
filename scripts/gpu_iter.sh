#!/bin/bash
# development loop on the GPU box: quick parity subset, phase stamps, one bench line.   usage: scripts/gpu_iter.sh <tag>
TAG=${1:-iter}
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_gpu_parity.py tests/test_gpu_random_sweep.py -m gpu -q -x --timeout 600 2>&1 | tail -4 > gpurun_out/$TAG/tests.txt
python scripts/stamp_phases.py > gpurun_out/$TAG/stamp_phases.txt 2>&1
python bench.py --no-cpu-baseline --no-train-leg > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/tests.txt; grep -v "amdgpu.ids" gpurun_out/$TAG/stamp_phases.txt
python - <<PY
import json
r = json.load(open("gpurun_out/$TAG/bench.json"))
print("ms_per_step %.5f  median %.5f  fwd launch_ms %.5f  frac %.4f  model_frac %.4f" % (r["ms_per_step"], r.get("ms_per_step_median", 0), r["roofline"]["launch_ms"], r["roofline"]["frac"], r["model_frac_of_mfma_peak"]))
PY
