import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_training as t
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
bad = 0
for i in range(40):
    M = int(rng.integers(1, 129)); R = int(rng.integers(1, 6))
    try:
        t.test_natgrad_step_matches_oracle(dev, M, R)
    except Exception as e:
        bad += 1; print("FAIL", M, R, str(e)[:300])
print("natgrad sweep done:", bad, "failures")
