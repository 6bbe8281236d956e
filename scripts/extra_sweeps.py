"""One-off robustness hunt: the two seeded random sweeps of tests/ with other seeds and more cases (development aid).
usage: python scripts/extra_sweeps.py [seed] [n]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_random_sweep as fw
import test_gpu_backward_sweep as bw
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
bad = 0
for name, mod, fn in (("forward", fw, fw.test_random_shape_matches_oracle), ("backward", bw, bw.test_random_shape_gradients_match_oracle)):
    for case in mod._cases(n, seed):
        try:
            fn(dev, case)
        except Exception as e:
            bad += 1
            print("FAIL", name, case, type(e).__name__, str(e)[:300])
print("done: %d failures" % bad)
