"""Graph-mode training steps only (training.Trainer(use_graph=True).step), for kernel profiles: scripts/prof_any.sh <tag> scripts/time_train_step.py
Usage: python scripts/time_train_step.py [--config 2] [--iters 50] [--op both|ng|adam]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import synthetic   # noqa: E402
from dgps_with_iwvi_amd.training import Trainer   # noqa: E402
from scripts.time_backward import CONFIGS   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--op", default="both")
a = ap.parse_args()
dev = torch.device("cuda:0")
model = synthetic.build_model(synthetic.make_spec(**CONFIGS[a.config], seed=0), dev)
tr = Trainer(model, use_graph=True)
for _ in range(3):
    tr.step()
fn = {"both": tr.step, "ng": lambda: tr._replay("ng", None), "adam": lambda: tr._replay("adam", None)}[a.op]
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    fn()
torch.cuda.synchronize()
print("config %d: %s: %.3f ms per call (%d calls after 3 warm-up steps)" % (a.config, a.op, (time.perf_counter() - t0) / a.iters * 1e3, a.iters))
