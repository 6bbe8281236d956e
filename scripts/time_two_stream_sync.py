#!/usr/bin/env python3
"""What a cross-stream dependency costs inside a captured graph (development aid): the evaluation's two launches on ONE stream against
the factorisation launch on a second stream with the two event edges an overlapped design would keep (precompute_i after forward_{i-1},
forward_i after precompute_i).  Nothing overlaps here -- the difference is the price of the edges."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS, Step
from dgps_with_iwvi_amd import synthetic

dev = torch.device("cuda:0")
cfg = CONFIGS[2]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
model = synthetic.build_model(spec, dev)
step = Step(model, spec, dev, "k", 1, exchange=False)
step.run(); torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev, priority=-1)
real_precompute = model.precompute


def run_two(main):
    side.wait_stream(main)
    with torch.cuda.stream(side):
        real_precompute(with_encoders=True)
    main.wait_stream(side)
    model.precompute = lambda *a, **k: None
    try:
        step.run()
    finally:
        model.precompute = real_precompute


graphs = {}
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step.run(); run_two(s)
    for name, fn in (("one stream", step.run), ("two streams, two edges per evaluation", lambda: run_two(s))):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(20):
                fn()
        graphs[name] = g
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for _ in range(20):
    for g in graphs.values(): g.replay()
torch.cuda.synchronize()
for rnd in range(3):
    for name, g in graphs.items():
        ts = []
        for _ in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6 / 20)
        print("%-40s %.2f us per evaluation (median of 10 replays of 20)" % (name, np.median(ts)))
