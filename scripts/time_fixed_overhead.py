import os, sys, time
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bench import CONFIGS, Step
from dgps_with_iwvi_amd import synthetic, _abi
dev = torch.device("cuda:0")
cfg = CONFIGS[2]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
model = synthetic.build_model(spec, dev)
step = Step(model, spec, dev, "k", 1, exchange=False)
step.run(); torch.cuda.synchronize()
graphs = {}
s = torch.cuda.Stream(device=dev)
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step.run()
    for n in (1, 5, 20, 40):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                step.run()
        graphs[n] = g
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
for n, g in graphs.items():
    first = []
    for _ in range(4):                                            # the first replays of a graph, one by one (bench.py --steps 20 times the SECOND)
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); first.append((time.perf_counter() - t0) * 1e6 / n)
    print("graph of %2d evaluations: replays 1-4, us per evaluation: %s" % (n, " ".join("%.2f" % v for v in first)))
    host, evs = [], []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record(); g.replay(); e1.record()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        host.append((time.perf_counter() - t0) * 1e6); evs.append(e0.elapsed_time(e1) * 1e3)
    spin = []
    for _ in range(15):                                           # the same, the end found by polling an event instead of a blocking synchronize
        ev = torch.cuda.Event()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay(); ev.record()
        while not ev.query():
            pass
        torch.cuda.synchronize()
        spin.append((time.perf_counter() - t0) * 1e6)
    print("graph of %2d evaluations: host, end found by polling an event: %.1f us (%.2f per evaluation)" % (n, np.median(spin), np.median(spin) / n))
    print("graph of %2d evaluations: host %.1f us (%.2f per evaluation), events %.1f us (%.2f per evaluation), enqueue %.1f us" % (
        n, np.median(host), np.median(host) / n, np.median(evs), np.median(evs) / n, t_enq * 1e6))
