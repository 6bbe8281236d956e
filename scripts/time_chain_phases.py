#!/usr/bin/env python3
"""k_bw_chain by prefixes (development aid): the value + gradient evaluation of a config with the chain kernels leaving after phase N
(IWVI_CHAIN_EXIT; results are wrong then, the timing of the prefix is not), per-kernel time from torch's profiler-free hipEvents around a
graph of the two chain launches is not available here -- so: rocprofv3-free estimate from the whole evaluation's replay time."""
import os, sys, time
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, backward, settings, synthetic

cfg = CONFIGS[int(sys.argv[1]) if len(sys.argv) > 1 else 2]
dev = torch.device("cuda:0")
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
settings.set_seed(1)
m = synthetic.build_model(spec, dev)
names = {0: "whole", 10: "inputs in LDS", 11: "heads", 1: "thin sums (dq_mu, dW)", 2: "phase 1: da", 3: "phase 2: dk", 4: "products over samples", 5: "kernel adjoint"}
base = None
if os.environ.get("P5_F32") == "1":                          # phase 5 on fp32 operands (the form before round 6)
    _abi.set_debug_option("IWVI_BW_P5_F32", 1)
for ex in (10, 11, 1, 2, 3, 4, 5, 0):
    _abi.set_debug_option("IWVI_CHAIN_EXIT", ex)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            backward.iw_elbo_and_gradients(m)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            backward.iw_elbo_and_gradients(m)
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50 * 1e6
    print("chains leave after %-24s value + gradient %7.1f us" % (names[ex], dt))
_abi.set_debug_option("IWVI_CHAIN_EXIT", 0)
