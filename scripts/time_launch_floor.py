#!/usr/bin/env python3
"""How much of the fused forward's launch duration is fixed cost?  Times graphs of 20 back-to-back launches of
k_dgp_forward with the kernel leaving (1) at once, (2) after its prologue, (0) normally (development aid)."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic
dev = torch.device("cuda:0")
cfg = CONFIGS[2]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
lib = _abi.lib()
lib.iwvi_debug_set_exit.restype = None
lib.iwvi_debug_set_exit.argtypes = [ctypes.c_int]
m.precompute(with_encoders=True)
for phase in (1, 2, 0):
    lib.iwvi_debug_set_exit(phase)
    fwd = lambda: m._fused_forward(B * K, K, B, (B, K), elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False))
    fwd(); torch.cuda.synchronize()
    s = torch.cuda.Stream(device=dev); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(20): keep = fwd()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 20 * 1e3)
    print("exit phase %d: %.2f us per launch (median of 20 graph replays of 20 launches)" % (phase, np.median(ts)))
lib.iwvi_debug_set_exit(0)
