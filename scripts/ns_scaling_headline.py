"""Development aid (round 6): what a workgroup of FEWER samples costs at the headline stack.  L = 2, M = 128, LV layer,
K = 16 (divides every chunk size), B = 256 NS so that every run is exactly 256 workgroups -- one per CU -- of 16 NS samples: the launch
time as a function of NS separates what does not shrink with the samples (the latency chains: prologue, LV, x~, epilogues, tail, the
dependent solve) from what does (the MFMA phases).  Two phase-offset groups on a CU pay the first part twice, in parallel, and share the
second: T_two >= max(T(NS_a), T(NS_b)) + offset, and it can only win if T(3) + what T(2) adds beside it stays under T(5).
   python scripts/ns_scaling_headline.py"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import _abi, synthetic
dev = torch.device("cuda:0")
K = 16
rows = []
for ns in (5, 4, 3, 2, 1):
    B = 256 * ns
    spec = synthetic.make_spec(seed=0, parity=True, n_data=8192, L=2, M=128, K=K, B=B, with_lv=True)
    m = synthetic.build_model(spec, dev)
    m.precompute(with_encoders=True)
    el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False)
    _abi.set_debug_option("IWVI_FW_MAX_NS", ns)
    for _ in range(3):
        m._fused_forward(B * K, K, B, (B, K), elbo=el)
    torch.cuda.synchronize()
    v = int(_abi.lib().iwvi_debug_last_forward_variant())
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(20):
                m._fused_forward(B * K, K, B, (B, K), elbo=el)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = float(np.median(ts))
    rows.append((ns, t))
    print("NS %d (16 NS = %2d samples per workgroup, 256 workgroups, variant 0x%x, lean %d): %.2f us per launch" % (ns, 16 * ns, v, (v >> 10) & 3, t))
_abi.set_debug_option("IWVI_FW_MAX_NS", 0)
ns = np.array([r[0] for r in rows if r[0] <= 4], float); t = np.array([r[1] for r in rows if r[0] <= 4])
w, L = np.polyfit(ns, t, 1)
print("general variants, fit T(NS) = %.2f + %.2f NS us: %.0f %% of a five-sub-tile workgroup's time does not shrink with its samples" % (L, w, 100 * L / (L + 5 * w)))
print("two groups of 3 + 2 sub-tiles sharing one CU: each alone T(3) = %.2f, T(2) = %.2f; the CU's MFMA phases at most as fast as today -> no schedule beats max(T(3), L + 5 w) = %.2f us" % (L + 3 * w, L + 2 * w, max(L + 3 * w, L + 5 * w)))
