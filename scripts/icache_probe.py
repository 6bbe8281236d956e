#!/usr/bin/env python3
"""Is the forward kernel paying for a cold instruction cache?  Twice the samples -> two workgroups per CU, one after the other."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic
dev = torch.device("cuda:0")
cfg = dict(CONFIGS[2]); cfg["K"] = cfg["K"] * 2
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
lib = _abi.lib()
lib.iwvi_debug_set_stamps.restype = None
lib.iwvi_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
NW = 32768
buf = torch.zeros(NW * 128, dtype=torch.int64, device=dev)
m.precompute(with_encoders=True)
for _ in range(3): m._fused_forward(B * K, K, B, (B, K))
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(buf.data_ptr(), NW)
m._fused_forward(B * K, K, B, (B, K))
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(None, 0)
full = buf.view(NW, 128).cpu().numpy(); full = full[full[:, 0] > 0]
s = full[:, :64].astype(np.float64)
print("workgroups:", len(s))
t0 = s[:, 0].min(); start = (s[:, 0] - t0) * 10e-3
early = start < np.median(start) - 5; late = start > np.median(start) + 5
print("early %d (start med %.1f us), late %d (start med %.1f us)" % (early.sum(), np.median(start[early]), late.sum(), np.median(start[late])))
names = {1: "input"}
for li, l in enumerate(spec["layers"]):
    ph = ["", "lv.mlp", "", "", "", "lv.out"] if l["type"] == "lv" else ["gp.xt", "gp.gram", "gp.stage1", "gp.stage2", "gp.epi1", "gp.epi2"]
    for k, n in enumerate(ph):
        if n: names[2 + li * 6 + k] = "L%d %s" % (li, n)
names[63] = "end"
prev = 0
for k in sorted(names):
    d = (s[:, k] - s[:, prev]) * 10e-3
    print("%-14s early %6.2f us   late %6.2f us" % (names[k], np.median(d[early]), np.median(d[late])))
    prev = k
print("span early %.2f late %.2f" % (np.median((s[:, 63] - s[:, 0])[early]) * 10e-3, np.median((s[:, 63] - s[:, 0])[late]) * 10e-3))
