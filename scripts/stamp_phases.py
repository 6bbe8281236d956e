#!/usr/bin/env python3
"""Phase timeline of the fused forward kernel from in-kernel wall-clock stamps (development aid)."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (read before HIP initialises)
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic

ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2)
ap.add_argument("--sustained", type=int, default=0, help="replay a graph of 20 evaluations this many times right before the stamped launch (the clocks of a busy device)")
args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.config]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
lib = _abi.lib()
lib.iwvi_debug_set_stamps.restype = None
lib.iwvi_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
NW = 32768
buf = torch.zeros(NW * 128, dtype=torch.int64, device=dev)
m.precompute(with_encoders=True)
el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False)   # the bound's own evaluation: the variant bench.py times
for _ in range(3):
    m._fused_forward(B * K, K, B, (B, K), elbo=el)
torch.cuda.synchronize()
if args.sustained:
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(20):
                m.precompute(with_encoders=True)
                m._fused_forward(B * K, K, B, (B, K), elbo=el)
        for _ in range(args.sustained):
            gr.replay()
        lib.iwvi_debug_set_stamps(buf.data_ptr(), NW)
        m._fused_forward(B * K, K, B, (B, K), elbo=el)
    torch.cuda.synchronize()
    lib.iwvi_debug_set_stamps(None, 0)
lib.iwvi_debug_set_stamps(buf.data_ptr(), NW) if not args.sustained else None
if not args.sustained:
    m._fused_forward(B * K, K, B, (B, K), elbo=el)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(None, 0)
full = buf.view(NW, 128).cpu().numpy()
full = full[full[:, 0] > 0]
s, cyc = full[:, :64], full[:, 64:]
print("workgroups stamped:", len(s))
t0 = s[:, 0].min()
names = {0: "start", 1: "input"}
for li, l in enumerate(spec["layers"]):
    ph = ["", "lv.mlp", "", "", "", "lv.out"] if l["type"] == "lv" else ["gp.xt", "gp.gram", "gp.stage1", "gp.stage2", "gp.epi1", "gp.epi2"]
    for k, n in enumerate(ph):
        if n: names[2 + li * 6 + k] = "L%d %s" % (li, n)
names[61] = "logw"; names[62] = "tail.lse"; names[63] = "tail.arrive"
for k, n in ((56, "p.ltab+x"), (40, "p.item-found"), (57, "p.noise-drawn"), (58, "p.(merged)"), (59, "p.vmcnt0")):
    print("%-14s at med %6.2f us after start" % (n, np.median(s[:, k] - s[:, 0]) * 10e-3))
prev = None
for k in sorted(names):
    col = s[:, k]
    if prev is not None:
        d = (col - s[:, prev]) * 10e-3     # 100 MHz ticks -> us
        dc = (cyc[:, k] - cyc[:, prev]).astype(np.float64)
        print("%-14s  dur med %6.2f us  max %6.2f  (%7.0f clk, %.2f GHz) | end at med %6.2f us" % (
            names[k], np.median(d), d.max(), np.median(dc), np.median(dc) / max(np.median(d), 1e-9) * 1e-3,
            np.median(col - t0) * 10e-3))
    prev = k
print("kernel span: %.2f us" % ((s[:, 63].max() - t0) * 10e-3))

w0 = full[:, 100:108].astype(np.float64); w1 = full[:, 110:118].astype(np.float64)
st = cyc[:, 2 + 1 * 6 + 2].astype(np.float64)[:, None]      # stage-1 end stamp (cycles) of layer 1
print("stage 2 (layer 1) per wave: start / end cycles after the stage-1 barrier (median over workgroups)")
print("  start", np.round(np.median(w0 - st, 0)).astype(int))
print("  end  ", np.round(np.median(w1 - st, 0)).astype(int))

s1 = full[:, 118:126].astype(np.float64); g0 = cyc[:, 2 + 1 * 6 + 1].astype(np.float64)[:, None]
print("stage 1 (layer 1) per wave end, cycles after the Gram barrier:", np.round(np.median(s1 - g0, 0)).astype(int))


