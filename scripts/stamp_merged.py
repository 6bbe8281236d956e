#!/usr/bin/env python3
"""Timeline of the merged launch (iwvi_dgp_forward_fused) from in-kernel wall-clock stamps: the factorising workgroups' phases, the
chunk workgroups' front, their waits for the factorisation, and the layers behind it (development aid)."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import argparse, ctypes, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic, settings

ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.config]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
lib = _abi.lib()
lib.iwvi_debug_set_stamps.restype = None
lib.iwvi_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
NW = 4096
buf = torch.zeros(NW * 128, dtype=torch.int64, device=dev)
settings.merged_launch = True
for _ in range(3):
    m._build_likelihood(None)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(buf.data_ptr(), NW)
m._build_likelihood(None)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(None, 0)
full = buf.view(NW, 128).cpu().numpy()
full = full[full[:, 50] >= 1000]
role = full[:, 50] - 1000
n_gp = sum(1 for l in spec["layers"] if l["type"] == "gp")
fac, fw = full[role < n_gp], full[role >= n_gp]
print("rows stamped:", len(full), "roles:", np.unique(role)[:6], "...")
if len(fac) and fac[:, 63].max() > 0:
    print("(resume mode: the factorising workgroups carry a chunk too; their forward stamps overwrite words 0, 1 of their rows)")
t0 = min(fac[:, 0].min() if len(fac) else 1 << 62, fw[:, 0].min())
us = lambda x: (x - t0) * 1e-2
print("workgroups: %d factorising, %d chunks" % (len(fac), len(fw)))
for r in fac:
    if r[63] > 0:
        print("factor ticket %d (resumed a chunk): published %.2f | snapshot in LDS %.2f | at the wait %.2f | factorisation seen %.2f | first GP layer done %.2f | log-weights %.2f us" % (
            r[50] - 1000, us(r[3]), us(r[1]), us(r[60]), us(r[61]), us(r[2 + [i for i, l in enumerate(spec["layers"]) if l["type"] == "gp"][0] * 6 + 5]), us(r[63])))
        continue
    print("factor ticket %d: start %.2f | centred %.2f | gen(0,1) done %.2f | column %d pass: %.2f -> %.2f -> %.2f -> %.2f | published %.2f us" % (
        r[50] - 1000, us(r[0]), us(r[1]), us(r[7]), 1, us(r[10]), us(r[11]), us(r[12]), us(r[13]), us(r[3])))
fgp = [i for i, l in enumerate(spec["layers"]) if l["type"] == "gp"][0]
def col(k): return us(fw[:, k])
def show(name, a):
    a = np.sort(a)
    print("%-34s med %6.2f  min %6.2f  max %6.2f   3 latest %s" % (name, np.median(a), a[0], a[-1], np.round(a[-3:], 2)))
show("chunk start", col(0))
show("table/inputs barrier", col(56))
show("independent copies + noise issued", col(57))
show("early flag seen, dependent copies", col(58))
show("prologue done", col(1))
show("front done (Gram computed)", col(60))
show("factorisation seen", col(61))
show("first GP layer: stage 1 done", col(2 + fgp * 6 + 2))
show("first GP layer: stage 2 done", col(2 + fgp * 6 + 3))
show("first GP layer: done", col(2 + fgp * 6 + 5))
last = len(spec["layers"]) - 1
show("last layer: done", col(2 + last * 6 + 5))
show("log-weights written", col(63))
n_pack = sum(l["q_mu"].shape[1] for l in spec["layers"] if l["type"] == "gp")
for r in full[(role >= n_gp) & (role < n_gp + n_pack)]:
    print("ticket %d (pack job): came to life %.2f | pack %.2f -> %.2f | chunk start %.2f | log-weights %.2f us" % (r[50] - 1000, us(r[52]), us(r[53]), us(r[54]), us(r[0]), us(r[63])))
    p_ = [(r[30 + k] - r[30]) * 1e-2 for k in range(6)]
    print("      inside: q_sqrt in LDS +%.2f | fp32 image + KL terms +%.2f | tree +%.2f | max +%.2f | S16 image +%.2f us" % tuple(p_[1:]))
for r in full[(role >= n_gp + n_pack) & (role < n_gp + n_pack + n_gp)]:
    print("ticket %d (helper): other chunk's x~ done %.2f, in HBM %.2f | own chunk: start %.2f | prologue done %.2f | front done %.2f | factorisation seen %.2f | log-weights %.2f us" % (
        r[50] - 1000, us(r[53]), us(r[54]), us(r[0]), us(r[1]), us(r[60]), us(r[61]), us(r[63])))
o_ = np.argsort(fw[:, 0])
print("first chunk workgroups to start (ticket, came to life, forward start, table barrier):", [(int(fw[i, 50] - 1000), round(float(us(fw[i, 52])), 2), round(float(us(fw[i, 0])), 2), round(float(us(fw[i, 56])), 2)) for i in o_[:5]])
print("came to life: min %.2f median %.2f max %.2f" % (us(full[:, 52]).min(), np.median(us(full[:, 52])), us(full[:, 52]).max()))
lw_ = np.argsort(fw[:, 63])[-6:]
print("last to write their log-weights (ticket, start, front done, seen, stage-1 done, log-weights):",
      [(int(fw[i, 50] - 1000), round(float(us(fw[i, 0])), 1), round(float(us(fw[i, 60])), 1), round(float(us(fw[i, 61])), 1), round(float(us(fw[i, 2 + fgp * 6 + 2])), 1), round(float(us(fw[i, 63])), 1)) for i in lw_])
late = np.argsort(full[:, 52])[-6:]
print("latest to come to life (ticket, block, XCC, us):", [(int(full[i, 50] - 1000), int(np.flatnonzero(buf.view(NW, 128).cpu().numpy()[:, 52] == full[i, 52])[0]), int(full[i, 51]), round(float(us(full[i, 52])), 2)) for i in late])
print("workgroups per XCC:", np.bincount(full[:, 51].astype(int), minlength=8))
print("span (first start -> last stamp 63): %.2f us" % us(fw[:, 63]).max())
