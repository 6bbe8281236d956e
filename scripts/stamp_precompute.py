#!/usr/bin/env python3
"""Phase timeline of the factorisation role of k_precompute (development aid)."""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (read before HIP initialises)
import argparse, ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--p", type=int, default=1); args = ap.parse_args()
os.environ["IWVI_PRE_STAMP_P"] = str(args.p)
dev = torch.device("cuda:0")
spec = synthetic.make_spec(seed=0, parity=True, n_data=8192, **dict(CONFIGS[args.config], B=64, K=2))
m = synthetic.build_model(spec, dev)
lib = _abi.lib()
lib.iwvi_debug_set_pre_stamps.restype = None
lib.iwvi_debug_set_pre_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 16, dtype=torch.int64, device=dev)
for _ in range(3): m.precompute()
torch.cuda.synchronize()
lib.iwvi_debug_set_pre_stamps(buf.data_ptr())
rows = []
for _ in range(20):
    m.precompute(); torch.cuda.synchronize()
    rows.append(buf.view(8, 16).cpu().numpy().copy())
lib.iwvi_debug_set_pre_stamps(None)
r = np.stack(rows)[:, 0, :7].astype(np.float64)      # layer 0
names = ["Zs+centre", "-", "Gram+cholesky (+Z~, zmax beside the last pass)", "-", "-", "(dense)"]
d = np.diff(r, axis=1) * 10e-3
for n, col in zip(names, d.T):
    print("%-10s med %6.2f us" % (n, np.median(col)))
print("total      med %6.2f us" % np.median((r[:, 6] - r[:, 0]) * 10e-3))
r2 = np.stack(rows)[:, 0, :].astype(np.float64)
for a, b, n in ((10, 11, "p=%d diagonal pass (w0)" % args.p), (10, 14, "   rows loaded"), (14, 15, "   16 columns"), (15, 11, "   stored"), (11, 12, "p=%d wait for the other waves" % args.p), (12, 13, "p=%d rows below" % args.p)):
    print("%-24s med %6.2f us" % (n, np.median((r2[:, b] - r2[:, a]) * 10e-3)))

print("%-24s med %6.2f us" % ("start of chol -> this pass", np.median((r2[:, 10] - r2[:, 2]) * 10e-3)))
print("%-24s med %6.2f us" % ("Gram block columns 0, 1 (up front)", np.median((r2[:, 7] - r2[:, 2]) * 10e-3)))
