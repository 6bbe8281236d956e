"""Per-wave sub-phase clocks of the super-block solve (stage 1, M > 240) of layer 1 -- needs a library built with -DIWVI_S2_STEP_STAMPS.
   python scripts/s1_sb_steps.py [--config 4]"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import argparse, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=4); args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.config]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
lib = _abi.lib()
NW = 32768
buf = torch.zeros(NW * 128, dtype=torch.int64, device=dev)
m.precompute(with_encoders=True)
el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False)
for _ in range(2):
    m._fused_forward(B * K, K, B, (B, K), elbo=el)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(buf.data_ptr(), NW)
m._fused_forward(B * K, K, B, (B, K), elbo=el)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(None, 0)
full = buf.view(NW, 128).cpu().numpy()
M = cfg["M"]; nsb = (M // 16 + 7) // 8
NWG = (B * K + 16 * min(5, max(1, (B * K + 4095) // 4096)) - 1) // (16 * min(5, max(1, (B * K + 4095) // 4096)))
R0 = NWG                                                          # per-wave rows start behind the workgroups' own (csrc/dgp_forward.hip: DBG_WSTAMP)
NSW = int(round(B * K / max(1, int((full[:NWG + 8, 0] > 0).sum())) / 16.0))
R0 = int((full[:17100, 0] > 0).sum())                              # = the launch's workgroup count
wgs = [w for w in range(64, 1900, 23) if full[R0 + w * 8, 48] > 0]
print("workgroups sampled:", len(wgs), " super-blocks:", nsb)
names = ["dense", "wait1", "tri", "wait2", "write", "wait3"]
tot = np.zeros(6)
for I in range(nsb):
    rows = []
    for w in range(8):
        st = np.array([[full[R0 + wg * 8 + w, 48 + 6 * I + k] for k in range(6)] + [full[R0 + wg * 8 + w, 48 + 6 * (I + 1)] if I + 1 < nsb else full[wg, 64 + 2 + 1 * 6 + 2]] for wg in wgs], dtype=np.float64)
        d = np.diff(st, axis=1)
        rows.append(np.median(d, axis=0))
    rows = np.array(rows)
    print("super-block %d (clocks, median over workgroups; rows = waves 0..7: block row rw = w < 4 ? w : 11 - w)" % I)
    for w in range(8):
        print("   wave %d  " % w + "  ".join("%s %6d" % (n, v) for n, v in zip(names, rows[w])))
    print("   max over waves of (dense, tri, write): %d %d %d ; sum of the phase maxima incl. waits: %d" % (rows[:, 0].max(), rows[:, 2].max(), rows[:, 4].max(), rows.sum(1).max()))
    tot += rows.max(0)
s1 = np.median([full[wg, 64 + 2 + 6 + 2] - full[wg, 64 + 2 + 6 + 1] for wg in wgs])
print("stage 1 of layer 1: %d clocks (median)" % s1)
