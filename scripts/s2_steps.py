import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import ctypes, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic
dev = torch.device("cuda:0")
cfg = CONFIGS[2]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
lib = _abi.lib()
lib.iwvi_debug_set_stamps.restype = None
lib.iwvi_debug_set_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int64]
NW = 32768
buf = torch.zeros(NW * 128, dtype=torch.int64, device=dev)
m.precompute(with_encoders=True)
el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False)
for _ in range(3):
    m._fused_forward(B * K, K, B, (B, K), elbo=el)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(buf.data_ptr(), NW)
m._fused_forward(B * K, K, B, (B, K), elbo=el)
torch.cuda.synchronize()
lib.iwvi_debug_set_stamps(None, 0)
full = buf.view(NW, 128).cpu().numpy()
for wg in (0, 100, 200):
    base = full[wg, 64 + 2 + 1*6 + 2]   # stage-1 end stamp cycles layer 1
    print("WG", wg, "wave start", (full[wg, 100:108] - base), "end", (full[wg, 110:118] - base))
    for w in range(8):
        row = full[256 + wg * 8 + w]      # (configs[2]: 256 workgroups; per-wave rows start behind them)
        st = row[:32][row[:32] > 0] - base
        print("  wave", w, "steps at", st.tolist(), "deltas", np.diff(st).tolist(), "| gram start/end, prefetch issued, barrier passed:", (row[32:36] - row[32]).tolist(), "| mean job", int(row[37] - row[36]), "| epi: i start, i end, barrier, k loop, d loop, outputs, end, barrier", (row[40:48] - row[40]).tolist())
