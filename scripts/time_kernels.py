#!/usr/bin/env python3
"""Per-stage timings (HIP events, median of N) of one IW-ELBO evaluation; development aid."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import synthetic
from dgps_with_iwvi_amd.layers import GPLayer

def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return float(np.median(ts)), float(np.min(ts))

ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); args = ap.parse_args()
dev = torch.device("cuda:0")
cfg = CONFIGS[args.config]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]
print("config", args.config, spec["name"])
print("precompute      med %.1f us  min %.1f" % timeit(m.precompute))
zs = [torch.randn(B, K, (l["latent_dim"] if l["type"] == "lv" else l["q_mu"].shape[1]), device=dev) for l in spec["layers"]]
X = m.X[:, None, :].expand(B, K, -1).contiguous(); Y = m.Y[:, None, :].expand(B, K, -1).contiguous()
XY = torch.cat([X, Y], -1)
F = X
for i, (layer, z) in enumerate(zip(m.layers, zs)):
    f = lambda: layer.propagate(F, inference_amorization_inputs=XY, is_sampled_local_regularizer=True, z=z, _precomputed=True)
    print("layer %d %-22s med %.1f us  min %.1f" % ((i, type(layer).__name__) + timeit(f)))
    F = f()[0]
print("full ELBO eager med %.1f us  min %.1f" % timeit(lambda: m._build_likelihood(zs)))
