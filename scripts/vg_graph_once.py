"""One value + gradient evaluation (backward.iw_elbo_and_gradients) captured into a hipGraph and replayed a few times -- for kernel traces
(rocprofv3 --kernel-trace; scripts/timeline_vg.py prints the last replay).   python scripts/vg_graph_once.py [--config 2] [--replays 6]"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import argparse, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import backward, synthetic
ap = argparse.ArgumentParser(); ap.add_argument("--config", type=int, default=2); ap.add_argument("--replays", type=int, default=6)
ap.add_argument("--no-fuse", action="store_true", help="the bound's adjoint heads by iwvi_iw_elbo_backward (two more launches) instead of the layer launch's tail")
ap.add_argument("--no-fuse-lv", action="store_true", help="the latent-variable layer's and the encoder's adjoints as two launches (iwvi_lv_layer_backward, iwvi_encoder_backward_act)")
a = ap.parse_args()
dev = torch.device("cuda:0")
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **CONFIGS[a.config])
model = synthetic.build_model(spec, dev)
import functools
_vg = backward.iw_elbo_and_gradients
backward.iw_elbo_and_gradients = functools.partial(_vg, fuse_heads=not a.no_fuse)
if a.no_fuse_lv:
    backward.lv_backward = functools.partial(backward.lv_backward, fused=False)
backward.iw_elbo_and_gradients(model); backward.iw_elbo_and_gradients(model)
torch.cuda.synchronize()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    backward.iw_elbo_and_gradients(model)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        keep = backward.iw_elbo_and_gradients(model)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.replays):
    g.replay()
torch.cuda.synchronize()
print("value + gradient, graph replay: %.4f ms" % ((time.perf_counter() - t0) / a.replays * 1e3))
reps = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    reps.append((time.perf_counter() - t0) / 50 * 1e3)
print("value + gradient, graph replay, median of 5 x 50: %.4f ms  (fused heads: %s)" % (sorted(reps)[2], not a.no_fuse))
