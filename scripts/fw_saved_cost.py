"""Development aid: what the layer launch of a value + gradient evaluation pays for what it leaves in HBM (configs[2]): the bound alone, + the
adjoint heads, + the final layer's saved operands (the natural-gradient op), + every layer's (the Adam op).   python scripts/fw_saved_cost.py"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import CONFIGS
from dgps_with_iwvi_amd import _abi, synthetic, settings
dev = torch.device("cuda:0")
cfg = CONFIGS[2]
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
m = synthetic.build_model(spec, dev)
B, K = cfg["B"], cfg["K"]; T = B * K
L = len(m.layers)
m.precompute(with_encoders=True)
ft = settings.float_type
w = torch.empty(T, dtype=ft, device=dev); dm = torch.empty(T, 1, dtype=ft, device=dev); dv = torch.empty(T, 1, dtype=ft, device=dev)
sums = torch.empty(3, dtype=torch.float64, device=dev)
el = lambda adj: dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False, **({"adj": dict(w=w, d_mean=dm, d_var=dv, sums=sums)} if adj else {}))
zf = [None] * L
cases = [("bound only", lambda: m._fused_forward(T, K, B, (B, K), elbo=el(False))),
         ("+ adjoint heads", lambda: m._fused_forward(T, K, B, (T,), zs=zf, sampled_kl=True, want_logw=True, elbo=el(True))),
         ("+ final layer saved (NatGrad op)", lambda: m._fused_forward(T, K, B, (T,), zs=zf, sampled_kl=True, want_layers=True, want_logw=True, want_saved=True, elbo=el(True), outputs_for={L - 1}, moments=False)),
         ("+ every layer saved (Adam op)", lambda: m._fused_forward(T, K, B, (T,), zs=zf, sampled_kl=True, want_layers=True, want_logw=True, want_saved=True, elbo=el(True)))]
for name, fn in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(20): keep = fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); 
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%-36s %6.1f us per launch (variant 0x%x)" % (name, e0.elapsed_time(e1) / 100 * 1e3, int(_abi.lib().iwvi_debug_last_forward_variant())))
