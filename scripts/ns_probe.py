"""Development aid: the layer kernel at an M = 512 stack with R latent GPs per inner layer, with the samples per workgroup capped
(IWVI_FW_MAX_NS) -- how the launch time scales with the operand reuse.   python scripts/ns_probe.py [--R 1] [--caps 3 4 5]"""
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import argparse, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import _abi, synthetic
ap = argparse.ArgumentParser()
ap.add_argument("--R", type=int, default=1); ap.add_argument("--M", type=int, default=512); ap.add_argument("--L", type=int, default=5)
ap.add_argument("--K", type=int, default=100); ap.add_argument("--B", type=int, default=8192); ap.add_argument("--caps", type=int, nargs="+", default=[3, 4, 5])
a = ap.parse_args()
dev = torch.device("cuda:0")
spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, L=a.L, M=a.M, K=a.K, B=a.B, R=a.R, with_lv=False)
m = synthetic.build_model(spec, dev)
B, K = a.B, a.K
m.precompute(with_encoders=True)
el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False)
for cap in a.caps:
    _abi.set_debug_option("IWVI_FW_MAX_NS", cap)
    for _ in range(2):
        m._fused_forward(B * K, K, B, (B, K), elbo=el)
    torch.cuda.synchronize()
    v = int(_abi.lib().iwvi_debug_last_forward_variant())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = m._fused_forward(B * K, K, B, (B, K), elbo=el)
    e1.record(); torch.cuda.synchronize()
    print("cap %d -> NS %d (variant 0x%x): %.3f ms per launch" % (cap, v & 0xff, v, e0.elapsed_time(e1) / 5))
_abi.set_debug_option("IWVI_FW_MAX_NS", 0)
