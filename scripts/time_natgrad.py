"""Kernel-level view of one NatGrad step (iwvi_natgrad_step) at M = 128, R = 1 (development aid; run under scripts/prof_any.sh)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import _abi
dev = torch.device("cuda:0")
M, R = 128, 1
g = torch.Generator().manual_seed(0)
q_mu = torch.randn(M, R, generator=g).to(dev)
q_sqrt = (torch.tril(torch.randn(R, M, M, generator=g)) * 0.05 + torch.eye(M)).to(dev)
dq_mu = (torch.randn(M, R, generator=g) * 1e-2).to(dev)
dq_sqrt = (torch.tril(torch.randn(R, M, M, generator=g)) * 1e-2).to(dev)
ws = torch.empty(_abi.lib().iwvi_natgrad_ws_bytes(M), dtype=torch.uint8, device=dev)
def step():
    _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(q_mu), _abi.ptr(q_sqrt), _abi.ptr(dq_mu), _abi.ptr(dq_sqrt), M, R, 1e-3, ws.data_ptr(), _abi.stream_ptr()))
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("natgrad step: %.1f us eager" % ((time.perf_counter() - t0) / 50 * 1e6))
