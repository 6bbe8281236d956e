"""One NatGrad step (iwvi_natgrad_step, M = 128, R = 1) as hipGraph replays of 20 steps: the default route (spread over four launches) and the
one-workgroup kernel (IWVI_NG_ONE_WG).  Development aid."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import _abi
dev = torch.device("cuda:0")
M, R = 128, 1
g = torch.Generator().manual_seed(0)
q_mu0 = torch.randn(M, R, generator=g).to(dev)
q_sqrt0 = (torch.tril(torch.randn(R, M, M, generator=g)) * 0.05 + torch.eye(M)).to(dev)
dq_mu = (torch.randn(M, R, generator=g) * 1e-2).to(dev)
dq_sqrt = (torch.tril(torch.randn(R, M, M, generator=g)) * 1e-2).to(dev)
ws = torch.empty(_abi.lib().iwvi_natgrad_ws_bytes(M), dtype=torch.uint8, device=dev)
for one in (1, 0, 1, 0):
    _abi.set_debug_option("IWVI_NG_ONE_WG", one)
    q_mu, q_sqrt = q_mu0.clone(), q_sqrt0.clone()
    def step():
        _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(q_mu), _abi.ptr(q_sqrt), _abi.ptr(dq_mu), _abi.ptr(dq_sqrt), M, R, 1e-6, ws.data_ptr(), _abi.stream_ptr()))
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
            for _ in range(20):
                step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    for _ in range(3): gr.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): gr.replay()
    torch.cuda.synchronize()
    print("%-28s %6.1f us per step" % ("one workgroup" if one else "spread over four launches", (time.perf_counter() - t0) / 200 * 1e6))
_abi.set_debug_option("IWVI_NG_ONE_WG", 0)
