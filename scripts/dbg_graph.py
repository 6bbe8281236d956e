import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import synthetic, backward
from dgps_with_iwvi_amd.training import Trainer
dev = torch.device("cuda:0")
spec = synthetic.make_spec(L=2, M=128, K=20, B=1024, with_lv=True, seed=0)
model = synthetic.build_model(spec, dev)
def timed(fn, n=20):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tr = Trainer(model, check_finite=False)
print("eager: natgrad_op %.3f  adam_op %.3f  grad %.3f" % (timed(lambda: tr.natgrad_op()), timed(lambda: tr.adam_op()), timed(lambda: backward.iw_elbo_and_gradients(model))))
trc = Trainer(model, check_finite=True)
print("eager with check: adam_op %.3f" % timed(lambda: trc.adam_op()))
trg = Trainer(model, use_graph=True)
trg.step(); trg.step(); torch.cuda.synchronize()
print("graph: ng replay %.3f  adam replay %.3f  step %.3f" % (timed(trg._graphs["ng"][1].replay), timed(trg._graphs["adam"][1].replay), timed(trg.step)))
import ctypes
from dgps_with_iwvi_amd import _abi
f = tr.final
dq_mu = torch.zeros_like(f.q_mu); dq_sqrt = torch.zeros_like(f.q_sqrt)
print("natgrad step alone %.3f" % timed(lambda: _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(f.q_mu), _abi.ptr(f.q_sqrt), _abi.ptr(dq_mu), _abi.ptr(dq_sqrt), f.num_inducing, f.num_outputs, 0.0, tr._ng_ws.data_ptr(), _abi.stream_ptr()))))
