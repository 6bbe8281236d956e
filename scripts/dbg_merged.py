import sys, torch, numpy as np
sys.path.insert(0, '.')
from dgps_with_iwvi_amd import synthetic, settings
dev = torch.device('cuda:0')
spec = synthetic.make_spec(L=2, M=128, B=64, K=20, R=5, with_lv=True, seed=11)
zs = [torch.as_tensor(z, dtype=torch.float32, device=dev) for z in synthetic.make_noise(spec, seed=12)]
res = []
for merged in (False, True):
    m = synthetic.build_model(spec, dev)
    B, K = 64, 20
    if not merged:
        m.precompute(with_encoders=True)
    logw, outs, _ = m._fused_forward(B * K, K, B, (B, K), zs=zs, sampled_kl=True, want_layers=True, merged=merged)
    torch.cuda.synchronize()
    res.append((logw.clone(), [{k: v.clone() for k, v in o.items()} for o in outs], [l.state().buf.clone() for l in m.layers if hasattr(l, 'q_sqrt')]))
    if merged:
        print("sync words", m._fz_ws().view(torch.int32)[:8].tolist())
a, b = res
print("logw maxdiff", (a[0] - b[0]).abs().max().item(), a[0][:4].tolist(), b[0][:4].tolist())
for i, (oa, ob) in enumerate(zip(a[1], b[1])):
    for k in oa:
        print("layer", i, k, (oa[k] - ob[k]).abs().max().item())
import ctypes
from dgps_with_iwvi_amd import _abi
for i, (sa, sb) in enumerate(zip(a[2], b[2])):
    l = [l for l in synthetic.build_model(spec, dev).layers if hasattr(l, 'q_sqrt')][i]
    offs = (ctypes.c_size_t * 8)()
    _abi.lib().iwvi_gp_state_offsets(l.num_inducing, l.num_outputs, offs)
    names = ["Lm", "Linv", "LsP", "LrTP", "QmuP", "ZtP", "cst", "kl"]
    o = list(offs) + [sa.numel()]
    for n, lo, hi in zip(names, o[:-1], o[1:]):
        d = (sa[lo:hi] != sb[lo:hi]).sum().item()
        print("state", i, n, "bytes differing", d, "of", hi - lo)
