import sys, torch, numpy as np
sys.path.insert(0, '.')
from dgps_with_iwvi_amd import synthetic, settings
dev = torch.device('cuda:0')
bad = 0
for (B, K, M, L) in [(1024, 20, 128, 2), (64, 20, 128, 2), (520, 7, 96, 3), (1024, 5, 128, 2)]:
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, R=5, with_lv=(K != 5), seed=3)
    res = []
    for merged in (False, True):
        settings.merged_launch = merged
        settings.set_seed(99)
        m = synthetic.build_model(spec, dev)
        gen = torch.Generator().manual_seed(1)
        vals = torch.empty(1500, dtype=torch.float64, device=dev)
        for it in range(1500):
            vals[it] = m._build_likelihood(None)
            if it % 7 == 0:
                for l in m.layers:
                    if hasattr(l, "q_sqrt"):
                        l.q_mu.add_(0.01 * torch.randn(l.q_mu.shape, generator=gen).to(dev))
                        l._Z().add_(0.003 * torch.randn(l._Z().shape, generator=gen).to(dev))
        torch.cuda.synchronize()
        res.append(vals.cpu())
        if merged:
            w = m._fz_ws().view(torch.int32)
            print("sync words gen/role/early/pack/done/timeout:", w[:6].tolist())
    eq = torch.equal(res[0], res[1])
    nd = int((res[0] != res[1]).sum())
    print((B, K, M, L), "bitwise equal over 1500 evaluations:", eq, "differing:", nd, "finite:", bool(torch.isfinite(res[1]).all()))
    bad += (not eq)
settings.merged_launch = False
print("FAIL" if bad else "OK")
