import sys, torch
sys.path.insert(0, '.')
from dgps_with_iwvi_amd import synthetic, settings
dev = torch.device("cuda:0")
for cfg in (dict(L=2, M=128, K=4, B=16), dict(L=3, M=256, K=4, B=16), dict(L=5, M=512, K=2, B=8), dict(L=2, M=128, K=4, B=16, Dx=1), dict(L=2, M=224, K=4, B=16, Dx=1),
            dict(L=2, M=64, K=4, B=16, Dx=1), dict(L=2, M=32, K=4, B=16, Dx=1), dict(L=2, M=128, K=4, B=16, Dx=2), dict(L=2, M=128, K=4, B=16, Dx=3), dict(L=2, M=128, K=4, B=16, Dx=4), dict(L=2, M=256, K=4, B=16, Dx=4), dict(L=2, M=512, K=4, B=16, Dx=5)):
    spec = synthetic.make_spec(seed=3, parity=True, n_data=4096, with_lv=False, **cfg)
    m = synthetic.build_model(spec, dev)
    print(cfg, [("%.1f" % r["diag_ratio"]) for r in m.autotune_f64()])
