import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from bench import CONFIGS
    from dgps_with_iwvi_amd import synthetic
    cfgn = int(sys.argv[2])
    spec = synthetic.make_spec(seed=0, parity=True, n_data=8192, **dict(CONFIGS[cfgn], B=64, K=2))
    m = synthetic.build_model(spec, torch.device("cuda:0"))
    for _ in range(5): m.precompute()
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); m.precompute(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    print("cfg %d stop=%s  precompute med %.1f us min %.1f" % (cfgn, os.environ.get("IWVI_DEBUG_STOP", "-"), np.median(ts), min(ts)))
else:
    for cfgn in (2,):
        for stop in ("2", "31", "32", "3"):
            env = dict(os.environ, IWVI_DEBUG_STOP=stop)
            subprocess.run([sys.executable, __file__, "child", str(cfgn)], env=env)
