"""A/B of host-side routes of the value + gradient evaluation that are chosen by an environment variable when the evaluation is queued
(IWVI_BW_BRANCH_ORDER, IWVI_BW_DENSE, IWVI_BW_PREPARE ...): one captured graph per setting, replayed alternately in ONE process; the
gradients of the settings compared on the same injected noise.
Usage: python scripts/ab_bw_env.py VAR=a,b [VAR2=c,d ...] [--config 2] [--rounds 5] [--iters 50]   (the cross product is timed)"""
import argparse
import itertools
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import backward, synthetic   # noqa: E402
from time_backward import CONFIGS   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("vars", nargs="+")
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    names = [v.split("=")[0] for v in a.vars]
    values = [v.split("=")[1].split(",") for v in a.vars]
    settings_ = [dict(zip(names, combo)) for combo in itertools.product(*values)]
    dev = torch.device("cuda:0")
    spec = synthetic.make_spec(**CONFIGS[a.config], seed=0)
    model = synthetic.build_model(spec, dev)
    B, K = spec["B"], spec["K"]
    gen = torch.Generator(device="cpu").manual_seed(7)
    zs = [torch.randn(B, K, getattr(l, "latent_dim", None) or l.num_outputs, generator=gen).to(dev) for l in model.layers]
    ref = None
    graphs = []
    for st in settings_:
        os.environ.update(st)
        e, g = backward.iw_elbo_and_gradients(model, zs=zs)
        torch.cuda.synchronize()
        cur = {k: v.double().cpu() for k, v in g.items()}
        if ref is None:
            ref = cur
        worst = max(float((cur[k] - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-30)) for k in ref)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                backward.iw_elbo_and_gradients(model)
        torch.cuda.current_stream().wait_stream(side)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            out = backward.iw_elbo_and_gradients(model)
        graphs.append((st, gr, out))
        print("%s: bound %.6f, largest relative gradient difference to the first setting %.2e" % (st, float(e), worst))
    for _, gr, _ in graphs:
        for _ in range(20):
            gr.replay()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        line = []
        for st, gr, _ in graphs:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                gr.replay()
            torch.cuda.synchronize()
            line.append("%s %.4f ms" % (",".join(st.values()), (time.perf_counter() - t0) / a.iters * 1e3))
        print("round %d: value + gradient  %s" % (r, "   ".join(line)))


if __name__ == "__main__":
    main()
