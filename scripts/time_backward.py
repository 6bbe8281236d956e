"""Wall time of one IW-ELBO value + gradient evaluation (backward.iw_elbo_and_gradients) and of one training step
(training.Trainer.step = NatGrad op + Adam op, two gradient evaluations) at a BASELINE.json configuration.
Usage: python scripts/time_backward.py [--config 2] [--iters 20]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import synthetic, backward   # noqa: E402
from dgps_with_iwvi_amd.training import Trainer   # noqa: E402

CONFIGS = {1: dict(L=2, M=128, K=5, B=1024, with_lv=False), 2: dict(L=2, M=128, K=20, B=1024, with_lv=True),
           3: dict(L=3, M=256, K=50, B=4096, with_lv=False), 4: dict(L=5, M=512, K=100, B=8192, with_lv=False)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only-gradient", action="store_true", help="time only the value+gradient evaluation (for kernel profiles)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = synthetic.make_spec(**CONFIGS[a.config], seed=0)
    model = synthetic.build_model(spec, dev)

    def timed(fn, n):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    T = spec["B"] * spec["K"]
    if a.only_gradient:
        grad = timed(lambda: backward.iw_elbo_and_gradients(model), a.iters)
        print("config %d: value+gradient %.3f ms (%d evaluations incl. 2 warm-up)" % (a.config, grad, a.iters + 2))
        return
    fwd = timed(lambda: model._build_likelihood(), a.iters)
    grad = timed(lambda: backward.iw_elbo_and_gradients(model), a.iters)
    graph_ms = float("nan")
    try:                                                      # the same evaluation replayed from a hipGraph (no host work per launch)
        if T * spec["layers"][-1]["Z"].shape[0] > 2e8:
            raise RuntimeError("skipped at this size (a captured evaluation pins its workspaces)")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                backward.iw_elbo_and_gradients(model)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = backward.iw_elbo_and_gradients(model)
        graph_ms = timed(g.replay, a.iters)
        e1 = float(out[0]); g.replay(); torch.cuda.synchronize(); e2 = float(out[0])
        print("graph replay: %.3f ms per value+gradient; two replays give different noise: %s (%.4f, %.4f)" % (graph_ms, e1 != e2, e1, e2))
    except Exception as e:
        print("graph capture failed: %s: %s" % (type(e).__name__, e))
    tr = Trainer(model)
    step = timed(lambda: tr.step(), a.iters)
    step_g = float("nan")
    try:
        trg = Trainer(model, use_graph=True)
        step_g = timed(lambda: trg.step(), a.iters)
    except Exception as e:
        print("graph-mode trainer failed: %s: %s" % (type(e).__name__, e))
    print("config %d: forward (fused, eager) %.3f ms | value+gradient %.3f ms (%.2e samples/s) | training step %.3f ms eager, %.3f ms as hipGraph replays (one per step: the data is not minibatched)"
          % (a.config, fwd, grad, T / grad * 1e3, step, step_g))


if __name__ == "__main__":
    main()
