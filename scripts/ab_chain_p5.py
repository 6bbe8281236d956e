"""A/B of the chain kernel's products over samples (phase 5): split-f16 operands (default) against the fp32 form (development switch
IWVI_BW_P5_F32), in ONE process, the two routes' graphs replayed alternately; plus the difference between the two routes' gradients
on the same injected noise.  Usage: python scripts/ab_chain_p5.py [--config 2] [--rounds 6] [--iters 50]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import _abi, backward, synthetic   # noqa: E402
from time_backward import CONFIGS   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = synthetic.make_spec(**CONFIGS[a.config], seed=0)
    model = synthetic.build_model(spec, dev)
    B, K = spec["B"], spec["K"]
    gen = torch.Generator(device="cpu").manual_seed(7)
    zs = []
    for l in model.layers:
        dim = getattr(l, "latent_dim", None) or l.num_outputs
        zs.append(torch.randn(B, K, dim, generator=gen).to(dev))
    grads = {}
    for name, val in (("f16", 0), ("f32", 1)):
        _abi.set_debug_option("IWVI_BW_P5_F32", val)
        e, g = backward.iw_elbo_and_gradients(model, zs=zs)
        torch.cuda.synchronize()
        grads[name] = (float(e), {k: v.double().cpu() for k, v in g.items()})
    print("bound: f16 %.6f  f32 %.6f" % (grads["f16"][0], grads["f32"][0]))
    for k in sorted(grads["f16"][1]):
        x, y = grads["f16"][1][k], grads["f32"][1][k]
        print("  %-12s max|d| %.3e   max|g| %.3e   rel %.2e" % (k, float((x - y).abs().max()), float(y.abs().max()),
                                                               float((x - y).abs().max() / y.abs().max().clamp_min(1e-30))))
    graphs = {}
    for name, val in (("f16", 0), ("f32", 1)):
        _abi.set_debug_option("IWVI_BW_P5_F32", val)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                backward.iw_elbo_and_gradients(model)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = backward.iw_elbo_and_gradients(model)
        graphs[name] = (g, out)
    _abi.set_debug_option("IWVI_BW_P5_F32", 0)
    for g, _ in graphs.values():
        for _ in range(20):
            g.replay()
    torch.cuda.synchronize()
    for r in range(a.rounds):
        line = []
        for name in ("f16", "f32"):
            g = graphs[name][0]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.iters):
                g.replay()
            torch.cuda.synchronize()
            line.append("%s %.4f ms" % (name, (time.perf_counter() - t0) / a.iters * 1e3))
        print("round %d: value + gradient  %s" % (r, "   ".join(line)))


if __name__ == "__main__":
    main()
