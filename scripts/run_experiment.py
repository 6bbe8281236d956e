"""The reference's experiment script (experiments/run_conditional_density_estimation.py) on this framework, end to end
on a synthetic conditional-density problem (no dataset files travel to the GPU box): build the model from a
configuration string, train with the reference's train_op (NatGrad + Adam), evaluate the test log-likelihood by KDE.

    python scripts/run_experiment.py --configuration L1_G5 --mode IWAE --iterations 500
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dgps_with_iwvi_amd import build_models, evaluation   # noqa: E402


def bimodal_data(n, rng):
    """y | x is a two-component mixture whose separation grows with x: a GP with Gaussian noise cannot fit it, a
    latent-variable layer can (the point of the reference's method)."""
    x = rng.uniform(-2, 2, (n, 1))
    side = rng.integers(0, 2, (n, 1)) * 2 - 1
    y = np.sin(2 * x) + side * (0.2 + 0.6 * (x + 2) / 4) + 0.05 * rng.standard_normal((n, 1))
    return x, y


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--mode", default="IWAE")
    p.add_argument("--configuration", default="L1_G5")
    p.add_argument("--M", type=int, default=64)
    p.add_argument("--num_IW_samples", type=int, default=5)
    p.add_argument("--minibatch_size", type=int, default=256)
    p.add_argument("--iterations", type=int, default=500)
    p.add_argument("--likelihood_variance", type=float, default=1e-2)
    p.add_argument("--gamma", type=float, default=1e-2)
    p.add_argument("--gamma_decay", type=float, default=0.98)
    p.add_argument("--lr", type=float, default=5e-3)
    p.add_argument("--lr_decay", type=float, default=0.98)
    p.add_argument("--fix_linear", type=int, default=1)
    p.add_argument("--num_predict_samples", type=int, default=2000)
    p.add_argument("--predict_batch_size", type=int, default=1000)
    p.add_argument("--n_train", type=int, default=2000)
    p.add_argument("--n_test", type=int, default=500)
    p.add_argument("--seed", type=int, default=0)
    ARGS = p.parse_args(argv)
    ARGS.fix_linear = bool(ARGS.fix_linear)
    rng = np.random.default_rng(ARGS.seed)
    np.random.seed(ARGS.seed)
    X, Y = bimodal_data(ARGS.n_train, rng)
    Xs, Ys = bimodal_data(ARGS.n_test, rng)
    mu, sd = Y.mean(), Y.std()
    Y, Ys = (Y - mu) / sd, (Ys - mu) / sd
    dev = torch.device("cuda:0")
    model = build_models.build_model(ARGS, X.astype(np.float32), Y.astype(np.float32), device=dev)
    before = evaluation.evaluate(model, Xs, Ys, ARGS.num_predict_samples, ARGS.predict_batch_size)
    t0 = time.perf_counter()
    for it in range(ARGS.iterations):
        elbo = model.train_op()
        if it % max(1, ARGS.iterations // 10) == 0:
            print("iteration %5d  ELBO %.2f" % (it, float(elbo)), flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = evaluation.evaluate(model, Xs, Ys, ARGS.num_predict_samples, ARGS.predict_batch_size, shapiro=True)
    res.update(test_loglik_before_training=before["test_loglik"], train_seconds=dt, ms_per_iteration=dt / max(ARGS.iterations, 1) * 1e3)
    res.update(ARGS.__dict__)
    print(res)
    return res


if __name__ == "__main__":
    main()
