"""The evaluation loop of experiments/run_conditional_density_estimation.py:128-169, batched on the GPU.

The reference walks the test set one point at a time: ``predict_y_samples(x, num_predict_samples)``, a Gaussian
``sklearn.neighbors.KernelDensity`` with Silverman's bandwidth fitted to the samples, its log density at y, the squared
error of the sample mean (and a Shapiro-Wilk statistic of the samples, a diagnostic).  Here ``predict_y_samples`` runs on
batches of test points (the layer kernels; prior-mode latent-variable layers, layers.py:73-81) and
``iwvi_kde_loglik`` evaluates every point's estimate in one launch."""
import numpy as np
import torch

from . import _abi, settings


def kde_log_density(samples, y):
    """samples [S, N] (device), y [N] -> (logp [N], sqerr [N], mean_std [N, 2]) through ``iwvi_kde_loglik``."""
    samples = _abi.dev_tensor(samples.contiguous(), "samples")
    S, N = samples.shape
    y = _abi.dev_tensor(y.reshape(-1).contiguous(), "y")
    if y.numel() != N:
        raise ValueError("y has %d entries, samples cover %d points" % (y.numel(), N))
    dev = samples.device
    logp, sq = (torch.empty(N, dtype=settings.float_type, device=dev) for _ in range(2))
    ms = torch.empty(N, 2, dtype=settings.float_type, device=dev)
    _abi.check(_abi.lib().iwvi_kde_loglik(_abi.ptr(samples), N, 1, _abi.ptr(y), N, S, _abi.ptr(logp), _abi.ptr(sq),
                                         _abi.ptr(ms), _abi.stream_ptr()))
    return logp, sq, ms


def evaluate(model, X_test, Y_test, num_predict_samples=2000, predict_batch_size=1000, shapiro=False):
    """-> dict(test_loglik, test_rmse[, test_shapiro_W_median]) as the reference's ``res`` (:167-169); Y one column."""
    dev = model.X.device
    X_test = torch.as_tensor(np.asarray(X_test, dtype=np.float32), device=dev) if not isinstance(X_test, torch.Tensor) else X_test.to(dev, settings.float_type)
    Y_test = torch.as_tensor(np.asarray(Y_test, dtype=np.float32), device=dev) if not isinstance(Y_test, torch.Tensor) else Y_test.to(dev, settings.float_type)
    if Y_test.dim() == 2 and Y_test.shape[1] != 1:
        raise ValueError("the reference's evaluation is for one output column")
    N = X_test.shape[0]
    if N == 0 or Y_test.shape[0] != N:
        raise ValueError("X_test has %d rows, Y_test %d" % (N, Y_test.shape[0]))
    logps, sqs, Ws = [], [], []
    for lo in range(0, N, predict_batch_size):
        x, y = X_test[lo:lo + predict_batch_size], Y_test[lo:lo + predict_batch_size]
        smp = model.predict_y_samples(x, num_predict_samples)[:, :, 0]              # [S, n]  (:154-156)
        lp, sq, ms = kde_log_density(smp, y)
        logps.append(lp)
        sqs.append(sq)
        if shapiro:                                                                # diagnostic only; host side like the reference (:164)
            from scipy.stats import shapiro as _shapiro
            z = ((smp - ms[:, 0]) / ms[:, 1]).cpu().numpy()
            Ws += [float(_shapiro(z[:, i])[0]) for i in range(z.shape[1])]
    res = {"test_loglik": float(torch.cat(logps).double().mean()), "test_rmse": float(torch.cat(sqs).double().mean()) ** 0.5}
    if shapiro:
        res["test_shapiro_W_median"] = float(np.median(Ws))
    return res
