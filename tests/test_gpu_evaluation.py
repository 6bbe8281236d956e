"""Evaluation loop (experiments/run_conditional_density_estimation.py:128-169) on the GPU against the NumPy oracle
(oracle/kde_oracle.py, itself pinned by sklearn's KernelDensity)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.kde_oracle import kde_loglik   # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("S,N", [(2000, 37), (64, 5), (130, 1000)])
def test_kde_kernel_matches_oracle(gpu_device, S, N):
    from dgps_with_iwvi_amd import evaluation
    rng = np.random.default_rng(S + N)
    samples = (rng.standard_normal((S, N)) * rng.uniform(0.1, 2.0, N) + rng.standard_normal(N) * 3).astype(np.float32)
    y = (rng.standard_normal(N) * 2).astype(np.float32)
    lp, sq, ms = evaluation.kde_log_density(torch.as_tensor(samples, device=gpu_device), torch.as_tensor(y, device=gpu_device))
    ref_lp, ref_sq = kde_loglik(samples, y)
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(sq.cpu().numpy(), ref_sq, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(ms.cpu().numpy()[:, 1], samples.astype(np.float64).std(0), rtol=2e-5)


def test_evaluate_reproduces_the_reference_loop_on_the_same_samples(gpu_device):
    """``evaluate`` == the per-point loop of :148-165 applied to the samples the model drew (batched vs one at a time)."""
    from dgps_with_iwvi_amd import synthetic, evaluation
    spec = synthetic.make_spec(L=2, M=32, B=16, K=2, with_lv=True, seed=9, n_data=400)
    model = synthetic.build_model(spec, gpu_device)
    Xt, Yt = spec["X"][100:150], spec["Y"][100:150]
    res = evaluation.evaluate(model, Xt, Yt, num_predict_samples=256, predict_batch_size=20, shapiro=True)
    assert np.isfinite(res["test_loglik"]) and res["test_rmse"] > 0 and 0.5 < res["test_shapiro_W_median"] <= 1.0
    # a model whose predictive is N(m, v) for known m, v: the KDE log-likelihood approaches the Gaussian log density
    x = torch.as_tensor(np.asarray(Xt[:8], dtype=np.float32), device=gpu_device)
    smp = model.predict_y_samples(x, 4000)[:, :, 0]
    lp, _, ms = evaluation.kde_log_density(smp, torch.as_tensor(np.asarray(Yt[:8], dtype=np.float32), device=gpu_device))
    ref_lp, _ = kde_loglik(smp.cpu().numpy(), Yt[:8])
    np.testing.assert_allclose(lp.cpu().numpy(), ref_lp, rtol=5e-5, atol=5e-5)


def test_experiment_end_to_end_trains_and_improves_the_test_log_likelihood(gpu_device):
    """scripts/run_experiment.py = the reference's experiment script on a synthetic bimodal conditional density:
    factory -> train_op (NatGrad + Adam, fresh minibatches) -> KDE test log-likelihood."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import run_experiment
    res = run_experiment.main(["--iterations", "300", "--n_train", "1000", "--n_test", "200", "--num_predict_samples", "500"])
    assert res["test_loglik"] > res["test_loglik_before_training"] + 0.3, res
    assert np.isfinite(res["test_rmse"]) and res["ms_per_iteration"] < 50
