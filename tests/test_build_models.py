"""Model factory (dgps_with_iwvi_amd/build_models.py; reference experiments/build_models.py:176-268, row F4)."""
import types

import numpy as np
import pytest
import torch


def _args(**kw):
    base = dict(mode="IWAE", configuration="L1_G5", M=16, likelihood_variance=0.01, minibatch_size=32, num_IW_samples=4)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _data(n=200, d=6, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, d))
    Y = np.sin(X.sum(1, keepdims=True)) + 0.1 * rng.standard_normal((n, 1))
    return X, Y


def test_parse_configuration():
    from dgps_with_iwvi_amd.build_models import parse_configuration
    assert parse_configuration("") == []
    assert parse_configuration("L1_G5_G12") == [("L", 1), ("G", 5), ("G", 12)]
    with pytest.raises(ValueError):
        parse_configuration("X3")


def test_layer_stack_follows_the_reference_recipe():
    from dgps_with_iwvi_amd.build_models import build_layers
    from dgps_with_iwvi_amd.layers import GPLayer, LatentVariableLayer
    from dgps_with_iwvi_amd.temp_workaround import SharedMixedMok
    X, _ = _data()
    np.random.seed(3)
    layers = build_layers("L1_G5_G5", X, 16)
    assert [type(l) for l in layers] == [LatentVariableLayer, GPLayer, GPLayer, GPLayer]
    lv, g1, g2, last = layers
    assert lv.latent_dim == 1 and lv.encoder.layer_dims[0] == X.shape[1] + 1
    D = X.shape[1]
    # first GP layer sees [x, w]: D + 1 inputs, 5 latent GPs mixed back to D + 1 ... the reference keeps D_out = D
    assert g1._Z().shape == (16, D + 1) and g1.num_outputs == 5 and isinstance(g1.kern, SharedMixedMok)
    W = g1.kern.W.cpu().numpy()
    P = np.linalg.svd(X, full_matrices=False)[2]
    assert W.shape == (D, 5) and np.allclose(W, P[:, :5], atol=1e-6)
    A = g1.mean_function.A.cpu().numpy()
    assert A.shape == (D + 1, D) and np.allclose(A[:D], np.eye(D)) and np.allclose(A[D], 0)
    assert np.allclose(g1._base_kern().lengthscales.cpu().numpy(), (D + 1) ** 0.5)
    # inner layers start with q_sqrt = 1e-5 I, the final layer with I; q_mu = 0
    assert np.allclose(g1.q_sqrt[0].cpu().numpy(), 1e-5 * np.eye(16)) and np.allclose(last.q_sqrt[0].cpu().numpy(), np.eye(16))
    assert float(g2.q_mu.abs().sum()) == 0.0
    assert g2._Z().shape == (16, D) and last._Z().shape == (16, D) and last.num_outputs == 1
    # deeper inducing inputs share their first columns with the first layer's (k-means centres of X)
    assert np.allclose(g2._Z().cpu().numpy()[:, :D], g1._Z().cpu().numpy()[:, :D])


def test_unsupported_modes_raise():
    from dgps_with_iwvi_amd.build_models import build_model
    X, Y = _data()
    for mode in ("CVAE", "SGHMC"):
        with pytest.raises(NotImplementedError):
            build_model(_args(mode=mode), X, Y, device=torch.device("cpu"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,config", [("IWAE", "L1_G5"), ("VI", "G5_G5"), ("IWAE", "")])
def test_factory_model_runs_and_round_trips_a_checkpoint(gpu_device, tmp_path, mode, config):
    from dgps_with_iwvi_amd import build_models as bm
    X, Y = _data()
    np.random.seed(1)
    model = bm.build_model(_args(mode=mode, configuration=config), X.astype(np.float32), Y.astype(np.float32), device=gpu_device)
    rng = np.random.default_rng(2)
    for layer in model.layers:                                     # move away from the initial state
        if hasattr(layer, "q_mu"):
            layer.q_mu = layer.q_mu + torch.as_tensor(rng.standard_normal(tuple(layer.q_mu.shape)), dtype=torch.float32, device=gpu_device)
    a = model.compute_log_likelihood()
    assert np.isfinite(a)
    path = str(tmp_path / "ckpt.npz")
    bm.save_checkpoint(model, path)
    np.random.seed(1)
    fresh = bm.build_model(_args(mode=mode, configuration=config), X.astype(np.float32), Y.astype(np.float32), device=gpu_device)
    bm.load_checkpoint(fresh, path)
    for l0, l1 in zip(model.layers, fresh.layers):
        if hasattr(l0, "q_mu"):
            assert torch.equal(l0.q_mu, l1.q_mu) and torch.equal(l0._Z(), l1._Z())
