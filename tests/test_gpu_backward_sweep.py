"""Seeded random sweep of small model shapes through the backward pass against the float64
gradient oracle: ragged sample counts, M not a multiple of 16 or 64, 1-3 GP layers, with and without a latent-variable
layer, IW and VI bounds.  D >= 4 (the family the stated float32 tolerance is for, see test_gpu_random_sweep.py).
Tolerance: max-norm relative 5e-3 per gradient array, ELBO relative 2e-4."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _cases(n=14, seed=77):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        out.append(dict(L=int(rng.integers(1, 4)), M=int(rng.choice([9, 16, 31, 48, 64, 100, 128])), B=int(rng.integers(2, 20)),
                        K=int(rng.integers(1, 9)), Dx=int(rng.integers(4, 10)), R=int(rng.integers(1, 6)),
                        with_lv=bool(rng.integers(0, 2)), seed=500 + i, vi=bool(i % 5 == 4)))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: "L%(L)d_M%(M)d_B%(B)d_K%(K)d_D%(Dx)d_R%(R)d_lv%(with_lv)d_vi%(vi)d" % c)
def test_random_shape_gradients_match_oracle(gpu_device, case):
    from dgps_with_iwvi_amd import synthetic, backward
    from dgps_with_iwvi_amd.models import DGP_VI
    from oracle.grad_oracle import iw_elbo_and_gradients
    case = dict(case)
    vi = case.pop("vi")
    spec = synthetic.make_spec(**case)
    zs = synthetic.make_noise(spec, seed=case["seed"] + 1)
    val, ref = iw_elbo_and_gradients(spec, zs, mode_vi=vi)
    B, K = spec["B"], spec["K"]
    if vi:
        model = synthetic.build_model(spec, gpu_device, cls=DGP_VI, num_samples=K)
        zd = [torch.as_tensor(np.asarray(z, dtype=np.float32).transpose(1, 0, 2).reshape(K * B, -1).copy(), device=gpu_device) for z in zs]
    else:
        model = synthetic.build_model(spec, gpu_device)
        zd = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in zs]
    elbo, grads = backward.iw_elbo_and_gradients(model, zd)
    assert abs(float(elbo) - val) <= 2e-4 * abs(val) + 2e-3 * B, (case, float(elbo), val)
    assert sorted(grads) == sorted(ref)
    for k, v in grads.items():
        got, want = v.cpu().numpy().astype(np.float64).reshape(ref[k].shape), ref[k]
        scale = max(np.abs(want).max(), 1e-9)
        assert np.abs(got - want).max() <= 5e-3 * scale, (case, k, np.abs(got - want).max(), scale)
