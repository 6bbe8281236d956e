"""The gradient oracle (oracle/grad_oracle.py, target of SURVEY.md section 8 row F1) against central finite
differences of the independent NumPy restatement of the forward pass, along random directions."""
import copy
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dgps_with_iwvi_amd import synthetic                    # noqa: E402
from oracle.from_spec import build_oracle, oracle_noise     # noqa: E402
from oracle.grad_oracle import iw_elbo_and_gradients        # noqa: E402


def _numpy_elbo(spec, zs):
    return float(build_oracle(spec).build_likelihood(oracle_noise(spec, zs)))


def _perturbed(spec, name, direction, eps):
    s = copy.deepcopy(spec)
    if name == "lik_var":
        s["lik_var"] = float(s["lik_var"] + eps * direction)
        return s
    li, key = name.split(".")
    L = s["layers"][int(li[1:])]
    if key.startswith("encW"):
        L["enc_W"][int(key[4:])] = L["enc_W"][int(key[4:])] + eps * direction
    elif key.startswith("encb"):
        L["enc_b"][int(key[4:])] = L["enc_b"][int(key[4:])] + eps * direction
    elif key == "var":
        L["var"] = float(L["var"] + eps * direction)
    elif key == "mfA":
        L["mf"] = (L["mf"][0], L["mf"][1] + eps * direction) + tuple(L["mf"][2:])
    else:
        L[key] = L[key] + eps * direction
    return s


@pytest.mark.parametrize("case", [dict(L=2, M=8, B=4, K=3, Dx=2, R=2, with_lv=True, seed=21),
                                  dict(L=2, M=12, B=5, K=4, Dx=3, R=2, with_lv=False, seed=22)])
def test_autodiff_gradients_match_finite_differences(case):
    spec = synthetic.make_spec(parity=True, n_data=64, **case)
    spec = {k: (np.asarray(v, dtype=np.float64) if isinstance(v, np.ndarray) else v) for k, v in spec.items()}
    for L in spec["layers"]:
        for k, v in list(L.items()):
            if isinstance(v, np.ndarray):
                L[k] = v.astype(np.float64)
            elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
                L[k] = [a.astype(np.float64) for a in v]
            elif isinstance(v, tuple):
                L[k] = tuple(a.astype(np.float64) if isinstance(a, np.ndarray) else a for a in v)
    zs = [z.astype(np.float64) for z in synthetic.make_noise(spec, seed=3)]
    val, grads = iw_elbo_and_gradients(spec, zs)
    assert abs(val - _numpy_elbo(spec, zs)) <= 1e-9 * abs(val)
    rng = np.random.default_rng(5)
    for name, g in grads.items():
        d = rng.standard_normal(g.shape)
        if name.endswith("q_sqrt"):
            d = np.tril(d)
        d = d / max(np.linalg.norm(d), 1e-30)
        eps = 1e-6
        fd = (_numpy_elbo(_perturbed(spec, name, d, eps), zs) - _numpy_elbo(_perturbed(spec, name, d, -eps), zs)) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) <= 2e-5 * max(1.0, abs(an), abs(fd)), (name, fd, an)


@pytest.mark.parametrize("name", ["tiny_L2_lv", "mid_L2_lv"])
def test_gradient_fixture_is_reproduced(name):
    """tests/golden/grad_<name>.npz (made by tests/golden/make_golden.py) == the gradient oracle on the stored inputs."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from make_golden import arrays_to_spec
    here = os.path.join(ROOT, "tests", "golden")
    a = dict(np.load(os.path.join(here, name + ".npz"), allow_pickle=True))
    g = dict(np.load(os.path.join(here, "grad_" + name + ".npz")))
    spec = arrays_to_spec(a)
    zs = [a["z%d" % i] for i in range(len(spec["layers"]))]
    val, grads = iw_elbo_and_gradients(spec, zs)
    assert abs(val - float(g["elbo"])) <= 1e-10 * abs(val)
    for k, v in grads.items():
        ref = g[k.replace(".", "_")]
        assert np.allclose(v, ref, rtol=1e-8, atol=1e-10 * max(1.0, float(np.abs(ref).max()))), k


def test_vi_mode_of_the_torch_restatement_equals_the_numpy_oracle():
    """mode_vi (models.py:49-86: analytic local KL, mean over S) of the differentiable restatement against the
    independent NumPy oracle's DGP_VI; noise [B, S, .] here, [S*B, .] (S-major tiling, models.py:50) there."""
    from oracle.ref_torch_cpu import CpuDGP
    import torch
    spec = synthetic.make_spec(L=2, M=16, B=7, K=4, Dx=3, R=2, with_lv=True, seed=5, n_data=500)
    zs = synthetic.make_noise(spec, seed=6)
    val = float(CpuDGP(spec, torch.float64).elbo_tensor(zs, mode_vi=True))
    zs_sn = [np.asarray(z).transpose(1, 0, 2).reshape(-1, z.shape[-1]) for z in zs]
    ref = build_oracle(spec, iw=False).build_likelihood(zs_sn)
    assert abs(val - ref) <= 1e-9 * abs(ref), (val, ref)


def test_matern52_of_the_torch_restatement_equals_the_numpy_oracle():
    import torch
    from oracle import iwvi_oracle as O
    from oracle.ref_torch_cpu import _matern52
    rng = np.random.default_rng(3)
    X, X2, ls = rng.standard_normal((7, 3)), rng.standard_normal((5, 3)), np.array([0.7, 1.3, 2.0])
    k = O.Matern52(3, variance=1.7, lengthscales=ls)
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)
    np.testing.assert_allclose(_matern52(t(X), t(X2), t(ls), 1.7).numpy(), k.K(X, X2), rtol=1e-12)
    np.testing.assert_allclose(_matern52(t(X), None, t(ls), 1.7).numpy(), k.K(X), rtol=1e-12, atol=1e-12)
