"""Backward pass against reverse-mode autodiff of the float64 restatement
(oracle/ref_torch_cpu.py, oracle/grad_oracle.py) -- the way the reference itself gets gradients
(TensorFlow autodiff, experiments/build_models.py:284-304).  Tolerance: float32 kernels vs float64 oracle,
max-norm relative 3e-3 per gradient array (stated per assert)."""
import math
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_torch_cpu import CpuDGP   # noqa: E402

pytestmark = pytest.mark.gpu


def _close(name, got, ref, rtol=3e-3):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max()
    assert err <= rtol * scale, "%s: max err %.3e vs scale %.3e" % (name, err, scale)


def _layer_reference(spec, li, F, z, cs, cm, cv, klw, kern=None):
    """d/d{F, Z, ls, var, q_mu, q_sqrt} of sum(cs*sample + cm*mean + cv*var) - klw*KL for GP layer li (float64 autograd)."""
    m = CpuDGP(spec, torch.float64)
    L = m.layers[li]
    leaf = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float64)).clone().requires_grad_(True)
    P = {k: leaf(L[k].detach().numpy()) for k in ("Z", "ls", "q_mu")}
    P["q_sqrt"] = leaf(L["q_sqrt"].detach().numpy())
    P["var"] = leaf(L["var"])
    Ft = leaf(F)
    Ld = dict(L, Z=P["Z"], ls=P["ls"], q_mu=P["q_mu"], q_sqrt=torch.tril(P["q_sqrt"]), var=P["var"], kern=kern)
    s, mu, v = m._conditional(Ld, Ft[None], False, torch.as_tensor(z, dtype=torch.float64)[None])
    if L["W"] is not None:
        s, mu, v = s @ L["W"].T, mu @ L["W"].T, v @ (L["W"] ** 2).T
    if L["A"] is not None:
        s, mu = s + Ft[None] @ L["A"], mu + Ft[None] @ L["A"]
    M, R = L["q_mu"].shape
    Lq = Ld["q_sqrt"]
    kl = 0.5 * ((P["q_mu"] ** 2).sum() - M * R - torch.log(torch.diagonal(Lq, dim1=-2, dim2=-1) ** 2).sum() + (Lq ** 2).sum())
    t = lambda a: torch.as_tensor(a, dtype=torch.float64)[None]
    obj = (t(cs) * s).sum() + (t(cm) * mu).sum() + (t(cv) * v).sum() - klw * kl
    obj.backward()
    g = {k: p.grad.numpy() for k, p in P.items()}
    g["F"] = Ft.grad.numpy()
    return g


@pytest.mark.parametrize("M,T,li", [(32, 70, 0), (128, 300, 0), (128, 300, 1), (40, 129, 0), (256, 520, 1)])
def test_gp_layer_backward_matches_autodiff(gpu_device, M, T, li):
    from dgps_with_iwvi_amd import synthetic, backward
    spec = synthetic.make_spec(L=2, M=M, B=8, K=2, with_lv=False, seed=M + li)
    model = synthetic.build_model(spec, gpu_device)
    layer = model.layers[li]
    rng = np.random.default_rng(T)
    D, R = layer._Z().shape[1], layer.num_outputs
    P = spec["layers"][li]["W"].shape[0] if spec["layers"][li]["W"] is not None else R
    F = rng.standard_normal((T, D)).astype(np.float32)
    z = rng.standard_normal((T, R)).astype(np.float32)
    cs, cm, cv = (rng.standard_normal((T, P)).astype(np.float32) for _ in range(3))
    dev = gpu_device
    tt = lambda a: torch.as_tensor(a, device=dev)
    saved = backward.gp_forward_saved(layer, tt(F), tt(z))
    np.testing.assert_array_equal(saved.noise.cpu().numpy(), z)
    out = backward.gp_backward(layer, saved, tt(cs), tt(cm), tt(cv), kl_weight=0.7)
    ref = _layer_reference(spec, li, F, z, cs, cm, cv, 0.7)
    _close("dF", out["dF"].cpu(), ref["F"])
    _close("dq_mu", out["dq_mu"].cpu(), ref["q_mu"])
    _close("dq_sqrt", out["dq_sqrt"].cpu(), np.tril(ref["q_sqrt"]))
    _close("dZ", out["dZ"].cpu(), ref["Z"])
    _close("dls", out["dls"].cpu(), ref["ls"])
    _close("dvariance", out["dvariance"].cpu()[0], ref["var"])
    # deterministic: the split-K partials are summed in a fixed order
    out2 = backward.gp_backward(layer, saved, tt(cs), tt(cm), tt(cv), kl_weight=0.7)
    for k in out:
        assert torch.equal(out[k], out2[k]), k


@pytest.mark.parametrize("Dx,M,T,li", [(3, 32, 70, 0), (2, 48, 129, 1)])
def test_gp_layer_backward_on_a_float64_route_layer(gpu_device, Dx, M, T, li):
    """A layer of input dimension <= 3 takes the float64 stage-1 route by the default rule: ``precompute_dense`` must keep
    IWVI_GP_F64_STAGE1 beside IWVI_GP_WANT_DENSE (the forward reads the state's plain z~ image, which only that bit writes), and a
    layer descriptor with IWVI_LAYER_F64_STAGE1 on a state prepared without the bit is refused."""
    from dgps_with_iwvi_amd import synthetic, backward, _abi
    from dgps_with_iwvi_amd.temp_workaround import precompute_states
    spec = synthetic.make_spec(L=2, M=M, B=8, K=2, Dx=Dx, R=2, with_lv=False, seed=M + li)
    model = synthetic.build_model(spec, gpu_device)
    layer = model.layers[li]
    assert layer.uses_f64_stage1()
    rng = np.random.default_rng(T)
    D, R = layer._Z().shape[1], layer.num_outputs
    P = spec["layers"][li]["W"].shape[0] if spec["layers"][li]["W"] is not None else R
    F = rng.standard_normal((T, D)).astype(np.float32)
    z = rng.standard_normal((T, R)).astype(np.float32)
    cs, cm, cv = (rng.standard_normal((T, P)).astype(np.float32) for _ in range(3))
    tt = lambda a: torch.as_tensor(a, device=gpu_device)
    # poison the state's buffer first: a forward that read an unwritten z~ image would show it
    layer.state().buf.fill_(0xFF)
    saved = backward.gp_forward_saved(layer, tt(F), tt(z))
    assert layer.state().f64_prepared
    m = CpuDGP(spec, torch.float64)
    s, mu, v = m._conditional(dict(m.layers[li]), torch.as_tensor(F, dtype=torch.float64)[None], False, torch.as_tensor(z, dtype=torch.float64)[None])
    gmv = saved.GMV.cpu().numpy().astype(np.float64)          # [T, 3R]: latent sample | mean | variance
    np.testing.assert_allclose(gmv[:, R:2 * R], mu[0].numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(gmv[:, 2 * R:], v[0].numpy(), rtol=1e-5, atol=1e-5)
    out = backward.gp_backward(layer, saved, tt(cs), tt(cm), tt(cv), kl_weight=0.7)
    ref = _layer_reference(spec, li, F, z, cs, cm, cv, 0.7)
    # the adjoint chain itself is float32: its own tolerance, on a moderately conditioned K_uu
    for name, key in (("dF", "F"), ("dq_mu", "q_mu"), ("dZ", "Z"), ("dls", "ls")):
        _close(name, out[name].cpu(), ref[key], rtol=2e-2)
    # the refusal: the same state refilled WITHOUT the bit
    d = layer.state_desc()
    d.flags &= ~_abi.GP_F64_STAGE1
    precompute_states([d])
    assert not layer.state().f64_prepared
    with pytest.raises(ValueError, match="float64 stage-1 route"):
        layer.fused_desc(tt(z), None)


os.environ.setdefault("IWVI_BW_FUSED", "1")      # exercise the fused per-sample kernel wherever the shapes allow it (it is gated by size otherwise)


def _model_grads(gpu_device, spec, zs):
    from dgps_with_iwvi_amd import synthetic, backward
    model = synthetic.build_model(spec, gpu_device)
    elbo, grads = backward.iw_elbo_and_gradients(model, [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in zs])
    return float(elbo), {k: v.detach().cpu().numpy() for k, v in grads.items()}


@pytest.mark.parametrize("M,T,li", [(32, 70, 0), (128, 256, 1), (64, 16384, 0)])
def test_matern52_layer_backward_matches_autodiff(gpu_device, M, T, li):
    """The adjoint of a layer with a Matern-5/2 kernel (the forward supports it; dk/dd2 = -(5/6) s2 (1 + sqrt5 r) exp(-sqrt5 r))."""
    from dgps_with_iwvi_amd import synthetic, backward, kernels
    spec = synthetic.make_spec(L=2, M=M, B=8, K=2, with_lv=False, seed=3 * M + li)
    model = synthetic.build_model(spec, gpu_device)
    layer = model.layers[li]
    old = layer._base_kern()
    new = kernels.Matern52(old.input_dim, variance=old.variance, lengthscales=old.lengthscales, ARD=True).to(gpu_device)
    if hasattr(layer.kern, "kernel"):
        layer.kern.kernel = new
    else:
        layer.kern = new
    rng = np.random.default_rng(T)
    D, R = layer._Z().shape[1], layer.num_outputs
    P = spec["layers"][li]["W"].shape[0] if spec["layers"][li]["W"] is not None else R
    F = rng.standard_normal((T, D)).astype(np.float32)
    z = rng.standard_normal((T, R)).astype(np.float32)
    cs, cm, cv = (rng.standard_normal((T, P)).astype(np.float32) for _ in range(3))
    tt = lambda a: torch.as_tensor(a, device=gpu_device)
    saved = backward.gp_forward_saved(layer, tt(F), tt(z))
    out = backward.gp_backward(layer, saved, tt(cs), tt(cm), tt(cv), kl_weight=1.0)
    Tr = min(T, 512)                                             # the float64 reference on a prefix (the rest has zero cotangent)
    if Tr < T:
        cs[Tr:], cm[Tr:], cv[Tr:] = 0, 0, 0
        out = backward.gp_backward(layer, saved, tt(cs), tt(cm), tt(cv), kl_weight=1.0)
    ref = _layer_reference(spec, li, F[:Tr], z[:Tr], cs[:Tr], cm[:Tr], cv[:Tr], 1.0, kern="matern52")
    _close("dF", out["dF"].cpu()[:Tr], ref["F"])
    for k_out, k_ref in (("dq_mu", "q_mu"), ("dZ", "Z"), ("dls", "ls")):
        _close(k_out, out[k_out].cpu(), ref[k_ref])
    _close("dq_sqrt", out["dq_sqrt"].cpu(), np.tril(ref["q_sqrt"]))
    _close("dvariance", out["dvariance"].cpu()[0], ref["var"])


@pytest.mark.parametrize("name", ["tiny_L2_lv", "mid_L2_lv"])
def test_iw_elbo_gradients_match_golden(gpu_device, name):
    """tests/golden/grad_*.npz: d ELBO / d parameters from the gradient oracle (float64 autodiff, pinned by finite differences)."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    from test_golden import load
    spec, zs, _ = load(os.path.join(here, "golden", name + ".npz"))
    with np.load(os.path.join(here, "golden", "grad_" + name + ".npz")) as f:
        ref = {k: f[k] for k in f.files}
    elbo, grads = _model_grads(gpu_device, spec, zs)
    assert abs(elbo - float(ref["elbo"])) <= 2e-4 * abs(float(ref["elbo"])), (elbo, float(ref["elbo"]))
    assert sorted(k.replace(".", "_") for k in grads) == sorted(k for k in ref if k != "elbo")
    for k, v in grads.items():
        r = ref[k.replace(".", "_")]
        _close(k, v.reshape(r.shape), r, rtol=5e-3)


@pytest.mark.parametrize("L,M,K,B,lv", [(2, 64, 5, 16, True), (3, 32, 4, 12, True), (2, 128, 3, 40, False), (1, 48, 6, 10, False),
                                        # T = B*K a multiple of 64 and M in {64, 128, 256}: the fused per-sample kernel (k_bw_mid)
                                        (2, 64, 4, 16, True), (2, 128, 8, 16, True), (3, 128, 4, 32, False), (2, 256, 8, 8, True),
                                        # ragged: M not a multiple of 16 (padded states), odd B and K
                                        (2, 40, 3, 7, True), (3, 50, 5, 3, False)])
def test_iw_elbo_gradients_match_oracle(gpu_device, L, M, K, B, lv):
    from dgps_with_iwvi_amd import synthetic
    from oracle.grad_oracle import iw_elbo_and_gradients
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, with_lv=lv, seed=7 * L + K)
    zs = synthetic.make_noise(spec, seed=2)
    val, ref = iw_elbo_and_gradients(spec, zs)
    elbo, grads = _model_grads(gpu_device, spec, zs)
    assert abs(elbo - val) <= 2e-4 * abs(val), (elbo, val)
    assert sorted(grads) == sorted(ref)
    for k, v in grads.items():
        _close(k, v.reshape(ref[k].shape), ref[k], rtol=5e-3)
    # bit-reproducible, with and without the side-stream overlap of the parameter-gradient branches
    from dgps_with_iwvi_amd import backward
    model = synthetic.build_model(spec, gpu_device)
    zd = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in zs]
    runs = [backward.iw_elbo_and_gradients(model, zd, overlap=o)[1] for o in (True, True, False)]
    torch.cuda.synchronize()
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
    # wrt="final_q" (what the natural-gradient op asks for): the same bound and the same two entries, bit for bit
    e_q, g_q = backward.iw_elbo_and_gradients(model, zd, wrt="final_q")
    assert sorted(g_q) == ["l%d.q_mu" % (L - 1 + int(lv)), "l%d.q_sqrt" % (L - 1 + int(lv))]
    assert float(e_q) == float(backward.iw_elbo_and_gradients(model, zd)[0])
    for k in g_q:
        assert torch.equal(g_q[k], runs[0][k]), k


@pytest.mark.parametrize("cfg", [1, 2, 3, dict(L=2, M=512, K=16, B=64, with_lv=True), dict(L=3, M=200, K=7, B=9, with_lv=False)])
def test_final_q_gradients_at_full_size_equal_the_full_adjoints(gpu_device, cfg):
    """BASELINE.json configs[1..3] at full size: in-chain G_r shares (configs[1], [2]), split-K GEMM for G_r (configs[3], M = 256); and two
    shapes off the streaming chain (M = 512; ragged M, T), where the two gradients come from the GEMMs over a_out / u_out."""
    from bench import CONFIGS
    from dgps_with_iwvi_amd import synthetic, backward
    spec = synthetic.make_spec(seed=0, parity=True, n_data=8192, **(CONFIGS[cfg] if isinstance(cfg, int) else cfg))
    model = synthetic.build_model(spec, gpu_device)
    zd = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in synthetic.make_noise(spec, seed=4)]
    e, g = backward.iw_elbo_and_gradients(model, zd)
    e_q, g_q = backward.iw_elbo_and_gradients(model, zd, wrt="final_q")
    assert float(e) == float(e_q) and len(g_q) == 2
    for k in g_q:
        assert torch.equal(g_q[k], g[k]), k


@pytest.mark.parametrize("dims,rows", [([9, 20, 20, 2], 100), ([5, 48, 48, 40, 4], 333), ([3, 64, 2], 64)])
def test_encoder_backward_matches_autodiff(gpu_device, dims, rows):
    """iwvi_encoder_backward (layers.py:137-152: tanh MLP with skip connections where widths match) vs float64 autograd."""
    import ctypes
    from dgps_with_iwvi_amd import _abi
    rng = np.random.default_rng(len(dims) * rows)
    XY = rng.standard_normal((rows, dims[0]))
    Ws = [rng.standard_normal((a, b)) * (2.0 / (a + b)) ** 0.5 for a, b in zip(dims[:-1], dims[1:])]
    bs = [rng.standard_normal(b) * 0.1 for b in dims[1:]]
    dout = rng.standard_normal((rows, dims[-1]))
    tW = [torch.tensor(w, requires_grad=True) for w in Ws]
    tb = [torch.tensor(b, requires_grad=True) for b in bs]
    H = torch.tensor(XY)
    for i, (W, b) in enumerate(zip(tW, tb)):
        H0 = H
        H = H @ W + b
        if i < len(tW) - 1:
            H = torch.tanh(H)
        if W.shape[0] == W.shape[1]:
            H = H + H0
    (H * torch.tensor(dout)).sum().backward()
    dev = gpu_device
    f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)
    dW_in, db_in = [f32(w) for w in Ws], [f32(b) for b in bs]
    dW, db = [torch.empty_like(w) for w in dW_in], [torch.empty_like(b) for b in db_in]
    cd = (ctypes.c_int32 * len(dims))(*dims)
    n = len(Ws)
    ws = torch.empty(_abi.lib().iwvi_encoder_backward_ws_bytes(rows, cd, n), dtype=torch.uint8, device=dev)
    xy, do = f32(XY), f32(dout)
    _abi.check(_abi.lib().iwvi_encoder_backward(_abi.ptr(xy), rows, _abi.ptr_array(dW_in), _abi.ptr_array(db_in), cd, n, _abi.ptr(do),
                                               _abi.ptr_array(dW), _abi.ptr_array(db), ws.data_ptr(), _abi.stream_ptr()))
    for i in range(n):
        _close("dW%d" % i, dW[i].cpu(), tW[i].grad.numpy(), rtol=1e-4)
        _close("db%d" % i, db[i].cpu(), tb[i].grad.numpy(), rtol=1e-4)


@pytest.mark.parametrize("Lw,B,K,with_dF,sampled", [(1, 1024, 20, True, 1), (2, 77, 7, True, 0), (3, 130, 70, False, 1)])
def test_fused_latent_variable_and_encoder_adjoint_equals_the_two_launches(gpu_device, Lw, B, K, with_dF, sampled):
    """iwvi_lv_encoder_backward (the training step's form: one launch) == iwvi_lv_layer_backward then iwvi_encoder_backward_act, bit for bit."""
    import ctypes
    from dgps_with_iwvi_amd import _abi
    dev = gpu_device
    rng = np.random.default_rng(Lw * B + K)
    f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)
    dims = [6, 20, 20, 2 * Lw]
    XY = f32(rng.standard_normal((B, dims[0])))
    Ws = [f32(rng.standard_normal((a, b)) * (2.0 / (a + b)) ** 0.5) for a, b in zip(dims[:-1], dims[1:])]
    bs = [f32(rng.standard_normal(b) * 0.1) for b in dims[1:]]
    enc_out = f32(rng.standard_normal((B, 2 * Lw)))
    eps = f32(rng.standard_normal((B * K, Lw)))
    ld, col0 = Lw + 4, 3
    dF = f32(rng.standard_normal((B * K, ld))) if with_dF else None
    w = f32(rng.random(B * K))
    cd = (ctypes.c_int32 * len(dims))(*dims)
    n = len(Ws)
    lib = _abi.lib()
    mu_p, sg_p = ctypes.c_void_p(enc_out.data_ptr()), ctypes.c_void_p(enc_out.data_ptr() + 4 * Lw)

    def outputs():
        return [torch.empty_like(t) for t in Ws], [torch.empty_like(t) for t in bs]
    ws = torch.empty(lib.iwvi_encoder_backward_ws_bytes(B, cd, n), dtype=torch.uint8, device=dev)
    dW1, db1 = outputs()
    d_enc = torch.empty(B, 2 * Lw, dtype=torch.float32, device=dev)
    _abi.check(lib.iwvi_lv_layer_backward(mu_p, sg_p, 2 * Lw, 1, _abi.ptr(eps), _abi.ptr(dF), ld if with_dF else 0, col0, _abi.ptr(w),
                                          Lw, B, K, sampled, _abi.ptr(d_enc), _abi.stream_ptr()))
    _abi.check(lib.iwvi_encoder_backward_act(_abi.ptr(XY), B, _abi.ptr_array(Ws), _abi.ptr_array(bs), cd, n, _abi.ACT_TANH, _abi.ptr(d_enc),
                                             _abi.ptr_array(dW1), _abi.ptr_array(db1), ws.data_ptr(), _abi.stream_ptr()))
    dW2, db2 = outputs()
    _abi.check(lib.iwvi_lv_encoder_backward(mu_p, sg_p, 2 * Lw, 1, _abi.ptr(eps), _abi.ptr(dF), ld if with_dF else 0, col0, _abi.ptr(w),
                                            Lw, B, K, sampled, _abi.ptr(XY), _abi.ptr_array(Ws), _abi.ptr_array(bs), cd, n, _abi.ACT_TANH,
                                            _abi.ptr_array(dW2), _abi.ptr_array(db2), ws.data_ptr(), _abi.stream_ptr()))
    torch.cuda.synchronize()
    for a, b in zip(dW1 + db1, dW2 + db2):
        assert float(a.abs().max()) > 0
        assert torch.equal(a, b)
    # an encoder whose output is not (means | raw) of this latent dimension is refused
    bad = (ctypes.c_int32 * len(dims))(*(dims[:-1] + [2 * Lw + 1]))
    assert lib.iwvi_lv_encoder_backward(mu_p, sg_p, 2 * Lw, 1, _abi.ptr(eps), None, 0, 0, _abi.ptr(w), Lw, B, K, sampled, _abi.ptr(XY),
                                        _abi.ptr_array(Ws), _abi.ptr_array(bs), bad, n, _abi.ACT_TANH, _abi.ptr_array(dW2), _abi.ptr_array(db2),
                                        ws.data_ptr(), _abi.stream_ptr()) == -1


def test_autograd_function_routes_the_hip_gradients(gpu_device):
    """backward.IwElbo: loss.backward() fills .grad of the model's own tensors with the HIP adjoints; a torch optimiser
    step on them changes what the kernels see."""
    from dgps_with_iwvi_amd import synthetic, backward
    spec = synthetic.make_spec(L=2, M=32, B=12, K=4, with_lv=True, seed=3)
    zs = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in synthetic.make_noise(spec, seed=4)]
    model = synthetic.build_model(spec, gpu_device)
    params = backward.parameter_list(model)
    for _, t in params:
        t.requires_grad_(True)
    elbo = backward.IwElbo.apply(model, zs, *[t for _, t in params])
    (-elbo).backward()
    with torch.no_grad():
        _, ref = backward.iw_elbo_and_gradients(model, zs)
    for n, t in params:
        assert torch.equal(t.grad, -ref[n].reshape(t.shape).to(t.dtype)), n
    before = float(elbo.detach())
    opt = torch.optim.SGD([t for n, t in params if n.endswith("q_mu")], lr=1e-4)
    opt.step()
    with torch.no_grad():
        after = float(backward.iw_elbo_and_gradients(model, zs)[0])
    assert after > before


@pytest.mark.parametrize("L,M,S,B,lv", [(2, 32, 3, 14, True), (2, 64, 2, 9, False)])
def test_vi_bound_gradients_match_oracle(gpu_device, L, M, S, B, lv):
    """DGP_VI (models.py:49-86): the same adjoints with uniform sample weights and the analytic local KL."""
    from dgps_with_iwvi_amd import synthetic, backward
    from dgps_with_iwvi_amd.models import DGP_VI
    from oracle.grad_oracle import iw_elbo_and_gradients
    spec = synthetic.make_spec(L=L, M=M, B=B, K=S, with_lv=lv, seed=41)
    zs = synthetic.make_noise(spec, seed=42)                                  # [B, S, dim]
    val, ref = iw_elbo_and_gradients(spec, zs, mode_vi=True)
    model = synthetic.build_model(spec, gpu_device, cls=DGP_VI, num_samples=S)
    zs_sn = [torch.as_tensor(np.asarray(z, dtype=np.float32).transpose(1, 0, 2).reshape(S * B, -1).copy(), device=gpu_device) for z in zs]
    fwd = model.compute_log_likelihood(zs_sn)                                 # the forward's own VI bound, same noise
    elbo, grads = backward.iw_elbo_and_gradients(model, zs_sn)
    assert abs(float(elbo) - val) <= 2e-4 * abs(val) and abs(fwd - val) <= 2e-4 * abs(val), (float(elbo), fwd, val)
    assert sorted(grads) == sorted(ref)
    for k, v in grads.items():
        _close(k, v.cpu().numpy().reshape(ref[k].shape), ref[k], rtol=5e-3)


def test_k_sharded_gradients_add_up_to_the_unsharded_gradient(gpu_device):
    """Two K-shards evaluated one after the other on one GPU (uneven split 4 + 3 of K = 7): the shares computed against the
    merged logsumexp, with the KL terms at 1/2 each, sum to the gradient of the unsharded model -- and of the oracle."""
    from dgps_with_iwvi_amd import synthetic, backward, sharding
    from oracle.grad_oracle import iw_elbo_and_gradients
    K, B = 7, 10
    spec = synthetic.make_spec(L=2, M=32, B=B, K=K, with_lv=True, seed=17)
    zs = synthetic.make_noise(spec, seed=18)
    val, ref = iw_elbo_and_gradients(spec, zs)
    tt = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=gpu_device)
    shards = [(0, 4), (4, 7)]
    models = [synthetic.build_model(dict(spec, K=hi - lo), gpu_device) for lo, hi in shards]
    pairs = []
    for (lo, hi), m in zip(shards, models):                          # pass 1: every shard's per-point (max, sum exp)
        def record(ms):
            pairs.append(ms.clone())
            return sharding.lse_from_pairs(ms[None])
        backward.iw_elbo_and_gradients(m, [tt(z[:, lo:hi]) for z in zs], exchange=record, K_total=K)
    lse = sharding.lse_from_pairs(torch.stack(pairs))
    total, elbos = None, []
    for (lo, hi), m in zip(shards, models):                          # pass 2: the shares against the merged normaliser
        e, g = backward.iw_elbo_and_gradients(m, [tt(z[:, lo:hi]) for z in zs], exchange=lambda ms: lse, K_total=K, kl_weight=0.5)
        elbos.append(float(e))
        total = g if total is None else {k: total[k] + g[k] for k in g}
    assert abs(elbos[0] - val) <= 2e-4 * abs(val) and abs(elbos[1] - val) <= 2e-4 * abs(val), (elbos, val)
    for k, v in total.items():
        _close(k, v.cpu().numpy().reshape(ref[k].shape), ref[k], rtol=5e-3)


@pytest.mark.parametrize("cfg,names", [
    (dict(L=2, M=128, B=1024, K=20, with_lv=True), ("l1.q_mu", "l1.Z", "l2.q_mu", "l1.ls", "l0.encW0", "l2.Z", "l1.q_sqrt", "l2.ls")),
    (dict(L=3, M=256, B=4096, K=50, with_lv=False), ("l0.q_mu", "l1.Z", "l2.q_mu", "l0.ls", "l1.q_sqrt", "l2.Z")),
    (dict(L=5, M=512, B=8192, K=100, with_lv=False), ("l0.q_mu", "l2.Z", "l4.q_mu", "l3.ls"))],
    ids=["configs2", "configs3", "configs4"])
def test_full_size_gradient_agrees_with_central_differences_of_the_forward(gpu_device, cfg, names):
    """BASELINE.json configs[2] / [3] / [4] at full size (20480 / 204800 / 819200 samples, where the float64 oracle is out of reach for a routine test): for a
    random direction d in each parameter group, the directional derivative g . d of the HIP backward pass against the central
    difference (ELBO(theta + eps d) - ELBO(theta - eps d)) / (2 eps) of the HIP forward path on the same injected noise.  The
    forward is float32 (relative noise ~1e-6 of |ELBO| ~ 6e6), so eps is sized to lift the difference two orders above it;
    tolerance 2 % (what is left is the truncation error of the difference quotient at eps = 0.02)."""
    from dgps_with_iwvi_amd import synthetic, backward
    spec = synthetic.make_spec(seed=0, parity=True, n_data=65536, **cfg)
    model = synthetic.build_model(spec, gpu_device)
    zs = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in synthetic.make_noise(spec, seed=1)]
    elbo0, grads = backward.iw_elbo_and_gradients(model, zs)
    f0 = model.compute_log_likelihood(zs)
    assert abs(float(elbo0) - f0) <= 1e-5 * abs(f0)                      # the saving forward and the benchmarked forward agree
    gen = torch.Generator(device="cpu").manual_seed(5)
    params = dict(backward.parameter_list(model))
    noise = 2e-6 * abs(f0)
    for name in names:
        p, g = params[name], grads[name].reshape(params[name].shape).double()
        d = torch.randn(p.shape, generator=gen).to(gpu_device)
        d = d / d.norm()
        if name.endswith("q_sqrt"):
            d = torch.tril(d)
        gd = float((g * d.double()).sum())
        eps = min(0.02, 100.0 * noise / max(abs(gd), 1e-30))
        with torch.no_grad():
            p.add_(eps * d); fp = model.compute_log_likelihood(zs)
            p.add_(-2 * eps * d); fm = model.compute_log_likelihood(zs)
            p.add_(eps * d)
        fd = (fp - fm) / (2 * eps)
        assert abs(fd - gd) <= 0.02 * abs(gd) + noise / eps, (name, fd, gd, eps)       # measured: 3e-5 .. 9e-3 relative


@pytest.mark.parametrize("M,R", [(128, 5), (40, 2), (256, 1), (512, 1)])
def test_dense_inverse_launch_equals_the_factorising_kernels_inverse(gpu_device, M, R):  # (R: the layer's own; the inverse does not depend on it)
    """``iwvi_gp_dense_inverse`` (Lm^-1 from the dense Lm of an ``IWVI_GP_WANT_LM`` state, one workgroup per 16-column block) against
    the inverse the factorising workgroup writes with ``IWVI_GP_WANT_DENSE``, and against float64 LAPACK on the same Lm."""
    import ctypes
    from dgps_with_iwvi_amd import _abi, synthetic
    from dgps_with_iwvi_amd.layers import GPLayer
    from dgps_with_iwvi_amd.temp_workaround import GpState, precompute_states
    spec = synthetic.make_spec(L=1, M=M, B=4, K=2, with_lv=False, seed=5)
    model = synthetic.build_model(spec, gpu_device)
    layer = [l for l in model.layers if isinstance(l, GPLayer)][0]
    Mp = (M + 15) // 16 * 16
    st_a, st_b = GpState(layer.num_inducing, layer.num_outputs, gpu_device), GpState(layer.num_inducing, layer.num_outputs, gpu_device)
    da, db = layer.state_desc(state=st_a), layer.state_desc(state=st_b)
    da.flags, db.flags = _abi.GP_WANT_DENSE, _abi.GP_WANT_LM
    precompute_states([da]); precompute_states([db])
    arr = (_abi.GpDesc * 1)(db)
    _abi.check(_abi.lib().iwvi_gp_dense_inverse(arr, 1, _abi.stream_ptr()))
    torch.cuda.synchronize()
    Lm_a, Li_a = (st_a.view(n, torch.float64, Mp * Mp).reshape(Mp, Mp) for n in ("Lm", "Linv"))
    Lm_b, Li_b = (st_b.view(n, torch.float64, Mp * Mp).reshape(Mp, Mp) for n in ("Lm", "Linv"))
    assert torch.equal(Lm_a, Lm_b)
    ref = torch.linalg.inv(Lm_b.cpu()).numpy()
    scale = np.abs(ref).max()
    assert np.abs(Li_b.cpu().numpy() - ref).max() <= 1e-9 * scale * max(1.0, np.linalg.cond(Lm_b.cpu().numpy()) * 1e-6)
    assert np.abs(Li_b.cpu().numpy() - Li_a.cpu().numpy()).max() <= 1e-9 * scale * max(1.0, np.linalg.cond(Lm_b.cpu().numpy()) * 1e-6)
    assert float(torch.triu(Li_b, 1).abs().max()) == 0.0


def test_inline_prepare_route_gives_the_default_routes_gradients(gpu_device, monkeypatch):
    """IWVI_BW_PREPARE=inline (one factorisation for both passes + iwvi_gp_dense_inverse) against the default (a second, dense
    factorisation on a side stream): same bound; gradients to float32 rounding of two float64 inverses that differ in the last bits."""
    from dgps_with_iwvi_amd import backward, synthetic
    spec = synthetic.make_spec(L=2, M=128, B=16, K=5, with_lv=True, seed=21)
    zs = synthetic.make_noise(spec, seed=3)
    model = synthetic.build_model(spec, gpu_device)
    zd = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in zs]
    e0, g0 = backward.iw_elbo_and_gradients(model, zd)
    monkeypatch.setenv("IWVI_BW_PREPARE", "inline")
    e1, g1 = backward.iw_elbo_and_gradients(model, zd)
    torch.cuda.synchronize()
    assert float(e0) == float(e1)
    for k in g0:
        a, b = g0[k].double().cpu().numpy(), g1[k].double().cpu().numpy()
        assert np.abs(a - b).max() <= 1e-4 * max(np.abs(a).max(), 1e-30), k


def test_branch_placement_does_not_change_a_bit(gpu_device, monkeypatch):
    """The last two parameter branches side by side (default for M <= 128) or both on the side stream (IWVI_BW_BRANCH_ORDER=old): the same
    launches in a different queue order -- the bound and every gradient must be bit-identical, eagerly and from a captured graph."""
    from dgps_with_iwvi_amd import backward, synthetic
    spec = synthetic.make_spec(L=3, M=64, B=16, K=5, with_lv=True, seed=13)
    zs = synthetic.make_noise(spec, seed=5)
    model = synthetic.build_model(spec, gpu_device)
    zd = [torch.as_tensor(np.asarray(z, dtype=np.float32), device=gpu_device) for z in zs]
    e0, g0 = backward.iw_elbo_and_gradients(model, zd)
    monkeypatch.setenv("IWVI_BW_BRANCH_ORDER", "old")
    e1, g1 = backward.iw_elbo_and_gradients(model, zd)
    monkeypatch.delenv("IWVI_BW_BRANCH_ORDER")
    torch.cuda.synchronize()
    assert float(e0) == float(e1) and sorted(g0) == sorted(g1)
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    # captured: warm up on a side stream, capture one evaluation, replay it
    s = torch.cuda.Stream(device=gpu_device)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        backward.iw_elbo_and_gradients(model, zd)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        e2, g2 = backward.iw_elbo_and_gradients(model, zd)
    graph.replay()
    torch.cuda.synchronize()
    assert float(e2) == float(e0)
    for k in g0:
        assert torch.equal(g0[k], g2[k]), k
