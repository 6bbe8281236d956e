"""The training step of experiments/build_models.py:284-304 on the GPU (NatGrad on the final layer's q(u), Adam on the
rest) against the NumPy float64 optimiser oracle (oracle/optim_oracle.py) driven by the gradient oracle."""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import optim_oracle as oo   # noqa: E402

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev)


@pytest.mark.parametrize("M,R", [(8, 1), (40, 2), (100, 3), (128, 1), (128, 5), (200, 1), (256, 1), (512, 1)])      # M <= 128: spread / one workgroup per latent GP; beyond: the multi-launch path (the final layers of configs[3] / [4])
def test_natgrad_step_matches_oracle(gpu_device, M, R):
    import ctypes
    from dgps_with_iwvi_amd import _abi
    lib = _abi.lib()
    # ABI 16: the workspace sized for R latent GPs always takes the spread route at M <= 128 (the R-independent size fell back
    # to the one-workgroup kernel beyond R = 3 without saying so); the route that ran is readable
    wsx = torch.empty(lib.iwvi_natgrad_ws_bytes_ex(M, R), dtype=torch.uint8, device=gpu_device)
    assert wsx.numel() >= lib.iwvi_natgrad_ws_bytes(M) > 0
    rng = np.random.default_rng(M)
    q_mu = rng.standard_normal((M, R)).astype(np.float32)
    q_sqrt = (np.tril(rng.standard_normal((R, M, M))) * 0.1 + np.eye(M)).astype(np.float32)
    g_mu = rng.standard_normal((M, R)).astype(np.float32)            # gradients of the ELBO
    g_sqrt = np.tril(rng.standard_normal((R, M, M))).astype(np.float32) * np.float32(0.5 * min(1.0, (64.0 / M) ** 0.5))   # keeps -2 theta_2' positive definite
    ref_mu, ref_sqrt = oo.natgrad_step(q_mu, q_sqrt, -g_mu.astype(np.float64), -g_sqrt.astype(np.float64), 0.05)
    d_mu, d_sqrt, dg_mu, dg_sqrt = (_t(a, gpu_device) for a in (q_mu, q_sqrt, g_mu, g_sqrt))
    ws = torch.empty(_abi.lib().iwvi_natgrad_ws_bytes(M), dtype=torch.uint8, device=gpu_device)
    _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(d_mu), _abi.ptr(d_sqrt), _abi.ptr(dg_mu), _abi.ptr(dg_sqrt), M, R, 0.05,
                                           ws.data_ptr(), _abi.stream_ptr()))
    np.testing.assert_allclose(d_mu.cpu().numpy(), ref_mu, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(d_sqrt.cpu().numpy(), ref_sqrt, rtol=2e-5, atol=2e-6)
    assert float(torch.triu(d_sqrt, 1).abs().max()) == 0.0
    x_mu, x_sqrt = _t(q_mu, gpu_device), _t(q_sqrt, gpu_device)
    _abi.check(lib.iwvi_natgrad_step_ex(_abi.ptr(x_mu), _abi.ptr(x_sqrt), _abi.ptr(dg_mu), _abi.ptr(dg_sqrt), M, R, 0.05,
                                        wsx.data_ptr(), wsx.numel(), _abi.stream_ptr()))
    assert lib.iwvi_debug_last_natgrad_route() == (2 if M <= 128 else 0)
    np.testing.assert_allclose(x_mu.cpu().numpy(), ref_mu, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(x_sqrt.cpu().numpy(), ref_sqrt, rtol=2e-5, atol=2e-6)
    with pytest.raises(_abi.IwviError):                          # a workspace smaller than the R-independent minimum is refused, not overrun
        _abi.check(lib.iwvi_natgrad_step_ex(_abi.ptr(x_mu), _abi.ptr(x_sqrt), _abi.ptr(dg_mu), _abi.ptr(dg_sqrt), M, R, 0.05,
                                            wsx.data_ptr(), 1024, _abi.stream_ptr()))
    if M <= 128:                                                 # the multi-launch path (kept for M > 128) on the same inputs
        e_mu, e_sqrt = _t(q_mu, gpu_device), _t(q_sqrt, gpu_device)
        _abi.set_debug_option("IWVI_NATGRAD_UNFUSED", 1)         # (a route switch of the library: iwvi_debug_set_option, not the environment)
        try:
            _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(e_mu), _abi.ptr(e_sqrt), _abi.ptr(dg_mu), _abi.ptr(dg_sqrt), M, R, 0.05,
                                                   ws.data_ptr(), _abi.stream_ptr()))
        finally:
            _abi.set_debug_option("IWVI_NATGRAD_UNFUSED", 0)
        np.testing.assert_allclose(e_mu.cpu().numpy(), d_mu.cpu().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(e_sqrt.cpu().numpy(), d_sqrt.cpu().numpy(), rtol=2e-6, atol=2e-7)
        # ... and the one-workgroup form of the step (the default spreads it over four launches: k_ng_chol, k_sb_inv, k_ng_rows)
        f_mu, f_sqrt = _t(q_mu, gpu_device), _t(q_sqrt, gpu_device)
        _abi.set_debug_option("IWVI_NG_ONE_WG", 1)
        try:
            _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(f_mu), _abi.ptr(f_sqrt), _abi.ptr(dg_mu), _abi.ptr(dg_sqrt), M, R, 0.05,
                                                   ws.data_ptr(), _abi.stream_ptr()))
        finally:
            _abi.set_debug_option("IWVI_NG_ONE_WG", 0)
        np.testing.assert_allclose(f_mu.cpu().numpy(), d_mu.cpu().numpy(), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(f_sqrt.cpu().numpy(), d_sqrt.cpu().numpy(), rtol=2e-6, atol=2e-7)
        # the step leaves the same bits when repeated on the same inputs (fixed summation orders; the last-arriving workgroup only reads)
        h_mu, h_sqrt = _t(q_mu, gpu_device), _t(q_sqrt, gpu_device)
        _abi.check(_abi.lib().iwvi_natgrad_step(_abi.ptr(h_mu), _abi.ptr(h_sqrt), _abi.ptr(dg_mu), _abi.ptr(dg_sqrt), M, R, 0.05,
                                               ws.data_ptr(), _abi.stream_ptr()))
        assert torch.equal(h_mu, d_mu) and torch.equal(h_sqrt, d_sqrt)


def test_adam_steps_match_oracle(gpu_device):
    from dgps_with_iwvi_amd import _abi
    rng = np.random.default_rng(0)
    shapes, pos = [(7, 3), (5,), (1,), (300,)], [False, True, True, False]
    params = [rng.standard_normal(s) if not p else rng.uniform(0.05, 3.0, s) for s, p in zip(shapes, pos)]
    ref = oo.Adam(params, pos, lr=0.01)
    dp = [_t(p, gpu_device) for p in params]
    st = [tuple(torch.empty_like(p) for _ in range(3)) for p in dp]

    def call(grads, t, init):
        arr = (_abi.AdamTensor * len(dp))()
        keep = []
        for i, p in enumerate(dp):
            a = arr[i]
            a.param, a.x, a.m, a.v, a.n, a.transform = p.data_ptr(), st[i][0].data_ptr(), st[i][1].data_ptr(), st[i][2].data_ptr(), p.numel(), int(pos[i])
            if grads is not None:
                keep.append(_t(grads[i], gpu_device)); a.grad = keep[-1].data_ptr()
        _abi.check(_abi.lib().iwvi_adam_step(arr, len(dp), 0.01, 0.9, 0.999, 1e-8, t, 0, init, _abi.stream_ptr()))
        torch.cuda.synchronize()

    call(None, 1, 1)
    for t in range(1, 6):
        grads = [rng.standard_normal(s) for s in shapes]
        new = ref.step(grads)
        call(grads, t, 0)
        for got, want in zip(dp, new):
            np.testing.assert_allclose(got.cpu().numpy(), want, rtol=3e-5, atol=3e-6)


class _OracleTrainer:
    """The same step on the NumPy side: gradient oracle + optimiser oracle, parameters kept in the spec."""

    def __init__(self, spec, lr, gamma, fix_linear=True):
        from oracle.grad_oracle import iw_elbo_and_gradients
        self.grad = iw_elbo_and_gradients
        self.spec, self.lr, self.gamma = spec, lr, gamma
        self.n = len(spec["layers"])
        self.names, pos = [], []
        for i, l in enumerate(spec["layers"]):
            if l["type"] == "lv":
                for j in range(len(l["enc_W"])):
                    self.names += ["l%d.encW%d" % (i, j), "l%d.encb%d" % (i, j)]; pos += [False, False]
            elif i == self.n - 1:
                self.names += ["l%d.Z" % i, "l%d.ls" % i, "l%d.var" % i]; pos += [False, True, True]
            else:
                self.names += ["l%d.Z" % i, "l%d.ls" % i, "l%d.q_mu" % i, "l%d.q_sqrt" % i]; pos += [False, True, False, False]
            if l["type"] != "lv" and not fix_linear:
                if l["W"] is not None:
                    self.names.append("l%d.W" % i); pos.append(False)
                if l["mf"][0] == "linear":
                    self.names.append("l%d.mfA" % i); pos.append(False)
        self.names.append("lik_var"); pos.append(True)
        self.adam = oo.Adam([self.get(k) for k in self.names], pos, lr)
        self.signif = None       # per parameter: entries whose gradient was never negligible (Adam normalises the step
        #                          size, so an entry whose gradient is within float32 error of 0 (the kernels are ~1e-3 of the
        #                          array's max-norm off the float64 oracle) moves by ~lr in the direction of rounding noise)

    def get(self, name):
        if name == "lik_var":
            return np.asarray(self.spec["lik_var"], dtype=np.float64)
        i, key = name.split(".")
        l = self.spec["layers"][int(i[1:])]
        if key.startswith("encW"):
            return l["enc_W"][int(key[4:])]
        if key.startswith("encb"):
            return l["enc_b"][int(key[4:])]
        if key == "mfA":
            return np.asarray(l["mf"][1], dtype=np.float64)
        return np.asarray(l[key], dtype=np.float64)

    def put(self, name, v):
        if name == "lik_var":
            self.spec["lik_var"] = float(v); return
        i, key = name.split(".")
        l = self.spec["layers"][int(i[1:])]
        if key.startswith("encW"):
            l["enc_W"][int(key[4:])] = v
        elif key.startswith("encb"):
            l["enc_b"][int(key[4:])] = v
        elif key == "var":
            l[key] = float(v)
        elif key == "mfA":
            l["mf"] = (l["mf"][0], v) + tuple(l["mf"][2:])
        else:
            l[key] = v

    def step(self, zs_ng, zs_adam):
        f = self.spec["layers"][-1]
        _, g = self.grad(self.spec, zs_ng)
        i = self.n - 1
        f["q_mu"], f["q_sqrt"] = oo.natgrad_step(f["q_mu"], f["q_sqrt"], -g["l%d.q_mu" % i], -g["l%d.q_sqrt" % i], self.gamma)
        val, g = self.grad(self.spec, zs_adam)
        sig = [np.abs(np.asarray(g[k])) > 5e-2 * max(np.abs(np.asarray(g[k])).max(), 1e-300) for k in self.names]
        self.signif = sig if self.signif is None else [a & b for a, b in zip(self.signif, sig)]
        new = self.adam.step([-np.asarray(g[k]) for k in self.names])
        for k, v in zip(self.names, new):
            self.put(k, v)
        return val


@pytest.mark.parametrize("L,M,K,B,lv,fix_linear", [(2, 32, 4, 12, True, True), (2, 64, 3, 16, False, True), (3, 32, 3, 10, True, False)])
def test_training_steps_follow_the_oracle_loop(gpu_device, L, M, K, B, lv, fix_linear):
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=L, M=M, B=B, K=K, with_lv=lv, seed=11)
    model = synthetic.build_model(spec, gpu_device)
    ospec = copy.deepcopy(spec)
    tr = Trainer(model, lr=5e-3, gamma=1e-2, fix_linear=fix_linear)
    ot = _OracleTrainer(ospec, 5e-3, 1e-2, fix_linear)
    assert sorted(n for n, _, _ in tr._entries) == sorted(ot.names)
    for s in range(3):
        zs_a, zs_b = synthetic.make_noise(spec, seed=100 + 2 * s), synthetic.make_noise(spec, seed=101 + 2 * s)
        e_gpu = float(tr.step([_t(z, gpu_device) for z in zs_a], [_t(z, gpu_device) for z in zs_b]))
        e_ref = ot.step(zs_a, zs_b)
        assert abs(e_gpu - e_ref) <= 3e-4 * abs(e_ref), (s, e_gpu, e_ref)
    sig = dict(zip(ot.names, ot.signif))
    n_sig = n_all = 0
    for name, p, _ in tr._entries:
        ref = np.asarray(ot.get(name), dtype=np.float64).reshape(-1)
        got = p.detach().cpu().numpy().astype(np.float64).reshape(-1)
        m = np.asarray(sig[name]).reshape(-1)
        n_sig += int(m.sum()); n_all += m.size
        np.testing.assert_allclose(got[m], ref[m], rtol=2e-4, atol=2e-4, err_msg=name)
        np.testing.assert_allclose(got, ref, rtol=0, atol=2 * 3.5 * 5e-3, err_msg=name)  # each side moves at most ~3 Adam steps of lr
    assert n_sig >= 100, (n_sig, n_all)         # (most of an inner q_sqrt is within rounding of zero gradient)
    f, fo = model.layers[-1], ospec["layers"][-1]
    np.testing.assert_allclose(f.q_mu.cpu().numpy(), fo["q_mu"], rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(f.q_sqrt.cpu().numpy(), fo["q_sqrt"], rtol=2e-3, atol=2e-4)
    assert model.likelihood.variance == pytest.approx(ospec["lik_var"], rel=1e-3)


def test_training_raises_the_bound_on_fixed_noise(gpu_device):
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=64, K=5, with_lv=True, seed=5)
    model = synthetic.build_model(spec, gpu_device)
    zs = [_t(z, gpu_device) for z in synthetic.make_noise(spec, seed=1)]
    before = model.compute_log_likelihood(zs)
    tr = Trainer(model, lr=5e-3, gamma=1e-2)
    for _ in range(25):
        tr.step()
    after = model.compute_log_likelihood(zs)
    assert after > before, (before, after)


def test_checkpoint_resume_continues_the_same_trajectory(gpu_device, tmp_path):
    """Parameters + optimiser state through build_models.save_checkpoint / load_checkpoint: 3 steps, save, restore into a
    fresh model and trainer, 2 more steps == 5 uninterrupted steps (same injected noise), bit for bit."""
    from dgps_with_iwvi_amd import synthetic, build_models
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=12, K=4, with_lv=True, seed=23)
    noise = [[_t(z, gpu_device) for z in synthetic.make_noise(spec, seed=300 + s)] for s in range(10)]

    def run(model, tr, steps):
        for s in steps:
            tr.step(noise[2 * s], noise[2 * s + 1])

    a = synthetic.build_model(spec, gpu_device); ta = Trainer(a)
    run(a, ta, range(5))
    b = synthetic.build_model(spec, gpu_device); tb = Trainer(b)
    run(b, tb, range(3))
    path = str(tmp_path / "ckpt.npz")
    build_models.save_checkpoint(b, path, tb)
    c = synthetic.build_model(spec, gpu_device); tc = Trainer(c)
    build_models.load_checkpoint(c, path, tc)
    assert tc.global_step == 3 and tc.adam_t == 3
    run(c, tc, range(3, 5))
    for (name, pa, _), (_, pc, _) in zip(ta._entries, tc._entries):
        assert torch.equal(pa, pc), name
    assert torch.equal(a.layers[-1].q_mu, c.layers[-1].q_mu) and torch.equal(a.layers[-1].q_sqrt, c.layers[-1].q_sqrt)
    assert a.likelihood.variance == c.likelihood.variance


def test_checkpoint_resume_with_minibatches_device_noise_and_trainable_linear_parts(gpu_device, tmp_path):
    """the session state a tf.train.Saver would restore besides the variables -- the shuffled minibatch iterator
    (gpflow.Minibatch(seed=0), models.py:25-26), the device noise counters, the Linear mean function's A (trainable with
    fix_linear=False) -- travels in the checkpoint: minibatch_size = 16 of 64 rows, noise drawn on the device, fix_linear=False;
    3 steps, save, restore into a fresh model + trainer, 3 more steps == 6 uninterrupted steps, bit for bit."""
    from dgps_with_iwvi_amd import synthetic, build_models, settings
    from dgps_with_iwvi_amd.models import DGP_IWVI
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=64, K=4, with_lv=True, seed=29)

    def fresh():
        settings.set_seed(11)
        m = synthetic.build_model(spec, gpu_device)
        m2 = DGP_IWVI(spec["X"], spec["Y"], m.layers, m.likelihood, num_samples=4, minibatch_size=16).to(gpu_device)
        return m2, Trainer(m2, fix_linear=False)

    a, ta = fresh()
    for _ in range(6):
        ta.step()
    b, tb = fresh()
    for _ in range(3):
        tb.step()
    path = str(tmp_path / "ckpt_mb.npz")
    build_models.save_checkpoint(b, path, tb)
    c, tc = fresh()
    tc.step()                                                    # disturb the fresh state: everything must come from the file
    build_models.load_checkpoint(c, path, tc)
    assert tc.global_step == 3 and torch.equal(c.X, b.X)         # the restored model looks at the same minibatch
    for _ in range(3):
        tc.step()
    for (name, pa, _), (_, pc, _) in zip(ta._entries, tc._entries):
        assert torch.equal(pa, pc), name
    assert torch.equal(a.layers[1].mean_function.A, c.layers[1].mean_function.A)
    assert torch.equal(a.layers[-1].q_sqrt, c.layers[-1].q_sqrt) and a.likelihood.variance == c.likelihood.variance
    assert int(a._words()[1]) == int(c._words()[1])


def test_graph_mode_training_step_equals_the_eager_one(gpu_device):
    """Trainer(use_graph=True): each op of a step replayed from a hipGraph, trained scalars and Adam's step count on the
    device -- the trajectory equals the eager trainer's bit for bit (same kernels, same device noise counters)."""
    from dgps_with_iwvi_amd import synthetic, settings
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=64, K=5, with_lv=True, seed=31)
    out = []
    for use_graph in (False, True):
        settings.set_seed(3)
        model = synthetic.build_model(spec, gpu_device)
        tr = Trainer(model, use_graph=use_graph, check_finite=False)
        vals = [float(tr.step()) for _ in range(6)]
        out.append((vals, [p.clone() for _, p, _ in tr._entries], model.likelihood.variance, model.layers[-1].q_sqrt.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for pa, pb in zip(out[0][1], out[1][1]):
        assert torch.equal(pa, pb)
    assert out[0][2] == out[1][2] and torch.equal(out[0][3], out[1][3])
    assert out[0][2] != spec["lik_var"]                          # the likelihood variance did move (and was read back lazily)


@pytest.mark.parametrize("case", ["full_batch_eager", "full_batch_graph", "minibatch_graph", "f64_route_graph"])
def test_one_factorisation_per_step_is_the_same_step(gpu_device, case):
    """Trainer(one_factorisation=True), the default: the Adam op re-packs the final layer's q(u) images instead of factorising every
    K_uu again (IWVI_GP_REUSE_FACTOR), and its dense factors were formed beside the natural-gradient update of the first op
    (IWVI_GP_FACTOR_ONLY on the layer whose q(u) is being written).  Same kernels on the same numbers: bounds and parameters are
    bit-identical to the step that factorises per op -- eager and captured, full-batch (one graph) and minibatched (two graphs), and
    with an inner layer on the float64 stage-1 route (its Lm^-1 and z~ images must survive the short precompute)."""
    from dgps_with_iwvi_amd import synthetic, settings
    from dgps_with_iwvi_amd.training import Trainer
    graph = not case.endswith("eager")
    if case == "f64_route_graph":
        spec = synthetic.make_spec(L=2, M=32, B=64, K=5, with_lv=False, seed=31, Dx=2)
    elif case == "minibatch_graph":
        spec = synthetic.make_spec(L=2, M=32, B=96, K=4, with_lv=True, seed=37, n_data=96)
    else:
        spec = synthetic.make_spec(L=2, M=32, B=64, K=5, with_lv=True, seed=31)
    out = []
    for one in (False, True):
        settings.set_seed(3)
        model = _minibatched_lv_model(spec, gpu_device, 16) if case == "minibatch_graph" else synthetic.build_model(spec, gpu_device)
        tr = Trainer(model, use_graph=graph, check_finite=False, one_factorisation=one)
        if case == "f64_route_graph":
            assert any(l.uses_f64_stage1() for l in model.layers if hasattr(l, "uses_f64_stage1"))
        vals = [float(tr.step()) for _ in range(6)]
        out.append((vals, [p.clone() for _, p, _ in tr._entries], model.layers[-1].q_mu.clone(), model.layers[-1].q_sqrt.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for pa, pb in zip(out[0][1], out[1][1]):
        assert torch.equal(pa, pb)
    assert torch.equal(out[0][2], out[1][2]) and torch.equal(out[0][3], out[1][3])


def test_graph_mode_follows_the_staircase_decay(gpu_device):
    """lr / gamma enter the captured update kernels by value: crossing a decay boundary (every 1000 steps, build_models.py:276-282)
    re-captures the two graphs, and the trajectory stays the eager one's bit for bit."""
    from dgps_with_iwvi_amd import synthetic, settings
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=16, B=16, K=2, with_lv=True, seed=5)
    out = []
    for use_graph in (False, True):
        settings.set_seed(9)
        model = synthetic.build_model(spec, gpu_device)
        tr = Trainer(model, use_graph=use_graph, check_finite=False, lr=1e-3, gamma=1e-3)
        tr.global_step = 996                                   # (the decay epoch is global_step // 1000)
        vals = [float(tr.step()) for _ in range(8)]
        out.append((vals, [p.clone() for _, p, _ in tr._entries], model.layers[-1].q_sqrt.clone()))
        if use_graph:
            assert set(tr._graphs) == {"step"} and tr._graphs["step"][0][0] == 1   # (key = (decay epoch, route key); full-batch data: the two ops are one graph) re-captured for epoch 1
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for pa, pb in zip(out[0][1], out[1][1]):
        assert torch.equal(pa, pb)
    assert torch.equal(out[0][2], out[1][2])


def test_a_diverged_natural_gradient_step_is_reported(gpu_device):
    """gamma far too large: -2 theta_2 turns indefinite, the Cholesky inside the step produces NaN; the trainer says so (the
    reference's TensorFlow Cholesky would raise at the same point) instead of training on."""
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=16, K=4, with_lv=True, seed=2)
    model = synthetic.build_model(spec, gpu_device)
    tr = Trainer(model, gamma=1e6)
    with pytest.raises(FloatingPointError):
        for _ in range(3):
            tr.step()
    # graph mode: the same check, every ``check_every`` steps
    model = synthetic.build_model(spec, gpu_device)
    tr = Trainer(model, gamma=1e6, use_graph=True, check_every=4)
    with pytest.raises(FloatingPointError):
        for _ in range(8):
            tr.step()
    assert tr.global_step == 4


def _minibatched_lv_model(spec, dev, mb):
    """A latent-variable model over ALL the spec's rows, stepping through shuffled minibatches of ``mb`` rows."""
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.models import DGP_IWVI
    m = synthetic.build_model(spec, dev)
    return DGP_IWVI(spec["X"], spec["Y"], m.layers, m.likelihood, num_samples=spec["K"], minibatch_size=mb).to(dev)


def test_graph_mode_with_minibatches_feeds_the_encoder_fresh_rows(gpu_device):
    """minibatch_size < N with a latent-variable layer: the encoder's [x, y] rows are refreshed OUTSIDE the captured op
    at every minibatch change (a cat keyed by the minibatch serial is not recorded into the graph when the key did not
    move inside it) -- the graph-mode trajectory is the eager one bit for bit over several different minibatches."""
    from dgps_with_iwvi_amd import synthetic, settings
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=96, K=4, with_lv=True, seed=37, n_data=96)
    out = []
    for use_graph in (False, True):
        settings.set_seed(11)
        model = _minibatched_lv_model(spec, gpu_device, 16)
        tr = Trainer(model, use_graph=use_graph, check_finite=False)
        vals = [float(tr.step()) for _ in range(7)]             # 14 minibatches: more than one epoch of 6
        out.append((vals, [p.clone() for _, p, _ in tr._entries], model._xy_minibatch().clone(), model.X.clone(), model.Y.clone()))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for pa, pb in zip(out[0][1], out[1][1]):
        assert torch.equal(pa, pb)
    # and the cached rows ARE the current minibatch's
    for o in out:
        assert torch.equal(o[2], torch.cat([o[3], o[4]], -1))


def test_graph_mode_host_scalars_follow_the_device_values(gpu_device):
    """Reading ``likelihood.variance`` / the final ``kern.variance`` in the middle of a graph-mode run must not freeze the
    host copy: every later read returns the device value of THAT moment (replays run no host code, so ``step()`` flags
    the copies stale itself)."""
    from dgps_with_iwvi_amd import synthetic, settings
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=2, M=32, B=32, K=4, with_lv=True, seed=41)
    settings.set_seed(5)
    model = synthetic.build_model(spec, gpu_device)
    tr = Trainer(model, use_graph=True, check_finite=False)
    kern = model.layers[-1]._base_kern()
    seen = []
    for _ in range(5):
        tr.step()
        lv, kv = model.likelihood.variance, kern.variance        # a mid-training read (evaluation, checkpoint)
        assert lv == float(tr._scalars[-1][0].item()) and kv == float(tr._scalars[0][0].item())
        seen.append((lv, kv))
    assert len(set(seen)) == 5, seen                             # the values moved at every step and were seen moving


def test_trainer_counts_ranks_not_group_arguments(gpu_device):
    """``group=None`` is the default group once torch.distributed is initialised: the N-shard bookkeeping and the
    capture mode follow the WORLD SIZE (1 rank: plain single-GPU training, the step is ONE graph; more ranks: graph segments around the
    collectives, tests/test_gpu_multirank.py::test_sharded_training_step_as_graph_segments)."""
    import torch.distributed as dist
    from dgps_with_iwvi_amd import synthetic
    from dgps_with_iwvi_amd.training import Trainer
    spec = synthetic.make_spec(L=1, M=16, B=16, K=2, seed=1)
    model = synthetic.build_model(spec, gpu_device)
    assert Trainer(model, use_graph=True).world == 1
    if not dist.is_initialized():
        import tempfile
        f = tempfile.NamedTemporaryFile(delete=False)
        dist.init_process_group("gloo", init_method="file://" + f.name, rank=0, world_size=1)
        try:
            assert Trainer(model, use_graph=True).world == 1     # 1 rank: still a single-GPU step
        finally:
            dist.destroy_process_group()
