"""Multi-process (world_size 2, gloo, CPU) tests of the sharding host logic (dgps_with_iwvi_amd/sharding.py).

No HIP kernel runs here: each rank's partial results come from the fp64 oracle (test infrastructure) and are
pushed through the same collectives, layouts and merge formulae the GPU ranks use; the merged ELBO must equal
the unsharded oracle's.  Covers K-shard with an uneven split (K = 7 over 2 ranks) and N-shard."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dgps_with_iwvi_amd import sharding, synthetic
from oracle.from_spec import build_oracle, oracle_noise


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        K_total, B = 7, 12
        spec = synthetic.make_spec(L=2, M=16, B=B * (world if mode == "n" else 1), K=K_total, Dx=3, R=2, with_lv=True,
                                   seed=21, n_data=4096)
        zs = synthetic.make_noise(spec, seed=22)                     # [B, K, .] for the WHOLE job
        full = build_oracle(spec)
        ref = full.build_likelihood(oracle_noise(spec, zs))
        if mode == "k":
            Ks = sharding.split_samples(K_total, world)
            k0 = sum(Ks[:rank])
            sl = slice(k0, k0 + Ks[rank])
            m = build_oracle(spec, num_samples=Ks[rank])
            L_NK, global_kls, _, _, _ = m.log_weights(oracle_noise(spec, [z[:, sl] for z in zs]))
            mx = L_NK.max(1)
            ms = torch.tensor(np.stack([mx, np.exp(L_NK - mx[:, None]).sum(1)], 1))       # [B, 2] exchange unit
            got = sharding.k_shard_elbo(ms, float(np.sum(global_kls)), K_total, spec["n_data"] / B)
        else:
            lo, hi = sharding.split_points(spec["B"], world)[rank]
            sub = dict(spec, X=spec["X"][lo:hi], Y=spec["Y"][lo:hi], B=hi - lo)
            m = build_oracle(sub)
            local = m.build_likelihood(oracle_noise(sub, [z[lo:hi] for z in zs]))       # scaled by n_data / B_local
            got = sharding.n_shard_elbo(torch.tensor(local, dtype=torch.float64))
        q.put((rank, float(got), float(ref)))
    finally:
        dist.destroy_process_group()


def _grad_worker(rank, world, port, q):
    """N-shard training: each rank's gradient (gradient oracle on its points) through sharding.allreduce_gradients."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.grad_oracle import iw_elbo_and_gradients
        spec = synthetic.make_spec(L=2, M=12, B=10, K=4, Dx=3, R=2, with_lv=True, seed=31, n_data=2048)
        zs = synthetic.make_noise(spec, seed=32)
        _, ref = iw_elbo_and_gradients(spec, zs)
        lo, hi = sharding.split_points(spec["B"], world)[rank] if world == 2 else (0, spec["B"])
        lo, hi = (0, 7) if rank == 0 else (7, 10)                   # uneven on purpose: weights B_r / B
        sub = dict(spec, X=spec["X"][lo:hi], Y=spec["Y"][lo:hi], B=hi - lo)
        _, g = iw_elbo_and_gradients(sub, [z[lo:hi] for z in zs])
        g = {k: torch.tensor(np.asarray(v)) for k, v in g.items()}
        g = sharding.allreduce_gradients(g, weight=(hi - lo) / spec["B"])
        err = max(float(np.abs(g[k].numpy() - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-12)) for k in ref)
        q.put((rank, err, float(g["lik_var"])))
    finally:
        dist.destroy_process_group()


class _LocalRows:
    """What ``sharding.resolve_n_shard`` reads of a model: a rank built from ITS OWN rows (models.py:18: num_data = X.shape[0])."""

    def __init__(self, X, B):
        self._X_all = torch.as_tensor(X)
        self.X = self._X_all[:B]
        self.num_data = self._X_all.shape[0]


def _num_data_worker(rank, world, port, q):
    """every rank constructs its model from its own rows; the job's data-term scale must still be N_total / B.
    resolve_n_shard fixes model.num_data and the gradient weight; the merged oracle gradient equals the unsharded one."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.grad_oracle import iw_elbo_and_gradients
        n_tot, rows = 30, [(0, 18), (18, 30)][rank]                 # uneven row counts AND uneven minibatches
        Bs = [6, 4]
        spec = synthetic.make_spec(L=2, M=12, B=10, K=3, Dx=3, R=2, with_lv=True, seed=51, n_data=n_tot)
        zs = synthetic.make_noise(spec, seed=52)
        # the job's minibatch = rank 0's first 6 local rows + rank 1's first 4 local rows
        pick = np.r_[0:6, 18:22]
        job = dict(spec, X=spec["X"][pick], Y=spec["Y"][pick])
        _, ref = iw_elbo_and_gradients(job, zs)
        fake = _LocalRows(spec["X"][rows[0]:rows[1]], Bs[rank])
        assert fake.num_data == rows[1] - rows[0]
        total, w = sharding.resolve_n_shard(fake)
        assert total == n_tot and fake.num_data == n_tot and abs(w - Bs[rank] / 10.0) < 1e-15
        lo = 0 if rank == 0 else 6
        sub = dict(spec, X=spec["X"][rows[0]:rows[0] + Bs[rank]], Y=spec["Y"][rows[0]:rows[0] + Bs[rank]], B=Bs[rank],
                   n_data=fake.num_data)
        _, g = iw_elbo_and_gradients(sub, [z[lo:lo + Bs[rank]] for z in zs])
        g = sharding.allreduce_gradients({k: torch.tensor(np.asarray(v)) for k, v in g.items()}, weight=w)
        err = max(float(np.abs(g[k].numpy() - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-12)) for k in ref)
        # a caller-set num_data that disagrees between ranks is an error, an agreeing one is kept
        fake2 = _LocalRows(spec["X"][rows[0]:rows[1]], Bs[rank]); fake2.num_data = 1000 + rank
        try:
            sharding.resolve_n_shard(fake2)
            bad = False
        except ValueError:
            bad = True
        fake3 = _LocalRows(spec["X"][rows[0]:rows[1]], Bs[rank]); fake3.num_data = 4096
        ok3 = sharding.resolve_n_shard(fake3)[0] == 4096
        q.put((rank, err, bad and ok3))
    finally:
        dist.destroy_process_group()


def test_ranks_built_from_local_rows_get_the_job_num_data():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_num_data_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, flags in res:
        assert err <= 1e-10 and flags, (rank, err, flags)


def _kgrad_worker(rank, world, port, q):
    """K-shard training: each rank's share of the gradient -- autodiff of sum_n sum_{k in rank} sg(exp(L_nk - LSE_n)) L_nk * scale
    - KL / world on the float64 restatement, LSE from sharding.lse_from_pairs on the all-gathered pairs -- summed by
    sharding.allreduce_gradients(weight=1): equals the unsharded gradient oracle."""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle.grad_oracle import iw_elbo_and_gradients
        from oracle.ref_torch_cpu import CpuDGP
        K = 7
        spec = synthetic.make_spec(L=2, M=12, B=9, K=K, Dx=3, R=2, with_lv=True, seed=41, n_data=1024)
        zs = synthetic.make_noise(spec, seed=42)
        val, ref = iw_elbo_and_gradients(spec, zs)
        Ks = sharding.split_samples(K, world)
        k0 = sum(Ks[:rank]); sl = slice(k0, k0 + Ks[rank])
        m = CpuDGP(dict(spec, K=Ks[rank]), torch.float64)
        leaves = {}
        for i, L in enumerate(m.layers):
            if L["type"] == "lv":
                L["W"] = [w.clone().requires_grad_(True) for w in L["W"]]
                leaves["l%d.encW0" % i] = L["W"][0]
            else:
                L["q_mu"] = L["q_mu"].clone().requires_grad_(True); leaves["l%d.q_mu" % i] = L["q_mu"]
                L["Z"] = L["Z"].clone().requires_grad_(True); leaves["l%d.Z" % i] = L["Z"]
        L_NK, glob = m.log_weights_tensor([z[:, sl] for z in zs])
        mx = L_NK.detach().max(1).values
        ms = torch.stack([mx, torch.exp(L_NK.detach() - mx[:, None]).sum(1)], 1)               # the forward path's exchange unit
        gathered = torch.empty((world,) + tuple(ms.shape), dtype=ms.dtype)
        dist.all_gather_into_tensor(gathered.view(-1), ms.contiguous().view(-1))
        lse = sharding.lse_from_pairs(gathered)
        share = (torch.exp(L_NK.detach() - lse[:, None]) * L_NK).sum() * (spec["n_data"] / spec["B"]) - glob / world
        share.backward()
        g = sharding.allreduce_gradients({k: v.grad.clone() for k, v in leaves.items()}, weight=1.0)
        err = max(float(np.abs(g[k].numpy() - ref[k]).max() / max(np.abs(ref[k]).max(), 1e-12)) for k in g)
        bound = float((lse - np.log(K)).sum() * (spec["n_data"] / spec["B"]) - glob.detach())
        q.put((rank, err, abs(bound - val) / abs(val)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_k_sharded_gradient_shares_sum_to_the_unsharded_gradient(world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_kgrad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, verr in res:
        assert err <= 1e-10 and verr <= 1e-12, (rank, err, verr)


def test_data_parallel_gradient_equals_unsharded():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, _ in res:
        assert err <= 1e-10, (rank, err)
    assert res[0][2] == res[1][2]


@pytest.mark.parametrize("mode,world", [("k", 2), ("n", 2), ("k", 4), ("n", 3)])     # K = 7 over 4 ranks: 2, 2, 2, 1 samples
def test_sharded_elbo_equals_unsharded(mode, world):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, got, ref in res:
        assert abs(got - ref) <= 1e-9 * abs(ref), (mode, rank, got, ref)
    assert all(r[1] == res[0][1] for r in res)                       # every rank holds the same answer


def test_split_helpers():
    assert sharding.split_samples(20, 8) == [3, 3, 3, 3, 2, 2, 2, 2]
    assert sum(sharding.split_samples(100, 8)) == 100 and sharding.split_samples(50, 8)[:2] == [7, 7]
    with pytest.raises(ValueError):
        sharding.split_samples(3, 8)
    r = sharding.split_points(1030, 8)
    assert r[0] == (0, 129) and r[-1][1] == 1030 and all(b - a in (128, 129) for a, b in r)


def test_merge_reference_matches_logsumexp():
    rng = np.random.default_rng(0)
    L = rng.standard_normal((6, 11)) * 5
    parts = [L[:, :4], L[:, 4:9], L[:, 9:]]
    ms = torch.tensor(np.stack([np.stack([p.max(1), np.exp(p - p.max(1, keepdims=True)).sum(1)], 1) for p in parts]))
    got = sharding.merge_lse_reference(ms, 11).numpy()
    ref = np.log(np.exp(L).sum(1)) - np.log(11)
    np.testing.assert_allclose(got, ref, rtol=1e-12)
