"""Multi-rank plumbing of bench.py and sharding.OverlappedExchange on ONE GPU (the pool hands out one GPU at a time; the
8-GPU run is the driver's).  bench.py is started as FRESH child processes -- 2 ranks over gloo, both on cuda:0 -- exactly
as ``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`` would start it (same env contract), with
``--check``: after its timed region every rank evaluates its shard of ONE job-wide injected noise draw through the very
path the timed loop uses (per-slot hipGraphs aside), rank 0 also evaluates the unsharded job, and the JSON line carries both."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(world, shard, extra=(), backend="gloo", timeout=600, train_leg=False):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), IWVI_BENCH_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "8", "--warmup", "2", "--config", "1",
               "--shard", shard, "--check", "--no-cpu-baseline", "--no-train-leg", *(("--sharded-train-leg",) if train_leg else ()), *extra]
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    outs = []
    for p in procs:
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out, err))
    for rc, out, err in outs:
        assert rc == 0, err[-3000:]
    lines = [l for l in outs[0][1].splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][1]
    assert all(not l.startswith("{") for _, o, _ in outs[1:] for l in o.splitlines())     # only rank 0 prints
    return json.loads(lines[0])


@pytest.mark.parametrize("shard", ["k", "n"])
def test_bench_two_ranks_on_one_gpu_merges_to_the_unsharded_elbo(gpu_device, shard):
    res = _run_bench(2, shard)
    assert res["n_gpus"] == 2 and res["n_ranks_seen"] == 2 and res["steps"] == 8
    assert res["config"]["sharding"] == shard + "-shard" and "one exchange per replay" in res["config"]["launch"]
    c = res["check"]
    assert c["K_total"] == (10 if shard == "k" else 5) and c["B_total"] == (1024 if shard == "k" else 2048)
    assert c["all_steps_equal"]
    # float32 logsumexp merged in a different order than the single-rank reduction
    assert c["rel_diff"] <= 2e-6, c
    assert np.isfinite(res["value"]) and res["value"] > 0 and np.isfinite(res["elbo"])
    assert res["exchange_ms"] is not None and 0.0 < res["exchange_ms"] < 50.0 and res["evaluations_per_exchange"] == 8


def test_bench_two_ranks_one_exchange_per_evaluation(gpu_device):
    """--xch-every 1: every evaluation is exchanged on its own (what a training loop pays), same merged value."""
    res = _run_bench(2, "k", extra=("--xch-every", "1"))
    assert res["evaluations_per_exchange"] == 1 and res["exchange_ms"] is not None and res["exchange_ms"] > 0.0
    assert res["check"]["all_steps_equal"] and res["check"]["rel_diff"] <= 2e-6, res["check"]


def test_bench_single_rank_line_has_the_contract_fields(gpu_device):
    res = _run_bench(1, "k", extra=("--median-iters", "50"))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "ms_per_step_median", "n_ranks_seen"):
        assert k in res, k
    assert res["n_ranks_seen"] == 1 and res["median_protocol"]["iters"] == 50
    assert res["check"]["rel_diff"] == 0.0                       # no sharding: the same evaluation twice
    r = res["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["limiter"] in ("mfma", "hbm", "latency/issue", "unknown") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] <= 1.0 and r["frac_fp32_equivalent"] >= r["frac"]       # the mix ceiling is never below the fp32-MFMA peak
    # round 4: the GEMM phases' own utilisation (north_star's "MFMA utilisation on the batched conditional GEMMs" as a number) and the
    # environment the line was measured in
    g = r["gemm_phase_mfma_util"]
    assert 0.0 < g["value"] <= 1.0 and len(g["per_layer"]) == 2 and all(0.0 < l["stage2"]["util"] <= 1.0 for l in g["per_layer"])
    assert res["environment"]["HIP_FORCE_DEV_KERNARG"] == os.environ.get("HIP_FORCE_DEV_KERNARG", "1")   # (an inherited value wins over the package's default)
    # round 5: the packed-operand stream every workgroup pulls out of L2 (informational; what bounds the M = 512 config)
    l2 = r["l2_operand_stream"]
    assert l2["workgroups"] * l2["bytes_per_workgroup"] == l2["bytes_per_launch"] and 0.0 < l2["TB_per_s"] < 20.0 and l2["samples_per_workgroup"] % 16 == 0
    # (the `no_dev_kernarg` leg -- the same loop in a fresh process without device-resident kernel arguments -- belongs to the full line only)


@pytest.mark.parametrize("shard", ["k", "n"])
def test_bench_exchange_path_on_a_real_rccl_communicator(gpu_device, shard):
    """The multi-GPU code path of bench.py -- per-slot hipGraphs, OverlappedExchange on a side stream, iwvi_lse_merge_steps /
    all-reduce -- on a 1-rank **RCCL** ("nccl") communicator (IWVI_BENCH_FORCE_XCH=1): what an 8-GPU launch runs, minus the peers."""
    port = _free_port()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               IWVI_BENCH_FORCE_XCH="1", IWVI_BENCH_BACKEND="nccl", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "50", "--warmup", "5", "--shard", shard, "--check",
           "--no-cpu-baseline", "--no-train-leg", "--median-iters", "0"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert "one exchange per replay" in res["config"]["launch"] and res["n_ranks_seen"] == 1
    assert res["check"]["all_steps_equal"] and res["check"]["rel_diff"] <= 2e-6, res["check"]
    assert np.isfinite(res["elbo"]) and res["value"] > 1e7      # (a sanity floor: > 10 M samples/s through the exchange path)


def _bare_env():
    """The caller's environment WITHOUT any launcher variable (and without the switches bench.py sets for its ranks)."""
    drop = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IWVI_BENCH_BACKEND",
            "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_FORCE_DEV_KERNARG", "IWVI_BENCH_LAUNCHED")
    return {k: v for k, v in os.environ.items() if k not in drop}


@pytest.mark.parametrize("shard", ["k", "n"])
def test_bench_gpus_2_starts_its_own_ranks(gpu_device, shard):
    """``python bench.py --gpus 2 ...`` with NO launcher and NO launcher environment starts its own two rank
    processes (before touching the GPU), sets dmabuf IPC + device kernel arguments for them, relays rank 0's single JSON line.
    Both ranks share cuda:0 here (one-GPU lease), hence --oversubscribe --backend gloo; on a node it is `python bench.py --gpus 8`."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--batch", "64", "--steps", "8", "--warmup", "2",
           "--shard", shard, "--check", "--no-cpu-baseline", "--no-train-leg", "--oversubscribe", "--backend", "gloo"]
    p = subprocess.run(cmd, env=_bare_env(), capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and p.stdout.strip() == lines[0], p.stdout          # stdout is exactly rank 0's line
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["n_ranks_seen"] == 2 and res["self_launched"] is True
    assert res["config"]["K_per_rank"] == [5, 5] and res["config"]["K_total"] == (10 if shard == "k" else 5)
    assert res["environment"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and res["environment"]["HIP_FORCE_DEV_KERNARG"] == "1"
    assert res["exchange_ms"] is not None and 0.0 < res["exchange_ms"] < 50.0 and res["evaluations_per_exchange"] == 8
    assert res["check"]["all_steps_equal"] and res["check"]["rel_diff"] <= 2e-6, res["check"]
    assert np.isfinite(res["value"]) and res["value"] > 0


def test_bench_refuses_more_ranks_than_devices(gpu_device):
    """--gpus N > device_count without the plumbing override is refused before any rank starts (one rank per GPU)."""
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "4", "--warmup", "1"],
                       env=_bare_env(), capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert p.returncode == 2 and "one rank per GPU" in p.stderr and not p.stdout.strip()


def test_bench_self_launch_reports_a_failing_rank(gpu_device):
    """A rank that dies takes the job down with a non-zero exit code instead of leaving its peers in a collective."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "1", "--batch", "64", "--steps", "4", "--warmup", "1",
           "--shard", "n", "--split-k", "--no-cpu-baseline", "--no-train-leg", "--oversubscribe", "--backend", "gloo"]   # (--split-k needs --shard k)
    p = subprocess.run(cmd, env=_bare_env(), capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode != 0 and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_overlapped_exchange_slot_reuse(gpu_device):
    """depth-2 staging ring, 7 submits of 3 evaluations each on a 1-rank RCCL communicator: every exchanged slot returns
    the ELBOs of exactly the evaluations written into it (no slot is overwritten while its exchange is still reading it)."""
    import torch.distributed as dist
    from dgps_with_iwvi_amd import sharding, synthetic
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu_device)
    try:
        spec = synthetic.make_spec(L=2, M=32, B=64, K=6, with_lv=True, seed=3, n_data=4096)
        model = synthetic.build_model(spec, gpu_device)
        rng = np.random.default_rng(0)
        for mode in ("k", "n"):
            xch = sharding.OverlappedExchange(mode, 1, 64, 6, 4096 / 64, gpu_device, depth=2, steps=3)
            want, got = [], []
            for it in range(7):
                slot = xch.before_step()
                assert slot == it % 2
                row = []
                for view in xch.slot_views(slot):
                    zs = [torch.as_tensor(z, dtype=torch.float32, device=gpu_device) for z in synthetic.make_noise(spec, seed=int(rng.integers(1 << 30)))]
                    if mode == "k":
                        _, glob = model.lse_partials(zs, K_total=6, out=view)
                    else:
                        model._build_likelihood(zs, out=view)
                        glob = None
                    row.append(zs)
                xch.submit(global_kls=glob)
                if it % 3 == 2:                                   # sometimes drain, sometimes let two exchanges be in flight
                    got.append(xch.finish().clone())
                    want.append(row)
            got.append(xch.finish().clone())
            want.append(row)
            for g, row in zip(got, want):
                ref = [model.compute_log_likelihood(zs) for zs in row]
                np.testing.assert_allclose(g.cpu().numpy(), ref, rtol=2e-6)
    finally:
        dist.destroy_process_group()


def test_bench_eight_ranks_split_k_at_the_configs3_stack(gpu_device):
    """BASELINE.json configs[3] -- 3-layer DGP, M=256, K=50, 'K-sharded across 8' -- as the driver will start it on an 8-GPU node, here as 8
    fresh rank processes over gloo on ONE GPU at a reduced batch: --split-k gives the ranks 7,7,6,6,6,6,6,6 of the job's 50 importance
    samples; the merged ELBO of the job-wide injected draw equals the single-rank evaluation of the whole job."""
    res = _run_bench(8, "k", extra=("--split-k", "--config", "3", "--batch", "64"), timeout=900)
    assert res["n_gpus"] == 8 and res["n_ranks_seen"] == 8 and res["scaling"] == "strong"
    assert res["config"]["K_total"] == 50 and res["config"]["K_per_rank"] == [7, 7, 6, 6, 6, 6, 6, 6]
    c = res["check"]
    assert c["K_total"] == 50 and c["B_total"] == 64 and c["all_steps_equal"]
    assert c["rel_diff"] <= 2e-6, c
    assert np.isfinite(res["value"]) and res["value"] > 0


def test_bench_eight_ranks_n_shard_at_the_configs3_stack(gpu_device):
    """The same stack N-sharded over 8 ranks (32 points each, all 50 samples): one scalar all-reduce per replay."""
    res = _run_bench(8, "n", extra=("--config", "3", "--batch", "32"), timeout=900)
    assert res["n_ranks_seen"] == 8 and res["scaling"] == "weak"
    c = res["check"]
    assert c["K_total"] == 50 and c["B_total"] == 256 and c["all_steps_equal"]
    assert c["rel_diff"] <= 2e-6, c


@pytest.mark.parametrize("rows", [(30, 18), (9, 8, 7, 6, 6, 5, 4, 3)])
def test_trainer_n_shard_ranks_built_from_local_rows(gpu_device, tmp_path, rows):
    """every rank constructs its model from its own rows (local num_data, models.py:18) and uneven minibatches
    (30 + 18 points).  Trainer(group=...) must scale the data term by the JOB's num_data and weigh the ranks by B_rank / B_job: the
    merged gradient of the two HIP ranks equals the single-process gradient of the whole minibatch on the same injected noise."""
    from dgps_with_iwvi_amd import backward, synthetic
    port = _free_port()
    out = str(tmp_path / "grad_rank%d.npz")
    worker = os.path.join(ROOT, "tests", "helpers", "nshard_trainer_worker.py")
    procs = []
    world = len(rows)                                             # 2 ranks, and the 8 of a full node (the gradient bucket of sharding.allreduce_gradients)
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, out, *[str(r) for r in rows]], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    for p in procs:
        so, se = p.communicate(timeout=600)
        assert p.returncode == 0, se[-3000:]
    spec = synthetic.make_spec(L=2, M=32, B=48, K=4, with_lv=True, seed=61, n_data=48)
    zs = synthetic.make_noise(spec, seed=62)
    model = synthetic.build_model(spec, gpu_device)
    elbo, ref = backward.iw_elbo_and_gradients(model, [torch.as_tensor(z, dtype=torch.float32, device=gpu_device) for z in zs])
    rs = [np.load(out % r) for r in range(world)]
    r0, r1 = rs[0], rs[-1]
    for r, n in zip(rs, rows):
        assert abs(float(r["weight"]) - n / 48) < 1e-12
        assert float(r["elbo"]) == float(r0["elbo"])             # every rank holds the same merged value
    assert abs(float(r0["elbo"]) - float(elbo)) <= 1e-5 * abs(float(elbo))
    for k, v in ref.items():
        a, b = r0[k], v.detach().double().cpu().numpy().reshape(r0[k].shape)
        assert np.array_equal(a, r1[k]), k
        assert np.abs(a - b).max() <= 2e-4 * max(np.abs(b).max(), 1e-6), (k, np.abs(a - b).max(), np.abs(b).max())


@pytest.mark.parametrize("shard", ["n", "k"])
def test_sharded_training_step_as_graph_segments(gpu_device, tmp_path, shard):
    """``Trainer(use_graph=True)`` with more than one rank captures a step as hipGraph SEGMENTS with the job's
    collectives between them (N-shard: 3 segments around the 2 gradient all-reduces; K-shard: 5 around 2 all-gathers of the [B, 2] pairs
    + 2 all-reduces).  Two gloo ranks on cuda:0: after five steps the parameters equal the eager sharded trainer's BIT FOR BIT (the same
    kernels in the same order on the same device-drawn noise), on both ranks, and both ranks hold the same parameters."""
    port = _free_port()
    out = str(tmp_path / "seg_rank%d.npz")
    worker = os.path.join(ROOT, "tests", "helpers", "sharded_graph_trainer_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, worker, out, shard, "5"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT))
    for p in procs:
        so, se = p.communicate(timeout=900)
        assert p.returncode == 0, se[-3000:]
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert int(r0["n_collectives"]) == (2 if shard == "n" else 4) and int(r0["n_graphs"]) == int(r0["n_collectives"]) + 1
    names = [k[len("eager."):] for k in r0.files if k.startswith("eager.") and k != "eager.ms"]
    assert "q_mu" in names and "q_sqrt" in names and len(names) > 8
    for r in (r0, r1):
        for n in names:
            assert np.array_equal(r["eager." + n], r["graph." + n]), (shard, n)
        assert np.all(np.isfinite(r["eager.elbo"]))
    for n in names:
        if n != "elbo":
            assert np.array_equal(r0["graph." + n], r1["graph." + n]), (shard, n)       # the ranks apply the same update
    print("sharded step, 2 gloo ranks on one GPU (%s-shard): eager %.3f ms, graph segments %.3f ms" % (shard, float(r0["eager.ms"]), float(r0["graph.ms"])))
    assert float(r0["graph.ms"]) <= float(r0["eager.ms"]) * 1.05          # never slower than the eager step it replaces


@pytest.mark.parametrize("shard", ["n", "k"])
def test_bench_two_ranks_report_the_sharded_training_step(gpu_device, shard):
    """``bench.py --gpus N`` reports ``training_step`` of the sharded job -- one exchange per op's evaluation -- eagerly and
    as hipGraph segments; the graph form is never slower than the eager step it replaces."""
    res = _run_bench(2, shard, train_leg=True, timeout=900)
    t = res["training_step"]
    assert "error" not in t, t
    assert t["collectives_per_step"] == (2 if shard == "n" else 4) and t["graph_segments"] == t["collectives_per_step"] + 1
    # (two gloo ranks sharing ONE GPU: every collective is a host round trip that idles the device, and the two processes' kernels interleave
    #  -- the ratio of the two forms moves by +-15 % between runs here; on this plumbing set-up only "same ballpark" is asserted.  The worker
    #  test above, with a model whose step is launch-bound, asserts graph <= eager)
    assert 0.0 < t["train_step_ms"] <= t["train_step_eager_ms"] * 1.3, t


@pytest.mark.parametrize("shard", ["n", "k"])
def test_graph_training_step_on_a_real_rccl_communicator(gpu_device, shard):
    """The graph form of a sharded training step on a real RCCL ("nccl") communicator -- one rank, the step's all-reduces / all-gathers
    forced on.  RCCL collectives are capturable: they are recorded INSIDE the step's graph (one graph, no cut; issued between two captures
    their work objects are polled by the process group's watchdog while the next capture is open -- hipErrorStreamCaptureUnsupported, which
    is what this test first found).  Three steps, parameters and bounds bit-identical to the eager step."""
    port = _free_port()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    worker = os.path.join(ROOT, "tests", "helpers", "rccl_one_rank_segments_worker.py")
    p = subprocess.run([sys.executable, worker, shard, "3"], env=env, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert res["same"] and res["finite"], res
    assert res["n_collectives"] == 0 and res["n_graphs"] == 1, res           # (the collectives are nodes of the one graph)
