"""bench.py --gpus N without a launcher, the parts that need no GPU: the rank processes are started with the
torch.distributed.run environment contract, rendezvous over gloo on 127.0.0.1, rank 0's line is relayed alone on stdout, a refused or
failing launch ends with a non-zero exit code.  The full path (two ranks evaluating on cuda:0) is tests/test_gpu_multirank.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DROP = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "IWVI_BENCH_BACKEND",
        "HSA_ENABLE_IPC_MODE_LEGACY", "HIP_FORCE_DEV_KERNARG", "IWVI_BENCH_LAUNCHED")


def _run(*argv, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in DROP}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, cwd=ROOT, timeout=timeout)


def test_self_launch_rendezvous_three_ranks():
    p = _run("--gpus", "3", "--rendezvous-only")
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout.count("\n") == 1                              # rank 0's line and nothing else on stdout
    res = json.loads(p.stdout)
    assert res["n_gpus"] == 3 and res["n_ranks_seen"] == 3 and res["rank_sum"] == 3 and res["self_launched"] is True
    assert res["environment"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "HIP_FORCE_DEV_KERNARG": "1", "MASTER_ADDR": "127.0.0.1", "LOCAL_WORLD_SIZE": "3"}


def test_self_launch_is_refused_beyond_the_device_count():
    import torch
    n = torch.cuda.device_count() + 2
    p = _run("--gpus", str(n))
    assert p.returncode == 2 and "one rank per GPU" in p.stderr and not p.stdout.strip()
    p = _run("--gpus", str(n), "--oversubscribe")                 # RCCL cannot share a device between ranks
    assert p.returncode == 2 and "gloo" in p.stderr


def test_a_dying_rank_ends_the_job_with_a_nonzero_code():
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered on the GPU by tests/test_gpu_multirank.py::test_bench_self_launch_reports_a_failing_rank")
    p = _run("--gpus", "2", "--oversubscribe", "--backend", "gloo", "--steps", "2", "--warmup", "1")   # no device here: both ranks exit 1
    assert p.returncode != 0 and "stopping the other ranks" in p.stderr and not p.stdout.strip()


def test_device_count_comes_from_the_kfd_topology(tmp_path, monkeypatch):
    """The launcher parent counts GPUs without a HIP / torch.cuda call: KFD nodes with simd_count > 0, narrowed by *_VISIBLE_DEVICES."""
    sys.path.insert(0, ROOT)
    import bench
    for i, simd in enumerate((0, 0, 1024, 1024, 1024)):             # two CPU nodes, three GPUs
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(v, raising=False)
    assert bench.visible_gpu_count(str(tmp_path)) == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,2")
    assert bench.visible_gpu_count(str(tmp_path)) == 2
    import torch
    assert bench.visible_gpu_count(str(tmp_path / "absent")) == torch.cuda.device_count()      # unreadable topology: the fallback
