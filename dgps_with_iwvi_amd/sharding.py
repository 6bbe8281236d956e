"""Multi-GPU sharding of the (K importance samples x minibatch) batch: one process per GPU,
``torch.distributed`` ("nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-device; this is the scale-out of its one data-parallel axis.
Parameters are replicated and every rank recomputes the (tiny) K_uu factorisation.  Two shardings
(SURVEY.md section 8 row E), both with a single small exchange per ELBO evaluation:

* **K-shard** (the one BASELINE.json names): every rank holds the same B points and its own K_r importance
  samples, sum K_r = K.  Exchange: all-gather of the per-point ``(max_k L, sum_k exp(L - max))`` pairs
  ``[B, 2]`` (8 KiB at B = 1024), then ``iwvi_lse_merge`` gives logsumexp over all K and the ELBO.
* **N-shard**: every rank owns B different points and all K samples; log-sum-exp is local, the exchange is one
  scalar all-reduce of the per-rank ELBO estimates (their mean: the KL terms are identical on every rank).

Messages are latency-bound (a few KiB), so the exchange runs on a side stream from a small staging ring and
overlaps the next evaluation's kernels.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _abi


def split_samples(K_total, world):
    """Importance samples per rank: ceil(K/G) for the first K % G ranks, floor(K/G) for the rest."""
    if K_total < world:
        raise ValueError("K=%d importance samples cannot be split over %d ranks" % (K_total, world))
    q, r = divmod(K_total, world)
    return [q + (1 if i < r else 0) for i in range(world)]


def split_points(n_points, world):
    """Contiguous point ranges [(lo, hi)] per rank, sizes differing by at most one."""
    q, r = divmod(n_points, world)
    out, lo = [], 0
    for i in range(world):
        hi = lo + q + (1 if i < r else 0)
        out.append((lo, hi))
        lo = hi
    return out


def merge_lse_reference(ms_all, K_total):
    """Merge gathered (max, sumexp) pairs [G, B, 2] -> logp [B] (torch ops; used on CPU tensors by the gloo tests,
    the GPU path is the ``iwvi_lse_merge`` kernel)."""
    m = ms_all[..., 0].max(0).values
    s = (ms_all[..., 1] * torch.exp(ms_all[..., 0] - m)).sum(0)
    return m + torch.log(s) - torch.log(torch.tensor(float(K_total), dtype=ms_all.dtype))


def merge_lse(ms_all, K_total, global_kls, scale):
    """Gathered [G, B, 2] -> (logp [B], elbo 0-dim float64) through ``iwvi_lse_merge`` (GPU)."""
    G, B, _ = ms_all.shape
    ms_all = _abi.dev_tensor(ms_all.contiguous(), "gathered lse pairs")
    glob = [_abi.dev_tensor(g.reshape(-1), "global kl", torch.float64) for g in global_kls]
    glob_n = (ctypes.c_int32 * max(len(glob), 1))(*[g.numel() for g in glob])
    logp = torch.empty(B, dtype=ms_all.dtype, device=ms_all.device)
    elbo = torch.empty(1, dtype=torch.float64, device=ms_all.device)
    _abi.check(_abi.lib().iwvi_lse_merge(_abi.ptr(ms_all), G, B, K_total, _abi.ptr_array(glob), glob_n, len(glob),
                                         float(scale), _abi.ptr(logp), _abi.ptr(elbo), _abi.stream_ptr()))
    return logp, elbo[0]


def k_shard_elbo(ms, global_kl_sum, K_total, scale, group=None, merge=None):
    """One K-sharded evaluation, synchronous form (tests, small jobs): ms [B, 2] local pairs -> global ELBO.
    ``merge(ms_all, K_total) -> logp`` defaults to the torch reference merge."""
    world = dist.get_world_size(group)
    gathered = torch.empty((world,) + tuple(ms.shape), dtype=ms.dtype, device=ms.device)
    dist.all_gather_into_tensor(gathered.view(-1), ms.contiguous().view(-1), group=group)
    logp = (merge or merge_lse_reference)(gathered, K_total)
    return logp.double().sum() * scale - global_kl_sum


def n_shard_elbo(local_elbo, group=None):
    """One N-sharded evaluation: the mean over ranks of the per-rank ELBO estimates (each already scaled by
    num_data / B and carrying the same global KL)."""
    world = dist.get_world_size(group)
    v = local_elbo.detach().clone().reshape(1)
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    return v[0] / world


class OverlappedExchange:
    """Per-step exchange on a side stream, overlapping the next step's kernels (GPU only).

    The step's output lives in a buffer that the next (graph-replayed) step overwrites, so the main stream first
    copies it into one of ``depth`` staging slots; the side stream then runs the collective (+ merge) from that
    slot.  Before a slot is reused the main stream waits for the exchange that last read it."""

    def __init__(self, mode, world, B, K_total, scale, device, depth=2, group=None):
        assert mode in ("k", "n")
        self.mode, self.world, self.B, self.K_total, self.scale, self.group = mode, world, B, K_total, scale, group
        self.comm = torch.cuda.Stream(device=device)
        self.depth, self.i = depth, 0
        shape = (B, 2) if mode == "k" else (1,)
        dtype = torch.float32 if mode == "k" else torch.float64
        self.stage = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.done = [None] * depth
        self.gathered = torch.empty((world,) + shape, dtype=dtype, device=device) if mode == "k" else None
        self.result = torch.zeros(1, dtype=torch.float64, device=device)

    def submit(self, out, global_kls=None):
        """out: ms [B, 2] (K-shard) or the local ELBO (N-shard), produced on the current stream."""
        slot = self.i % self.depth
        self.i += 1
        main = torch.cuda.current_stream()
        if self.done[slot] is not None:
            main.wait_event(self.done[slot])
        self.stage[slot].copy_(out.reshape(self.stage[slot].shape), non_blocking=True)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            if self.mode == "k":
                dist.all_gather_into_tensor(self.gathered.view(-1), self.stage[slot].view(-1), group=self.group)
                _, elbo = merge_lse(self.gathered, self.K_total, global_kls, self.scale)
                self.result.copy_(elbo.reshape(1))
            else:
                dist.all_reduce(self.stage[slot], op=dist.ReduceOp.SUM, group=self.group)
                self.result.copy_(self.stage[slot] / self.world)
            ev = torch.cuda.Event()
            ev.record(self.comm)
            self.done[slot] = ev

    def finish(self):
        torch.cuda.current_stream().wait_stream(self.comm)
        return self.result
