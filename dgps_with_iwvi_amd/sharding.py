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
overlaps the next evaluations' kernels; a forward-only loop may put several evaluations into one collective.
"""
import ctypes

import torch
import torch.distributed as dist

from . import _abi


# While ``training.Trainer`` records a sharded step as hipGraph SEGMENTS, every collective of the step is a cut between two segments
# (``run_collective``): a collective is host-driven for gloo and library-driven for RCCL, neither belongs inside a captured graph here.
_RECORDER = None
# test knob: issue the collectives of a training step even on a ONE-rank group (a 1-rank RCCL communicator is the only real "nccl" backend a
# single-GPU box offers: tests/test_gpu_multirank.py runs the segmented step against it)
_FORCE_ONE_RANK_COLLECTIVES = False


def run_collective(fn, group=None):
    """A collective of a training step (``fn()`` issues it on the current stream, on tensors that live as long as the step's graphs do):
    run now.  While ``training.Trainer`` records a step: an RCCL ("nccl") collective is CAPTURED like any other stream work -- the step stays
    one graph -- (it is capturable, and issued between two captures its work object would be polled by the process group's watchdog thread
    while the next capture is open: hipErrorStreamCaptureUnsupported, seen on a 1-rank communicator); a host-driven one (gloo: the CPU tests
    and one-GPU plumbing runs) is remembered as the cut between two graph segments and replayed in place."""
    if _RECORDER is None or dist.get_backend(group) == "nccl":
        fn()
    else:
        _RECORDER.cut(fn)


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
    """Exchange of ``steps`` consecutive, independent evaluations in ONE collective on a side stream, overlapping
    the next evaluations' kernels (GPU only).

    The evaluations write their results straight into a staging slot (``slot_views(slot)[e]``: a [B, 2] buffer of
    (max, sumexp) pairs for the K-shard, a 1-element float64 for the N-shard -- pass it as ``out=`` to
    ``lse_partials`` / ``_build_likelihood``, typically inside a captured graph, one graph per slot).  ``submit``
    then runs, on the side stream, one all-gather of the whole slot [steps, B, 2] + ``iwvi_lse_merge_steps``
    (K-shard) or one all-reduce of [steps] (N-shard).  The messages are a few KiB to a few hundred KiB and
    latency-bound, so one collective per ``steps`` evaluations costs about what one per evaluation does.
    A forward-only evaluation loop can batch like this because evaluation e+1 does not consume evaluation e's
    result; a training loop would exchange every step (``steps=1``).
    Before a slot is overwritten the main stream waits for the exchange that last read it (``before_step``).
    Everything the per-call path touches (buffers, events, the merge's argument block) is created once."""

    def __init__(self, mode, world, B, K_total, scale, device, depth=2, group=None, steps=1, timed=False):
        assert mode in ("k", "n")
        self.mode, self.world, self.B, self.K_total, self.scale, self.group = mode, world, B, K_total, float(scale), group
        self.steps = steps
        self.comm = torch.cuda.Stream(device=device)
        self.depth, self.i = depth, 0
        shape = (steps, B, 2) if mode == "k" else (steps,)
        dtype = torch.float32 if mode == "k" else torch.float64
        self.stage = [torch.zeros(shape, dtype=dtype, device=device) for _ in range(depth)]
        self.ready = [torch.cuda.Event() for _ in range(depth)]
        self.done = [torch.cuda.Event() for _ in range(depth)]
        self.used = [False] * depth
        self.gathered = torch.empty((world,) + shape, dtype=dtype, device=device) if mode == "k" else None
        self.result = torch.zeros(steps, dtype=torch.float64, device=device)     # the last exchanged slot's ELBOs
        self._glob_key, self._glob_args = None, None
        # timed: events around each exchange on the communication stream (collective + merge), read back by exchange_ms()
        self.timed = bool(timed)
        self._t = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(depth)] if timed else None
        self._t_ms = []
        self._t_open = [False] * depth                           # slot holds a recorded pair of events nobody has read yet

    def slot(self):
        return self.i % self.depth

    def slot_views(self, slot):
        """Per-evaluation output buffers of a slot: [B, 2] float32 views (K-shard) / 1-element float64 views."""
        st = self.stage[slot]
        return [st[e] if self.mode == "k" else st[e:e + 1] for e in range(self.steps)]

    def _merge_args(self, global_kls):
        key = tuple(int(g.data_ptr()) for g in global_kls)
        if key != self._glob_key:
            glob = [_abi.dev_tensor(g.reshape(-1), "global kl", torch.float64) for g in global_kls]
            self._glob_keep = glob
            self._glob_args = (_abi.ptr_array(glob), (ctypes.c_int32 * max(len(glob), 1))(*[g.numel() for g in glob]), len(glob))
            self._glob_key = key
        return self._glob_args

    def before_step(self):
        """Main stream: make sure the slot about to be overwritten is no longer being read; returns the slot."""
        slot = self.slot()
        if self.used[slot]:
            torch.cuda.current_stream().wait_event(self.done[slot])
        return slot

    def submit(self, global_kls=None):
        """Exchange the slot the evaluations just issued on the current stream have filled."""
        slot = self.slot()
        self.i += 1
        main = torch.cuda.current_stream()
        self.ready[slot].record(main)
        if self.timed and self.used[slot]:                       # this slot's previous exchange: before_step only QUEUED a wait for it on the
            self._collect(slot)                                  # device, so it may still be running -- its sample is then dropped (counted)
        self.used[slot] = True
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(self.ready[slot])
            if self.timed:
                self._t[slot][0].record(self.comm)
            if self.mode == "k":
                dist.all_gather_into_tensor(self.gathered.view(-1), self.stage[slot].view(-1), group=self.group)
                arr, counts, n = self._merge_args(global_kls)
                _abi.check(_abi.lib().iwvi_lse_merge_steps(_abi.ptr(self.gathered), self.world, self.steps, self.B, self.K_total,
                                                           arr, counts, n, self.scale, None, _abi.ptr(self.result), _abi.stream_ptr()))
            else:
                dist.all_reduce(self.stage[slot], op=dist.ReduceOp.SUM, group=self.group)
                self.result.copy_(self.stage[slot], non_blocking=True)     # sums over ranks; finish() divides
            if self.timed:
                self._t[slot][1].record(self.comm)
                self._t_open[slot] = True
            self.done[slot].record(self.comm)

    def _collect(self, slot):
        """Read a slot's pair of events ONCE (a second call before the slot is re-recorded adds nothing: no duplicate samples)."""
        if not self._t_open[slot]:
            return
        a, b = self._t[slot]
        if b.query():
            self._t_ms.append(a.elapsed_time(b))
        else:                                                    # still running and about to be re-recorded: this exchange is not sampled
            self.t_dropped = getattr(self, "t_dropped", 0) + 1
        self._t_open[slot] = False

    def exchange_ms(self):
        """Median side-stream time of one exchange (collective + merge kernel) over the timed submits so far; None if not timed."""
        if not self.timed:
            return None
        torch.cuda.current_stream().wait_stream(self.comm)
        torch.cuda.synchronize()
        for slot in range(self.depth):
            if self.used[slot]:
                self._collect(slot)
        if not self._t_ms:
            return None
        v = sorted(self._t_ms)
        return float(v[len(v) // 2])

    def finish(self):
        """Wait for the exchanges; returns the last slot's per-evaluation ELBOs [steps]."""
        torch.cuda.current_stream().wait_stream(self.comm)
        return self.result / self.world if self.mode == "n" else self.result


def allreduce_gradients(grads, weight=None, group=None):
    """Data-parallel (N-shard) training: every rank holds d ELBO_r / d theta of its own points, with
    ELBO_r = (num_data / B_r) * sum_{n in rank} (...) - KL (models.py:144-150 on the local minibatch).  The job's
    gradient is the B_r / B weighted mean (the KL terms then count once): ONE all-reduce of one flat bucket
    (parameters are <= R*M^2 floats per layer, a few MiB in all -- a single ring pass over xGMI), then views back.
    ``weight`` = B_r / B (default 1 / world: equal local batches).  Returns the same dict, reduced in place."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not _FORCE_ONE_RANK_COLLECTIVES):
        return grads
    world = dist.get_world_size(group)
    w = 1.0 / world if weight is None else float(weight)
    names = sorted(grads)
    flat = [grads[k].reshape(-1) for k in names]
    dtype = torch.float64 if any(t.dtype == torch.float64 for t in flat) else flat[0].dtype
    bucket = torch.cat([t.to(dtype) for t in flat]) * w
    run_collective(lambda: dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=group), group)
    o = 0
    for k, t in zip(names, flat):
        n = t.numel()
        grads[k] = bucket[o:o + n].to(grads[k].dtype).reshape(grads[k].shape)
        o += n
    return grads


def resolve_n_shard(model, group=None, num_data_total=None):
    """N-shard bookkeeping, once, at trainer construction.  The reference scales the data term by ``num_data / B``
    with ``num_data = X.shape[0]`` (models.py:18,80-81); a rank built from ITS OWN rows of the data set therefore holds
    a local ``num_data`` and would weigh the likelihood 1 / world too lightly against the KL terms after the
    1 / world gradient average.  This sets ``model.num_data`` to the job total -- ``num_data_total`` if given, else the
    all-reduced sum of the ranks' row counts when ``model.num_data`` still is the constructor's default (the local row
    count), else the value the caller set, which must then agree on every rank -- and returns
    ``(num_data_total, weight)`` with weight = B_rank / B_job for ``allreduce_gradients``."""
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rows = int(model._X_all.shape[0])
    B = int(model.X.shape[0])
    if world == 1:
        if num_data_total is not None:
            model.num_data = int(num_data_total)
        return int(model.num_data), 1.0
    dev = model._X_all.device if dist.get_backend(group) != "gloo" else torch.device("cpu")
    v = torch.tensor([rows, B, int(model.num_data), -int(model.num_data)], dtype=torch.int64, device=dev)
    tot = v.clone()
    dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    mx = v.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    if num_data_total is None:
        if int(model.num_data) == rows:                       # untouched default: local rows -> the job's rows
            num_data_total = int(tot[0])
        else:                                                 # set by the caller: it has to be the same number everywhere
            if int(mx[2]) != -int(mx[3]):
                raise ValueError("model.num_data differs between ranks (%d..%d): pass num_data_total" % (-int(mx[3]), int(mx[2])))
            num_data_total = int(model.num_data)
    model.num_data = int(num_data_total)
    return int(num_data_total), float(B) / float(int(tot[1]))


def lse_from_pairs(ms_all):
    """Gathered (max, sum exp) pairs [G, B, 2] -> logsumexp over all the job's samples [B] (not yet minus log K)."""
    m = ms_all[..., 0].max(0).values
    return m + torch.log((ms_all[..., 1] * torch.exp(ms_all[..., 0] - m)).sum(0))


def k_shard_gradients(model, zs=None, K_total=None, group=None, wrt="all"):
    """(Every rank must draw its own noise: give each a different ``settings.set_seed`` -- the kernels key their Philox
    streams by (seed, step, layer, sample, quad), so equal seeds would duplicate the samples across ranks.)

    K-sharded training step input: every rank holds all B points and its own K_r of the job's K_total importance
    samples.  One all-gather of the [B, 2] pairs (the exchange of the forward path) gives the job's logsumexp per
    point; each rank's adjoint then runs with its share of the softmax weights, and ONE all-reduce (sum) of the flat
    gradient bucket finishes d ELBO / d theta -- the KL terms enter with 1 / world per rank.  -> (elbo, grads)."""
    from .backward import iw_elbo_and_gradients
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    K_total = int(K_total or model.num_samples * world)

    def exchange(ms):
        if world == 1 and not (_FORCE_ONE_RANK_COLLECTIVES and dist.is_available() and dist.is_initialized()):
            return lse_from_pairs(ms[None])
        gathered = torch.empty((world,) + tuple(ms.shape), dtype=ms.dtype, device=ms.device)
        src = ms.contiguous().view(-1)
        run_collective(lambda: dist.all_gather_into_tensor(gathered.view(-1), src, group=group), group)
        return lse_from_pairs(gathered)

    elbo, g = iw_elbo_and_gradients(model, zs, exchange=exchange, K_total=K_total, kl_weight=1.0 / world, wrt=wrt)
    g = allreduce_gradients(g, weight=1.0, group=group)             # a sum: the shares add up
    return elbo, g
