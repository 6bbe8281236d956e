"""The reference's training step (experiments/build_models.py:270-304) on the GPU: per step
``op_increment`` -> ``op_ng`` (GPflow NatGradOptimizer on the final layer's (q_mu, q_sqrt), gamma with a staircase
exponential decay) -> ``op_adam`` (TensorFlow AdamOptimizer on every other trainable parameter, lr with the same
decay).  Each op evaluates the IW-ELBO and its gradient on its own fresh samples, like two ``session.run`` calls.

Trainable set as built by ``build_models.py`` with its defaults (``fix_linear=True``): inner layers Z, lengthscales,
q_mu, q_sqrt (kernel variance fixed, :213; W and the linear mean function's A too unless ``fix_linear=False``, :224-227); final layer Z, lengthscales,
kernel variance; encoders; the likelihood variance.  Gradients: ``backward.iw_elbo_and_gradients``; update rules:
``iwvi_natgrad_step`` / ``iwvi_adam_step`` (csrc/backward.hip)."""
import ctypes

import numpy as np
import torch

from . import _abi, settings
from .backward import iw_elbo_and_gradients
from .layers import GPLayer, LatentVariableLayer
from .temp_workaround import SharedMixedMok


def staircase_decay(base, step, rate, every=1000):
    """tf.train.exponential_decay(base, step, every, rate, staircase=True)."""
    return base * rate ** (step // every)


class _SegmentedGraph:
    """A training step (or one op of it) of a SHARDED job as hipGraph segments with the job's collectives between them.  Recording: the op
    runs once under capture; at every collective (``sharding.run_collective``) the running capture ends, the collective is issued eagerly
    (on whatever the dry capture left in its buffers: the ranks record at the same step, so the calls match up) and remembered, and a new
    capture begins in the same memory pool.  ``replay``: the segments and the collectives in their recorded order on the current stream."""

    def __init__(self):
        self.items, self._g, self._pool, self.n_graphs, self.n_collectives = [], None, None, 0, 0

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        if self._pool is None:
            self._pool = torch.cuda.graph_pool_handle()          # one memory pool for all segments: a buffer filled in one is read in the next
        self._g = g                                              # (set first: a failing capture_begin is ended by record()'s handler)
        g.capture_begin(pool=self._pool, capture_error_mode="thread_local")

    def _end(self):
        self._g.capture_end()
        self.items.append(self._g)
        self.n_graphs += 1
        self._g = None

    def cut(self, fn):
        self._end()
        fn()
        self.items.append(fn)
        self.n_collectives += 1
        self._begin()

    def record(self, op):
        from . import sharding
        if sharding._RECORDER is not None:
            raise RuntimeError("a segmented capture is already recording")
        sharding._RECORDER = self
        try:
            self._begin()
            try:
                out = op()
            except BaseException:
                if self._g is not None:
                    try:
                        self._g.capture_end()
                    except Exception:
                        pass
                raise
            self._end()
        finally:
            sharding._RECORDER = None
        return out

    def replay(self):
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            else:
                # (the collectives that end up here are host-driven -- gloo -- and wait for the stream anyway; the explicit wait orders them
                #  behind the segment by the host instead of by an event recorded right behind a multi-queue graph launch)
                torch.cuda.current_stream().synchronize()
                it()


class Trainer:
    def __init__(self, model, lr=5e-3, gamma=1e-2, lr_decay=0.98, gamma_decay=0.98, fix_linear=True,
                 beta1=0.9, beta2=0.999, epsilon=1e-8, group=None, shard_weight=None, shard="n", num_data_total=None,
                 use_graph=False, check_finite=True, check_every=100, autotune_f64=True, K_total=None, one_factorisation=True):
        """``group`` / ``shard_weight``: data-parallel training over the ranks of a torch.distributed group (each rank's
        model holds its own minibatch rows, N-shard): gradients are merged by ``sharding.allreduce_gradients`` with
        weight B_rank / B_job (default: from the all-reduced local batch sizes) before either update, so every rank
        applies the same step.  Give every rank its own rows of the data set (the minibatch iterator is seeded identically
        everywhere, models.py:25-26); the data term must still be scaled by the JOB's ``num_data`` (models.py:18,80-81), so
        ``model.num_data`` is set to ``num_data_total`` -- by default the sum of the ranks' row counts
        (``sharding.resolve_n_shard``).
        ``shard="k"``: every rank holds all the points and its own share of the importance samples instead
        (``sharding.k_shard_gradients``: one all-gather of the per-point pairs + one gradient all-reduce).

        The two trained host scalars of the reference -- the final layer's kernel variance and the likelihood variance -- live
        in 1-element device tensors that Adam updates in place and every kernel reads when it runs (``variance_dev`` of the
        descriptors); the model's host copies are refreshed lazily when read.  ``use_graph=True`` (noise drawn on the
        device): each of the two ops of a step is captured once into a hipGraph and replayed (as ONE graph when the data is not minibatched) -- no host work per launch, no
        device-to-host copy per step; the graphs are re-captured when the staircase decay changes lr / gamma.  With more than one rank the
        step is captured as graph SEGMENTS with the job's collectives between them (``_SegmentedGraph``: N-shard: [value + gradient, bucket]
        | all-reduce | [update; next op's value + gradient, bucket] | all-reduce | [update]; K-shard: the all-gather of the [B, 2] pairs cuts
        each op once more) -- the same kernels in the same order as the eager sharded step, so the parameters agree bit for bit.  ``check_finite``:
        one small D2H that raises when the bound or the final layer's q(u) went non-finite (the reference's Cholesky raises) -- every
        step in eager mode, every ``check_every`` steps in graph mode.

        ``autotune_f64`` (default on, while ``settings.f64_stage1 == "auto"``): ``model.autotune_f64()`` -- the float64 stage-1 route per layer
        from the measured diag(Lm) ratio of the CURRENT parameters -- runs here and again at every staircase epoch (1000 steps: the graphs
        are re-captured there anyway), so that a layer whose K_uu becomes ill-conditioned DURING training (lengthscales grow, inducing
        inputs cluster) leaves the float32 solve; captured graphs are dropped whenever a layer's route moves (``model.route_key()``).

        ``one_factorisation`` (default on; single rank, N-layout): a step factorises every K_uu ONCE.  The natural-gradient op moves
        nothing but the final layer's q(u) (build_models.py:288-295), so the Adam op that follows it in ``step`` re-packs that layer's q(u)
        images (IWVI_GP_REUSE_FACTOR, ~5 us) instead of factorising every layer again, and the dense float64 factors its adjoints read
        are formed beside the natural-gradient update of the first op, whose kernels leave the chip idle (``backward.prefactor_dense``).
        Same kernels on the same numbers: the parameters after a step are bit-identical to ``one_factorisation=False``.  The ops called
        on their own (``natgrad_op`` / ``adam_op``) always factorise."""
        self.model = model
        self.one_factorisation = bool(one_factorisation)
        self.K_total = None if K_total is None else int(K_total)   # K-shard with an uneven split of the job's samples (default: num_samples x ranks)
        self.autotune = bool(autotune_f64)
        self._tuned_epoch = None
        self.route_reports = []                                # [(global step, autotune_f64's report)]
        self.group, self.shard_weight, self.shard = group, shard_weight, shard
        import torch.distributed as dist
        # group=None means the DEFAULT group once torch.distributed is up: what counts is the number of ranks
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if shard == "n" and (self.world > 1 or group is not None or num_data_total is not None):
            from .sharding import resolve_n_shard
            _, w = resolve_n_shard(model, group, num_data_total)
            if self.shard_weight is None:
                self.shard_weight = w
        elif num_data_total is not None:
            model.num_data = int(num_data_total)
        self.lr, self.gamma, self.lr_decay, self.gamma_decay = lr, gamma, lr_decay, gamma_decay
        self.betas, self.epsilon = (beta1, beta2), epsilon
        self.global_step = 0
        self.adam_t = 0
        self.use_graph, self.check_finite, self.check_every = bool(use_graph), bool(check_finite), max(1, int(check_every))
        self._graphs = {}                                      # op name -> (decay epoch, CUDAGraph, elbo tensor)
        dev = model.X.device
        self._t_dev = torch.zeros(1, dtype=torch.int64, device=dev)     # Adam's step count, on the device
        ft = settings.float_type
        self.final = model.layers[-1]
        if not isinstance(self.final, GPLayer):
            raise ValueError("the last layer must be a GPLayer")
        # (gradient name, get tensor, transform); host scalars get a 1-element device master that is copied back
        self._entries = []
        self._scalars = []                                     # (device tensor, setter)
        n = len(model.layers)
        for i, l in enumerate(model.layers):
            if isinstance(l, LatentVariableLayer):
                for j in range(len(l.encoder.Ws)):
                    self._entries.append(("l%d.encW%d" % (i, j), l.encoder.Ws[j], 0))
                    self._entries.append(("l%d.encb%d" % (i, j), l.encoder.bs[j], 0))
                continue
            k = l._base_kern()
            self._entries.append(("l%d.Z" % i, l._Z(), 0))
            self._entries.append(("l%d.ls" % i, k.lengthscales, 1))
            if i == n - 1:
                t = torch.full((1,), k.variance, dtype=ft, device=dev)
                self._entries.append(("l%d.var" % i, t, 1))
                k.bind_device_variance(t)                      # kernels read it on the device from now on
                self._scalars.append((t, k))
            else:
                self._entries.append(("l%d.q_mu" % i, l.q_mu, 0))
                self._entries.append(("l%d.q_sqrt" % i, l.q_sqrt, 0))
            if not fix_linear:                                 # build_models.py:224-227
                if isinstance(l.kern, SharedMixedMok):
                    self._entries.append(("l%d.W" % i, l.kern.W, 0))
                if l.mean_function.mf_type == _abi.MF_LINEAR:
                    self._entries.append(("l%d.mfA" % i, l.mean_function.A, 0))
        t = torch.full((1,), model.likelihood.variance, dtype=ft, device=dev)
        self._entries.append(("lik_var", t, 1))
        model.likelihood.bind_device_variance(t)
        self._scalars.append((t, model.likelihood))
        for name, t, _ in self._entries:
            _abi.dev_tensor(t, name)
        self._state = [tuple(torch.empty_like(t) for _ in range(3)) for _, t, _ in self._entries]
        self._adam_call(None, init=True)
        self._autotune_if_due()
        M = self.final.num_inducing
        # (sized for THIS layer's R: the spread route of the step needs ~340 KB per latent GP at M = 128; include/iwvi_hip.h, ABI 16)
        self._ng_ws = torch.empty(_abi.lib().iwvi_natgrad_ws_bytes_ex(M, self.final.num_outputs), dtype=torch.uint8, device=dev)

    def _autotune_if_due(self):
        """Once per staircase epoch (and at construction): the measured route of every GP layer; a moved route invalidates the graphs."""
        epoch = self.global_step // 1000
        if not self.autotune or settings.f64_stage1 != "auto" or self._tuned_epoch == epoch:
            return
        self._tuned_epoch = epoch
        self.route_reports.append((self.global_step, self.model.autotune_f64()))

    def _adam_call(self, grads, init=False, lr=0.0):
        n = len(self._entries)
        arr = (_abi.AdamTensor * n)()
        keep = []
        for i, ((name, p, tr), (x, m, v)) in enumerate(zip(self._entries, self._state)):
            a = arr[i]
            a.param, a.x, a.m, a.v, a.n, a.transform = p.data_ptr(), x.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), tr
            if not init:
                g = grads[name].reshape(-1)
                if g.dtype == torch.float64:                     # (the head kernel's float64 sums: read as they are, no conversion launch)
                    a.transform = tr | _abi.ADAM_GRAD_F64
                else:
                    g = g.to(settings.float_type)
                g = _abi.dev_tensor(g.contiguous(), "grad " + name, g.dtype)
                if g.numel() != p.numel():
                    raise ValueError("gradient %s has %d entries, parameter has %d" % (name, g.numel(), p.numel()))
                keep.append(g)
                a.grad = g.data_ptr()
        if init:
            _abi.check(_abi.lib().iwvi_adam_step(arr, n, lr, self.betas[0], self.betas[1], self.epsilon, 1, 1, 1, _abi.stream_ptr()))
        else:                                                  # the step count lives on the device (graph replay)
            _abi.check(_abi.lib().iwvi_adam_step_dev(arr, n, lr, self.betas[0], self.betas[1], self.epsilon,
                                                    self._t_dev.data_ptr(), 1, _abi.stream_ptr()))
        return keep

    def _gradients(self, zs, advance=True, wrt="all", **one_factor):
        from .sharding import allreduce_gradients, k_shard_gradients
        if advance:
            self.model.next_minibatch()                          # gpflow.Minibatch: a new batch per session.run (models.py:21-26)
        if self.shard == "k":
            return k_shard_gradients(self.model, zs, K_total=self.K_total, group=self.group, wrt=wrt)
        elbo, g = iw_elbo_and_gradients(self.model, zs, wrt=wrt, **one_factor)
        g["__elbo__"] = elbo.reshape(1)                          # rides in the same bucket: the job's bound
        g = allreduce_gradients(g, weight=self.shard_weight, group=self.group)
        return g.pop("__elbo__")[0], g

    def _one_factor(self):
        """``step`` shares one factorisation between its two ops: single rank (a sharded step's collectives sit between the first op's
        evaluation and its update -- the dense factors would be in flight across them), N-layout."""
        from . import sharding as _sh
        return self.one_factorisation and self.world == 1 and self.shard != "k" and not _sh._FORCE_ONE_RANK_COLLECTIVES

    def natgrad_op(self, zs=None, _advance=True, _prefactor=False):
        """``op_ng``: one ELBO + gradient evaluation, natural-gradient step on the final layer's q(u).  Only that layer's
        (q_mu, q_sqrt) gradients are formed (``wrt="final_q"``): the op reads nothing else."""
        elbo, g = self._gradients(zs, _advance, wrt="final_q", **({"prefactor": True} if _prefactor else {}))
        i = len(self.model.layers) - 1
        f = self.final
        gamma = staircase_decay(self.gamma, self.global_step, self.gamma_decay)
        dq_mu = _abi.dev_tensor(g["l%d.q_mu" % i].contiguous(), "dq_mu")
        dq_sqrt = _abi.dev_tensor(g["l%d.q_sqrt" % i].contiguous(), "dq_sqrt")
        _abi.check(_abi.lib().iwvi_natgrad_step_ex(_abi.ptr(f.q_mu), _abi.ptr(f.q_sqrt), _abi.ptr(dq_mu), _abi.ptr(dq_sqrt),
                                                  f.num_inducing, f.num_outputs, gamma, self._ng_ws.data_ptr(), self._ng_ws.numel(), _abi.stream_ptr()))
        if _prefactor:
            # the dense float64 factors the NEXT op's adjoints read, beside this update (a handful of small launches): queued behind it -- in a
            # captured step the first successor keeps the hardware queue -- but ordered only behind the evaluation; joined here
            from . import backward as _bw
            ps = _bw.prepare_stream(f.q_mu.device)
            _bw.prefactor_dense(self.model, ps, self.model._prefactor_after, skip_q_of={i})
            if _prefactor == "join":                             # an op captured on its own must end with every stream joined; otherwise the next
                torch.cuda.current_stream().wait_stream(ps)      # op's preparation follows on the same stream and ITS join covers this work (a
                                                                 # join here is a cross-queue wait in front of the next op's launches: ~10 us)
        return elbo

    def adam_op(self, zs=None, _advance=True, _q_moved=False):
        """``op_adam``: one ELBO + gradient evaluation, Adam step on everything but the final layer's q(u)."""
        elbo, g = self._gradients(zs, _advance, **({"q_moved": {len(self.model.layers) - 1}} if _q_moved else {}))
        self._adam_call(g, lr=staircase_decay(self.lr, self.global_step, self.lr_decay))
        for _, owner in self._scalars:                         # host copies are refreshed lazily, when somebody reads them
            owner.mark_device_variance_changed()
        if self.check_finite and not self._capturing and not self.use_graph:
            self._raise_if_not_finite(elbo)
        return elbo

    def _raise_if_not_finite(self, elbo):
        # one small D2H: the bound and the final layer's q(u) must be finite (a natural-gradient step that leaves
        # -2 theta_2 indefinite makes TensorFlow's Cholesky raise in the reference; here it would go on as NaN)
        ok = torch.isfinite(elbo) & torch.isfinite(self.final.q_sqrt).all() & torch.isfinite(self.final.q_mu).all()
        if not bool(ok.item()):
            raise FloatingPointError("training step %d: non-finite bound or final-layer q(u) (the natural-gradient step left the "
                                     "precision matrix indefinite: lower gamma)" % self.global_step)

    _capturing = False
    _cap_stream = None

    def _capture_stream(self):
        """ONE capture stream per trainer (backward keeps per-stream side streams: a fresh stream per re-capture leaks them)."""
        if self._cap_stream is None:
            self._cap_stream = torch.cuda.Stream(device=self.model.X.device)
        return self._cap_stream

    def _replay(self, name, op):
        """Capture ``op`` (one evaluation + its update) into a hipGraph on first use -- and again whenever the staircase decay
        moves lr / gamma, which enter the update kernels by value -- then replay it."""
        epoch = (self.global_step // 1000, self.model.route_key())
        ent = self._graphs.get(name)
        if ent is not None and ent[0] == epoch:
            ent[1].replay()
            return ent[2]
        first = ent is None
        self._capturing = True
        try:
            side = self._capture_stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                if first:
                    eager = op()                               # this call's evaluation, launched eagerly (it also warms the allocator pools)
                from . import sharding as _sh
                if self.world > 1 or _sh._FORCE_ONE_RANK_COLLECTIVES:   # sharded: graph segments with the collectives between them
                    torch.cuda.synchronize()
                    g = _SegmentedGraph()
                    elbo = g.record(op)
                    torch.cuda.synchronize()
                else:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                        elbo = op()                            # recorded, not executed
            torch.cuda.current_stream().wait_stream(side)
        finally:
            self._capturing = False
        self._graphs[name] = (epoch, g, elbo)
        if first:
            return eager
        g.replay()                                             # re-capture after a decay step: this call's evaluation
        return elbo

    def step(self, zs_ng=None, zs_adam=None):
        """``model.train_op`` (build_models.py:297-300); returns the ELBO seen by the Adam op."""
        self.global_step += 1
        self._autotune_if_due()
        if self.use_graph:
            if zs_ng is not None or zs_adam is not None:
                raise ValueError("use_graph draws the noise on the device (captured graphs cannot take per-step host arguments)")
            if self.model.minibatch_size is None:
                # full-batch data: nothing host-driven separates the two ops, so the step is ONE replay (two replays on one stream start
                # about 10 us apart: 2 % of a configs[2] step)
                self._advance_outside_graph()
                elbo = self._replay("step", self._both_ops)
            else:
                self._advance_outside_graph()
                one = self._one_factor()
                self._replay("ng", lambda: self.natgrad_op(None, _advance=False, _prefactor="join" if one else False))
                self._advance_outside_graph()
                elbo = self._replay("adam", lambda: self.adam_op(None, _advance=False, _q_moved=one))
            for _, owner in self._scalars:                     # a replay runs no host code: flag the host copies here
                owner.mark_device_variance_changed()
            if self.check_finite and self.global_step % self.check_every == 0:
                self._raise_if_not_finite(elbo)
            return elbo
        one = self._one_factor()
        self.natgrad_op(zs_ng, _prefactor="open" if one else False)
        return self.adam_op(zs_adam, _q_moved=one)

    def _both_ops(self):
        one = self._one_factor()
        self.natgrad_op(None, _advance=False, _prefactor="open" if one else False)
        return self.adam_op(None, _advance=False, _q_moved=one)

    def _advance_outside_graph(self):
        """Everything of a minibatch change that is host-driven runs here, never inside the captured op: the in-place
        gather into the model's X / Y buffers AND the [x, y] rows the encoders read (``_xy_minibatch`` re-runs its
        ``cat`` only when the minibatch key moved -- inside the captured op the key is unchanged, so the cat would
        not be recorded and every replay would read the first step's rows)."""
        self.model.next_minibatch()
        if any(isinstance(l, LatentVariableLayer) for l in self.model.layers):
            self.model._xy_minibatch()

    def sync_scalars(self):
        """Refresh the host copies of the device-resident scalars now (otherwise: lazily, on first read)."""
        return [owner.variance for _, owner in self._scalars]

    # -- checkpoint / resume (reference: gpflow Saver of the whole session, run_conditional_density_estimation.py:95-125) --
    def state_dict(self):
        self.adam_t = int(self._t_dev.item())                  # Adam's step count lives on the device
        out = {"global_step": np.int64(self.global_step), "adam_t": np.int64(self.adam_t)}
        for (name, _, _), (x, m, v) in zip(self._entries, self._state):
            out["x." + name], out["m." + name], out["v." + name] = (a.detach().cpu().numpy() for a in (x, m, v))
        return out

    def load_state_dict(self, state):
        """After the model's parameters have been restored in place (that also restores the device masters of the trained
        scalars, which are bound to the model's ``variance`` attributes): the optimiser's own state and the step counters."""
        self.global_step, self.adam_t = int(state["global_step"]), int(state["adam_t"])
        self._t_dev.fill_(self.adam_t)
        self._graphs = {}
        self._tuned_epoch = None                               # the restored parameters get their own measurement
        self._autotune_if_due()
        for (name, _, _), (x, m, v) in zip(self._entries, self._state):
            for key, dst in (("x.", x), ("m.", m), ("v.", v)):
                dst.copy_(torch.as_tensor(np.asarray(state[key + name]), dtype=dst.dtype, device=dst.device).reshape(dst.shape))
        # (the device masters of the trained scalars already hold the checkpoint's exact values: restoring the model's
        # ``variance`` attributes writes through to the bound device tensors; recomputing them from x would be 1 ulp off)
        return self
