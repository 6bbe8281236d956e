"""Model classes of the reference's ``dgps_with_iwvi/models.py`` (DGP_VI :9-107, DGP_IWVI :110-150)
with the same constructor / method signatures, running on the HIP kernels behind include/iwvi_hip.h.

``DGP_IWVI._build_likelihood`` is the north-star function; ``E_log_p_Y`` (the name used in the older
doubly-stochastic DGP code and in BASELINE.json) is an alias for its per-point log-weight stage.

One ELBO evaluation is TWO launches: ``iwvi_model_precompute`` (per GP layer: Gram + float64 Cholesky + operand packing; the
encoders of the latent-variable layers) and ``iwvi_dgp_forward``, whose workgroups each carry a chunk of samples through the
tiling of X/Y over K, every layer, the Gaussian variational expectation and the local regularisers in LDS; the last workgroup
to finish does the log-sum-exp over K, the scaled sum and subtracts the global KLs.

Differences from the reference, all documented in DESIGN.md:
  * ``zs`` (one N(0,1) array or None per layer) injects the noise tf.random_normal draws in-graph; None
    draws inside the kernel from a counter-based Philox stream (``settings.seed``, per-model step counter);
  * the IW path asks the final layer for marginal variances only (``full_cov_over_samples=False``):
    the reference builds the [B, Dy, K, K] covariance and keeps its diagonal (models.py:129-133),
    the result is identical; set the flag to follow the reference literally (layer-by-layer launches).
    Inner layers: a SharedMixedMok layer samples marginally in the reference too (temp_workaround.py:125-129), so the
    fused launch is exact for the benchmark family; an inner GPLayer with a PLAIN kernel draws its K samples per point
    jointly there (:149-155, full_cov=True) -- such a model is detected (``_joint_over_samples``) and evaluated along
    the literal layer-by-layer path, never with marginal draws.
"""
import ctypes

import numpy as np
import torch

from . import _abi, settings
from .layers import GPLayer, LatentVariableLayer, RegularizerType
from .temp_workaround import draw_normal, precompute_states


def _data(x):
    t = torch.as_tensor(np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x)
    return t.to(dtype=settings.float_type, device=settings.default_device()).contiguous()


class DGP_VI:
    def __init__(self, X, Y, layers, likelihood, num_samples=1, minibatch_size=None, name=None):
        self.likelihood = likelihood
        self.num_data = X.shape[0]                                    # models.py:18
        self.num_samples = num_samples
        self._X_all, self._Y_all = _data(X), _data(Y)
        self.minibatch_size = minibatch_size
        self._mb_rng = np.random.RandomState(0)                       # Minibatch(seed=0), models.py:25-26
        self._mb_perm, self._mb_pos = None, 0
        self._mb_serial = 0                                           # bumped by every minibatch change: cache key
        self.X, self.Y = self._X_all, self._Y_all
        if minibatch_size is not None:
            self.next_minibatch()
        self.layers = list(layers)
        self.name = name
        self.full_cov_over_samples = False
        # IW-ELBO only: evaluate a leading latent-variable layer inside the precompute launch (beside the
        # factorisations) instead of inside the layer kernel; keep_lv_noise exports its draws (tests)
        self.lv_in_precompute = False
        self.keep_lv_noise = False
        self._dev_words = None

    # -- data ---------------------------------------------------------------------------------
    def next_minibatch(self):
        """Advance to the next minibatch (shuffled epochs, X and Y aligned like gpflow.Minibatch)."""
        if self.minibatch_size is None:
            return
        n = self._X_all.shape[0]          # the rows THIS model holds (num_data may be the job total of a data-parallel run)
        b = min(self.minibatch_size, n)
        if self._mb_perm is None or self._mb_pos + b > n:
            self._mb_perm = torch.as_tensor(self._mb_rng.permutation(n), device=self._X_all.device)
            self._mb_pos = 0
        idx = self._mb_perm[self._mb_pos:self._mb_pos + b]
        self._mb_pos += b
        if self.X.shape[0] == b and self.X.data_ptr() != self._X_all.data_ptr():
            # same buffers every step (a captured hipGraph of the step keeps reading them)
            torch.index_select(self._X_all, 0, idx, out=self.X)
            torch.index_select(self._Y_all, 0, idx, out=self.Y)
        else:
            self.X, self.Y = self._X_all[idx].contiguous(), self._Y_all[idx].contiguous()
        self._mb_serial += 1         # the caching allocator reuses addresses: pointers alone cannot key the per-minibatch caches

    def to(self, device):
        self._X_all, self._Y_all = self._X_all.to(device), self._Y_all.to(device)
        self.X, self.Y = self.X.to(device), self.Y.to(device)
        self._mb_serial += 1
        for layer in self.layers:
            layer.to(device)
        return self

    def _words(self):
        """[ticket, rng step, rng ticket] device words, zeroed once (never per call)."""
        dev = self.X.device
        if self._dev_words is None or self._dev_words.device != dev:
            self._dev_words = torch.zeros(4, dtype=torch.int64, device=dev)
        return self._dev_words

    # -- reference API ------------------------------------------------------------------------
    def precompute(self, with_encoders=False, sample_first=None, dense=False, q_moved=None):
        """Gram + Cholesky + operand packing of every GP layer: one ABI call, one launch.  ``with_encoders``:
        the same launch also evaluates the encoder MLP of every latent-variable layer on the current minibatch
        (it does not depend on the factorisation, so it runs beside it instead of inside the layer kernel).
        ``sample_first`` = dict(K, sampled_kl, want_z): a latent-variable layer at position 0 is evaluated in full
        there (its K samples per row, their regulariser); ``_fused_forward(stack_from=1)`` continues from it."""
        encs, keep = [], []
        if with_encoders:
            for i, l in enumerate(self.layers):
                if isinstance(l, LatentVariableLayer) and len(encs) < 2:
                    smp = None
                    if sample_first is not None and i == 0:
                        smp = dict(sample_first, X=self.X, layer_index=0, seed=settings.seed,
                                   rng_state=ctypes.c_void_p(self._words().data_ptr() + 8))
                    e, k = l.enc_desc(self._xy_minibatch(), sample=smp)
                    if e is not None:                            # (None: a custom-activation encoder, already evaluated by torch ops)
                        encs.append(e)
                    keep.append(k)
                    l._enc_key = self._mb_key()
        if q_moved is None:
            descs = [l.state_desc() for l in self.layers if isinstance(l, GPLayer)]
        else:
            # ``q_moved`` (layer indices; the CALLER vouches for it -- training.Trainer.step between its two ops): since the last full precompute
            # on these state buffers nothing but these layers' q(u) has changed.  Their q(u) images are rewritten (IWVI_GP_REUSE_FACTOR), every
            # factorisation stays as it is, a layer of which nothing moved is not in the call at all; the encoders run as always.
            descs = []
            for i in sorted(q_moved):
                d = self.layers[i].state_desc()
                d.flags |= _abi.GP_REUSE_FACTOR
                descs.append(d)
            if dense or not descs:
                raise ValueError("q_moved: at least one layer, and not together with dense")
        if dense:                                                    # the adjoints read the dense float64 Lm, Lm^-1
            for d in descs:                                          # ("lm": the factor only; Lm^-1 by iwvi_gp_dense_inverse)
                d.flags |= _abi.GP_WANT_LM if dense == "lm" else _abi.GP_WANT_DENSE
        precompute_states(descs, encs)
        if dense == "lm" and descs:
            arr = (_abi.GpDesc * len(descs))(*descs)
            _abi.check(_abi.lib().iwvi_gp_dense_inverse(arr, len(descs), _abi.stream_ptr()))

    def autotune_f64(self, threshold=300.0):
        """Record every GP layer's MEASURED float64-route choice from the factor itself: max / min of diag(Lm) of the CURRENT parameters
        (one precompute launch with the dense factors for all layers + ONE read-back: a synchronisation, so it is a calibration call -- after
        building or loading a model, at every staircase epoch of a training run (``training.Trainer`` does both) -- not part of an evaluation).
        diag(Lm) spans [sqrt(jitter), sigma] -- at most 1e3 at the default jitter and unit variance.  Measured (scripts/diag_ratio.py): 14 / 28 / 85
        at the BASELINE stacks (8-D, M = 128 / 256 / 512), 430 at 4-D M = 128, 820-990 for 1-3-D inputs at any M >= 32; the default
        ``threshold`` 300 separates what float32 holds at the stated tolerance from what it does not (profiles/r05_f64_route_error.txt).
        A layer's explicit ``f64_stage1 = True / False`` still wins over the measurement.  Returns [{layer, diag_ratio, f64_stage1}]."""
        gps = [(i, l) for i, l in enumerate(self.layers) if isinstance(l, GPLayer)]
        self.precompute(dense=True)
        ratios = []
        for _, l in gps:
            st = l.state()
            d = torch.diagonal(st.view("Lm", torch.float64, st.Mp * st.Mp).view(st.Mp, st.Mp)[:st.M, :st.M])
            ratios.append(d.max() / d.min())
        ratios = torch.stack(ratios).tolist() if ratios else []    # the one device-to-host copy
        rep = []
        for (i, l), ratio in zip(gps, ratios):
            l._f64_measured = bool(ratio >= threshold) if ratio == ratio else True      # (NaN: the float32 image of the factor broke down)
            rep.append(dict(layer=i, diag_ratio=float(ratio), f64_stage1=l.uses_f64_stage1()))
        self.precompute()                                         # (the states as an evaluation expects them: flags of the new choice)
        return rep

    def route_key(self):
        """What a captured graph bakes in besides the shapes: each GP layer's arithmetic route (``training.Trainer`` re-captures when it moves)."""
        return tuple((l.uses_f64_stage1(), bool(settings.fw_f32_stage2)) for l in self.layers if isinstance(l, GPLayer))

    def propagate(self, X, full_cov=False, inference_amorization_inputs=None,
                  is_sampled_local_regularizer=False, zs=None, _precomputed=False, _kl_parts=False, _last_sample=True):
        """reference models.py:31-46 -> (samples[1:], means, covs, kls, kl_types); one launch per layer.
        ``_last_sample=False``: the final layer's sample is not needed (None is returned in its place)."""
        if not _precomputed:
            self.precompute()
        samples, means, covs, kls, kl_types = [X, ], [], [], [], []
        zs = [None] * len(self.layers) if zs is None else zs
        if len(zs) != len(self.layers):
            raise ValueError("zs needs one entry per layer")
        for i, (layer, z) in enumerate(zip(self.layers, zs)):
            sample, mean, cov, kl = layer.propagate(samples[-1], full_cov=full_cov,
                                                    inference_amorization_inputs=inference_amorization_inputs,
                                                    is_sampled_local_regularizer=is_sampled_local_regularizer,
                                                    z=z, _precomputed=True, _kl_parts=_kl_parts,
                                                    _want_sample=_last_sample or i + 1 < len(self.layers))
            samples.append(sample)
            means.append(mean)
            covs.append(cov)
            kls.append(kl)
            kl_types.append(layer.regularizer_type)
        return samples[1:], means, covs, kls, kl_types

    # -- fused forward ------------------------------------------------------------------------
    def _fused_forward(self, T, row_div, row_mod, lead, zs=None, sampled_kl=True, want_layers=False,
                       want_logw=True, use_encoder=True, elbo=None, stack_from=0, want_saved=False, outputs_for=None, moments=True):
        """``iwvi_dgp_forward`` over the current minibatch: every layer + log-weights in one launch.
        Row t of the flattened batch reads data row (t // row_div) % row_mod.  ``elbo`` = dict(B, K, stride_b,
        stride_k, mode_vi, want_ms, K_total): also run the reduction of models.py:138-150 in the tail of the
        launch.  Returns (logw [T] or None, per-layer dict lists when ``want_layers``, (elbo, logp, ms) or None).
        ``stack_from=1`` (needs ``elbo``): layer 0 was evaluated by ``precompute(sample_first=...)``; the launch
        starts from its samples [T, Dx+Lw] and its per-sample regulariser.
        ``outputs_for`` (with ``want_layers``): only these layers (indices into the stack) get output buffers, the others write
        nothing; ``moments=False``: no sample / mean / var rows either (only what ``want_saved`` adds) -- the natural-gradient op needs
        the final layer's a, noise and latent moments alone."""
        dev = self.X.device
        layers = self.layers[stack_from:]
        n = len(layers)
        if n > _abi.MAX_STACK:
            raise ValueError("more than %d layers in one fused launch" % _abi.MAX_STACK)
        zs = [None] * n if zs is None else zs
        if len(zs) != n:
            raise ValueError("zs needs one entry per layer")
        if stack_from:
            first = self.layers[0]
            if stack_from != 1 or elbo is None or getattr(first, "_smp_X", None) is None or first._smp_X.shape[0] != T:
                raise ValueError("stack_from=1 needs elbo and precompute(sample_first=...) on this minibatch")
            X = first._smp_X
        else:
            X = _abi.dev_tensor(self.X.contiguous(), "X")
        Y = _abi.dev_tensor(self.Y.contiguous(), "Y")
        XY = self._xy_minibatch() if use_encoder and any(isinstance(l, LatentVariableLayer) for l in layers) else None
        descs = (_abi.LayerDesc * n)()
        keep, outs = [], []
        D = X.shape[1]
        for i, (layer, z) in enumerate(zip(layers, zs)):
            if isinstance(layer, GPLayer):
                R = layer.num_outputs
                z2 = None if z is None else _abi.dev_tensor(z.reshape(T, R).contiguous(), "z")
                o = None
                if want_layers and (outputs_for is None or i in outputs_for):
                    P = layer.kern.W.shape[0] if hasattr(layer.kern, "W") else R
                    o = {k: torch.empty(*lead, P, dtype=settings.float_type, device=dev) for k in ("sample", "mean", "var")} if moments else {}
                    if want_saved:                               # what the adjoint of this layer needs (backward.py)
                        Mp = layer.state().Mp
                        o["a_out"] = torch.empty(T, Mp, dtype=settings.float_type, device=dev)
                        from .backward import needs_saved_u
                        if needs_saved_u(layer, T):     # only the GEMM path of the adjoint reads u_r = L_r^T a
                            o["u_out"] = torch.empty(R, T, Mp, dtype=settings.float_type, device=dev)
                        o["noise_out"] = torch.empty(T, R, dtype=settings.float_type, device=dev)
                        o["gmv_out"] = torch.empty(T, 3 * R, dtype=settings.float_type, device=dev)
                d, k = layer.fused_desc(z2, o)
                if d.D != D:
                    raise ValueError("layer %d expects %d inputs, got %d" % (i, d.D, D))
                D = d.P
            elif isinstance(layer, LatentVariableLayer):
                Lw = layer.latent_dim
                z2 = None if z is None else _abi.dev_tensor(z.reshape(T, Lw).contiguous(), "z")
                o = None
                if want_layers and (outputs_for is None or i in outputs_for):
                    o = {k: torch.empty(*lead, D + Lw, dtype=settings.float_type, device=dev) for k in ("sample", "mean", "var")} if moments else {}
                    o["kl_local"] = torch.empty(*lead, Lw, dtype=settings.float_type, device=dev)
                    if want_saved:
                        o["noise_out"] = torch.empty(T, Lw, dtype=settings.float_type, device=dev)
                # encoder output of THIS minibatch from the last precompute launch, if there is one
                eo = layer._enc_out if (use_encoder and getattr(layer, "_enc_key", None) == self._mb_key()) else None
                if eo is None and use_encoder and layer.encoder.custom_act is not None:      # a custom activation never runs inside the kernels
                    layer.enc_desc(self._xy_minibatch())
                    layer._enc_key = self._mb_key()
                    eo = layer._enc_out
                d, k = layer.fused_desc(D, z2, o, sampled_kl=sampled_kl, use_encoder=use_encoder, enc_out=eo)
                D += Lw
            else:
                raise TypeError("the fused forward knows GPLayer and LatentVariableLayer; use propagate() for custom layers")
            descs[i] = d
            keep.append(k)
            outs.append(o)
        logw = torch.empty(T, dtype=settings.float_type, device=dev) if want_logw else None
        words = self._words()
        ed, red = None, None
        if elbo is not None:
            B, K = elbo["B"], elbo["K"]
            glob = [_abi.dev_tensor(g.reshape(-1), "global kl", torch.float64) for g in self._global_kls()]
            glob_p = _abi.ptr_array(glob)
            glob_n = (ctypes.c_int32 * max(len(glob), 1))(*[g.numel() for g in glob])
            logp = torch.empty(B, dtype=settings.float_type, device=dev)
            # caller-provided result buffers (e.g. a slot of an exchange staging ring) or fresh ones
            val = elbo.get("elbo_out")
            val = torch.empty(1, dtype=torch.float64, device=dev) if val is None else _abi.dev_tensor(val.view(-1), "elbo_out", torch.float64)
            ms = elbo.get("ms_out")
            if ms is not None:
                ms = _abi.dev_tensor(ms, "ms_out", settings.float_type)
                if tuple(ms.shape) != (B, 2):
                    raise ValueError("ms_out must be [B, 2]")
            elif elbo.get("want_ms"):
                ms = torch.empty(B, 2, dtype=settings.float_type, device=dev)
            ed = _abi.ElboDesc()
            ed.B, ed.K, ed.stride_b, ed.stride_k = B, K, elbo["stride_b"], elbo["stride_k"]
            ed.kl_global, ed.kl_global_counts, ed.n_glob = glob_p, glob_n, len(glob)
            ed.scale = float(self.num_data) / float(B)                     # models.py:80-81, :144-145
            ed.K_total, ed.mode_vi = elbo.get("K_total") or K, 1 if elbo["mode_vi"] else 0
            ed.out_lse_ms = None if ms is None else ms.data_ptr()
            ed.lik_variance_dev = self.likelihood.desc_variance()[1]
            ws = torch.empty((T + 15) // 16, dtype=torch.float64, device=dev)
            ed.out_logp, ed.out_elbo, ed.ws = logp.data_ptr(), val.data_ptr(), ws.data_ptr()
            adj = elbo.get("adj")                                # heads of the bound's adjoint from the same launch (backward.py)
            if adj is not None:
                ed.adj_w, ed.adj_dmean, ed.adj_dvar, ed.adj_sums = (adj["w"].data_ptr(), adj["d_mean"].data_ptr(),
                                                                    adj["d_var"].data_ptr(), adj["sums"].data_ptr())
            if stack_from:
                ed.lw_init, ed.noise_layer_base, ed.x_per_sample = self.layers[0]._smp_kl.data_ptr(), stack_from, 1
            keep.append((glob, glob_p, glob_n, ws))
            red = (val[0], logp, ms)
        args = (descs, n, _abi.ptr(X), X.shape[1], _abi.ptr(XY), 0 if XY is None else XY.shape[1],
                _abi.ptr(Y) if want_logw else None, Y.shape[1], T, row_div, row_mod,
                self.likelihood.desc_variance()[0] if (ed is not None or not want_logw) else self.likelihood.variance,
                settings.seed, ctypes.c_void_p(words.data_ptr() + 8), _abi.ptr(logw),
                None if ed is None else ctypes.byref(ed), _abi.stream_ptr())
        _abi.check(_abi.lib().iwvi_dgp_forward(*args))
        return logw, outs, red

    def _mb_key(self):
        return (self._mb_serial, self.X.data_ptr(), self.Y.data_ptr(), self.X.shape[0])

    def invalidate_minibatch_caches(self):
        """Call after changing ``model.X`` / ``model.Y`` IN PLACE (the [x, y] rows and encoder outputs are cached per minibatch)."""
        self._mb_serial += 1

    def _xy_minibatch(self):
        """[x_b, y_b] rows of the current minibatch (models.py:53 / :116 before tiling), cached per minibatch."""
        key = self._mb_key()
        if getattr(self, "_xy_key", None) != key:
            cur = getattr(self, "_xy_cache", None)
            shape = (self.X.shape[0], self.X.shape[1] + self.Y.shape[1])
            if cur is not None and tuple(cur.shape) == shape and cur.device == self.X.device:
                torch.cat([self.X, self.Y], -1, out=cur)          # same buffer every minibatch (graph capture)
            else:
                self._xy_cache = torch.cat([self.X, self.Y], -1).contiguous()
            self._xy_key = key
        return self._xy_cache

    def _global_kls(self):
        return [l.state().kl_parts for l in self.layers if l.regularizer_type is RegularizerType.GLOBAL]

    def _reduce_logw(self, logw, global_kls, B, K, stride_b, stride_k, mode_vi, want_ms=False, K_total=None):
        """``iwvi_logw_reduce``: logsumexp/mean over K + scaled sum - global KLs (models.py:138-150, :69-86)."""
        dev = logw.device
        glob = [_abi.dev_tensor(g.reshape(-1), "global kl", torch.float64) for g in global_kls]
        glob_n = (ctypes.c_int32 * max(len(glob), 1))(*[g.numel() for g in glob])
        logp = torch.empty(B, dtype=settings.float_type, device=dev)
        elbo = torch.empty(1, dtype=torch.float64, device=dev)
        ms = torch.empty(B, 2, dtype=settings.float_type, device=dev) if want_ms else None
        scale = float(self.num_data) / float(B)                        # models.py:80-81, :144-145
        _abi.check(_abi.lib().iwvi_logw_reduce(
            _abi.ptr(logw), B, K, stride_b, stride_k, _abi.ptr_array(glob), glob_n, len(glob),
            scale, K_total or K, 1 if mode_vi else 0, _abi.ptr(ms), _abi.ptr(logp), _abi.ptr(elbo),
            _abi.ptr(self._words()), _abi.stream_ptr()))
        return elbo[0], logp, ms

    def _reduce(self, fmean, fvar, Y, local_kls, global_kls, B, K, stride_b, stride_k, mode_vi,
                want_ms=False, K_total=None):
        """``iwvi_iw_elbo_reduce`` on explicit final-layer moments (the layer-by-layer path)."""
        dev = fmean.device
        Dy = Y.shape[-1]
        fmean = _abi.dev_tensor(fmean.contiguous(), "final mean")
        fvar = _abi.dev_tensor(fvar.contiguous(), "final var")
        Y = _abi.dev_tensor(Y.contiguous(), "Y")
        kls = [_abi.dev_tensor(k.contiguous(), "local kl") for k in local_kls]
        if len(kls) > _abi.MAX_KL:
            raise ValueError("more than %d latent-variable layers" % _abi.MAX_KL)
        kl_dims = (ctypes.c_int32 * max(len(kls), 1))(*[k.shape[-1] for k in kls])
        glob = [_abi.dev_tensor(g.reshape(-1), "global kl", torch.float64) for g in global_kls]
        glob_n = (ctypes.c_int32 * max(len(glob), 1))(*[g.numel() for g in glob])
        logp = torch.empty(B, dtype=settings.float_type, device=dev)
        elbo = torch.empty(1, dtype=torch.float64, device=dev)
        ms = torch.empty(B, 2, dtype=settings.float_type, device=dev) if want_ms else None
        scale = float(self.num_data) / float(B)                        # models.py:80-81, :144-145
        # (a trained likelihood variance is read on the device: no device-to-host copy here -- there must be none inside a captured step --
        #  and a replayed graph follows the value)
        lik_host, lik_dev = self.likelihood.desc_variance()
        _abi.check(_abi.lib().iwvi_iw_elbo_reduce_dev(
            _abi.ptr(fmean), _abi.ptr(fvar), _abi.ptr(Y), lik_host, lik_dev, B, K, Dy,
            stride_b, stride_k, _abi.ptr_array(kls), kl_dims, len(kls), _abi.ptr_array(glob), glob_n, len(glob),
            scale, K_total or K, 1 if mode_vi else 0, _abi.ptr(ms), _abi.ptr(logp), _abi.ptr(elbo),
            _abi.ptr(self._words()), _abi.stream_ptr()))
        return elbo[0], logp, ms

    def _build_likelihood(self, zs=None):
        """The VI bound, reference models.py:49-86 (2-D [S*N, D] tiling, mean over S)."""
        S, N = self.num_samples, self.X.shape[0]
        self.precompute(with_encoders=True)
        # tile(X, [S, 1]) (:50-53): row t = s*N + n reads data row t % N; analytic local KL (:58-61)
        _, _, red = self._fused_forward(S * N, 1, N, (S * N,), zs=zs, sampled_kl=False,
                                        elbo=dict(B=N, K=S, stride_b=1, stride_k=N, mode_vi=True))
        return red[0]

    def compute_log_likelihood(self, zs=None):
        """gpflow ``Model.compute_log_likelihood`` (reference tests/test_gp_layer.py:50): host float."""
        return float(self._build_likelihood(zs).item())

    likelihood_tensor = property(lambda self: self._build_likelihood())

    def _build_predict(self, X, full_cov=False, zs=None):
        _, means, covs, _, _ = self.propagate(X, full_cov=full_cov, zs=zs, _last_sample=False)   # :89-91
        return means[-1], covs[-1]

    def predict_f(self, X, zs=None):
        return self._build_predict(_data(X), False, zs)

    def predict_f_full_cov(self, X, zs=None):
        return self._build_predict(_data(X), True, zs)

    def predict_f_multisample(self, X, S, zs=None):
        X = _data(X)
        X_tiled = X[None, :, :].expand(S, *X.shape).contiguous()       # :97
        _, means, covs, _, _ = self.propagate(X_tiled, zs=zs)
        return means[-1], covs[-1]

    def predict_y_samples(self, X, S, zs=None, z_y=None):
        X = _data(X)
        X_tiled = X[None, :, :].expand(S, *X.shape).contiguous()       # :104
        _, means, covs, _, _ = self.propagate(X_tiled, zs=zs)
        m, v = self.likelihood.predict_mean_and_var(means[-1], covs[-1])   # :105
        z = draw_normal(m.shape, m.device) if z_y is None else z_y
        return m + z * v ** 0.5                                        # :106-107


class DGP_IWVI(DGP_VI):
    def _joint_over_samples(self):
        """True when some non-final GPLayer has a plain kernel and K > 1: the reference then samples that layer with the
        full [K, K] covariance over the importance samples (models.py:122-125 -> temp_workaround.py:149-155), which the
        fused marginal-sampling launch does not reproduce."""
        return self.num_samples > 1 and any(isinstance(l, GPLayer) and not hasattr(l.kern, "W") for l in self.layers[:-1])

    def _literal(self):
        return self.full_cov_over_samples or self._joint_over_samples()

    def _forward_iw(self, zs=None, _last_sample=True):
        """models.py:113-133 with every per-layer output exposed (tests / API parity).  Default: one fused
        launch, marginal variances everywhere.  With ``full_cov_over_samples`` the reference is followed
        literally (explicit tiling, full_cov=True, matrix_diag_part of the [B, Dy, K, K] covariance)."""
        B, K = self.X.shape[0], self.num_samples
        if self._literal():
            X_tiled = self.X[:, None, :].expand(B, K, self.X.shape[1]).contiguous()     # :113
            Y_tiled = self.Y[:, None, :].expand(B, K, self.Y.shape[1]).contiguous()     # :114
            XY = torch.cat([X_tiled, Y_tiled], -1)                                       # :116
            samples, means, covs, kls, kl_types = self.propagate(
                X_tiled, full_cov=True, inference_amorization_inputs=XY,
                is_sampled_local_regularizer=True, zs=zs, _kl_parts=True, _last_sample=_last_sample)   # :122-125
            local_kls = [kl for kl, t in zip(kls, kl_types) if t is RegularizerType.LOCAL]
            global_kls = [kl for kl, t in zip(kls, kl_types) if t is RegularizerType.GLOBAL]
            cov = covs[-1]
            if cov.dim() == 4:                                                            # [B, Dy, K, K]
                cov = torch.diagonal(cov, dim1=-2, dim2=-1).transpose(1, 2).contiguous()  # :133
            return means[-1], cov, local_kls, global_kls, samples, means, covs
        self.precompute(with_encoders=True)
        _, outs, _ = self._fused_forward(B * K, K, B, (B, K), zs=zs, sampled_kl=True, want_layers=True)
        samples, means, covs = ([o[k] for o in outs] for k in ("sample", "mean", "var"))
        local_kls = [o["kl_local"] for o, l in zip(outs, self.layers) if l.regularizer_type is RegularizerType.LOCAL]
        return means[-1], covs[-1], local_kls, self._global_kls(), samples, means, covs

    def _logw(self, zs=None):
        """Per-sample log-weights L_NK [B*K] (models.py:134-142) and the global KL shares."""
        B, K = self.X.shape[0], self.num_samples
        if self._literal():
            return None
        self.precompute(with_encoders=True)
        return self._fused_forward(B * K, K, B, (B, K), zs=zs, sampled_kl=True)[0]

    def _elbo_parts(self, zs=None, want_ms=False, K_total=None, ms_out=None, elbo_out=None):
        B, K = self.X.shape[0], self.num_samples
        if self._literal():                                          # literal reference path, layer by layer
            fmean, fvar, local_kls, global_kls, _, _, _ = self._forward_iw(zs, _last_sample=False)
            return self._reduce(fmean, fvar, self.Y, local_kls, global_kls, B, K, stride_b=K, stride_k=1,
                                mode_vi=False, want_ms=want_ms, K_total=K_total)
        el = dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False, want_ms=want_ms, K_total=K_total,
                  ms_out=ms_out, elbo_out=elbo_out)
        if self.lv_in_precompute and isinstance(self.layers[0], LatentVariableLayer) and (zs is None or zs[0] is None):
            # the leading latent-variable layer rides in the precompute launch, beside the factorisations
            self.precompute(with_encoders=True, sample_first=dict(K=K, sampled_kl=True, want_z=self.keep_lv_noise))
            return self._fused_forward(B * K, K, B, (B, K), zs=None if zs is None else zs[1:], sampled_kl=True,
                                       elbo=el, stack_from=1)[2]
        self.precompute(with_encoders=True)
        return self._fused_forward(B * K, K, B, (B, K), zs=zs, sampled_kl=True, elbo=el)[2]

    def _build_likelihood(self, zs=None, out=None):
        """The importance-weighted ELBO, reference models.py:112-150 (``out``: optional 1-element float64 result buffer)."""
        return self._elbo_parts(zs, elbo_out=out)[0]

    def E_log_p_Y(self, zs=None):
        """Per-point ``logsumexp_k(L_nk) - log K`` [B] (models.py:134-148); name from BASELINE.json."""
        return self._elbo_parts(zs)[1]

    def lse_partials(self, zs=None, K_total=None, out=None):
        """(max_k L, sum_k exp(L - max)) per point [B, 2] + the global KLs: the K-sharded exchange unit
        (``out``: optional [B, 2] buffer to write the pairs into)."""
        _, _, ms = self._elbo_parts(zs, want_ms=True, K_total=K_total, ms_out=out)
        return ms, self._global_kls()
