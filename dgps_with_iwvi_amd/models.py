"""Model classes of the reference's ``dgps_with_iwvi/models.py`` (DGP_VI :9-107, DGP_IWVI :110-150)
with the same constructor / method signatures, running on the HIP kernels behind include/iwvi_hip.h.

``DGP_IWVI._build_likelihood`` is the north-star function; ``E_log_p_Y`` (the name used in the older
doubly-stochastic DGP code and in BASELINE.json) is an alias for its per-point log-weight stage.

Differences from the reference, all documented in DESIGN.md:
  * ``zs`` (one N(0,1) array or None per layer) injects the noise tf.random_normal draws in-graph;
  * the IW path asks the final layer for marginal variances only (``full_cov_over_samples=False``):
    the reference builds the [B, Dy, K, K] covariance and keeps its diagonal (models.py:129-133),
    the result is identical; set the flag to follow the reference literally;
  * all per-step Gram/Cholesky work of every GP layer is batched into one ``iwvi_gp_precompute`` call.
"""
import ctypes

import numpy as np
import torch

from . import _abi, settings
from .layers import GPLayer, RegularizerType
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
        self.X, self.Y = self._X_all, self._Y_all
        if minibatch_size is not None:
            self.next_minibatch()
        self.layers = list(layers)
        self.name = name
        self.full_cov_over_samples = False
        self._ticket = None

    # -- data ---------------------------------------------------------------------------------
    def next_minibatch(self):
        """Advance to the next minibatch (shuffled epochs, X and Y aligned like gpflow.Minibatch)."""
        if self.minibatch_size is None:
            return
        n, b = self.num_data, min(self.minibatch_size, self.num_data)
        if self._mb_perm is None or self._mb_pos + b > n:
            self._mb_perm = torch.as_tensor(self._mb_rng.permutation(n), device=self._X_all.device)
            self._mb_pos = 0
        idx = self._mb_perm[self._mb_pos:self._mb_pos + b]
        self._mb_pos += b
        self.X, self.Y = self._X_all[idx].contiguous(), self._Y_all[idx].contiguous()

    def to(self, device):
        self._X_all, self._Y_all = self._X_all.to(device), self._Y_all.to(device)
        self.X, self.Y = self.X.to(device), self.Y.to(device)
        for layer in self.layers:
            layer.to(device)
        return self

    # -- reference API ------------------------------------------------------------------------
    def precompute(self):
        """Gram + Cholesky + operand packing of every GP layer: one ABI call, two launches."""
        precompute_states([l.state_desc() for l in self.layers if isinstance(l, GPLayer)])

    def propagate(self, X, full_cov=False, inference_amorization_inputs=None,
                  is_sampled_local_regularizer=False, zs=None, _precomputed=False, _kl_parts=False):
        """reference models.py:31-46 -> (samples[1:], means, covs, kls, kl_types)."""
        if not _precomputed:
            self.precompute()
        samples, means, covs, kls, kl_types = [X, ], [], [], [], []
        zs = [None] * len(self.layers) if zs is None else zs
        if len(zs) != len(self.layers):
            raise ValueError("zs needs one entry per layer")
        for layer, z in zip(self.layers, zs):
            sample, mean, cov, kl = layer.propagate(samples[-1], full_cov=full_cov,
                                                    inference_amorization_inputs=inference_amorization_inputs,
                                                    is_sampled_local_regularizer=is_sampled_local_regularizer,
                                                    z=z, _precomputed=True, _kl_parts=_kl_parts)
            samples.append(sample)
            means.append(mean)
            covs.append(cov)
            kls.append(kl)
            kl_types.append(layer.regularizer_type)
        return samples[1:], means, covs, kls, kl_types

    def _reduce(self, fmean, fvar, Y, local_kls, global_kls, B, K, stride_b, stride_k, mode_vi,
                want_ms=False, K_total=None):
        """``iwvi_iw_elbo_reduce``: var-exp + local terms + logsumexp/mean over K + scaled sum - global KLs."""
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
        if self._ticket is None or self._ticket.device != dev:
            self._ticket = torch.zeros(1, dtype=torch.int64, device=dev)      # zeroed once, never per call
        scale = float(self.num_data) / float(B)                        # models.py:80-81, :144-145
        _abi.check(_abi.lib().iwvi_iw_elbo_reduce(
            _abi.ptr(fmean), _abi.ptr(fvar), _abi.ptr(Y), self.likelihood.variance, B, K, Dy,
            stride_b, stride_k, _abi.ptr_array(kls), kl_dims, len(kls), _abi.ptr_array(glob), glob_n, len(glob),
            scale, K_total or K, 1 if mode_vi else 0, _abi.ptr(ms), _abi.ptr(logp), _abi.ptr(elbo),
            _abi.ptr(self._ticket), _abi.stream_ptr()))
        return elbo[0], logp, ms

    def _build_likelihood(self, zs=None):
        """The VI bound, reference models.py:49-86 (2-D [S*N, D] tiling, mean over S)."""
        S, N = self.num_samples, self.X.shape[0]
        X_tiled = self.X.repeat(S, 1)                                  # :50
        Y_tiled = self.Y.repeat(S, 1)                                  # :51
        XY = torch.cat([X_tiled, Y_tiled], -1)                         # :53
        _, means, covs, kls, kl_types = self.propagate(X_tiled, full_cov=False,
                                                       inference_amorization_inputs=XY,
                                                       is_sampled_local_regularizer=False, zs=zs, _kl_parts=True)
        local_kls = [kl for kl, t in zip(kls, kl_types) if t is RegularizerType.LOCAL]
        global_kls = [kl for kl, t in zip(kls, kl_types) if t is RegularizerType.GLOBAL]
        elbo, _, _ = self._reduce(means[-1], covs[-1], self.Y, local_kls, global_kls, N, S,
                                  stride_b=1, stride_k=N, mode_vi=True)
        return elbo

    def compute_log_likelihood(self, zs=None):
        """gpflow ``Model.compute_log_likelihood`` (reference tests/test_gp_layer.py:50): host float."""
        return float(self._build_likelihood(zs).item())

    likelihood_tensor = property(lambda self: self._build_likelihood())

    def _build_predict(self, X, full_cov=False, zs=None):
        _, means, covs, _, _ = self.propagate(X, full_cov=full_cov, zs=zs)   # :89-91
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
    def _forward_iw(self, zs=None):
        """models.py:113-133.  Default: the tiling of X, Y over K (:113-116) happens inside the first
        layer's kernel (``bcast_K``) and every layer returns marginal variances.  With
        ``full_cov_over_samples`` the reference is followed literally (explicit tiling, full_cov=True,
        matrix_diag_part of the [B, Dy, K, K] covariance)."""
        B, K = self.X.shape[0], self.num_samples
        if self.full_cov_over_samples:
            X_tiled = self.X[:, None, :].expand(B, K, self.X.shape[1]).contiguous()     # :113
            Y_tiled = self.Y[:, None, :].expand(B, K, self.Y.shape[1]).contiguous()     # :114
            XY = torch.cat([X_tiled, Y_tiled], -1)                                       # :116
            samples, means, covs, kls, kl_types = self.propagate(
                X_tiled, full_cov=True, inference_amorization_inputs=XY,
                is_sampled_local_regularizer=True, zs=zs, _kl_parts=True)                # :122-125
        else:
            self.precompute()
            XY_b = self._xy_minibatch()
            zs = [None] * len(self.layers) if zs is None else zs
            if len(zs) != len(self.layers):
                raise ValueError("zs needs one entry per layer")
            samples, means, covs, kls, kl_types = [], [], [], [], []
            F = self.X
            for i, (layer, z) in enumerate(zip(self.layers, zs)):
                kw = dict(_bcast_K=K) if i == 0 else dict(_bcast_XY=K)
                s_, m_, c_, kl_ = layer.propagate(F, full_cov=False, inference_amorization_inputs=XY_b,
                                                  is_sampled_local_regularizer=True, z=z, _precomputed=True,
                                                  _kl_parts=True, **kw)
                samples.append(s_); means.append(m_); covs.append(c_); kls.append(kl_)
                kl_types.append(layer.regularizer_type)
                F = s_
        local_kls = [kl for kl, t in zip(kls, kl_types) if t is RegularizerType.LOCAL]
        global_kls = [kl for kl, t in zip(kls, kl_types) if t is RegularizerType.GLOBAL]
        cov = covs[-1]
        if cov.dim() == 4:                                                            # [B, Dy, K, K]
            cov = torch.diagonal(cov, dim1=-2, dim2=-1).transpose(1, 2).contiguous()  # :133
        return means[-1], cov, local_kls, global_kls, samples, means, covs

    def _xy_minibatch(self):
        """[x_b, y_b] rows of the current minibatch (models.py:116 before tiling), cached per minibatch."""
        key = (self.X.data_ptr(), self.Y.data_ptr(), self.X.shape[0])
        if getattr(self, "_xy_key", None) != key:
            self._xy_cache, self._xy_key = torch.cat([self.X, self.Y], -1).contiguous(), key
        return self._xy_cache

    def _build_likelihood(self, zs=None):
        """The importance-weighted ELBO, reference models.py:112-150."""
        B, K = self.X.shape[0], self.num_samples
        fmean, fvar, local_kls, global_kls, _, _, _ = self._forward_iw(zs)
        elbo, _, _ = self._reduce(fmean, fvar, self.Y, local_kls, global_kls, B, K,
                                  stride_b=K, stride_k=1, mode_vi=False)
        return elbo

    def E_log_p_Y(self, zs=None):
        """Per-point ``logsumexp_k(L_nk) - log K`` [B] (models.py:134-148); name from BASELINE.json."""
        B, K = self.X.shape[0], self.num_samples
        fmean, fvar, local_kls, global_kls, _, _, _ = self._forward_iw(zs)
        _, logp, _ = self._reduce(fmean, fvar, self.Y, local_kls, global_kls, B, K,
                                  stride_b=K, stride_k=1, mode_vi=False)
        return logp

    def lse_partials(self, zs=None, K_total=None):
        """(max_k L, sum_k exp(L - max)) per point [B, 2] + the global KLs: the K-sharded exchange unit."""
        B, K = self.X.shape[0], self.num_samples
        fmean, fvar, local_kls, global_kls, _, _, _ = self._forward_iw(zs)
        _, _, ms = self._reduce(fmean, fvar, self.Y, local_kls, global_kls, B, K, stride_b=K, stride_k=1,
                                mode_vi=False, want_ms=True, K_total=K_total)
        return ms, global_kls
