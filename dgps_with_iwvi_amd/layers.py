"""Layer classes of the reference's ``dgps_with_iwvi/layers.py`` with the same constructor and
``propagate`` signatures; parameters are float32 torch tensors on a ROCm device and every
forward runs in the HIP kernels behind include/iwvi_hip.h.

Reference -> here: RegularizerType (layers.py:9-11), GPLayer (:14-50), LatentVariableLayer (:53-105),
Encoder (:108-152).  Extra keyword ``z`` on ``propagate`` injects the N(0,1) draw the reference takes
from ``tf.random_normal`` (temp_workaround.py:89,94; layers.py:86).
"""
import ctypes
import enum

import numpy as np
import torch

from . import _abi, settings
from .features import InducingFeature, InducingPoints, MixedKernelSharedMof
from .mean_functions import Zero
from .temp_workaround import (GpState, SharedMixedMok, draw_normal, gauss_kl,
                              multisample_sample_conditional, precompute_states)


class RegularizerType(enum.Enum):
    LOCAL = 0
    GLOBAL = 1


def _tensor(x, device=None):
    t = torch.as_tensor(np.asarray(x, dtype=np.float32) if not isinstance(x, torch.Tensor) else x)
    return t.to(dtype=settings.float_type, device=device or settings.default_device()).contiguous().clone()


class GPLayer:
    regularizer_type = RegularizerType.GLOBAL

    def __init__(self, kern, Z, num_outputs, mean_function=None, name=None):
        self.num_inducing = len(Z)
        if self.num_inducing > _abi.MAX_M:
            raise ValueError("num_inducing %d > %d" % (self.num_inducing, _abi.MAX_M))
        self.feature = Z if isinstance(Z, InducingFeature) else InducingPoints(Z)
        dev = self._Z().device
        self.q_mu = torch.zeros(self.num_inducing, num_outputs, dtype=settings.float_type, device=dev)
        # full [R, M, M] storage; like gpflow's LowerTriangular transform only the lower band is used
        self.q_sqrt = torch.eye(self.num_inducing, dtype=settings.float_type, device=dev).repeat(num_outputs, 1, 1)
        self.kern = kern.to(dev)
        self.mean_function = (mean_function or Zero()).to(dev)
        self.num_outputs = num_outputs
        self.name = name
        self._state = None
        self.f64_stage1 = None          # None: settings.f64_stage1 decides (auto: input dimension <= 3); True / False: this layer's own choice
        self._f64_measured = None       # models.DGP_VI.autotune_f64: the choice measured on the factor of the current parameters

    def uses_f64_stage1(self):
        """Does this layer take the float64 stage-1 route (include/iwvi_hip.h: IWVI_LAYER_F64_STAGE1)?"""
        return settings.use_f64_stage1(self._Z().shape[1], self.f64_stage1, getattr(self, "_f64_measured", None))

    # -- plumbing -------------------------------------------------------------------------
    def _Z(self):
        f = self.feature.feat if isinstance(self.feature, MixedKernelSharedMof) else self.feature
        return f.Z

    def _base_kern(self):
        return self.kern.kernel if isinstance(self.kern, SharedMixedMok) else self.kern

    def to(self, device):
        self.feature.to(device)
        self.kern.to(device)
        self.mean_function.to(device)
        self.q_mu, self.q_sqrt = self.q_mu.to(device), self.q_sqrt.to(device)
        self._state = None
        return self

    def state(self):
        dev = self.q_mu.device
        if self._state is None or self._state.buf.device != dev or \
                (self._state.M, self._state.R) != (self.num_inducing, self.num_outputs):
            self._state = GpState(self.num_inducing, self.num_outputs, dev)
        return self._state

    def state_dense(self):
        """A second factorisation buffer that holds the DENSE float64 Lm, Lm^-1 for the adjoint (backward.py): it is filled
        on a side stream beside the forward, which reads the first one."""
        dev = self.q_mu.device
        st = getattr(self, "_state_dense", None)
        if st is None or st.buf.device != dev or (st.M, st.R) != (self.num_inducing, self.num_outputs):
            st = self._state_dense = GpState(self.num_inducing, self.num_outputs, dev)
        return st

    def state_desc(self, state=None):
        """``iwvi_gp_desc`` of this layer's current parameters (for a batched precompute); ``state``: the buffer to fill."""
        q_mu = _abi.dev_tensor(self.q_mu.contiguous(), "q_mu")
        q_sqrt = _abi.dev_tensor(self.q_sqrt.contiguous(), "q_sqrt")
        if q_sqrt.shape != (self.num_outputs, self.num_inducing, self.num_inducing):
            raise ValueError("q_sqrt must be [R, M, M], got %s" % (tuple(q_sqrt.shape),))
        d = (state or self.state()).desc(_abi.dev_tensor(self._Z(), "Z"), self._base_kern(), q_mu, q_sqrt,
                                         settings.jitter_level)
        if self.uses_f64_stage1():
            d.flags |= _abi.GP_F64_STAGE1        # the dense float64 Lm^-1 and the plain z~ for the forward's float64 Gram + solve
        return d

    def precompute(self):
        precompute_states([self.state_desc()])

    @property
    def kl(self):
        """KL[q(u)||p(u)] of the last precompute (0-dim float64 device tensor)."""
        return self.state().kl

    def fused_desc(self, z=None, outputs=None, draw=True):
        """``iwvi_layer_desc`` of this layer for ``iwvi_dgp_forward`` (state must be precomputed).
        ``outputs``: optional dict sample/mean/var -> [T, P] tensors;  z: [T, R] noise, or None ->
        drawn inside the kernel (``draw``) / zero."""
        d = _abi.LayerDesc()
        kern = self._base_kern()
        M, D = self._Z().shape
        R = self.num_outputs
        W = _abi.dev_tensor(self.kern.W, "W") if isinstance(self.kern, SharedMixedMok) else None
        P = W.shape[0] if W is not None else R
        if W is not None and W.shape[1] != R:
            raise ValueError("W is %s but there are %d latent GPs" % (tuple(W.shape), R))
        mf = self.mean_function
        d.type, d.state = _abi.LAYER_GP, self.state().buf.data_ptr()
        d.M, d.D, d.R, d.P = M, D, R, P
        d.kern_type, d.mf_type = kern.kern_type, mf.mf_type
        d.variance, d.variance_dev = kern.desc_variance()
        d.flags = (_abi.LAYER_F32_STAGE2 if (settings.fw_f32_stage2 or not settings.split16_variance_ok(M, d.variance)) else 0) | (_abi.LAYER_F64_STAGE1 if self.uses_f64_stage1() else 0)
        if (d.flags & _abi.LAYER_F64_STAGE1) and not self.state().f64_prepared:
            # the float64 Gram + solve read the state's plain z~ and dense Lm^-1, which only a precompute with IWVI_GP_F64_STAGE1 writes
            raise ValueError("layer takes the float64 stage-1 route but its state was last precomputed without IWVI_GP_F64_STAGE1 "
                             "(use GPLayer.state_desc() / precompute(), which set the bit)")
        keep = [W]
        if W is not None:
            d.W = W.data_ptr()
        if mf.mf_type == _abi.MF_LINEAR:
            if tuple(mf.A.shape) != (D, P):
                raise ValueError("Linear mean function A is %s, layer needs (%d, %d)" % (tuple(mf.A.shape), D, P))
            d.mf_A = _abi.dev_tensor(mf.A, "mean_function.A").data_ptr()
            if mf.b is not None:
                d.mf_b = _abi.dev_tensor(mf.b, "mean_function.b").data_ptr()
        if z is not None:
            d.noise = _abi.dev_tensor(z, "z").data_ptr()
            keep.append(z)
        d.zero_noise = 0 if draw else 1
        for k, t in (outputs or {}).items():
            setattr(d, k, t.data_ptr())
        return d, keep

    # -- reference API --------------------------------------------------------------------
    def propagate(self, F, full_cov=False, z=None, _precomputed=False, _kl_parts=False, _want_sample=True, **kwargs):
        """reference layers.py:35-50 -> (samples, mean, cov, kl).  ``_want_sample=False``: the caller does not consume this
        layer's sample (the final layer of a predict call, models.py:89-91), so a full-covariance draw is skipped."""
        if not _precomputed:
            self.precompute()
        samples, mean, cov = multisample_sample_conditional(
            F, self.feature, self.kern, self.q_mu, full_cov=full_cov, q_sqrt=self.q_sqrt, white=True,
            z=z, state=self.state(), mean_function=self.mean_function,      # layers.py:46-48 fused into the kernels
            precomputed=True, want_sample=_want_sample, f64_stage1=self.uses_f64_stage1())
        # layers.py:44 (computed by the precompute); _kl_parts hands the model the R per-GP shares
        # so that the ELBO reduction sums them without an extra launch
        kl = self.state().kl_parts if _kl_parts else self.kl
        return samples, mean, cov, kl


def _activation_code(activation_func):
    """``Encoder(activation_func=...)`` (reference layers.py:109,119: default ``tf.nn.tanh``) -> IWVI_ACT_*.  Accepts None, a
    name, or the torch callable of one of the activations the kernels implement."""
    if activation_func is None:
        return _abi.ACT_TANH
    names = {"tanh": _abi.ACT_TANH, "relu": _abi.ACT_RELU, "sigmoid": _abi.ACT_SIGMOID, "softplus": _abi.ACT_SOFTPLUS,
             "identity": _abi.ACT_IDENTITY, "linear": _abi.ACT_IDENTITY}
    if isinstance(activation_func, str):
        key = activation_func.lower()
    else:
        key = getattr(activation_func, "__name__", type(activation_func).__name__).lower()
    if key in names:
        return names[key]
    raise NotImplementedError("Encoder activation %r: the HIP kernels implement %s" % (activation_func, sorted(set(names))))


class Encoder:
    """MLP [input_dim, *network_dims, 2*latent_dim] with skip connections (reference :108-152), hidden activation
    ``activation_func`` (default tanh like the reference; also relu / sigmoid / softplus / identity, by name or as the torch
    callable).  Evaluated inside ``iwvi_lv_layer_forward``; ``__call__`` runs that kernel in encoder-only mode."""

    def __init__(self, latent_dim, input_dim, network_dims, activation_func=None, name=None):
        # A built-in activation (tanh, relu, sigmoid, softplus, identity: by name or as the torch callable) runs inside the HIP kernels.
        # ANY other callable on torch tensors (reference layers.py:119 takes any TensorFlow op) is honoured too: the MLP is then
        # evaluated with torch ops on the device (``torch_raw``) and handed to the kernels as a precomputed encoder output
        # (iwvi_layer_desc.enc_out -- the route the precompute launch's encoder rows already take), its weight gradients by
        # torch.autograd from the kernels' d(encoder output).  The user's function has to run somewhere; nothing else changes route.
        self.custom_act = None
        try:
            self.act = _activation_code(activation_func)
        except NotImplementedError:
            if not callable(activation_func):
                raise
            self.act, self.custom_act = _abi.ACT_IDENTITY, activation_func
        self.latent_dim = latent_dim
        self.layer_dims = [input_dim, *network_dims, latent_dim * 2]
        if len(self.layer_dims) - 1 > _abi.MAX_ENC:
            raise ValueError("encoder deeper than %d layers" % _abi.MAX_ENC)
        self.Ws, self.bs = [], []
        for din, dout in zip(self.layer_dims[:-1], self.layer_dims[1:]):
            xavier_std = (2. / (din + dout)) ** 0.5                                     # :124
            self.Ws.append(_tensor(np.random.randn(din, dout) * xavier_std))
            self.bs.append(_tensor(np.zeros(dout)))
        self.name = name

    def to(self, device):
        self.Ws = [w.to(device) for w in self.Ws]
        self.bs = [b.to(device) for b in self.bs]
        return self

    _TORCH_ACTS = {_abi.ACT_TANH: torch.tanh, _abi.ACT_RELU: torch.relu, _abi.ACT_SIGMOID: torch.sigmoid,
                   _abi.ACT_SOFTPLUS: torch.nn.functional.softplus, _abi.ACT_IDENTITY: (lambda x: x)}

    def torch_raw(self, Z, Ws=None, bs=None):
        """The MLP of reference layers.py:137-147 with torch ops: [rows, input_dim] -> [rows, 2*latent_dim] = (means | raw), where
        q_sqrt = softplus(raw - 3) (:149-150) is applied by the consumer.  ``Ws`` / ``bs``: stand-ins for the weights (autograd leaves)."""
        act = self.custom_act if self.custom_act is not None else self._TORCH_ACTS[self.act]
        Ws, bs = (self.Ws if Ws is None else Ws), (self.bs if bs is None else bs)
        n = len(bs)
        for i, (W, b, din, dout) in enumerate(zip(Ws, bs, self.layer_dims[:-1], self.layer_dims[1:])):
            Z0 = Z
            Z = Z @ W + b                                        # :141
            if i < n - 1:
                Z = act(Z)                                       # :143-144
            if dout == din:
                Z = Z + Z0                                       # :146-147
        return Z

    def abi_args(self):
        if self.custom_act is not None:
            raise RuntimeError("an Encoder with a custom activation is evaluated by torch_raw(), not inside the kernels")
        Ws = [_abi.dev_tensor(w, "encoder W") for w in self.Ws]
        bs = [_abi.dev_tensor(b, "encoder b") for b in self.bs]
        dims = (ctypes.c_int32 * len(self.layer_dims))(*self.layer_dims)
        return _abi.ptr_array(Ws), _abi.ptr_array(bs), dims, len(Ws), (Ws, bs)

    def __call__(self, Z):
        """-> (q_mu, q_sqrt), each [..., latent_dim]."""
        Z = _abi.dev_tensor(Z.contiguous(), "encoder input")
        lead, T = Z.shape[:-1], Z[..., 0].numel()
        if Z.shape[-1] != self.layer_dims[0]:
            raise ValueError("encoder expects %d features, got %d" % (self.layer_dims[0], Z.shape[-1]))
        Lw = self.latent_dim
        if self.custom_act is not None:
            raw = self.torch_raw(Z.reshape(T, -1))
            return raw[:, :Lw].reshape(*lead, Lw), torch.nn.functional.softplus(raw[:, Lw:] - 3.0).reshape(*lead, Lw)
        dummy = torch.zeros(T, 1, dtype=settings.float_type, device=Z.device)
        mean = torch.empty(T, 1 + Lw, dtype=settings.float_type, device=Z.device)
        cov = torch.empty(T, 1 + Lw, dtype=settings.float_type, device=Z.device)
        Wp, bp, dims, n, keep = self.abi_args()
        _abi.check(_abi.lib().iwvi_lv_layer_forward_act(
            _abi.ptr(dummy), _abi.ptr(Z.reshape(T, -1)), None, Wp, bp, dims, n, self.act, 1, Lw, 0,
            None, _abi.ptr(mean), _abi.ptr(cov), None, T, _abi.stream_ptr()))
        return mean[:, 1:].reshape(*lead, Lw), cov[:, 1:].sqrt().reshape(*lead, Lw)


class LatentVariableLayer:
    regularizer_type = RegularizerType.LOCAL

    def __init__(self, latent_dim, XY_dim=None, encoder=None, name=None):
        self.latent_dim = latent_dim
        if encoder is None:
            assert XY_dim, 'must pass XY_dim or else an encoder'                        # :67
            encoder = Encoder(latent_dim, XY_dim, [20, 20])
        self.encoder = encoder
        self.name = name
        # the reference's placeholders_with_default (layers.py:60-64): in prior mode (no recognition inputs) q_mu / q_sqrt may be
        # FED, e.g. for plotting; None = the defaults 0 / 1.  Tensors broadcastable to [..., latent_dim].
        self.q_mu_placeholder = None
        self.q_sqrt_placeholder = None

    def to(self, device):
        self.encoder.to(device)
        return self

    def enc_desc(self, XY, sample=None):
        """``iwvi_enc_desc``: evaluate this layer's encoder for every row of XY [rows, XY_dim] inside the
        model's precompute launch; the result lands in ``self._enc_out`` [rows, 2*latent_dim].

        ``sample`` = dict(X [rows, Dx], K, sampled_kl, layer_index, seed, rng_state (device address), want_z):
        the same launch also evaluates the layer itself (layers.py:83-103) for K samples of every row, into
        ``self._smp_X`` [rows*K, Dx+latent_dim], ``self._smp_kl`` [rows*K] and, on request, ``self._smp_z``."""
        XY = _abi.dev_tensor(XY.contiguous(), "encoder input")
        if XY.shape[-1] != self.encoder.layer_dims[0]:
            raise ValueError("encoder expects %d features, got %d" % (self.encoder.layer_dims[0], XY.shape[-1]))
        rows = XY.shape[0]
        if getattr(self, "_enc_out", None) is None or self._enc_out.shape[0] != rows or self._enc_out.device != XY.device:
            self._enc_out = torch.empty(rows, 2 * self.latent_dim, dtype=settings.float_type, device=XY.device)
        if self.encoder.custom_act is not None:                  # a user-supplied activation: torch evaluates the MLP, same output buffer
            if sample is not None:
                raise NotImplementedError("the leading-LV-layer-in-the-precompute-launch route needs a built-in encoder activation")
            with torch.no_grad():
                self._enc_out.copy_(self.encoder.torch_raw(XY))
            return None, (XY,)
        Wp, bp, dims, n, k2 = self.encoder.abi_args()
        e = _abi.EncDesc()
        e.XY, e.rows, e.enc_W, e.enc_b, e.dims, e.n_enc = XY.data_ptr(), rows, Wp, bp, dims, n
        e.latent_dim, e.out, e.act = self.latent_dim, self._enc_out.data_ptr(), self.encoder.act
        keep = [XY, Wp, bp, dims, k2]
        if sample is not None:
            X = _abi.dev_tensor(sample["X"].contiguous(), "X")
            if X.shape[0] != rows:
                raise ValueError("X and the encoder input need the same number of rows")
            K, Dx, Lw = int(sample["K"]), X.shape[1], self.latent_dim
            cur = getattr(self, "_smp_X", None)
            if cur is None or tuple(cur.shape) != (rows * K, Dx + Lw) or cur.device != XY.device:
                self._smp_X = torch.empty(rows * K, Dx + Lw, dtype=settings.float_type, device=XY.device)
                self._smp_kl = torch.empty(rows * K, dtype=settings.float_type, device=XY.device)
                self._smp_z = None
            if sample.get("want_z") and self._smp_z is None:
                self._smp_z = torch.empty(rows * K, Lw, dtype=settings.float_type, device=XY.device)
            e.X, e.Dx, e.K = X.data_ptr(), Dx, K
            e.sampled_kl, e.layer_index = 1 if sample.get("sampled_kl", True) else 0, int(sample.get("layer_index", 0))
            e.seed, e.rng_state = int(sample["seed"]), sample["rng_state"]
            e.sample_X, e.sample_kl = self._smp_X.data_ptr(), self._smp_kl.data_ptr()
            e.sample_z = self._smp_z.data_ptr() if sample.get("want_z") else None
            keep.append(X)
        return e, tuple(keep)

    def fused_desc(self, D, z=None, outputs=None, sampled_kl=True, use_encoder=True, draw=True, enc_out=None):
        """``iwvi_layer_desc`` of this layer for ``iwvi_dgp_forward`` (D = width of the incoming F).
        ``enc_out``: precomputed encoder output [rows, 2*latent_dim] (``enc_desc``) instead of the weights."""
        d = _abi.LayerDesc()
        d.type, d.D, d.latent_dim, d.sampled_kl = _abi.LAYER_LV, D, self.latent_dim, 1 if sampled_kl else 0
        keep = []
        if use_encoder and enc_out is not None:
            d.enc_out = _abi.dev_tensor(enc_out, "enc_out").data_ptr()
            keep.append(enc_out)
        elif use_encoder:
            Wp, bp, dims, n, k2 = self.encoder.abi_args()
            d.enc_W, d.enc_b, d.enc_dims, d.n_enc, d.enc_act = Wp, bp, dims, n, self.encoder.act
            keep += [Wp, bp, dims, k2]
        if z is not None:
            d.noise = _abi.dev_tensor(z, "z").data_ptr()
            keep.append(z)
        d.zero_noise = 0 if draw else 1
        for k, t in (outputs or {}).items():
            setattr(d, k, t.data_ptr())
        return d, keep

    def propagate(self, F, inference_amorization_inputs=None, is_sampled_local_regularizer=False,
                  z=None, **kwargs):
        """reference layers.py:72-105 -> (samples, mean, cov, kl) with kl [..., latent_dim]."""
        F = _abi.dev_tensor(F.contiguous(), "F")
        D, Lw, dev = F.shape[-1], self.latent_dim, F.device
        lead = F.shape[:-1]
        T = int(np.prod(lead)) if len(lead) else 1
        XY = inference_amorization_inputs
        if XY is not None:
            XY = _abi.dev_tensor(XY.contiguous(), "inference_amorization_inputs")
            if XY.shape[:-1] != lead:
                raise ValueError("inference_amorization_inputs %s does not match F %s" % (tuple(XY.shape), tuple(F.shape)))
            if XY.shape[-1] != self.encoder.layer_dims[0]:
                raise ValueError("encoder expects %d features, got %d" % (self.encoder.layer_dims[0], XY.shape[-1]))
            XY = XY.reshape(T, -1)
        z2 = draw_normal((T, Lw), dev) if z is None else _abi.dev_tensor(z.reshape(T, Lw).contiguous(), "z")
        outs = [torch.empty(T, D + Lw, dtype=settings.float_type, device=dev) for _ in range(3)]
        kl = torch.empty(T, Lw, dtype=settings.float_type, device=dev)
        if XY is None and (self.q_mu_placeholder is not None or self.q_sqrt_placeholder is not None):
            return self._propagate_fed(F.reshape(T, D), z2, outs, kl, lead, is_sampled_local_regularizer)
        if XY is not None and self.encoder.custom_act is not None:
            with torch.no_grad():
                enc_out = self.encoder.torch_raw(XY).contiguous()
            return self._propagate_fed(F.reshape(T, D), z2, outs, kl, lead, is_sampled_local_regularizer, enc_out=enc_out)
        if XY is None and self.encoder.custom_act is not None:   # prior mode: N(0, 1), as fed defaults
            return self._propagate_fed(F.reshape(T, D), z2, outs, kl, lead, is_sampled_local_regularizer)
        Wp, bp, dims, n, keep = self.encoder.abi_args()
        _abi.check(_abi.lib().iwvi_lv_layer_forward_act(
            _abi.ptr(F.reshape(T, D)), _abi.ptr(XY), _abi.ptr(z2), Wp, bp, dims, n, self.encoder.act, D, Lw,
            1 if is_sampled_local_regularizer else 0,
            _abi.ptr(outs[0]), _abi.ptr(outs[1]), _abi.ptr(outs[2]), _abi.ptr(kl), T, _abi.stream_ptr()))
        s, m, c = (o.view(*lead, D + Lw) for o in outs)
        return s, m, c, kl.view(*lead, Lw)

    def _propagate_fed(self, F2, z2, outs, kl, lead, sampled, enc_out=None):
        """Prior mode with FED q_mu / q_sqrt (the reference's placeholders, layers.py:60-64,78-81): the values travel as a
        precomputed "encoder output" [T, 2*latent_dim] = (q_mu | raw) with q_sqrt = softplus(raw - 3), through the same kernel."""
        T, D = F2.shape
        Lw, dev = self.latent_dim, F2.device
        ft = settings.float_type
        if enc_out is not None:                                  # (an encoder evaluated outside the kernels: custom activation)
            return self._run_pre_encoded(F2, z2, outs, kl, lead, sampled, _abi.dev_tensor(enc_out, "enc_out"))
        mu = torch.zeros(T, Lw, dtype=ft, device=dev) if self.q_mu_placeholder is None else \
            torch.as_tensor(self.q_mu_placeholder, dtype=ft, device=dev).expand(*lead, Lw).reshape(T, Lw)
        sg = torch.ones(T, Lw, dtype=ft, device=dev) if self.q_sqrt_placeholder is None else \
            torch.as_tensor(self.q_sqrt_placeholder, dtype=ft, device=dev).expand(*lead, Lw).reshape(T, Lw)
        if bool((sg <= 0).any()):
            raise ValueError("q_sqrt_placeholder must be positive")
        sg64 = sg.double()
        raw = torch.where(sg64 > 20.0, sg64, torch.log(torch.expm1(sg64))) + 3.0          # softplus^-1, float64 then rounded once
        enc_out = torch.cat([mu, raw.to(ft)], -1).contiguous()
        return self._run_pre_encoded(F2, z2, outs, kl, lead, sampled, enc_out)

    def _run_pre_encoded(self, F2, z2, outs, kl, lead, sampled, enc_out):
        """The layer kernel on a precomputed "encoder output" [T, 2*latent_dim] = (q_mu | raw), q_sqrt = softplus(raw - 3)."""
        T, D = F2.shape
        Lw = self.latent_dim
        d = _abi.LayerDesc()
        d.type, d.D, d.latent_dim, d.sampled_kl = _abi.LAYER_LV, D, Lw, 1 if sampled else 0
        d.enc_out = enc_out.data_ptr()
        d.noise, d.zero_noise = z2.data_ptr(), 1
        d.sample, d.mean, d.var, d.kl_local = (t.data_ptr() for t in (*outs, kl))
        descs = (_abi.LayerDesc * 1)(d)
        _abi.check(_abi.lib().iwvi_dgp_forward(descs, 1, _abi.ptr(F2), D, None, 0, None, 0, T, 1, T, 1.0, 0, None, None, None,
                                              _abi.stream_ptr()))
        s, m, c = (o.view(*lead, D + Lw) for o in outs)
        return s, m, c, kl.view(*lead, Lw)
