"""Batched sparse-GP conditionals of the reference's ``dgps_with_iwvi/temp_workaround.py``,
same names and argument meaning, executed by the gfx950 HIP kernels behind include/iwvi_hip.h.

Reference -> here
  independent_multisample_sample_conditional (temp_workaround.py:12-98)  -> same name
  SharedMixedMok                             (temp_workaround.py:107-115) -> same name
  multisample_sample_conditional             (temp_workaround.py:118-161) -> same name
  gauss_kl                                   (temp_workaround.py:167-188) -> same name (KL branch)

Deviations (all documented in DESIGN.md):
  * noise is an explicit optional argument ``z`` (the reference draws tf.random_normal inside the graph);
    z=None draws from the library's Philox stream;
  * ``white=False`` (:63-65; never used by GPLayer, layers.py:42): the second back-substitution is applied once to
    the operands -- f_w = Lm^-1 f, q_sqrt_w = Lm^-1 tril(q_sqrt) (``iwvi_unwhiten``) -- instead of per sample;
  * the full-covariance sample follows the intended ``fmean_SRN1 + chol(fvar) z`` (the reference's
    line :95 has a broadcasting bug);
  * float32 per-sample arithmetic with a float64 factorisation; variances are clamped at 0.
"""
import ctypes
import weakref

import torch

from . import _abi, settings
from .features import InducingPoints, MixedKernelSharedMof
from .kernels import Stationary


class SharedMixedMok:
    """Linear mixing of latent GPs that share one kernel: f = W g, W [P, L] (reference :107-115)."""

    def __init__(self, kernel, W, name=None):
        self.kernel = kernel
        W = torch.as_tensor(W) if not isinstance(W, torch.Tensor) else W
        self.W = W.to(dtype=settings.float_type, device=settings.default_device()).contiguous().clone()
        self.name = name

    def to(self, device):
        self.kernel.to(device)
        self.W = self.W.to(device)
        return self


_STATES = weakref.WeakValueDictionary()      # state buffer address -> its GpState (precompute_states records what each launch prepared)


class GpState:
    """Owner of one layer's per-step factorisation buffer (``iwvi_gp_desc.state``)."""

    def __init__(self, M, R, device):
        self.M, self.R = int(M), int(R)
        nbytes = _abi.lib().iwvi_gp_state_bytes(self.M, self.R)
        if nbytes == 0:
            raise ValueError("bad layer size M=%d R=%d" % (M, R))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        offs = (ctypes.c_size_t * 8)()
        _abi.check(_abi.lib().iwvi_gp_state_offsets(self.M, self.R, offs))
        self.offsets = dict(zip(["Lm", "Linv", "LsP", "LrTP", "QmuP", "ZtP", "cst", "kl"], list(offs)))
        self.Mp = (self.M + 15) // 16 * 16
        self._keep = None
        self._redo_dense = None
        self.f64_prepared = False      # the last precompute of this buffer ran with IWVI_GP_F64_STAGE1 (dense Lm^-1 + plain z~ present)
        _STATES[self.buf.data_ptr()] = self

    def view(self, name, dtype, numel):
        off = self.offsets[name]
        nbytes = numel * torch.empty((), dtype=dtype).element_size()
        return self.buf[off:off + nbytes].view(dtype)

    @property
    def kl_parts(self):
        """[R] float64 device tensor: each latent GP's share of KL[q(u) || p(u)] (last precompute)."""
        return self.view("kl", torch.float64, self.R)

    @property
    def kl(self):
        """0-dim float64 device tensor: KL[q(u) || p(u)] of the last precompute."""
        return self.kl_parts.sum()

    def _dense(self, name):
        """The dense float64 factors are debug / API outputs: the hot path does not write them, so
        reading one re-runs the last precompute with IWVI_GP_WANT_DENSE."""
        if self._redo_dense is None:
            raise RuntimeError("no precompute has run on this state yet")
        d = self._redo_dense
        d.flags |= _abi.GP_WANT_DENSE
        precompute_states([d])
        return self.view(name, torch.float64, self.Mp * self.Mp).view(self.Mp, self.Mp)[:self.M, :self.M]

    @property
    def Lm(self):
        return self._dense("Lm")

    @property
    def Linv(self):
        return self._dense("Linv")

    def desc(self, Z, kern, q_mu, q_sqrt, jitter):
        M, D = Z.shape
        R = q_mu.shape[1]
        if (M, R) != (self.M, self.R):
            raise ValueError("state sized for (M=%d,R=%d), got (M=%d,R=%d)" % (self.M, self.R, M, R))
        d = _abi.GpDesc()
        d.Z, d.lengthscales = Z.data_ptr(), kern.lengthscales.data_ptr()
        d.q_mu, d.q_sqrt, d.state = q_mu.data_ptr(), q_sqrt.data_ptr(), self.buf.data_ptr()
        d.variance, vdev = kern.desc_variance()
        d.variance_dev = vdev
        d.jitter = float(jitter)
        d.M, d.D, d.R, d.kern_type, d.flags = M, D, R, kern.kern_type, 0
        self._keep = (Z, q_mu, q_sqrt, kern.lengthscales)     # keep operands alive until the launch ran
        self._redo_dense = d
        return d


def precompute_states(descs, encs=()):
    """One ``iwvi_model_precompute`` call (one launch per 8 layers) for any number of GP layers, plus the
    encoder MLPs of the model's latent-variable layers in the same launch (``encs``: EncDesc list)."""
    if not descs:
        return
    arr = (_abi.GpDesc * len(descs))(*descs)
    for d in descs:
        st = _STATES.get(d.state)
        if st is not None:
            st.f64_prepared = bool(d.flags & _abi.GP_F64_STAGE1)
    if encs:
        earr = (_abi.EncDesc * len(encs))(*encs)
        _abi.check(_abi.lib().iwvi_model_precompute(arr, len(descs), earr, len(encs), _abi.stream_ptr()))
    else:
        _abi.check(_abi.lib().iwvi_gp_precompute(arr, len(descs), _abi.stream_ptr()))


def _prep_q_sqrt(q_sqrt, f):
    """q_sqrt None | [M,R] diagonal | [R,M,M] -> [R,M,M] (lower triangle is what the kernels read)."""
    M, R = f.shape
    if q_sqrt is None:
        return torch.zeros(R, M, M, dtype=f.dtype, device=f.device)
    if q_sqrt.dim() == 2:                                     # reference :72-73
        return torch.diag_embed(q_sqrt.t().contiguous()).contiguous()
    if q_sqrt.dim() == 3:                                     # reference :74-78
        return q_sqrt.contiguous()
    raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.dim()))   # reference :80-81


def _unwrap_feat(feat):
    if isinstance(feat, InducingPoints):
        return feat.Z
    if isinstance(feat, torch.Tensor):
        return feat
    raise TypeError("feat must be InducingPoints")


def draw_normal(shape, device):
    """N(0,1) draws from the library's counter-based stream (``iwvi_fill_normal``)."""
    out = torch.empty(shape, dtype=settings.float_type, device=device)
    n = out.numel()
    if n:
        _abi.dev_tensor(out, "noise")
        off = settings.next_noise_offset(n)
        _abi.check(_abi.lib().iwvi_fill_normal(_abi.ptr(out), n, settings.seed, off, _abi.stream_ptr()))
    return out


def _layer_flags(f64, M=0, variance=0.0):
    f32 = settings.fw_f32_stage2 or not settings.split16_variance_ok(M, variance)
    return (_abi.LAYER_F32_STAGE2 if f32 else 0) | (_abi.LAYER_F64_STAGE1 if f64 else 0)


def _forward_diag(state, kern, D, R, F2, z2, W, mean_function, want=(True, True, True), f64=False):
    """[T, D] -> sample/mean/var [T, P] through ``iwvi_gp_layer_forward_ex`` (``f64``: the float64 stage-1 route)."""
    T = F2.shape[0]
    P = W.shape[0] if W is not None else R
    dev = F2.device
    outs = [torch.empty(T, P, dtype=settings.float_type, device=dev) if w else None for w in want]
    mf_type, mfA, mfb = _abi.MF_ZERO, None, None
    if mean_function is not None:
        mf_type, mfA, mfb = mean_function.mf_type, mean_function.A, mean_function.b
        if mf_type == _abi.MF_LINEAR:
            if tuple(mfA.shape) != (D, P):
                raise ValueError("Linear mean function A is %s, layer needs (%d, %d)" % (tuple(mfA.shape), D, P))
            _abi.dev_tensor(mfA, "mean_function.A")
    _abi.check(_abi.lib().iwvi_gp_layer_forward_ex(
        _abi.ptr(state.buf), state.M, D, R, P, kern.kern_type, kern.variance,
        _abi.ptr(F2), _abi.ptr(z2), _abi.ptr(W), mf_type, _abi.ptr(mfA), _abi.ptr(mfb),
        _abi.ptr(outs[0]), _abi.ptr(outs[1]), _abi.ptr(outs[2]), T, 1, _layer_flags(f64, state.M, kern.desc_variance()[0]), _abi.stream_ptr()))
    return outs


def _check_common(Xnew, full_output_cov, white, precomputed=False):
    if full_output_cov:
        raise NotImplementedError                              # reference :36-37
    if not white and precomputed:
        raise ValueError("a precomputed state holds whitened operands; white=False needs precomputed=False")
    if Xnew.dim() not in (2, 3):
        raise ValueError("Xnew must be [N, D] or [S, N, D]")


def _factorise(state, Z, kern, f, q_sqrt, white, f64=False):
    """Fill ``state`` for q(u) = (f, q_sqrt).  white=False (reference :63-65): the extra solve with Lm^T is folded
    into the operands once -- f_w = Lm^-1 f, q_sqrt_w[r] = Lm^-1 tril(q_sqrt[r]) -- and the whitened kernels run unchanged."""
    Z = _abi.dev_tensor(Z, "Z")
    q3 = _abi.dev_tensor(_prep_q_sqrt(q_sqrt, f), "q_sqrt")
    d = state.desc(Z, kern, f, q3, settings.jitter_level)
    if f64:
        d.flags |= _abi.GP_F64_STAGE1
    if white:
        precompute_states([d])
        return
    d.flags |= _abi.GP_WANT_DENSE
    precompute_states([d])
    M, R = f.shape
    f_w, q_w = torch.empty_like(f), torch.empty_like(q3)
    _abi.check(_abi.lib().iwvi_unwhiten(_abi.ptr(state.buf), M, R, _abi.ptr(f), _abi.ptr(q3) if q_sqrt is not None else None,
                                        _abi.ptr(f_w), _abi.ptr(q_w) if q_sqrt is not None else None, _abi.stream_ptr()))
    if q_sqrt is None:
        q_w.zero_()
    d = state.desc(Z, kern, f_w, q_w, settings.jitter_level)
    if f64:
        d.flags |= _abi.GP_F64_STAGE1
    precompute_states([d])


def independent_multisample_sample_conditional(Xnew, feat, kern, f, *, full_cov=False, full_output_cov=False,
                                               q_sqrt=None, white=False, z=None, state=None,
                                               mean_function=None, precomputed=False, want_sample=True, f64_stage1=None):
    """Multisample, single-output GP conditional (reference temp_workaround.py:12-98).

    :param Xnew: [S, N, D] (also accepts [N, D], the 2-D ``sample_conditional`` path of :157-161)
    :param f: [M, R];  q_sqrt: [R, M, M], [M, R] or None;  white: whitened representation of q(u) or not
    :return: sample [S,N,R], mean [S,N,R], var [S,N,R] (full_cov=False) or [S,R,N,N] (full_cov=True);
             for 2-D input: [N,R], [N,R], [N,R] | [R,N,N].
    ``state``/``precomputed``/``mean_function`` are used by GPLayer to reuse the per-step factorisation
    and to fuse the mean-function add; plain callers leave them at their defaults.  ``f64_stage1``: K_uf, Lm^-1 k and
    Kdiag - sum A^2 (:44,51,59) in float64 like the reference's float_type (None: ``settings.f64_stage1`` decides -- by default
    for inputs of dimension <= 3).  ``want_sample=False`` (full_cov
    only) returns None for the sample: a TF graph never evaluates an unfetched sample (models.py:89-91 fetches the
    final mean and covariance only), an eager library has to be told.
    """
    _check_common(Xnew, full_output_cov, white, precomputed)
    if not isinstance(kern, Stationary):
        raise TypeError("kern must be a stationary kernel (RBF / Matern52)")
    Z = _unwrap_feat(feat)
    Xnew = _abi.dev_tensor(Xnew.contiguous(), "Xnew")
    f = _abi.dev_tensor(f.contiguous(), "f")
    M, R = f.shape
    D = Xnew.shape[-1]
    if Z.shape != (M, D):
        raise ValueError("feature is %s, expected (%d, %d)" % (tuple(Z.shape), M, D))
    f64 = settings.use_f64_stage1(D, f64_stage1)
    if state is None:
        state = GpState(M, R, Xnew.device)
    if not precomputed:
        _factorise(state, Z, kern, f, q_sqrt, white, f64)
    lead = Xnew.shape[:-1]
    F2 = Xnew.reshape(-1, D)
    T = F2.shape[0]
    if not full_cov:
        z2 = draw_normal((T, R), Xnew.device) if z is None else _abi.dev_tensor(z.reshape(T, R).contiguous(), "z")
        s, m, v = _forward_diag(state, kern, D, R, F2, z2, None, mean_function, f64=f64)
        return s.view(*lead, R), m.view(*lead, R), v.view(*lead, R)
    # full covariance over the second axis (reference :45,56,83,93-96)
    S, N = (1, lead[0]) if Xnew.dim() == 2 else lead
    mean = torch.empty(S, N, R, dtype=settings.float_type, device=Xnew.device)
    cov = torch.empty(S, R, N, N, dtype=settings.float_type, device=Xnew.device)
    ws = torch.empty(_abi.lib().iwvi_gp_fullcov_ws_bytes(T, M, R), dtype=torch.uint8, device=Xnew.device)
    mf_type, mfA, mfb = _abi.MF_ZERO, None, None
    if mean_function is not None:                              # layers.py:46-48, fused: the sample below inherits it
        mf_type, mfA, mfb = mean_function.mf_type, mean_function.A, mean_function.b
        if mf_type == _abi.MF_LINEAR and tuple(mfA.shape) != (D, R):
            raise ValueError("Linear mean function A is %s, layer needs (%d, %d)" % (tuple(mfA.shape), D, R))
        if mf_type == _abi.MF_IDENTITY and D != R:
            raise ValueError("Identity mean function needs D == R")
    _abi.check(_abi.lib().iwvi_gp_layer_fullcov_ex(_abi.ptr(state.buf), M, D, R, kern.kern_type, kern.variance,
                                                  _abi.ptr(F2), S, N, mf_type, _abi.ptr(mfA), _abi.ptr(mfb),
                                                  _abi.ptr(mean), _abi.ptr(cov), _abi.ptr(ws), _layer_flags(f64), _abi.stream_ptr()))
    if not want_sample:
        return (None, mean[0], cov[0]) if Xnew.dim() == 2 else (None, mean, cov)
    zz = draw_normal((S, R, N, 1), Xnew.device) if z is None else _abi.dev_tensor(z.reshape(S, R, N, 1).contiguous(), "z")
    # K11: the jointly Gaussian sample (:93-96).  For the FINAL layer this is dead code for the ELBO (its sample is never
    # consumed, models.py:122-134) and the block may be singular (X tiled over K gives rank-1 blocks): the kernel's
    # factorisation is rounding-tolerant (non-positive pivot -> zero column) where tf.cholesky would fail.  The 2-D
    # gpflow path adds jitter inside its _sample_mvn, the 3-D path of this file (:94-95) does not.
    sample = torch.empty(S, N, R, dtype=settings.float_type, device=Xnew.device)
    nws = _abi.lib().iwvi_mvn_sample_ws_bytes(S, N, R)
    ws2 = torch.empty(nws, dtype=torch.uint8, device=Xnew.device) if nws else None
    _abi.check(_abi.lib().iwvi_mvn_sample(_abi.ptr(mean), _abi.ptr(cov), _abi.ptr(zz), _abi.ptr(sample), S, N, R,
                                          settings.jitter_level if Xnew.dim() == 2 else 0.0, _abi.ptr(ws2),
                                          _abi.stream_ptr()))
    if Xnew.dim() == 2:
        return sample[0], mean[0], cov[0]
    return sample, mean, cov


def multisample_sample_conditional(Xnew, feat, kern, f, *, full_cov=False, full_output_cov=False,
                                   q_sqrt=None, white=False, z=None, state=None, mean_function=None,
                                   precomputed=False, want_sample=True, f64_stage1=None):
    """Dispatcher of reference temp_workaround.py:118-161."""
    if isinstance(kern, SharedMixedMok) and isinstance(feat, MixedKernelSharedMof):      # :123
        _check_common(Xnew, False, white, precomputed)
        base, Z = kern.kernel, _unwrap_feat(feat.feat)
        Xnew = _abi.dev_tensor(Xnew.contiguous(), "Xnew")
        f = _abi.dev_tensor(f.contiguous(), "f")
        M, R = f.shape
        D = Xnew.shape[-1]
        W = _abi.dev_tensor(kern.W, "W")
        if W.shape[1] != R:
            raise ValueError("W is %s but there are %d latent GPs" % (tuple(W.shape), R))
        f64 = settings.use_f64_stage1(D, f64_stage1)
        if state is None:
            state = GpState(M, R, Xnew.device)
        if not precomputed:
            _factorise(state, Z, base, f, q_sqrt, white, f64)
        lead = Xnew.shape[:-1]
        F2 = Xnew.reshape(-1, D)
        T = F2.shape[0]
        # full_cov is forced to False on this branch (reference :125-129, :134-138)
        z2 = draw_normal((T, R), Xnew.device) if z is None else _abi.dev_tensor(z.reshape(T, R).contiguous(), "z")
        s, m, v = _forward_diag(state, base, D, R, F2, z2, W, mean_function, f64=f64)   # mixing fused (:142-145)
        P = W.shape[0]
        return s.view(*lead, P), m.view(*lead, P), v.view(*lead, P)
    assert not isinstance(kern, SharedMixedMok)                                          # :149
    return independent_multisample_sample_conditional(
        Xnew, feat, kern, f, full_cov=full_cov, full_output_cov=full_output_cov, q_sqrt=q_sqrt,
        white=white, z=z, state=state, mean_function=mean_function, precomputed=precomputed, want_sample=want_sample,
        f64_stage1=f64_stage1)


def gauss_kl(q_mu, q_sqrt, K=None):
    """Whitened KL[q(u) || p(u)] summed over outputs (reference :167-188, KL branch :186-188).
    Returns a 0-dim float64 device tensor.  The SGHMC branch (q_sqrt None) is out of scope."""
    if q_sqrt is None or K is not None:
        raise NotImplementedError("only the whitened KL branch is on the IW-ELBO path")
    q_mu = _abi.dev_tensor(q_mu.contiguous(), "q_mu")
    q_sqrt = _abi.dev_tensor(_prep_q_sqrt(q_sqrt, q_mu), "q_sqrt")
    M, R = q_mu.shape
    out = torch.empty(1, dtype=torch.float64, device=q_mu.device)
    _abi.check(_abi.lib().iwvi_gauss_kl(_abi.ptr(q_mu), _abi.ptr(q_sqrt), M, R, _abi.ptr(out), _abi.stream_ptr()))
    return out[0]
