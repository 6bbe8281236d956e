"""Backward pass of the IW-ELBO path: host side of ``csrc/backward.hip``.

The reference obtains its gradients from TensorFlow's autodiff of the graph of ``models.py:112-150``
(``experiments/build_models.py:284-304``); here every layer has a hand-written adjoint behind the C-ABI
(``iwvi_gp_layer_backward`` ...), driven layer by layer in reverse.  First version: correct and
deterministic, intermediates in HBM; not fused like the forward yet."""
import ctypes

import torch

from . import _abi, settings
from .layers import GPLayer, SharedMixedMok
from .temp_workaround import precompute_states


_SIDE = {}


def _side_stream(device, which=0):
    """Two extra HIP streams per device: a layer's parameter-gradient branch runs there (as two concurrent chains) beside the
    adjoint of the layer below."""
    # one set per CALLER stream: side streams that took part in a hipGraph capture on one stream are not reused for work
    # forked from another (HIP then reports "capturing stream has unjoined work" for the later capture)
    key = (device.type, device.index, which, torch.cuda.current_stream(device).cuda_stream)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def prepare_stream(device):
    """The stream the adjoint's preparation runs on (dense factors, packed operands) for the calling stream."""
    return _side_stream(device, 2)


class GpSaved:
    """What one GP layer's forward leaves for its adjoint."""
    __slots__ = ("F", "noise", "A", "U", "sample", "mean", "var", "T", "GMV")


def needs_saved_u(layer, T):
    """The adjoint's streaming chain (csrc/backward.hip: k_bw_chain; M <= 512, M and T multiples of 16 -- T of 32 beyond M = 256 --, its tiles within the LDS) works
    from a = Lm^-1 k alone; every other shape takes the GEMM path, which also reads the forward's u_r = L_r^T a.  The library
    decides (``iwvi_gp_layer_backward_needs_u``)."""
    R = layer.num_outputs
    P = layer.kern.W.shape[0] if isinstance(layer.kern, SharedMixedMok) else R
    return bool(_abi.lib().iwvi_gp_layer_backward_needs_u(int(T), layer.num_inducing, layer._Z().shape[1], R, P))


def _words(device):
    return torch.zeros(4, dtype=torch.int64, device=device)


def precompute_dense(layers):
    """One precompute launch for all GP layers with IWVI_GP_WANT_DENSE (the adjoints read the dense Lm, Lm^-1)."""
    descs = []
    for l in layers:
        if isinstance(l, GPLayer):
            d = l.state_desc()
            d.flags |= _abi.GP_WANT_DENSE          # keep IWVI_GP_F64_STAGE1: the forward of a float64-route layer reads this state's z~ image
            descs.append(d)
    precompute_states(descs)


def gp_forward_saved(layer, F, z=None, words=None, precomputed=False):
    """``GPLayer.propagate`` on per-sample rows F [T, D] (marginal variances) that also keeps a = Lm^-1 k,
    u_r = L_r^T a and the draws.  Factorises with IWVI_GP_WANT_DENSE unless ``precomputed`` (``precompute_dense``)."""
    if not isinstance(layer, GPLayer):
        raise TypeError("gp_forward_saved needs a GPLayer")
    F = _abi.dev_tensor(F.contiguous(), "F")
    T, D = F.shape
    dev = F.device
    if not precomputed:
        precompute_dense([layer])
    R, Mp = layer.num_outputs, layer.state().Mp
    P = layer.kern.W.shape[0] if isinstance(layer.kern, SharedMixedMok) else R
    s = GpSaved()
    s.F, s.T, s.GMV = F, T, None
    s.A = torch.empty(T, Mp, dtype=settings.float_type, device=dev)
    s.U = torch.empty(R, T, Mp, dtype=settings.float_type, device=dev) if needs_saved_u(layer, T) else None
    s.noise = torch.empty(T, R, dtype=settings.float_type, device=dev)
    s.sample, s.mean, s.var = (torch.empty(T, P, dtype=settings.float_type, device=dev) for _ in range(3))
    z2 = None if z is None else _abi.dev_tensor(z.reshape(T, R).contiguous(), "z")
    s.GMV = torch.empty(T, 3 * R, dtype=settings.float_type, device=dev)
    outs = dict(sample=s.sample, mean=s.mean, var=s.var, a_out=s.A, noise_out=s.noise, gmv_out=s.GMV)
    if s.U is not None:
        outs["u_out"] = s.U
    ld, keep = layer.fused_desc(z2, outs)
    descs = (_abi.LayerDesc * 1)(ld)
    words = _words(dev) if words is None else words
    _abi.check(_abi.lib().iwvi_dgp_forward(
        descs, 1, _abi.ptr(F), D, None, 0, None, 0, T, 1, T, 1.0,
        settings.seed, ctypes.c_void_p(words.data_ptr() + 8), None, None, _abi.stream_ptr()))
    return s


def _param_desc(layer, dense_state=None):
    """The parameter part of an ``iwvi_gp_bwd_desc`` (all ``iwvi_gp_layer_backward_prepare`` reads) + the tensors to keep alive."""
    M, R = layer.num_inducing, layer.num_outputs
    kern = layer._base_kern()
    b = _abi.GpBwdDesc()
    Z, q_mu, q_sqrt = (_abi.dev_tensor(t.contiguous(), n) for t, n in ((layer._Z(), "Z"), (layer.q_mu, "q_mu"), (layer.q_sqrt, "q_sqrt")))
    b.state = (dense_state or layer.state()).buf.data_ptr()
    b.Z, b.lengthscales = Z.data_ptr(), kern.lengthscales.data_ptr()
    b.q_mu, b.q_sqrt = q_mu.data_ptr(), q_sqrt.data_ptr()
    b.variance, b.variance_dev = kern.desc_variance()
    b.M, b.D, b.R, b.kern_type = M, Z.shape[1], R, kern.kern_type
    b.P = layer.kern.W.shape[0] if isinstance(layer.kern, SharedMixedMok) else R
    b.flags = _abi.BW_F32_CHAIN if settings.bw_f32_chain else 0     # per call, not per process (include/iwvi_hip.h)
    return b, [Z, q_mu, q_sqrt]


def prepare_alloc(model, T):
    """Workspace + dense state per GP layer -> {layer index: (workspace, dense state)} (allocated on the caller's stream BEFORE the
    forward is queued: ``prepare_side`` may then run on another stream ordered only after this point)."""
    dev = model.X.device
    out = {}
    for i, l in enumerate(model.layers):
        if isinstance(l, GPLayer):
            D = l._Z().shape[1]
            ws = torch.empty(_abi.lib().iwvi_gp_layer_backward_ws_bytes(T, l.num_inducing, D, l.num_outputs), dtype=torch.uint8, device=dev)
            out[i] = (ws, l.state_dense())
    return out


def prepare_inline(model, T):
    """The adjoint's operands from the forward's OWN factorisation, on the caller's stream: ``model.precompute(dense="lm")`` (one
    factorisation for both passes, the dense factor written too), ``iwvi_gp_dense_inverse`` (Lm^-1 on nbk CUs per layer) and the
    parameter-only part of every layer's adjoint in one launch.  No second factorisation, no second state buffer, nothing to join:
    the dense factorisation of ``prepare_side`` holds one CU per layer for twice as long as the fast one, and two workgroups of the
    layer kernel -- which needs every CU -- wait for it.  -> {layer index: (workspace, state)} like ``prepare_side``; the caller must
    NOT run ``model.precompute`` again."""
    dev = model.X.device
    gps = [(i, l) for i, l in enumerate(model.layers) if isinstance(l, GPLayer)]
    out = {}
    for i, l in gps:
        D = l._Z().shape[1]
        ws = torch.empty(_abi.lib().iwvi_gp_layer_backward_ws_bytes(T, l.num_inducing, D, l.num_outputs), dtype=torch.uint8, device=dev)
        out[i] = (ws, l.state())
    model.precompute(with_encoders=True, dense="lm")
    if len(gps) <= _abi.MAX_STACK:
        arr = (_abi.GpBwdDesc * len(gps))()
        wsp = (ctypes.c_void_p * len(gps))()
        keeps = []
        for k, (i, l) in enumerate(gps):
            b, keep = _param_desc(l, out[i][1])
            arr[k] = b
            wsp[k] = out[i][0].data_ptr()
            keeps.append(keep)
        _abi.check(_abi.lib().iwvi_gp_layers_backward_prepare(arr, len(gps), T, wsp, _abi.stream_ptr()))
    else:
        for i, l in gps:
            b, keep = _param_desc(l, out[i][1])
            _abi.check(_abi.lib().iwvi_gp_layer_backward_prepare(ctypes.byref(b), T, out[i][0].data_ptr(), _abi.stream_ptr()))
    return out


def prefactor_dense(model, stream, after, skip_q_of=()):
    """On ``stream``, behind the event ``after``: the dense float64 Lm and Lm^-1 of every GP layer into its second state buffer -- the part of
    ``prepare_side`` that depends on Z and the kernel parameters only.  A training step queues it beside the natural-gradient update of its
    first op (whose kernels leave most of the chip idle); the second op's ``prepare_side(dense_ready=...)`` then starts from it.  Layers in
    ``skip_q_of`` get the factorisation only (IWVI_GP_FACTOR_ONLY): their q(u) is being written meanwhile."""
    gps = [(i, l) for i, l in enumerate(model.layers) if isinstance(l, GPLayer)]
    stream.wait_event(after)
    with torch.cuda.stream(stream):
        descs = []
        for i, l in gps:
            d = l.state_desc(state=l.state_dense())
            d.flags |= _abi.GP_WANT_LM | (_abi.GP_FACTOR_ONLY if i in skip_q_of else 0)
            descs.append(d)
        precompute_states(descs)
        arr_d = (_abi.GpDesc * len(descs))(*descs)
        _abi.check(_abi.lib().iwvi_gp_dense_inverse(arr_d, len(descs), _abi.stream_ptr()))


def prepare_side(model, T, stream, out=None, after=None, dense_ready=None, flags=0):
    """On ``stream``, beside the forward: the dense float64 factors of every GP layer (into its second state buffer) and the
    parameter-only part of its adjoint (``iwvi_gp_layer_backward_prepare``: scaled inducing inputs, packed S_r = L_r L_r^T and
    Lm^-T).  -> {layer index: (workspace, dense state)}; the caller joins ``stream`` before the first ``gp_backward``.
    ``out`` = ``prepare_alloc``'s result, ``after`` = an event recorded on the caller's stream right after it: the side work then
    depends on nothing queued later -- the caller can queue its own precompute + forward FIRST, so that in a captured graph they
    continue the caller's hardware queue (the first successor captured does; the other pays ~10 us of cross-queue dispatch)."""
    gps = [(i, l) for i, l in enumerate(model.layers) if isinstance(l, GPLayer)]
    if out is None:
        out = prepare_alloc(model, T)
    if stream != torch.cuda.current_stream():
        if after is not None:
            stream.wait_event(after)
        else:
            stream.wait_stream(torch.cuda.current_stream())
    import os
    # the factorisation leaves only the dense factor, Lm^-1 comes from iwvi_gp_dense_inverse (nbk workgroups per layer): the factorising
    # workgroups then hold their CUs for ~29 us instead of ~52 (two workgroups of the layer kernel wait for them) -- 0.268 -> 0.2635 ms per
    # value + gradient at configs[2] (scripts/ab_bw_env.py, round 6).  IWVI_BW_DENSE=full: the factorising workgroup inverts as well
    lm = os.environ.get("IWVI_BW_DENSE", "lm") == "lm"
    with torch.cuda.stream(stream):
        descs = []
        for i, l in gps:
            d = l.state_desc(state=out[i][1])
            if dense_ready is not None:                          # ``prefactor_dense`` has run on these buffers; only these layers' q(u) moved since
                if i not in dense_ready:
                    continue
                d.flags |= _abi.GP_REUSE_FACTOR
            else:
                d.flags |= _abi.GP_WANT_LM if lm else _abi.GP_WANT_DENSE
            descs.append(d)
        precompute_states(descs)
        if lm and dense_ready is None:
            arr_d = (_abi.GpDesc * len(descs))(*descs)
            _abi.check(_abi.lib().iwvi_gp_dense_inverse(arr_d, len(descs), _abi.stream_ptr()))
        if 1 < len(gps) <= _abi.MAX_STACK:                       # every layer's operands in one launch
            arr = (_abi.GpBwdDesc * len(gps))()
            wsp = (ctypes.c_void_p * len(gps))()
            keeps = []
            for k, (i, l) in enumerate(gps):
                b, keep = _param_desc(l, out[i][1])
                b.flags |= flags
                arr[k] = b
                wsp[k] = out[i][0].data_ptr()
                keeps.append(keep)
            _abi.check(_abi.lib().iwvi_gp_layers_backward_prepare(arr, len(gps), T, wsp, _abi.stream_ptr()))
        else:
            for i, l in gps:
                b, keep = _param_desc(l, out[i][1])
                b.flags |= flags
                _abi.check(_abi.lib().iwvi_gp_layer_backward_prepare(ctypes.byref(b), T, out[i][0].data_ptr(), _abi.stream_ptr()))
    return out


def gp_backward(layer, saved, d_sample=None, d_mean=None, d_var=None, kl_weight=1.0, want_dF=True, side_stream=None, keep=None,
                side_stream2=None, prepared=None, q_only=False, defer_params=False):
    """``iwvi_gp_layer_backward``: upstream gradients [T, P] -> dict(dF [T, D], dZ, dls, dvariance, dq_mu, dq_sqrt).
    ``prepared`` = (workspace, dense state) from ``prepare_side``.  ``q_only``: dict(dq_mu, dq_sqrt) alone -- the library then
    skips everything those two do not need (no prepared operands, no dense factors).
    ``defer_params`` (with side streams): only the per-sample chain is queued now (``desc.phase = 1``: dF is on its way); the
    dict carries ``"_finish"``, a callable that queues the parameter branch (``phase = 2``) on the side streams -- the caller runs
    it after queueing whatever should FOLLOW the chain on the same hardware queue of a captured graph (the next layer's chain)."""
    dev = saved.F.device
    T, D = saved.F.shape
    M, R = layer.num_inducing, layer.num_outputs
    kern = layer._base_kern()
    W = _abi.dev_tensor(layer.kern.W, "W") if isinstance(layer.kern, SharedMixedMok) else None
    P = W.shape[0] if W is not None else R
    ft = settings.float_type
    out = dict(dq_mu=torch.empty(M, R, dtype=ft, device=dev), dq_sqrt=torch.empty(R, M, M, dtype=ft, device=dev))
    if not q_only:
        out.update(dZ=torch.empty(M, D, dtype=ft, device=dev), dls=torch.empty(D, dtype=ft, device=dev),
                   dvariance=torch.empty(1, dtype=ft, device=dev))
        if want_dF:
            out["dF"] = torch.empty(T, D, dtype=ft, device=dev)
        if W is not None:
            out["dW"] = torch.empty(P, R, dtype=ft, device=dev)
        if layer.mean_function.mf_type == _abi.MF_LINEAR:
            out["dmf_A"] = torch.empty(D, P, dtype=ft, device=dev)
    b, keep_t = _param_desc(layer, prepared[1] if prepared else None)
    b.P = P
    b.prepared = 1 if prepared else 0
    mf = layer.mean_function
    b.mf_type = mf.mf_type
    keep_t.append(W)
    if W is not None:
        b.W = W.data_ptr()
    if mf.mf_type == _abi.MF_LINEAR:
        b.mf_A = _abi.dev_tensor(mf.A, "mean_function.A").data_ptr()
    b.F, b.noise, b.A = saved.F.data_ptr(), saved.noise.data_ptr(), saved.A.data_ptr()
    if saved.U is not None:
        b.U = saved.U.data_ptr()
    if saved.GMV is not None:
        b.GMV = saved.GMV.data_ptr()
    for name, t in (("d_sample", d_sample), ("d_mean", d_mean), ("d_var", d_var)):
        if t is not None:
            t = _abi.dev_tensor(t.reshape(T, P).contiguous(), name)
            keep_t.append(t)
            setattr(b, name, t.data_ptr())
    b.kl_weight = float(kl_weight)
    for k, t in out.items():
        setattr(b, k, t.data_ptr())
    ws = prepared[0] if prepared else torch.empty(_abi.lib().iwvi_gp_layer_backward_ws_bytes(T, M, D, R), dtype=torch.uint8, device=dev)
    if side_stream is not None:                                  # parameter gradients beside the next layer's adjoint
        b.side_stream = ctypes.c_void_p(side_stream.cuda_stream)
        if side_stream2 is not None:
            b.side_stream2 = ctypes.c_void_p(side_stream2.cuda_stream)
        keep.append((ws, out, keep_t))                           # alive until the caller has joined the streams
    if defer_params and side_stream is not None and not q_only:
        b.phase = 1
        _abi.check(_abi.lib().iwvi_gp_layer_backward(ctypes.byref(b), T, ws.data_ptr(), _abi.stream_ptr()))
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())

        def finish(stream=None):
            # ``stream``: queue the branch there instead (the caller's own stream, once nothing else is left to queue on it)
            st = side_stream if stream is None else stream
            if st != torch.cuda.current_stream() or stream is None:
                st.wait_event(ev)
            b.phase = 2                                          # (same descriptor: the workspace layout depends on the two side streams)
            b.side_stream = ctypes.c_void_p(st.cuda_stream)
            _abi.check(_abi.lib().iwvi_gp_layer_backward(ctypes.byref(b), T, ws.data_ptr(), ctypes.c_void_p(st.cuda_stream)))
        out["_finish"] = finish
        return out
    _abi.check(_abi.lib().iwvi_gp_layer_backward(ctypes.byref(b), T, ws.data_ptr(), _abi.stream_ptr()))
    return out


def lv_backward(layer, XY, enc_out, eps, dF_next, col0, w, B, K, sampled_kl=True, fused=True):
    """``iwvi_lv_layer_backward`` + ``iwvi_encoder_backward`` -> (dW list, db list) of the layer's encoder.
    ``enc_out`` [B, 2*latent_dim] = (means | raw) as the precompute launch leaves it (``layer._enc_out``)."""
    dev = enc_out.device
    Lw = layer.latent_dim
    ft = settings.float_type
    enc_out = _abi.dev_tensor(enc_out, "enc_out")
    if layer.encoder.custom_act is None and fused:
        # one launch: the encoder's workgroups form the d(means | raw) of their own rows (iwvi_lv_encoder_backward)
        Wp, bp, dims, n, keep = layer.encoder.abi_args()
        dW = [torch.empty_like(t) for t in keep[0]]
        db = [torch.empty_like(t) for t in keep[1]]
        dWp, dbp = _abi.ptr_array(dW), _abi.ptr_array(db)
        ws = torch.empty(_abi.lib().iwvi_encoder_backward_ws_bytes(B, dims, n), dtype=torch.uint8, device=dev)
        XY = _abi.dev_tensor(XY.contiguous(), "encoder input")
        _abi.check(_abi.lib().iwvi_lv_encoder_backward(
            ctypes.c_void_p(enc_out.data_ptr()), ctypes.c_void_p(enc_out.data_ptr() + 4 * Lw), 2 * Lw, 1,
            _abi.ptr(eps), _abi.ptr(dF_next), 0 if dF_next is None else dF_next.shape[1], col0,
            _abi.ptr(w), Lw, B, K, 1 if sampled_kl else 0,
            _abi.ptr(XY), Wp, bp, dims, n, layer.encoder.act, dWp, dbp, ws.data_ptr(), _abi.stream_ptr()))
        return dW, db
    d_enc = torch.empty(B, 2 * Lw, dtype=ft, device=dev)
    _abi.check(_abi.lib().iwvi_lv_layer_backward(
        ctypes.c_void_p(enc_out.data_ptr()), ctypes.c_void_p(enc_out.data_ptr() + 4 * Lw), 2 * Lw, 1,
        _abi.ptr(eps), _abi.ptr(dF_next), 0 if dF_next is None else dF_next.shape[1], col0,
        _abi.ptr(w), Lw, B, K, 1 if sampled_kl else 0, _abi.ptr(d_enc), _abi.stream_ptr()))
    if layer.encoder.custom_act is not None:
        # a user-supplied activation: the encoder is torch ops (layers.Encoder.torch_raw), so are its weight gradients -- the
        # vector-Jacobian product of d(encoder output), which the kernel above formed
        enc = layer.encoder
        Ws = [t.detach().clone().requires_grad_(True) for t in enc.Ws]
        bs = [t.detach().clone().requires_grad_(True) for t in enc.bs]
        with torch.enable_grad():
            raw = enc.torch_raw(_abi.dev_tensor(XY.contiguous(), "encoder input"), Ws, bs)
            gr = torch.autograd.grad(raw, Ws + bs, grad_outputs=d_enc)
        return list(gr[:len(Ws)]), list(gr[len(Ws):])
    Wp, bp, dims, n, keep = layer.encoder.abi_args()
    dW = [torch.empty_like(t) for t in keep[0]]
    db = [torch.empty_like(t) for t in keep[1]]
    dWp, dbp = _abi.ptr_array(dW), _abi.ptr_array(db)
    ws = torch.empty(_abi.lib().iwvi_encoder_backward_ws_bytes(B, dims, n), dtype=torch.uint8, device=dev)
    XY = _abi.dev_tensor(XY.contiguous(), "encoder input")
    _abi.check(_abi.lib().iwvi_encoder_backward_act(_abi.ptr(XY), B, Wp, bp, dims, n, layer.encoder.act, _abi.ptr(d_enc), dWp, dbp,
                                                   ws.data_ptr(), _abi.stream_ptr()))
    return dW, db


def iw_elbo_and_gradients(model, zs=None, mode_vi=None, exchange=None, K_total=None, kl_weight=1.0, overlap=True, wrt="all", fuse_heads=True,
                          prefactor=False, q_moved=None):
    """``wrt="final_q"``: only the final layer's 'l<i>.q_mu' / 'l<i>.q_sqrt' (all that the natural-gradient op of
    build_models.py:288-295 uses): same forward and bound, and of the adjoints only the final layer's two sums over samples.

    ``mode_vi`` (default: the model is a DGP_VI, not a DGP_IWVI): the bound of models.py:49-86 instead -- analytic
    local KL, mean over the S samples; ``zs`` then in that model's layout [S*N, dim] (S-major tiling, models.py:50).

    K-sharded training (``sharding.k_shard_gradients``): ``exchange(ms [B, 2]) -> lse [B]`` turns this rank's per-point
    (max, sum exp) pairs into the logsumexp over ALL ``K_total`` samples of the job; the gradients returned are then
    this rank's share (weights exp(L - lse)), to be SUMMED over the ranks, with the KL terms weighted ``kl_weight``
    (1 / world) so that they count once; the value returned is the job's bound.

    One factorisation per training step (``training.Trainer.step``; build_models.py:288-300 runs the two ops back to back and the first one
    moves nothing but the final layer's q(u)): ``prefactor=True`` with ``wrt="final_q"`` also queues the dense float64 factors of every GP
    layer beside the natural-gradient update the caller queues next: it leaves the event they may start from in ``model._prefactor_after``;
    the caller queues its update, then ``prefactor_dense(model, prepare_stream(device), model._prefactor_after, skip_q_of={final})``, and
    joins that stream.  ``q_moved={final layer index}`` with ``wrt="all"`` is the second half: the
    caller vouches that since that evaluation only these layers' q(u) has changed, so no K_uu is factorised again -- the q(u) images of both
    state buffers are rewritten (IWVI_GP_REUSE_FACTOR), the adjoint's operands packed, and the layer launch follows ~5 us after the op starts.

    The IW-ELBO of the current minibatch (models.py:112-150) and its gradient w.r.t. every parameter the
    reference trains (build_models.py:284-304): -> (elbo [0-dim float64 tensor], dict) with the names of
    oracle/grad_oracle.py: 'l<i>.Z', 'l<i>.ls', 'l<i>.var', 'l<i>.q_mu', 'l<i>.q_sqrt', 'l<i>.W', 'l<i>.mfA' (layers with
    a mixing matrix / linear mean function), 'l<i>.encW<j>', 'l<i>.encb<j>', 'lik_var'.  ``zs``: one noise tensor per layer ([B, K, dim]) or None -> drawn."""
    from .layers import LatentVariableLayer
    from .models import DGP_IWVI
    from .temp_workaround import draw_normal
    dev = model.X.device
    ft = settings.float_type
    B, K = model.X.shape[0], model.num_samples
    T = B * K
    layers = model.layers
    mode_vi = (not isinstance(model, DGP_IWVI)) if mode_vi is None else bool(mode_vi)
    zs = [None] * len(layers) if zs is None else list(zs)
    if len(zs) != len(layers):
        raise ValueError("zs needs one entry per layer")
    if mode_vi:                                                  # [S*N, dim] -> the kernels' point-major order t = n*S + s
        zs = [None if z is None else z.reshape(K, B, -1).transpose(0, 1).contiguous() for z in zs]
    X = _abi.dev_tensor(model.X.contiguous(), "X")
    Y = _abi.dev_tensor(model.Y.contiguous(), "Y")
    for layer in layers:
        if not isinstance(layer, (GPLayer, LatentVariableLayer)):
            raise TypeError("the backward pass knows GPLayer and LatentVariableLayer")
    if not mode_vi and getattr(model, "_joint_over_samples", lambda: False)():
        raise NotImplementedError("an inner GPLayer with a plain (non-SharedMixedMok) kernel draws its K samples jointly "
                                  "(temp_workaround.py:149-155); the hand-written adjoints cover the marginal-sampling stacks only")
    n_lv = sum(isinstance(l, LatentVariableLayer) for l in layers)
    if n_lv > 2:
        raise NotImplementedError("the backward pass reads the encoder outputs of the precompute launch, which evaluates at most "
                                  "2 latent-variable layers (%d in this model)" % n_lv)
    has_lv = any(isinstance(l, LatentVariableLayer) for l in layers)
    XY = model._xy_minibatch() if has_lv else None
    # beside the forward, on its own stream: dense float64 factors + the parameter-only part of every layer's adjoint
    cur = torch.cuda.current_stream()
    import os
    if wrt not in ("all", "final_q"):
        raise ValueError("wrt is 'all' or 'final_q'")
    final_q = wrt == "final_q"
    prep_stream = _side_stream(dev, 2) if (overlap and not final_q) else cur
    # (queued BEFORE the caller's own precompute: queued after it -- ordered by an event only -- the dense factorisation lands beside the
    # layer kernel instead, whose workgroups then wait for its two CUs: 0.355 -> 0.382 ms at configs[2])
    n_gp = sum(isinstance(l, GPLayer) for l in layers)
    inline = (not final_q) and n_gp <= _abi.MAX_STACK and os.environ.get("IWVI_BW_PREPARE") == "inline"
    if q_moved is not None and (final_q or not overlap or inline):
        raise ValueError("q_moved goes with wrt='all', overlap=True and the side-stream preparation")
    if q_moved is not None:
        # the caller's own (short) precompute + layer launch are captured FIRST: behind the previous op's last node the first successor
        # captured continues its hardware queue, the preparation -- 12 us of work, needed 50 us later -- takes the cross-queue dispatch
        q_moved = set(q_moved)
        alloc = prepare_alloc(model, T)
        after = torch.cuda.Event()
        after.record(cur)
        model.precompute(with_encoders=True, q_moved=q_moved)
        # (the dense state's q(u) images are NOT refreshed: the adjoint reads nothing of them but the split-f16 scale of S_r = L_r L_r^T, which
        #  the packing takes from L_r itself here (IWVI_BW_OWN_QSCALE) -- the preparation is k_prepare_all alone, ordered behind the start
        #  of this op only and over long before the layer launch ends: no cross-queue join in front of the first chain)
        prepared = prepare_side(model, T, prep_stream, out=alloc, after=after, dense_ready=set(), flags=_abi.BW_OWN_QSCALE)
    elif inline:
        # IWVI_BW_PREPARE=inline: one factorisation for both passes (prepare_inline).  Measured, not the default: the inversion launch sits
        # in front of the layer kernel (17 us at M = 128) where the side stream's dense factorisation costs the layer kernel 7 us and a
        # join 10 -- configs[2] 0.318 vs 0.309 ms per value + gradient, configs[3] 8.98 vs 8.90 ms
        prepared = prepare_inline(model, T)
        prep_stream = cur
    else:
        prepared = {} if final_q else prepare_side(model, T, prep_stream)
        # forward: one factorisation launch (packed operands, encoders) + ONE fused layer launch that also leaves what the
        # adjoints need in HBM (a = Lm^-1 k, the draws, every layer's output rows)
        model.precompute(with_encoders=True)
    zflat = [None if z is None else z.reshape(T, -1) for z in zs]
    Dy = Y.shape[1]
    w = torch.empty(T, dtype=ft, device=dev)
    d_mean, d_var = torch.empty(T, Dy, dtype=ft, device=dev), torch.empty(T, Dy, dtype=ft, device=dev)
    sums = torch.empty(3, dtype=torch.float64, device=dev)
    # The heads of the bound's adjoint (w, d / d final mean and variance, the sums) come out of the layer launch's own tail when the
    # bound is importance-weighted, unsharded, and every point's K samples sit in one chunk of the launch (include/iwvi_hip.h:
    # iwvi_elbo_desc.adj_*): two launches less in front of the first chain.  The library refuses (before launching anything) when the
    # launch's chunk does not hold whole points; the separate iwvi_iw_elbo_backward below then does it.
    fused_heads = False
    if (not mode_vi) and exchange is None and (K_total is None or int(K_total) == K) and fuse_heads:
        try:
            # (wrt = "final_q", the natural-gradient op: with the heads fused, of everything the launch can leave in HBM only the final
            #  layer's a, draws and latent moments are read -- no per-layer rows, nothing of the inner layers: ~25 MB of stores less at configs[2])
            slim = wrt == "final_q"
            _, outs, _ = model._fused_forward(T, K, B, (T,), zs=zflat, sampled_kl=True, want_layers=True, want_logw=True, want_saved=True,
                                              elbo=dict(B=B, K=K, stride_b=K, stride_k=1, mode_vi=False,
                                                        adj=dict(w=w, d_mean=d_mean, d_var=d_var, sums=sums)),
                                              outputs_for={len(layers) - 1} if slim else None, moments=not slim)
            fused_heads = True
        except _abi.IwviError as e:
            if e.rc != _abi.ERR_UNSUPPORTED:                     # the library's "this stack cannot fuse the heads": the two-launch form below
                raise
    if not fused_heads:
        _, outs, _ = model._fused_forward(T, K, B, (T,), zs=zflat, sampled_kl=not mode_vi, want_layers=True, want_logw=False,
                                          want_saved=True)
    # (the layer launch itself advances the device-resident noise counter: the next evaluation draws fresh noise)
    saved = []
    F = None
    slim_q = fused_heads and wrt == "final_q"                    # only the final layer left anything (its adjoint reads a, noise, latent moments)
    for i, (layer, o) in enumerate(zip(layers, outs)):
        if slim_q:
            if i < len(layers) - 1:
                saved.append(("skipped",))
                continue
            s = GpSaved()
            s.F = torch.empty(T, layer._Z().shape[1], dtype=ft, device=dev)        # (shape only: the q-only adjoint does not read the inputs)
            s.T, s.A, s.U, s.noise, s.GMV = T, o["a_out"], o.get("u_out"), o["noise_out"], o["gmv_out"]
            s.sample = s.mean = s.var = None
            saved.append(("gp", s))
            continue
        if F is None and isinstance(layer, GPLayer):                              # first layer: the tiled inputs (models.py:113)
            F = X[:, None, :].expand(B, K, X.shape[1]).reshape(T, -1).contiguous()
        if isinstance(layer, LatentVariableLayer):
            D_in = X.shape[1] if F is None else F.shape[1]
            saved.append(("lv", layer._enc_out, o["noise_out"], o["kl_local"], D_in))
        else:
            s = GpSaved()
            s.F, s.T, s.A, s.U, s.noise, s.GMV = F, T, o["a_out"], o.get("u_out"), o["noise_out"], o["gmv_out"]
            s.sample, s.mean, s.var = o["sample"], o["mean"], o["var"]
            saved.append(("gp", s))
        F = o["sample"]
    if saved[-1][0] != "gp":
        raise ValueError("the last layer must be a GPLayer")
    fin = saved[-1][1]
    kls = [s[3] for s in saved if s[0] == "lv"]                  # (only the unfused heads read them)
    klp = _abi.ptr_array(kls)
    kld = (ctypes.c_int32 * max(len(kls), 1))(*[k.shape[1] for k in kls])
    glob = [_abi.dev_tensor(g.reshape(-1), "global kl", torch.float64) for g in model._global_kls()]
    glob_p = _abi.ptr_array(glob)
    glob_n = (ctypes.c_int32 * max(len(glob), 1))(*[g.numel() for g in glob])
    ws = torch.empty(2 * B, dtype=torch.float64, device=dev)
    scale = float(model.num_data) / float(B)
    lse_g = None
    if exchange is not None:
        if mode_vi:
            raise ValueError("the K-sharded exchange is for the importance-weighted bound")
        _, _, ms = model._reduce(fin.mean, fin.var, Y, kls, [], B, K, stride_b=K, stride_k=1, mode_vi=False, want_ms=True)
        from . import sharding
        if sharding._RECORDER is not None and prep_stream != cur:
            cur.wait_stream(prep_stream)                         # a segmented capture cuts the graph at the exchange: no forked work may be left open across the cut
        lse_g = _abi.dev_tensor(exchange(ms).to(ft).contiguous(), "lse_global")
    lik_host, lik_dev = model.likelihood.desc_variance()
    if not fused_heads:
        _abi.check(_abi.lib().iwvi_iw_elbo_backward_dev(
            _abi.ptr(fin.mean), _abi.ptr(fin.var), _abi.ptr(Y), Dy, klp, kld, len(kls), B, K,
            lik_host, lik_dev, scale, 1 if mode_vi else 0, _abi.ptr(w), _abi.ptr(d_mean), _abi.ptr(d_var),
            glob_p, glob_n, len(glob), _abi.ptr(lse_g), int(K_total or K),
            ctypes.c_void_p(sums.data_ptr()), ctypes.c_void_p(ws.data_ptr()), _abi.stream_ptr()))
    grads = {"lik_var": sums[1]}
    elbo = sums[2]                                               # scale * sum_n(...) - sum of the global KLs, formed on the device
    if final_q:
        i = len(layers) - 1
        g = gp_backward(layers[i], fin, d_mean=d_mean, d_var=d_var, kl_weight=kl_weight, q_only=True)
        if prefactor:                                            # the point the dense factors may start from; the CALLER queues them (prefactor_dense)
            done = torch.cuda.Event()                            # AFTER its update: the first successor captured behind these launches keeps their
            done.record(cur)                                     # hardware queue, and that has to be the update, not the side work
            model._prefactor_after = done
        return elbo, {"l%d.q_mu" % i: g["dq_mu"], "l%d.q_sqrt" % i: g["dq_sqrt"]}
    if prep_stream != cur:
        cur.wait_stream(prep_stream)                             # dense factors and packed adjoint operands are ready
    side = _side_stream(dev) if overlap else None
    held = []
    dF = None
    pending = None                                               # the parameter branch of the layer above, queued once this layer's chain is
    # the LAST branch to be queued (the lowest GP layer that has one) runs as two chains: the adjoint of the factorisation on the side
    # stream, every other sum on the caller's own stream -- nothing else is left to run there (0.364 -> 0.350 ms at configs[2])
    deferred = [j for j, l in enumerate(layers) if isinstance(l, GPLayer) and j > 0]
    last_deferred = min(deferred) if (deferred and overlap) else -1
    # With two or more branches: a chain kernel fills every CU, so the branch of the layer above the lowest one does not get to run beside
    # the lowest layer's chain anyway -- on the side stream it ends up IN FRONT of the lowest layer's branch, one after the other.  It is
    # therefore queued on the caller's stream behind everything else there, and the lowest layer's branch runs whole on the side stream:
    # the two branches then run side by side (configs[2]: 0.331 -> 0.307 ms per value + gradient).  IWVI_BW_BRANCH_ORDER=old: both on the side stream.
    # (M <= 128 only: at M = 256 the branches are GEMM-sized and do overlap the chains -- configs[3]: 8.88 ms this way round, 9.09 ms the other)
    # (Round 6, measured and not kept: every branch but the lowest queued on the side stream right behind its own chain, the lowest as
    # two chains -- 0.268 -> 0.281 ms: the first successor captured behind a chain takes over its hardware queue, so the NEXT chain pays
    # the cross-queue dispatch; and of a branch only its reduction runs beside a chain kernel (k_gl_tril does not fit next to a chain
    # workgroup's LDS, the float64 products need 304 registers a wave): the rest queues up in front of the lowest branch.)
    on_cur = sorted(deferred)[1] if (len(deferred) >= 2 and overlap and os.environ.get("IWVI_BW_BRANCH_ORDER") != "old"
                                     and layers[sorted(deferred)[1]].num_inducing <= 128) else -1
    finish_on_cur = None
    for i in range(len(layers) - 1, -1, -1):
        layer, s = layers[i], saved[i]
        if s[0] == "gp":
            last = i == len(layers) - 1
            g = gp_backward(layer, s[1], d_sample=None if last else dF, d_mean=d_mean if last else None,
                            d_var=d_var if last else None, kl_weight=kl_weight, want_dF=i > 0,
                            side_stream=side if i > 0 else None, keep=held, prepared=prepared.get(i), defer_params=True,
                            # (one side stream: with the chains of consecutive layers back to back on the caller's stream -- defer_params --
                            # a second one for the Cholesky-adjoint chain no longer pays: 0.400 -> 0.369 ms at configs[2] without it; alternating
                            # the layers' branches between two side streams: 0.356 -> 0.379 ms)
                            side_stream2=cur if (i == last_deferred and on_cur < 0) else None)
            for k_out, k_name in (("dZ", "Z"), ("dls", "ls"), ("dvariance", "var"), ("dq_mu", "q_mu"), ("dq_sqrt", "q_sqrt"),
                                  ("dW", "W"), ("dmf_A", "mfA")):
                if k_out in g:
                    grads["l%d.%s" % (i, k_name)] = g[k_out]
            dF = g.get("dF")
            if i == on_cur:
                finish_on_cur = g.pop("_finish", None)
                if pending is not None:
                    pending()
                pending = None
                continue
            if pending is not None:
                pending()
            pending = g.pop("_finish", None)
        else:
            _, enc_out, eps, _, D_in = s
            dW, db = lv_backward(layer, XY, enc_out, eps, dF, D_in, w, B, K, not mode_vi)
            for j, (a, b) in enumerate(zip(dW, db)):
                grads["l%d.encW%d" % (i, j)], grads["l%d.encb%d" % (i, j)] = a, b
            dF = None if (dF is None or i == 0) else dF[:, :D_in].contiguous()
    if pending is not None:
        pending()
    if finish_on_cur is not None:
        finish_on_cur(cur)
    if side is not None:
        cur.wait_stream(side)                                    # join: the parameter gradients are complete on the caller's stream
    del held
    return elbo, grads


def parameter_list(model):
    """[(name, tensor)] of the model's tensor parameters, named like the gradients (host-scalar parameters -- kernel
    and likelihood variances -- are not tensors and are left to ``training.Trainer``)."""
    from .layers import LatentVariableLayer
    out = []
    for i, l in enumerate(model.layers):
        if isinstance(l, LatentVariableLayer):
            for j, (w, b) in enumerate(zip(l.encoder.Ws, l.encoder.bs)):
                out += [("l%d.encW%d" % (i, j), w), ("l%d.encb%d" % (i, j), b)]
        elif isinstance(l, GPLayer):
            out += [("l%d.Z" % i, l._Z()), ("l%d.ls" % i, l._base_kern().lengthscales), ("l%d.q_mu" % i, l.q_mu), ("l%d.q_sqrt" % i, l.q_sqrt)]
            if isinstance(l.kern, SharedMixedMok):
                out.append(("l%d.W" % i, l.kern.W))
            if l.mean_function.mf_type == _abi.MF_LINEAR:
                out.append(("l%d.mfA" % i, l.mean_function.A))
    return out


class IwElbo(torch.autograd.Function):
    """The IW-ELBO as a differentiable torch op: ``IwElbo.apply(model, zs, *tensors)`` with
    ``tensors = [t for _, t in parameter_list(model)]`` -- the model's OWN parameter tensors (the kernels read the
    model; the arguments only tell autograd where the gradients go).  Forward and backward are the hand-written HIP
    kernels (``iw_elbo_and_gradients``); nothing is traced."""

    @staticmethod
    def forward(ctx, model, zs, *tensors):
        params = parameter_list(model)
        if len(tensors) != len(params) or any(a.data_ptr() != b.data_ptr() for a, (_, b) in zip(tensors, params)):
            raise ValueError("pass the model's own parameter tensors, in parameter_list(model) order")
        elbo, grads = iw_elbo_and_gradients(model, zs)
        ctx.grads = [grads[n].reshape(t.shape).to(t.dtype) for n, t in params]
        return elbo.clone()

    @staticmethod
    def backward(ctx, g):
        return (None, None) + tuple((g * gr.to(torch.float64)).to(gr.dtype) for gr in ctx.grads)
