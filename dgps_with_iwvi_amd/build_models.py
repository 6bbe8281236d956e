"""Model factory: the layer stack of the reference's experiments from a ``configuration`` string
(reference ``experiments/build_models.py:176-268``; SURVEY.md section 8 row F4 -- host code around the hot path).

``build_model(ARGS, X, Y)`` with ``ARGS.mode`` in {'VI', 'IWAE'}, ``ARGS.configuration`` such as ``'L1_G5'`` or
``'G5_G5'`` ('G<r>': a GP layer of r latent GPs mixed to the input width, 'L<d>': a latent-variable layer of d
dimensions), ``ARGS.M`` inducing points, ``ARGS.likelihood_variance``, ``ARGS.minibatch_size``,
``ARGS.num_IW_samples``.  The training op the reference attaches (``model.train_op``: natural gradient + Adam,
:270-304) is ``attach_train_op`` over ``training.Trainer`` (row F1); SGHMC and the CVAE baseline are out of scope
and raise.

Initial values follow the reference: first-layer inducing inputs by k-means of X (or X padded with N(0,1) rows when
N <= M), deeper layers N(0,1) with their first columns taken from them (:218-219, :239-240); RBF-ARD kernels with
lengthscales sqrt(D_in), variance 1 (:212, :238); mixing W = leading right-singular vectors of X (:186, :216-217);
identity-padded linear mean function (:205-208); inner-layer q_sqrt scaled by 1e-5 (:276-278).
"""
import numpy as np
import torch
from scipy.cluster.vq import kmeans2

from . import settings
from .features import InducingPoints, MixedKernelSharedMof
from .kernels import RBF
from .layers import GPLayer, LatentVariableLayer
from .likelihoods import Gaussian
from .mean_functions import Linear
from .models import DGP_IWVI, DGP_VI
from .temp_workaround import SharedMixedMok


def parse_configuration(configuration):
    """'L1_G5_G5' -> [('L', 1), ('G', 5), ('G', 5)]; '' -> [] (a single GP layer: SVGP)."""
    out = []
    for tok in (configuration.split("_") if configuration else []):
        if len(tok) < 2 or tok[0] not in "GL" or not tok[1:].isdigit():
            raise ValueError("configuration token %r: expected G<num_gps> or L<latent_dim>" % tok)
        out.append((tok[0], int(tok[1:])))
    return out


def build_layers(configuration, X, M, rng=None):
    """The reference's layer stack (everything but the model object): list of layers, on the CPU."""
    rng = np.random if rng is None else rng
    X = np.asarray(X, dtype=np.float64)
    N, D = X.shape
    if N > M:
        Z = kmeans2(X, M, minit="points")[0]
    else:
        Z = np.concatenate([X.copy(), rng.randn(M - N, D)], 0)
    P = np.linalg.svd(X, full_matrices=False)[2]
    DX, DY = D, 1
    D_in = D_out = D
    layers = []
    for kind, d in parse_configuration(configuration):
        if kind == "G":
            A = np.zeros((D_in, D_out))
            k = min(D_in, D_out)
            A[:k, :k] = np.eye(k)
            W = np.zeros((D_out, d))
            W[:, :min(d, DX)] = P[:, :min(d, DX)]
            ZZ = rng.randn(M, D_in)
            ZZ[:, :min(D_in, DX)] = Z[:, :min(D_in, DX)]
            kern = SharedMixedMok(RBF(D_in, lengthscales=float(D_in) ** 0.5, variance=1.0, ARD=True), W=W)
            layer = GPLayer(kern, MixedKernelSharedMof(InducingPoints(ZZ)), d, mean_function=Linear(A=A))
            layer.q_sqrt = layer.q_sqrt * 1e-5                       # inner layers start nearly deterministic
            layers.append(layer)
            D_in = D_out
        else:
            D_in += d
            layers.append(LatentVariableLayer(d, XY_dim=DX + 1))
    ZZ = rng.randn(M, D_in)
    ZZ[:, :min(D_in, DX)] = Z[:, :min(D_in, DX)]
    layers.append(GPLayer(RBF(D_in, lengthscales=float(D_in) ** 0.5, variance=1.0, ARD=True), InducingPoints(ZZ), DY))
    return layers


def build_model(ARGS, X, Y, apply_name=True, device=None):
    if ARGS.mode not in ("VI", "IWAE"):
        raise NotImplementedError("mode %r: the VI / IWAE models (with their NatGrad + Adam train_op) are built here; SGHMC "
                                  "and the CVAE baseline are outside the scope of this port (SURVEY.md section 8)" % (ARGS.mode,))
    layers = build_layers(ARGS.configuration, X, ARGS.M)
    lik = Gaussian(ARGS.likelihood_variance)
    name = "Model" if apply_name else None
    if ARGS.mode == "VI":
        model = DGP_VI(X, Y, layers, lik, minibatch_size=ARGS.minibatch_size, name=name)
    else:
        model = DGP_IWVI(X, Y, layers, lik, minibatch_size=ARGS.minibatch_size,
                         num_samples=ARGS.num_IW_samples, name=name)
    model = model.to(device or settings.default_device())
    if model.X.is_cuda and settings.f64_stage1 == "auto" and getattr(ARGS, "autotune_f64", True):
        # the float64 stage-1 route per layer from the measured diag(Lm) ratio of the initial values (k-means inducing inputs can
        # cluster: an 8-D layer is not well-conditioned by its dimension alone); the Trainer repeats it at every staircase epoch
        model.f64_route_report = model.autotune_f64()
    attach_train_op(model, ARGS)
    return model


def attach_train_op(model, ARGS):
    """``model.train_op()`` / ``model.global_step`` as built by the reference (build_models.py:270-304): NatGrad on the
    final layer's q(u) then Adam on the rest, lr / gamma with a staircase decay.  The optimiser state lives in a
    ``training.Trainer`` created at the first call (it needs the ROCm device).  Modes 'IWAE' (models.py:112-150) and
    'VI' (:49-86: uniform sample weights, analytic local KL)."""
    state = {}

    def trainer():
        if "t" not in state:
            from .training import Trainer
            state["t"] = Trainer(model, lr=getattr(ARGS, "lr", 5e-3), gamma=getattr(ARGS, "gamma", 1e-2),
                                 lr_decay=getattr(ARGS, "lr_decay", 0.98), gamma_decay=getattr(ARGS, "gamma_decay", 0.98),
                                 fix_linear=getattr(ARGS, "fix_linear", True),
                                 # every op of a step replayed from a hipGraph (the reference's session.run of a prebuilt graph, in effect)
                                 use_graph=bool(getattr(ARGS, "use_graph", True)))
        return state["t"]

    model.trainer = trainer
    model.train_op = lambda: trainer().step()
    return model


# ---- checkpoint / resume of the parameters (reference: gpflow Saver, run_conditional_density_estimation.py:95-125) ----
def state_dict(model):
    """Every parameter tensor of the stack, as CPU float arrays keyed 'layers.<i>.<name>'."""
    out = {"likelihood.variance": np.float64(model.likelihood.variance)}
    for i, layer in enumerate(model.layers):
        p = "layers.%d." % i
        if isinstance(layer, GPLayer):
            kern = layer._base_kern()
            out[p + "Z"] = layer._Z().detach().cpu().numpy()
            out[p + "lengthscales"] = torch.as_tensor(kern.lengthscales).detach().cpu().numpy()
            out[p + "variance"] = np.float64(kern.variance)
            out[p + "q_mu"] = layer.q_mu.detach().cpu().numpy()
            out[p + "q_sqrt"] = layer.q_sqrt.detach().cpu().numpy()
            if isinstance(layer.kern, SharedMixedMok):
                out[p + "W"] = layer.kern.W.detach().cpu().numpy()
            if layer.mean_function.A is not None:                     # Linear: trainable when fix_linear=False (:224-227)
                out[p + "mf_A"] = layer.mean_function.A.detach().cpu().numpy()
                out[p + "mf_b"] = layer.mean_function.b.detach().cpu().numpy()
        elif isinstance(layer, LatentVariableLayer) and layer.encoder is not None:
            for j, (w, b) in enumerate(zip(layer.encoder.Ws, layer.encoder.bs)):
                out[p + "enc_W%d" % j] = w.detach().cpu().numpy()
                out[p + "enc_b%d" % j] = b.detach().cpu().numpy()
    # the session state a tf.train.Saver restores besides the variables: where the shuffled minibatch iterator stands
    # (gpflow.Minibatch(seed=0), models.py:25-26) and the noise streams' counters
    kind, keys, pos, has_gauss, cached = model._mb_rng.get_state()
    out["minibatch.rng_keys"], out["minibatch.rng_pos"] = np.asarray(keys), np.int64(pos)
    out["minibatch.pos"] = np.int64(model._mb_pos)
    if model._mb_perm is not None:
        out["minibatch.perm"] = model._mb_perm.detach().cpu().numpy()
    if model._dev_words is not None:
        out["rng.step"] = np.int64(int(model._dev_words[1].item()))
    out["rng.seed"], out["rng.offset"] = np.int64(settings.seed), np.int64(settings._offset)
    return out


def load_state_dict(model, state):
    """In place (``copy_``) wherever the stored array has the parameter's shape, so that a ``training.Trainer`` built on the
    model keeps pointing at live tensors; a parameter of another shape is replaced."""
    dev = model.X.device
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=settings.float_type, device=dev)

    def put(obj, name, value, index=None):
        cur = getattr(obj, name) if index is None else getattr(obj, name)[index]
        new = t(value)
        if isinstance(cur, torch.Tensor) and cur.shape == new.shape and cur.device == new.device:
            with torch.no_grad():
                cur.copy_(new)
        elif index is None:
            setattr(obj, name, new)
        else:
            getattr(obj, name)[index] = new

    model.likelihood.variance = float(state["likelihood.variance"])
    for i, layer in enumerate(model.layers):
        p = "layers.%d." % i
        if isinstance(layer, GPLayer):
            kern = layer._base_kern()
            with torch.no_grad():
                layer._Z().copy_(t(state[p + "Z"]))
            put(kern, "lengthscales", state[p + "lengthscales"])
            kern.variance = float(state[p + "variance"])
            put(layer, "q_mu", state[p + "q_mu"])
            put(layer, "q_sqrt", state[p + "q_sqrt"])
            if isinstance(layer.kern, SharedMixedMok):
                put(layer.kern, "W", state[p + "W"])
            if layer.mean_function.A is not None and p + "mf_A" in state:
                put(layer.mean_function, "A", state[p + "mf_A"])
                put(layer.mean_function, "b", state[p + "mf_b"])
            layer._state = None                                       # factorisation depends on Z / lengthscales
        elif isinstance(layer, LatentVariableLayer) and layer.encoder is not None:
            for j in range(len(layer.encoder.Ws)):
                put(layer.encoder, "Ws", state[p + "enc_W%d" % j], j)
                put(layer.encoder, "bs", state[p + "enc_b%d" % j], j)
    if "minibatch.rng_keys" in state:                                 # resume the batch order and the noise streams
        model._mb_rng.set_state(("MT19937", np.asarray(state["minibatch.rng_keys"], dtype=np.uint32),
                                 int(state["minibatch.rng_pos"]), 0, 0.0))
        model._mb_pos = int(state["minibatch.pos"])
        model._mb_perm = (torch.as_tensor(np.asarray(state["minibatch.perm"]), device=dev)
                          if "minibatch.perm" in state else None)
        if model.minibatch_size is not None and model._mb_perm is not None and model._mb_pos > 0:
            b = min(model.minibatch_size, model._X_all.shape[0])
            idx = model._mb_perm[model._mb_pos - b:model._mb_pos]     # the batch the saved model was looking at
            model.X, model.Y = model._X_all[idx].contiguous(), model._Y_all[idx].contiguous()
        model._mb_serial += 1
        if "rng.step" in state:
            model._words()[1] = int(state["rng.step"])
        settings.seed, settings._offset = int(state["rng.seed"]), int(state["rng.offset"])
    return model


def save_checkpoint(model, path, trainer=None):
    """Parameters (+ the optimiser state of ``trainer``: Adam moments and unconstrained variables, step counters) -> .npz."""
    out = state_dict(model)
    if trainer is not None:
        out.update({"trainer." + k: v for k, v in trainer.state_dict().items()})
    np.savez(path, **out)


def load_checkpoint(model, path, trainer=None):
    with np.load(path) as f:
        state = {k: f[k] for k in f.files}
    load_state_dict(model, {k: v for k, v in state.items() if not k.startswith("trainer.")})
    if trainer is not None:
        trainer.load_state_dict({k[len("trainer."):]: v for k, v in state.items() if k.startswith("trainer.")})
    return model
