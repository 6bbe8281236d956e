"""NumPy float64 restatement of the DGPs_with_IWVI importance-weighted ELBO path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned**: the
reference holds no golden vectors and cannot run here; this file follows the
reference line by line in *algorithm*, not in code (it is NumPy with explicit
noise arguments, the reference is a TensorFlow-1 graph with unseeded
``tf.random_normal``).  Citations are relative to ``/root/reference``.

Every stochastic site of the reference takes its standard-normal draw as an
argument here so that the HIP path and the oracle can be fed identical noise:

* ``temp_workaround.py:89``  marginal GP sample      -> ``z`` of shape [S, N, R]
* ``temp_workaround.py:94``  full-cov GP sample      -> ``z`` of shape [S, R, N, 1]
* ``layers.py:86``           latent-variable sample  -> ``z`` of shape q_mu.shape

GPflow-1.x formulas restated (third-party arithmetic on the path, SURVEY.md
section 8 row A-3P): stationary kernels with the squared distance formed as
``|x|^2 + |x2|^2 - 2 x.x2`` on lengthscale-scaled inputs, ``Kuu = K(Z,Z) +
jitter I``, Gaussian variational expectations, whitened ``gauss_kl``, mean
functions acting on the last axis, ``LowerTriangular`` = lower band of q_sqrt.
"""
import numpy as np

DEFAULT_JITTER = 1e-6  # gpflow.settings.numerics.jitter_level default


# --------------------------------------------------------------------------
# kernels (GPflow 1.x gpflow/kernels.py Stationary / RBF / Matern52)
# --------------------------------------------------------------------------
class Stationary:
    """variance * f(scaled distance); lengthscales scalar or [D] (ARD)."""

    def __init__(self, input_dim, variance=1.0, lengthscales=1.0):
        self.input_dim = int(input_dim)
        self.variance = float(variance)
        self.lengthscales = np.broadcast_to(
            np.asarray(lengthscales, dtype=np.float64), (self.input_dim,)).copy()

    def scaled_square_dist(self, X, X2=None):
        X = np.asarray(X, np.float64) / self.lengthscales
        Xs = np.sum(X * X, -1)
        if X2 is None:
            X2, X2s = X, Xs
        else:
            X2 = np.asarray(X2, np.float64) / self.lengthscales
            X2s = np.sum(X2 * X2, -1)
        # batched over leading dims: [..., N, D] x [..., N2, D] -> [..., N, N2]
        return (-2.0 * np.matmul(X, np.swapaxes(X2, -1, -2))
                + Xs[..., :, None] + X2s[..., None, :])

    def Kdiag(self, X):
        return np.full(np.shape(X)[:-1], self.variance, dtype=np.float64)


class RBF(Stationary):
    def K(self, X, X2=None):
        return self.variance * np.exp(-0.5 * self.scaled_square_dist(X, X2))


class Matern52(Stationary):
    def K(self, X, X2=None):
        r = np.sqrt(self.scaled_square_dist(X, X2) + 1e-12)
        s5 = np.sqrt(5.0)
        return self.variance * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * np.exp(-s5 * r)


def Kuu(Z, kern, jitter=DEFAULT_JITTER):
    """gpflow.features.Kuu(InducingPoints): K(Z,Z) + jitter I (temp_workaround.py:39)."""
    return kern.K(Z) + jitter * np.eye(len(Z))


def Kuf(Z, kern, Xnew):
    """gpflow.features.Kuf(InducingPoints): K(Z, Xnew) -> [M, N] (temp_workaround.py:44)."""
    return kern.K(Z, Xnew)


# --------------------------------------------------------------------------
# mean functions, likelihood (GPflow 1.x)
# --------------------------------------------------------------------------
class Zero:
    def __call__(self, X):
        return np.zeros(np.shape(X)[:-1] + (1,))


class Identity:
    def __call__(self, X):
        return np.asarray(X, np.float64)


class Linear:
    def __init__(self, A, b=None):
        self.A = np.asarray(A, np.float64)
        self.b = np.zeros(self.A.shape[1]) if b is None else np.asarray(b, np.float64)

    def __call__(self, X):
        return np.tensordot(np.asarray(X, np.float64), self.A, [[-1], [0]]) + self.b


class Gaussian:
    def __init__(self, variance=1.0):
        self.variance = float(variance)

    def variational_expectations(self, Fmu, Fvar, Y):
        return (-0.5 * np.log(2 * np.pi) - 0.5 * np.log(self.variance)
                - 0.5 * ((Y - Fmu) ** 2 + Fvar) / self.variance)

    def predict_mean_and_var(self, Fmu, Fvar):
        return Fmu, Fvar + self.variance


def tril(q_sqrt):
    """gpflow.transforms.LowerTriangular / tf.matrix_band_part(q_sqrt, -1, 0)."""
    return np.tril(np.asarray(q_sqrt, np.float64))


def solve_lower(L, B):
    from scipy.linalg import solve_triangular
    return solve_triangular(L, B, lower=True)


# --------------------------------------------------------------------------
# temp_workaround.py:12-98
# --------------------------------------------------------------------------
def independent_multisample_sample_conditional(Xnew, Z, kern, f, *, full_cov=False,
                                               full_output_cov=False, q_sqrt=None,
                                               white=False, z=None,
                                               jitter=DEFAULT_JITTER,
                                               intended_full_cov_sample=True):
    """Batched sparse-GP conditional over the leading axis S with optional full
    covariance over the second axis N, plus the reparameterised sample.

    Returns (sample [S,N,R], fmean [S,N,R], fvar [S,N,R] | [S,R,N,N]).
    ``z`` is the injected N(0,1) noise: [S,N,R] (marginal) or [S,R,N,1] (full).

    ``intended_full_cov_sample``: the reference adds ``fmean`` ([S,N,R]) instead
    of ``fmean_SRN1`` at temp_workaround.py:95 (a broadcasting bug, SURVEY.md
    section 3.3); True restates the intended ``fmean_SRN1 + chol(fvar) z``.
    """
    if full_output_cov:
        raise NotImplementedError                                   # :36-37
    Xnew = np.asarray(Xnew, np.float64)
    f = np.asarray(f, np.float64)
    S, N, D = Xnew.shape                                            # :41
    Kmm = Kuu(Z, kern, jitter)                                      # :39
    M = Kmm.shape[0]
    Kmn = Kuf(Z, kern, Xnew.reshape(S * N, D))                      # :44   M x SN
    Knn = kern.K(Xnew) if full_cov else kern.Kdiag(Xnew)            # :45
    R = f.shape[1]                                                  # :47
    Lm = np.linalg.cholesky(Kmm)                                    # :48
    A_M_SN = solve_lower(Lm, Kmn)                                   # :51
    A = A_M_SN.reshape(M, S, N).transpose(1, 0, 2)                  # :52   S x M x N
    if full_cov:
        fvar = Knn - np.matmul(A.transpose(0, 2, 1), A)             # :56   S x N x N
        fvar = np.tile(fvar[:, None, :, :], [1, R, 1, 1])           # :57
    else:
        fvar = Knn - np.sum(A * A, -2)                              # :59   S x N
        fvar = np.tile(fvar[:, None, :], [1, R, 1])                 # :60   S x R x N
    if not white:
        from scipy.linalg import solve_triangular
        A_M_SN = solve_triangular(Lm.T, A_M_SN, lower=False)        # :64
        A = A_M_SN.reshape(M, S, N).transpose(1, 0, 2)              # :65
    fmean = np.matmul(A.transpose(0, 2, 1), f[None])                # :68   S x N x R
    if q_sqrt is not None:
        q_sqrt = np.asarray(q_sqrt, np.float64)
        if q_sqrt.ndim == 2:
            LTA = A[:, None, :, :] * q_sqrt.T[None, :, :, None]     # :73   S x R x M x N
        elif q_sqrt.ndim == 3:
            LTA = np.einsum('rMm,sMn->srmn', tril(q_sqrt), A)       # :78
        else:
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        if full_cov:
            fvar = fvar + np.matmul(LTA.transpose(0, 1, 3, 2), LTA)  # :83
        else:
            fvar = fvar + np.sum(LTA * LTA, 2)                      # :85   S x R x N
    if not full_cov:
        fvar = fvar.transpose(0, 2, 1)                              # :90   S x N x R
        zz = np.zeros_like(fmean) if z is None else np.asarray(z, np.float64)
        sample = fmean + zz * fvar ** 0.5                           # :91
    else:
        fmean_SRN1 = fmean.transpose(0, 2, 1)[:, :, :, None]        # :93
        if z is None:
            # The TF graph only evaluates tf.cholesky(fvar) when the sample is fetched; the
            # IW-ELBO never fetches the final layer's sample (models.py:122-134), and with a
            # single layer fvar is exactly singular (X tiled over K).  z=None == "not fetched".
            sample = fmean.copy()
        else:
            zz = np.asarray(z, np.float64)
            chol = np.linalg.cholesky(fvar)
            if intended_full_cov_sample:
                sample_SRN1 = fmean_SRN1 + np.matmul(chol, zz)
            else:
                sample_SRN1 = fmean + np.matmul(chol, zz)           # :95 as written
            sample = sample_SRN1[:, :, :, 0].transpose(0, 2, 1)     # :96
    return sample, fmean, fvar                                      # :98


# --------------------------------------------------------------------------
# gpflow.conditionals.base_conditional / sample_conditional (2-D inputs)
# call sites: temp_workaround.py:134-138, :157-161
# --------------------------------------------------------------------------
def sample_conditional(Xnew, Z, kern, f, *, full_cov=False, full_output_cov=False,
                       q_sqrt=None, white=False, z=None, jitter=DEFAULT_JITTER):
    """2-D path: Xnew [N,D] -> (sample [N,R], mean [N,R], var [N,R] | [R,N,N])."""
    if full_output_cov:
        raise NotImplementedError
    Xnew = np.asarray(Xnew, np.float64)
    f = np.asarray(f, np.float64)
    N = Xnew.shape[0]
    R = f.shape[1]
    Kmm = Kuu(Z, kern, jitter)
    Kmn = Kuf(Z, kern, Xnew)
    Knn = kern.K(Xnew) if full_cov else kern.Kdiag(Xnew)
    Lm = np.linalg.cholesky(Kmm)
    A = solve_lower(Lm, Kmn)                                        # M x N
    if full_cov:
        fvar = np.tile((Knn - A.T @ A)[None], [R, 1, 1])            # R x N x N
    else:
        fvar = np.tile((Knn - np.sum(A * A, 0))[None], [R, 1])      # R x N
    if not white:
        from scipy.linalg import solve_triangular
        A = solve_triangular(Lm.T, A, lower=False)
    fmean = A.T @ f                                                 # N x R
    if q_sqrt is not None:
        q_sqrt = np.asarray(q_sqrt, np.float64)
        if q_sqrt.ndim == 2:
            LTA = A[None] * q_sqrt.T[:, :, None]                    # R x M x N
        elif q_sqrt.ndim == 3:
            LTA = np.matmul(tril(q_sqrt).transpose(0, 2, 1), A[None])  # R x M x N
        else:
            raise ValueError("Bad dimension for q_sqrt: %s" % str(q_sqrt.ndim))
        if full_cov:
            fvar = fvar + np.matmul(LTA.transpose(0, 2, 1), LTA)
        else:
            fvar = fvar + np.sum(LTA * LTA, 1)
    if not full_cov:
        fvar = fvar.T                                               # N x R
        zz = np.zeros_like(fmean) if z is None else np.asarray(z, np.float64)
        sample = fmean + zz * fvar ** 0.5
    else:
        zz = np.zeros((R, N, 1)) if z is None else np.asarray(z, np.float64)
        chol = np.linalg.cholesky(fvar + jitter * np.eye(N))        # gpflow _sample_mvn jitter
        sample = (fmean.T[:, :, None] + np.matmul(chol, zz))[:, :, 0].T
    return sample, fmean, fvar


# --------------------------------------------------------------------------
# temp_workaround.py:107-161
# --------------------------------------------------------------------------
class SharedMixedMok:
    """R latent GPs sharing one kernel, mixed by W [P x R] (temp_workaround.py:107-115)."""

    def __init__(self, kernel, W):
        self.kernel = kernel
        self.W = np.asarray(W, np.float64)


def multisample_sample_conditional(Xnew, Z, kern, f, *, full_cov=False,
                                   full_output_cov=False, q_sqrt=None, white=False,
                                   z=None, jitter=DEFAULT_JITTER):
    Xnew = np.asarray(Xnew, np.float64)
    if isinstance(kern, SharedMixedMok):                            # :123
        if Xnew.ndim == 3:
            sample, gmean, gvar = independent_multisample_sample_conditional(
                Xnew, Z, kern.kernel, f, white=white, q_sqrt=q_sqrt,
                full_output_cov=False, full_cov=False, z=z, jitter=jitter)  # :125-129
        else:
            sample, gmean, gvar = sample_conditional(
                Xnew, Z, kern.kernel, f, white=white, q_sqrt=q_sqrt,
                full_output_cov=False, full_cov=False, z=z, jitter=jitter)  # :134-138
        W = kern.W
        f_sample = np.matmul(sample, W.T)                           # :143
        f_mu = np.matmul(gmean, W.T)                                # :144
        f_var = np.matmul(gvar, (W ** 2).T)                         # :145
        return f_sample, f_mu, f_var
    if Xnew.ndim == 3:                                              # :150
        return independent_multisample_sample_conditional(
            Xnew, Z, kern, f, full_cov=full_cov, full_output_cov=full_output_cov,
            q_sqrt=q_sqrt, white=white, z=z, jitter=jitter)
    return sample_conditional(Xnew, Z, kern, f, full_cov=full_cov,
                              full_output_cov=full_output_cov, q_sqrt=q_sqrt,
                              white=white, z=z, jitter=jitter)      # :157-161


def gauss_kl(q_mu, q_sqrt, K=None):
    """temp_workaround.py:167-188, KL branch (:186-188) -> gpflow gauss_kl, white.

    KL[N(q_mu, L L^T) || N(0, I)] summed over the R columns of q_mu.
    """
    if q_sqrt is None or K is not None:
        raise NotImplementedError("SGHMC / unwhitened branches are out of scope")
    q_mu = np.asarray(q_mu, np.float64)
    L = tril(q_sqrt)
    M, R = q_mu.shape
    mahalanobis = np.sum(q_mu ** 2)
    constant = -float(M * R)
    logdet_qcov = np.sum(np.log(np.square(np.diagonal(L, axis1=-2, axis2=-1))))
    trace = np.sum(L ** 2)
    return 0.5 * (mahalanobis + constant - logdet_qcov + trace)


# --------------------------------------------------------------------------
# layers.py
# --------------------------------------------------------------------------
LOCAL, GLOBAL = 0, 1                                                # layers.py:9-11


class GPLayer:
    regularizer_type = GLOBAL                                       # layers.py:15

    def __init__(self, kern, Z, num_outputs, mean_function=None, jitter=DEFAULT_JITTER):
        self.Z = np.asarray(Z, np.float64)
        self.num_inducing = len(self.Z)
        self.q_mu = np.zeros((self.num_inducing, num_outputs))       # layers.py:21
        self.q_sqrt = np.tile(np.eye(self.num_inducing)[None], [num_outputs, 1, 1])  # :24
        self.kern = kern
        self.mean_function = mean_function or Zero()                 # :30
        self.num_outputs = num_outputs
        self.jitter = jitter

    def propagate(self, F, full_cov=False, z=None, **kwargs):
        samples, mean, cov = multisample_sample_conditional(
            F, self.Z, self.kern, self.q_mu, full_cov=full_cov,
            q_sqrt=self.q_sqrt, white=True, z=z, jitter=self.jitter)  # layers.py:36-42
        kl = gauss_kl(self.q_mu, self.q_sqrt)                        # :44
        mf = self.mean_function(F)                                   # :46
        return samples + mf, mean + mf, cov, kl                      # :47-50


class Encoder:
    """layers.py:108-152 -- tanh MLP with skip connections on equal dims."""

    def __init__(self, latent_dim, input_dim, network_dims, rng=None, activation_func=None):
        self.activation_func = np.tanh if activation_func is None else activation_func   # :119 (default tf.nn.tanh)
        self.latent_dim = latent_dim
        self.layer_dims = [input_dim, *network_dims, latent_dim * 2]
        rng = np.random.default_rng(0) if rng is None else rng
        self.Ws, self.bs = [], []
        for din, dout in zip(self.layer_dims[:-1], self.layer_dims[1:]):
            std = (2.0 / (din + dout)) ** 0.5                        # :125 xavier
            self.Ws.append(rng.standard_normal((din, dout)) * std)
            self.bs.append(np.zeros(dout))

    def __call__(self, Z):
        Z = np.asarray(Z, np.float64)
        n = len(self.bs)
        for i, (W, b, din, dout) in enumerate(zip(self.Ws, self.bs,
                                                  self.layer_dims[:-1], self.layer_dims[1:])):
            Z0 = Z
            Z = np.matmul(Z, W) + b                                  # :141
            if i < n - 1:
                Z = self.activation_func(Z)                          # :143-144
            if dout == din:
                Z = Z + Z0                                           # :146-147
        means, raw = np.split(Z, 2, axis=-1)                         # :149
        q_sqrt = np.logaddexp(0.0, raw - 3.0)                        # :150 softplus
        return means, q_sqrt


class LatentVariableLayer:
    regularizer_type = LOCAL                                         # layers.py:54

    def __init__(self, latent_dim, XY_dim=None, encoder=None):
        self.latent_dim = latent_dim
        if encoder is None:
            assert XY_dim, 'must pass XY_dim or else an encoder'     # :67
            encoder = Encoder(latent_dim, XY_dim, [20, 20])
        self.encoder = encoder

    def propagate(self, F, inference_amorization_inputs=None,
                  is_sampled_local_regularizer=False, z=None, **kwargs):
        F = np.asarray(F, np.float64)
        if inference_amorization_inputs is None:                     # :73-81 prior; the placeholders (:60-64) may be fed
            shape = F.shape[:-1] + (self.latent_dim,)
            q_mu = np.broadcast_to(np.asarray(getattr(self, "q_mu_placeholder", 0.0), np.float64), shape) * np.ones(shape)
            q_sqrt = np.broadcast_to(np.asarray(getattr(self, "q_sqrt_placeholder", 1.0), np.float64), shape) * np.ones(shape)
        else:
            q_mu, q_sqrt = self.encoder(inference_amorization_inputs)  # :83
        zz = np.zeros_like(q_mu) if z is None else np.asarray(z, np.float64)
        W = q_mu + zz * q_sqrt                                       # :86-87
        samples = np.concatenate([F, W], -1)                         # :89
        mean = np.concatenate([F, q_mu], -1)                         # :90
        cov = np.concatenate([np.zeros_like(F), q_sqrt ** 2], -1)    # :91
        if is_sampled_local_regularizer:                             # :98-100
            log_q = -0.5 * ((W - q_mu) / q_sqrt) ** 2 - np.log(q_sqrt) - 0.5 * np.log(2 * np.pi)
            log_p = -0.5 * W ** 2 - 0.5 * np.log(2 * np.pi)
            kl = log_q - log_p
        else:                                                        # :101-103
            kl = 0.5 * (q_sqrt ** 2 + q_mu ** 2 - 1.0) - np.log(q_sqrt)
        return samples, mean, cov, kl


# --------------------------------------------------------------------------
# models.py
# --------------------------------------------------------------------------
class DGP_VI:
    def __init__(self, X, Y, layers, likelihood, num_samples=1, num_data=None):
        self.X = np.asarray(X, np.float64)
        self.Y = np.asarray(Y, np.float64)
        self.layers = list(layers)
        self.likelihood = likelihood
        self.num_samples = num_samples
        # models.py:18 num_data = X.shape[0] of the *full* data set; X here is the minibatch
        self.num_data = self.X.shape[0] if num_data is None else num_data

    def propagate(self, X, full_cov=False, inference_amorization_inputs=None,
                  is_sampled_local_regularizer=False, zs=None):
        """models.py:31-46.  ``zs`` = one noise array (or None) per layer."""
        samples, means, covs, kls, kl_types = [X], [], [], [], []
        zs = [None] * len(self.layers) if zs is None else zs
        for layer, z in zip(self.layers, zs):
            sample, mean, cov, kl = layer.propagate(
                samples[-1], full_cov=full_cov,
                inference_amorization_inputs=inference_amorization_inputs,
                is_sampled_local_regularizer=is_sampled_local_regularizer, z=z)
            samples.append(sample)
            means.append(mean)
            covs.append(cov)
            kls.append(kl)
            kl_types.append(layer.regularizer_type)
        return samples[1:], means, covs, kls, kl_types

    def build_likelihood(self, zs=None):
        """models.py:49-86 (VI bound; 2-D [S*N, D] tiling, mean over S)."""
        S = self.num_samples
        X_tiled = np.tile(self.X, [S, 1])                            # :50
        Y_tiled = np.tile(self.Y, [S, 1])                            # :51
        XY = np.concatenate([X_tiled, Y_tiled], -1)                  # :53
        _, means, covs, kls, kl_types = self.propagate(
            X_tiled, full_cov=False, inference_amorization_inputs=XY,
            is_sampled_local_regularizer=False, zs=zs)               # :58-61
        local_kls = [kl for kl, t in zip(kls, kl_types) if t == LOCAL]
        global_kls = [kl for kl, t in zip(kls, kl_types) if t == GLOBAL]
        var_exp = self.likelihood.variational_expectations(means[-1], covs[-1], Y_tiled)
        L_SN = np.sum(var_exp, -1)                                   # :69
        N = self.X.shape[0]
        L_S_N = L_SN.reshape(S, N)                                   # :72
        if local_kls:
            local = np.sum(np.concatenate(local_kls, -1), -1).reshape(S, N)
            L_S_N = L_S_N - local                                    # :74-78
        scale = self.num_data / N                                    # :80-81
        logp = np.mean(L_S_N, 0)                                     # :84
        return np.sum(logp) * scale - np.sum(global_kls)             # :86

    def build_predict(self, X, full_cov=False, zs=None):
        _, means, covs, _, _ = self.propagate(X, full_cov=full_cov, zs=zs)  # :89-91
        return means[-1], covs[-1]


class DGP_IWVI(DGP_VI):
    def log_weights(self, zs=None):
        """models.py:113-142 -> L_NK [N, K] plus the global KLs and final mean/var."""
        K = self.num_samples
        X_tiled = np.tile(self.X[:, None, :], [1, K, 1])             # :113
        Y_tiled = np.tile(self.Y[:, None, :], [1, K, 1])             # :114
        XY = np.concatenate([X_tiled, Y_tiled], -1)                  # :116
        samples, means, covs, kls, kl_types = self.propagate(
            X_tiled, full_cov=True, inference_amorization_inputs=XY,
            is_sampled_local_regularizer=True, zs=zs)                # :122-125
        local_kls = [kl for kl, t in zip(kls, kl_types) if t == LOCAL]
        global_kls = [kl for kl, t in zip(kls, kl_types) if t == GLOBAL]
        cov = covs[-1]
        if cov.ndim == 4:                                            # [N, Dy, K, K]
            cov_diag = np.diagonal(cov, axis1=-2, axis2=-1).transpose(0, 2, 1)  # :133
        else:                                                        # SharedMixedMok last layer
            cov_diag = cov
        var_exp = self.likelihood.variational_expectations(means[-1], cov_diag, Y_tiled)  # :134
        L_NK = np.sum(var_exp, 2)                                    # :138
        if local_kls:
            L_NK = L_NK - np.sum(np.concatenate(local_kls, -1), 2)   # :140-142
        return L_NK, global_kls, means, covs, samples

    def build_likelihood(self, zs=None):
        """models.py:112-150 (the IW-ELBO)."""
        L_NK, global_kls, _, _, _ = self.log_weights(zs)
        K = self.num_samples
        scale = self.num_data / self.X.shape[0]                      # :144-145
        m = np.max(L_NK, 1, keepdims=True)
        lse = m[:, 0] + np.log(np.sum(np.exp(L_NK - m), 1))          # tf.reduce_logsumexp
        logp = lse - np.log(K)                                       # :148
        return np.sum(logp) * scale - np.sum(global_kls)             # :150
