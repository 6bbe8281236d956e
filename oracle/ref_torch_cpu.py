"""CPU baseline: the reference's op sequence restated with torch-CPU (MKL), float64 like the
reference's default dtype (or float32).  TEST / BENCH INFRASTRUCTURE ONLY: used by bench.py's
``cpu_baseline`` leg ("kind": "port") and by tests as a second opinion on the NumPy oracle.

It materialises the same intermediates the TF graph does (temp_workaround.py:39-91: Gram ->
Cholesky -> triangular solve -> einsum('rMm,sMn->srmn') -> square-reduce -> sample), including the
full [B, Dy, K, K] covariance of the final layer (models.py:122-133), so its cost is the
reference's cost on the same host cores, not an optimised re-derivation.
"""
import math

import numpy as np
import torch


def _rbf(X, X2, ls, var):
    X = X / ls
    Xs = (X * X).sum(-1)
    if X2 is None:
        X2, X2s = X, Xs
    else:
        X2 = X2 / ls
        X2s = (X2 * X2).sum(-1)
    d = -2 * X @ X2.transpose(-1, -2) + Xs[..., :, None] + X2s[..., None, :]
    return var * torch.exp(-0.5 * d)


def _matern52(X, X2, ls, var):
    """GPflow 1.x Matern52 (r = sqrt(scaled square distance + 1e-12)), as oracle/iwvi_oracle.py."""
    X = X / ls
    Xs = (X * X).sum(-1)
    if X2 is None:
        X2, X2s = X, Xs
    else:
        X2 = X2 / ls
        X2s = (X2 * X2).sum(-1)
    d = -2 * X @ X2.transpose(-1, -2) + Xs[..., :, None] + X2s[..., None, :]
    r = torch.sqrt(d + 1e-12)
    s5 = math.sqrt(5.0)
    return var * (1.0 + s5 * r + 5.0 / 3.0 * r * r) * torch.exp(-s5 * r)


class CpuDGP:
    def __init__(self, spec, dtype=torch.float64):
        self.dtype = dtype
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        self.layers = []
        for l in spec["layers"]:
            if l["type"] == "lv":
                self.layers.append(dict(type="lv", Lw=l["latent_dim"], act=l.get("act", torch.tanh), W=[t(w) for w in l["enc_W"]],
                                        b=[t(b) for b in l["enc_b"]], dims=l["dims"]))
            else:
                self.layers.append(dict(type="gp", Z=t(l["Z"]), ls=t(l["ls"]), var=l["var"], q_mu=t(l["q_mu"]),
                                        q_sqrt=torch.tril(t(l["q_sqrt"])), W=None if l["W"] is None else t(l["W"]),
                                        A=t(l["mf"][1]) if l["mf"][0] == "linear" else None))
        B = spec["B"]
        self.X, self.Y = t(spec["X"][:B]), t(spec["Y"][:B])
        self.K, self.lik_var, self.n_data = spec["K"], spec["lik_var"], spec["n_data"]

    def _conditional(self, L, F, full_cov, z):
        S, N, D = F.shape
        M = L["Z"].shape[0]
        R = L["q_mu"].shape[1]
        _k = _matern52 if L.get("kern") == "matern52" else _rbf
        Kmm = _k(L["Z"], None, L["ls"], L["var"]) + 1e-6 * torch.eye(M, dtype=self.dtype)        # :39
        Kmn = _k(L["Z"], F.reshape(S * N, D), L["ls"], L["var"])                                  # :44
        Lm = torch.linalg.cholesky(Kmm)                                                           # :48
        A = torch.linalg.solve_triangular(Lm, Kmn, upper=False)                                   # :51
        A = A.reshape(M, S, N).permute(1, 0, 2)                                                   # :52
        if full_cov:
            Knn = _k(F, None, L["ls"], L["var"])                                                  # :45
            fvar = (Knn - A.transpose(1, 2) @ A)[:, None].repeat(1, R, 1, 1)                       # :56-57
        else:
            fvar = (L["var"] - (A * A).sum(-2))[:, None].repeat(1, R, 1)                           # :59-60
        fmean = A.transpose(1, 2) @ L["q_mu"][None].repeat(S, 1, 1)                                # :68
        # :78 einsum('rMm,sMn->srmn') evaluated the way TF evaluates it: one GEMM [R*m, M] x [M, S*N] on the
        # un-transposed solve result, then the reshape/transposes -- the [S,R,M,N] intermediate is materialised
        LTA = (L["q_sqrt"].transpose(1, 2).reshape(R * M, M) @ A.permute(1, 0, 2).reshape(M, S * N))
        LTA = LTA.reshape(R, M, S, N).permute(2, 0, 1, 3)
        if full_cov:
            fvar = fvar + LTA.transpose(2, 3) @ LTA                                                # :83
            return fmean, fmean, fvar
        fvar = (fvar + (LTA * LTA).sum(2)).transpose(1, 2)                                         # :85,:90
        return fmean + z * fvar ** 0.5, fmean, fvar                                                # :91

    def elbo(self, zs):
        return float(self.elbo_tensor(zs))

    def elbo_tensor(self, zs, mode_vi=False):
        L_NK, glob = self.log_weights_tensor(zs, mode_vi)
        B, K = L_NK.shape
        logp = L_NK.mean(1) if mode_vi else torch.logsumexp(L_NK, 1) - math.log(K)                 # :84 / :148
        return logp.sum() * (self.n_data / B) - glob                                               # :150

    def log_weights_tensor(self, zs, mode_vi=False):
        """Per-sample log-weights L_NK [B, K] (models.py:134-142) and the summed global KL, as tensors.  ``elbo_tensor``
        (the IW-ELBO, differentiable w.r.t. whichever parameter tensors require grad: the gradient oracle of
        oracle/grad_oracle.py) is built on it.  ``mode_vi``: the bound of DGP_VI instead (models.py:49-86: analytic
        local KL, mean over the samples), same [B, K, .] noise layout."""
        B, K = self.X.shape[0], self.K
        F = self.X[:, None, :].repeat(1, K, 1)                                                     # models.py:113
        Yt = self.Y[:, None, :].repeat(1, K, 1)
        XY = torch.cat([F, Yt], -1)
        local, glob = [], []
        mean = cov = None
        for L, z in zip(self.layers, zs):
            z = None if z is None else torch.as_tensor(z, dtype=self.dtype)
            if L["type"] == "lv":
                H = XY
                n = len(L["W"])
                for i, (W, b) in enumerate(zip(L["W"], L["b"])):                                   # layers.py:137-147
                    H0 = H
                    H = H @ W + b
                    if i < n - 1:
                        H = L.get("act", torch.tanh)(H)                                            # layers.py:119 (default tf.nn.tanh); spec["layers"][i]["act"]
                    if W.shape[0] == W.shape[1]:
                        H = H + H0
                mu, raw = H.split(L["Lw"], -1)
                sg = torch.nn.functional.softplus(raw - 3.0)
                Wl = mu + z * sg
                F = torch.cat([F, Wl], -1)
                if mode_vi:
                    local.append(0.5 * (sg ** 2 + mu ** 2 - 1.0) - torch.log(sg) + 0.0 * Wl)       # layers.py:101-103
                else:
                    local.append((-0.5 * ((Wl - mu) / sg) ** 2 - torch.log(sg)) - (-0.5 * Wl ** 2))    # :98-100
                continue
            if L["W"] is not None:                                                                 # SharedMixedMok branch
                s, m, v = self._conditional(L, F, False, z)
                s, m, v = s @ L["W"].T, m @ L["W"].T, v @ (L["W"] ** 2).T                          # :142-145
                mf = F @ L["A"] if L["A"] is not None else 0.0
                F, mean, cov = s + mf, m + mf, v
            else:
                _, mean, cov = self._conditional(L, F, True, None)                                  # full_cov over K
                F = mean
            M, R = L["q_mu"].shape
            Lq = L["q_sqrt"]
            glob.append(0.5 * ((L["q_mu"] ** 2).sum() - M * R
                               - torch.log(torch.diagonal(Lq, dim1=-2, dim2=-1) ** 2).sum() + (Lq ** 2).sum()))
        if cov.dim() == 4:
            cov = torch.diagonal(cov, dim1=-2, dim2=-1).transpose(1, 2)                            # models.py:133
        lik_var = torch.as_tensor(self.lik_var, dtype=self.dtype)
        ve = -0.5 * math.log(2 * math.pi) - 0.5 * torch.log(lik_var) \
            - 0.5 * ((Yt - mean) ** 2 + cov) / lik_var                                             # :134
        L_NK = ve.sum(2)
        for kl in local:
            L_NK = L_NK - kl.sum(2)
        return L_NK, sum(glob)
