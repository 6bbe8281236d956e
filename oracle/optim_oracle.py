"""Optimiser oracle for SURVEY.md section 8 row F1 (TEST INFRASTRUCTURE ONLY): the two update rules the reference
applies per training step (experiments/build_models.py:284-304), restated in NumPy float64.

  * ``natgrad_step``: GPflow 1.x ``NatGradOptimizer(gamma)`` with the default natural parameterisation on a whitened
    (q_mu [M, R], q_sqrt [R, M, M]).  GPflow is not vendored in /root/reference (unpinned,
    1.3 <= v < 2.0); the algorithm restated here is the published one (Salimbeni, Eleftheriadis, Hensman 2018,
    "Natural gradients in practice", eq. 3-5 / GPflow ``training/natgrad_optimizer.py``):
        eta = (m, S + m m^T),  theta = (S^-1 m, -1/2 S^-1),  theta <- theta - gamma * dLoss/d eta,
    then ``natural_to_meanvarsqrt`` (cholesky(-2 theta_2), inverse, S = X^T X, mu = S theta_1, cholesky(S)).
    Pinned by a known answer (tests/test_optim_oracle.py): for a conjugate model one step with gamma = 1 lands on
    the optimal q(u) from any start.
  * ``adam_step``: ``tf.train.AdamOptimizer`` (lr_t = lr sqrt(1-b2^t)/(1-b1^t); x -= lr_t m / (sqrt(v) + eps)) on
    GPflow's unconstrained variables; positive parameters use ``transforms.Log1pe`` (p = softplus(x) + 1e-6).
"""
import numpy as np


def _sym_grad_S_from_grad_L(L, Lbar):
    """d/dS (symmetric) of a function given its gradient w.r.t. the lower factor L of S = L L^T."""
    n = L.shape[0]
    P = np.tril(L.T @ np.tril(Lbar))
    P[np.arange(n), np.arange(n)] *= 0.5
    Li = np.linalg.inv(L)
    Q = Li.T @ P @ Li
    return 0.5 * (Q + Q.T)


def natgrad_step(q_mu, q_sqrt, g_mu, g_sqrt, gamma):
    """One step on the LOSS gradients (g = d loss / d q_mu, d loss / d tril(q_sqrt)); returns new (q_mu, q_sqrt)."""
    q_mu, q_sqrt = np.array(q_mu, dtype=np.float64), np.array(q_sqrt, dtype=np.float64)
    out_mu, out_sqrt = np.empty_like(q_mu), np.empty_like(q_sqrt)
    for r in range(q_mu.shape[1]):
        L, m = np.tril(q_sqrt[r]), q_mu[:, r]
        Sbar = _sym_grad_S_from_grad_L(L, np.asarray(g_sqrt[r], dtype=np.float64))
        g1 = np.asarray(g_mu, dtype=np.float64)[:, r] - 2.0 * Sbar @ m          # d loss / d eta_1
        g2 = Sbar                                                               # d loss / d eta_2
        Sinv = np.linalg.inv(L @ L.T)
        th1, th2 = Sinv @ m - gamma * g1, -0.5 * Sinv - gamma * g2
        X = np.linalg.inv(np.linalg.cholesky(-2.0 * th2))
        S = X.T @ X
        out_mu[:, r] = S @ th1
        out_sqrt[r] = np.linalg.cholesky(S)
    return out_mu, out_sqrt


def softplus(x):
    return np.logaddexp(0.0, x)


def to_unconstrained(p, positive):
    p = np.asarray(p, dtype=np.float64)
    if not positive:
        return p.copy()
    y = p - 1e-6
    return np.where(y > 30.0, y, np.log(np.expm1(np.minimum(y, 30.0))))


def to_constrained(x, positive):
    return softplus(x) + 1e-6 if positive else x.copy()


class Adam:
    """State for a list of parameters; ``step(grads_of_loss)`` returns the new constrained values."""

    def __init__(self, params, positive, lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.pos = list(positive)
        self.x = [to_unconstrained(p, s) for p, s in zip(params, self.pos)]
        self.m = [np.zeros_like(x) for x in self.x]
        self.v = [np.zeros_like(x) for x in self.x]
        self.lr, self.b1, self.b2, self.eps, self.t = lr, beta1, beta2, eps, 0

    def step(self, grads, lr=None):
        self.t += 1
        lr = self.lr if lr is None else lr
        lr_t = lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        out = []
        for i, g in enumerate(grads):
            g = np.asarray(g, dtype=np.float64)
            if self.pos[i]:
                g = g * (1.0 - np.exp(-softplus(self.x[i])))        # chain rule through softplus
            self.m[i] = self.b1 * self.m[i] + (1 - self.b1) * g
            self.v[i] = self.b2 * self.v[i] + (1 - self.b2) * g * g
            self.x[i] = self.x[i] - lr_t * self.m[i] / (np.sqrt(self.v[i]) + self.eps)
            out.append(to_constrained(self.x[i], self.pos[i]))
        return out


def staircase_decay(base, step, decay_rate, decay_steps=1000):
    """tf.train.exponential_decay(base, step, decay_steps, decay_rate, staircase=True) (build_models.py:289-291)."""
    return base * decay_rate ** (step // decay_steps)
