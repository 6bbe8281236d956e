"""Known-answer pins of the optimiser oracle (oracle/optim_oracle.py)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import optim_oracle as oo   # noqa: E402


def test_natgrad_with_unit_step_solves_a_conjugate_problem_in_one_step():
    """loss(m, S) = KL[N(m, S) || N(m*, S*)]: the natural gradient with gamma = 1 lands on (m*, S*) from any start
    (the defining property of a natural-gradient step in an exponential family)."""
    rng = np.random.default_rng(0)
    n = 6
    A = rng.standard_normal((n, n)); S_star = A @ A.T + n * np.eye(n); m_star = rng.standard_normal(n)
    L0 = np.tril(rng.standard_normal((n, n))) * 0.3 + np.eye(n); m0 = rng.standard_normal(n)
    P = np.linalg.inv(S_star)
    # gradients of the KL w.r.t. m and L (S = L L^T): dKL/dm = P (m - m*), dKL/dS = 1/2 (P - S^-1) -> dKL/dL = 2 dKL/dS L
    S0 = L0 @ L0.T
    g_m = P @ (m0 - m_star)
    g_L = np.tril(2.0 * (0.5 * (P - np.linalg.inv(S0))) @ L0)
    mu, Ls = oo.natgrad_step(m0[:, None], L0[None], g_m[:, None], g_L[None], 1.0)
    np.testing.assert_allclose(mu[:, 0], m_star, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(Ls[0] @ Ls[0].T, S_star, rtol=1e-9, atol=1e-10)


def test_natgrad_small_step_is_a_descent_direction_and_keeps_S_positive_definite():
    rng = np.random.default_rng(1)
    n = 5
    A = rng.standard_normal((n, n)); S_star = A @ A.T + n * np.eye(n); m_star = rng.standard_normal(n)
    P = np.linalg.inv(S_star)
    kl = lambda m, L: 0.5 * (np.trace(P @ L @ L.T) + (m - m_star) @ P @ (m - m_star) - n
                             + np.log(np.linalg.det(S_star)) - 2 * np.log(np.diag(L)).sum())
    L0 = np.eye(n); m0 = np.zeros(n)
    g_m = P @ (m0 - m_star)
    g_L = np.tril((P - np.linalg.inv(L0 @ L0.T)) @ L0)
    mu, Ls = oo.natgrad_step(m0[:, None], L0[None], g_m[:, None], g_L[None], 0.1)
    assert kl(mu[:, 0], Ls[0]) < kl(m0, L0)
    assert np.all(np.diag(Ls[0]) > 0)


def test_adam_first_step_moves_by_lr_against_the_gradient_sign_and_respects_positivity():
    p = [np.array([1.0, -2.0, 3.0]), np.array([0.5, 2.0])]
    opt = oo.Adam(p, [False, True], lr=0.01)
    new = opt.step([np.array([0.3, -0.2, 10.0]), np.array([5.0, -5.0])])
    np.testing.assert_allclose(new[0], p[0] - 0.01 * np.sign([0.3, -0.2, 10.0]), rtol=0, atol=1e-7)   # |step 1| = lr
    x0 = oo.to_unconstrained(p[1], True)
    np.testing.assert_allclose(oo.to_constrained(x0, True), p[1], rtol=1e-12)
    np.testing.assert_allclose(new[1], oo.to_constrained(x0 - 0.01 * np.sign([5.0, -5.0]), True), rtol=1e-6)
    assert np.all(new[1] > 0)


def test_staircase_decay():
    assert oo.staircase_decay(0.1, 999, 0.5) == 0.1 and oo.staircase_decay(0.1, 1000, 0.5) == 0.05
    assert oo.staircase_decay(0.1, 2500, 0.5) == 0.025
