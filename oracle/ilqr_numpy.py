"""SECOND, independently written restatement of the reference's iLQR -- CPU ORACLE (test infrastructure).

Purpose (parity item 1d): `oracle/mas_oracle.c` is pinned by the reference's own fixtures only for
rollout / objective / finite differences; `iLQR::solve` is pinned by nothing the reference holds.  This module
restates `solve` a second time, from the reference text alone and in a different style -- dense numpy matrices,
`numpy.linalg.cholesky` / `numpy.linalg.solve` instead of hand-written loops, Python callbacks like the
reference's std::function hooks -- so that a transcription error in one restatement shows up as a disagreement
between the two.  It is NOT bit-exact with the C oracle and is not meant to be: BLAS/LAPACK order their sums
differently, and the algorithm amplifies last-bit differences (tests/test_oracle_sensitivity.py).  The tests
compare the two at short horizons, where the problem is benign, to finite-difference-noise tolerance.

Only tests/ may import this module.  Every function cites the reference lines it follows
(/root/reference/include/multi_agent_solver/...).
"""
from __future__ import annotations

import math

import numpy as np


# ------------------------------------------------------------------ integrator.hpp:19-48 ------
def integrate_rk4(x, u, dt, f):
    k1 = f(x, u)
    k2 = f(x + 0.5 * dt * k1, u)
    k3 = f(x + 0.5 * dt * k2, u)
    k4 = f(x + dt * k3, u)
    return x + (dt / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)


def integrate_horizon(x0, U, dt, f):
    """U is nu x T (one column per step, types.hpp:14-19); returns nx x (T+1)."""
    T = U.shape[1]
    X = np.zeros((x0.size, T + 1))
    X[:, 0] = x0
    x = x0.copy()
    for t in range(T):
        x = integrate_rk4(x, U[:, t], dt, f)
        X[:, t + 1] = x
    return X


# ------------------------------------------------------------------ ocp.hpp:14-28 ------------
def trajectory_cost(X, U, stage, terminal):
    c = 0.0
    T = U.shape[1]
    for t in range(T):
        c += stage(X[:, t], U[:, t], t)
    return c + terminal(X[:, T])


# ------------------------------------------------------------------ finite_differences.hpp ----
def _unit(n, i, eps):
    d = np.zeros(n)
    d[i] = eps
    return d


def _safe(v):                                             # :95-107
    return v if math.isfinite(v) else 0.0


def fd_dynamics_state_jacobian(f, x, u):                  # :53-71
    n, eps = x.size, 1e-6
    A = np.zeros((n, n))
    for i in range(n):
        dx = _unit(n, i, eps)
        A[:, i] = (f(x + dx, u) - f(x - dx, u)) / (2 * eps)
    return A


def fd_dynamics_control_jacobian(f, x, u):                # :73-92
    n, m, eps = x.size, u.size, 1e-6
    B = np.zeros((n, m))
    for i in range(m):
        du = _unit(m, i, eps)
        B[:, i] = (f(x, u + du) - f(x, u - du)) / (2 * eps)
    return B


def fd_cost_state_gradient(c, x, u, t):                   # :110-122
    eps = 1e-6
    return np.array([(c(x + _unit(x.size, i, eps), u, t) - c(x - _unit(x.size, i, eps), u, t)) / (2 * eps) for i in range(x.size)])


def fd_cost_control_gradient(c, x, u, t):                 # :124-136
    eps = 1e-6
    return np.array([(c(x, u + _unit(u.size, i, eps), t) - c(x, u - _unit(u.size, i, eps), t)) / (2 * eps) for i in range(u.size)])


def _fd_hessian(g, z):
    """:138-210,226-261: 3-point diagonal, 4-point off-diagonal with (i,j) and (j,i) evaluated separately."""
    n, eps = z.size, 1e-5
    H = np.zeros((n, n))
    for i in range(n):
        d = _unit(n, i, eps)
        H[i, i] = (_safe(g(z + d)) - 2 * _safe(g(z)) + _safe(g(z - d))) / (eps * eps)
    for i in range(n):
        for j in range(n):
            if i != j:
                di, dj = _unit(n, i, eps), _unit(n, j, eps)
                H[i, j] = (_safe(g(z + di + dj)) - _safe(g(z + di - dj)) - _safe(g(z - di + dj)) + _safe(g(z - di - dj))) / (4 * eps * eps)
    return H


def fd_cost_state_hessian(c, x, u, t):                    # :138-173
    return _fd_hessian(lambda z: c(z, u, t), x)


def fd_cost_control_hessian(c, x, u, t):                  # :175-210
    return _fd_hessian(lambda z: c(x, z, t), u)


def fd_terminal_gradient(tc, x):                          # :212-224
    eps = 1e-6
    return np.array([(tc(x + _unit(x.size, i, eps)) - tc(x - _unit(x.size, i, eps))) / (2 * eps) for i in range(x.size)])


def fd_terminal_hessian(tc, x):                           # :226-261
    return _fd_hessian(tc, x)


def fd_cost_cross_term(c, x, u, t):                       # :263-287, nu x nx
    m, n, eps = u.size, x.size, 1e-6
    H = np.zeros((m, n))
    for i in range(m):
        for j in range(n):
            du, dx = _unit(m, i, eps), _unit(n, j, eps)
            H[i, j] = (_safe(c(x + dx, u + du, t)) - _safe(c(x - dx, u + du, t)) - _safe(c(x + dx, u - du, t)) + _safe(c(x - dx, u - du, t))) / (4 * eps * eps)
    return H


# ------------------------------------------------------------------ workloads ----------------
class Problem:
    """What an OCP carries (ocp.hpp:30-100) after initialize_problem installed FD hooks for every derivative the
    example left empty (ocp.hpp:117-135).  `analytic` maps hook names to callables for the examples that set them."""

    def __init__(self, f, stage, terminal, x0, T, dt, nu, bounds=None, analytic=None):
        self.f, self.stage, self.terminal = f, stage, terminal
        self.x0, self.T, self.dt, self.nu, self.bounds = np.asarray(x0, dtype=float), T, dt, nu, bounds
        self.analytic = analytic or {}

    def hook(self, name, fd):
        return self.analytic.get(name, fd)


def single_track(x0, T=100, dt=0.5):
    """examples/models/single_track_model.hpp:23-44; examples/multi_agent_single_track.cpp:46-67,103-110"""
    L, R, vt, w_track, w_speed, w_delta, w_acc = 2.5, 20.0, 5.0, 1.0, 1.0, 0.001, 0.001

    def f(x, u):
        psi, v = x[2], x[3]
        return np.array([v * math.cos(psi), v * math.sin(psi), v * math.tan(u[0]) / L, u[1]])

    def stage(x, u, t):
        dist = abs(math.sqrt(x[0] * x[0] + x[1] * x[1]) - R)
        se = x[3] - vt
        return w_track * dist * dist + w_speed * se * se + w_delta * u[0] * u[0] + w_acc * u[1] * u[1]

    return Problem(f, stage, lambda x: 0.0, x0, T, dt, 2, bounds=(np.array([-0.5, -0.5]), np.array([0.5, 0.5])))


def lqr_identity(x0, T=50, dt=0.1):
    """examples/multi_agent_lqr.cpp:21-76: A=B=Q=R=Qf=I, every derivative analytic"""
    n = 4
    I = np.eye(n)
    analytic = dict(A=lambda x, u: I.copy(), B=lambda x, u: I.copy(), lx=lambda x, u, t: 2 * x, lu=lambda x, u, t: 2 * u,
                    lxx=lambda x, u, t: 2 * I, luu=lambda x, u, t: 2 * I, lux=lambda x, u, t: np.zeros((n, n)),
                    vx=lambda x: 2 * x, vxx=lambda x: 2 * I)
    return Problem(lambda x, u: x + u, lambda x, u, t: float(x @ x + u @ u), lambda x: float(x @ x), x0, T, dt, n, analytic=analytic)


def single_track_lane(x0, T=80, dt=0.1):
    """examples/single_track_ocp.cpp:14-116: analytic A, B (models/single_track_model.hpp:49-82) and l_x, l_u, l_xx, l_uu
    (:62-94); the cross term and the zero terminal cost stay finite-differenced"""
    L, vd, w_lane, w_speed, w_delta, w_acc = 2.5, 1.0, 10.0, 1.0, 0.1, 0.1

    def f(x, u):
        psi, v = x[2], x[3]
        return np.array([v * math.cos(psi), v * math.sin(psi), v * math.tan(u[0]) / L, u[1]])

    def stage(x, u, t):
        return w_lane * (x[1] * x[1]) + w_speed * ((x[3] - vd) * (x[3] - vd)) + w_delta * (u[0] * u[0]) + w_acc * (u[1] * u[1])

    def A(x, u):
        psi, v = x[2], x[3]
        M = np.zeros((4, 4))
        M[0, 2], M[0, 3] = -v * math.sin(psi), math.cos(psi)
        M[1, 2], M[1, 3] = v * math.cos(psi), math.sin(psi)
        M[2, 3] = math.tan(u[0]) / L
        return M

    def B(x, u):
        M = np.zeros((4, 2))
        M[2, 0] = x[3] / (L * math.cos(u[0]) * math.cos(u[0]))
        M[3, 1] = 1.0
        return M

    analytic = dict(A=A, B=B, lx=lambda x, u, t: np.array([0.0, 2.0 * w_lane * x[1], 0.0, 2.0 * w_speed * (x[3] - vd)]),
                    lu=lambda x, u, t: np.array([2.0 * w_delta * u[0], 2.0 * w_acc * u[1]]),
                    lxx=lambda x, u, t: np.diag([0.0, 2.0 * w_lane, 0.0, 2.0 * w_speed]),
                    luu=lambda x, u, t: np.diag([2.0 * w_delta, 2.0 * w_acc]))
    return Problem(f, stage, lambda x: 0.0, x0, T, dt, 2, bounds=(np.array([-0.7, -1.0]), np.array([0.7, 1.0])), analytic=analytic)


def pendulum(T=50, dt=0.05):
    """examples/models/pendulum_model.hpp:8-20; examples/pendulum_swing_up.cpp:29-117"""
    g, l, m, b = 9.81, 1.0, 1.0, 0.1
    mgl = m * g * l

    def f(x, u):
        return np.array([x[1], (g / l) * math.sin(x[0]) + u[0] / (m * l * l) - (b / (m * l * l)) * x[1]])

    def stage(x, u, k):
        s = k / (float(T) - 1.0)
        late = s * s
        early = 1.0 - late
        w_e, w_s, w_o = 2.0 * (0.2 + 0.8 * early), 2.0 * (0.2 + 0.8 * late), 0.05 * (0.2 + 0.8 * late)
        E = 0.5 * m * l * l * x[1] * x[1] + mgl * math.cos(x[0])
        ee = (E - mgl) / mgl
        return w_e * ee * ee + w_s * (1.0 - math.cos(x[0])) + w_o * x[1] * x[1] + 0.05 * u[0] * u[0]

    def terminal(x):
        return 500.0 * (1.0 - math.cos(x[0])) + 100.0 * x[1] * x[1]

    x0 = np.array([math.pi - 0.05, 0.0])
    U0 = np.array([[0.2 * 5.0 * math.sin(2.0 * math.pi * (k * dt)) for k in range(T)]])
    p = Problem(f, stage, terminal, x0, T, dt, 1, bounds=(np.array([-5.0]), np.array([5.0])))
    return p, U0


# ------------------------------------------------------------------ ilqr.hpp:92-193 ----------
def linearise(p: Problem, x, u, t):
    """the seven derivative blocks of one step, ilqr.hpp:106-113"""
    return dict(A=p.hook("A", lambda x, u: fd_dynamics_state_jacobian(p.f, x, u))(x, u),
                B=p.hook("B", lambda x, u: fd_dynamics_control_jacobian(p.f, x, u))(x, u),
                lx=p.hook("lx", lambda x, u, t: fd_cost_state_gradient(p.stage, x, u, t))(x, u, t),
                lu=p.hook("lu", lambda x, u, t: fd_cost_control_gradient(p.stage, x, u, t))(x, u, t),
                lxx=p.hook("lxx", lambda x, u, t: fd_cost_state_hessian(p.stage, x, u, t))(x, u, t),
                luu=p.hook("luu", lambda x, u, t: fd_cost_control_hessian(p.stage, x, u, t))(x, u, t),
                lux=p.hook("lux", lambda x, u, t: fd_cost_cross_term(p.stage, x, u, t))(x, u, t))


def symmetrise(V, aliased=False):
    """v_xx = 0.5 * (v_xx + v_xx.transpose()), ilqr.hpp:102,192.  Mathematically the mean of V and V^T (aliased=False).
    Eigen evaluates this coefficient-wise expression WITHOUT a temporary although source and destination alias, in
    column-major order: entry (i,j) is overwritten with 0.5*(V[i,j] + V[j,i]) using whatever V[j,i] holds at that
    moment, so an upper entry sees its already-rewritten mirror (aliased=True, the C oracle's and the HIP path's
    reading; could not be checked against a real Eigen build)."""
    if not aliased:
        return 0.5 * (V + V.T)
    W = np.array(V, dtype=float)
    n = W.shape[0]
    for j in range(n):
        for i in range(n):
            W[i, j] = 0.5 * (W[i, j] + W[j, i])
    return W


def backward_pass(p: Problem, X, U, derivs_in=None, terminal_in=None, aliased_symmetrisation=False):
    """One linearisation + Riccati sweep on (X, U).  Returns k [T][nu], K [T][nu,nx], reg_events, derivs[t].
    derivs_in / terminal_in: use these derivative blocks (list of dicts per step; (v_x, v_xx)) instead of evaluating
    them -- lets a test feed both restatements the SAME finite-difference noise and compare the recursion alone."""
    T, nu = p.T, p.nu
    nx = X.shape[0]
    if terminal_in is not None:
        vx, vxx = np.array(terminal_in[0], dtype=float), np.array(terminal_in[1], dtype=float)
    else:
        vx = p.hook("vx", lambda x: fd_terminal_gradient(p.terminal, x))(X[:, T])              # :92-95
        vxx = p.hook("vxx", lambda x: fd_terminal_hessian(p.terminal, x))(X[:, T])             # :97-100
    vxx = symmetrise(vxx, aliased_symmetrisation)                                               # :102
    ks, Ks, derivs, regs = [None] * T, [None] * T, [None] * T, 0
    I = np.eye(nu)
    for t in range(T - 1, -1, -1):
        x, u = X[:, t], U[:, t]
        d = derivs_in[t] if derivs_in is not None else linearise(p, x, u, t)
        A, B, lx, lu, lxx, luu, lux = (np.asarray(d[n], dtype=float) for n in ("A", "B", "lx", "lu", "lxx", "luu", "lux"))
        qx = lx + A.T @ vx                                                                      # :115-119
        qu = lu + B.T @ vx
        qxx = lxx + A.T @ vxx @ A
        qux = lux + B.T @ vxx @ A
        quu = luu + B.T @ vxx @ B
        quu_reg, reg = quu.copy(), 1e-6                                                         # :172-182
        while True:
            try:
                Lf = np.linalg.cholesky(quu_reg)
                if np.isfinite(Lf).all() or not np.isfinite(quu_reg).all():
                    break
            except np.linalg.LinAlgError:
                pass
            if not math.isfinite(reg):
                Lf = np.full((nu, nu), np.nan)          # Eigen's LLT "succeeds" on a +inf / NaN pivot (not <= 0)
                break
            quu_reg = quu_reg + reg * I
            reg *= 10.0
            regs += 1
        quu_inv = np.linalg.solve(Lf @ Lf.T, I) if np.isfinite(Lf).all() else np.full((nu, nu), np.nan)   # :183
        k = -quu_inv @ qu                                                                       # :185-186
        K = -quu_inv @ qux
        vx = qx + K.T @ qu + qux.T @ k + K.T @ quu @ k                                          # :188-191 (unregularised q_uu)
        vxx = qxx + K.T @ qux + qux.T @ K + K.T @ quu @ K
        vxx = symmetrise(vxx, aliased_symmetrisation)                                           # :192
        ks[t], Ks[t] = k, K
        derivs[t] = d
    return ks, Ks, regs, derivs


# ------------------------------------------------------------------ ilqr.hpp:195-228 ---------
def line_search(p: Problem, X, U, ks, Ks, merit):
    """alpha = 1, 1/2, ... while alpha >= 1e-3; first strict decrease of the merit wins.
    Returns (X, U, best_merit, accepted alpha or None)."""
    alpha = 1.0
    while alpha >= 1e-3:                                                                        # :199-206
        Xt = np.zeros_like(X)
        Ut = np.zeros_like(U)
        Xt[:, 0] = p.x0
        for t in range(p.T):
            Ut[:, t] = U[:, t] + alpha * ks[t] + Ks[t] @ (Xt[:, t] - X[:, t])                   # :210-211
            if p.bounds is not None:                                                            # :213-214, constraint_helpers.hpp:107-114
                Ut[:, t] = np.maximum(np.minimum(Ut[:, t], p.bounds[1]), p.bounds[0])
            Xt[:, t + 1] = integrate_rk4(Xt[:, t], Ut[:, t], p.dt, p.f)                         # :216
        trial = trajectory_cost(Xt, Ut, p.stage, p.terminal)                                    # :219
        if trial < merit:                                                                       # :220-226
            return Xt, Ut, trial, alpha
        alpha *= 0.5
    return X, U, merit, None


# ------------------------------------------------------------------ ilqr.hpp:58-273 ----------
def solve(p: Problem, U0, max_iterations=100, tolerance=1e-5):
    """iLQR::solve without constraint functions (merit == objective, :379-407) and with max_ms = infinity.
    Returns dict(X, U, cost, iterations, costs=[cost after each iteration], alphas=[accepted alpha or None])."""
    U = np.array(U0, dtype=float)
    X = integrate_horizon(p.x0, U, p.dt, p.f)                                                   # :75
    cost = trajectory_cost(X, U, p.stage, p.terminal)                                           # :76
    merit = cost                                                                                # :78
    costs, alphas, it = [], [], 0
    for it in range(1, max_iterations + 1):                                                     # :82
        ks, Ks, _, _ = backward_pass(p, X, U)
        X, U, best_merit, accepted = line_search(p, X, U, ks, Ks, merit)
        improvement = merit - best_merit                                                        # :230
        cost = trajectory_cost(X, U, p.stage, p.terminal)                                       # :233
        merit = best_merit
        costs.append(cost)
        alphas.append(accepted)
        if improvement < tolerance:                                                             # :269-271
            break
    return dict(X=X, U=U, cost=cost, iterations=it if max_iterations > 0 else 0, costs=costs, alphas=alphas)
