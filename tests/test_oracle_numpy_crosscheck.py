"""The C oracle (oracle/mas_oracle.c) against a second, independently written restatement of iLQR::solve
(oracle/ilqr_numpy.py: dense numpy / LAPACK, written from ilqr.hpp:58-273 and finite_differences.hpp alone).

Neither is the reference, and the reference holds no fixture for iLQR::solve, so this is the
strongest pin available for a6-a9: two restatements in different styles must tell the same story.  They cannot be
bit-equal (LAPACK orders sums differently; numpy's 0.5*(V+V^T) is the true mean while Eigen's aliased statement is
not), so the comparisons are at finite-difference-noise tolerance, stated per assertion, and
full solves are compared at short horizons where the iteration is benign (tests/test_oracle_sensitivity.py).
CPU only; glibc trig on both sides."""
import numpy as np
import pytest

from oracle import ilqr_numpy as R


@pytest.fixture()
def libm_oracle(oracle):
    oracle.set_trig_mode("libm")
    yield oracle
    oracle.set_trig_mode("det")


def _cols(a):            # oracle arrays are [T, n] (one row per step); the numpy restatement uses n x T columns
    return np.ascontiguousarray(a.T)


def test_rollout_and_objective_agree_to_rounding(libm_oracle):
    O = libm_oracle
    rng = np.random.default_rng(3)
    x0 = O.single_track_x0(5)[2]
    U = 0.3 * rng.standard_normal((40, 2))
    b = O.backward_pass(O.MODEL_SINGLE_TRACK, x0, U, 40, 0.5)
    p = R.single_track(x0, 40, 0.5)
    X = R.integrate_horizon(p.x0, _cols(U), p.dt, p.f)
    # same formulas, same libm, numpy vector ops instead of scalar loops: agreement to a few ulp of the state norm
    assert np.abs(X.T - b["X"]).max() <= 1e-12 * (1 + np.abs(b["X"]).max())
    assert abs(R.trajectory_cost(X, _cols(U), p.stage, p.terminal) - b["cost"]) <= 1e-12 * abs(b["cost"])


@pytest.mark.parametrize("model", ["single_track", "pendulum", "lqr"])
def test_backward_pass_derivatives_and_gains(libm_oracle, model):
    """ilqr.hpp:92-193 once, on the same trajectory: FD derivatives to FD noise, gains to the noise the Riccati
    recursion carries over T steps."""
    O = libm_oracle
    rng = np.random.default_rng(5)
    if model == "single_track":
        T, dt, mid = 20, 0.5, O.MODEL_SINGLE_TRACK
        x0 = O.single_track_x0(7)[3]
        U = 0.2 * rng.standard_normal((T, 2))
        p = R.single_track(x0, T, dt)
    elif model == "pendulum":
        T, dt, mid = 20, 0.05, O.MODEL_PENDULUM
        p, U0 = R.pendulum(T, dt)
        x0, U = p.x0, U0.T.copy()
    else:
        T, dt, mid = 20, 0.1, O.MODEL_LQR_IDENTITY
        x0 = np.array([1.0, -0.5, 0.25, 2.0])
        U = 0.2 * rng.standard_normal((T, 4))
        p = R.lqr_identity(x0, T, dt)
    b = O.backward_pass(mid, x0, U, T, dt)
    X = R.integrate_horizon(p.x0, _cols(U), p.dt, p.f)
    ks, Ks, regs, derivs = R.backward_pass(p, X, _cols(U))
    for t in range(T):
        d = derivs[t]
        ref = {n: np.asarray(b["derivs"][n][t]).reshape(np.shape(d[n])) for n in d}    # the oracle wrapper squeezes nu = 1
        # first derivatives, eps = 1e-6: the two evaluate the same difference quotient; rounding of f enters as ~1e-16/1e-6
        np.testing.assert_allclose(d["A"], ref["A"], rtol=0, atol=1e-8)
        np.testing.assert_allclose(d["B"], ref["B"], rtol=0, atol=1e-8)
        scale = 1.0 + np.abs(ref["lx"]).max()
        np.testing.assert_allclose(d["lx"], ref["lx"], rtol=0, atol=1e-7 * scale)
        np.testing.assert_allclose(d["lu"], ref["lu"], rtol=0, atol=1e-7 * scale)
        # second differences, eps = 1e-5 / 1e-6: rounding of the cost (|c| * 2^-52) is divided by eps^2
        c = abs(p.stage(X[:, t], _cols(U)[:, t], t)) + 1.0
        np.testing.assert_allclose(d["lxx"], ref["lxx"], rtol=0, atol=4 * c * 2.3e-16 / 1e-10 * 4)
        np.testing.assert_allclose(d["luu"], ref["luu"], rtol=0, atol=4 * c * 2.3e-16 / 1e-10 * 4)
        np.testing.assert_allclose(d["lux"], ref["lux"], rtol=0, atol=4 * c * 2.3e-16 / 1e-12)
    assert regs == b["reg_events"]                       # the x10 ladder takes the same number of rungs
    kn, Kn = np.array(ks).reshape(b["k"].shape), np.array(Ks).reshape(b["K"].shape)
    if model == "lqr":                                    # analytic derivatives: only LAPACK-vs-loop rounding is left
        np.testing.assert_allclose(kn, b["k"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(Kn, b["K"], rtol=0, atol=1e-13)
    else:                                                 # FD noise in l_uu (~1e-6 of 2e-3) carried through T steps
        np.testing.assert_allclose(kn, b["k"], rtol=2e-2, atol=2e-3 * (1 + np.abs(b["k"]).max()))
        np.testing.assert_allclose(Kn, b["K"], rtol=2e-2, atol=2e-3 * (1 + np.abs(b["K"]).max()))


def _oracle_derivs(b, T):
    names = ("A", "B", "lx", "lu", "lxx", "luu", "lux")
    return [{n: np.atleast_1d(b["derivs"][n][t]) for n in names} for t in range(T)]


@pytest.mark.parametrize("model", ["single_track", "pendulum"])
def test_riccati_recursion_on_identical_derivatives(libm_oracle, model):
    """a6 alone: both restatements are fed the SAME derivative blocks (the C oracle's), so finite-difference noise
    drops out and what is compared is ilqr.hpp:115-119,172-192 -- Q assembly, the x10 ladder, LLT solve, gains, value
    update, symmetrisation -- hand-written loops against LAPACK.  Run twice: with the C oracle's reading of the
    aliased `v_xx = 0.5*(v_xx + v_xx^T)` (Eigen evaluates it in place) the gains agree to 1e-9
    relative; with the mathematical mean instead they agree to 1e-3 (V is asymmetric only through FD noise, and the
    recursion amplifies that over the horizon) -- which bounds what the one Eigen reading that could not be checked
    against a real Eigen build can change.  Ladder rungs must be equal in both."""
    O = libm_oracle
    rng = np.random.default_rng(11)
    if model == "single_track":
        T, dt, mid = 30, 0.5, O.MODEL_SINGLE_TRACK
        x0 = O.single_track_x0(7)[5]
        U = 0.2 * rng.standard_normal((T, 2))
        p = R.single_track(x0, T, dt)
    else:
        T, dt, mid = 30, 0.05, O.MODEL_PENDULUM
        p, U0 = R.pendulum(T, dt)
        x0, U = p.x0, U0.T.copy()
    b = O.backward_pass(mid, x0, U, T, dt)
    d_in = _oracle_derivs(b, T)
    if model == "pendulum":
        for d in d_in:
            d["A"] = d["A"].reshape(2, 2); d["B"] = d["B"].reshape(2, 1); d["lxx"] = d["lxx"].reshape(2, 2)
            d["luu"] = d["luu"].reshape(1, 1); d["lux"] = d["lux"].reshape(1, 2)
    # the swing-up recursion is itself ill-conditioned (955 ladder rungs in 30 steps, gains up to 2.7e3): rounding
    # differences between LAPACK and the loops grow from 1e-10 at step T-10 to 2e-4 at step 0, so the sharp
    # comparison covers the last ten steps there and the whole horizon gets the loose bound
    sharp = slice(0, T) if model == "single_track" else slice(T - 10, T)
    for aliased, rtol in ((True, 1e-8), (False, 1e-3)):
        ks, Ks, regs, _ = R.backward_pass(p, b["X"].T.copy(), _cols(U), derivs_in=d_in, aliased_symmetrisation=aliased)
        assert regs == b["reg_events"]
        kn, Kn = np.array(ks).reshape(b["k"].shape), np.array(Ks).reshape(b["K"].shape)
        for sl, tol in ((sharp, rtol), (slice(0, T), 1e-3)):
            np.testing.assert_allclose(kn[sl], b["k"][sl], rtol=tol, atol=tol * 1e-2 * (1 + np.abs(b["k"][sl]).max()))
            np.testing.assert_allclose(Kn[sl], b["K"][sl], rtol=tol, atol=tol * 1e-2 * (1 + np.abs(b["K"][sl]).max()))


@pytest.mark.parametrize("model", ["single_track", "pendulum"])
def test_one_iteration_on_identical_gains(libm_oracle, model):
    """a7/a8 alone: the numpy line search (ilqr.hpp:195-228: alpha ladder, feedback law, clamp, RK4, first strict
    decrease) run with the C oracle's gains must accept the same alpha and land on the cost the C oracle reports
    after max_iterations = 1, to rounding."""
    O = libm_oracle
    if model == "single_track":
        T, dt, mid = 30, 0.5, O.MODEL_SINGLE_TRACK
        x0s = O.single_track_x0(5)
        cases = [(x0s[a], np.zeros((T, 2)), R.single_track(x0s[a], T, dt)) for a in range(5)]
    else:
        T, dt, mid = 30, 0.05, O.MODEL_PENDULUM
        p, U0 = R.pendulum(T, dt)
        cases = [(p.x0, U0.T.copy(), p)]
    for x0, U, p in cases:
        b = O.backward_pass(mid, x0, U, T, dt)
        ks = [np.atleast_1d(b["k"][t]) for t in range(T)]
        Ks = [np.atleast_2d(b["K"][t]).reshape(p.nu, p.x0.size) for t in range(T)]
        X = b["X"].T.copy()
        Xn, Un, merit, alpha = R.line_search(p, X, _cols(U), ks, Ks, b["cost"])
        o = O.solve_batch(mid, x0.reshape(1, -1), U.reshape(1, T, -1), T, dt, strategy="single", max_iterations=1, tolerance=1e-5)
        assert abs(merit - o["cost"][0]) <= 1e-11 * abs(o["cost"][0]), (merit, o["cost"][0])
        np.testing.assert_allclose(Xn.T, o["X"][0], rtol=0, atol=1e-10 * (1 + np.abs(o["X"][0]).max()))
        np.testing.assert_allclose(Un.T, o["U"][0], rtol=0, atol=1e-10)
        hist = o["stats"]["alpha_hist"]           # index j counts acceptances of alpha = 2^-j, index 10 = no acceptance
        j = 10 if alpha is None else int(round(-np.log2(alpha)))
        assert hist[j] == 1 and sum(hist) == 1


def test_full_solve_lqr_is_the_same_iteration(libm_oracle):
    """Analytic derivatives: nothing chaotic is left, the two restatements must take the same path:
    same iteration count, costs to 1e-12."""
    O = libm_oracle
    x0 = O.lqr_x0(1)
    o = O.solve_batch(O.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy="single")
    r = R.solve(R.lqr_identity(x0[0], 50, 0.1), np.zeros((4, 50)))
    assert r["iterations"] == int(o["iters"][0])
    assert abs(r["cost"] - o["cost"][0]) <= 1e-12 * o["cost"][0]
    np.testing.assert_allclose(r["X"].T, o["X"][0], rtol=0, atol=1e-11)


def test_full_solve_lane_following_mostly_analytic(libm_oracle):
    """examples/single_track_ocp.cpp: analytic A, B, l_x, l_u, l_xx, l_uu, only the cross term is finite-differenced
    (a separable cost: l_ux is pure rounding noise ~ c * 2^-52 / 4e-12).  The reference program runs 10 iterations."""
    O = libm_oracle
    x0 = np.array([[0.0, 1.0, 0.0, 0.0]])
    o = O.solve_batch(O.MODEL_SINGLE_TRACK_LANE, x0, None, 80, 0.1, strategy="single", max_iterations=10)
    r = R.solve(R.single_track_lane(x0[0], 80, 0.1), np.zeros((2, 80)), max_iterations=10)
    assert r["iterations"] == int(o["iters"][0])
    assert abs(r["cost"] - o["cost"][0]) <= 1e-10 * o["cost"][0], (r["cost"], o["cost"][0])    # measured: 1e-16
    assert abs(r["cost"] - 508.59306030493815) <= 1e-9      # tests/golden/reference_readme_costs.json records the README's 24.4039 as unreproducible


@pytest.mark.parametrize("T", [10, 20])
def test_full_solve_single_track_short_horizon(libm_oracle, T):
    """All derivatives finite-differenced: l_uu = 2e-3 carries noise c*2^-52/eps^2 ~ 2e-4 and l_ux (true value 0) noise
    ~5e-3 at the start (c ~ 90), so the iterates of ANY two evaluations differ after the first unsaturated step.  The
    yardstick is the C oracle against itself with v0 moved by one ulp (tests/test_oracle_sensitivity.py: < 1e-2 at
    H = 20): the two restatements must agree at least that well, and exactly while the clamp saturates the step."""
    O = libm_oracle
    x0 = O.single_track_x0(6)
    o = O.solve_batch(O.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="single")
    o1 = O.solve_batch(O.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="single", max_iterations=1)
    for a in range(6):
        r = R.solve(R.single_track(x0[a], T, 0.5), np.zeros((2, T)))
        assert abs(r["costs"][0] - o1["cost"][a]) <= 1e-9 * o1["cost"][a]          # first iteration: same step
        assert abs(r["cost"] - o["cost"][a]) <= 2e-3 * o["cost"][a], (a, r["cost"], o["cost"][a])     # measured: 0 .. 8.6e-4
        assert abs(r["iterations"] - int(o["iters"][a])) <= 1                                             # measured: equal
        assert np.abs(r["U"]).max() <= 0.5


def test_full_solve_pendulum_first_iterations(libm_oracle):
    """Pendulum swing-up walks hundreds of ladder rungs and is as chaotic as the single track at full length; what the
    two restatements must share is the start."""
    O = libm_oracle
    T, dt = 20, 0.05
    p, U0 = R.pendulum(T, dt)
    x0 = p.x0.reshape(1, 2)
    for its, tol in ((1, 1e-6), (2, 1e-4), (3, 1e-3)):
        o = O.solve_batch(O.MODEL_PENDULUM, x0, U0.T.reshape(1, T, 1), T, dt, strategy="single", max_iterations=its, tolerance=1e-4)
        r = R.solve(p, U0, max_iterations=its, tolerance=1e-4)
        assert abs(r["cost"] - o["cost"][0]) <= tol * abs(o["cost"][0]), (its, r["cost"], o["cost"][0])
