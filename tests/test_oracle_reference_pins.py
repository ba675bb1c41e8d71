"""The oracle against the reference's own gtest known answers (tests/ocp_tests.cpp) and the
oracle's closed-form known answers.  CPU only."""
import json
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = json.load(open(os.path.join(HERE, "golden", "reference_unit_tests.json")))


def test_initialize_problem_sets_defaults_and_best_cost(oracle):
    out = oracle.ref_case("initialize")
    exp = REF["initialize"]
    assert out[0] == exp["best_cost"]                  # EXPECT_DOUBLE_EQ(ocp.best_cost, 0.0)
    assert np.isfinite(out[1]) and np.isfinite(out[2])  # FD hooks installed, right sizes
    assert out[3:7].tolist() == exp["states"]


def test_update_initial_with_best_copies_trajectories(oracle):
    out = oracle.ref_case("update_initial")
    assert out[:4].tolist() == REF["update_initial"]["initial_controls"]
    assert out[4:].tolist() == REF["update_initial"]["initial_states"]


def test_build_global_problem_merges_agents(oracle):
    out = oracle.ref_case("global")
    exp = REF["global"]
    assert out[0:2].tolist() == exp["block_ids"]
    assert out[2:5].tolist() == [exp["state_dim"], exp["control_dim"], exp["horizon_steps"]]
    assert out[5] == exp["dt"]
    assert out[6:9].tolist() == exp["input_lower_bounds"]
    assert out[9:12].tolist() == exp["initial_state"]
    assert out[12:15].tolist() == exp["dynamics_at_linspaced"]      # EXPECT_DOUBLE_EQ on each
    assert out[15] == exp["stage_cost_at_linspaced"]
    assert out[16] == exp["terminal_cost_at_linspaced"]
    assert out[17] == 1.0


def test_fd_gradient_matches_analytical_for_quadratic_objective(oracle):
    out = oracle.ref_case("fd_gradient")
    exp = REF["fd_gradient"]
    assert abs(out[0] - exp["expected"][0]) <= exp["abs_tol"]
    assert abs(out[1] - exp["expected"][1]) <= exp["abs_tol"]
    assert out[2:5].tolist() == exp["rk4_of_xdot_equals_u"]          # RK4 of x' = u is exactly x + dt*u


# ---- closed-form known answers (reference-independent) ----------------
def test_lqr_rollout_is_the_degree4_taylor_map(oracle):
    # RK4 on x' = x + u with a zero-order hold is x+ = p(h) x + (p(h) - 1) u, p = 1+h+h^2/2+h^3/6+h^4/24
    T, h = 10, 0.1
    p = 1 + h + h * h / 2 + h ** 3 / 6 + h ** 4 / 24
    rng = np.random.default_rng(0)
    U = rng.standard_normal((T, 4))
    b = oracle.backward_pass(oracle.MODEL_LQR_IDENTITY, [1.0, -2.0, 0.5, 0.0], U, T, h)
    x = np.array([1.0, -2.0, 0.5, 0.0])
    for t in range(T):
        np.testing.assert_allclose(b["X"][t], x, rtol=0, atol=1e-13 * (1 + np.abs(x).max()))
        x = p * x + (p - 1) * U[t]
    np.testing.assert_allclose(b["X"][T], x, rtol=0, atol=1e-13 * (1 + np.abs(x).max()))


def test_lqr_backward_pass_is_the_scalar_riccati_recursion(oracle):
    # A=B=I used as the discrete Jacobians (the reference's convention, ilqr.hpp:106-119),
    # l_xx=l_uu=2I, V_T=2I  =>  per coordinate: K = -V/(2+V), V <- 2 + V - V^2/(2+V)
    T, h = 50, 0.1
    b = oracle.backward_pass(oracle.MODEL_LQR_IDENTITY, [1.0, 0.0, 0.0, 0.0], np.zeros((T, 4)), T, h)
    V = 2.0
    for t in range(T - 1, -1, -1):
        K = -V / (2 + V)
        np.testing.assert_allclose(b["K"][t], K * np.eye(4), rtol=0, atol=2e-15)
        V = 2 + V - V * V / (2 + V)
    assert b["reg_events"] == 0


def test_fd_derivatives_of_the_single_track_cost_match_calculus(oracle):
    # l_uu of the example cost is exactly 2*w*I = 2e-3*I; central differences recover it to FD accuracy
    x0 = oracle.single_track_x0(5)[3]
    b = oracle.backward_pass(oracle.MODEL_SINGLE_TRACK, x0, np.full((4, 2), 0.1), 4, 0.5)
    d = b["derivs"]
    np.testing.assert_allclose(d["luu"][0], 2e-3 * np.eye(2), atol=2e-4)
    np.testing.assert_allclose(d["lu"][0], [2e-3 * 0.1, 2e-3 * 0.1], atol=1e-8)
    x = b["X"][0]
    r = math.hypot(x[0], x[1])
    gx = 2 * (r - 20.0) * x[0] / r
    np.testing.assert_allclose(d["lx"][0][0], gx, atol=1e-6)
    np.testing.assert_allclose(d["lx"][0][3], 2 * (x[3] - 5.0), atol=1e-8)
    # A: d(v cos psi)/d psi = -v sin psi etc. (continuous-time Jacobian, no I + dt*A)
    np.testing.assert_allclose(d["A"][0][0, 2], -x[3] * math.sin(x[2]), atol=1e-8)
    np.testing.assert_allclose(d["A"][0][1, 3], math.sin(x[2]), atol=1e-8)
    np.testing.assert_allclose(d["B"][0][3, 1], 1.0, atol=1e-9)


def test_oracle_matches_committed_golden_vectors(oracle):
    """Regression pin of the oracle (vectors are oracle-generated; see make_oracle_vectors.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mov", os.path.join(HERE, "golden", "make_oracle_vectors.py"))
    gold = json.load(open(os.path.join(HERE, "golden", "oracle_vectors.json")))["cases"]
    mov = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mov)
    for name, c in gold.items():
        x0, U0 = mov.inputs(c["model"], c["N"], c["T"], c["dt"])
        r = oracle.solve_batch(c["model"], x0, U0, c["T"], c["dt"], strategy=c["strategy"], max_outer=c["max_outer"],
                               max_iterations=c["max_iterations"], tolerance=c["tolerance"], threads=1)
        exp = np.array([float.fromhex(v) for v in c["cost"]])
        assert np.array_equal(r["cost"], exp), name      # det trig mode: pure IEEE arithmetic, exact on any host
        assert r["iters"].tolist() == c["iters"], name
        assert [[float.hex(v) for v in r["X"][a, -1]] for a in range(c["N"])] == c["x_final"], name


def test_centralized_blocked_agrees_with_dense_stacked_solve(oracle):
    """The block-diagonal reading (what the HIP path implements) against the literal stacked
    solve (centralized.hpp + build_global_ocp) at small N: same cost up to the FD noise the
    dense Hessian carries in its cross-agent blocks."""
    x0 = oracle.single_track_x0(4)
    dense = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 10, 0.5, strategy="centralized_dense")
    blocked = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 10, 0.5, strategy="centralized")
    assert abs(dense["total_cost"] - blocked["total_cost"]) / blocked["total_cost"] < 5e-3
