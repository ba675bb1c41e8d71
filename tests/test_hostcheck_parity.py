"""Host logic + lane arithmetic of the product sources, replayed on the CPU (tests/hostcheck,
-DMAS_HOST_EMULATION) and compared with the independent oracle.  Same libm on both sides, so
agreement is BIT-EXACT: costs, trajectories, iteration counts and regularisation events.
The HIP build of the same sources is checked on the GPU by tests/test_gpu_parity.py."""
import math
import os

import numpy as np
import pytest

from conftest import run_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(e, o, total_exact=True):
    assert np.array_equal(e["cost"], o["cost"])
    assert np.array_equal(e["X"], o["X"])
    assert np.array_equal(e["U"], o["U"])
    assert e["iters"].tolist() == o["iters"].tolist()
    assert e["stats"]["reg_events"] == o["stats"]["reg_events"]
    if "radii" in o:                                # trust-region strategy: the radii every agent ends with (nash.hpp:231,241)
        assert "radii" in e and np.array_equal(e["radii"], o["radii"])
    if total_exact:
        assert e["total_cost"] == o["total_cost"]
    else:
        assert abs(e["total_cost"] - o["total_cost"]) <= 1e-12 * abs(o["total_cost"])


@pytest.mark.parametrize("strategy", ["single", "sequential", "trustregion", "linesearch"])
def test_single_track_h100_strategies(hostcheck_lib, oracle, strategy):
    x0 = oracle.single_track_x0(8)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy=strategy)
    _same(e, o)
    assert e["stats"]["agent_iterations"] == int(o["iters"].sum())


def test_single_track_centralized_blocked(hostcheck_lib, oracle):
    x0 = oracle.single_track_x0(8)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, "centralized")
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy="centralized")
    _same(e, o, total_exact=False)   # merit summed agent-major here, step-major in the oracle


def test_ragged_agent_count_and_padding(hostcheck_lib, oracle):
    x0 = oracle.single_track_x0(70)     # 70 agents -> padded to 128 lanes
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 20, 0.5, "sequential", max_outer=3)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 20, 0.5, strategy="sequential", max_outer=3)
    _same(e, o)


def test_single_agent_and_short_horizon(hostcheck_lib, oracle):
    x0 = oracle.single_track_x0(1)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 1, 0.5, "single")
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 1, 0.5, strategy="single")
    _same(e, o)


def test_lqr_identity_sequential(hostcheck_lib, oracle):
    x0 = oracle.lqr_x0(3)
    e = run_batch(hostcheck_lib, oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, "sequential")
    o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy="sequential")
    _same(e, o)
    assert e["iters"].tolist() == [109, 109, 109]      # first solve hits max_iterations=100, then 9 x 1


@pytest.mark.parametrize("strategy,outer", [("sequential", 10), ("trustregion", 3), ("linesearch", 2)])
def test_lqr_identity_against_the_oracle_in_libm_mode(hostcheck_lib, oracle, strategy, outer):
    """no trigonometry in this workload: the product's lane arithmetic must meet the oracle's glibc-faithful mode bit for bit too --
    the parity statement that does not pass through include/mas_dettrig.h (the GPU tier repeats it on the device)"""
    rng = np.random.default_rng(5)
    x0 = oracle.lqr_x0(9) + 0.2 * rng.standard_normal((9, 4))
    oracle.set_trig_mode("libm")
    try:
        o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy=strategy, max_outer=outer)
    finally:
        oracle.set_trig_mode("det")
    e = run_batch(hostcheck_lib, oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy, max_outer=outer)
    _same(e, o)


@pytest.mark.parametrize("T", [50, 60])
def test_pendulum_plumbing_config(hostcheck_lib, oracle, T):
    x0, U0 = oracle.pendulum_problem(T, 0.05)
    e = run_batch(hostcheck_lib, oracle.MODEL_PENDULUM, x0, U0, T, 0.05, "single", max_iterations=1000, tolerance=1e-4)
    o = oracle.solve_batch(oracle.MODEL_PENDULUM, x0, U0, T, 0.05, strategy="single", max_iterations=1000, tolerance=1e-4)
    _same(e, o)


def test_lane_following_analytic_derivatives(hostcheck_lib, oracle):
    x0 = np.array([[0.0, 1.0, 0.0, 0.0]])
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK_LANE, x0, None, 80, 0.1, "single", max_iterations=10)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LANE, x0, None, 80, 0.1, strategy="single", max_iterations=10)
    _same(e, o)


def test_bicycle6_trust_region(hostcheck_lib, oracle):
    x0 = oracle.bicycle6_x0(4)
    e = run_batch(hostcheck_lib, oracle.MODEL_BICYCLE6, x0, None, 200, 0.25, "trustregion")
    o = oracle.solve_batch(oracle.MODEL_BICYCLE6, x0, None, 200, 0.25, strategy="trustregion")
    _same(e, o)


def test_staged_backward_pass(hostcheck_lib, oracle):
    """linearisation sweep + Riccati recursion outputs, stage by stage"""
    from multi_agent_solver_amd.batch import BatchSolver
    x0 = oracle.single_track_x0(5)
    rng = np.random.default_rng(1)
    U = 0.3 * rng.standard_normal((5, 30, 2))
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, 5, 30, 0.5, lib=hostcheck_lib) as s:
        s.set_initial_states(x0)
        s.set_controls(U)
        s.initialize_problem()
        got = s.debug_backward()
    for a in range(5):
        ref = oracle.backward_pass(oracle.MODEL_SINGLE_TRACK, x0[a], U[a], 30, 0.5)
        for name in ("A", "B", "lx", "lu", "lxx", "luu", "lux"):
            assert np.array_equal(got["derivs"][name][a], ref["derivs"][name]), name
        assert np.array_equal(got["k"][a], ref["k"])
        assert np.array_equal(got["K"][a], ref["K"])


def test_missing_required_params_raise_like_params_at(hostcheck_lib):
    from multi_agent_solver_amd.batch import BatchSolver
    with BatchSolver(1, 2, 5, 0.5, lib=hostcheck_lib) as s:
        with pytest.raises(KeyError):
            s.set_params({"max_iterations": 10, "tolerance": 1e-5})     # no max_ms


def test_max_iterations_limit_and_status(hostcheck_lib, oracle):
    x0 = oracle.single_track_x0(4)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, "single", max_iterations=2)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy="single", max_iterations=2)
    _same(e, o)
    assert set(e["status"].tolist()) <= {0, 1}
    assert (e["iters"] <= 2).all()


# ---- augmented-Lagrangian branch of iLQR (ilqr.hpp:121-170,236-260,379-407) ----------
@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 3), ("trustregion", 3), ("linesearch", 3)])
def test_constrained_single_track_augmented_lagrangian(hostcheck_lib, oracle, strategy, outer):
    x0 = oracle.single_track_x0(5)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, 30, 0.5, strategy, max_outer=outer)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, 30, 0.5, strategy=strategy, max_outer=outer)
    _same(e, o)
    if strategy == "sequential":            # the constraints bite: speed limit 4.5 < target speed 5, delta = 0.1 a
        un = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 30, 0.5, strategy, max_outer=outer)
        assert un["X"][:, :, 3].max() > 5.0 and e["X"][:, :, 3].max() < 4.6          # unconstrained runs exceed the limit
        assert np.abs(e["U"][:, :, 0] - 0.1 * e["U"][:, :, 1]).max() < 0.05


def test_constrained_model_optional_penalty_keys(hostcheck_lib, oracle):
    from multi_agent_solver_amd.batch import BatchSolver
    x0 = oracle.single_track_x0(3)
    keys = dict(penalty=3.0, penalty_increase=2.0, constraint_tolerance=1e-3, inequality_activation_tolerance=1e-5)
    with BatchSolver(oracle.MODEL_SINGLE_TRACK_LIMITED, 3, 20, 0.5, lib=hostcheck_lib) as s:
        s.set_params({"max_iterations": 60, "tolerance": 1e-5, "max_ms": math.inf, **keys})
        s.set_initial_states(x0); s.set_controls(None); s.initialize_problem()
        s.run_sequential(2)
        e = s.solution()
    try:
        oracle.set_al_params(*keys.values())
        o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, 20, 0.5, strategy="sequential", max_outer=2, max_iterations=60)
    finally:
        oracle.set_al_params()
    assert np.array_equal(e["cost"], o["cost"]) and e["iters"].tolist() == o["iters"].tolist()


def test_centralized_ignores_function_constraints_like_the_stacked_ocp(hostcheck_lib, oracle):
    # build_global_ocp (multi_agent_problem.hpp:52-127) stacks dynamics, costs and bounds -- no constraint functions
    x0 = oracle.single_track_x0(4)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, 20, 0.5, "centralized")
    u = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 20, 0.5, "centralized")
    assert np.array_equal(e["cost"], u["cost"]) and np.array_equal(e["X"], u["X"])


def test_exit_codes_and_the_time_limit(hostcheck_lib, oracle):
    """ilqr.hpp:82-90,269-271 through the host loop: converged / max_iterations / elapsed_ms > max_ms.  max_ms = -1 is
    the deterministic use of the clock (0 > -1 at the first check: no iteration runs); a generous finite budget must
    change nothing, including the chained solves of the sequential strategy (per-solve start stamps)."""
    from multi_agent_solver_amd.batch import EXIT_CONVERGED, EXIT_MAX_ITER, EXIT_TIME
    N, T, dt = 5, 30, 0.5
    x0 = oracle.single_track_x0(N)
    init = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, "single", max_iterations=0)
    assert (init["iters"] == 0).all() and (init["status"] == EXIT_MAX_ITER).all()
    for strategy, outer in (("single", 1), ("sequential", 3), ("trustregion", 2), ("linesearch", 2)):
        for mi, ms, code in ((100, math.inf, EXIT_CONVERGED), (2, math.inf, EXIT_MAX_ITER), (100, -1.0, EXIT_TIME), (100, 1e7, EXIT_CONVERGED)):
            e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy, max_outer=outer, max_iterations=mi, max_ms=ms)
            o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy=strategy, max_outer=outer, max_iterations=mi, max_ms=ms)
            _same(e, o)
            assert (e["status"] & 3).tolist() == o["status"].tolist(), (strategy, mi, ms)      # last solve of every agent
            if ms < 0 or (strategy == "single" and mi == 100):
                assert (e["status"] == code).all(), (strategy, mi, ms, e["status"])
            elif strategy == "single":
                assert (e["status"] == code).sum() > 0
            if ms < 0:
                assert np.array_equal(e["X"], init["X"]) and np.array_equal(e["cost"], init["cost"]) and (e["iters"] == 0).all()


def test_debug_key_prints_the_reference_trace(hostcheck_lib, oracle, capfd):
    """`debug` (ilqr.hpp:45,79-80,262-267): one "initial" line per solve and one line per iteration of every agent, with
    the reference's wording; values at stream precision (6 significant digits)."""
    from multi_agent_solver_amd.batch import BatchSolver
    x0 = oracle.single_track_x0(2)
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, 2, 20, 0.5, lib=hostcheck_lib) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf, "debug": 1.0})
        s.set_initial_states(x0); s.set_controls(None); s.initialize_problem()
        s.solve()
        e = s.solution()
    out = capfd.readouterr().out.splitlines()
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 20, 0.5, strategy="single")
    assert np.array_equal(e["cost"], o["cost"])
    for a in range(2):
        mine = [ln for ln in out if ln.startswith(f"[agent {a}] ")]
        assert mine[0].startswith(f"[agent {a}] iLQR initial cost=") and " merit=" in mine[0]
        its = [ln for ln in mine if " iLQR iter " in ln]
        assert len(its) == int(o["iters"][a])
        assert [int(ln.split("iLQR iter ")[1].split(":")[0]) for ln in its] == list(range(len(its)))
        last = its[-1]
        assert f"cost={o['cost'][a]:.6g}" in last and "d_merit=" in last and last.endswith("eq_violation=0 ineq_violation=0")


@pytest.mark.parametrize("strategy,outer", [("sequential", 4), ("trustregion", 4)])
def test_perturbed_agents_finish_their_solves_in_different_rounds(hostcheck_lib, oracle, strategy, outer):
    """chained outer iterations (no host barrier between them): agents with different iteration counts are in different
    outer iterations at the same time; for the trust-region strategy every agent runs its epilogue (step norm, re-roll
    of a too-long step, accept / revert) in the round its own solve ends"""
    N, T = 40, 30
    rng = np.random.default_rng(0)
    x0 = oracle.single_track_x0(N) + 0.1 * rng.standard_normal((N, 4))
    U0 = 0.05 * rng.standard_normal((N, T, 2))
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, U0, T, 0.5, strategy, max_outer=outer)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, U0, T, 0.5, strategy=strategy, max_outer=outer)
    _same(e, o)
    assert len(set(o["iters"].tolist())) > 3


@pytest.mark.parametrize("N,T", [(2, 10), (4, 10), (10, 10), (3, 25)])
def test_centralized_dense_literal_stacked_solve(hostcheck_lib, oracle, N, T):
    """mas_hip_run_centralized_dense: ONE iLQR on the stacked OCP with finite differences of the SUMMED cost
    (centralized.hpp:18-38, multi_agent_problem.hpp:52-127) -- the reference's literal reading, dense cross-agent
    Hessians and all.  Bit-equal to the oracle's dense restatement: per-agent costs, trajectories, iterations, total."""
    x0 = oracle.single_track_x0(N)
    e = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "centralized_dense")
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="centralized_dense")
    _same(e, o)
    b = run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "centralized")
    assert not np.array_equal(b["cost"], e["cost"])                      # the block-diagonal reading is a different number ...
    assert abs(b["total_cost"] - e["total_cost"]) <= 2e-2 * e["total_cost"]   # ... close to it


def test_centralized_dense_other_models_and_limits(hostcheck_lib, oracle):
    x0 = oracle.lqr_x0(3)                                                 # stacked OCP: FD even where the agents have analytic hooks
    e = run_batch(hostcheck_lib, oracle.MODEL_LQR_IDENTITY, x0, None, 12, 0.1, "centralized_dense")
    o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, x0, None, 12, 0.1, strategy="centralized_dense")
    _same(e, o)
    from multi_agent_solver_amd._capi import MasHipError
    with pytest.raises(MasHipError):                                      # 65 agents: beyond the literal solve's limit, loudly
        run_batch(hostcheck_lib, oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(65), None, 5, 0.5, "centralized_dense")


def test_dependency_levels_of_the_vehicle_models(hostcheck_lib):
    """mas_kernel_levels.hpp rolls an open-loop trajectory out level by level (all time steps of a level side by side, then serial
    additions); that equals integrate_horizon bit for bit exactly when the levels the models declare (M::state_level) are true --
    held here against the serial rollout on random trajectories, unknown components replaced by garbage."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "hostcheck", "build", "levels_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "levels_check: 0 mismatches" in r.stdout, (r.stdout, r.stderr)


@pytest.mark.parametrize("strategy,okey,outer", [("sequential", "sequential", 3), ("trustregion", "trustregion", 3), ("linesearch", "linesearch", 2),
                                                   ("centralized", "centralized", 1), ("single", "single", 1)])
def test_user_defined_model_through_the_driver(hostcheck_lib, oracle, strategy, okey, outer):
    """A workload of the kind mas_hip_register_model compiles at run time -- FdModel<Self, NX, NU> with the reference's three callbacks, every
    derivative finite-differenced -- instantiated in the host emulation (tests/hostcheck/user_model_check.cpp): the single-track workload
    written again must give the oracle's bits (cost, iterations, x_T, u_0 of every agent).  The GPU tier holds the same struct, compiled by
    hipcc into a plugin, against the built-in model (tests/test_gpu_register_model.py)."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "hostcheck", "build", "user_model_check")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    N, T = 6, 25
    x0 = oracle.single_track_x0(N) + 0.03 * np.arange(N * 4).reshape(N, 4) / (N * 4)
    r = subprocess.run([exe, str(N), str(T), strategy], input=" ".join(float(v).hex() for v in x0.ravel()), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy=okey, max_outer=outer)
    lines = [l.split() for l in r.stdout.splitlines() if l.startswith("agent ")]
    assert len(lines) == N
    for a, w in enumerate(lines):
        assert float.fromhex(w[3]) == o["cost"][a] and int(w[5]) == int(o["iters"][a]), (a, w)
        assert [float.fromhex(v) for v in w[7:11]] == o["X"][a, T].tolist()
        assert [float.fromhex(v) for v in w[12:14]] == o["U"][a, 0].tolist()
