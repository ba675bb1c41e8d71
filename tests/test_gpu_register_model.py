"""User-defined workloads (mas_hip_register_model): the reference takes a workload as callbacks (types.hpp:21-50) and finite-differences
every derivative hook an example leaves empty (ocp.hpp:117-135); here a workload is a struct of static functions, and a NEW one is
compiled against the kernel templates at run time -- no rebuild of libmas_hip.so.

1. The single-track workload written AGAIN as a user model (plain FdModel: no hoisted tan(delta), no four-lane RK4, no dependency
   levels, no skipped zero terminal cost) must give the bits of the built-in model 1 and of the oracle, for every strategy: same
   arithmetic per evaluation, other kernel forms.
2. A workload the library has never seen (a cart with quadratic drag: two states, one control) against the SECOND restatement of the
   reference's solve, oracle/ilqr_numpy.py, which takes Python callbacks like the reference takes std::function: finite-difference-noise
   tolerance, as in tests/test_oracle_numpy_crosscheck.py -- and against the C oracle solving the SAME struct compiled for the CPU
   (oracle.register_model): every bit.  A second unseen workload (trigonometry, input bounds, a time-dependent stage cost) the same way
   under every strategy.
3. A struct that does not compile: MAS_HIP_ERR_INVALID with the compiler's message."""
import math

import numpy as np
import pytest

from conftest import run_batch

pytestmark = pytest.mark.gpu

USER_SINGLE_TRACK = r"""
struct UserSingleTrack : FdModel<UserSingleTrack, 4, 2> {
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = lb[1] = -0.5; ub[0] = ub[1] = 0.5; }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {     // examples/models/single_track_model.hpp:23-44
    const double psi = x[2], v = x[3], L = P.p[6];
    d[0] = v * mas_dt_cos(psi);
    d[1] = v * mas_dt_sin(psi);
    d[2] = (v * mas_dt_tan(u[0])) / L;
    d[3] = u[1];
  }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {        // examples/multi_agent_single_track.cpp:46-64
    const double dist = fabs(sqrt(s[0] * s[0] + s[1] * s[1]) - P.p[0]);
    const double se = s[3] - P.p[1];
    return P.p[2] * dist * dist + P.p[3] * se * se + P.p[4] * c[0] * c[0] + P.p[5] * c[1] * c[1];
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
};
"""

DRAG_CART = r"""
struct DragCart : FdModel<DragCart, 2, 1> {
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    d[0] = x[1];
    d[1] = u[0] - P.p[0] * x[1] * fabs(x[1]);
  }
  MAS_HD static double stage_cost(const double* x, const double* u, int, const ModelParams& P) {
    const double e = x[0] - P.p[1];
    return e * e + 0.1 * x[1] * x[1] + 0.01 * u[0] * u[0];
  }
  MAS_HD static double terminal_cost(const double* x, const ModelParams& P) { const double e = x[0] - P.p[1]; return 10.0 * e * e + x[1] * x[1]; }
};
"""

WHEELED_ARM = r"""
struct WheeledArm : FdModel<WheeledArm, 5, 2> {                 // a unicycle carrying a joint: position, heading, speed, joint angle
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = -1.0; ub[0] = 1.0; lb[1] = -0.7; ub[1] = 0.7; }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    d[0] = x[3] * mas_dt_cos(x[2]);
    d[1] = x[3] * mas_dt_sin(x[2]);
    d[2] = x[3] * u[1];
    d[3] = u[0] - P.p[3] * x[3];
    d[4] = u[1] * mas_dt_cos(x[4]) - P.p[4] * mas_dt_sin(x[4]);
  }
  MAS_HD static double stage_cost(const double* x, const double* u, int t, const ModelParams& P) {
    const double ex = x[0] - P.p[0], ey = x[1] - P.p[1], ej = x[4] - P.p[2];
    const double w = t < 12 ? 1.0 : 2.0;
    return w * (ex * ex + ey * ey) + 0.5 * ej * ej + 0.05 * u[0] * u[0] + 0.05 * u[1] * u[1];
  }
  MAS_HD static double terminal_cost(const double* x, const ModelParams& P) {
    const double ex = x[0] - P.p[0], ey = x[1] - P.p[1];
    return P.p[5] * (ex * ex + ey * ey) + x[3] * x[3];
  }
};
"""

SIX_BY_THREE = r"""
struct SixByThree : FdModel<SixByThree, 6, 3> {                 // two coupled, damped double integrators with a steering-like third axis
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    d[0] = x[3] + 0.1 * mas_dt_sin(x[2]);
    d[1] = x[4];
    d[2] = x[5] - 0.2 * x[2];
    d[3] = u[0] - P.p[0] * x[3] * fabs(x[3]) + 0.3 * x[1];
    d[4] = u[1] - 0.4 * x[4] + 0.2 * x[0] * mas_dt_cos(x[2]);
    d[5] = u[2] - 0.5 * x[5];
  }
  MAS_HD static double stage_cost(const double* x, const double* u, int, const ModelParams& P) {
    const double e0 = x[0] - P.p[1], e1 = x[1] - P.p[2];
    return e0 * e0 + 0.8 * e1 * e1 + 0.3 * x[2] * x[2] + 0.1 * (x[3] * x[3] + x[4] * x[4] + x[5] * x[5]) + 0.05 * (u[0] * u[0] + u[1] * u[1]) + 0.02 * u[2] * u[2];
  }
  MAS_HD static double terminal_cost(const double* x, const ModelParams& P) {
    const double e0 = x[0] - P.p[1], e1 = x[1] - P.p[2];
    return 5.0 * (e0 * e0 + e1 * e1) + x[3] * x[3] + x[4] * x[4];
  }
};
"""

SINGLE_TRACK_PARAMS = [20.0, 5.0, 1.0, 1.0, 0.001, 0.001, 2.5] + [0.0] * 9


@pytest.fixture(scope="module")
def gpu_lib(product_lib):
    assert product_lib.mas_hip_device_count() > 0
    # the module's six workloads, registered from six threads: the library runs the compiler outside its registry lock, so they compile side by
    # side (one compile time instead of six) and the tests below find their plugins in the cache
    from concurrent.futures import ThreadPoolExecutor
    from multi_agent_solver_amd.batch import register_model
    from test_oracle_external_models import USER_LANE, USER_LIMITED
    structs = [(USER_SINGLE_TRACK, "UserSingleTrack"), (DRAG_CART, "DragCart"), (WHEELED_ARM, "WheeledArm"), (SIX_BY_THREE, "SixByThree"),
               (USER_LANE, "UserLane"), (USER_LIMITED, "UserLimited")]
    with ThreadPoolExecutor(len(structs)) as pool:
        ids = list(pool.map(lambda sn: register_model(sn[0], sn[1], lib=product_lib), structs))
    assert len(set(ids)) == len(ids) and min(ids) >= 100
    return product_lib


@pytest.fixture(scope="module")
def user_single_track(gpu_lib):
    from multi_agent_solver_amd.batch import register_model
    return register_model(USER_SINGLE_TRACK, "UserSingleTrack", lib=gpu_lib)          # about a minute of hipcc


@pytest.mark.parametrize("strategy,outer", [("sequential", 3), ("trustregion", 3), ("linesearch", 2), ("centralized", 1), ("centralized_dense", 1)])
def test_single_track_rewritten_as_a_user_model(gpu_lib, oracle, user_single_track, strategy, outer):
    assert user_single_track >= 100
    N, T = (6, 12) if strategy == "centralized_dense" else (70, 30)
    rng = np.random.default_rng(3)
    x0 = oracle.single_track_x0(N) + 0.05 * rng.standard_normal((N, 4))
    mine = run_batch(gpu_lib, user_single_track, x0, None, T, 0.5, strategy, max_outer=outer, params=np.array(SINGLE_TRACK_PARAMS))
    builtin = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy, max_outer=outer)
    for key in ("cost", "X", "U", "iters", "status"):
        assert np.array_equal(mine[key], builtin[key]), key
    assert mine["total_cost"] == builtin["total_cost"] and mine["stats"]["reg_events"] == builtin["stats"]["reg_events"]
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy=strategy, max_outer=outer)
    assert np.array_equal(mine["cost"], o["cost"]) and np.array_equal(mine["X"], o["X"]) and np.array_equal(mine["U"], o["U"])
    assert mine["iters"].tolist() == o["iters"].tolist()


def test_a_workload_the_library_has_never_seen(gpu_lib, oracle):
    from multi_agent_solver_amd.batch import register_model
    from oracle import ilqr_numpy as inp
    mid = register_model(DRAG_CART, "DragCart", lib=gpu_lib)
    drag, target, T, dt = 0.3, 2.0, 20, 0.1
    params = np.array([drag, target] + [0.0] * 14)
    x0 = np.array([[0.0, 0.0], [0.5, -0.2], [-1.0, 1.0]])
    e = run_batch(gpu_lib, mid, x0, None, T, dt, "single", params=params, no_input_bounds=True)
    # the same struct compiled for the CPU and solved by the C oracle: every bit
    c = oracle.solve_batch(oracle.register_model(DRAG_CART, "DragCart"), x0, None, T, dt, strategy="single", params=params)
    for key in ("cost", "X", "U"):
        assert np.array_equal(e[key], c[key]), key
    assert e["iters"].tolist() == c["iters"].tolist()
    for a in range(x0.shape[0]):
        p = inp.Problem(lambda x, u: np.array([x[1], u[0] - drag * x[1] * abs(x[1])]),
                        lambda x, u, t: (x[0] - target) ** 2 + 0.1 * x[1] * x[1] + 0.01 * u[0] * u[0],
                        lambda x: 10.0 * (x[0] - target) ** 2 + x[1] * x[1], x0[a], T, dt, 1)
        r = inp.solve(p, np.zeros((1, T)))
        assert abs(e["cost"][a] - r["cost"]) <= 5e-3 * abs(r["cost"]), (a, e["cost"][a], r["cost"])         # two FD solves: noise-level agreement
        assert abs(int(e["iters"][a]) - r["iterations"]) <= 1
        assert np.abs(e["X"][a].T - r["X"]).max() <= 5e-2


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 3), ("trustregion", 3), ("linesearch", 2), ("centralized", 1), ("centralized_dense", 1)])
def test_an_unseen_workload_with_trigonometry_and_bounds_under_every_strategy(gpu_lib, oracle, strategy, outer):
    """Neither side has this model built in: the product compiles the struct with hipcc against the kernel templates, the oracle compiles the
    same text with g++ against the models header in its host form -- bit-equal results, as for the built-in workloads."""
    from multi_agent_solver_amd.batch import register_model
    mid, oid = register_model(WHEELED_ARM, "WheeledArm", lib=gpu_lib), oracle.register_model(WHEELED_ARM, "WheeledArm")
    N, T = (5, 10) if strategy == "centralized_dense" else (40, 24)
    rng = np.random.default_rng(17)
    x0 = np.column_stack([rng.uniform(-2, 2, N), rng.uniform(-2, 2, N), rng.uniform(-math.pi, math.pi, N), rng.uniform(0, 1, N), rng.uniform(-0.5, 0.5, N)])
    params = np.array([1.5, -0.5, 0.8, 0.05, 0.02, 4.0] + [0.0] * 10)
    e = run_batch(gpu_lib, mid, x0, None, T, 0.2, strategy, max_outer=outer, params=params)
    c = oracle.solve_batch(oid, x0, None, T, 0.2, strategy=strategy, max_outer=outer, params=params)
    for key in ("cost", "X", "U"):
        assert np.array_equal(e[key], c[key]), key
    assert e["iters"].tolist() == c["iters"].tolist()
    # (the block-diagonal centralised merit is summed agent-major on the GPU and step-major in the oracle: DESIGN.md section 3)
    assert abs(e["total_cost"] - c["total_cost"]) <= (1e-12 * abs(c["total_cost"]) if strategy == "centralized" else 0.0)


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 2), ("trustregion", 3)])
def test_a_user_model_with_analytic_derivative_hooks(gpu_lib, oracle, strategy, outer):
    """The reference's optional derivative callbacks (types.hpp:30-50; ocp.hpp:117-135 finite-differences only the hooks left empty): a user
    struct declares kAnalyticA ... and provides A / B / lx / lu / lxx / luu.  The lane-following workload written that way gives the bits of the
    built-in model 4 and of the oracle (which takes the same hooks from the same struct: oracle.register_model)."""
    from multi_agent_solver_amd.batch import register_model
    from test_oracle_external_models import USER_LANE
    mid, oid = register_model(USER_LANE, "UserLane", lib=gpu_lib), oracle.register_model(USER_LANE, "UserLane")
    N, T = 90, 40
    rng = np.random.default_rng(4)
    x0 = np.column_stack([np.zeros(N), rng.uniform(-1.5, 1.5, N), rng.uniform(-0.3, 0.3, N), rng.uniform(2.0, 8.0, N)])
    params = oracle.default_params(oracle.MODEL_SINGLE_TRACK_LANE, T)
    mine = run_batch(gpu_lib, mid, x0, None, T, 0.1, strategy, max_outer=outer, params=params)
    builtin = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK_LANE, x0, None, T, 0.1, strategy, max_outer=outer)
    c = oracle.solve_batch(oid, x0, None, T, 0.1, strategy=strategy, max_outer=outer, params=params)
    for key in ("cost", "X", "U"):
        assert np.array_equal(mine[key], builtin[key]) and np.array_equal(mine[key], c[key]), key
    assert mine["iters"].tolist() == builtin["iters"].tolist() == c["iters"].tolist()
    assert mine["stats"]["reg_events"] == builtin["stats"]["reg_events"]


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 2)])
def test_a_user_model_with_function_constraints(gpu_lib, oracle, strategy, outer):
    """h(x, u) = 0 and g(x, u) <= 0 of a user workload (ocp.hpp:61-68: NEQ / NIN + equality() / inequality() of the struct) through the
    augmented-Lagrangian branch (ilqr.hpp:236-260): the constrained test workload written as a user struct gives the bits of the built-in
    model 6 and of the oracle (same struct, oracle.register_model -> mas_oracle_external_set_constraints)."""
    from multi_agent_solver_amd.batch import register_model
    from test_oracle_external_models import USER_LIMITED
    mid, oid = register_model(USER_LIMITED, "UserLimited", lib=gpu_lib), oracle.register_model(USER_LIMITED, "UserLimited")
    N, T = 70, 30
    x0 = oracle.single_track_x0(N)
    params = oracle.default_params(oracle.MODEL_SINGLE_TRACK_LIMITED, T)
    mine = run_batch(gpu_lib, mid, x0, None, T, 0.5, strategy, max_outer=outer, params=params)
    builtin = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, T, 0.5, strategy, max_outer=outer)
    c = oracle.solve_batch(oid, x0, None, T, 0.5, strategy=strategy, max_outer=outer, params=params)
    for key in ("cost", "X", "U"):
        assert np.array_equal(mine[key], builtin[key]) and np.array_equal(mine[key], c[key]), key
    assert mine["iters"].tolist() == builtin["iters"].tolist() == c["iters"].tolist()


@pytest.mark.parametrize("strategy,outer", [("sequential", 2), ("centralized", 1), ("trustregion", 2)])
def test_a_six_state_three_control_workload(gpu_lib, oracle, strategy, outer):
    """Tiles of 126 doubles: the centralised backward pass has LDS for 8 wavefronts per workgroup, not the 16 its widest form takes -- forms
    that do not fit are not instantiated (found by tools/fuzz_models.py: the registration failed with the compiler's 'local memory exceeds limit')."""
    from multi_agent_solver_amd.batch import register_model
    mid, oid = register_model(SIX_BY_THREE, "SixByThree", lib=gpu_lib), oracle.register_model(SIX_BY_THREE, "SixByThree")
    N, T = 150, 30
    rng = np.random.default_rng(31)
    x0 = rng.uniform(-1.0, 1.0, (N, 6))
    params = np.array([0.2, 1.0, -0.5] + [0.0] * 13)
    e = run_batch(gpu_lib, mid, x0, None, T, 0.1, strategy, max_outer=outer, params=params, no_input_bounds=True)
    c = oracle.solve_batch(oid, x0, None, T, 0.1, strategy=strategy, max_outer=outer, params=params)
    for key in ("cost", "X", "U"):
        assert np.array_equal(e[key], c[key]), key
    assert e["iters"].tolist() == c["iters"].tolist()
    if strategy == "centralized":       # every width of the shared ladder the LDS holds, and a request beyond it (clamped)
        for wpb in (16, 8, 2, 1):
            other = run_batch(gpu_lib, mid, x0, None, T, 0.1, strategy, max_outer=outer, params=params, no_input_bounds=True, options={"central_wpb": wpb})
            assert all(np.array_equal(e[key], other[key]) for key in ("cost", "X", "U", "iters"))


def test_the_second_registration_loads_the_cached_plugin(gpu_lib, user_single_track):
    import time
    from multi_agent_solver_amd.batch import register_model
    t0 = time.time()
    again = register_model(USER_SINGLE_TRACK, "UserSingleTrack", lib=gpu_lib)
    assert again != user_single_track and time.time() - t0 < 5.0          # (a new id for the same plugin; the compiler did not run)


def test_a_model_that_does_not_compile_is_refused_with_the_compilers_words(gpu_lib):
    from multi_agent_solver_amd._capi import MasHipError, ERR_INVALID
    from multi_agent_solver_amd.batch import register_model
    with pytest.raises(MasHipError) as err:
        register_model("struct Broken : FdModel<Broken, 2, 1> { this is not C++ };", "Broken", lib=gpu_lib)
    assert err.value.code == ERR_INVALID and "error" in str(err.value)
    with pytest.raises(MasHipError):
        register_model("struct X {};", "not an identifier", lib=gpu_lib)
