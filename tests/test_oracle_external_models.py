"""The oracle's external models (oracle.register_model): a user-defined workload of the product -- the struct mas_hip_register_model takes --
compiled for the CPU and solved by the C oracle, so that the GPU tier can hold a workload NEITHER side has built in to every bit
(tests/test_gpu_register_model.py).  Pinned here without a GPU:
  1. the single-track workload written as such a struct gives the bits of the oracle's own model 1 (examples/models/single_track_model.hpp:23-44,
     examples/multi_agent_single_track.cpp:46-64) under every strategy;
  2. a workload with no built-in twin agrees with the SECOND restatement of the reference's solve (oracle/ilqr_numpy.py, Python callbacks)
     to finite-difference noise, as tests/test_oracle_numpy_crosscheck.py does for the built-in ones."""
import numpy as np
import pytest

from test_gpu_register_model import DRAG_CART, SINGLE_TRACK_PARAMS, USER_SINGLE_TRACK


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 3), ("trustregion", 3), ("linesearch", 2), ("centralized", 1),
                                            ("centralized_dense", 1), ("centralized_dense_structured", 1)])
def test_single_track_as_an_external_model_is_the_built_in_one(oracle, strategy, outer):
    mid = oracle.register_model(USER_SINGLE_TRACK, "UserSingleTrack")
    assert mid >= 100 and oracle.model_dims(mid)[:2] == (4, 2)
    N, T = (4, 10) if strategy.startswith("centralized_dense") else (9, 25)
    x0 = oracle.single_track_x0(N) + 0.02 * np.random.default_rng(2).standard_normal((N, 4))
    a = oracle.solve_batch(mid, x0, None, T, 0.5, strategy=strategy, max_outer=outer, params=np.array(SINGLE_TRACK_PARAMS))
    b = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy=strategy, max_outer=outer)
    for key in ("cost", "X", "U", "iters"):
        assert np.array_equal(a[key], b[key]), key
    assert a["total_cost"] == b["total_cost"]


def test_an_external_model_without_a_twin_against_the_numpy_restatement(oracle):
    from oracle import ilqr_numpy as inp
    drag, target, T, dt = 0.3, 2.0, 20, 0.1
    x0 = np.array([[0.0, 0.0], [0.5, -0.2], [-1.0, 1.0]])
    c = oracle.solve_batch(oracle.register_model(DRAG_CART, "DragCart"), x0, None, T, dt, strategy="single", params=np.array([drag, target] + [0.0] * 14))
    for a in range(x0.shape[0]):
        p = inp.Problem(lambda x, u: np.array([x[1], u[0] - drag * x[1] * abs(x[1])]),
                        lambda x, u, t: (x[0] - target) ** 2 + 0.1 * x[1] * x[1] + 0.01 * u[0] * u[0],
                        lambda x: 10.0 * (x[0] - target) ** 2 + x[1] * x[1], x0[a], T, dt, 1)
        r = inp.solve(p, np.zeros((1, T)))
        assert abs(c["cost"][a] - r["cost"]) <= 5e-3 * abs(r["cost"])
        assert abs(int(c["iters"][a]) - r["iterations"]) <= 1
        assert np.abs(c["X"][a].T - r["X"]).max() <= 5e-2


# the lane-following workload (examples/single_track_ocp.cpp:14-116) as a user struct WITH the derivative hooks its example wires: analytic A, B,
# l_x, l_u, l_xx, l_uu (models/single_track_model.hpp:49-82, single_track_ocp.cpp:62-94); the cross term and the terminal derivatives stay FD
USER_LANE = r"""
struct UserLane : FdModel<UserLane, 4, 2> {
  static constexpr bool kAnalyticA = true, kAnalyticB = true, kAnalyticLx = true, kAnalyticLu = true, kAnalyticLxx = true, kAnalyticLuu = true;
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = -0.7; lb[1] = -1.0; ub[0] = 0.7; ub[1] = 1.0; }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    const double psi = x[2], v = x[3], L = P.p[6];
    d[0] = v * mas_dt_cos(psi); d[1] = v * mas_dt_sin(psi); d[2] = (v * mas_dt_tan(u[0])) / L; d[3] = u[1];
  }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {
    const double lane_error = s[1], speed_error = s[3] - P.p[0];
    return P.p[1] * (lane_error * lane_error) + P.p[2] * (speed_error * speed_error) + P.p[3] * (c[0] * c[0]) + P.p[4] * (c[1] * c[1]);
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
  MAS_HD static void A(const double* x, const double* u, double* M, const ModelParams& P) {
    const double psi = x[2], v = x[3], L = P.p[6];
    for (int i = 0; i < 16; ++i) M[i] = 0.0;
    M[2 * 4 + 0] = -v * mas_dt_sin(psi); M[3 * 4 + 0] = mas_dt_cos(psi);
    M[2 * 4 + 1] = v * mas_dt_cos(psi);  M[3 * 4 + 1] = mas_dt_sin(psi);
    M[3 * 4 + 2] = mas_dt_tan(u[0]) / L;
  }
  MAS_HD static void B(const double* x, const double* u, double* M, const ModelParams& P) {
    for (int i = 0; i < 8; ++i) M[i] = 0.0;
    M[0 * 4 + 2] = x[3] / (P.p[6] * mas_dt_cos(u[0]) * mas_dt_cos(u[0]));
    M[1 * 4 + 3] = 1.0;
  }
  MAS_HD static void lx(const double* x, const double*, int, double* g, const ModelParams& P) { g[0] = 0.0; g[1] = 2.0 * P.p[1] * x[1]; g[2] = 0.0; g[3] = 2.0 * P.p[2] * (x[3] - P.p[0]); }
  MAS_HD static void lu(const double*, const double* u, int, double* g, const ModelParams& P) { g[0] = 2.0 * P.p[3] * u[0]; g[1] = 2.0 * P.p[4] * u[1]; }
  MAS_HD static void lxx(const double*, const double*, int, double* H, const ModelParams& P) { for (int i = 0; i < 16; ++i) H[i] = 0.0; H[1 * 4 + 1] = 2.0 * P.p[1]; H[3 * 4 + 3] = 2.0 * P.p[2]; }
  MAS_HD static void luu(const double*, const double*, int, double* H, const ModelParams& P) { H[0] = 2.0 * P.p[3]; H[1] = 0.0; H[2] = 0.0; H[3] = 2.0 * P.p[4]; }
};
"""


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 2), ("trustregion", 3)])
def test_an_external_model_with_analytic_hooks_is_the_built_in_lane_following_model(oracle, strategy, outer):
    """The derivative hooks of an external model (kAnalytic* of the struct -> mas_oracle_external_set_hooks) take the place of the finite
    differences exactly where the built-in lane-following model has them (ocp.hpp:117-135: FD only for hooks left empty)."""
    mid = oracle.register_model(USER_LANE, "UserLane")
    N, T = 7, 30
    rng = np.random.default_rng(4)
    x0 = np.column_stack([np.zeros(N), rng.uniform(-1.5, 1.5, N), rng.uniform(-0.3, 0.3, N), rng.uniform(2.0, 8.0, N)])
    params = oracle.default_params(oracle.MODEL_SINGLE_TRACK_LANE, T)
    b = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LANE, x0, None, T, 0.1, strategy=strategy, max_outer=outer)
    a = oracle.solve_batch(mid, x0, None, T, 0.1, strategy=strategy, max_outer=outer, params=params)
    for key in ("cost", "X", "U", "iters"):
        assert np.array_equal(a[key], b[key]), key


# function constraints of a user workload (ocp.hpp:61-68): NEQ / NIN + equality() / inequality() of the struct; the constrained test workload
# (model 6: single track, steering tied to the acceleration, speed limits) written as a user struct
USER_LIMITED = r"""
struct UserLimited : FdModel<UserLimited, 4, 2> {
  static constexpr int NEQ = 1, NIN = 2;
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = lb[1] = -0.5; ub[0] = ub[1] = 0.5; }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    const double psi = x[2], v = x[3], L = P.p[6];
    d[0] = v * mas_dt_cos(psi); d[1] = v * mas_dt_sin(psi); d[2] = (v * mas_dt_tan(u[0])) / L; d[3] = u[1];
  }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {
    const double dist = fabs(sqrt(s[0] * s[0] + s[1] * s[1]) - P.p[0]);
    const double se = s[3] - P.p[1];
    return P.p[2] * dist * dist + P.p[3] * se * se + P.p[4] * c[0] * c[0] + P.p[5] * c[1] * c[1];
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
  MAS_HD static void equality(const double*, const double* u, double* out, const ModelParams& P) { out[0] = u[0] - P.p[9] * u[1]; }
  MAS_HD static void inequality(const double* x, const double*, double* out, const ModelParams& P) { out[0] = x[3] - P.p[8]; out[1] = -x[3]; }
};
"""


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 2)])
def test_an_external_model_with_function_constraints_is_the_built_in_constrained_model(oracle, strategy, outer):
    mid = oracle.register_model(USER_LIMITED, "UserLimited")
    N, T = 6, 20
    x0 = oracle.single_track_x0(N)
    params = oracle.default_params(oracle.MODEL_SINGLE_TRACK_LIMITED, T)
    b = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, T, 0.5, strategy=strategy, max_outer=outer)
    a = oracle.solve_batch(mid, x0, None, T, 0.5, strategy=strategy, max_outer=outer, params=params)
    for key in ("cost", "X", "U", "iters"):
        assert np.array_equal(a[key], b[key]), key
