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
