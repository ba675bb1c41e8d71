"""The literal centralised solve (centralized.hpp:18-38: ONE iLQR on the stacked OCP, every cost derivative a finite difference of
the SUMMED cost) costs 4 (N nx)^2 evaluations of the N-agent cost per time step: the plain restatement (`centralized_dense`,
mas_ilqr_solve on the stacked OCP through the generic FD hooks) needs 17 minutes for 64 agents at H = 100.
`centralized_dense_structured` (oracle/mas_oracle_dense_impl.h) computes the same numbers with exact re-use only -- the stored prefix
of the summed cost's own chain of additions, the exact zeros of the stacked A and B -- so that config 2's 256 agents become
computable.  Here: it IS the same solve (np.array_equal on everything, wherever both can run), and it reproduces the totals the
plain restatement left in profiles/r3_c2_readings_curve.json at 32 and 64 agents."""
import json
import math
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ring(n):
    th = 2 * np.pi * np.arange(n) / n
    return np.stack([20 * np.cos(th), 20 * np.sin(th), 1.57 + th, np.full(n, 4.0)], 1)


CASES = [
    (1, 2, 10, 0.5), (1, 3, 20, 0.5), (1, 5, 30, 0.5), (1, 8, 100, 0.5),      # single-track (the C2 workload), up to its horizon
    (3, 2, 12, 0.05),                                                            # pendulum: non-trivial terminal cost
    (2, 3, 10, 0.1),                                                             # identity LQR: analytic hooks are NOT inherited by the stacked OCP
    (5, 2, 8, 0.1),                                                              # bicycle6
]


@pytest.mark.parametrize("model,n,T,dt", CASES)
def test_structured_solve_is_the_literal_solve(oracle, model, n, T, dt):
    oracle.set_trig_mode("det")
    nx, nu, _ = oracle.model_dims(model)
    if model == 1:
        x0 = ring(n)
    elif model == 3:
        x0 = np.array([[math.pi - 0.05, 0.0], [math.pi - 0.1, 0.1]])
    else:
        x0 = np.tile(np.eye(nx)[0], (n, 1)) + 0.1 * np.arange(n)[:, None] + (np.array([0, 0, 0, 2.0, 0, 0])[None, :nx] if model == 5 else 0)
    a = oracle.solve_batch(model, x0, None, T, dt, strategy="centralized_dense")
    b = oracle.solve_batch(model, x0, None, T, dt, strategy="centralized_dense_structured")
    for key in ("X", "U", "cost", "iters"):
        assert np.array_equal(a[key], b[key]), key
    assert a["total_cost"] == b["total_cost"]
    assert a["stats"] == b["stats"]            # iterations, rungs of the ladder, line-search trials, accepted step sizes


@pytest.mark.parametrize("n", [32, 64])
def test_structured_solve_reproduces_the_committed_literal_totals(oracle, n):
    oracle.set_trig_mode("det")
    rows = {r["agents"]: r for r in json.load(open(os.path.join(ROOT, "profiles", "r3_c2_readings_curve.json")))["curve"]}
    b = oracle.solve_batch(1, ring(n), None, 100, 0.5, strategy="centralized_dense_structured")
    assert b["total_cost"] == rows[n]["literal_total_cost"]           # 99 s / 17 min of the plain restatement; here 0.5 s / 3.4 s
    assert int(b["iters"][0]) == rows[n]["iterations_literal"]
