"""Randomised parity sweep on the GPU: random models, agent counts, horizons, strategies, warm starts, bounds and iteration
limits and kernel forms (fused / five-kernel rounds, one- / four-lane line search), every result compared bit for bit with the CPU oracle (det trig).  Not part of the test suite (run time grows with
the case count); the fixed-seed tests cover the same code paths with chosen cases.
usage: python tools/fuzz_parity.py [cases] [seed]"""
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_solver_amd.batch import BatchSolver  # noqa: E402
from oracle import oracle as O  # noqa: E402

BIG = int(os.environ.get("FUZZ_BIG", "1"))      # agent-count multiplier (FUZZ_BIG=8: batches of thousands, slower oracle)
ONLY = int(os.environ.get("FUZZ_MODEL", "0"))   # 1..6: that model only, with the large agent counts (e.g. 5: bicycle6 across the 16 / 32 / 64-lane backward forms)
ONLY_STRATEGY = os.environ.get("FUZZ_STRATEGY", "")   # e.g. centralized_dense: that strategy only, with larger stacked problems
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
O.set_trig_mode("det")
DT = {1: 0.5, 2: 0.1, 3: 0.05, 4: 0.1, 5: 0.25, 6: 0.5}
bad = 0
t0 = time.time()
for c in range(cases):
    model = ONLY if ONLY else int(rng.choice([1, 1, 1, 2, 3, 4, 5, 6]))
    nx, nu, _ = O.model_dims(model)
    strategy = str(rng.choice(["single", "sequential", "trustregion", "linesearch", "centralized", "centralized_dense"]))
    N = int(rng.integers(1, 700 * BIG)) if (model in (1, 2, 6) or ONLY) else int(rng.integers(1, 80 * BIG))
    T = int(rng.integers(1, 60))
    if model == 3:
        N, T = int(rng.integers(1, 4)), int(rng.integers(5, 40))
    if strategy == "centralized":
        N = min(N, 200 * min(BIG, 5))
    if ONLY_STRATEGY:
        strategy = ONLY_STRATEGY
    if strategy == "centralized_dense":
        N, T = int(rng.integers(1, 7)), int(rng.integers(1, 9))
        if ONLY_STRATEGY:                      # FUZZ_STRATEGY=centralized_dense: stacked problems of up to 240 states, held against the oracle's structured restatement
            N, T = int(rng.integers(1, 41)), int(rng.integers(1, 41))
    if model == 5:
        T = int(rng.integers(1, 120))
    max_outer = int(rng.integers(1, 5))
    max_it = int(rng.choice([1, 3, 100]))
    tol = float(rng.choice([1e-5, 1e-3]))
    if model == 1 or model == 6:
        x0 = O.single_track_x0(N) + 0.2 * rng.standard_normal((N, 4))
    elif model == 5:
        x0 = O.bicycle6_x0(N) + 0.1 * rng.standard_normal((N, 6))
    elif model == 4:
        x0 = np.tile([0.0, 1.0, 0.0, 0.0], (N, 1)) + 0.3 * rng.standard_normal((N, 4))
    elif model == 3:
        x0 = np.tile([math.pi - 0.05, 0.0], (N, 1)) + 0.05 * rng.standard_normal((N, 2))
    else:
        x0 = rng.standard_normal((N, nx))
    U0 = None if rng.random() < 0.4 or strategy.startswith("centralized") else 0.1 * rng.standard_normal((N, T, nu))
    # kernel forms (bit-identical alternatives, mas_hip_set_option): two-kernel fused rounds or the five-kernel round, the four-lane line
    # search never / up to 1600 live agents / always, the LDS-tile backward kernel with 16 lanes per agent only / 32 / 64 lanes for few live agents
    forms = {"fused": int(rng.choice([0, 1, 1, 1])), "forward_quad_max": int(rng.choice([0, 1600, 1600, 1000000])),
             "backward_wide_max": int(rng.choice([0, 4096, 4096])), "backward_wave_max": int(rng.choice([0, 1024, 1024]))}
    with BatchSolver(model, N, T, DT[model]) as s:
        for key, value in forms.items():
            s.set_option(key, value)
        s.set_params({"max_iterations": max_it, "tolerance": tol, "max_ms": math.inf})
        s.set_initial_states(x0); s.set_controls(U0); s.initialize_problem()
        s.run(strategy, max_outer)
        e = s.solution(); st = s.stats()
    o = O.solve_batch(model, x0, U0, T, DT[model], strategy=("centralized_dense_structured" if strategy == "centralized_dense" and N > 6 else strategy),
                      max_outer=max_outer, max_iterations=max_it, tolerance=tol)
    eq = lambda a, b: np.array_equal(a, b, equal_nan=True)      # (pendulum with T = 1: its cost divides by T - 1 -> NaN on both sides)
    ok = (eq(e["cost"], o["cost"]) and eq(e["X"], o["X"]) and eq(e["U"], o["U"]) and
          e["iters"].tolist() == o["iters"].tolist() and st["reg_events"] == o["stats"]["reg_events"])
    if strategy != "centralized":
        ok = ok and (e["total_cost"] == o["total_cost"] or (math.isnan(e["total_cost"]) and math.isnan(o["total_cost"])))
    if (o["status"] >= 0).all():
        ok = ok and (e["status"] & 3).tolist() == o["status"].tolist()
    if not ok:
        bad += 1
        print("MISMATCH", dict(case=c, model=model, strategy=strategy, N=N, T=T, max_outer=max_outer, max_it=max_it, tol=tol, warm=U0 is not None, **forms))
print(f"{cases} cases, seed {seed}: {bad} mismatches, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
