"""Generate tests/golden/oracle_vectors.json from the CPU oracle.

These vectors are ORACLE-generated, not reference-generated: the reference cannot be built in
this image (Eigen 3.4 absent), so they are a regression pin of the oracle itself (and a second
check for the HIP path on a box where only the fixtures travel), not an independent pin of
iLQR::solve.  Run from the repo root:  python tests/golden/make_oracle_vectors.py
"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, model, N, T, dt, strategy, max_outer, max_iterations, tolerance
    ("single_track_N3_T10_sequential", O.MODEL_SINGLE_TRACK, 3, 10, 0.5, "sequential", 10, 100, 1e-5),
    ("single_track_N3_T100_sequential", O.MODEL_SINGLE_TRACK, 3, 100, 0.5, "sequential", 10, 100, 1e-5),
    ("single_track_N8_T100_single", O.MODEL_SINGLE_TRACK, 8, 100, 0.5, "single", 1, 100, 1e-5),
    ("single_track_N8_T100_trustregion", O.MODEL_SINGLE_TRACK, 8, 100, 0.5, "trustregion", 10, 100, 1e-5),
    ("single_track_N8_T100_linesearch", O.MODEL_SINGLE_TRACK, 8, 100, 0.5, "linesearch", 10, 100, 1e-5),
    ("single_track_N8_T100_centralized", O.MODEL_SINGLE_TRACK, 8, 100, 0.5, "centralized", 1, 100, 1e-5),
    ("single_track_N2_T10_centralized_dense", O.MODEL_SINGLE_TRACK, 2, 10, 0.5, "centralized_dense", 1, 100, 1e-5),
    ("single_track_N4_T10_centralized_dense", O.MODEL_SINGLE_TRACK, 4, 10, 0.5, "centralized_dense", 1, 100, 1e-5),
    ("single_track_N10_T10_centralized_dense", O.MODEL_SINGLE_TRACK, 10, 10, 0.5, "centralized_dense", 1, 100, 1e-5),   # README.md:176's shape
    ("lqr_N2_T10_sequential", O.MODEL_LQR_IDENTITY, 2, 10, 0.1, "sequential", 10, 100, 1e-5),
    ("lqr_N2_T50_sequential", O.MODEL_LQR_IDENTITY, 2, 50, 0.1, "sequential", 10, 100, 1e-5),
    ("pendulum_T50_single", O.MODEL_PENDULUM, 1, 50, 0.05, "single", 1, 1000, 1e-4),
    ("pendulum_T60_single", O.MODEL_PENDULUM, 1, 60, 0.05, "single", 1, 1000, 1e-4),
    ("lane_T80_single", O.MODEL_SINGLE_TRACK_LANE, 1, 80, 0.1, "single", 1, 10, 1e-5),
    ("bicycle6_N4_T200_trustregion", O.MODEL_BICYCLE6, 4, 200, 0.25, "trustregion", 10, 100, 1e-5),
]


def inputs(model, N, T, dt):
    if model == O.MODEL_SINGLE_TRACK:
        return O.single_track_x0(N), None
    if model == O.MODEL_LQR_IDENTITY:
        return O.lqr_x0(N), None
    if model == O.MODEL_PENDULUM:
        return O.pendulum_problem(T, dt)
    if model == O.MODEL_SINGLE_TRACK_LANE:
        import numpy as np
        return np.array([[0.0, 1.0, 0.0, 0.0]]), None
    if model == O.MODEL_BICYCLE6:
        return O.bicycle6_x0(N), None
    raise ValueError(model)


def main():
    O.set_trig_mode("det")   # portable bit-reproducible trig (include/mas_dettrig.h): vectors are exact on any host
    out = {"_generator": "tests/golden/make_oracle_vectors.py", "_origin": "CPU oracle (oracle/mas_oracle.c) in 'det' trig mode, NOT the reference", "cases": {}}
    for name, model, N, T, dt, strat, max_outer, max_it, tol in CASES:
        x0, U0 = inputs(model, N, T, dt)
        r = O.solve_batch(model, x0, U0, T, dt, strategy=strat, max_outer=max_outer, max_iterations=max_it, tolerance=tol, threads=1)
        out["cases"][name] = dict(model=model, N=N, T=T, dt=dt, strategy=strat, max_outer=max_outer, max_iterations=max_it, tolerance=tol,
                                  x0=x0.tolist(), cost=[float.hex(c) for c in r["cost"]], cost_decimal=r["cost"].tolist(),
                                  iters=r["iters"].tolist(), total_cost=float.hex(r["total_cost"]),
                                  x_final=[[float.hex(v) for v in r["X"][a, -1]] for a in range(N)],
                                  u_first=[[float.hex(v) for v in r["U"][a, 0]] for a in range(N)],
                                  reg_events=r["stats"]["reg_events"], trials=r["stats"]["trials"], alpha_hist=r["stats"]["alpha_hist"])
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")


if __name__ == "__main__":
    main()
