"""Randomised WORKLOADS: structs of the kind mas_hip_register_model takes, generated at random (2-6 states, 1-3 controls; dynamics = a random
stable linear part + products, sin / cos of states, a drag term; a quadratic stage cost with a non-quadratic term, a terminal cost; input bounds
or none), compiled by hipcc against the kernel templates (the product) AND by g++ for the C oracle (oracle.register_model), then solved under
every strategy on random batches -- every result compared bit for bit.  Neither side has any of these models built in.  Not part of the test
suite (a minute of hipcc per model).   usage: python tools/fuzz_models.py [models] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multi_agent_solver_amd.batch import BatchSolver, register_model  # noqa: E402
from oracle import oracle as O  # noqa: E402

n_models = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
MAX_NX, MAX_NU = int(os.environ.get("FUZZ_MAX_NX", "6")), int(os.environ.get("FUZZ_MAX_NU", "3"))     # the library takes 1 .. 8 of either
O.set_trig_mode("det")


def lit(v):
    return repr(float(np.round(v, 6)))


def make_struct(name):
    nx, nu = int(rng.integers(2, MAX_NX + 1)), int(rng.integers(1, MAX_NU + 1))
    lines = []
    for i in range(nx):
        terms = []
        for j in range(nx):                                   # a damped linear part
            if i == j:
                terms.append(f"{lit(-rng.uniform(0.05, 0.6))} * x[{j}]")
            elif rng.random() < 0.4:
                terms.append(f"{lit(rng.uniform(-0.8, 0.8))} * x[{j}]")
        for k in range(nu):
            if rng.random() < 0.6 or i == nx - 1 - k % nx:
                terms.append(f"{lit(rng.uniform(-1.0, 1.0))} * u[{k}]")
        kind = rng.integers(0, 5)
        j, k2 = int(rng.integers(0, nx)), int(rng.integers(0, nx))
        if kind == 0:
            terms.append(f"{lit(rng.uniform(-0.5, 0.5))} * mas_dt_sin(x[{j}])")
        elif kind == 1:
            terms.append(f"{lit(rng.uniform(-0.5, 0.5))} * x[{j}] * mas_dt_cos(x[{k2}])")
        elif kind == 2:
            terms.append(f"{lit(rng.uniform(-0.2, 0.2))} * x[{j}] * x[{k2}]")
        elif kind == 3:
            terms.append(f"{lit(-rng.uniform(0.0, 0.3))} * x[{j}] * fabs(x[{j}])")
        else:
            terms.append(f"P.p[{int(rng.integers(0, 4))}] * u[{int(rng.integers(0, nu))}] * mas_dt_cos(x[{j}])")
        lines.append(f"    d[{i}] = " + " + ".join(terms) + ";")
    q = [lit(rng.uniform(0.1, 2.0)) for _ in range(nx)]
    r = [lit(rng.uniform(0.01, 0.5)) for _ in range(nu)]
    stage = " + ".join([f"{q[i]} * (x[{i}] - P.p[{4 + i}]) * (x[{i}] - P.p[{4 + i}])" for i in range(nx)] + [f"{r[k]} * u[{k}] * u[{k}]" for k in range(nu)])
    extra = [f"{lit(rng.uniform(0.05, 0.5))} * (1.0 - mas_dt_cos(x[{int(rng.integers(0, nx))}]))", f"{lit(rng.uniform(0.01, 0.2))} * x[0] * x[0] * x[{nx - 1}] * x[{nx - 1}]",
             "(t < 5 ? 0.5 : 0.0) * x[0] * x[0]"][int(rng.integers(0, 3))]
    term = " + ".join(f"{lit(rng.uniform(0.5, 5.0))} * (x[{i}] - P.p[{4 + i}]) * (x[{i}] - P.p[{4 + i}])" for i in range(nx))
    bounded = rng.random() < 0.5
    bounds = ""
    if bounded:
        b = [lit(rng.uniform(0.3, 1.5)) for _ in range(nu)]
        bounds = ("  static constexpr bool kDefaultBounds = true;\n  MAS_HD static void default_bounds(double* lb, double* ub) { "
                  + " ".join(f"lb[{k}] = -{b[k]}; ub[{k}] = {b[k]};" for k in range(nu)) + " }\n")
    zero_term = rng.random() < 0.25
    src = (f"struct {name} : FdModel<{name}, {nx}, {nu}> {{\n{bounds}"
           f"  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {{\n" + "\n".join(lines) + "\n  }\n"
           f"  MAS_HD static double stage_cost(const double* x, const double* u, int t, const ModelParams& P) {{ return {stage} + {extra}; }}\n"
           f"  MAS_HD static double terminal_cost(const double* x, const ModelParams& P) {{ return {'0.0' if zero_term else term}; }}\n}};\n")
    return src, nx, nu, bounded


bad = 0
t0 = time.time()
for m in range(n_models):
    name = f"Fuzz{seed}_{m}"
    src, nx, nu, bounded = make_struct(name)
    mid, oid = register_model(src, name), O.register_model(src, name)
    params = np.concatenate([rng.uniform(-0.5, 0.5, 4), rng.uniform(-1.0, 1.0, nx), np.zeros(12 - nx)])      # p[0..3]: dynamics constants, p[4..]: the cost's targets
    for strategy in ("single", "sequential", "trustregion", "linesearch", "centralized", "centralized_dense"):
        N = int(rng.integers(1, 5)) if strategy == "centralized_dense" else int(rng.integers(1, 300))
        T = int(rng.integers(1, 8)) if strategy == "centralized_dense" else int(rng.integers(1, 50))
        dt = float(rng.choice([0.05, 0.1, 0.2]))
        outer = int(rng.integers(1, 4))
        max_it = int(rng.choice([2, 100]))
        x0 = rng.uniform(-1.5, 1.5, (N, nx))
        U0 = None if rng.random() < 0.5 or strategy.startswith("centralized") else 0.1 * rng.standard_normal((N, T, nu))
        with BatchSolver(mid, N, T, dt, params=params, no_input_bounds=not bounded) as s:
            s.set_params({"max_iterations": max_it, "tolerance": 1e-5, "max_ms": float("inf")})
            s.set_initial_states(x0)
            s.set_controls(U0)
            s.initialize_problem()
            s.run(strategy, outer)
            e = s.solution()
        o = O.solve_batch(oid, x0, U0, T, dt, strategy=strategy, max_outer=outer, max_iterations=max_it, tolerance=1e-5, params=params)
        ok = all(np.array_equal(e[k], o[k], equal_nan=True) for k in ("cost", "X", "U")) and e["iters"].tolist() == o["iters"].tolist()
        if not ok:
            bad += 1
            print("MISMATCH", name, strategy, "N", N, "T", T, "nx", nx, "nu", nu, "max |dcost|", np.nanmax(np.abs(e["cost"] - o["cost"])), flush=True)
            print(src, flush=True)
    print(f"model {m}: nx {nx} nu {nu} bounded {bounded}: six strategies done, {bad} mismatches so far, {time.time() - t0:.0f} s", flush=True)
print(f"{n_models} random workloads x 6 strategies: {bad} mismatches (seed {seed})")
sys.exit(1 if bad else 0)
