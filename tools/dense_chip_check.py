"""The literal centralised solve as chip-wide launches (mas_dense_chip.hpp) against the oracle's structured restatement and against the
one-workgroup form, with timings: python tools/dense_chip_check.py [max_agents]  (GPU box)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import run_batch
from multi_agent_solver_amd import _capi
from oracle import oracle

lib = _capi.load()
oracle.build(); oracle.set_trig_mode("det")
top = int(sys.argv[1]) if len(sys.argv) > 1 else 64
low = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cases = [(1, 2, 10), (1, 3, 40), (1, 16, 12), (3, 2, 12), (2, 3, 10), (5, 2, 8), (1, 8, 100), (1, 16, 100), (1, 32, 100), (1, 64, 100), (1, 128, 100), (1, 256, 100)]
for model, N, T in cases:
    if N > top or N < low: continue
    nx, nu, _ = oracle.model_dims(model)
    if model == 1: x0 = oracle.single_track_x0(N); dt = 0.5
    elif model == 3: x0 = np.array([[np.pi - 0.05, 0.0], [np.pi - 0.1, 0.1]]); dt = 0.05
    else: x0 = np.tile(np.eye(nx)[0], (N, 1)) + 0.1 * np.arange(N)[:, None] + (np.array([0, 0, 0, 2.0, 0, 0])[None, :nx] if model == 5 else 0); dt = 0.1
    if N > 64:                      # against tests/golden/centralized_dense_large.json (the oracle needs ten minutes at 256 agents)
        import hashlib, json
        from multi_agent_solver_amd.batch import BatchSolver
        g = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "centralized_dense_large.json")))["cases"][str(N)]
        with BatchSolver(model, N, T, dt, lib=lib) as sv:
            sv.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": float("inf")})
            sv.set_initial_states(x0); sv.set_controls(None); sv.initialize_problem()
            times = []
            for rep in range(2):
                sv.set_profiling(rep == 1)
                t0 = time.time(); sv.run("centralized_dense", 1); times.append(round(time.time() - t0, 3))
            e = sv.solution(); e["stats"] = sv.stats(); kt = sv.kernel_times()
        ok = dict(total=float(e["total_cost"]).hex() == g["total_cost"], cost=[float(v).hex() for v in e["cost"]] == g["cost"], iters=int(e["iters"][0]) == g["iterations"],
                  X=hashlib.sha256(np.ascontiguousarray(e["X"]).tobytes()).hexdigest() == g["sha256_X"], U=hashlib.sha256(np.ascontiguousarray(e["U"]).tobytes()).hexdigest() == g["sha256_U"])
        print(dict(model=model, N=N, T=T, ok=all(ok.values()), same=ok, run_s=times, iters=int(e["iters"][0]), total=e["total_cost"]), flush=True)
        print({k: (v["launches"], round(v["total_ms"], 2)) for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])}, flush=True)
        continue
    t0 = time.time(); e = run_batch(lib, model, x0, None, T, dt, "centralized_dense"); t1 = time.time()
    o = oracle.solve_batch(model, x0, None, T, dt, strategy="centralized_dense_structured"); t2 = time.time()
    same = {k: bool(np.array_equal(e[k], o[k])) for k in ("X", "U", "cost", "iters")}
    same["total"] = e["total_cost"] == o["total_cost"]; same["regs"] = e["stats"]["reg_events"] == o["stats"]["reg_events"]
    line = dict(model=model, N=N, T=T, ok=all(same.values()), gpu_s=round(t1 - t0, 3), oracle_s=round(t2 - t1, 3), iters=int(e["iters"][0]), total=e["total_cost"], oracle_total=o["total_cost"])
    if not line["ok"]: line["same"] = same
    if N <= 32:
        old = run_batch(lib, model, x0, None, T, dt, "centralized_dense", options={"dense_chip": 0})
        line["one_workgroup_equal"] = all(np.array_equal(e[k], old[k]) for k in ("X", "U", "cost", "iters")) and e["total_cost"] == old["total_cost"]
    print(line, flush=True)
