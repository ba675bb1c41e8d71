"""Per-kernel timing of one BASELINE config shard on one GPU (HIP events / in-kernel stamps).
usage: python tools/profile_config.py c2|c3|c4|c5 [agents]"""
import math, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0, lqr_x0, bicycle6_x0
cfg = sys.argv[1]
CFG = {"c2": (1, 256, 100, 0.5, "centralized", 1), "c3": (1, 4096, 100, 0.5, "sequential", 10),
       "c4": (2, 4096, 50, 0.1, "sequential", 10), "c5": (5, 8192, 200, 0.25, "trustregion", 10)}
model, N, T, dt, strat, outer = CFG[cfg]
if len(sys.argv) > 2: N = int(sys.argv[2])
x0 = {1: single_track_x0, 2: lqr_x0, 5: bicycle6_x0}[model](N)
s = BatchSolver(model, N, T, dt)
s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
s.set_initial_states(x0); s.set_controls(None); s.initialize_problem(); s.save_initial()
s.run(strat, outer)
s.restore_initial(); s.reset_stats(); s.set_profiling(True)
t0 = time.perf_counter(); s.run(strat, outer); el = time.perf_counter() - t0
st = s.stats(); kt = s.kernel_times()
print(cfg, f"N={N} T={T} {strat}: {1e3*el:.2f} ms, {st['agent_iterations']/el/1e6:.3f} M agent-it/s, rounds {st['batch_iterations']}, converged {st['converged_solves']}/{st['solves']}")
for k, v in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"   {k:18s} {v['launches']:5d} x {1e3*v['total_ms']/v['launches']:9.1f} us = {v['total_ms']:8.2f} ms")
