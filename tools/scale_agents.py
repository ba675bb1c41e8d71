"""Per-kernel time versus agents per GPU (single-track, H=100): shows which kernels are latency-bound
(time flat in N) and where the chip starts to fill."""
import math, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
T = 100
for N in (1024, 2048, 4096, 8192, 16384, 32768):
    s = BatchSolver(1, N, T, 0.5)
    s.set_params({"max_iterations": 3, "tolerance": 1e-5, "max_ms": math.inf})
    s.set_initial_states(single_track_x0(N)); s.set_controls(None); s.initialize_problem()
    s.solve(); s.initialize_problem()
    s.set_profiling(True)
    s.solve()
    kt = s.kernel_times()
    print(N, {k: round(1e3 * v["total_ms"] / v["launches"], 1) for k, v in kt.items() if k in ("backward", "forward", "sweep", "sweep_inkernel", "select", "commit")})
    s.close()
