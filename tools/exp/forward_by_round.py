"""Why does the forward kernel take 2.5x longer in the late rounds of a solve?  Runs one solve (all agents start
together) with the debug trace on; run under rocprofv3 --kernel-trace to get the per-launch durations."""
import math, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
N = int(sys.argv[1]); total = int(sys.argv[2]) if len(sys.argv) > 2 else N; first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
s = BatchSolver(1, N, 100, 0.5)
s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf, "debug": 1})
s.set_initial_states(single_track_x0(N, first=first, total=total)); s.set_controls(None); s.initialize_problem()
s.solve()
print("iters", s.solution()["iters"].max())
