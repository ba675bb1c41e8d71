"""One C3 strategy run (4096 single-track agents, sequential, max_outer 10) for counter collection."""
import math, sys
sys.path.insert(0, "/root/repo")
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
s = BatchSolver(1, N, 100, 0.5)
s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
s.set_initial_states(single_track_x0(N)); s.set_controls(None); s.initialize_problem()
s.run_sequential(10)
print("iterations", s.stats()["agent_iterations"])
