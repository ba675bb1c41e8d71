"""one literal centralised solve (mas_hip_run_centralized_dense) of the C2 workload at N agents, for rocprofv3:
   rocprofv3 --kernel-trace --stats -d out -o d -- python3 tools/dense_one_run.py [agents]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd import BatchSolver
from multi_agent_solver_amd.batch import single_track_x0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with BatchSolver(1, N, 100, 0.5) as s:
    s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
    s.set_initial_states(single_track_x0(N)); s.set_controls(None); s.initialize_problem()
    s.run("centralized_dense", 1)
    print(s.total_cost(), s.stats())
