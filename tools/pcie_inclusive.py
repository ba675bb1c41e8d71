"""C3 through the HOST-buffer entry points (one PCIe copy each way): rate including the transfers."""
import math, sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
N, T = 4096, 100
x0 = single_track_x0(N); U0 = np.zeros((N, T, 2))
s = BatchSolver(1, N, T, 0.5)
s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
for rep in range(3):
    s.reset_stats()
    t0 = time.perf_counter()
    s.set_initial_states(x0); s.set_controls(U0); s.initialize_problem()
    t1 = time.perf_counter()
    s.run_sequential(10)
    t2 = time.perf_counter()
    sol = s.solution()
    t3 = time.perf_counter()
    it = s.stats()["agent_iterations"]
    print(f"upload+init {1e3*(t1-t0):.2f} ms, strategy {1e3*(t2-t1):.2f} ms, download {1e3*(t3-t2):.2f} ms -> {it/(t3-t0)/1e6:.2f} M agent-it/s incl. PCIe, {it/(t2-t1)/1e6:.2f} resident")
