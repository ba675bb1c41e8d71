"""64 identical copies of the slowest agent of the 64-agent shard: every lane stays active in every round.
If the forward kernel still slows down in the late rounds the cause is the data, not the exec mask."""
import math, sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
def run(x0, tag):
    N = x0.shape[0]
    s = BatchSolver(1, N, 100, 0.5)
    s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf, "debug": 1})
    s.set_initial_states(x0); s.set_controls(None); s.initialize_problem()
    print("RUN", tag, flush=True)
    s.solve()
    it = s.solution()["iters"]
    s.close()
    return it
x0 = single_track_x0(64, first=0, total=4096)
it = run(x0, "shard")
k = int(np.argmax(it)); print("slowest agent", k, "iters", it[k], flush=True)
run(np.repeat(x0[k:k + 1], 64, axis=0), "identical")
