import math, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
N, T = 4096, 100
s = BatchSolver(1, N, T, 0.5)
s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
s.set_initial_states(single_track_x0(N)); s.set_controls(None); s.initialize_problem(); s.save_initial()
s.restore_initial(); s.run_sequential(10)
for prof in (False, True, False, True):
    s.set_profiling(prof); s.reset_stats()
    t0 = time.perf_counter()
    for _ in range(5):
        s.restore_initial(); s.run_sequential(10)
    el = time.perf_counter() - t0
    st = s.stats()
    print(f"profiling={prof}: {1e3*el/5:.2f} ms/step, {st['agent_iterations']/el/1e6:.2f} M agent-it/s, rounds {st['batch_iterations']}")
    s.kernel_times()
