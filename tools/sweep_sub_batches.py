import json, math, sys, time
sys.path.insert(0, "/root/repo")
import torch
from multi_agent_solver_amd import BatchSolver, MODEL_SINGLE_TRACK
from multi_agent_solver_amd.batch import single_track_x0
N, T = 4096, 100
x0 = single_track_x0(N)
for S in (1, 2, 4, 8):
    s = BatchSolver(MODEL_SINGLE_TRACK, N, T, 0.5)
    s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf, "sub_batches": S})
    s.set_initial_states(x0); s.set_controls(None); s.initialize_problem(); s.save_initial()
    s.restore_initial(); s.run_sequential(10)
    s.reset_stats()
    t0 = time.perf_counter()
    for _ in range(5):
        s.restore_initial(); s.run_sequential(10)
    el = time.perf_counter() - t0
    st = s.stats()
    print(f"sub_batches={S}: {1e3*el/5:.2f} ms/step, {st['agent_iterations']/el/1e6:.2f} M agent-it/s, rounds {st['batch_iterations']}")
    s.close()
