"""Rounds and time of every rank's shard for the weak-scaling bench (4096 agents per GPU, agent i of N_total at
theta = 2*pi*i/N_total): what the slowest rank of an N-GPU run will take, measured one shard at a time on one GPU."""
import math, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
PER = 4096
for world in (1, 2, 4, 8):
    rows = []
    for rank in range(world):
        s = BatchSolver(1, PER, 100, 0.5)
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(single_track_x0(PER, first=rank * PER, total=world * PER)); s.set_controls(None)
        s.initialize_problem(); s.save_initial(); s.run_sequential(10)
        s.restore_initial(); s.reset_stats()
        t0 = time.perf_counter(); s.run_sequential(10); el = time.perf_counter() - t0
        st = s.stats(); rows.append((st["batch_iterations"], round(1e3 * el, 2), st["agent_iterations"])); s.close()
    worst = max(r[1] for r in rows)
    print(f"world {world}: rounds/ms/iterations per rank {rows}  -> slowest {worst} ms, aggregate {sum(r[2] for r in rows) / worst / 1e3:.2f} M agent-it/s")
