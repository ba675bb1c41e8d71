"""--strategy trustregion on the reference's own single-track workload at C3's size (4096 agents, H = 100, max_outer 10): two-kernel rounds with
the trust-region epilogue inside the fused kernel (default) against the five-kernel round ("fused_trust": 0), same process, results compared.
    python tools/trust_c3_timing.py [agents] [steps]   (GPU box)"""
import json, math, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multi_agent_solver_amd import BatchSolver
from multi_agent_solver_amd.batch import single_track_x0

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
out, sols = {}, {}
for name, opts in (("fused_epilogue", {}), ("five_kernel_round", {"fused_trust": 0})):
    with BatchSolver(1, N, 100, 0.5) as s:
        for k, v in opts.items(): s.set_option(k, v)
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(single_track_x0(N)); s.set_controls(None); s.initialize_problem(); s.save_initial()
        for _ in range(2): s.restore_initial(); s.run("trustregion", 10)
        s.reset_stats()
        t0 = time.perf_counter()
        for _ in range(steps): s.restore_initial(); s.run("trustregion", 10)
        el = time.perf_counter() - t0
        st = s.stats(); sols[name] = s.solution(); sols[name]["radii"] = s.trust_radii()
        s.set_profiling(True); s.restore_initial(); s.run("trustregion", 10); kt = s.kernel_times()
        out[name] = dict(ms_per_run=1e3 * el / steps, agent_iterations_per_s=st["agent_iterations"] / el, rounds=st["batch_iterations"] // steps,
                         kernels_avg_us={k: round(1e3 * v["total_ms"] / max(1, v["launches"]), 1) for k, v in kt.items()}, launches={k: v["launches"] for k, v in kt.items()})
a, b = sols["fused_epilogue"], sols["five_kernel_round"]
out["identical"] = all(np.array_equal(a[k], b[k]) for k in ("X", "U", "cost", "iters", "radii"))
out["speedup"] = out["five_kernel_round"]["ms_per_run"] / out["fused_epilogue"]["ms_per_run"]
print(json.dumps(out, indent=1))
