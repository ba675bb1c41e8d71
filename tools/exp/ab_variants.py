"""A/B timing of libmas_hip variants (tools/exp/build_variants.sh) on one workload: per-kernel HIP-event averages of 3 strategy
runs each, and a bit-for-bit comparison of every variant's result with the first one's.
usage: ab_variants.py <workload> name1 name2 ..."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from multi_agent_solver_amd import _capi, BatchSolver  # noqa: E402
import math  # noqa: E402

wname = sys.argv[1]
w = dict(bench.WORKLOADS[wname])
N = w["agents"]
ref = None
for name in sys.argv[2:]:
    lib = _capi.load(os.path.join(ROOT, "tools", "exp", "variants", f"libmas_hip_{name}.so"))
    s = BatchSolver(w["model"], N, w["T"], w["dt"], lib=lib)
    s.set_params({"max_iterations": w["max_iterations"], "tolerance": w["tolerance"], "max_ms": math.inf})
    s.set_initial_states(bench.initial_states(w, N, 0, N)); s.set_controls(None); s.initialize_problem(); s.save_initial()
    bench.one_step(s, w)
    s.reset_stats()
    t0 = time.perf_counter()
    for _ in range(3):
        bench.one_step(s, w)
    el = (time.perf_counter() - t0) / 3
    s.set_profiling(True)
    for _ in range(3):
        bench.one_step(s, w)
    kt = s.kernel_times()
    sol = s.solution()
    s.close()
    same = "ref" if ref is None else ("same" if all(np.array_equal(sol[k], ref[k]) for k in ("cost", "X", "U", "iters")) else "DIFFERENT")
    if ref is None:
        ref = sol
    print(f"{name:16s} {1e3*el:8.3f} ms/step  {same:9s} " + "  ".join(f"{k}={1e3*v['total_ms']/max(1,v['launches']):.1f}" for k, v in sorted(kt.items()) if "inkernel" not in k and v["total_ms"] > 0.05))
