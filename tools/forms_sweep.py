"""Run time of one bench workload under alternative kernel forms (mas_hip_set_option): bit-identical results, different launches.
usage: forms_sweep.py <c2|c3|c4|c5> key=v1,v2,... [steps]        e.g.  forms_sweep.py c5 forward_quad_max=0,1600,4800,100000"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
import bench  # noqa: E402

wname = sys.argv[1]
key, values = sys.argv[2].split("=")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = dict(bench.WORKLOADS[wname])
N = w["agents"]
for v in values.split(","):
    s = bench.make_solver(w, N, 0, N, 0)
    s.set_option(key, int(v))
    for _ in range(2):
        bench.one_step(s, w)
    s.synchronize() if hasattr(s, "synchronize") else torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bench.one_step(s, w)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    st = s.stats()
    print(f"{wname} {key}={v}: {ms:.3f} ms per run, total cost {s.solution()['cost'].sum():.9g}", flush=True)
    s.close()
