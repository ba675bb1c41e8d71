"""Does a shard of INDEPENDENT agents run faster as K handles of N / K agents on K streams of one GPU (host threads) than as one handle?
Rounds in which every agent is live follow throughput (HBM in the sweep, LDS in the backward pass, FP64 issue in the line search): K
pipelines offset in time overlap kernels with different limits.  usage: split_handles.py [c3|c4|c5] [steps]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402

wname = sys.argv[1] if len(sys.argv) > 1 else "c5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
w = dict(bench.WORKLOADS[wname])
N = w["agents"]
out = {"workload": wname, "agents": N, "steps": steps, "splits": {}}
ref = None
for K in (1, 2, 4, 8):
    per = N // K
    hs = [bench.make_solver(w, per, k * per, N, 0) for k in range(K)]
    def loop(s, n):
        for _ in range(n):
            bench.one_step(s, w)
        s.synchronize() if hasattr(s, "synchronize") else None
    for s in hs:
        loop(s, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=loop, args=(s, steps)) for s in hs]
    [t.start() for t in th]
    [t.join() for t in th]
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    cost = np.concatenate([s.solution()["cost"] for s in hs])
    its = sum(s.stats()["agent_iterations"] for s in hs)
    if ref is None:
        ref = cost
    out["splits"][K] = {"ms_per_step": ms, "bit_equal_to_one_handle": bool(np.array_equal(cost, ref))}
    print(K, "handles:", round(ms, 3), "ms per step, equal:", np.array_equal(cost, ref), flush=True)
    for s in hs:
        s.close()
print(json.dumps(out))
