"""tests/golden/centralized_dense_large.json: the literal centralised solve (oracle, `centralized_dense_structured`, det trig) on BASELINE
config 2's workload -- single-track, H = 100, agents on the circle -- at 128 and 256 agents.  About 10 s and 10 minutes of eight cores;
the GPU tier compares mas_hip_run_centralized_dense with these vectors (per-agent costs bit for bit, SHA-256 of the trajectories).
    python tools/make_dense_goldens.py [agents ...]"""
import hashlib, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle

def main():
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256]
    oracle.build(); oracle.set_trig_mode("det")
    path = os.path.join(ROOT, "tests", "golden", "centralized_dense_large.json")
    out = json.load(open(path)) if os.path.exists(path) else {"_what": __doc__, "cases": {}}
    for n in sizes:
        t0 = time.time()
        r = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(n), None, 100, 0.5, strategy="centralized_dense_structured")
        out["cases"][str(n)] = dict(
            agents=n, horizon=100, dt=0.5, trig="det", total_cost=float(r["total_cost"]).hex(), total_cost_decimal=float(r["total_cost"]),
            iterations=int(r["iters"][0]), reg_events=int(r["stats"]["reg_events"]), trials=int(r["stats"]["trials"]),
            cost=[float(v).hex() for v in r["cost"]],
            sha256_X=hashlib.sha256(np.ascontiguousarray(r["X"]).tobytes()).hexdigest(),
            sha256_U=hashlib.sha256(np.ascontiguousarray(r["U"]).tobytes()).hexdigest(),
            oracle_seconds=round(time.time() - t0, 1))
        print(n, out["cases"][str(n)]["total_cost_decimal"], out["cases"][str(n)]["iterations"], out["cases"][str(n)]["oracle_seconds"], flush=True)
        json.dump(out, open(path, "w"), indent=1)

if __name__ == "__main__":
    main()
