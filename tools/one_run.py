"""One strategy run of a BASELINE workload shard (bench.py's WORKLOADS) for counter collection under rocprofv3.
usage: one_run.py [c3|c2|c4|c5] [agents] [--algorithmic out.json] [--option key=value ...]
--algorithmic: no profiler needed; writes the per-kernel launch counts and algorithmic bytes of the same
(deterministic) run, which tools/summarise_profiles.py holds the PMC byte counters against."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

argv = list(sys.argv[1:])
alg_out = None
if "--algorithmic" in argv:
    i = argv.index("--algorithmic")
    alg_out = argv[i + 1]
    del argv[i:i + 2]
options = []
while "--option" in argv:        # kernel forms (mas_hip_set_option): bit-identical results, different launches
    i = argv.index("--option")
    options.append(argv[i + 1].split("="))
    del argv[i:i + 2]
args = argv
wname = args[0] if args else "c3"
w = dict(bench.WORKLOADS[wname])
N = int(args[1]) if len(args) > 1 else w["agents"]
s = bench.make_solver(w, N, 0, N, 0)
for key, value in options:
    s.set_option(key, int(value))
if alg_out:
    s.set_profiling(True)
s.reset_stats()
bench.one_step(s, w)
st = s.stats()
print("workload", wname, "agents", N, "iterations", st["agent_iterations"], "rounds", st["batch_iterations"])
if alg_out:
    json.dump({"workload": wname, "agents": N, "stats": st, "kernels": s.kernel_times()}, open(alg_out, "w"), indent=1)
