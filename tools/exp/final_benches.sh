#!/bin/bash
# the plain (un-profiled) bench lines of all four workloads, as committed under profiles/<tag>_bench_<w>.json
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/final_lines
for w in c3 c2 c4 c5; do
  timeout -k 10 500 python bench.py --workload $w > gpurun_out/final_lines/${TAG}_bench_$w.json 2> gpurun_out/final_lines/${TAG}_bench_$w.err || tail -3 gpurun_out/final_lines/${TAG}_bench_$w.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/final_lines/${TAG}_bench_$w.json").read())
print("$w value %.4g ms/step %.3f frac %.3f parity %s cpu %.4g"%(j["value"], j["ms_per_step"], j["roofline"]["frac"], (j.get("parity") or {}).get("vs_det_oracle"), (j.get("cpu_baseline") or {}).get("value", 0)))
PY
done
