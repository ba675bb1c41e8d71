#!/bin/bash
# short bench lines of the given workloads (default: all four), no tests
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in ${@:-c3 c2 c4 c5}; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline --no-parity > gpurun_out/b_$w.json 2> gpurun_out/b_$w.err || tail -3 gpurun_out/b_$w.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/b_$w.json").read())
print("$w value %.4g  ms/step %.3f  frac %.3f"%(j["value"], j["ms_per_step"], j["roofline"]["frac"]), {k:round(v["avg_us"],1) for k,v in j["kernels"].items()})
PY
done
