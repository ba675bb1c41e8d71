#!/bin/bash
# C5 with different upper bounds of the four-lanes-per-rollout line search (MAS_HIP_FORWARD_QUAD_MAX slots)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for q in ${QMAX:-1600 2400 3200 4800 8192}; do
  MAS_HIP_FORWARD_QUAD_MAX=$q timeout -k 10 300 python bench.py --workload c5 --no-cpu-baseline --no-parity --steps 3 > gpurun_out/qt5_$q.json 2> gpurun_out/qt5_$q.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/qt5_$q.json").read())
print("quad_max $q: value %.4g ms/step %.3f fwd %.1f bwd %.1f"%(j["value"], j["ms_per_step"], j["kernels"]["forward"]["avg_us"], j["kernels"]["backward"]["avg_us"]))
PY
done
