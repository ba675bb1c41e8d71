#!/bin/bash
# C3 with different upper bounds of the four-lanes-per-rollout line search (MAS_HIP_FORWARD_QUAD_MAX slots; 0 = never)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for q in ${QMAX:-0 1600 2400 3200 4096}; do
  MAS_HIP_FORWARD_QUAD_MAX=$q timeout -k 10 200 python bench.py --no-cpu-baseline --no-parity --steps 8 > gpurun_out/qt_$q.json 2> gpurun_out/qt_$q.err
  python - <<PY
import json
j=json.loads(open("gpurun_out/qt_$q.json").read())
print("quad_max $q: value %.4g ms/step %.3f fwd %.1f bwd %.1f"%(j["value"], j["ms_per_step"], j["kernels"]["forward"]["avg_us"], j["kernels"]["backward"]["avg_us"]), "c2 %.3f"%j["also_config2"]["ms_per_step"])
PY
done
