#!/bin/bash
# round-3 kernel-iteration loop on the GPU box: the whole GPU test tier, then a C3 bench line (+ C2 inside it)
# usage: tools/exp/check.sh <tag> [pytest -k expression]
TAG=${1:-x}; K=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ -n "$K" ]; then timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/ck_${TAG}_tests.txt 2>&1; else timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/ck_${TAG}_tests.txt 2>&1; fi
tail -3 gpurun_out/ck_${TAG}_tests.txt
timeout -k 10 300 python bench.py --no-cpu-baseline ${BENCH_ARGS:-} > gpurun_out/ck_${TAG}_bench.json 2> gpurun_out/ck_${TAG}_bench.err || tail -5 gpurun_out/ck_${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/ck_${TAG}_bench.json").read())
print("value %.4g  ms/step %.3f  frac %.3f parity %s"%(j["value"], j["ms_per_step"], j["roofline"]["frac"], (j.get("parity") or {}).get("vs_det_oracle")))
print({k:round(v["avg_us"],1) for k,v in j["kernels"].items()})
if j.get("also_config2"): print("c2 %.3f"%j["also_config2"]["ms_per_step"], j["also_config2"]["kernels_avg_us"], (j["also_config2"].get("parity") or {}).get("vs_det_oracle"))
PY
