#!/bin/bash
# A/B builds of the fused linearise + Riccati kernel (run on the GPU box): MAS_HIPCC_EXTRA variants, bench per workload
cd "$GRAFT_REPO_ROOT"
WL=${WL:-c4}
IFS='|' read -ra VARS <<< "${VARIANTS:-|-DMAS_EXP_NO_LOADS|-DMAS_EXP_NO_COMMIT}"
for V in "${VARS[@]}"; do
  MAS_HIPCC_EXTRA="$V" python3 -c "import __graft_entry__ as g; g.build_hip(force=True)" > gpurun_out/fv_build.log 2>&1 || { echo "build failed: $V"; tail -5 gpurun_out/fv_build.log; continue; }
  for W in $WL; do
    timeout -k 10 200 python3 bench.py --workload $W --no-cpu-baseline --no-parity --no-also --steps 10 > gpurun_out/fv.json 2> gpurun_out/fv.err || { echo "bench failed: $V"; tail -3 gpurun_out/fv.err; continue; }
    python3 - "$V" $W <<'PY'
import json, sys
d = json.load(open('gpurun_out/fv.json'))
print(repr(sys.argv[1]), sys.argv[2], "ms %.3f" % d['ms_per_step'], {k: (v['avg_us'], v['launches']) for k, v in d['kernels'].items() if k in ('linearise_riccati', 'forward', 'backward', 'sweep')})
PY
  done
done
