#!/bin/bash
# A/B builds of the fused linearise + Riccati kernel: helper wavefronts, ring slots, priorities (run on the GPU box)
cd "$GRAFT_REPO_ROOT"
for V in "-DMAS_FUSED_HELPERS=4 -DMAS_FUSED_RING=2" "-DMAS_FUSED_HELPERS=4 -DMAS_FUSED_RING=3"; do
  MAS_HIPCC_EXTRA="$V" python3 -c "import __graft_entry__ as g; g.build_hip(force=True)" > gpurun_out/fv_build.log 2>&1 || { echo "build failed: $V"; tail -5 gpurun_out/fv_build.log; continue; }
  for W in c3; do
    timeout -k 10 200 python3 bench.py --workload $W --no-cpu-baseline --no-parity --no-also --steps 10 > gpurun_out/fv.json 2> gpurun_out/fv.err || { echo "bench failed: $V"; tail -3 gpurun_out/fv.err; continue; }
    python3 - "$V" $W <<'PY'
import json, sys
d = json.load(open('gpurun_out/fv.json'))
print(sys.argv[1], sys.argv[2], "ms %.3f" % d['ms_per_step'], {k: v['avg_us'] for k, v in d['kernels'].items() if k in ('linearise_riccati', 'forward')})
PY
  done
done
