#!/bin/bash
# A/B builds of libmas_hip.so on the GPU box: one bench line per (build variant, workload).
#   WL="c3 c4" VARIANTS="|-DMAS_FUSED_HELPERS=2 -DMAS_FUSED_RING=2|-DMAS_FUSED_NO_PRIO" bash tools/ab_builds.sh
# VARIANTS: '|'-separated values of MAS_HIPCC_EXTRA (extra hipcc options, __graft_entry__.build_hip); the empty one is the product build.
# Macros that exist for this purpose: MAS_FUSED_HELPERS / MAS_FUSED_RING (helper wavefronts and ring slots of the fused kernel),
# MAS_FUSED_NO_PRIO (no s_setprio), MAS_LPA6 (lanes per agent of the LDS-tile backward kernel for nx = 6), MAS_HIP_ENV_SWITCHES
# (kernel forms from MAS_HIP_<KEY> environment variables instead of mas_hip_set_option).
cd "$GRAFT_REPO_ROOT"
WL=${WL:-c3}
IFS='|' read -ra VARS <<< "${VARIANTS:-|}"
for V in "${VARS[@]}"; do
  MAS_HIPCC_EXTRA="$V" python3 -c "import __graft_entry__ as g; g.build_hip(force=True)" > gpurun_out/ab_build.log 2>&1 || { echo "build failed: $V"; tail -5 gpurun_out/ab_build.log; continue; }
  for W in $WL; do
    timeout -k 10 300 python3 bench.py --workload $W --no-cpu-baseline --no-parity --no-also --no-unfused --steps 10 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "bench failed: $V"; tail -3 gpurun_out/ab.err; continue; }
    python3 - "$V" $W <<'PY'
import json, sys
d = json.load(open('gpurun_out/ab.json'))
print(repr(sys.argv[1]), sys.argv[2], "ms %.3f" % d['ms_per_step'], {k: (v['avg_us'], v['launches']) for k, v in d['kernels'].items() if v['share_of_kernel_time'] > 0.02})
PY
  done
done
