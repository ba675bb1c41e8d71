#!/bin/bash
# Profile evidence of one bench workload on the GPU box (run through gpurun from the repo root):
#   tools/collect_profiles.sh c3 [round-tag]
#   1. rocprofv3 --kernel-trace --stats of `bench.py --workload W --steps 5 --warmup 1`  -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, MI355X_MICROARCH.md) of one strategy run
#   3. four SQ counter passes of one strategy run
#   4. tools/summarise_profiles.py -> profiles/<tag>_*_W.{json,csv}
# Counter passes never combine --pmc with a sys/hip/hsa trace (gpurun refuses that); the program after `--` is python3 itself.
set -e
W=${1:-c3}; TAG=${2:-r4}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${TAG}_${W}
rm -rf "$OUT"; mkdir -p "$OUT"
python3 -c "from multi_agent_solver_amd.source_hash import source_hash; print(source_hash())" > "$OUT/source_hash.txt"      # the sources these counters are collected from
python3 tools/one_run.py "$W" --algorithmic "$OUT/algorithmic.json" > "$OUT/algorithmic.log" 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o s -- python3 bench.py --workload "$W" --steps 5 --warmup 1 --no-cpu-baseline --no-also --no-parity > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
echo "stats pass done"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d "$OUT/pmc_$C" -o p -- python3 tools/one_run.py "$W" > "$OUT/pmc_$C.log" 2>&1
  echo "pmc $C done"
done
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_WAVES" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d "$OUT/sq$i" -o p -- python3 tools/one_run.py "$W" > "$OUT/sq$i.log" 2>&1 || echo "SQ set $i failed"
  echo "sq set $i done"
done
python3 tools/summarise_profiles.py "$W" "$TAG" "$OUT"
# (profiles/ does not travel back from the GPU box: run the same summarise command on the merged gpurun_out/ directory afterwards)
