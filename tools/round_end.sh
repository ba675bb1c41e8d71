#!/bin/bash
# Everything a change under csrc/ or include/ asks for, in ONE gpurun call (the box is acquired once):
#   full GPU tier -> counters of c3 c2 c5 c4 at the current sources -> bench lines that quote them -> trust-region timing -> literal-solve kernel stats
# Afterwards, locally: tools/round_end_local.sh (summaries from the merged gpurun_out/ into profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -n 3 gpurun_out/r4_gpu_tests.txt
[ $rc -eq 0 ] || exit 1
for W in c3 c2 c5 c4; do bash tools/collect_profiles.sh $W r4 > gpurun_out/col_$W.log 2>&1 || { echo "collection of $W failed"; tail -n 5 gpurun_out/col_$W.log; exit 1; }; echo "collected $W"; done
timeout -k 10 300 python bench.py > gpurun_out/r4_bench_c3.json 2> gpurun_out/r4_bench_c3.err || exit 1
timeout -k 10 200 python bench.py --workload c4 > gpurun_out/r4_bench_c4.json 2> gpurun_out/r4_bench_c4.err || exit 1
timeout -k 10 300 python bench.py --workload c5 > gpurun_out/r4_bench_c5.json 2> gpurun_out/r4_bench_c5.err || exit 1
timeout -k 10 200 python tools/trust_c3_timing.py > gpurun_out/trust_c3.log 2>&1 || exit 1
rm -rf gpurun_out/dense_stats
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dense_stats -o d -- python3 tools/dense_one_run.py 256 > gpurun_out/dense_one_run.log 2>&1 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" | tail -n 1
echo "round_end done"
