set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.txt 2>&1
tail -3 gpurun_out/final_gpu_tests.txt
timeout -k 10 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
rm -rf gpurun_out/final_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_prof -o r1 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-also > gpurun_out/final_bench_under_rocprof.json 2> gpurun_out/final_rocprof.err
find gpurun_out/final_prof -type f | head -20
