set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r3_t1.txt 2>&1 || { tail -40 gpurun_out/r3_t1.txt; exit 1; }
tail -3 gpurun_out/r3_t1.txt
