set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "not maximum" > gpurun_out/r3_t1.txt 2>&1 || { tail -30 gpurun_out/r3_t1.txt; exit 1; }
tail -3 gpurun_out/r3_t1.txt
for W in c5 c3; do
timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline --no-parity --no-also --no-unfused > gpurun_out/r3_b1_$W.json 2> gpurun_out/r3_b1.err || { tail -20 gpurun_out/r3_b1.err; exit 1; }
python - $W <<'PY'
import json, sys
d=json.load(open('gpurun_out/r3_b1_%s.json' % sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items() if v['share_of_kernel_time'] > 0.01})
PY
done
