set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "staged or single_track_h100 or lqr_identity or fused_round or kernel_forms or lane" > gpurun_out/r3_t1.txt 2>&1 || { tail -30 gpurun_out/r3_t1.txt; exit 1; }
tail -3 gpurun_out/r3_t1.txt
for W in c3 c4; do
timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-parity --no-also --steps 10 > gpurun_out/r3_b1_$W.json 2> gpurun_out/r3_b1.err || { tail -20 gpurun_out/r3_b1.err; exit 1; }
python - $W <<'PY'
import json, sys
d=json.load(open('gpurun_out/r3_b1_%s.json' % sys.argv[1]))
print(sys.argv[1], d['value'], d['ms_per_step'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})
PY
done
