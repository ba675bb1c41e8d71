#!/bin/bash
# full GPU tier + the driver's bench command (run through gpurun from the repo root)
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r4_gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r4_gpu_tests.txt
timeout -k 10 400 python bench.py > gpurun_out/r4_bench_c3.json 2> gpurun_out/r4_bench_c3.err; echo "bench rc=$?"; tail -3 gpurun_out/r4_bench_c3.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_bench_c3.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k: d['roofline'][k] for k in ('kernel','frac','avg_launch_us','traffic')})
print('unfused', {k: d['unfused_round'][k] for k in ('ms_per_step','speedup_of_the_fused_round')}, 'sweep frac', d['unfused_round']['roofline_sweep']['frac'])
print('parity', d['parity']['vs_det_oracle'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['spread_rel'])
print('c2', d['also_config2']['ms_per_step'], d['also_config2']['parity']['vs_det_oracle'])
PY
