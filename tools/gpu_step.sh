#!/bin/bash
# scratch: GPU tests + bench lines of c5 / c4
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/step_tests.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/step_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/forms_sweep.py c5 backward_wave_max=0,512,1024,2048 5 || exit 1
for w in c5; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-also --no-unfused > gpurun_out/step_bench_$w.json 2> gpurun_out/step_bench_$w.err || exit 1
  python -c "import json; d=json.load(open('gpurun_out/step_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d.get('parity',{}).get('vs_det_oracle'))"
done
timeout -k 10 600 python tools/fuzz_parity.py 300 11 > gpurun_out/step_fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/step_fuzz.txt
