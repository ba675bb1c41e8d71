# plain bench lines of every workload at HEAD -> gpurun_out/r3_bench_<w>.json (copied into profiles/ afterwards)
set -e
mkdir -p gpurun_out
for W in c3 c2 c4 c5; do
  timeout -k 10 500 python bench.py --workload $W --steps 10 --warmup 2 > gpurun_out/r3_bench_$W.json 2> gpurun_out/r3_bench_$W.err || { echo "bench $W failed"; tail -5 gpurun_out/r3_bench_$W.err; exit 1; }
  python - $W <<'PY'
import json, sys
d = json.load(open('gpurun_out/r3_bench_%s.json' % sys.argv[1]))
r = d['roofline']
print(sys.argv[1], 'value %.4g' % d['value'], 'ms %.3f' % d['ms_per_step'], r['kernel'], 'frac %.3f' % r['frac'], 'traffic', r['traffic'] and '%.3g' % r['traffic'], 'parity', d['parity']['vs_det_oracle'] if d['parity'] else None,
      {k: v['avg_us'] for k, v in d['kernels'].items() if v['share_of_kernel_time'] > 0.03}, (d.get('unfused_round') or {}).get('ms_per_step'))
PY
done
