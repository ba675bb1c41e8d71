#!/bin/bash
# fast kernel-iteration loop on the GPU box: the core parity tests, a short bench line, and the SQ instruction / LDS-conflict counters
# usage: tools/exp/quick_check.sh <tag> [workload]
TAG=${1:-x}; W=${2:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single_track_h100 or staged or lqr_identity or pendulum or edge_cases or perturbed or config2 or bicycle6_trust or constrained" > gpurun_out/qc_${TAG}_tests.txt 2>&1
tail -2 gpurun_out/qc_${TAG}_tests.txt
timeout -k 10 300 python bench.py --workload $W --no-cpu-baseline --no-parity > gpurun_out/qc_${TAG}_bench.json 2> gpurun_out/qc_${TAG}_bench.err
python - <<PY
import json
j=json.loads(open("gpurun_out/qc_${TAG}_bench.json").read())
print("value %.4g  ms/step %.3f  frac %.3f"%(j["value"], j["ms_per_step"], j["roofline"]["frac"]))
print({k:v["avg_us"] for k,v in j["kernels"].items()})
if j.get("also_config2"): print("c2", j["also_config2"]["ms_per_step"], j["also_config2"]["kernels_avg_us"])
PY
rm -rf gpurun_out/qc_${TAG}_sq
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d gpurun_out/qc_${TAG}_sq -o p -- python3 tools/exp/one_run.py $W > gpurun_out/qc_${TAG}_sq.log 2>&1
python - <<PY
import csv,glob,collections
t=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/qc_${TAG}_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="backward" if "backward_tile" in n else "forward" if "ForwardBody" in n else "sweep" if "sweep" in n else "trust" if "TrustRegion" in n else None
        if k: t[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,c in t.items():
    w=c.get("SQ_WAVES",1) or 1
    print(k, {n: round(v/w,1) for n,v in c.items()})
PY
