#!/bin/bash
# SQ_LDS_BANK_CONFLICT of the backward kernel for each libmas_hip variant (tools/exp/build_variants.sh)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  rm -rf gpurun_out/cf_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/cf_$v -o p -- python3 tools/exp/ab_variants.py c3 $v > gpurun_out/cf_$v.log 2>&1
  python3 - <<PY
import csv,glob,collections
t=collections.defaultdict(float)
for f in glob.glob("gpurun_out/cf_$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "backward_tile" in r["Kernel_Name"]: t[r["Counter_Name"]]+=float(r["Counter_Value"])
w=t.get("SQ_WAVES",1) or 1
print("$v", {n: round(x/w,1) for n,x in t.items()})
PY
done
