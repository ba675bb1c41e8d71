#!/bin/bash
# after tools/round_end.sh through gpurun: the merged gpurun_out/ -> profiles/ (profiles/ written on the GPU box does not travel back)
set -e
cd "$(dirname "$0")/.."
for W in c3 c2 c5 c4; do python tools/summarise_profiles.py $W r4 gpurun_out/prof_r4_$W | tail -n 1; done
python - <<'PY'
import glob, json, shutil
for w in ("c3", "c4", "c5"):
    j = json.loads([l for l in open(f"gpurun_out/r4_bench_{w}.json") if l.startswith("{")][-1])
    r = j["roofline"]
    print(w, round(j["value"]), "per s,", round(j["ms_per_step"], 3), "ms, frac", round(r["frac"], 4), "traffic", r.get("traffic"), "parity", (j.get("parity") or {}).get("vs_det_oracle"))
    assert r.get("traffic"), "the bench line did not find counters of the current sources"
    json.dump(j, open(f"profiles/r4_bench_{w}.json", "w"), indent=1)
shutil.copy("profiles/r4_bench_c3.json", "profiles/r4_bench_c3_with_literal_c2.json")
shutil.copy(glob.glob("gpurun_out/dense_stats/**/*kernel_stats.csv", recursive=True)[0], "profiles/r4_literal_c2_kernel_stats.csv")
log = open("gpurun_out/trust_c3.log").read()
json.dump(json.loads(log[log.index("{"):]), open("profiles/r4_trust_region_c3_sized_timing.json", "w"), indent=1)
PY
tail -n 2 gpurun_out/r4_gpu_tests.txt
