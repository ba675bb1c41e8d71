#!/bin/bash
# duration of every sweep / backward / forward launch of one strategy run, in launch order (rocprofv3 kernel trace)
W=${1:-c3}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/bylaunch_$W
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bylaunch_$W -o t -- python3 tools/exp/one_run.py $W > gpurun_out/bylaunch_$W.log 2>&1
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/bylaunch_$W/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        k="sweep" if "sweep" in n else "backward" if "backward_" in n else "forward" if ("ForwardBody" in n or "forward_quad" in n) else "commit" if "CommitBody" in n else "select" if "SelectBody" in n else None
        if k: rows.append((int(r["Start_Timestamp"]), k, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, int(r["Grid_Size_X"])))
rows.sort()
import collections
by=collections.defaultdict(list)
for _,k,us,g in rows: by[k].append(us)
for k,v in by.items(): print(k, " ".join("%.0f"%x for x in v))
PY
