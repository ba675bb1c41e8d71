#!/bin/bash
# duration of every launch of one strategy run, in launch order, and the gaps between consecutive launches (rocprofv3 kernel trace)
#   tools/by_launch.sh [c3|c2|c4|c5] [agents] [--option key=value ...]
W=${1:-c3}; N=${2:-}; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/bylaunch_$W
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/bylaunch_$W -o t -- python3 tools/one_run.py $W $N "$@" > gpurun_out/bylaunch_$W.log 2>&1
python3 - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("gpurun_out/bylaunch_$W/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = ("fused" if "linearise_riccati" in n else "prologue" if "round_prologue" in n else "trust" if ("trust_" in n or "Trust" in n) else "sweep" if "sweep" in n else "backward" if "backward_" in n else
             "forward" if ("ForwardBody" in n or "forward_quad" in n) else "commit" if "Commit" in n else "select" if "SelectBody" in n else
             "rollout" if "rollout" in n.lower() else "other")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
by = collections.defaultdict(list)
for s, e, k in rows:
    by[k].append((e - s) / 1e3)
for k, v in by.items():
    print(k, len(v), "launches:", " ".join("%.0f" % x for x in v))
gaps = [(rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1)]
print("gaps_us:", " ".join("%.1f" % g for g in gaps))
# the strategy run proper: from the first fused / sweep launch to the last launch
first = next((i for i, r in enumerate(rows) if r[2] in ("fused", "sweep")), 0)
span = (rows[-1][1] - rows[first][0]) / 1e3
busy = sum((e - s) / 1e3 for s, e, _ in rows[first:])
print("span_us %.0f kernel_us %.0f gap_us %.0f launches %d" % (span, busy, span - busy, len(rows) - first))
PY
