"""Turn the rocprofv3 output of tools/collect_profiles.sh into the small committed summaries under profiles/:
  <tag>_bench_<W>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of bench.py (per-kernel calls / average ns)
  <tag>_bench_<W>_under_rocprof.json the bench line printed in that run (its roofline.avg_launch_us is held against the csv)
  <tag>_pmc_traffic_<W>.json         FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE per kernel over one strategy run vs algorithmic bytes
  <tag>_sq_counters_<W>.json         SQ counters per kernel over one strategy run + derived issue / wait / LDS-conflict percentages
usage: summarise_profiles.py <workload> <tag> <dir>"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, TAG, D = sys.argv[1], sys.argv[2], sys.argv[3]
PROF = os.path.join(ROOT, "profiles")
os.makedirs(PROF, exist_ok=True)

KERNELS = (("linearise_riccati", ("mas_linearise_riccati_kernel",)), ("sweep", ("mas_sweep_kernel", "SweepBody")), ("central_backward", ()),
           ("backward", ("mas_backward_tile_kernel", "mas_backward_row_kernel", "BackwardBody")), ("forward", ("ForwardBody",)), ("select", ("SelectBody",)),
           ("commit", ("CommitBody", "CommitSlotBody")), ("trust_epilogue", ("mas_trust_epilogue_kernel",)), ("trust_reroll", ("TrustRerollBody",)), ("trust_norm", ("TrustNormBody",)), ("trust_copy", ("TrustCopyBody",)),
           ("trust_region", ("TrustRegion",)), ("rollout_cost", ("RolloutCostBody",)))


def classify(name):
    head = name.split("(")[0]
    # <Model, 16, true, WPB> (LDS-tile form) / <Model, true, WPB> (registers + DPP form): the shared ladder of the centralised strategy
    if ("mas_backward_tile_kernel" in head or "mas_backward_row_kernel" in head) and ", true" in head:
        return "central_backward"
    if "mas_backward_row_pipelined_kernel" in head:      # the shared ladder behind a speculative step
        return "central_backward"
    for key, pats in KERNELS:
        if any(p in name for p in pats):
            return key
    return None


def commit_id():
    if os.environ.get("GRAFT_COMMIT"):
        return os.environ["GRAFT_COMMIT"]
    try:
        return subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        return os.environ.get("GRAFT_COMMIT", "working tree at collection time")


def one(pattern):
    hits = glob.glob(os.path.join(D, pattern), recursive=True)
    return hits[0] if hits else None


def counters(csv_path):
    """{kernel: {counter: sum over dispatches}}, {kernel: dispatches}, {kernel: {counter: value of the first dispatch}}, {kernel: total ns}"""
    tot, n, first, ns = {}, {}, {}, {}
    seen = set()
    with open(csv_path) as f:
        for row in csv.DictReader(f):
            k = classify(row["Kernel_Name"])
            if not k:
                continue
            c, v = row["Counter_Name"], float(row["Counter_Value"])
            tot.setdefault(k, {}).setdefault(c, 0.0)
            tot[k][c] += v
            first.setdefault(k, {}).setdefault(c, v)
            key = (row["Dispatch_Id"], k)
            if key not in seen:
                seen.add(key)
                n[k] = n.get(k, 0) + 1
                ns[k] = ns.get(k, 0) + int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return tot, n, first, ns


sys.path.insert(0, ROOT)
from multi_agent_solver_amd.source_hash import source_hash  # noqa: E402

_stamp = os.path.join(D, "source_hash.txt")      # written on the GPU box at collection time; the hash of the sources at summarise time only without it
meta = {"workload": W, "commit": commit_id(), "source_hash": (open(_stamp).read().strip() if os.path.exists(_stamp) else source_hash()), "collected_by": "tools/collect_profiles.sh " + W + " " + TAG}
alg = json.load(open(os.path.join(D, "algorithmic.json")))

# ---- 1. kernel stats of the bench run
stats = one("stats/**/*kernel_stats.csv")
if stats:
    shutil.copy(stats, os.path.join(PROF, f"{TAG}_bench_{W}_kernel_stats.csv"))
bj = os.path.join(D, "bench_under_rocprof.json")
if os.path.exists(bj):
    lines = [l for l in open(bj).read().splitlines() if l.startswith("{")]
    if lines:
        j = json.loads(lines[-1])
        r = j.get("roofline") or {}
        kname = r.get("kernel", "mas_sweep_kernel")
        j["_note"] = ("bench.py run UNDER rocprofv3 --kernel-trace --stats (profiler overhead included in `value`); its roofline.avg_launch_us is the HIP-event average to hold "
                      "against AverageNs of " + kname + " in the kernel_stats.csv of the same run")
        j["source_hash"] = meta["source_hash"]
        if stats:
            with open(stats) as f:
                for row in csv.DictReader(f):
                    if kname in row["Name"]:
                        j["_rocprof_roofline_kernel"] = {"kernel": kname, "calls": int(row["Calls"]), "average_us": float(row["AverageNs"]) / 1e3, "total_ms": float(row["TotalDurationNs"]) / 1e6}
                        if r.get("algorithmic_bytes_per_launch") and r.get("launches"):
                            # bench's instrumented pass and the whole profiled process share the same kernels; rocprof averages over
                            # warm-up, timed and instrumented passes alike, so compare average durations, not totals
                            a = r["algorithmic_bytes_per_launch"] / (float(row["AverageNs"]) * 1e-9) / 1e9
                            j["_rocprof_roofline_kernel"]["achieved_GBs_at_rocprof_average"] = a
                            j["_rocprof_roofline_kernel"]["frac_at_rocprof_average"] = a / 8000.0
                        break
        json.dump(j, open(os.path.join(PROF, f"{TAG}_bench_{W}_under_rocprof.json"), "w"), indent=1)

# ---- 2. HBM traffic
f_csv, w_csv = one("pmc_FETCH_SIZE/**/*counter_collection.csv"), one("pmc_WRITE_SIZE/**/*counter_collection.csv")
if f_csv and w_csv:
    ft, fn, ff, _ = counters(f_csv)
    wt, wn, wf, _ = counters(w_csv)
    out = dict(meta)
    out["_what"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over ONE strategy run (tools/one_run.py); counters are in KiB; "
                    "FETCH_SIZE is doubled (gfx950 reports 1/2 of a coalesced streaming read, MI355X_MICROARCH.md), WRITE_SIZE taken as is; "
                    "algorithmic bytes: the same run's per-kernel totals (SURVEY.md 8(d) per agent x active agents per launch)")
    out["kernels"] = {}
    for k in sorted(set(ft) | set(wt)):
        fetch = 2.0 * 1024.0 * ft.get(k, {}).get("FETCH_SIZE", 0.0)
        write = 1024.0 * wt.get(k, {}).get("WRITE_SIZE", 0.0)
        a = alg["kernels"].get(k, {})
        e = {"dispatches": fn.get(k, wn.get(k, 0)), "fetch_bytes_corrected": fetch, "write_bytes": write, "measured_bytes": fetch + write,
             "first_dispatch": {"fetch_bytes_corrected": 2.0 * 1024.0 * ff.get(k, {}).get("FETCH_SIZE", 0.0), "write_bytes": 1024.0 * wf.get(k, {}).get("WRITE_SIZE", 0.0)}}
        if a.get("algorithmic_bytes"):
            e["algorithmic_bytes"] = a["algorithmic_bytes"]
            e["algorithmic_launches"] = a["launches"]
            e["measured_over_algorithmic"] = (fetch + write) / a["algorithmic_bytes"]
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(PROF, f"{TAG}_pmc_traffic_{W}.json"), "w"), indent=1)

# ---- 3. SQ counters
sq = {}
disp, dur = {}, {}
for p in sorted(glob.glob(os.path.join(D, "sq*/**/*counter_collection.csv"), recursive=True)):
    t, n, _, ns = counters(p)
    for k, cs in t.items():
        sq.setdefault(k, {}).update(cs)
        disp[k] = n[k]
        dur[k] = ns[k]
if sq:
    out = dict(meta)
    out["_what"] = ("rocprofv3 --pmc SQ counter sets (four passes) over ONE strategy run (tools/one_run.py), summed over every dispatch of a kernel; "
                    "SQ cycle counters count quad-cycles per wavefront.  derived: issue = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES, valu_issue = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, "
                    "wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES, lds_bank_conflict = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS, cycles_per_valu = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU, "
                    "chip_valu_busy = 4 * SQ_ACTIVE_INST_VALU / (kernel ns under the profiler * 2.4 GHz * 1024 SIMDs) -- the FP64 VALU issue slots of the whole chip the kernel fills (no MFMA is used)")
    out["kernels"] = {}
    for k, c in sq.items():
        e = {"dispatches": disp[k], "kernel_ns_under_profiler": dur[k], "counters": c}
        wc = c.get("SQ_WAVE_CYCLES")
        d = {}
        if wc:
            for name, key in (("issue_pct", "SQ_ACTIVE_INST_ANY"), ("valu_issue_pct", "SQ_ACTIVE_INST_VALU"), ("wait_pct", "SQ_WAIT_ANY"), ("lds_issue_pct", "SQ_ACTIVE_INST_LDS")):
                if key in c:
                    d[name] = round(100.0 * c[key] / wc, 2)
        if c.get("SQ_ACTIVE_INST_LDS"):
            d["lds_bank_conflict_pct_of_lds_cycles"] = round(100.0 * c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_ACTIVE_INST_LDS"], 2)
        if c.get("SQ_INSTS_VALU"):
            d["cycles_per_valu_instruction"] = round(4.0 * c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_INSTS_VALU"], 2)
        if c.get("SQ_WAVES"):
            for key in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_LDS_BANK_CONFLICT"):
                if key in c:
                    d[key.lower() + "_per_wavefront"] = round(c[key] / c["SQ_WAVES"], 1)
        if dur.get(k) and "SQ_ACTIVE_INST_VALU" in c:
            d["chip_valu_busy_pct"] = round(100.0 * 4.0 * c["SQ_ACTIVE_INST_VALU"] / (dur[k] * 2.4 * 1024), 2)
        e["derived"] = d
        out["kernels"][k] = e
    json.dump(out, open(os.path.join(PROF, f"{TAG}_sq_counters_{W}.json"), "w"), indent=1)
print("summaries written for", W, "->", sorted(os.path.basename(p) for p in glob.glob(os.path.join(PROF, f"{TAG}_*_{W}*"))))
