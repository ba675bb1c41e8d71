"""bench.py prints ONE JSON line with the contract's fields (runs on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _check_line(j, n_gpus):
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict)):
        assert isinstance(j[key], typ), key
    assert j["n_gpus"] == n_gpus and j["scaling"] == "weak" and j["dtype"] == "f64" and j["vs_baseline"] is None
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    # the contract's figure (algorithmic bytes / kernel time against the HBM peak) ...
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    # ... and what actually binds the kernel (round 4): the fused kernel keeps its tiles in LDS and follows the recursion's FP64 chain
    assert r["bound"] == ("fp64_issue" if r["kernel"] == "mas_linearise_riccati_kernel" else "hbm") and "frac_is" in r
    f = r["fp64"]
    assert f["unit"] == "TFLOP/s" and f["peak"] == 78.6 and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-12 and 0.0 < f["frac"] < 1.0
    assert "issue_fill" in f and "issue_fill_source" in f           # a number only with SQ counters collected from these very sources, else null + reason
    assert (f["issue_fill"] is None) or (0.0 < f["issue_fill"] < 1.0)
    ls = r["line_search"]
    assert ls["bound"] == "fp64_issue" and 0.0 < ls["fp64"]["frac"] < 1.0
    rf = r["round_fill"]
    assert ("value" in rf) and ((rf["value"] is None and "reason" in rf) or (0.0 < rf["value"] < 1.0))
    assert "hbm_measured" in r and ((r["hbm_measured"] is None) == (r["traffic"] is None))
    assert j["value"] > 0 and j["ms_per_step"] > 0


def test_single_gpu_line_has_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--agents-per-gpu", "512"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    _check_line(j, 1)
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    assert j["also_config2"]["value"] > 0
    # parity block: exact against the pure-IEEE-trig oracle, measured deviation against the glibc-trig one
    p = j["parity"]
    assert p["vs_det_oracle"] == "exact" and p["compared_agents"] == 512
    assert set(p["vs_libm_oracle"]) >= {"median_rel", "max_rel", "total_rel", "frac_le_1e-5", "north_star_tolerance_1e-5_met"}
    assert j["also_config2"]["parity"]["vs_det_oracle"] == "exact"
    # round 4: BASELINE configs[1] also under the reference's LITERAL centralised algorithm (chip-wide launches), held against the oracle's committed vectors
    lit = j["also_config2"]["literal_stacked_solve"]
    assert lit["agents"] == 256 and lit["vs_oracle_structured_restatement"] == "exact" and lit["iterations"] == 8 and 0.0 < lit["seconds_per_solve"] < 60.0
    assert lit["block_diagonal_vs_literal_total_rel"] > 0.0 and lit["unit"] == "agent-iterations/s"
    # round 3: the four-state models run two-kernel rounds; the roofline kernel is the fused linearise + Riccati kernel (sweep +
    # backward bytes of SURVEY 8(d)), and the sweep as a kernel of its own is measured in the same run with the five-kernel round
    r = j["roofline"]
    assert r["kernel"] == "mas_linearise_riccati_kernel" and r["algorithmic_bytes_per_agent"] == 51392 + 54432
    assert r["fp64"]["flops_per_agent_launch"] == 100 * (12 * 150 + 118 * 25) + 100 * 802      # 4.8e5 (linearisation) + ~700-800 per backward step
    u = j["unfused_round"]
    assert u["ms_per_step"] > 0 and u["speedup_of_the_fused_round"] > 0.9 and r_has_both_fracs(u["roofline_sweep"])
    assert "traffic" in r and "traffic_source" in r          # a number only with counters collected from these very sources, else null + reason
    pt = p["population_tolerance"]
    assert pt["met"] is True and pt["agents"] == 512 and set(pt["worst_over_variants"]) == set(pt["bounds"])
    cb = j["cpu_baseline"]
    assert cb["repeats"] >= 3 and cb["min"] <= cb["value"] <= cb["max"]


def r_has_both_fracs(r):
    return r["kernel"] == "mas_sweep_kernel" and "inkernel_span" in r and 0.0 < r["inkernel_span"]["frac"] < 1.0 and r["algorithmic_bytes_per_agent"] == 51392


@pytest.mark.parametrize("workload,agents,bytes_per_agent", [("c4", 256, 38592 + 8 * 50 * 88 + 32 + 8 * 50 * 20), ("c5", 128, 185984)])
def test_config4_and_config5_are_bench_workloads(workload, agents, bytes_per_agent):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps", "1", "--warmup", "1",
                          "--agents-per-gpu", str(agents)], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    _check_line(j, 1)
    assert j["roofline"]["algorithmic_bytes_per_agent"] == bytes_per_agent
    assert j["parity"]["vs_det_oracle"] == "exact"
    assert j["cpu_baseline"]["value"] > 0


def test_rccl_path_initialises_with_one_rank():
    """torch.distributed over RCCL (backend nccl) with a single rank: the code path the driver uses for N > 1"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29544", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0",
                          "--agents-per-gpu", "256", "--force-dist", "--no-cpu-baseline", "--no-also", "--gather-solution"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    _check_line(j, 1)
    g = j["solution_all_gather"]            # the optional publish all-gather went through RCCL on device tensors
    assert g["ms"] > 0 and g["bytes_received_per_rank"] == 256 * (101 * 4 + 100 * 2 + 1) * 8


_RCCL_LINESEARCH = r"""
import math, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
from multi_agent_solver_amd.dist import make_cost_exchange
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
N, T = 64, 40
def run(exchange):
    with BatchSolver(1, N, T, 0.5) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(single_track_x0(N)); s.set_controls(None)
        if exchange: s.set_cost_exchange(make_cost_exchange(N))
        s.run_line_search(3)
        return s.solution()
a, b = run(False), run(True)
assert np.array_equal(a["cost"], b["cost"]) and np.array_equal(a["X"], b["X"]) and np.array_equal(a["iters"], b["iters"])
dist.destroy_process_group()
print("RCCL_LINESEARCH_OK")
"""


def test_line_search_cost_exchange_over_rccl_matches_the_local_sum():
    """the one collective of the path (total cost of the line-search strategy, nash.hpp:45-49) through RCCL, one rank"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29546")
    out = subprocess.run([sys.executable, "-c", _RCCL_LINESEARCH % {"root": ROOT}], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0 and "RCCL_LINESEARCH_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-2000:])
