#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X batched iLQR path.

Metric (BASELINE.json): iLQR iterations/sec x agents on the single-track H=100 workload.
A "step" is one complete strategy run (the span the reference times around mas::solve,
examples/multi_agent_single_track.cpp:123-126) over one batch of synthetic agents whose inputs are
already resident in HBM:  restore initial (x0, U0=0) -> strategy -> results stay on the device.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2|c4|c5] [--agents-per-gpu A]

Workloads (BASELINE.json `configs`; per-GPU shard sizes, weak scaling over ranks):
  c3  configs[2]  single-track, sequential (Nash), 4096 agents per GPU, H=100           <- default, the headline
  c2  configs[1]  single-track, centralized, 256 agents, H=100 (one GPU; measured as `also_config2` in the c3 run)
  c4  configs[3]  identity LQR, sequential, 4096 agents per GPU (32768 on 8), H=50, + RCCL all-gather of the Solution
  c5  configs[4]  bicycle6 (nx=6), trust-region, 8192 agents per GPU (65536 on 8), H=200

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
one rank per GPU, agents sharded by id range, no data-path collective (agents are
decoupled); RCCL is used only to take the MAX of the ranks' times and the SUM of their iteration counts -- and, for
c4, for the publish step of the Nash strategies (nash.hpp:66-71) timed separately as `solution_all_gather`.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline     -- the kernel that holds the linearisation sweep north_star names: for the four-state models the fused linearise +
                  Riccati kernel (its tiles never reach HBM), else the sweep kernel itself; algorithmic bytes
                  over the kernel's launch durations measured in this run with HIP events bound to each dispatch
  unfused_round-- (c3, c4) the same steps with the five-kernel round: the gain of the fusion, and the sweep's own HBM roofline
  utilisation  -- backward / forward pass: issue %, wait %, LDS bank-conflict % (SQ counters cannot be read in-process:
                  taken from the committed rocprofv3 --pmc passes under profiles/, source stated)
  parity       -- the timed workload's result against the CPU oracle: exact vs its pure-IEEE-trig mode (the parity
                  definition) and the measured deviation vs its glibc-trig mode
  cpu_baseline -- the CPU restatement of the reference algorithm (oracle, OpenMP over agents as nash.hpp:59),
                  timed on this host on a bounded sample of the same workload
  kernels      -- per-kernel time shares of the timed region
  also_config2 -- the same measurement for BASELINE configs[1] (256 agents, centralized), N=1 only
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_PEAK_TFLOPS = 78.6 # MI355X FP64 vector peak (256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz), SURVEY.md 8(d)
PROFILE_ROUND = "r4"    # profiles/<round>_pmc_traffic_<workload>.json, profiles/<round>_sq_counters_<workload>.json


def backward_flops_per_step(nx, nu):
    """FP64 operations of one backward Riccati step (ilqr.hpp:115-119,183-192) counted on the dense products: SURVEY.md 8(d) quotes ~700 at (4, 2)"""
    return (4 * nx ** 3 + 4 * nx * nx * nu + 2 * nx * nu * nu + 2 * nx * nx + 2 * nx * nu      # A^T V A, B^T V A, B^T V B, q_x, q_u
            + nu ** 3 // 3 + nu ** 3 + 4 * nu * nu * nx + 2 * nu * nu                         # LLT, its inverse, K, k, K^T q_uu
            + 6 * nx * nu + 6 * nx * nx * nu + nx * nx)                                        # v_x, v_xx, symmetrisation


from multi_agent_solver_amd.dist import shard_range  # noqa: E402  contiguous ranges of the id-sorted block list, one per rank

M_SINGLE_TRACK, M_LQR, M_BICYCLE6 = 1, 2, 5

WORKLOADS = {
    "c3": dict(name="multi_agent_single_track --solver ilqr --strategy sequential, 4096 agents per GPU, H=100 (BASELINE configs[2])",
               model=M_SINGLE_TRACK, nx=4, nu=2, agents=4096, T=100, dt=0.5, strategy="sequential", max_outer=10, max_iterations=100, tolerance=1e-5,
               sweep_bytes_per_agent=51392, backward_bytes_per_agent=54432,
               # per (agent, t) 12 dynamics evaluations x ~150 + 118 cost evaluations x ~25 FP64 operations = 4.8e5 per agent-iteration at T = 100
               sweep_flops_per_agent=100 * (12 * 150 + 118 * 25), forward_flops_per_agent_trial=100 * (4 * 150 + 25 + 30)),
    "c2": dict(name="multi_agent_single_track --solver ilqr --strategy centralized, 256 agents, H=100 (BASELINE configs[1])",
               model=M_SINGLE_TRACK, nx=4, nu=2, agents=256, T=100, dt=0.5, strategy="centralized", max_outer=1, max_iterations=100, tolerance=1e-5,
               sweep_bytes_per_agent=51392, backward_bytes_per_agent=54432,
               sweep_flops_per_agent=100 * (12 * 150 + 118 * 25), forward_flops_per_agent_trial=100 * (4 * 150 + 25 + 30)),
    "c4": dict(name="multi_agent_lqr --solver ilqr --strategy sequential, 4096 agents per GPU (32768 on 8), H=50 (BASELINE configs[3])",
               model=M_LQR, nx=4, nu=4, agents=4096, T=50, dt=0.1, strategy="sequential", max_outer=10, max_iterations=100, tolerance=1e-5,
               sweep_bytes_per_agent=38592, backward_bytes_per_agent=8 * 50 * 88 + 32 + 8 * 50 * 20,
               sweep_flops_per_agent=50 * 16, forward_flops_per_agent_trial=50 * (4 * 8 + 16 + 40)),       # analytic derivatives (2x, 2u), x' = x + u
    "c5": dict(name="multi_agent_single_track --solver ilqr --strategy trustregion, nx=6 bicycle model, 8192 agents per GPU (65536 on 8), H=200 (BASELINE configs[4])",
               model=M_BICYCLE6, nx=6, nu=2, agents=8192, T=200, dt=0.25, strategy="trustregion", max_outer=10, max_iterations=100, tolerance=1e-5,
               sweep_bytes_per_agent=185984, backward_bytes_per_agent=8 * 200 * 108 + 48 + 8 * 200 * 14,
               sweep_flops_per_agent=200 * (16 * 160 + 216 * 30), forward_flops_per_agent_trial=200 * (4 * 160 + 30 + 40)),   # 16 dynamics + 216 cost evaluations per (agent, t)
}


def initial_states(w, n_local, first, n_total):
    """the reference's closed-form generators; rows are agents first .. first+n_local-1"""
    from multi_agent_solver_amd.batch import single_track_x0, lqr_x0, bicycle6_x0
    if w["model"] == M_SINGLE_TRACK:
        return single_track_x0(n_local, first=first, total=n_total)
    if w["model"] == M_LQR:
        return lqr_x0(n_local)
    return bicycle6_x0(n_local, first=first, total=n_total)


def make_solver(w, n_local, first, n_total, device):
    from multi_agent_solver_amd import BatchSolver
    s = BatchSolver(w["model"], n_local, w["T"], w["dt"], device=device)
    s.set_params({"max_iterations": w["max_iterations"], "tolerance": w["tolerance"], "max_ms": math.inf})
    s.set_initial_states(initial_states(w, n_local, first, n_total))
    s.set_controls(None)
    s.initialize_problem()
    s.save_initial()
    return s


def one_step(s, w):
    s.restore_initial()
    s.run(w["strategy"], w["max_outer"])


def measure(w, n_local, first, n_total, device, steps, warmup, sync, barrier, after=None, unfused_reference=False):
    """Timed region: K steps with per-kernel instrumentation OFF (HIP events around every dispatch cost ~11 % of
    wall time on this latency-bound loop).  The kernel times / roofline come from a second pass of the same K
    steps, same process, same resident inputs, with the instrumentation ON."""
    s = make_solver(w, n_local, first, n_total, device)
    for _ in range(warmup):
        one_step(s, w)
    s.reset_stats()
    barrier(); sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step(s, w)
    sync(); barrier()
    elapsed = time.perf_counter() - t0
    st = s.stats()
    s.reset_stats()
    s.set_profiling(True)
    sync()
    t1 = time.perf_counter()
    for _ in range(steps):
        one_step(s, w)
    sync()
    st["profiled_pass_elapsed"] = time.perf_counter() - t1
    kt = s.kernel_times()
    s.set_profiling(False)
    if unfused_reference:
        # the same K steps with the five-kernel round (mas_hip_set_option "fused" 0; bit-identical results): what the fused round
        # buys, measured in this very run, and the standalone linearisation sweep -- the HBM-bound kernel north_star names -- alone
        s.set_option("fused", 0)
        one_step(s, w)
        s.reset_stats()
        sync(); t2 = time.perf_counter()
        for _ in range(steps):
            one_step(s, w)
        sync()
        st["unfused_elapsed"] = time.perf_counter() - t2
        st["unfused_agent_iterations"] = s.stats()["agent_iterations"]
        s.reset_stats()
        s.set_profiling(True)
        for _ in range(steps):
            one_step(s, w)
        sync()
        st["unfused_kernels"] = s.kernel_times()
        s.set_profiling(False)
        s.set_option("fused", 1)
        one_step(s, w)                       # the solution read below is the product path's
    if after is not None:
        st["after"] = after(s)
    sol = s.solution()
    s.close()
    return elapsed, st, kt, sol


def literal_stacked_solve_block(w, device, block_diagonal_total):
    """BASELINE configs[1] under the reference's LITERAL algorithm -- one iLQR on the stacked OCP, every cost derivative a finite difference of the
    SUMMED cost (centralized.hpp:18-38, multi_agent_problem.hpp:94-122) -- on the device (mas_hip_run_centralized_dense, chip-wide launches:
    csrc/mas_dense_chip.hpp), timed, and held against the CPU oracle's structured restatement of the same solve through the committed vectors
    tests/golden/centralized_dense_large.json (tools/make_dense_goldens.py: ten minutes of eight cores at 256 agents)."""
    import hashlib
    import numpy as np
    s = make_solver(w, w["agents"], 0, w["agents"], device)
    times = []
    for _ in range(2):                                   # (the first run allocates the stacked problem's 3.5 GB)
        s.restore_initial(); s.reset_stats()
        t0 = time.perf_counter(); s.run("centralized_dense", 1); times.append(time.perf_counter() - t0)
    st, sol, total = s.stats(), s.solution(), s.total_cost()
    s.set_profiling(True); s.restore_initial(); s.run("centralized_dense", 1); kt = s.kernel_times(); s.close()
    out = {"what": "the reference's literal centralised algorithm at this size: 4 (N nx)^2 evaluations of the N-agent cost per time step, dense n = N nx / m = N nu "
                   "Riccati recursion (LLT of 512 x 512 with the x10 ladder, its inverse, K, V) -- the same numbers by exact re-use only (DESIGN.md section 4)",
           "agents": w["agents"], "seconds_per_solve": times[1], "first_solve_seconds": times[0], "iterations": int(st["agent_iterations"] // w["agents"]),
           "value": st["agent_iterations"] / times[1], "unit": "agent-iterations/s", "total_cost": total,
           "block_diagonal_reading_total_cost": block_diagonal_total,
           "block_diagonal_vs_literal_total_rel": abs(block_diagonal_total - total) / abs(total),
           "kernels_ms": {n: round(k["total_ms"], 2) for n, k in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])[:8]}}
    path = os.path.join(ROOT, "tests", "golden", "centralized_dense_large.json")
    g = json.load(open(path))["cases"].get(str(w["agents"])) if os.path.exists(path) else None
    if g is not None and w["T"] == g["horizon"]:
        same = (float(total).hex() == g["total_cost"] and [float(v).hex() for v in sol["cost"]] == g["cost"] and
                hashlib.sha256(np.ascontiguousarray(sol["X"]).tobytes()).hexdigest() == g["sha256_X"] and
                hashlib.sha256(np.ascontiguousarray(sol["U"]).tobytes()).hexdigest() == g["sha256_U"] and out["iterations"] == g["iterations"])
        out["vs_oracle_structured_restatement"] = "exact" if same else "DIFFERENT"
        out["oracle_seconds_for_the_same_solve"] = g["oracle_seconds"]
    return out


def oracle_run(w, x0, trig, threads=0):
    from oracle import oracle as O
    O.set_trig_mode(trig)
    try:
        return O.solve_batch(w["model"], x0, None, w["T"], w["dt"], strategy=w["strategy"], max_outer=w["max_outer"],
                             max_iterations=w["max_iterations"], tolerance=w["tolerance"], threads=threads)
    finally:
        O.set_trig_mode("libm")


def cpu_baseline(w, n_total):
    """CPU restatement of the reference algorithm (oracle/), OpenMP over agents like nash.hpp:59.
    Bounded sample: every k-th agent of the workload, whole strategy."""
    from oracle import oracle as O
    threads = O.max_threads()
    n = min(n_total, 64 * threads)
    stride = max(1, n_total // n)
    x0 = initial_states(w, n_total, 0, n_total)[::stride][:n]
    rates, el = [], 0.0
    while (el < 8.0 or len(rates) < 3) and len(rates) < 20:            # bounded: about 10 s of CPU work, at least three repeats
        t0 = time.perf_counter()
        r = oracle_run(w, x0, "libm", threads)      # std::sin/cos/tan, as the reference binary calls them
        dt = time.perf_counter() - t0
        el += dt
        rates.append(int(r["stats"]["total_iterations"]) * (x0.shape[0] if w["strategy"] == "centralized" else 1) / dt)
    rates.sort()
    median = rates[len(rates) // 2] if len(rates) % 2 else 0.5 * (rates[len(rates) // 2 - 1] + rates[len(rates) // 2])
    return dict(value=median, unit="agent-iterations/s", cores=threads, kind="port", statistic="median of the repeats", repeats=len(rates),
                min=rates[0], max=rates[-1], spread_rel=(rates[-1] - rates[0]) / median,
                sample=f"{x0.shape[0]} agents (every {stride}th of {n_total}), full '{w['strategy']}' strategy run repeated {len(rates)}x, {el:.1f} s total; "
                       f"CPU restatement of reference algorithm (oracle/, gcc -O2 -ffp-contract=off, OpenMP schedule(static) over agents as nash.hpp:59, glibc trig)")


def parity_block(w, sol, n_total, max_agents=4096):
    """The timed workload's final result against the CPU oracle, outside the timed region.
    vs_det_oracle : the oracle with the shared pure-IEEE sin/cos/tan (include/mas_dettrig.h) -- the parity definition of
                    this repository: "exact" means costs, trajectories, controls and iteration
                    counts are bit-identical for every compared agent.
    vs_libm_oracle: the same oracle with glibc's sin/cos/tan, i.e. what a glibc-linked reference binary computes.
                    north_star asks for <= 1e-5 relative cost error; the algorithm amplifies last-bit differences of the
                    trig functions chaotically (tests/test_oracle_sensitivity.py), so this is reported as measured."""
    import numpy as np
    x0_all = initial_states(w, n_total, 0, n_total)
    stride = max(1, -(-n_total // max_agents))
    idx = np.arange(0, n_total, stride)
    if w["strategy"] == "centralized":
        idx = np.arange(n_total)                      # agents share alpha / termination / ladder: the whole problem or nothing
    x0 = x0_all[idx]
    t0 = time.perf_counter()
    det = oracle_run(w, x0, "det")
    exact = (np.array_equal(sol["cost"][idx], det["cost"]) and np.array_equal(sol["X"][idx], det["X"]) and
             np.array_equal(sol["U"][idx], det["U"]) and sol["iters"][idx].tolist() == det["iters"].tolist())
    rel_det = np.abs(sol["cost"][idx] - det["cost"]) / np.abs(det["cost"])
    lib = oracle_run(w, x0, "libm")
    rel = np.abs(sol["cost"][idx] - lib["cost"]) / np.abs(lib["cost"])
    out = {
        "compared_agents": int(idx.size), "of": int(n_total), "sample": ("all agents" if stride == 1 or w["strategy"] == "centralized" else f"every {stride}th agent (agents are independent: a subset is re-solved alone by the oracle)"),
        "definition": "bit-exact against the CPU oracle in its pure-IEEE-trig mode (DESIGN.md section 2.1); the glibc-trig deviation is what north_star's <=1e-5 asks about",
        "vs_det_oracle": "exact" if exact else {"max_rel_cost": float(rel_det.max()), "iters_equal_frac": float(np.mean(sol["iters"][idx] == det["iters"]))},
        "vs_libm_oracle": {"median_rel": float(np.median(rel)), "max_rel": float(rel.max()),
                           "total_rel": float(abs(sol["cost"][idx].sum() - lib["cost"].sum()) / abs(lib["cost"].sum())),
                           "frac_le_1e-5": float(np.mean(rel <= 1e-5)), "iters_equal_frac": float(np.mean(sol["iters"][idx] == lib["iters"])),
                           "north_star_tolerance_1e-5_met": bool(rel.max() <= 1e-5)},
        "oracle_seconds": round(time.perf_counter() - t0, 2),
    }
    if w["model"] == M_SINGLE_TRACK and w["strategy"] == "sequential" and stride == 1:
        # the tolerance that can be true (tests/test_population_tolerance.py): the GPU result's population
        # statistics against every faithful variant of the CPU arithmetic, all agents
        from oracle import oracle as O
        from oracle.population import POPULATION_BOUNDS_512, POPULATION_BOUNDS_4096, VARIANTS, oracle_variant, population_deviation, population_statistics
        bounds = POPULATION_BOUNDS_4096 if idx.size >= 2048 else POPULATION_BOUNDS_512      # (a smaller population is noisier)
        mine = population_statistics({"cost": sol["cost"][idx], "iters": sol["iters"][idx], "status": sol["status"][idx]})
        worst, per = {k: 0.0 for k in bounds}, {}
        for name, kw in VARIANTS.items():
            dev = population_deviation(mine, population_statistics(oracle_variant(O, x0, T=w["T"], dt=w["dt"], **kw)))
            per[name] = {k: float(v) for k, v in dev.items()}
            for k, v in dev.items():
                worst[k] = max(worst[k], float(v))
        out["population_tolerance"] = {
            "statement": "total cost, mean iterations, converged fraction and the 10/25/50/75/90 % quantiles of the per-agent cost of the GPU result against the CPU "
                         "restatement under glibc trig / Eigen-style pairwise product sums / non-aliased symmetrisation / one-ulp inputs / all of them (the per-agent 1e-5 is "
                         "unattainable: one rounding anywhere moves single agents by per cents)",
            "agents": int(idx.size), "bounds": bounds, "worst_over_variants": worst, "per_variant": per,
            "met": bool(all(worst[k] <= bounds[k] for k in bounds))}
    if w["strategy"] == "centralized":
        # the reference's literal centralised algorithm (finite differences of the SUMMED cost) is computable up to 64 agents on the device (seconds beyond 16) and at any size by the oracle: its deviation
        # from the block-diagonal reading AT THIS HORIZON, next to its own deviation under one-ulp inputs (tests/test_centralized_readings.py)
        from oracle import oracle as O
        from oracle.population import centralized_reading_deviation
        O.set_trig_mode("det")
        try:
            curve = [centralized_reading_deviation(O, n, T=w["T"], dt=w["dt"]) for n in (4, 8, 16)]
        finally:
            O.set_trig_mode("libm")
        out["vs_reference_stacked_solve"] = {
            "statement": "NOT the same number: the reference finite-differences the summed cost of the stacked problem (centralized.hpp:18-38, multi_agent_problem.hpp:94-122), "
                         "whose Hessians carry FD noise in the cross-agent blocks; this path solves the block-diagonal reading (the literal solve is 4*(4N)^2 evaluations of "
                         "the N-agent cost per time step: on the device up to 64 agents, mas_hip_run_centralized_dense, bit-equal to the oracle's; profiles/r3_c2_readings_curve.json holds this curve to 64 agents).  At this horizon the two "
                         "differ by tens of per cent of the total cost -- and so does the literal solve from ITSELF when one input coordinate moves by one ulp: both stall on "
                         "the same finite-difference noise after a few iterations.  At T = 10 they agree to 5e-3 (tests/test_gpu_parity.py).",
            "curve_at_this_horizon": curve,
            "block_diagonal_vs_literal_total_rel_max": max(c["block_diagonal_vs_literal_total_rel"] for c in curve),
            "literal_under_one_ulp_inputs_total_rel_max": max(max(c["literal_under_one_ulp_inputs_total_rel"]) for c in curve)}
    return out


def _load_profile(kind, workload):
    """a committed counter summary, or (None, reason) when there is none or it was collected from OTHER kernel sources than
    the ones this library is built from (multi_agent_solver_amd/source_hash.py): stale counters are never quoted"""
    from multi_agent_solver_amd.source_hash import source_hash
    path = os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{kind}_{workload}.json")
    rel = os.path.relpath(path, ROOT)
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return None, f"{rel}: not collected"
    have, want = prof.get("source_hash"), source_hash()
    if have != want:
        return None, f"{rel}: collected from other kernel sources (source_hash {have} != {want}); re-run tools/collect_profiles.sh {workload}"
    return prof, rel


ROOFLINE_KERNELS = {
    # name in kernel_times -> (kernel symbol, what its algorithmic bytes are, what bounds it)
    "linearise_riccati": ("mas_linearise_riccati_kernel",
                          "SURVEY.md 8(d), linearisation sweep + backward pass: read 8[(nx+nu)T+nx] + write 8[T*ND+nx+nx^2] (the sweep), read 8[T*ND+nx] + write 8T(nu+nu*nx) (the recursion).  "
                          "The fused kernel keeps the derivative tiles in LDS (or, for analytic models, in registers): they never reach HBM, so `traffic` is a fraction of these bytes",
                          "the recursion's dependent FP64 chain (one wavefront per SIMD, DESIGN.md section 4); quoted against HBM as the contract prescribes"),
    "sweep": ("mas_sweep_kernel",
              "SURVEY.md 8(d): read 8[(nx+nu)T+nx] + write 8[T*ND+nx+nx^2]; the kernel itself moves 8(2nx+nx^2) B less per agent (x_T, terminal v_x/v_xx belong to the backward kernel)",
              "HBM (streaming store of the derivative tiles)"),
}


def _issue_fill(workload, name):
    """share of the chip's FP64 issue slots a kernel fills (SQ_ACTIVE_INST_VALU over all SIMDs' cycles of the kernel's launches), from the committed
    rocprofv3 --pmc SQ passes; (None, reason) without counters collected from exactly these kernel sources"""
    sq, src = _load_profile("sq_counters", workload)
    if not sq:
        return None, src
    k = sq.get("kernels", {}).get(name, {}).get("derived", {})
    if "chip_valu_busy_pct" not in k:
        return None, f"{src}: no such kernel in the profile"
    return k["chip_valu_busy_pct"] / 100.0, f"derived: {src} (SQ_ACTIVE_INST_VALU / (kernel time x 1024 SIMDs), source_hash {sq.get('source_hash')})"


def fp64_block(k, agent_launches, flops_per_agent_launch, workload, name, what):
    """nominal FP64 operations (evaluation counts x operations per evaluation, dense-product counts for the recursion) / kernel time
    against the FP64 vector peak, and the measured issue fill next to it.  The nominal count is NOT the executed instruction count: the
    compiler merges the common subexpressions of the 130 perturbed evaluations of a sweep item, and the
    reference's arithmetic forbids FMA contraction, so one issued instruction is one operation, not two."""
    tf = agent_launches * flops_per_agent_launch / (k["total_ms"] * 1e-3) / 1e12
    fill, fill_src = _issue_fill(workload, name)
    return dict(achieved=tf, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=tf / FP64_PEAK_TFLOPS, flops_per_agent_launch=flops_per_agent_launch,
                flops_note=what, issue_fill=fill, issue_fill_source=fill_src)


def round_fill_of(kt, workload):
    """time-weighted FP64 issue fill over the two serial chains of a round (recursion kernel + line search): how full the chip is while
    the run's critical path executes; null without SQ counters collected from these kernel sources"""
    chain = [n for n in ("linearise_riccati", "backward", "central_backward", "forward") if kt.get(n, {}).get("total_ms", 0) > 0]
    parts, tot, acc = {}, 0.0, 0.0
    for n in chain:
        fill, src = _issue_fill(workload, n)
        if fill is None:
            return dict(value=None, reason=src)
        parts[n] = dict(ms=round(kt[n]["total_ms"], 4), issue_fill=fill)
        tot += kt[n]["total_ms"]; acc += kt[n]["total_ms"] * fill
    return dict(value=(acc / tot if tot > 0 else None), kernels=parts, what="sum(kernel time x issue fill) / sum(kernel time) over the chain kernels of the instrumented pass")


def roofline_of(kt, w, workload, name=None):
    """The contract's figure: achieved = algorithmic bytes (SURVEY.md 8(d) per agent x active agents of each launch) / launch durations, against
    the HBM peak.  Kernel: the fused linearise + Riccati kernel where the round is fused (four-state models), else the linearisation sweep.
    `frac` uses the HIP events bound to each dispatch of the instrumented pass (every launch of the kernel) -- the quantity rocprofv3
    --kernel-trace averages, so profiles/*_kernel_stats.csv can be held against it.  `bound` says what ACTUALLY limits the kernel: the sweep
    kernel streams its tiles to HBM ("hbm"); the fused kernel keeps them in LDS and is limited by the recursion's dependent FP64 chain
    ("fp64_issue") -- for it `frac` is the equivalent throughput of an unfused implementation, `hbm_measured` the HBM utilisation from the PMC
    byte counters and `fp64` the figure against the limit that binds.  `inkernel_span` (sweep only) is the same sum over the kernel's own
    first-workgroup-start .. last-workgroup-end REFCLK stamps (no dispatch ramp)."""
    if name is None:
        name = "linearise_riccati" if kt.get("linearise_riccati", {}).get("launches") else "sweep"
    k = kt.get(name)
    if not k or k["launches"] == 0 or k["total_ms"] <= 0:
        return None
    symbol, what, bound_by = ROOFLINE_KERNELS[name]
    fused = name == "linearise_riccati"
    bytes_per_agent = (w["sweep_bytes_per_agent"] + w["backward_bytes_per_agent"]) if fused else w["sweep_bytes_per_agent"]
    agent_launches = k["algorithmic_bytes"] / bytes_per_agent            # sum over launches of the agents active in them
    achieved = k["algorithmic_bytes"] / (k["total_ms"] * 1e-3) / 1e9
    out = dict(bound=("fp64_issue" if fused else "hbm"), kernel=symbol, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
               frac_is=("equivalent unfused throughput: SURVEY.md 8(d)'s sweep + backward bytes / kernel time / HBM peak -- the contract's figure; the tiles stay in LDS, "
                        "so this is NOT the kernel's HBM utilisation (hbm_measured) and NOT what binds it (fp64)" if fused else
                        "SURVEY.md 8(d)'s sweep bytes / kernel time / HBM peak: the kernel streams these bytes, hbm_measured agrees with it to a few per cent"),
               timing="HIP events bound to each dispatch (hipExtLaunchKernelGGL), all launches of the instrumented pass",
               launches=k["launches"], avg_launch_us=1e3 * k["total_ms"] / k["launches"],
               algorithmic_bytes_per_launch=k["algorithmic_bytes"] / k["launches"],
               algorithmic_bytes_per_agent=bytes_per_agent,
               algorithmic_bytes_note=what, limited_by=bound_by)
    ik = kt.get(name + "_inkernel")
    if ik and ik["launches"] and ik["total_ms"] > 0:
        a = ik["algorithmic_bytes"] / (ik["total_ms"] * 1e-3) / 1e9
        out["inkernel_span"] = dict(achieved=a, frac=a / HBM_PEAK_GBS, launches=ik["launches"], avg_launch_us=1e3 * ik["total_ms"] / ik["launches"])
    full = kt.get(name + "_inkernel_full")
    if full and full["launches"] and full["total_ms"] > 0:
        fa = full["algorithmic_bytes"] / (full["total_ms"] * 1e-3) / 1e9
        out["all_agents_active"] = dict(launches=full["launches"], avg_launch_us=1e3 * full["total_ms"] / full["launches"], achieved=fa, frac=fa / HBM_PEAK_GBS,
                                        timing="in-kernel span", note="launches in which every agent of the batch is still active (first round of each solve): the kernel at its design point")
    # HBM traffic: PMC counters cannot be read from inside this process.  DERIVED figure: the measured/algorithmic ratio of
    # the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate passes, corrected as MI355X_MICROARCH.md
    # prescribes: FETCH_SIZE x2 on gfx950) applied to this run's algorithmic bytes per launch.  null (with the reason) without a
    # profile collected from exactly these kernel sources.
    pmc, src = _load_profile("pmc_traffic", workload)
    out["traffic"], out["traffic_source"], out["hbm_measured"] = None, src, None
    if pmc and name in pmc.get("kernels", {}):
        ratio = pmc["kernels"][name].get("measured_over_algorithmic")
        if ratio:
            out["traffic"] = ratio * out["algorithmic_bytes_per_launch"]
            out["traffic_source"] = f"derived: {src} (FETCH_SIZE x2 + WRITE_SIZE = {ratio:.4f} x algorithmic bytes over the launches of one strategy run; collected at {pmc.get('commit', 'unknown commit')}, source_hash {pmc.get('source_hash')})"
            out["hbm_measured"] = dict(achieved=ratio * achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=ratio * achieved / HBM_PEAK_GBS,
                                       what="HBM bytes of the PMC counters / kernel time: the kernel's actual HBM utilisation")
    # against the limit that binds the fused kernel (and, for reference, the sweep): nominal FP64 operations / time / vector peak + measured issue fill
    flops = w["sweep_flops_per_agent"] + (w["T"] * backward_flops_per_step(w["nx"], w["nu"]) if fused else 0)
    out["fp64"] = fp64_block(k, agent_launches, flops, workload, name,
                             "SURVEY.md 8(d): linearisation (evaluation counts x operations per evaluation)" + (" + backward recursion (dense products of ilqr.hpp:115-119,183-192)" if fused else ""))
    fw = kt.get("forward")
    if fw and fw["launches"] and fw["total_ms"] > 0:
        NUM_ALPHAS = 10          # alpha = 1, 1/2, ... >= 1e-3 (ilqr.hpp:199-206): kNumAlphas, csrc/mas_bodies.hpp
        out["line_search"] = dict(bound="fp64_issue", kernel="ForwardBody / mas_forward_quad_kernel", launches=fw["launches"], avg_launch_us=1e3 * fw["total_ms"] / fw["launches"],
                                  limited_by="the rollout's dependent FP64 chain (T steps per trial; ten trials side by side)",
                                  fp64=fp64_block(fw, agent_launches, NUM_ALPHAS * w["forward_flops_per_agent_trial"], workload, "forward",
                                                  "ten trial rollouts per agent and launch: T x (4 dynamics evaluations + stage cost + feedback law)"))
    out["round_fill"] = round_fill_of(kt, workload)
    return out


def utilisation_of(workload):
    """backward / forward pass against the SIMD they run on: SQ counters of the committed rocprofv3 --pmc passes
    (north_star: 'LDS/MFMA utilisation on the backward pass'; no MFMA is used -- fused multiply-adds would break the
    reference's SSE2 rounding)."""
    sq, src = _load_profile("sq_counters", workload)
    if not sq:
        return {"source": None, "reason": src}
    out = {"source": f"{src} (rocprofv3 --pmc, SQ counters of one strategy run; collected at {sq.get('commit', 'unknown commit')}, source_hash {sq.get('source_hash')})"}
    for name in ("linearise_riccati", "backward", "central_backward", "forward", "sweep"):
        k = sq.get("kernels", {}).get(name)
        if k and "derived" in k:
            out[name] = k["derived"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--agents-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--no-unfused", action="store_true", help="skip the A/B pass with the five-kernel round (c3 / c4, N = 1)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for a single rank (exercises the RCCL path on one GPU)")
    ap.add_argument("--gather-solution", action="store_true",
                    help="after the timed region, also time the all-gather of the whole Solution over RCCL (the publish step of coupled "
                         "Nash problems, nash.hpp:66-71); reported as 'solution_all_gather', never part of 'value'.  On by default for c4 with N > 1.")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: multi_agent_solver_amd has no CPU fallback")
    device = local_rank % torch.cuda.device_count()      # one rank per GPU (several per GPU only in the gloo rehearsal)
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", device))   # nccl == RCCL on ROCm
        else:
            dist.init_process_group(backend=args.backend)

    def sync():
        torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    w = dict(WORKLOADS[args.workload])
    per_gpu = args.agents_per_gpu or w["agents"]
    if per_gpu != w["agents"]:
        w["name"] = w["name"].replace(f"{w['agents']} agents", f"{per_gpu} agents") + " [agent count overridden]"
    n_total = per_gpu * world
    lo, hi = shard_range(n_total, world, rank)

    gather = None
    if (args.gather_solution or (args.workload == "c4" and world > 1)) and dist is not None:
        def gather(solver):
            from multi_agent_solver_amd.dist import all_gather_solution
            times = []
            for _ in range(3):
                barrier(); sync(); t0 = time.perf_counter()
                full = all_gather_solution(solver, n_total)
                sync(); barrier(); times.append(time.perf_counter() - t0)
            nbytes = sum(v.numel() * 8 for v in full.values())
            return {"ms": 1e3 * min(times), "bytes_received_per_rank": nbytes, "GBs_per_rank": nbytes / min(times) / 1e9, "collective": "all_gather x3 (X, U, cost)",
                    "what": "publish step of the Nash strategies (update_initial_with_best, nash.hpp:66-71): every rank receives every agent's trajectory; "
                            "off the timed path because decoupled agents (every shipped example) do not consume it"}
    elapsed, st, kt, sol = measure(w, hi - lo, lo, n_total, device, args.steps, args.warmup, sync, barrier, after=gather,
                                   unfused_reference=(world == 1 and not args.no_unfused and args.workload in ("c3", "c4")))
    vals = torch.tensor([elapsed, float(st["agent_iterations"]), float(st["converged_solves"]), float(st["solves"])],
                        dtype=torch.float64, device=("cuda" if args.backend == "nccl" else "cpu"))
    if dist is not None:
        tmax = vals[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sums = vals[1:].clone(); dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        elapsed, (iters, conv, solves) = float(tmax.item()), [float(v) for v in sums.tolist()]
    else:
        iters, conv, solves = float(vals[1]), float(vals[2]), float(vals[3])

    if rank == 0:
        total_kernel_ms = sum(k["total_ms"] for n, k in kt.items() if "_inkernel" not in n) or 1.0
        kernels = {n: dict(launches=k["launches"], total_ms=round(k["total_ms"], 4), share_of_kernel_time=round(k["total_ms"] / total_kernel_ms, 4),
                           avg_us=round(1e3 * k["total_ms"] / max(1, k["launches"]), 3),
                           achieved_GBs=(round(k["algorithmic_bytes"] / (k["total_ms"] * 1e-3) / 1e9, 2) if k["algorithmic_bytes"] > 0 and k["total_ms"] > 0 else None))
                   for n, k in sorted(kt.items(), key=lambda kv: -kv[1]["total_ms"])}
        out = {
            "metric": "ilqr_agent_iterations_per_sec", "value": iters / elapsed, "unit": "agent-iterations/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": w["name"], "agents_per_gpu": per_gpu, "agents_total": n_total, "horizon": w["T"], "nx": w["nx"], "nu": w["nu"],
                       "strategy": w["strategy"], "max_outer": w["max_outer"], "max_iterations": w["max_iterations"],
                       "tolerance": w["tolerance"], "sharding": f"agents by id range over {world} rank(s), no data-path collective"},
            "converged_agent_trajectories_per_sec": (n_total * args.steps) / elapsed if w["strategy"] != "centralized" else None,
            "agent_iterations_per_step": iters / args.steps,
            "converged_solves_fraction": conv / max(1.0, solves),
            "roofline": roofline_of(kt, w, args.workload),
            "utilisation": utilisation_of(args.workload),
            "kernels": kernels,
            "kernel_time_fraction_of_wall": total_kernel_ms * 1e-3 / st["profiled_pass_elapsed"],
            "instrumented_pass_ms_per_step": 1e3 * st["profiled_pass_elapsed"] / args.steps,
        }
        if st.get("after") is not None:
            out["solution_all_gather"] = st["after"]
        if "unfused_elapsed" in st:
            uk = st["unfused_kernels"]
            out["unfused_round"] = {
                "what": "the same K steps with the five-kernel round (sweep -> backward -> line search -> select -> commit; mas_hip_set_option 'fused' 0), same process, "
                        "same resident inputs, bit-identical results (tests/test_gpu_parity.py): what the two-kernel round buys, and the linearisation sweep as a kernel of its own",
                "ms_per_step": 1e3 * st["unfused_elapsed"] / args.steps, "value": st["unfused_agent_iterations"] / st["unfused_elapsed"],
                "speedup_of_the_fused_round": st["unfused_elapsed"] / elapsed,
                "kernels_avg_us": {n: round(1e3 * k["total_ms"] / max(1, k["launches"]), 2) for n, k in uk.items() if "_inkernel" not in n},
                "roofline_sweep": roofline_of(uk, w, args.workload, "sweep")}
        out["parity"] = parity_block(w, sol, n_total) if (world == 1 and not args.no_parity) else None
        out["cpu_baseline"] = cpu_baseline(w, n_total) if (world == 1 and not args.no_cpu_baseline) else None
        if world == 1 and not args.no_also and args.workload == "c2":
            out["literal_stacked_solve"] = literal_stacked_solve_block(w, device, float(sol["total_cost"]))
        if world == 1 and not args.no_also and args.workload == "c3":
            w2 = dict(WORKLOADS["c2"])
            el2, st2, kt2, sol2 = measure(w2, w2["agents"], 0, w2["agents"], device, max(3, args.steps), 1, sync, barrier)
            out["also_config2"] = {"workload": w2["name"], "value": st2["agent_iterations"] / el2, "unit": "agent-iterations/s",
                                   "ms_per_step": 1e3 * el2 / max(3, args.steps), "roofline": roofline_of(kt2, w2, "c2", "sweep"),
                                   "kernels_avg_us": {n: round(1e3 * k["total_ms"] / max(1, k["launches"]), 2) for n, k in kt2.items() if "_inkernel" not in n},
                                   "parity": None if args.no_parity else parity_block(w2, sol2, w2["agents"])}
            out["also_config2"]["literal_stacked_solve"] = literal_stacked_solve_block(w2, device, float(sol2["total_cost"]))
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
