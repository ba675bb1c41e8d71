#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X batched iLQR path.

Metric (BASELINE.json): iLQR iterations/sec x agents on the single-track H=100 workload.
A "step" is one complete strategy run (the span the reference times around mas::solve,
examples/multi_agent_single_track.cpp:123-126) over one batch of synthetic agents whose inputs are
already resident in HBM:  restore initial (x0, U0=0) -> strategy -> results stay on the device.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c3|c2] [--agents-per-gpu A]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`:
one rank per GPU, agents sharded by id range (weak scaling: A agents per GPU), no data-path
collective (agents are decoupled); RCCL is used only to take the MAX of
the ranks' times and the SUM of their iteration counts.

Rank 0 prints ONE JSON line.  Besides the contract's fields it carries
  roofline     -- the linearisation sweep (the kernel north_star names), HIP-event timed in this run
  cpu_baseline -- the CPU restatement of the reference algorithm (oracle, OpenMP over agents as
                  nash.hpp:59), timed on this host on a bounded sample of the same workload
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


from multi_agent_solver_amd.dist import shard_range  # noqa: E402  contiguous ranges of the id-sorted block list, one per rank


WORKLOADS = {
    # BASELINE.json configs[2]
    "c3": dict(name="multi_agent_single_track --solver ilqr --strategy sequential, 4096 agents per GPU, H=100 (BASELINE configs[2])",
               agents=4096, T=100, dt=0.5, strategy="sequential", max_outer=10, max_iterations=100, tolerance=1e-5),
    # BASELINE.json configs[1]
    "c2": dict(name="multi_agent_single_track --solver ilqr --strategy centralized, 256 agents, H=100 (BASELINE configs[1])",
               agents=256, T=100, dt=0.5, strategy="centralized", max_outer=1, max_iterations=100, tolerance=1e-5),
}


def make_solver(w, n_local, first, n_total, device):
    from multi_agent_solver_amd import BatchSolver, MODEL_SINGLE_TRACK
    from multi_agent_solver_amd.batch import single_track_x0
    s = BatchSolver(MODEL_SINGLE_TRACK, n_local, w["T"], w["dt"], device=device)
    s.set_params({"max_iterations": w["max_iterations"], "tolerance": w["tolerance"], "max_ms": math.inf})
    s.set_initial_states(single_track_x0(n_local, first=first, total=n_total))
    s.set_controls(None)
    s.initialize_problem()
    s.save_initial()
    return s


def one_step(s, w):
    s.restore_initial()
    s.run(w["strategy"], w["max_outer"])


def measure(w, n_local, first, n_total, device, steps, warmup, sync, barrier, after=None):
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
    if after is not None:
        st["after"] = after(s)
    sol = s.solution()
    s.close()
    return elapsed, st, kt, sol


def cpu_baseline(w, n_total, budget_agents=None):
    """CPU restatement of the reference algorithm (oracle/), OpenMP over agents like nash.hpp:59.
    Bounded sample: every k-th agent of the workload, whole strategy."""
    from oracle import oracle as O
    import numpy as np
    from multi_agent_solver_amd.batch import single_track_x0
    threads = O.max_threads()
    O.set_trig_mode("libm")      # std::sin/cos/tan, as the reference binary calls them
    n = budget_agents or min(n_total, 64 * threads)
    stride = max(1, n_total // n)
    x0 = single_track_x0(n_total)[::stride][:n]
    its, el, reps, r = 0, 0.0, 0, None
    while el < 8.0 and reps < 20:            # bounded: about 10 s of CPU work
        t0 = time.perf_counter()
        r = O.solve_batch(O.MODEL_SINGLE_TRACK, x0, None, w["T"], w["dt"], strategy=w["strategy"], max_outer=w["max_outer"],
                          max_iterations=w["max_iterations"], tolerance=w["tolerance"], threads=threads)
        el += time.perf_counter() - t0
        its += int(r["stats"]["total_iterations"]) * (x0.shape[0] if w["strategy"] == "centralized" else 1)
        reps += 1
    return dict(value=its / el, unit="agent-iterations/s", cores=threads, kind="port",
                sample=f"{x0.shape[0]} agents (every {stride}th of {n_total}), full '{w['strategy']}' strategy run repeated {reps}x, {el:.1f} s total; "
                       f"CPU restatement of reference algorithm (oracle/, gcc -O2 -ffp-contract=off, OpenMP schedule(static) over agents as nash.hpp:59, glibc trig)"), r, x0


def roofline_of(kt, name="sweep"):
    # prefer the in-kernel REFCLK timing of the sweep (kernel execution span, what rocprofv3 --kernel-trace
    # reports); the host-side event timing of the same launches carries ~5 us of dispatch overhead each
    k = kt.get(name + "_inkernel") or kt.get(name)
    if not k or k["launches"] == 0 or k["total_ms"] <= 0:
        return None
    achieved = k["algorithmic_bytes"] / (k["total_ms"] * 1e-3) / 1e9
    # HBM traffic per launch: PMC counters cannot be read from inside this process; the ratio measured/algorithmic
    # of the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic_*.json, corrected as MI355X_MICROARCH.md
    # prescribes) is applied to this run's algorithmic bytes per launch.  None when no such profile is present.
    traffic, traffic_src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_traffic_c3.json")))
        if name == "sweep":
            ratio = (pmc["sweep"]["fetch_bytes_corrected_KB"] + pmc["after_chunk_major_tile_layout"]["sweep_WRITE_SIZE_KB_first_dispatch"]) / \
                    (pmc["sweep"]["algorithmic_read_KB"] + pmc["sweep"]["algorithmic_write_KB"])
            traffic = ratio * k["algorithmic_bytes"] / k["launches"]
            traffic_src = f"profiles/r1_pmc_traffic_c3.json (FETCH_SIZE x2 + WRITE_SIZE = {ratio:.4f} x algorithmic bytes on a full launch)"
    except (OSError, KeyError, ValueError):
        pass
    full = kt.get(name + "_inkernel_full")
    full_info = None
    if full and full["launches"] and full["total_ms"] > 0:
        fa = full["algorithmic_bytes"] / (full["total_ms"] * 1e-3) / 1e9
        full_info = dict(launches=full["launches"], avg_launch_us=1e3 * full["total_ms"] / full["launches"], achieved=fa, frac=fa / HBM_PEAK_GBS,
                         note="launches in which every agent of the batch is still active (first round of each solve)")
    return dict(bound="hbm", kernel=name, achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                all_agents_active=full_info,
                traffic=traffic, traffic_source=traffic_src, timing=("in-kernel s_memrealtime span" if (name + "_inkernel") in kt else "hip events"),
                launches=k["launches"], avg_launch_us=1e3 * k["total_ms"] / k["launches"],
                algorithmic_bytes_per_launch=k["algorithmic_bytes"] / k["launches"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--agents-per-gpu", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl == RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed even for a single rank (exercises the RCCL path on one GPU)")
    ap.add_argument("--gather-solution", action="store_true",
                    help="after the timed region, also time the all-gather of the whole Solution over RCCL (the publish step of coupled "
                         "Nash problems, nash.hpp:66-71); reported as 'solution_all_gather', never part of 'value'")
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
    if args.gather_solution and dist is not None:
        def gather(solver):
            from multi_agent_solver_amd.dist import all_gather_solution
            times = []
            for _ in range(3):
                barrier(); sync(); t0 = time.perf_counter()
                full = all_gather_solution(solver, n_total)
                sync(); barrier(); times.append(time.perf_counter() - t0)
            nbytes = sum(v.numel() * 8 for v in full.values())
            return {"ms": 1e3 * min(times), "bytes_received_per_rank": nbytes, "GBs_per_rank": nbytes / min(times) / 1e9, "collective": "all_gather x3 (X, U, cost)"}
    elapsed, st, kt, _ = measure(w, hi - lo, lo, n_total, device, args.steps, args.warmup, sync, barrier, after=gather)
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
            "config": {"workload": w["name"], "agents_per_gpu": per_gpu, "agents_total": n_total, "horizon": w["T"], "nx": 4, "nu": 2,
                       "strategy": w["strategy"], "max_outer": w["max_outer"], "max_iterations": w["max_iterations"],
                       "tolerance": w["tolerance"], "sharding": f"agents by id range over {world} rank(s), no data-path collective"},
            "converged_agent_trajectories_per_sec": (n_total * args.steps) / elapsed if w["strategy"] != "centralized" else None,
            "agent_iterations_per_step": iters / args.steps,
            "converged_solves_fraction": conv / max(1.0, solves),
            "roofline": roofline_of(kt, "sweep"),
            "kernels": kernels,
            "kernel_time_fraction_of_wall": total_kernel_ms * 1e-3 / st["profiled_pass_elapsed"],
            "instrumented_pass_ms_per_step": 1e3 * st["profiled_pass_elapsed"] / args.steps,
        }
        if st.get("after") is not None:
            out["solution_all_gather"] = st["after"]
        if world == 1 and not args.no_cpu_baseline:
            cb, _, _ = cpu_baseline(w, n_total)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_also and args.workload == "c3":
            w2 = dict(WORKLOADS["c2"])
            el2, st2, kt2, _ = measure(w2, w2["agents"], 0, w2["agents"], device, max(3, args.steps), 1, sync, barrier)
            out["also_config2"] = {"workload": w2["name"], "value": st2["agent_iterations"] / el2, "unit": "agent-iterations/s",
                                   "ms_per_step": 1e3 * el2 / max(3, args.steps), "roofline": roofline_of(kt2, "sweep")}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
