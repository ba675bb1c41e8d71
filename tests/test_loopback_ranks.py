"""The library's multi-rank logic with MORE THAN ONE RANK, without a second GPU: mas_hip_comm_create_all, Comm::all_gather_shards (even
and uneven shards: one in-place all-gather / grouped in-place broadcasts by displacement), mas_hip_all_gather_solution and
mas_hip_set_comm_cost_exchange -> JointSumBody over the all-gathered costs, driven by 2, 3 and 8 ranks = host threads of this
process over the in-process loopback transport (csrc/mas_rccl.hpp, mas_hip_comm_use_loopback) -- every result bit-equal to ONE
handle holding all agents.  CPU tier: the host emulation of the product sources (the "device" buffers are host memory);
tests/test_gpu_rccl.py repeats it with three shards on one GPU.  The collectives have run over librccl itself with one rank only
(DESIGN.md section 6): what this pins is everything above the table of function pointers."""
import ctypes as C
import math
import os
import subprocess
import threading

import numpy as np
import pytest

from conftest import run_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_ranks(lib, n_ranks, model, x0_all, T, dt, strategy, outer, alloc, to_numpy, device=0, fail_rank=None):
    """every rank on its own thread and handle; returns per-rank dict(sol, full=(X, U, cost) all-gathered on that rank)"""
    from multi_agent_solver_amd.batch import BatchSolver, Comm
    N = x0_all.shape[0]
    Comm.use_loopback(True, lib)
    try:
        comms = Comm.create_all([device] * n_ranks, lib=lib)
    finally:
        Comm.use_loopback(False, lib)                   # (communicators keep the table they were made by)
    out, errors = [None] * n_ranks, [None] * n_ranks

    def worker(r):
        try:
            lo, hi = comms[r].shard_range(N)
            if r == fail_rank:
                raise RuntimeError("this rank fails before its first collective")
            with BatchSolver(model, hi - lo, T, dt, lib=lib, device=device) as s:
                s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
                s.set_initial_states(x0_all[lo:hi])
                s.set_controls(None)
                s.initialize_problem()
                if strategy == "linesearch":
                    s.set_comm_cost_exchange(comms[r], N)          # the total cost of ALL ranks' agents decides the joint step
                s.run(strategy, outer)
                sol = s.solution()
                sol["total_cost_as_the_handle_reports_it"] = s.total_cost()
                nx, nu = s.nx, s.nu
                X, U, c = alloc((N, T + 1, nx)), alloc((N, T, nu)), alloc((N,))
                s.all_gather_solution_device(comms[r], N, X[1], U[1], c[1])
                out[r] = dict(sol=sol, full=(to_numpy(X[0]), to_numpy(U[0]), to_numpy(c[0])), range=(lo, hi))
        except Exception as e:      # noqa: BLE001 -- a failing rank releases its peers (no rank may wait forever)
            errors[r] = e
            comms[r].abort()

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(n_ranks)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    hung = [r for r, t in enumerate(threads) if t.is_alive()]
    for c in comms:
        c.close()
    return out, errors, hung


def host_alloc(shape):
    a = np.full(shape, np.nan)
    return a, a.ctypes.data


def check_against_one_handle(lib, oracle, n_ranks, N, model, T, dt, strategy, outer, alloc=host_alloc, to_numpy=lambda a: a, device=0):
    rng = np.random.default_rng(100 * n_ranks + N)
    base = {oracle.MODEL_LQR_IDENTITY: oracle.lqr_x0, oracle.MODEL_BICYCLE6: oracle.bicycle6_x0}.get(model, oracle.single_track_x0)(N)
    x0 = base + 0.05 * rng.standard_normal(base.shape)
    whole = run_batch(lib, model, x0, None, T, dt, strategy, max_outer=outer, device=device)
    out, errors, hung = run_ranks(lib, n_ranks, model, x0, T, dt, strategy, outer, alloc, to_numpy, device=device)
    assert not hung and all(e is None for e in errors), (hung, errors)
    for r, o in enumerate(out):
        lo, hi = o["range"]
        for key in ("X", "U", "cost", "iters"):
            assert np.array_equal(o["sol"][key], whole[key][lo:hi]), (r, key)          # the shard's own result
        if strategy == "linesearch":
            # the total of ALL ranks' costs as THIS rank's device formed it (all-gather by displacement + one wavefront's serial sum in
            # global block order): the very bits the one handle's sum has -- a wrong displacement or a stale shard would show here
            assert o["sol"]["total_cost_as_the_handle_reports_it"] == whole["total_cost"], r
        X, U, c = o["full"]                                                           # what the all-gather left on THIS rank: everybody's
        assert np.array_equal(X, whole["X"]) and np.array_equal(U, whole["U"]) and np.array_equal(c, whole["cost"]), r
    return whole


@pytest.mark.parametrize("n_ranks,N", [(2, 8), (2, 7), (3, 10), (3, 9), (8, 19), (8, 16)])
def test_line_search_total_over_all_ranks_and_the_solution_all_gather(hostcheck_lib, oracle, n_ranks, N):
    """even (N % ranks == 0: one in-place all-gather) and uneven shards (grouped in-place broadcasts), 2 / 3 / 8 ranks: the line search
    accepts or rejects its joint step on the total of ALL ranks' costs -- summed in global block order on every rank -- so every shard
    must end exactly where one handle holding all agents ends; and the publish all-gather leaves the whole Solution on every rank"""
    check_against_one_handle(hostcheck_lib, oracle, n_ranks, N, oracle.MODEL_SINGLE_TRACK, 20, 0.5, "linesearch", 3)


@pytest.mark.parametrize("model_name,N,T,dt,strategy", [("MODEL_LQR_IDENTITY", 11, 12, 0.1, "sequential"), ("MODEL_BICYCLE6", 7, 16, 0.25, "trustregion")])
def test_the_configs_that_shard_without_an_exchange(hostcheck_lib, oracle, model_name, N, T, dt, strategy):
    """C4's and C5's shapes over three uneven shards: no exchange on the data path, the all-gather of the Solution by displacement"""
    check_against_one_handle(hostcheck_lib, oracle, 3, N, getattr(oracle, model_name), T, dt, strategy, 2)


def test_a_failing_rank_releases_its_peers(hostcheck_lib, oracle):
    """one of three ranks fails before its first collective and aborts its communicator: the others must come back with an error
    from the collective they wait in -- never hang (run_sharded could wait forever)"""
    x0 = oracle.single_track_x0(9)
    out, errors, hung = run_ranks(hostcheck_lib, 3, oracle.MODEL_SINGLE_TRACK, x0, 10, 0.5, "linesearch", 2, host_alloc, lambda a: a, fail_rank=1)
    assert not hung
    assert isinstance(errors[1], RuntimeError)
    assert all(errors[r] is not None and out[r] is None for r in (0, 2)), errors      # released with an error status, not with a result


def test_shard_displacements(hostcheck_lib):
    """shard_range / shard_pieces (mas_rccl.hpp) for 1 .. 9 ranks and 1 .. 70 agents: tests/hostcheck/shards_check.cpp"""
    exe = os.path.join(ROOT, "tests", "hostcheck", "build", "shards_check")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, (r.stdout, r.stderr)
