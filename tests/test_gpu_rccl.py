"""Native multi-GPU entry points of the C ABI (include/mas_hip.h: mas_hip_comm_*, mas_hip_all_gather_solution,
mas_hip_set_comm_cost_exchange) on the one GPU a test box has: a one-rank RCCL communicator made and used by the library, and
two handles driven by two host threads on one device (the shape of one process driving several GPUs)."""
import math
import threading

import numpy as np
import pytest

from conftest import run_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm(product_lib):
    from multi_agent_solver_amd.batch import Comm
    assert product_lib.mas_hip_device_count() > 0
    c = Comm(Comm.unique_id(product_lib), 1, 0, 0, lib=product_lib)
    yield c
    c.close()


def test_shard_ranges_are_contiguous_and_cover(product_lib, comm):
    import ctypes as C
    for n_total, world in ((10, 3), (4096, 8), (7, 8), (65536, 8)):
        spans = []
        for r in range(world):
            lo, hi = C.c_int(), C.c_int()
            product_lib.mas_hip_shard_range(n_total, world, r, C.byref(lo), C.byref(hi))
            spans.append((lo.value, hi.value))
        assert spans[0][0] == 0 and spans[-1][1] == n_total and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert comm.shard_range(100) == (0, 100)


def test_one_rank_all_gather_of_the_solution_through_rccl(product_lib, oracle, comm):
    """the publish step through the library's RCCL path: layout conversion on the device straight into the caller's contiguous
    buffers, one ncclAllGather per array on the handle's stream; equal to the host read-back and to the oracle"""
    import torch
    from multi_agent_solver_amd.batch import BatchSolver
    N, T = 70, 25
    x0 = oracle.single_track_x0(N)
    dev = torch.device("cuda", 0)
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, N, T, 0.5, lib=product_lib) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(x0); s.set_controls(None); s.initialize_problem()
        s.run_sequential(2)
        X = torch.full((N, T + 1, 4), float("nan"), dtype=torch.float64, device=dev)
        U = torch.full((N, T, 2), float("nan"), dtype=torch.float64, device=dev)
        c = torch.full((N,), float("nan"), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        s.all_gather_solution_device(comm, N, X.data_ptr(), U.data_ptr(), c.data_ptr())
        host = s.solution()
        with pytest.raises(Exception):                  # a handle whose agent count is not its rank's shard is refused
            s.all_gather_solution_device(comm, N + 1, X.data_ptr(), U.data_ptr(), c.data_ptr())
    assert np.array_equal(X.cpu().numpy(), host["X"]) and np.array_equal(U.cpu().numpy(), host["U"]) and np.array_equal(c.cpu().numpy(), host["cost"])
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="sequential", max_outer=2)
    assert np.array_equal(host["X"], o["X"]) and np.array_equal(host["cost"], o["cost"])


@pytest.mark.parametrize("model,T,dt", [("MODEL_SINGLE_TRACK", 40, 0.5), ("MODEL_LQR_IDENTITY", 20, 0.1)])
def test_line_search_total_cost_on_the_device_and_through_rccl(product_lib, oracle, comm, model, T, dt):
    """detail::run_line_search (nash.hpp:92-180) with its joint step decided on the device: local sum, and the sum over `all ranks'
    agents` through the library's RCCL exchange (one rank) -- both bit-equal to the oracle"""
    from multi_agent_solver_amd.batch import BatchSolver
    mid = getattr(oracle, model)
    N = 37
    rng = np.random.default_rng(3)
    base = oracle.lqr_x0(N) if model == "MODEL_LQR_IDENTITY" else oracle.single_track_x0(N)
    x0 = base + 0.05 * rng.standard_normal(base.shape)
    o = oracle.solve_batch(mid, x0, None, T, dt, strategy="linesearch", max_outer=4)
    for use_comm in (False, True):
        with BatchSolver(mid, N, T, dt, lib=product_lib) as s:
            s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
            s.set_initial_states(x0); s.set_controls(None); s.initialize_problem()
            if use_comm:
                s.set_comm_cost_exchange(comm, N)
            s.run_line_search(4)
            e = s.solution()
        assert np.array_equal(e["cost"], o["cost"]) and np.array_equal(e["X"], o["X"]) and np.array_equal(e["U"], o["U"]), use_comm
        assert e["iters"].tolist() == o["iters"].tolist() and e["total_cost"] == o["total_cost"]


def test_two_handles_two_host_threads_one_device(product_lib, oracle):
    """one process driving several handles from one host thread each (what mas_amd::MultiAgentProblem does with several GPUs; here both
    handles share the one GPU of the box): the id-sorted block list cut by mas_hip_shard_range, every shard solved concurrently,
    bit-equal to one handle holding all agents"""
    N, T = 150, 40
    x0 = oracle.single_track_x0(N)
    whole = run_batch(product_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "trustregion", max_outer=3)
    spans = [(0, 75), (75, 150)]
    parts = [None, None]

    def work(k):
        lo, hi = spans[k]
        parts[k] = run_batch(product_lib, oracle.MODEL_SINGLE_TRACK, x0[lo:hi], None, T, 0.5, "trustregion", max_outer=3)

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for key in ("cost", "X", "U", "iters", "status", "radii"):
        assert np.array_equal(np.concatenate([p[key] for p in parts]), whole[key]), key
    assert sum(p["stats"]["agent_iterations"] for p in parts) == whole["stats"]["agent_iterations"]


def _torch_alloc(shape):
    import torch
    t = torch.full(shape, float("nan"), dtype=torch.float64, device=torch.device("cuda", 0))
    torch.cuda.synchronize()
    return t, t.data_ptr()


@pytest.mark.parametrize("n_ranks,N,strategy", [(3, 10, "linesearch"), (3, 9, "linesearch"), (2, 130, "linesearch"), (4, 70, "trustregion")])
def test_several_ranks_on_one_gpu_over_the_loopback_transport(product_lib, oracle, n_ranks, N, strategy):
    """The library's multi-rank logic with more than one rank ON THE DEVICE: n shards of one problem on the box's single GPU, each with its
    own handle, stream and host thread, their communicators made by mas_hip_comm_create_all over the in-process loopback transport
    (RCCL refuses several ranks per device; csrc/mas_rccl.hpp) -- device-side layout conversion straight to the shard's displacement,
    even and uneven shards, the line search's total over all ranks' costs summed by one wavefront in global block order.  Every
    shard bit-equal to one handle holding all agents; the all-gathered Solution complete on every rank; the total the very bits of
    the one handle's.  (tests/test_loopback_ranks.py: the same through the host emulation, to 8 ranks.)"""
    from test_loopback_ranks import check_against_one_handle
    model = oracle.MODEL_SINGLE_TRACK
    check_against_one_handle(product_lib, oracle, n_ranks, N, model, 30, 0.5, strategy, 3, alloc=_torch_alloc, to_numpy=lambda t: t.cpu().numpy())


def test_a_failing_rank_releases_its_peers_on_the_device(product_lib, oracle):
    from test_loopback_ranks import run_ranks
    x0 = oracle.single_track_x0(9)
    out, errors, hung = run_ranks(product_lib, 3, oracle.MODEL_SINGLE_TRACK, x0, 10, 0.5, "linesearch", 2, _torch_alloc, lambda t: t.cpu().numpy(), fail_rank=2)
    assert not hung and isinstance(errors[2], RuntimeError)
    assert all(errors[r] is not None and out[r] is None for r in (0, 1)), errors


def test_two_ranks_on_two_gpus_over_rccl(product_lib, oracle):
    """librccl itself with MORE than one rank: two GPUs, uneven shards (grouped in-place broadcasts) and the line-search exchange,
    bit-equal to one handle.  Skipped on a one-GPU box -- which every box of this pool is: the collectives have run over librccl with
    one rank only; everything above the table of function pointers is pinned by the loopback tests."""
    if product_lib.mas_hip_device_count() < 2:
        pytest.skip("needs two GPUs")
    import torch
    from multi_agent_solver_amd.batch import BatchSolver, Comm
    N, T = 11, 20
    x0 = oracle.single_track_x0(N)
    whole = run_batch(product_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "linesearch", max_outer=3)
    comms = Comm.create_all([0, 1], lib=product_lib)
    out, errors = [None, None], [None, None]

    def worker(r):
        try:
            lo, hi = comms[r].shard_range(N)
            with BatchSolver(oracle.MODEL_SINGLE_TRACK, hi - lo, T, 0.5, lib=product_lib, device=r) as s:
                s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
                s.set_initial_states(x0[lo:hi]); s.set_controls(None); s.initialize_problem()
                s.set_comm_cost_exchange(comms[r], N)
                s.run_line_search(3)
                sol = s.solution()
                dev = torch.device("cuda", r)
                X = torch.full((N, T + 1, 4), float("nan"), dtype=torch.float64, device=dev)
                torch.cuda.synchronize(dev)
                s.all_gather_solution_device(comms[r], N, X.data_ptr(), 0, 0)
                out[r] = (lo, hi, sol, X.cpu().numpy())
        except Exception as e:      # noqa: BLE001
            errors[r] = e
            comms[r].abort()

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads) and errors == [None, None], errors
    for lo, hi, sol, X in out:
        assert np.array_equal(sol["cost"], whole["cost"][lo:hi]) and sol["total_cost"] == whole["total_cost"] and np.array_equal(X, whole["X"])
    for c in comms:
        c.close()
