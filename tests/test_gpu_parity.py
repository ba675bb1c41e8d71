"""Parity of the HIP path (through the C ABI, libmas_hip.so) with the CPU oracle on a real MI355X.

Tolerance: ZERO.  north_star asks for <= 1e-5 relative cost error, but the reference algorithm is
chaotic on this workload (tests/test_oracle_sensitivity.py: one ulp of input moves converged costs by
per cents), so the only meaningful statement is bit-exactness: same operation order, no FMA
contraction, IEEE-correct sqrt and division on both sides, and sin/cos/tan from the shared pure-IEEE
include/mas_dettrig.h (the oracle runs in its "det" trig mode).  Every comparison below is
np.array_equal on costs, trajectories and iteration counts; the only exception is the centralised
merit, which the GPU sums agent-major and the oracle step-major (1e-12 relative).
"""
import math
import os

import numpy as np
import pytest

from conftest import run_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu_lib(product_lib):
    assert product_lib.mas_hip_device_count() > 0, "no HIP device: the product has no CPU fallback"
    return product_lib


def _same(e, o, total_exact=True):
    assert np.array_equal(e["cost"], o["cost"]), np.abs(e["cost"] - o["cost"]).max()
    assert e["iters"].tolist() == o["iters"].tolist()
    assert np.array_equal(e["X"], o["X"])
    assert np.array_equal(e["U"], o["U"])
    assert e["stats"]["reg_events"] == o["stats"]["reg_events"]
    if (o["status"] >= 0).all():                    # the oracle reports exit reasons for the Nash strategies
        assert (e["status"] & 3).tolist() == o["status"].tolist()
    if "radii" in o:                                # trust-region strategy: the radii every agent ends with (nash.hpp:231,241)
        assert "radii" in e and np.array_equal(e["radii"], o["radii"])
    if total_exact:
        assert e["total_cost"] == o["total_cost"]
    else:
        assert abs(e["total_cost"] - o["total_cost"]) <= 1e-12 * abs(o["total_cost"])


@pytest.mark.parametrize("strategy", ["single", "sequential", "trustregion", "linesearch", "centralized"])
def test_single_track_h100(gpu_lib, oracle, strategy):
    x0 = oracle.single_track_x0(8)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy=strategy)
    _same(e, o, total_exact=(strategy != "centralized"))


@pytest.mark.parametrize("fused", [1, 0])
def test_staged_sweep_and_backward(gpu_lib, oracle, fused):
    """derivative tiles, gains and rollout of ONE linearisation + backward pass against the oracle, element by element: through the
    fused linearise + Riccati kernel (its tiles dumped from the LDS ring) and through the separate sweep and backward kernels"""
    from multi_agent_solver_amd.batch import BatchSolver
    N, T = 70, 30
    x0 = oracle.single_track_x0(N)
    rng = np.random.default_rng(1)
    U = 0.3 * rng.standard_normal((N, T, 2))
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, N, T, 0.5, lib=gpu_lib) as s:
        s.set_option("fused", fused)
        s.set_initial_states(x0)
        s.set_controls(U)
        s.initialize_problem()
        got = s.debug_backward()
        sol = s.solution()
    for a in (0, 1, 33, 63, 64, 69):
        ref = oracle.backward_pass(oracle.MODEL_SINGLE_TRACK, x0[a], U[a], T, 0.5)
        assert np.array_equal(sol["X"][a], ref["X"])
        assert sol["cost"][a] == ref["cost"]
        for name in ("A", "B", "lx", "lu", "lxx", "luu", "lux"):
            assert np.array_equal(got["derivs"][name][a], ref["derivs"][name]), name
        assert np.array_equal(got["k"][a], ref["k"]) and np.array_equal(got["K"][a], ref["K"])


def test_lqr_identity(gpu_lib, oracle):
    x0 = oracle.lqr_x0(5)
    e = run_batch(gpu_lib, oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, "sequential")
    o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy="sequential")
    _same(e, o)


@pytest.mark.parametrize("strategy,outer", [("sequential", 10), ("trustregion", 3), ("linesearch", 2), ("centralized", 1)])
def test_lqr_identity_against_the_oracle_in_libm_mode(gpu_lib, oracle, strategy, outer):
    """The one parity statement that does NOT pass through the sin / cos / tan header the product and the oracle's `det` mode share
    (include/mas_dettrig.h): the identity LQR has no trigonometry, so the oracle in its glibc-faithful `libm` mode -- what a
    glibc-linked reference binary computes -- must be met bit for bit too (perturbed agents, so that they differ)."""
    rng = np.random.default_rng(5)
    x0 = oracle.lqr_x0(70) + 0.2 * rng.standard_normal((70, 4))
    oracle.set_trig_mode("libm")
    try:
        o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy=strategy, max_outer=outer)
    finally:
        oracle.set_trig_mode("det")
    e = run_batch(gpu_lib, oracle.MODEL_LQR_IDENTITY, x0, None, 50, 0.1, strategy, max_outer=outer)
    _same(e, o, total_exact=(strategy != "centralized"))


@pytest.mark.parametrize("T", [50, 60])
def test_pendulum(gpu_lib, oracle, T):
    x0, U0 = oracle.pendulum_problem(T, 0.05)
    e = run_batch(gpu_lib, oracle.MODEL_PENDULUM, x0, U0, T, 0.05, "single", max_iterations=1000, tolerance=1e-4)
    o = oracle.solve_batch(oracle.MODEL_PENDULUM, x0, U0, T, 0.05, strategy="single", max_iterations=1000, tolerance=1e-4)
    _same(e, o)


def test_lane_following(gpu_lib, oracle):
    x0 = np.array([[0.0, 1.0, 0.0, 0.0]])
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK_LANE, x0, None, 80, 0.1, "single", max_iterations=10)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LANE, x0, None, 80, 0.1, strategy="single", max_iterations=10)
    _same(e, o)


def test_bicycle6_trust_region(gpu_lib, oracle):
    x0 = oracle.bicycle6_x0(4)
    e = run_batch(gpu_lib, oracle.MODEL_BICYCLE6, x0, None, 200, 0.25, "trustregion")
    o = oracle.solve_batch(oracle.MODEL_BICYCLE6, x0, None, 200, 0.25, strategy="trustregion")
    _same(e, o)                                    # (includes the trust radii)
    assert set(np.unique(np.log(o["radii"]) / np.log(1.5)).round(6)) != {0.0}      # and they moved: accepted steps grow the radius by 1.5, reverted ones halve it


def test_edge_cases_ragged_and_tiny(gpu_lib, oracle):
    for N, T in ((1, 1), (1, 100), (65, 7), (130, 20)):
        x0 = oracle.single_track_x0(N)
        e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "sequential", max_outer=2)
        o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="sequential", max_outer=2)
        _same(e, o)


@pytest.mark.parametrize("strategy,outer", [("sequential", 4), ("trustregion", 3)])
def test_perturbed_agents_converge_at_different_rounds(gpu_lib, oracle, strategy, outer):
    """SURVEY 8(d) robustness variant: x0 perturbed by a seeded N(0, 0.1^2), so agents are not rotations of one
    another and retire in different rounds -- mixed wavefronts for the compacted sweep, the chained solves of the
    sequential strategy and the whole-wavefront forward pass; random warm-start controls on top."""
    N, T = 300, 60
    rng = np.random.default_rng(0)
    x0 = oracle.single_track_x0(N) + 0.1 * rng.standard_normal((N, 4))
    U0 = 0.05 * rng.standard_normal((N, T, 2))
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, U0, T, 0.5, strategy, max_outer=outer)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, U0, T, 0.5, strategy=strategy, max_outer=outer)
    _same(e, o)
    assert len(set(o["iters"].tolist())) > 3          # the agents really do finish at different times


def test_golden_fixtures_on_the_gpu(gpu_lib):
    """the committed (oracle-generated, det-trig) vectors, without running the oracle"""
    import importlib.util
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("mov", os.path.join(here, "golden", "make_oracle_vectors.py"))
    mov = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mov)
    gold = json.load(open(os.path.join(here, "golden", "oracle_vectors.json")))["cases"]
    for name, c in gold.items():
        x0, U0 = mov.inputs(c["model"], c["N"], c["T"], c["dt"])     # (centralized_dense: the literal stacked solve, mas_hip_run_centralized_dense)
        e = run_batch(gpu_lib, c["model"], x0, U0, c["T"], c["dt"], c["strategy"], max_outer=c["max_outer"],
                      max_iterations=c["max_iterations"], tolerance=c["tolerance"])
        assert [float.hex(v) for v in e["cost"]] == c["cost"], name
        assert e["iters"].tolist() == c["iters"], name
        assert [[float.hex(v) for v in e["X"][a, -1]] for a in range(c["N"])] == c["x_final"], name


@pytest.mark.parametrize("N,T", [(2, 10), (4, 10), (10, 10), (16, 12), (3, 40)])
@pytest.mark.parametrize("chip", [1, 0])
def test_centralized_dense_literal_stacked_solve(gpu_lib, oracle, N, T, chip):
    """mas_hip_run_centralized_dense: the reference's LITERAL centralised strategy -- one iLQR on the stacked OCP, finite
    differences of the summed cost, dense cross-agent Hessians, dense LLT and gains (centralized.hpp:18-38,
    multi_agent_problem.hpp:52-127) -- bit-equal to the oracle's PLAIN dense restatement (mas_ilqr_solve on the stacked OCP through
    the generic FD hooks), as chip-wide launches with exact re-use (csrc/mas_dense_chip.hpp) and in the one-workgroup form."""
    x0 = oracle.single_track_x0(N)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "centralized_dense", options={"dense_chip": chip})
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="centralized_dense")
    _same(e, o)


@pytest.mark.parametrize("model,N,T,dt", [("MODEL_PENDULUM", 2, 12, 0.05), ("MODEL_LQR_IDENTITY", 3, 10, 0.1), ("MODEL_BICYCLE6", 2, 8, 0.1),
                                          ("MODEL_SINGLE_TRACK_LANE", 5, 20, 0.1), ("MODEL_SINGLE_TRACK", 7, 30, 0.5)])
def test_centralized_dense_chip_other_models(gpu_lib, oracle, model, N, T, dt):
    """the chip-wide literal solve on the other workloads (non-trivial terminal cost, analytic hooks that the stacked OCP does NOT
    inherit, six states, perturbed and clamped agents, a budget of iterations): against the plain restatement, the structured one and
    the one-workgroup form"""
    mid = getattr(oracle, model)
    nx, nu, _ = oracle.model_dims(mid)
    rng = np.random.default_rng(5)
    if model == "MODEL_PENDULUM":
        x0 = np.array([[math.pi - 0.05, 0.0], [math.pi - 0.1, 0.1]])
    elif nx == 4 and model != "MODEL_LQR_IDENTITY":
        x0 = oracle.single_track_x0(N) + 0.05 * rng.standard_normal((N, 4))
    else:
        x0 = np.tile(np.eye(nx)[0], (N, 1)) + 0.1 * np.arange(N)[:, None] + (np.array([0, 0, 0, 2.0, 0, 0])[None, :nx] if nx == 6 else 0)
    kw = dict(max_iterations=6) if model == "MODEL_SINGLE_TRACK" else {}
    e = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized_dense", **kw)
    one = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized_dense", options={"dense_chip": 0}, **kw)
    for key in ("cost", "X", "U", "iters"):
        assert np.array_equal(e[key], one[key]), key
    assert e["total_cost"] == one["total_cost"] and e["stats"]["reg_events"] == one["stats"]["reg_events"]
    for strategy in ("centralized_dense", "centralized_dense_structured"):
        _same(e, oracle.solve_batch(mid, x0, None, T, dt, strategy=strategy, **kw))


def test_centralized_against_the_dense_stacked_goldens(gpu_lib):
    """mas_hip_run_centralized implements the block-diagonal reading of the stacked problem; the reference
    (centralized.hpp:18-38 on build_global_ocp, multi_agent_problem.hpp:94-122) finite-differences the SUMMED cost, so
    its stacked Hessians carry noise in the cross-agent blocks (~14 % of l_uu at 256 agents) and
    its answer is a different number.  Against the committed `centralized_dense` vectors (the oracle's literal stacked
    solve) at N = 2, 4, 10, T = 10 -- tolerance: total cost 5e-3 relative, per-agent cost 1e-2 relative, same iteration
    count (measured on the oracle pair: 2.6e-4 / 3.2e-5 / 1.3e-3 on the total, <= 4.9e-3 per agent)."""
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    gold = json.load(open(os.path.join(here, "golden", "oracle_vectors.json")))["cases"]
    seen = 0
    for name, c in gold.items():
        if c["strategy"] != "centralized_dense":
            continue
        seen += 1
        x0 = np.array(c["x0"])
        e = run_batch(gpu_lib, c["model"], x0, None, c["T"], c["dt"], "centralized", max_iterations=c["max_iterations"], tolerance=c["tolerance"])
        dense_cost = np.array([float.fromhex(v) for v in c["cost"]])
        dense_total = float.fromhex(c["total_cost"])
        assert abs(e["total_cost"] - dense_total) <= 5e-3 * dense_total, name
        assert (np.abs(e["cost"] - dense_cost) <= 1e-2 * dense_cost).all(), name
        assert e["iters"].tolist() == c["iters"], name
    assert seen == 3


def test_exit_codes_and_the_time_limit(gpu_lib, oracle):
    """ilqr.hpp:82-90,269-271: a solve ends on convergence, on max_iterations, or when elapsed_ms > max_ms at the top
    of an iteration.  max_ms = -1 is the one deterministic use of the clock (0 > -1: no iteration runs at all)."""
    from multi_agent_solver_amd.batch import EXIT_CONVERGED, EXIT_MAX_ITER, EXIT_TIME
    N, T, dt = 70, 40, 0.5
    x0 = oracle.single_track_x0(N)
    init = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, "single", max_iterations=0)
    assert (init["iters"] == 0).all() and (init["status"] == EXIT_MAX_ITER).all()          # the loop body never runs
    for strategy, outer in (("single", 1), ("sequential", 3), ("trustregion", 2)):
        for mi, ms, code in ((100, math.inf, EXIT_CONVERGED), (2, math.inf, EXIT_MAX_ITER), (100, -1.0, EXIT_TIME)):
            e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy, max_outer=outer, max_iterations=mi, max_ms=ms)
            o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy=strategy, max_outer=outer, max_iterations=mi, max_ms=ms)
            _same(e, o)
            # _same compared every agent's exit with the oracle's; here: the intended exit really occurs
            if ms < 0 or (strategy == "single" and mi == 100):
                assert (e["status"] == code).all(), (strategy, mi, ms, e["status"])
            elif strategy == "single":
                assert (e["status"] == code).sum() > N // 2 and set(e["status"].tolist()) <= {EXIT_CONVERGED, EXIT_MAX_ITER}
            if ms < 0:      # nothing ran: the initial rollout is the answer
                assert np.array_equal(e["X"], init["X"]) and np.array_equal(e["cost"], init["cost"]) and (e["iters"] == 0).all()
    # a generous finite budget changes nothing (per-solve clocks on the device, chained solves stay chained)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, "sequential", max_outer=3, max_ms=1e7)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy="sequential", max_outer=3)
    _same(e, o)
    # and a tiny one stops every agent early with EXIT_TIME or before (never more iterations than the unlimited run)
    t = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(2048), None, 100, dt, "single", max_ms=0.0)
    assert ((t["status"] & 3) == EXIT_TIME).sum() > 0 and (t["iters"] >= 1).all() and np.isfinite(t["cost"]).all()


def test_full_size_config3_properties_and_full_parity(gpu_lib, oracle):
    """BASELINE config 3: 4096 single-track agents, H=100, sequential, max_outer 10.
    Size-independent properties + exact oracle parity on ALL 4096 agents (the oracle solves them in seconds)."""
    from multi_agent_solver_amd.batch import BatchSolver
    N, T, dt = 4096, 100, 0.5
    x0 = oracle.single_track_x0(N)
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, N, T, dt, lib=gpu_lib) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(x0)
        s.set_controls(None)
        s.initialize_problem()
        init = s.solution()
        s.run_sequential(10)
        e = s.solution()
        st = s.stats()
        s.run_sequential(1)          # idempotence: one more round never increases any cost
        e2 = s.solution()
    assert np.isfinite(e["cost"]).all()
    assert (e["cost"] <= init["cost"]).all()                       # strict-decrease line search => monotone
    assert (e2["cost"] <= e["cost"]).all()
    assert (np.abs(e["U"]) <= 0.5).all()                           # clamp_controls
    assert np.array_equal(e["X"][:, 0, :], x0)
    assert st["agent_iterations"] == int(e["iters"].sum())
    assert (e["iters"] >= 10).all() and (e["iters"] <= 1000).all()
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy="sequential")
    assert np.array_equal(e["cost"], o["cost"])
    assert np.array_equal(e["X"], o["X"]) and np.array_equal(e["U"], o["U"])
    assert e["iters"].tolist() == o["iters"].tolist()
    assert e["status"].tolist() == o["status"].tolist()
    assert st["reg_events"] == o["stats"]["reg_events"] and st["converged_solves"] == o["stats"]["converged_solves"]
    assert e["total_cost"] == o["total_cost"]


def test_config2_centralized_256(gpu_lib, oracle):
    """BASELINE config 2: 256 agents centralised (block-diagonal reading), H=100."""
    N, T, dt = 256, 100, 0.5
    x0 = oracle.single_track_x0(N)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, "centralized")
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, dt, strategy="centralized")
    _same(e, o, total_exact=False)


def test_config4_lqr_shard(gpu_lib, oracle):
    """BASELINE config 4 per-GPU shard: 4096 identity-LQR agents, H=50 (identical agents => identical results)."""
    N, T, dt = 4096, 50, 0.1
    e = run_batch(gpu_lib, oracle.MODEL_LQR_IDENTITY, oracle.lqr_x0(N), None, T, dt, "sequential")
    o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, oracle.lqr_x0(N), None, T, dt, strategy="sequential")
    _same(e, o)
    assert (e["cost"] == e["cost"][0]).all() and (e["X"] == e["X"][0]).all()      # identical agents => identical results


def test_device_pointer_interface_with_torch(gpu_lib, oracle):
    import torch
    from multi_agent_solver_amd.batch import BatchSolver
    N, T = 64, 20
    x0 = oracle.single_track_x0(N)
    dev = torch.device("cuda:0")
    x0_t = torch.from_numpy(x0).to(dev)
    U_t = torch.zeros((N, T, 2), dtype=torch.float64, device=dev)
    X_out = torch.empty((N, T + 1, 4), dtype=torch.float64, device=dev)
    U_out = torch.empty((N, T, 2), dtype=torch.float64, device=dev)
    c_out = torch.empty((N,), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, N, T, 0.5, lib=gpu_lib) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states_device(x0_t.data_ptr())
        s.set_controls_device(U_t.data_ptr())
        s.initialize_problem()
        s.run_sequential(2)
        s.solution_device(X_out.data_ptr(), U_out.data_ptr(), c_out.data_ptr())
        host = s.solution()
    torch.cuda.synchronize()
    assert np.array_equal(X_out.cpu().numpy(), host["X"])
    assert np.array_equal(c_out.cpu().numpy(), host["cost"])


def test_torch_tensors_without_a_manual_synchronize(gpu_lib, oracle):
    """the handle works on its own stream.  The *_torch helpers order the handle's copies after torch's
    current stream (stream-ordering contract in include/mas_hip.h); here the inputs are produced by queued torch
    kernels and the outputs are freshly zero-filled, with no torch.cuda.synchronize() anywhere in the test."""
    import torch
    from multi_agent_solver_amd.batch import BatchSolver
    N, T = 512, 20
    x0 = oracle.single_track_x0(N)
    dev = torch.device("cuda:0")
    big = torch.ones((4096, 4096), device=dev)
    for _ in range(10):                                 # keep torch's stream busy while the tensors below are produced
        big = big @ big * 1e-4
    x0_t = torch.from_numpy(x0).to(dev) * 1.0
    U_t = torch.zeros((N, T, 2), dtype=torch.float64, device=dev) + 0.0
    with BatchSolver(oracle.MODEL_SINGLE_TRACK, N, T, 0.5, lib=gpu_lib) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states_torch(x0_t)
        s.set_controls_torch(U_t)
        s.initialize_problem()
        s.run_sequential(2)
        for _ in range(10):
            big = big @ big * 1e-4
        X_out = torch.zeros((N, T + 1, 4), dtype=torch.float64, device=dev)
        U_out = torch.zeros((N, T, 2), dtype=torch.float64, device=dev)
        c_out = torch.zeros((N,), dtype=torch.float64, device=dev)
        s.solution_torch(X_out, U_out, c_out)
        host = s.solution()
    assert np.array_equal(X_out.cpu().numpy(), host["X"]) and np.array_equal(U_out.cpu().numpy(), host["U"])
    assert np.array_equal(c_out.cpu().numpy(), host["cost"])
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="sequential", max_outer=2)
    assert np.array_equal(host["cost"], o["cost"])


def test_handle_on_the_callers_stream(gpu_lib, oracle):
    """mas_hip_create_on_stream: the handle works on torch's current stream, so its copies are ordered with the tensors'
    producers and consumers by the stream itself -- the other way to satisfy the *_device contract (include/mas_hip.h)"""
    import torch
    from multi_agent_solver_amd.batch import BatchSolver
    N, T = 256, 20
    x0 = oracle.single_track_x0(N)
    dev = torch.device("cuda:0")
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        x0_t = torch.from_numpy(x0).to(dev, non_blocking=True) + 0.0
        U_t = torch.zeros((N, T, 2), dtype=torch.float64, device=dev)
        with BatchSolver(oracle.MODEL_SINGLE_TRACK, N, T, 0.5, lib=gpu_lib, stream=stream.cuda_stream) as s:
            s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
            s.set_initial_states_device(x0_t.data_ptr())
            s.set_controls_device(U_t.data_ptr())
            s.initialize_problem()
            s.run_trust_region(2)
            X_out = torch.zeros((N, T + 1, 4), dtype=torch.float64, device=dev)
            c_out = torch.zeros((N,), dtype=torch.float64, device=dev)
            s.solution_device(X_out.data_ptr(), 0, c_out.data_ptr())
            total = (c_out.sum() * 1.0).item()
            host = s.solution()
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="trustregion", max_outer=2)
    assert np.array_equal(X_out.cpu().numpy(), o["X"]) and np.array_equal(c_out.cpu().numpy(), o["cost"]) and np.array_equal(host["U"], o["U"])
    assert abs(total - o["cost"].sum()) <= 1e-9 * o["cost"].sum()


def test_centralized_single_workgroup_fallback(gpu_lib, oracle):
    """when the workgroups of the rung-agreement kernel cannot all be resident (occupancy check), the
    centralised backward pass falls back to the single-workgroup kernel -- same results, bit for bit."""
    x0 = oracle.single_track_x0(70)
    fast = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 40, 0.5, "centralized")
    slow = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 40, 0.5, "centralized", options={"force_single_workgroup_central": 1})
    for key in ("cost", "X", "U", "iters"):
        assert np.array_equal(fast[key], slow[key]), key
    assert fast["stats"]["reg_events"] == slow["stats"]["reg_events"] and fast["total_cost"] == slow["total_cost"]


@pytest.mark.parametrize("strategy,outer", [("single", 1), ("sequential", 3), ("trustregion", 3)])
def test_constrained_single_track_augmented_lagrangian(gpu_lib, oracle, strategy, outer):
    """augmented-Lagrangian branch (ilqr.hpp:121-170,236-260,379-407) on this repository's constrained workload"""
    x0 = oracle.single_track_x0(70)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, 30, 0.5, strategy, max_outer=outer)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK_LIMITED, x0, None, 30, 0.5, strategy=strategy, max_outer=outer)
    _same(e, o)


def test_config5_bicycle6_shard_properties_and_full_parity(gpu_lib, oracle):
    """BASELINE config 5 per-GPU shard: 8192 bicycle6 agents (this repository's own nx=6 model), H=200,
    trust-region strategy, max_outer 10.  Properties + exact oracle parity on ALL 8192 agents (about a minute of
    oracle time on the GPU box's host cores)."""
    from multi_agent_solver_amd.batch import BatchSolver, bicycle6_x0
    N, T, dt = 8192, 200, 0.25
    x0 = bicycle6_x0(N)
    with BatchSolver(oracle.MODEL_BICYCLE6, N, T, dt, lib=gpu_lib) as s:
        s.set_params({"max_iterations": 100, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(x0)
        s.set_controls(None)
        s.initialize_problem()
        init = s.solution()
        s.run_trust_region(10)
        e = s.solution()
        radii = s.trust_radii()
    assert np.isfinite(e["cost"]).all() and (e["cost"] <= init["cost"]).all()      # accept only if the cost decreases (nash.hpp:228)
    assert (np.abs(e["U"]) <= 0.5).all()
    assert ((radii > 0) & np.isfinite(radii)).all()
    o = oracle.solve_batch(oracle.MODEL_BICYCLE6, x0, None, T, dt, strategy="trustregion")
    assert np.array_equal(e["cost"], o["cost"]) and np.array_equal(e["X"], o["X"]) and np.array_equal(e["U"], o["U"])
    assert e["iters"].tolist() == o["iters"].tolist()
    assert e["status"].tolist() == o["status"].tolist()


def test_maximum_sizes(gpu_lib, oracle):
    """the limits the ABI states: 1024 agents for the block-diagonal centralised strategy (1025 is refused, loudly), 64 for
    the literal one, and a 65536-agent batch (BASELINE configs[4]'s total on ONE device: 7 GB of 288) for the per-agent path"""
    from multi_agent_solver_amd._capi import MasHipError
    N, T = 1024, 12
    x0 = oracle.single_track_x0(N)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "centralized")
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, strategy="centralized")
    _same(e, o, total_exact=False)
    with pytest.raises(MasHipError):
        run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(1025), None, 5, 0.5, "centralized")
    with pytest.raises(MasHipError):
        run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(257), None, 5, 0.5, "centralized_dense")
    with pytest.raises(MasHipError):
        run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(65), None, 5, 0.5, "centralized_dense", options={"dense_chip": 0})
    N, T = 65536, 100
    x0 = oracle.single_track_x0(N)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, T, 0.5, "sequential", max_outer=2)
    sample = np.arange(0, N, 37)                              # 1772 agents re-solved alone by the oracle (agents are independent)
    o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0[sample], None, T, 0.5, strategy="sequential", max_outer=2)
    assert np.array_equal(e["cost"][sample], o["cost"]) and np.array_equal(e["X"][sample], o["X"]) and np.array_equal(e["U"][sample], o["U"])
    assert e["iters"][sample].tolist() == o["iters"].tolist()
    assert np.isfinite(e["cost"]).all() and (np.abs(e["U"]) <= 0.5).all() and e["stats"]["agent_iterations"] == int(e["iters"].sum())


@pytest.mark.gpu
def test_division_by_an_invariant_divisor_is_the_ieee_quotient():
    """div_by (mas_platform.hpp): q = p*y, r = fma(-q, b, p), fma(r, y, q), v_div_fixup -- against the device's own IEEE
    division, bit for bit, on 2 x 10 million random numerators per divisor (the finite-difference steps, the wheelbase, random
    ones) and the special values.  tests/hip/div_by_check.hip"""
    import subprocess
    exe = os.path.join(ROOT, "tests", "hip", "build", "div_by_check")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    r = subprocess.run([exe, "10"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, (r.stdout, r.stderr[-500:])


@pytest.mark.gpu
def test_sin_cos_tan_on_the_device_are_the_oracles_bits_and_within_an_ulp_of_glibc():
    """include/mas_dettrig.h evaluated on the DEVICE: (1) the same bits as the host evaluation of the same header -- what the CPU
    oracle's `det` mode computes, i.e. what every trigonometric parity test silently relies on -- and (2) <= 1 ulp (sin, cos) /
    <= 3 ulp (tan) from glibc, densely over the argument ranges the workloads reach (steering angles, headings of a lap, pendulum
    angles) and +-1e5.  tests/hip/dettrig_check.hip; tests/test_dettrig.py holds (2) for the CPU build only."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "hip", "build", "dettrig_check")
    assert os.path.exists(exe), "build it with __graft_entry__.build()"
    r = subprocess.run([exe, "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and " 0 mismatches between device and host evaluation" in r.stdout and "within 1 / 1 / 3" in r.stdout, (r.stdout, r.stderr[-500:])


@pytest.mark.parametrize("model,T,dt,strategy,outer", [("MODEL_SINGLE_TRACK", 60, 0.5, "sequential", 4), ("MODEL_SINGLE_TRACK", 40, 0.5, "trustregion", 3),
                                                       ("MODEL_SINGLE_TRACK", 40, 0.5, "centralized", 1), ("MODEL_LQR_IDENTITY", 30, 0.1, "sequential", 3),
                                                       ("MODEL_SINGLE_TRACK_LANE", 50, 0.1, "sequential", 2), ("MODEL_BICYCLE6", 80, 0.25, "trustregion", 3),
                                                       ("MODEL_SINGLE_TRACK", 100, 0.5, "trustregion", 10), ("MODEL_SINGLE_TRACK_LANE", 30, 0.1, "trustregion", 4)])
def test_kernel_forms_agree(gpu_lib, oracle, model, T, dt, strategy, outer):
    """The second half of round 2 added a second form of three kernels: line search / initial rollout / trust-region re-roll with
    four lanes per rollout (DPP quads) beside one lane per rollout, and the backward pass in registers + DPP (NX = 4) beside the
    LDS-tile form.  Which form runs depends on the number of live agents and the model; here the same perturbed problem is
    solved with the new forms (default: 330 agents are below the four-lane bound from round 0 on) and with the round-1 forms
    forced through mas_hip_set_option -- equal results (and both equal to the oracle).  Round 3: the default is also the two-kernel
    fused round (mas_fused_hip.hpp) for the four-state models; "fused": 0 gives the five-kernel round with the new forms.  For
    the six-state model the default at 330 agents is the LDS-tile kernel with 64 lanes per agent ("backward_wave_max": 0: 32 lanes, and
    "backward_wide_max": 0: 16 lanes).  Round 4: the chained trust-region strategy on the four-state models also runs two-kernel rounds, its
    epilogue (step norm, re-roll by dependency levels, accept / revert) inside the fused kernel (fused_trust_phase); "fused_trust": 0
    gives the five-kernel round with the epilogue as a kernel of its own."""
    mid = getattr(oracle, model)
    N = 330
    rng = np.random.default_rng(7)
    base = {"MODEL_LQR_IDENTITY": lambda: oracle.lqr_x0(N), "MODEL_BICYCLE6": lambda: oracle.bicycle6_x0(N)}.get(model, lambda: oracle.single_track_x0(N))()
    x0 = base + 0.05 * rng.standard_normal(base.shape)
    new = run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer)
    old = run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"fused": 0, "forward_quad_max": 0, "backward_tile": 1, "backward_wide_max": 0, "backward_wave_max": 0})
    unfused = run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"fused": 0})
    fused_one_lane = run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"forward_quad_max": 0})
    others = [old, unfused, fused_one_lane]
    if model == "MODEL_BICYCLE6":     # default here: a whole wavefront per agent in the backward pass; 32 lanes per agent; (old: 16)
        others.append(run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"backward_wave_max": 0}))
    if strategy == "trustregion":     # round 4: the chained epilogue as one kernel (re-roll by dependency levels) against norm / re-roll / copy kernels
        others.append(run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"trust_fused_max": 0}))
        others.append(run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"fused_trust": 0}))
        others.append(run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"trust_fused_max": 0, "trust_reroll_levels": 0}))   # (re-roll with four lanes per rollout)
        others.append(run_batch(gpu_lib, mid, x0, None, T, dt, strategy, max_outer=outer, options={"trust_fused_max": 0, "trust_reroll_levels": 0, "fused": 0, "forward_quad_max": 0}))
        assert all(np.array_equal(new["radii"], other["radii"]) for other in others)
    for other in others:
        for key in ("cost", "X", "U", "iters", "status"):
            assert np.array_equal(new[key], other[key]), key
        assert new["stats"]["reg_events"] == other["stats"]["reg_events"]
        assert new["stats"]["agent_iterations"] == other["stats"]["agent_iterations"] and new["stats"]["converged_solves"] == other["stats"]["converged_solves"]
    o = oracle.solve_batch(mid, x0, None, T, dt, strategy=strategy, max_outer=outer)
    _same(new, o, total_exact=(strategy != "centralized"))


@pytest.mark.parametrize("N,T,strategy,outer,kw", [(330, 30, "sequential", 3, {}), (1700, 50, "sequential", 2, {}), (1, 7, "single", 1, {}), (70, 1, "trustregion", 2, {}),
                                                   (65, 23, "linesearch", 2, {}), (200, 20, "centralized", 1, {}), (4096, 50, "sequential", 2, {}),
                                                   (300, 25, "sequential", 3, {"input_bounds": ([-0.3] * 4, [0.3] * 4)}), (300, 25, "sequential", 2, {"max_iterations": 2})])
def test_staged_line_search_of_the_identity_lqr(gpu_lib, oracle, N, T, strategy, outer, kw):
    """kStagedLineSearch: the per-agent line search of the identity LQR rolls out the first step size alone and the nine further ones only for
    agents whose first trial is no strict decrease (two launches instead of one with ten trials: a launch is bound by the bytes of its trial
    planes).  Same decisions as all ten at once ("forward_staged": 0) -- in particular in the rounds in which NO trial decreases the merit (every
    solve's last) and with input bounds that make the full step fail; the strategies that sum trial costs over agents keep all ten trials."""
    rng = np.random.default_rng(43)
    x0 = oracle.lqr_x0(N) + 0.1 * rng.standard_normal((N, 4))
    run_kw = {k: v for k, v in kw.items() if k == "max_iterations"}
    solver_kw = {}
    if "input_bounds" in kw:
        solver_kw["input_bounds"] = (np.array(kw["input_bounds"][0]), np.array(kw["input_bounds"][1]))
    new = run_batch(gpu_lib, oracle.MODEL_LQR_IDENTITY, x0, None, T, 0.1, strategy, max_outer=outer, **run_kw, **solver_kw)
    old = run_batch(gpu_lib, oracle.MODEL_LQR_IDENTITY, x0, None, T, 0.1, strategy, max_outer=outer, options={"forward_staged": 0}, **run_kw, **solver_kw)
    for key in ("cost", "X", "U", "iters", "status"):
        assert np.array_equal(new[key], old[key]), key
    for key in ("reg_events", "agent_iterations", "converged_solves", "solves"):
        assert new["stats"][key] == old["stats"][key], key
    if strategy in ("sequential", "trustregion", "single") and "input_bounds" not in kw:
        sample = np.arange(0, N, max(1, N // 40))
        o = oracle.solve_batch(oracle.MODEL_LQR_IDENTITY, x0[sample], None, T, 0.1, strategy=strategy, max_outer=outer, **run_kw)
        assert np.array_equal(new["cost"][sample], o["cost"]) and np.array_equal(new["X"][sample], o["X"]) and np.array_equal(new["U"][sample], o["U"])


@pytest.mark.parametrize("model,N,T,dt,outer", [("MODEL_SINGLE_TRACK", 2100, 41, 0.5, 3), ("MODEL_SINGLE_TRACK", 1, 7, 0.5, 2), ("MODEL_SINGLE_TRACK", 17, 3, 0.5, 2),
                                                ("MODEL_LQR_IDENTITY", 1700, 50, 0.1, 3), ("MODEL_SINGLE_TRACK_LANE", 1650, 30, 0.1, 2)])
def test_fused_round_agrees_across_the_line_search_forms(gpu_lib, oracle, model, N, T, dt, outer):
    """Two-kernel fused rounds (mas_fused_hip.hpp) on batches that start above the four-lane bound of the line search (trial
    planes indexed by agent) and fall below it as agents retire (planes indexed by slot): the commit inside the fused kernel
    has to find each agent's accepted trial in either layout.  Horizons that are not multiples of the helpers' four-step
    chunks, a single agent, a workgroup with one agent.  Against the five-kernel round and, on a sample, the oracle."""
    mid = getattr(oracle, model)
    rng = np.random.default_rng(5)
    base = oracle.lqr_x0(N) if model == "MODEL_LQR_IDENTITY" else oracle.single_track_x0(N)
    x0 = base + 0.05 * rng.standard_normal(base.shape)
    new = run_batch(gpu_lib, mid, x0, None, T, dt, "sequential", max_outer=outer)
    old = run_batch(gpu_lib, mid, x0, None, T, dt, "sequential", max_outer=outer, options={"fused": 0})
    for key in ("cost", "X", "U", "iters", "status"):
        assert np.array_equal(new[key], old[key]), key
    for key in ("reg_events", "agent_iterations", "converged_solves", "solves"):
        assert new["stats"][key] == old["stats"][key], key
    sample = np.arange(0, N, max(1, N // 60))
    o = oracle.solve_batch(mid, x0[sample], None, T, dt, strategy="sequential", max_outer=outer)
    assert np.array_equal(new["cost"][sample], o["cost"]) and np.array_equal(new["X"][sample], o["X"]) and np.array_equal(new["U"][sample], o["U"])
    assert new["iters"][sample].tolist() == o["iters"].tolist()


@pytest.mark.parametrize("model,N,T,dt,options", [("MODEL_LQR_IDENTITY", 200, 4, 0.1, {}), ("MODEL_LQR_IDENTITY", 200, 4, 0.1, {"dense_ladder_blocked": 0}),
                                                 ("MODEL_LQR_IDENTITY", 256, 3, 0.1, {}), ("MODEL_BICYCLE6", 140, 3, 0.25, {})])
def test_centralized_dense_wide_stacked_problems(gpu_lib, oracle, model, N, T, dt, options):
    """the chip-wide literal solve at its limits: 800 and 1024 stacked controls (two rows of the factor per thread in the LLT kernels and in
    LLT::solve, both ladder forms), 840 stacked states of the six-state model -- against the oracle's structured restatement"""
    mid = getattr(oracle, model)
    nx, _, _ = oracle.model_dims(mid)
    rng = np.random.default_rng(2)
    x0 = (oracle.lqr_x0(N) if model == "MODEL_LQR_IDENTITY" else oracle.bicycle6_x0(N)) + 0.1 * rng.standard_normal((N, nx))
    e = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized_dense", max_iterations=3, options=options)
    _same(e, oracle.solve_batch(mid, x0, None, T, dt, strategy="centralized_dense_structured", max_iterations=3))


@pytest.mark.parametrize("N", [2, 8, 16, 32, 64])
def test_both_centralised_readings_at_config2s_horizon(gpu_lib, oracle, N):
    """C2's horizon (H = 100): the reference's literal stacked solve on the device is bit-equal to the oracle's dense restatement (the plain
    one up to 16 agents, where it needs seconds; the structured one -- the same numbers, tests/test_centralized_structured.py -- at every
    size), the block-diagonal reading to the oracle's blocked one; how far the two readings lie apart there -- tens of per cent, inside
    the literal solve's own one-ulp band -- is tests/test_centralized_readings.py (CPU) and bench.py's C2 parity block"""
    x0 = oracle.single_track_x0(N)
    for strategy in ("centralized_dense", "centralized"):
        e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy)
        against = [strategy] if strategy == "centralized" else (["centralized_dense_structured"] + (["centralized_dense"] if N <= 16 else []))
        for name in against:
            o = oracle.solve_batch(oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy=name)
            assert np.array_equal(e["cost"], o["cost"]) and np.array_equal(e["X"], o["X"]) and np.array_equal(e["U"], o["U"]), name
            assert e["iters"].tolist() == o["iters"].tolist() and e["stats"]["reg_events"] == o["stats"]["reg_events"]
            if strategy == "centralized":
                assert abs(e["total_cost"] - o["total_cost"]) <= 1e-12 * abs(o["total_cost"])
            else:
                assert e["total_cost"] == o["total_cost"]


def test_literal_centralised_solve_against_the_committed_curve(gpu_lib, oracle):
    """The committed oracle curve (profiles/r3_c2_readings_curve.json, the PLAIN restatement: 99 s / 17 minutes of a core at 32 / 64 agents; its
    rows up to 16 agents are pinned to a fresh computation by tests/test_centralized_readings.py): total cost and iteration count of both
    readings on the device."""
    import json
    rows = {r["agents"]: r for r in json.load(open(os.path.join(ROOT, "profiles", "r3_c2_readings_curve.json")))["curve"]}
    for n in (32, 64):
        row = rows[n]
        x0 = oracle.single_track_x0(n)
        for strategy, total, iterations in (("centralized_dense", row["literal_total_cost"], row["iterations_literal"]),
                                            ("centralized", row["block_diagonal_total_cost"], row["iterations_block_diagonal"])):
            e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, x0, None, 100, 0.5, strategy)
            assert abs(e["total_cost"] - total) <= (0.0 if strategy == "centralized_dense" else 1e-12 * abs(total)), (strategy, e["total_cost"], total)
            assert int(e["iters"][0]) == iterations, strategy


@pytest.mark.parametrize("N,blocked", [(128, 1), (128, 0), (256, 1)])
def test_literal_centralised_solve_at_config2s_size(gpu_lib, oracle, N, blocked):
    """BASELINE configs[1] -- 256 agents, H = 100 -- under the reference's LITERAL algorithm: n = 1024 stacked states, m = 512 stacked controls,
    eight iterations of a dense Riccati recursion, 5 s on the device.  The oracle's structured restatement needs ten minutes of eight cores
    there, so the comparison is with its committed vectors (tests/golden/centralized_dense_large.json, tools/make_dense_goldens.py):
    per-agent costs and the total bit for bit, SHA-256 of the trajectories and controls, iterations, rungs of the ladder."""
    import hashlib
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "centralized_dense_large.json")))["cases"][str(N)]
    # (blocked: the ladder's sixteen LLTs factorised across the chip panel by panel, m >= 256; 0: one workgroup per rung, column by column)
    e = run_batch(gpu_lib, oracle.MODEL_SINGLE_TRACK, oracle.single_track_x0(N), None, g["horizon"], g["dt"], "centralized_dense", options={"dense_ladder_blocked": blocked})
    assert float(e["total_cost"]).hex() == g["total_cost"]
    assert [float(v).hex() for v in e["cost"]] == g["cost"]
    assert int(e["iters"][0]) == g["iterations"] and e["stats"]["reg_events"] == g["reg_events"]
    assert hashlib.sha256(np.ascontiguousarray(e["X"]).tobytes()).hexdigest() == g["sha256_X"]
    assert hashlib.sha256(np.ascontiguousarray(e["U"]).tobytes()).hexdigest() == g["sha256_U"]


@pytest.mark.parametrize("model,N,T", [("MODEL_SINGLE_TRACK", 10, 30), ("MODEL_SINGLE_TRACK", 70, 40), ("MODEL_SINGLE_TRACK", 600, 25),
                                       ("MODEL_LQR_IDENTITY", 40, 20), ("MODEL_SINGLE_TRACK_LANE", 90, 30)])
def test_centralized_speculative_pass_agrees(gpu_lib, oracle, model, N, T):
    """The centralised backward pass hides the rung agreement behind a speculative step (mas_backward_row_pipelined_kernel): one
    workgroup (N = 10), several on one XCD (70, 40, 90), more than an XCD holds (600); NU = 2 and NU = 4; perturbed agents, so that
    the agreed rung often differs from the one a wavefront assumed and steps are redone.  Same results as the plain kernel with
    the shared ladder, as the LDS-tile kernel, and as the oracle."""
    mid = getattr(oracle, model)
    rng = np.random.default_rng(11)
    base = oracle.lqr_x0(N) if model == "MODEL_LQR_IDENTITY" else oracle.single_track_x0(N)
    x0 = base + 0.05 * rng.standard_normal(base.shape)
    dt = 0.1 if model != "MODEL_SINGLE_TRACK" else 0.5
    new = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1)
    plain = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1, options={"central_pipeline": 0})
    tile = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1, options={"central_pipeline": 0, "backward_tile": 1})
    # round 4: an iteration is four launches -- the sweep commits what the iteration before accepted, the select publishes and clears the
    # agreement's words; "central_fold": 0 gives commit kernel + memset back.  Also with an iteration limit that ends the solve on an
    # ACCEPTED step (the last trial is committed by the copy behind the loop) and with a time limit (no iteration enqueued ahead).
    unfolded = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1, options={"central_fold": 0})
    for kw in ({"max_iterations": 2}, {"max_ms": 1e7}):
        a = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1, **kw)
        b = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1, options={"central_fold": 0}, **kw)
        assert all(np.array_equal(a[key], b[key]) for key in ("cost", "X", "U", "iters", "status")) and a["total_cost"] == b["total_cost"]
    two = oracle.solve_batch(mid, x0, None, T, dt, strategy="centralized", max_outer=1, max_iterations=2)
    a = run_batch(gpu_lib, mid, x0, None, T, dt, "centralized", max_outer=1, max_iterations=2)
    assert np.array_equal(a["X"], two["X"]) and np.array_equal(a["U"], two["U"]) and np.array_equal(a["cost"], two["cost"])
    for other in (plain, tile, unfolded):
        for key in ("cost", "X", "U", "iters"):
            assert np.array_equal(new[key], other[key]), key
        assert new["stats"]["reg_events"] == other["stats"]["reg_events"] and new["total_cost"] == other["total_cost"]
    o = oracle.solve_batch(mid, x0, None, T, dt, strategy="centralized", max_outer=1)
    _same(new, o, total_exact=False)
