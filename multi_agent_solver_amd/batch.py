"""BatchSolver -- thin Python handle over the C ABI (include/mas_hip.h).

One BatchSolver = one `mas_hip_solver` = N agents of one registered workload model resident on
one GPU.  Arrays crossing this boundary use the reference layout (per agent one Eigen column
per time step): states [N, T+1, nx], controls [N, T, nu].
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import numpy as np

from . import _capi

MODEL_SINGLE_TRACK = 1
MODEL_LQR_IDENTITY = 2
MODEL_PENDULUM = 3
MODEL_SINGLE_TRACK_LANE = 4
MODEL_BICYCLE6 = 5
MODEL_SINGLE_TRACK_LIMITED = 6

EXIT_CONVERGED, EXIT_MAX_ITER, EXIT_TIME, EXIT_NONFINITE = 0, 1, 2, 4


def _dp(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


class Comm:
    """RCCL communicator made by the library (include/mas_hip.h: mas_hip_comm_*), one per rank / GPU.  `unique_id()` on one rank,
    the 128 bytes to the others by any channel (torch.distributed broadcast, a file, MPI), then `Comm(id, n_ranks, rank, device)`
    on every rank (collective)."""

    def __init__(self, unique_id: bytes, n_ranks: int, rank: int, device: int = 0, lib=None, _handle=None):
        self._lib = lib or _capi.product_lib()
        self.n_ranks, self.rank, self.device = n_ranks, rank, device
        if _handle is not None:                     # made by create_all
            self._h = _handle
            return
        buf = C.create_string_buffer(bytes(unique_id), 128)
        h = C.c_void_p()
        rc = self._lib.mas_hip_comm_create(C.cast(buf, C.c_void_p), n_ranks, rank, device, C.byref(h))
        if rc != _capi.OK:
            raise _capi.MasHipError(rc, (self._lib.mas_hip_comm_last_error() or b"").decode())
        self._h = h

    @staticmethod
    def create_all(devices, lib=None):
        """mas_hip_comm_create_all (ncclCommInitAll): one process, one communicator per entry of `devices`, each to be driven from its own thread"""
        lib = lib or _capi.product_lib()
        n = len(devices)
        devs = (C.c_int * n)(*devices)
        out = (C.c_void_p * n)()
        rc = lib.mas_hip_comm_create_all(n, devs, out)
        if rc != _capi.OK:
            raise _capi.MasHipError(rc, (lib.mas_hip_comm_last_error() or b"").decode())
        return [Comm(b"", n, r, devices[r], lib=lib, _handle=C.c_void_p(out[r])) for r in range(n)]

    @staticmethod
    def use_loopback(enabled: bool, lib=None):
        """communicators made after this call use the in-process loopback transport (tests; include/mas_hip.h)"""
        (lib or _capi.product_lib()).mas_hip_comm_use_loopback(1 if enabled else 0)

    def abort(self):
        """release every peer waiting for this rank inside a collective (ncclCommAbort); later collectives fail at once; close() still frees it"""
        if self._h:
            self._lib.mas_hip_comm_abort(self._h)

    @staticmethod
    def unique_id(lib=None) -> bytes:
        lib = lib or _capi.product_lib()
        buf = C.create_string_buffer(128)
        rc = lib.mas_hip_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != _capi.OK:
            raise _capi.MasHipError(rc, (lib.mas_hip_comm_last_error() or b"").decode())
        return buf.raw

    def shard_range(self, n_total: int, rank: int = None):
        lo, hi = C.c_int(), C.c_int()
        self._lib.mas_hip_shard_range(n_total, self.n_ranks, self.rank if rank is None else rank, C.byref(lo), C.byref(hi))
        return lo.value, hi.value

    def close(self):
        if self._h:
            self._lib.mas_hip_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class BatchSolver:
    def __init__(self, model_id: int, n_agents: int, horizon_steps: int, dt: float, params=None,
                 input_bounds=None, no_input_bounds: bool = False, device: int = 0, stream: int = 0, lib=None):
        self._lib = lib if lib is not None else _capi.product_lib()
        self._h = C.c_void_p()
        nx, nu = C.c_int(), C.c_int()
        if self._lib.mas_hip_model_dims(model_id, C.byref(nx), C.byref(nu)) != _capi.OK:
            raise ValueError(f"unknown model id {model_id}")
        self.model_id, self.N, self.T, self.dt = model_id, int(n_agents), int(horizon_steps), float(dt)
        self.nx, self.nu = nx.value, nu.value
        desc = _capi.ProblemDesc()
        desc.model_id, desc.n_agents, desc.horizon_steps, desc.device, desc.dt = model_id, n_agents, horizon_steps, device, dt
        self._keep = []
        if params is not None:
            p = np.zeros(16)
            p[:len(params)] = np.asarray(params, dtype=np.float64)
            self._keep.append(p)
            desc.params = _dp(p)
        if input_bounds is not None:
            lb = np.ascontiguousarray(input_bounds[0], dtype=np.float64)
            ub = np.ascontiguousarray(input_bounds[1], dtype=np.float64)
            self._keep += [lb, ub]
            desc.input_lower_bounds, desc.input_upper_bounds = _dp(lb), _dp(ub)
        desc.no_input_bounds = 1 if no_input_bounds else 0
        if stream:
            rc = self._lib.mas_hip_create_on_stream(C.byref(desc), C.c_void_p(stream), C.byref(self._h))
        else:
            rc = self._lib.mas_hip_create(C.byref(desc), C.byref(self._h))
        if rc != _capi.OK:
            msg = self._lib.mas_hip_last_error(None)
            self._h = C.c_void_p()
            raise _capi.MasHipError(rc, msg.decode() if msg else "create failed")

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._lib.mas_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != _capi.OK:
            msg = self._lib.mas_hip_last_error(self._h)
            raise _capi.MasHipError(rc, msg.decode() if msg else "")

    # -- agents sharded over GPUs: RCCL inside the library (include/mas_hip.h) ---------------
    def all_gather_solution_device(self, comm: "Comm", n_total: int, X_all_ptr: int, U_all_ptr: int, cost_all_ptr: int):
        """every rank receives every agent's trajectory: DEVICE buffers [n_total][T+1][nx], [n_total][T][nu], [n_total] (0 = skip)"""
        self._check(self._lib.mas_hip_all_gather_solution(self._h, comm._h, n_total, C.c_void_p(X_all_ptr or None), C.c_void_p(U_all_ptr or None),
                                                          C.c_void_p(cost_all_ptr or None)))

    def set_comm_cost_exchange(self, comm, n_total: int = 0):
        """line-search strategy: total cost over all ranks' agents formed on the device (None: local sum)"""
        self._check(self._lib.mas_hip_set_comm_cost_exchange(self._h, comm._h if comm is not None else None, n_total))

    def set_option(self, key: str, value: int):
        """choose between bit-identical kernel forms (include/mas_hip.h: mas_hip_set_option); results never depend on it"""
        self._check(self._lib.mas_hip_set_option(self._h, key.encode(), int(value)))

    # -- parameters (iLQR::set_params, ilqr.hpp:39-55) ------------------------------------
    def set_params(self, params: dict):
        for key in ("max_iterations", "tolerance", "max_ms"):
            if key not in params:   # params.at(key) throws std::out_of_range in the reference
                raise KeyError(key)
        for key, value in params.items():
            self._check(self._lib.mas_hip_set_param(self._h, key.encode(), float(value)))

    # -- problem data --------------------------------------------------------------------
    def set_initial_states(self, x0):
        x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(self.N, self.nx)
        self._check(self._lib.mas_hip_set_initial_states_host(self._h, _dp(x0)))

    def set_controls(self, U=None):
        if U is None:
            self._check(self._lib.mas_hip_set_controls_host(self._h, None))
        else:
            U = np.ascontiguousarray(U, dtype=np.float64).reshape(self.N, self.T, self.nu)
            self._check(self._lib.mas_hip_set_controls_host(self._h, _dp(U)))

    # Device-pointer entry points -- stream-ordering contract (include/mas_hip.h): the handle runs on its own stream.  The
    # caller makes the buffer's contents ready before the call (e.g. torch.cuda.current_stream().synchronize(), or create
    # the BatchSolver with stream=torch.cuda.current_stream().cuda_stream); on return the handle has finished with it.
    def set_initial_states_device(self, ptr: int):
        self._check(self._lib.mas_hip_set_initial_states_device(self._h, C.c_void_p(ptr)))

    def set_initial_states_torch(self, t):
        """x0 from a torch CUDA tensor [N, nx] (float64, contiguous), ordered after the work queued on torch's current stream"""
        import torch
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == (self.N, self.nx)
        torch.cuda.current_stream(t.device).synchronize()
        self.set_initial_states_device(t.data_ptr())

    def set_controls_torch(self, t):
        import torch
        assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and tuple(t.shape) == (self.N, self.T, self.nu)
        torch.cuda.current_stream(t.device).synchronize()
        self.set_controls_device(t.data_ptr())

    def solution_torch(self, X=None, U=None, cost=None):
        """write the Solution into preallocated torch CUDA tensors (any may be None); complete on return"""
        import torch
        dev = next(t.device for t in (X, U, cost) if t is not None)
        torch.cuda.current_stream(dev).synchronize()       # pending torch writes to the buffers (e.g. their zero-fill) come first
        self.solution_device(X.data_ptr() if X is not None else 0, U.data_ptr() if U is not None else 0, cost.data_ptr() if cost is not None else 0)

    def set_controls_device(self, ptr: int):
        self._check(self._lib.mas_hip_set_controls_device(self._h, C.c_void_p(ptr)))

    def initialize_problem(self):
        self._check(self._lib.mas_hip_initialize_problem(self._h))

    def save_initial(self):
        self._check(self._lib.mas_hip_save_initial(self._h))

    def restore_initial(self):
        self._check(self._lib.mas_hip_restore_initial(self._h))

    # -- agents sharded over ranks ---------------------------------------------------------
    def set_cost_exchange(self, fn):
        """fn(local_costs: np.ndarray[n_local]) -> float: total cost over ALL ranks' agents, summed serially in global
        block order (mas_hip_set_cost_exchange; nash.hpp:45-49).  None restores the local sum."""
        if fn is None:
            self._exchange_cb = None
            self._check(self._lib.mas_hip_set_cost_exchange(self._h, _capi.COST_EXCHANGE_FN(0), None))
            return

        def _cb(_ctx, costs, n, out):
            try:
                out[0] = float(fn(np.ctypeslib.as_array(costs, shape=(n,)).copy()))
                return 0
            except Exception:   # never let an exception cross the C boundary
                return 1

        self._exchange_cb = _capi.COST_EXCHANGE_FN(_cb)   # keep the trampoline alive as long as the handle uses it
        self._check(self._lib.mas_hip_set_cost_exchange(self._h, self._exchange_cb, None))

    # -- hot path ------------------------------------------------------------------------
    def solve(self):
        self._check(self._lib.mas_hip_solve(self._h))

    def run_sequential(self, max_outer: int):
        self._check(self._lib.mas_hip_run_sequential(self._h, int(max_outer)))

    def run_trust_region(self, max_outer: int):
        self._check(self._lib.mas_hip_run_trust_region(self._h, int(max_outer)))

    def run_line_search(self, max_outer: int):
        self._check(self._lib.mas_hip_run_line_search(self._h, int(max_outer)))

    def run_centralized(self):
        self._check(self._lib.mas_hip_run_centralized(self._h))

    def run_centralized_dense(self):
        """the reference's literal stacked solve (FD of the summed cost, dense Riccati recursion), N <= 256: chip-wide launches (csrc/mas_dense_chip.hpp)"""
        self._check(self._lib.mas_hip_run_centralized_dense(self._h))

    def run(self, strategy: str, max_outer: int = 10):
        if strategy == "sequential":
            self.run_sequential(max_outer)
        elif strategy == "trustregion":
            self.run_trust_region(max_outer)
        elif strategy == "linesearch":
            self.run_line_search(max_outer)
        elif strategy == "centralized":
            self.run_centralized()
        elif strategy == "centralized_dense":
            self.run_centralized_dense()
        elif strategy == "single":
            self.solve()
        else:
            raise ValueError(f"Unknown strategy '{strategy}'.")

    # -- results -------------------------------------------------------------------------
    def solution(self):
        X = np.zeros((self.N, self.T + 1, self.nx))
        U = np.zeros((self.N, self.T, self.nu))
        cost = np.zeros(self.N)
        iters = np.zeros(self.N, dtype=np.int32)
        status = np.zeros(self.N, dtype=np.int32)
        i32 = C.POINTER(C.c_int32)
        self._check(self._lib.mas_hip_get_solution_host(self._h, _dp(X), _dp(U), _dp(cost), iters.ctypes.data_as(i32),
                                                        status.ctypes.data_as(i32)))
        return dict(X=X, U=U, cost=cost, iters=iters, status=status, total_cost=self.total_cost())

    def solution_device(self, X_ptr: int = 0, U_ptr: int = 0, cost_ptr: int = 0):
        self._check(self._lib.mas_hip_get_solution_device(self._h, C.c_void_p(X_ptr), C.c_void_p(U_ptr), C.c_void_p(cost_ptr)))

    def total_cost(self) -> float:
        v = C.c_double()
        self._check(self._lib.mas_hip_total_cost(self._h, C.byref(v)))
        return v.value

    def trust_radii(self) -> np.ndarray:
        r = np.zeros(self.N)
        self._check(self._lib.mas_hip_get_trust_radii_host(self._h, _dp(r)))
        return r

    def debug_backward(self):
        """One linearisation sweep + backward pass on the current trajectory (parity tests)."""
        nx, nu, T, N = self.nx, self.nu, self.T, self.N
        nd = 2 * nx * nx + 2 * nx * nu + nu * nu + nx + nu
        D = np.zeros((N, T, nd))
        k = np.zeros((N, T, nu))
        Kc = np.zeros((N, T, nx, nu))
        regs = C.c_int64()
        self._check(self._lib.mas_hip_debug_backward_host(self._h, _dp(D), _dp(k), _dp(Kc), C.byref(regs)))
        derivs, off = {}, 0
        for name, (r, c) in (("A", (nx, nx)), ("B", (nx, nu)), ("lx", (nx, 1)), ("lu", (nu, 1)),
                             ("lxx", (nx, nx)), ("luu", (nu, nu)), ("lux", (nu, nx))):
            blk = D[:, :, off:off + r * c].reshape(N, T, c, r).transpose(0, 1, 3, 2)
            derivs[name] = blk[..., 0] if c == 1 else blk
            off += r * c
        return dict(derivs=derivs, k=k, K=Kc.transpose(0, 1, 3, 2), reg_events=regs.value)

    # -- measurement ---------------------------------------------------------------------
    def stats(self) -> dict:
        st = _capi.Stats()
        self._check(self._lib.mas_hip_get_stats(self._h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def reset_stats(self):
        self._check(self._lib.mas_hip_reset_stats(self._h))

    def set_profiling(self, enabled: bool):
        self._check(self._lib.mas_hip_set_profiling(self._h, 1 if enabled else 0))

    def kernel_times(self) -> dict:
        buf = (_capi.KernelTime * 64)()
        n = C.c_int()
        self._check(self._lib.mas_hip_get_kernel_times(self._h, buf, 64, C.byref(n)))
        return {buf[i].name.decode(): dict(launches=buf[i].launches, total_ms=buf[i].total_ms,
                                           algorithmic_bytes=buf[i].algorithmic_bytes) for i in range(min(n.value, 64))}


def register_model(struct_source: str, struct_name: str, source_root: Optional[str] = None, lib=None) -> int:
    """mas_hip_register_model: compile a user-defined workload (a struct deriving from mas::FdModel<Self, NX, NU>, csrc/mas_models.hpp)
    against the kernel templates with the machine's hipcc, load it, and return its model id for BatchSolver(model_id, ...)."""
    lib = lib or _capi.load()
    mid = C.c_int()
    rc = lib.mas_hip_register_model(struct_source.encode(), struct_name.encode(), source_root.encode() if source_root else None, C.byref(mid))
    if rc != _capi.OK:
        raise _capi.MasHipError(rc, (lib.mas_hip_last_error(None) or b"").decode())
    return mid.value


# ----------------------------------------------------------------- workload generators --
def single_track_x0(N: int, track_radius: float = 20.0, first: int = 0, total: Optional[int] = None) -> np.ndarray:
    """Agent i of `total` at theta = 2*pi*i/total (examples/multi_agent_single_track.cpp:41-45,113-117);
    rows are agents first .. first+N-1 (a shard of the id-sorted block list)."""
    total = N if total is None else total
    x0 = np.zeros((N, 4))
    for r in range(N):
        theta = 2.0 * math.pi * (first + r) / total
        x0[r] = (track_radius * math.cos(theta), track_radius * math.sin(theta), 1.57 + theta, 4.0)
    return x0


def lqr_x0(N: int, nx: int = 4) -> np.ndarray:
    x0 = np.zeros((N, nx))
    x0[:, 0] = 1.0   # examples/multi_agent_lqr.cpp:29-32
    return x0


def pendulum_problem(T: int = 60, dt: float = 0.05):
    """examples/pendulum_swing_up.cpp:36-44,105-113"""
    x0 = np.array([[math.pi - 0.05, 0.0]])
    torque_max = 5.0
    U0 = np.array([0.2 * torque_max * math.sin(2.0 * math.pi * (k * dt)) for k in range(T)]).reshape(1, T, 1)
    return x0, U0


def bicycle6_x0(N: int, track_radius: float = 20.0, first: int = 0, total: Optional[int] = None) -> np.ndarray:
    x0 = np.zeros((N, 6))
    x0[:, :4] = single_track_x0(N, track_radius, first, total)
    return x0
