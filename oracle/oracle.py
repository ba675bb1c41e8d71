"""ctypes wrapper around oracle/build/libmas_oracle.so -- CPU ORACLE (test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module, and only as the checker / CPU baseline.  The product package
(multi_agent_solver_amd) never imports it.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libmas_oracle.so")

MODEL_SINGLE_TRACK = 1
MODEL_LQR_IDENTITY = 2
MODEL_PENDULUM = 3
MODEL_SINGLE_TRACK_LANE = 4
MODEL_BICYCLE6 = 5
MODEL_SINGLE_TRACK_LIMITED = 6

STRATEGY = {"sequential": 0, "trustregion": 1, "linesearch": 2, "centralized_dense": 3,
            "centralized": 4, "single": 5, "centralized_dense_structured": 6}


class StrategyStats(C.Structure):
    _fields_ = [("total_iterations", C.c_long), ("converged_solves", C.c_long),
                ("reg_events", C.c_long), ("trials", C.c_long), ("alpha_hist", C.c_int * 12)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
            for f in ("mas_oracle.c", "mas_oracle_dense_impl.h", "mas_oracle.h", "ref_unit_cases.c", "Makefile", "../include/mas_dettrig.h")):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        d = C.POINTER(C.c_double)
        i = C.POINTER(C.c_int)
        _lib.mas_oracle_solve_batch.restype = C.c_double
        _lib.mas_oracle_solve_batch.argtypes = [C.c_int, d, C.c_int, C.c_int, C.c_double, d, d, d, d, i,
                                                C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                                C.POINTER(StrategyStats)]
        _lib.mas_model_dims.argtypes = [C.c_int, i, i, i]
        _lib.mas_model_default_params.argtypes = [C.c_int, C.c_int, d]
        _lib.mas_oracle_max_threads.restype = C.c_int
        _lib.mas_oracle_set_trig_mode.argtypes = [C.c_int]
        _lib.mas_oracle_set_arith_variant.argtypes = [C.c_int, C.c_int]
        _lib.mas_oracle_get_trig_mode.restype = C.c_int
        _lib.mas_oracle_set_al_params.argtypes = [C.c_double] * 4
        _lib.mas_oracle_set_exit_buffer.argtypes = [i]
        _lib.mas_oracle_set_radii_buffer.argtypes = [d]
        for name in ("mas_ref_case_initialize", "mas_ref_case_update_initial", "mas_ref_case_global",
                     "mas_ref_case_fd_gradient"):
            getattr(_lib, name).argtypes = [d]
            getattr(_lib, name).restype = C.c_int
        _lib.mas_model_make_ocp.restype = C.c_void_p
        _lib.mas_model_make_ocp.argtypes = [C.c_int, C.c_int, C.c_double, d, d, d, C.c_size_t]
        _lib.mas_ocp_free.argtypes = [C.c_void_p]
        _lib.mas_ilqr_backward_pass.restype = C.c_long
        _lib.mas_ilqr_backward_pass.argtypes = [C.c_void_p, d, d, d, d, d]
        _lib.mas_integrate_horizon.argtypes = [C.c_void_p, d, d, C.c_double, d]
        _lib.mas_ocp_objective.restype = C.c_double
        _lib.mas_ocp_objective.argtypes = [C.c_void_p, d, d]
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def model_dims(model_id: int):
    nx, nu, npar = C.c_int(), C.c_int(), C.c_int()
    if lib().mas_model_dims(model_id, C.byref(nx), C.byref(nu), C.byref(npar)):
        raise ValueError(f"unknown model {model_id}")
    return nx.value, nu.value, npar.value


def default_params(model_id: int, T: int) -> np.ndarray:
    p = np.zeros(16)
    lib().mas_model_default_params(model_id, T, _p(p))
    return p


_EXT_KEEP = []      # loaded shims and their callbacks stay alive
_EXT_IDS = {}       # (struct_source, struct_name) -> id: a workload registered again gets the id it has (the C table holds 64 entries)


def register_model(struct_source: str, struct_name: str) -> int:
    """A user-defined workload of the product (a struct deriving from mas::FdModel<Self, NX, NU>, the source mas_hip_register_model takes)
    compiled for the CPU: g++ with the oracle's own options over multi_agent_solver_amd/csrc/mas_models.hpp in its host form (MAS_HD = inline,
    `/` for div_by, sin / cos / tan from include/mas_dettrig.h), its three callbacks -- and the analytic derivative hooks it declares
    (kAnalyticA ... kAnalyticTerm, types.hpp:30-50) -- entered as an external model.  Returns the oracle's
    model id (>= 100) for solve_batch.  Test infrastructure: what it checks is the PRODUCT's treatment of the same struct."""
    import hashlib
    import tempfile
    if (struct_source, struct_name) in _EXT_IDS:
        return _EXT_IDS[(struct_source, struct_name)]
    root = os.path.dirname(_HERE)
    headers = b"".join(open(os.path.join(root, q), "rb").read() for q in ("multi_agent_solver_amd/csrc/mas_models.hpp", "multi_agent_solver_amd/csrc/mas_platform.hpp", "include/mas_dettrig.h"))
    tag = hashlib.sha256(b"shim v4" + (struct_source + struct_name).encode() + headers).hexdigest()[:16]
    d = os.path.join(tempfile.gettempdir(), f"mas_oracle_models_{os.getuid()}")
    os.makedirs(d, exist_ok=True)
    so, src = os.path.join(d, f"{struct_name}_{tag}.so"), os.path.join(d, f"{struct_name}_{tag}.{os.getpid()}.cpp")
    if not os.path.exists(so):
        with open(src, "w") as f:
            f.write(f"""#define MAS_HOST_EMULATION 1
#include <cstring>
#include "{root}/multi_agent_solver_amd/csrc/mas_models.hpp"
namespace mas {{
{struct_source}
}}
using M = mas::{struct_name};
static mas::ModelParams pack(const double* p) {{ mas::ModelParams P; std::memcpy(P.p, p, sizeof(P.p)); return P; }}
extern "C" void ext_dyn(const double* x, const double* u, double* d, const double* p) {{ M::dynamics(x, u, d, pack(p)); }}
extern "C" double ext_stage(const double* x, const double* u, int t, const double* p) {{ return M::stage_cost(x, u, t, pack(p)); }}
extern "C" double ext_term(const double* x, const double* p) {{ return M::terminal_cost(x, pack(p)); }}
extern "C" int ext_info(int* nx, int* nu, double* lb, double* ub) {{ *nx = M::NX; *nu = M::NU; M::default_bounds(lb, ub); return M::kDefaultBounds ? 1 : 0; }}
// the derivative hooks the struct provides (kAnalytic* = true): bit k of ext_hooks_mask() <-> entry k of the table handed to the oracle
// (templates, so that a hook the struct does not have is never named)
template <class Q> void hA(const double* x, const double* u, double* o, const double* p) {{ if constexpr (Q::kAnalyticA) Q::A(x, u, o, pack(p)); }}
template <class Q> void hB(const double* x, const double* u, double* o, const double* p) {{ if constexpr (Q::kAnalyticB) Q::B(x, u, o, pack(p)); }}
template <class Q> void hlx(const double* x, const double* u, int t, double* o, const double* p) {{ if constexpr (Q::kAnalyticLx) Q::lx(x, u, t, o, pack(p)); }}
template <class Q> void hlu(const double* x, const double* u, int t, double* o, const double* p) {{ if constexpr (Q::kAnalyticLu) Q::lu(x, u, t, o, pack(p)); }}
template <class Q> void hlxx(const double* x, const double* u, int t, double* o, const double* p) {{ if constexpr (Q::kAnalyticLxx) Q::lxx(x, u, t, o, pack(p)); }}
template <class Q> void hluu(const double* x, const double* u, int t, double* o, const double* p) {{ if constexpr (Q::kAnalyticLuu) Q::luu(x, u, t, o, pack(p)); }}
template <class Q> void hlux(const double* x, const double* u, int t, double* o, const double* p) {{ if constexpr (Q::kAnalyticLux) Q::lux(x, u, t, o, pack(p)); }}
template <class Q> void htg(const double* x, double* o, const double* p) {{ if constexpr (Q::kAnalyticTerm) Q::term_grad(x, o, pack(p)); }}
template <class Q> void hth(const double* x, double* o, const double* p) {{ if constexpr (Q::kAnalyticTerm) Q::term_hess(x, o, pack(p)); }}
extern "C" void ext_A(const double* x, const double* u, double* o, const double* p) {{ hA<M>(x, u, o, p); }}
extern "C" void ext_B(const double* x, const double* u, double* o, const double* p) {{ hB<M>(x, u, o, p); }}
extern "C" void ext_lx(const double* x, const double* u, int t, double* o, const double* p) {{ hlx<M>(x, u, t, o, p); }}
extern "C" void ext_lu(const double* x, const double* u, int t, double* o, const double* p) {{ hlu<M>(x, u, t, o, p); }}
extern "C" void ext_lxx(const double* x, const double* u, int t, double* o, const double* p) {{ hlxx<M>(x, u, t, o, p); }}
extern "C" void ext_luu(const double* x, const double* u, int t, double* o, const double* p) {{ hluu<M>(x, u, t, o, p); }}
extern "C" void ext_lux(const double* x, const double* u, int t, double* o, const double* p) {{ hlux<M>(x, u, t, o, p); }}
extern "C" void ext_tg(const double* x, double* o, const double* p) {{ htg<M>(x, o, p); }}
extern "C" void ext_th(const double* x, double* o, const double* p) {{ hth<M>(x, o, p); }}
extern "C" void ext_eq(const double* x, const double* u, double* o, const double* p) {{ M::equality(x, u, o, pack(p)); }}
extern "C" void ext_in(const double* x, const double* u, double* o, const double* p) {{ M::inequality(x, u, o, pack(p)); }}
extern "C" void ext_constraint_dims(int* neq, int* nin) {{ *neq = M::NEQ; *nin = M::NIN; }}
extern "C" int ext_hooks_mask() {{
  return (M::kAnalyticA ? 1 : 0) | (M::kAnalyticB ? 2 : 0) | (M::kAnalyticLx ? 4 : 0) | (M::kAnalyticLu ? 8 : 0) | (M::kAnalyticLxx ? 16 : 0) |
         (M::kAnalyticLuu ? 32 : 0) | (M::kAnalyticLux ? 64 : 0) | (M::kAnalyticTerm ? 128 + 256 : 0);
}}
""")
        fma = ["-mfma"] if any(" fma " in l + " " for l in open("/proc/cpuinfo") if l.startswith("flags")) else []
        tmp_so = f"{so}.{os.getpid()}.tmp"       # (several test processes may build the same shim: the finished file appears at once)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fno-fast-math", *fma, "-fPIC", "-shared", src, "-o", tmp_so])
        os.replace(tmp_so, so)
    shim = C.CDLL(so)
    nx, nu = C.c_int(), C.c_int()
    lb, ub = (C.c_double * 8)(), (C.c_double * 8)()
    shim.ext_info.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    bounded = shim.ext_info(C.byref(nx), C.byref(nu), lb, ub)
    f = lib().mas_oracle_register_external_model
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    mid = f(nx.value, nu.value, C.cast(shim.ext_dyn, C.c_void_p), C.cast(shim.ext_stage, C.c_void_p), C.cast(shim.ext_term, C.c_void_p),
            lb if bounded else None, ub if bounded else None)
    if mid < 0:
        raise RuntimeError("the oracle's external-model table is full")
    neq, nin = C.c_int(), C.c_int()
    shim.ext_constraint_dims(C.byref(neq), C.byref(nin))
    if neq.value or nin.value:
        g = lib().mas_oracle_external_set_constraints
        g.restype = C.c_int
        g.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        if g(mid, neq.value, C.cast(shim.ext_eq, C.c_void_p), nin.value, C.cast(shim.ext_in, C.c_void_p)) != 0:
            raise RuntimeError("mas_oracle_external_set_constraints failed")
    mask = shim.ext_hooks_mask()
    if mask:
        names = ("ext_A", "ext_B", "ext_lx", "ext_lu", "ext_lxx", "ext_luu", "ext_lux", "ext_tg", "ext_th")
        table = (C.c_void_p * 9)(*[C.cast(getattr(shim, n), C.c_void_p) if (mask >> k) & 1 else None for k, n in enumerate(names)])
        h = lib().mas_oracle_external_set_hooks
        h.restype = C.c_int
        h.argtypes = [C.c_int, C.c_void_p]
        if h(mid, C.cast(table, C.c_void_p)) != 0:
            raise RuntimeError("mas_oracle_external_set_hooks failed")
    _EXT_KEEP.append(shim)
    _EXT_IDS[(struct_source, struct_name)] = mid
    return mid


def set_trig_mode(mode: str) -> None:
    """'libm': platform sin/cos/tan, as a glibc-linked reference binary calls them (default).
    'det': include/mas_dettrig.h, the bit-reproducible implementation the HIP path uses."""
    lib().mas_oracle_set_trig_mode({"libm": 0, "det": 1}[mode])


def set_arith_variant(sum_order: str = "index", symmetrise: str = "aliased") -> None:
    """readings of Eigen the restatement cannot check (mas_oracle.c): products summed in index order or SSE2-packet-pairwise;
    V = 0.5 (V + V^T) aliased in place or as the mean of the old values.  Parity tests use the defaults."""
    lib().mas_oracle_set_arith_variant({"index": 0, "pairwise": 1}[sum_order], {"aliased": 0, "mean": 1}[symmetrise])


def get_trig_mode() -> str:
    return "det" if lib().mas_oracle_get_trig_mode() else "libm"


def set_al_params(penalty=10.0, penalty_increase=5.0, constraint_tolerance=1e-4, inequality_activation_tolerance=1e-6) -> None:
    """optional iLQR keys penalty / penalty_increase / constraint_tolerance / inequality_activation_tolerance (ilqr.hpp:47-54)"""
    lib().mas_oracle_set_al_params(penalty, penalty_increase, constraint_tolerance, inequality_activation_tolerance)


def max_threads() -> int:
    return lib().mas_oracle_max_threads()


def solve_batch(model_id, x0, U0, T, dt, strategy="sequential", max_outer=10, max_iterations=100,
                tolerance=1e-5, max_ms=math.inf, params=None, threads=0):
    """Run a strategy over N agents.  x0 [N,nx]; U0 [N,T,nu] (reference column layout: one
    control column per step) or None for zeros.  Returns dict(X [N,T+1,nx], U [N,T,nu], cost [N],
    iters [N], total_cost, stats) and, for the trust-region strategy, radii [N]."""
    nx, nu, _ = model_dims(model_id)
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, nx)
    N = x0.shape[0]
    U = np.zeros((N, T, nu)) if U0 is None else np.ascontiguousarray(U0, dtype=np.float64).reshape(N, T, nu).copy()
    X = np.zeros((N, T + 1, nx))
    cost = np.zeros(N)
    iters = np.zeros(N, dtype=np.int32)
    st = StrategyStats()
    par = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
    if par is not None and par.size < 16:
        par = np.concatenate([par, np.zeros(16 - par.size)])
    status = np.full(N, -1, dtype=np.int32)        # exit reason of each agent's last solve; stays -1 for the centralised strategies
    lib().mas_oracle_set_exit_buffer(status.ctypes.data_as(C.POINTER(C.c_int)))
    radii = np.ones(N) if strategy == "trustregion" else None   # final trust radii (nash.hpp:184,231,241)
    lib().mas_oracle_set_radii_buffer(_p(radii))
    try:
        total = lib().mas_oracle_solve_batch(model_id, _p(par), N, T, float(dt), _p(x0), _p(U), _p(X), _p(cost),
                                             iters.ctypes.data_as(C.POINTER(C.c_int)), STRATEGY[strategy], max_outer,
                                             max_iterations, float(tolerance), float(max_ms), threads, C.byref(st))
    finally:
        lib().mas_oracle_set_exit_buffer(None)
        lib().mas_oracle_set_radii_buffer(None)
    extra = {} if radii is None else {"radii": radii}
    return dict(X=X, U=U, cost=cost, iters=iters, status=status, total_cost=total, **extra,
                stats=dict(total_iterations=st.total_iterations, converged_solves=st.converged_solves,
                           reg_events=st.reg_events, trials=st.trials, alpha_hist=list(st.alpha_hist)))


def backward_pass(model_id, x0, U, T, dt, params=None):
    """Roll out U from x0, then one backward pass (linearisation + Riccati) for ONE agent.
    Returns X [T+1,nx], cost, k [T,nu], K [T,nu,nx] (K[t,i,j]), derivs dict, reg_events."""
    nx, nu, _ = model_dims(model_id)
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(nx)
    U = np.ascontiguousarray(U, dtype=np.float64).reshape(T, nu)
    par = None if params is None else np.ascontiguousarray(params, dtype=np.float64)
    o = lib().mas_model_make_ocp(model_id, T, float(dt), _p(x0), _p(U), _p(par), 0)
    try:
        X = np.zeros((T + 1, nx))
        lib().mas_integrate_horizon(o, _p(x0), _p(U), float(dt), _p(X))
        cost = lib().mas_ocp_objective(o, _p(X), _p(U))
        k = np.zeros((T, nu))
        Kc = np.zeros((T, nx, nu))  # column-major nu x nx per step
        ds = nx * nx + nx * nu + nx + nu + nx * nx + nu * nu + nu * nx
        D = np.zeros((T, ds))
        regs = lib().mas_ilqr_backward_pass(o, _p(X), _p(U), _p(k), _p(Kc), _p(D))
    finally:
        lib().mas_ocp_free(o)
    off = 0
    derivs = {}
    for name, (r, c) in (("A", (nx, nx)), ("B", (nx, nu)), ("lx", (nx, 1)), ("lu", (nu, 1)),
                         ("lxx", (nx, nx)), ("luu", (nu, nu)), ("lux", (nu, nx))):
        blk = D[:, off:off + r * c].reshape(T, c, r).transpose(0, 2, 1)  # col-major -> [t, row, col]
        derivs[name] = blk[:, :, 0] if c == 1 else blk
        off += r * c
    return dict(X=X, cost=cost, k=k, K=Kc.transpose(0, 2, 1), derivs=derivs, reg_events=regs)


def ref_case(name: str) -> np.ndarray:
    out = np.zeros(32)
    n = getattr(lib(), f"mas_ref_case_{name}")(_p(out))
    return out[:n].copy()


# ----------------------------------------------------------------- workloads --
def single_track_x0(N: int, track_radius: float = 20.0) -> np.ndarray:
    """examples/multi_agent_single_track.cpp:41-45,113-117: agent i at theta = 2*pi*i/N."""
    x0 = np.zeros((N, 4))
    for i in range(N):
        theta = 2.0 * math.pi * i / N
        x0[i] = (track_radius * math.cos(theta), track_radius * math.sin(theta), 1.57 + theta, 4.0)
    return x0


def lqr_x0(N: int, nx: int = 4) -> np.ndarray:
    x0 = np.zeros((N, nx))
    x0[:, 0] = 1.0  # examples/multi_agent_lqr.cpp:29-32
    return x0


def pendulum_problem(T: int = 60, dt: float = 0.05):
    """examples/pendulum_swing_up.cpp:36-44,105-113."""
    x0 = np.array([[math.pi - 0.05, 0.0]])
    torque_max = 5.0
    U0 = np.array([0.2 * torque_max * math.sin(2.0 * math.pi * (k * dt)) for k in range(T)]).reshape(1, T, 1)
    return x0, U0


def bicycle6_x0(N: int, track_radius: float = 20.0) -> np.ndarray:
    x0 = np.zeros((N, 6))
    st = single_track_x0(N, track_radius)
    x0[:, :4] = st
    return x0
