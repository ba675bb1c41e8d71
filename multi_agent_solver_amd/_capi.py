"""ctypes binding of include/mas_hip.h (libmas_hip.so).

The product library is `multi_agent_solver_amd/lib/libmas_hip.so`, built by
`__graft_entry__.build()` (hipcc, gfx950).  If it is missing or cannot be loaded this module
raises -- there is no CPU fallback and nothing here imports the oracle.

`load(path)` exists so the test-only host emulation (tests/hostcheck) can be driven through the
very same binding; the package itself never calls it with anything but the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmas_hip.so")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_RUNTIME, ERR_STATE = 0, 1, 2, 3, 4

EXPORTS = [
    "mas_hip_abi_version", "mas_hip_device_count", "mas_hip_model_dims", "mas_hip_model_default_params", "mas_hip_register_model",
    "mas_hip_create", "mas_hip_create_on_stream", "mas_hip_destroy", "mas_hip_last_error",
    "mas_hip_set_param", "mas_hip_check_params", "mas_hip_set_option",
    "mas_hip_set_initial_states_host", "mas_hip_set_controls_host",
    "mas_hip_set_initial_states_device", "mas_hip_set_controls_device",
    "mas_hip_initialize_problem", "mas_hip_save_initial", "mas_hip_restore_initial",
    "mas_hip_solve", "mas_hip_run_sequential", "mas_hip_run_trust_region", "mas_hip_run_line_search",
    "mas_hip_run_centralized", "mas_hip_run_centralized_dense",
    "mas_hip_get_solution_host", "mas_hip_get_solution_device", "mas_hip_total_cost",
    "mas_hip_set_cost_exchange", "mas_hip_get_trust_radii_host", "mas_hip_debug_backward_host",
    "mas_hip_get_stats", "mas_hip_reset_stats", "mas_hip_set_profiling", "mas_hip_get_kernel_times",
    "mas_hip_comm_unique_id", "mas_hip_comm_create", "mas_hip_comm_create_all", "mas_hip_comm_destroy", "mas_hip_comm_abort", "mas_hip_comm_use_loopback", "mas_hip_comm_last_error",
    "mas_hip_shard_range", "mas_hip_all_gather_solution", "mas_hip_set_comm_cost_exchange",
]


class ProblemDesc(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("n_agents", C.c_int32), ("horizon_steps", C.c_int32),
                ("device", C.c_int32), ("dt", C.c_double), ("params", C.POINTER(C.c_double)),
                ("input_lower_bounds", C.POINTER(C.c_double)), ("input_upper_bounds", C.POINTER(C.c_double)),
                ("no_input_bounds", C.c_int32), ("reserved", C.c_int32)]


# int (*mas_hip_cost_exchange_fn)(void* ctx, const double* local_costs, int32_t n_local, double* global_total)
COST_EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double))


class Stats(C.Structure):
    _fields_ = [("agent_iterations", C.c_int64), ("converged_solves", C.c_int64), ("solves", C.c_int64),
                ("reg_events", C.c_int64), ("batch_iterations", C.c_int64), ("wall_ms", C.c_double)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_int64), ("total_ms", C.c_double),
                ("algorithmic_bytes", C.c_double)]


class MasHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"mas_hip status {code}: {message}")
        self.code = code


def _preload_torch_hip_runtime() -> None:
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Two HIP runtimes
    in one process cannot both initialise the GPU, so when torch is installed its copy is loaded first
    (RTLD_GLOBAL, without importing torch) and libmas_hip.so then binds to it by SONAME.  The same holds for
    librccl.so.1, which libmas_hip.so opens on first use of its mas_hip_comm_* entry points: torch's copy first, so that the
    library and a later `import torch` share ONE RCCL built against that one HIP runtime."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    # libamdhip64 globally (libmas_hip.so binds its HIP symbols to it); librccl LOCALLY: it only has to be the instance a later
    # dlopen("librccl.so.1") finds by SONAME -- with its symbols in the global scope, a process that also holds torch and an OpenMP
    # library aborted at exit (double free)
    for name, mode in (("libamdhip64.so", C.RTLD_GLOBAL), ("librccl.so", C.RTLD_LOCAL)):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=mode)
            except OSError:
                if name == "libamdhip64.so":
                    raise


def load(path: str = LIB_PATH) -> C.CDLL:
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); multi_agent_solver_amd has no CPU fallback")
    if os.path.basename(path) == "libmas_hip.so":
        _preload_torch_hip_runtime()
    lib = C.CDLL(path)
    d, vp, i32 = C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int32)
    lib.mas_hip_last_error.restype = C.c_char_p
    lib.mas_hip_last_error.argtypes = [vp]
    lib.mas_hip_model_dims.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mas_hip_register_model.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_int)]
    lib.mas_hip_model_default_params.argtypes = [C.c_int, C.c_int, d]
    lib.mas_hip_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(vp)]
    lib.mas_hip_create_on_stream.argtypes = [C.POINTER(ProblemDesc), vp, C.POINTER(vp)]
    lib.mas_hip_destroy.argtypes = [vp]
    lib.mas_hip_set_param.argtypes = [vp, C.c_char_p, C.c_double]
    lib.mas_hip_check_params.argtypes = [vp]
    lib.mas_hip_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.mas_hip_set_initial_states_host.argtypes = [vp, d]
    lib.mas_hip_set_controls_host.argtypes = [vp, d]
    lib.mas_hip_set_initial_states_device.argtypes = [vp, vp]
    lib.mas_hip_set_controls_device.argtypes = [vp, vp]
    for name in ("mas_hip_initialize_problem", "mas_hip_save_initial", "mas_hip_restore_initial", "mas_hip_solve",
                 "mas_hip_run_centralized", "mas_hip_run_centralized_dense", "mas_hip_reset_stats"):
        getattr(lib, name).argtypes = [vp]
    for name in ("mas_hip_run_sequential", "mas_hip_run_trust_region", "mas_hip_run_line_search",
                 "mas_hip_set_profiling"):
        getattr(lib, name).argtypes = [vp, C.c_int]
    lib.mas_hip_get_solution_host.argtypes = [vp, d, d, d, i32, i32]
    lib.mas_hip_get_solution_device.argtypes = [vp, vp, vp, vp]
    lib.mas_hip_total_cost.argtypes = [vp, d]
    lib.mas_hip_set_cost_exchange.argtypes = [vp, COST_EXCHANGE_FN, C.c_void_p]
    lib.mas_hip_get_trust_radii_host.argtypes = [vp, d]
    lib.mas_hip_debug_backward_host.argtypes = [vp, d, d, d, C.POINTER(C.c_int64)]
    lib.mas_hip_get_stats.argtypes = [vp, C.POINTER(Stats)]
    lib.mas_hip_get_kernel_times.argtypes = [vp, C.POINTER(KernelTime), C.c_int, C.POINTER(C.c_int)]
    lib.mas_hip_comm_unique_id.argtypes = [vp]
    lib.mas_hip_comm_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    lib.mas_hip_comm_create_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(vp)]
    lib.mas_hip_comm_destroy.argtypes = [vp]
    lib.mas_hip_comm_abort.argtypes = [vp]
    lib.mas_hip_comm_use_loopback.argtypes = [C.c_int]
    lib.mas_hip_comm_last_error.restype = C.c_char_p
    lib.mas_hip_shard_range.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.mas_hip_shard_range.restype = None
    lib.mas_hip_all_gather_solution.argtypes = [vp, vp, C.c_int, vp, vp, vp]
    lib.mas_hip_set_comm_cost_exchange.argtypes = [vp, vp, C.c_int]
    return lib


_product_lib = None


def product_lib() -> C.CDLL:
    global _product_lib
    if _product_lib is None:
        _product_lib = load(LIB_PATH)
    return _product_lib
