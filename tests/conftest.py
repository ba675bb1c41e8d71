import math
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): oracle/build/libmas_oracle.so via ctypes."""
    from oracle import oracle as O
    O.build()
    # "det": sin/cos/tan from include/mas_dettrig.h, the same instruction sequence the HIP path runs, so
    # product-vs-oracle comparisons are bit-exact.  Tests that want the glibc-faithful mode switch
    # explicitly (tests/test_oracle_sensitivity.py) and switch back.
    O.set_trig_mode("det")
    return O


@pytest.fixture(scope="session")
def hostcheck_lib():
    """TEST-ONLY host emulation of the kernels' lane bodies (g++ build of the product sources
    with -DMAS_HOST_EMULATION); lets the driver logic and arithmetic be checked without a GPU."""
    import __graft_entry__ as g
    from multi_agent_solver_amd import _capi
    return _capi.load(g.build_hostcheck())


@pytest.fixture(scope="session")
def product_lib():
    import __graft_entry__ as g
    from multi_agent_solver_amd import _capi
    return _capi.load(g.build_hip())


def run_batch(lib, model, x0, U0, T, dt, strategy, max_outer=10, max_iterations=100, tolerance=1e-5, max_ms=math.inf, options=None, **kw):
    """options: kernel forms (mas_hip_set_option) -- bit-identical alternatives of the same launches"""
    from multi_agent_solver_amd.batch import BatchSolver
    with BatchSolver(model, x0.shape[0], T, dt, lib=lib, **kw) as s:
        for key, value in (options or {}).items():
            s.set_option(key, value)
        s.set_params({"max_iterations": max_iterations, "tolerance": tolerance, "max_ms": max_ms})
        s.set_initial_states(x0)
        s.set_controls(U0)
        s.initialize_problem()
        s.run(strategy, max_outer)
        out = s.solution()
        out["stats"] = s.stats()
        if strategy == "trustregion":
            out["radii"] = s.trust_radii()
    return out
