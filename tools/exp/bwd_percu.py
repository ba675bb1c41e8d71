"""experiment: is the backward pass slowed by sharing a CU?  1024 agents as 256 single-wavefront workgroups (one per CU) against 64
workgroups of 4 wavefronts (four per CU on 64 CUs)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from multi_agent_solver_amd import _capi
from multi_agent_solver_amd.batch import BatchSolver, single_track_x0
lib = _capi.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "libmas_hip_bwdexp.so"))
for N in (1024, 2048, 4096):
    for wpb4 in (False, True):
        if wpb4: os.environ["MAS_EXP_BWD_WPB4"] = "1"
        else: os.environ.pop("MAS_EXP_BWD_WPB4", None)
        s = BatchSolver(1, N, 100, 0.5, lib=lib)
        s.set_params({"max_iterations": 3, "tolerance": 1e-5, "max_ms": math.inf})
        s.set_initial_states(single_track_x0(N)); s.set_controls(None); s.initialize_problem()
        s.solve(); s.initialize_problem(); s.set_profiling(True); s.solve()
        kt = s.kernel_times()
        print(N, "4 waves per workgroup" if wpb4 else "1 wave per workgroup ", round(1e3 * kt["backward"]["total_ms"] / kt["backward"]["launches"], 1), "us")
        s.close()
