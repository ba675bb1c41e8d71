/*
 * mas_hip.h -- C ABI of the MI355X batched iLQR path (libmas_hip.so).
 *
 * Drop-in boundary for `--solver ilqr` of markomiz/multi_agent_solver.  The reference has
 * no FFI: its "plugin" API is a closed std::variant of solver classes
 * (include/multi_agent_solver/solvers/solver.hpp:17-45) whose alternatives provide
 * set_params(SolverParams) and solve(OCP&), called once per agent inside the OpenMP
 * loops of include/multi_agent_solver/strategies/nash.hpp:59-64,200-212 or once on the
 * stacked problem (strategies/centralized.hpp:22-23).  The GPU batching boundary is one
 * level above solve(OCP&): the whole per-agent loop of a strategy becomes one call here.
 * INTEGRATION.md shows the reference-side binding (a `HipILQR` variant alternative and
 * batched strategy bodies) that a maintainer would add on top of these entry points.
 *
 * Conventions
 *  - plain C, opaque handle, integer status codes (0 = MAS_HIP_OK), no exceptions;
 *  - every array at this boundary is FP64 in the REFERENCE layout: per agent a
 *    column-major Eigen matrix, i.e. states [N][T+1][nx] and controls [N][T][nu] with the
 *    column of one time step contiguous (types.hpp:14-19); the SoA device layout is internal;
 *  - `*_host` entry points take host pointers (one PCIe copy each way); `*_device` ones take
 *    device pointers in the same layout (used with torch tensors' data_ptr());
 *  - a handle is bound to one HIP device and one stream and is not thread-safe
 *    (one handle per GPU, as one rank per GPU under torch.distributed; several handles may be driven from several host threads);
 *  - while mas_hip_solve / mas_hip_run_* execute, the calling thread POLLS one pinned 64-bit word per round of kernels (the device publishes
 *    each round's counters as a single store; an event per round cost 6 us of device idle time per round): one host core is busy for the
 *    duration of the call, which returns after the handle's stream has drained.
 *
 * Stream ordering of the `*_device` entry points
 *    A handle made by mas_hip_create works on its OWN non-blocking stream; one made by mas_hip_create_on_stream works on
 *    the caller's stream.  With its own stream nothing orders the handle's copies against work the caller has queued on
 *    other streams, so the contract is: (1) the contents of an input buffer (set_*_device) and any pending writes to an
 *    output buffer (get_solution_device; e.g. an allocator's zero-fill) must be COMPLETE before the call -- synchronise the
 *    producing stream, or share one stream through mas_hip_create_on_stream; (2) every `*_device` call synchronises the
 *    handle's stream before it returns: inputs are no longer referenced, outputs are complete.
 *
 * Device exclusivity of the centralised strategy
 *    mas_hip_run_centralized agrees on one regularisation rung per time step across all its workgroups with a spin barrier
 *    in global memory, which needs all of them (N/16 four-wavefront workgroups, N <= 1024; up to 32 of them are placed on one
 *    XCD) resident on the device at the same time.  The library checks that they fit this device (occupancy query) and otherwise uses a slower single-workgroup
 *    kernel, but it cannot see OTHER work: do not run kernels from other streams or processes on the device while a
 *    centralised solve is in flight.  If the barrier starves, the call fails with MAS_HIP_ERR_RUNTIME ("gave up waiting for
 *    the rung agreement") instead of hanging.
 *
 * Three documented departures from a glibc-linked reference binary
 *  - sin/cos/tan come from include/mas_dettrig.h (IEEE + - * / and explicit fused multiply-adds, fma(); <= 1 ulp from glibc for sin/cos, <= 3 ulp for tan) so
 *    that GPU and CPU oracle agree bit for bit.  Its argument range is |x| <= 1.6e6 (three-term Cody-Waite reduction);
 *    beyond it the functions return NaN where glibc returns a value.  A line-search trial that reaches such a heading or
 *    steering angle gets a NaN merit and is rejected (NaN < best is false), and a non-finite accepted cost is flagged with
 *    MAS_HIP_EXIT_NONFINITE.  None of the shipped workloads comes within five orders of magnitude of the limit.
 *  - divisions by a kernel-invariant divisor (the finite-difference steps 2 eps, eps^2, 4 eps^2; the wheelbase) are formed on the device as
 *    multiply + exact residual + fused correction + v_div_fixup (div_by, csrc/mas_platform.hpp): the IEEE quotient bit for bit wherever
 *    the quotient is normal and the residual exactly representable: |p| >= 2^-969 and |p / b| >= 2^-1022, or p == 0
 *    (tests/hip/div_by_check.hip).  A finite-difference numerator below ~2e-292 -- differences of costs or dynamics of that
 *    magnitude -- may round differently from `/` in its last bit; no shipped workload comes within 270 orders of magnitude of it.
 *  - max_ms (ilqr.hpp:84-90) is honoured per solve with a clock started on the device when the agent's solve begins; the
 *    check for iteration i+1 is made at the end of iteration i (a few microseconds before the reference's check at the top
 *    of the loop).  All agents of a batch run concurrently, so a budget buys each agent about as many iterations as the
 *    batch completes rounds in that time -- fewer than N serial CPU solves would get when N is large.
 */
#ifndef MAS_HIP_H
#define MAS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAS_HIP_ABI_VERSION 1

enum mas_hip_status {
  MAS_HIP_OK = 0,
  MAS_HIP_ERR_INVALID = 1,     /* bad argument / unknown model / missing parameter */
  MAS_HIP_ERR_NO_DEVICE = 2,   /* no HIP device: the product never falls back to the CPU */
  MAS_HIP_ERR_RUNTIME = 3,     /* a HIP call failed; see mas_hip_last_error()            */
  MAS_HIP_ERR_STATE = 4        /* call order violated (e.g. solve before set_initial)    */
};

/* Registered workload models (device functors replace the reference's std::function
 * callbacks, types.hpp:21-50, which cannot run on a GPU).  Values match the oracle's. */
enum mas_hip_model {
  MAS_HIP_MODEL_SINGLE_TRACK      = 1, /* examples/multi_agent_single_track.cpp:32-72, models/single_track_model.hpp:23-44; FD derivatives */
  MAS_HIP_MODEL_LQR_IDENTITY      = 2, /* examples/multi_agent_lqr.cpp:21-76; analytic derivatives */
  MAS_HIP_MODEL_PENDULUM          = 3, /* examples/pendulum_swing_up.cpp:29-117, models/pendulum_model.hpp:8-20; FD */
  MAS_HIP_MODEL_SINGLE_TRACK_LANE = 4, /* examples/single_track_ocp.cpp:14-116; analytic except cross term */
  MAS_HIP_MODEL_BICYCLE6          = 5, /* this repository's own 6-state model (BASELINE config 5) */
  MAS_HIP_MODEL_SINGLE_TRACK_LIMITED = 6 /* this repository's own: single track + function constraints (speed limit, v >= 0,
                                            steering/acceleration coupling); exercises the augmented-Lagrangian branch */
};

/* per-agent exit status of the most recent solve (extension: the reference's solve()
 * returns nothing, ilqr.hpp:58-273) */
enum mas_hip_exit {
  MAS_HIP_EXIT_CONVERGED = 0,  /* improvement < tolerance, ilqr.hpp:269-271 */
  MAS_HIP_EXIT_MAX_ITER  = 1,  /* iter == max_iterations,  ilqr.hpp:82      */
  MAS_HIP_EXIT_TIME      = 2,  /* elapsed_ms > max_ms,     ilqr.hpp:84-90   */
  MAS_HIP_EXIT_NONFINITE = 4   /* flag OR-ed in: a non-finite cost was met   */
};

typedef struct mas_hip_solver mas_hip_solver;

/* Problem description: what an OCP carries besides its callbacks (ocp.hpp:30-60). */
typedef struct mas_hip_problem_desc {
  int32_t model_id;                   /* enum mas_hip_model                                   */
  int32_t n_agents;                   /* N >= 1                                               */
  int32_t horizon_steps;              /* T >= 1                                               */
  int32_t device;                     /* HIP device ordinal                                   */
  double  dt;
  const double* params;               /* model constants (16 doubles) or NULL = example's     */
  const double* input_lower_bounds;   /* nu values or NULL; clamping needs both (ilqr.hpp:213) */
  const double* input_upper_bounds;   /* NULL,NULL = the model definition's own bounds; pass   */
  int32_t no_input_bounds;            /* 1 to force "no bounds"                               */
  int32_t reserved;
} mas_hip_problem_desc;

/* ---- lifetime ---- */
int mas_hip_abi_version(void);
int mas_hip_device_count(void);
int mas_hip_model_dims(int model_id, int* nx, int* nu);

/* User-defined workloads.  The reference takes a workload as std::function callbacks (types.hpp:21-50: dynamics, stage_cost,
 * terminal_cost; every derivative hook an example leaves empty is finite-differenced, ocp.hpp:117-135); callbacks cannot run on the
 * device, so a workload is a C++ struct of static functions that the kernel templates are instantiated with.  The built-in ones
 * (enum mas_hip_model) are compiled into libmas_hip.so; a new one needs NO rebuild of the library:
 *   struct_source   the definition of one struct deriving from mas::FdModel<Self, NX, NU> (csrc/mas_models.hpp), e.g.
 *                     struct Unicycle : FdModel<Unicycle, 3, 2> {
 *                       MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) { ... }
 *                       MAS_HD static double stage_cost(const double* x, const double* u, int t, const ModelParams& P) { ... }
 *                       MAS_HD static double terminal_cost(const double* x, const ModelParams& P) { ... } };
 *                   1 <= NX, NU <= 8.  The reference's OPTIONAL callbacks are members of the same struct: a derivative hook (types.hpp:30-50)
 *                   is `static constexpr bool kAnalyticA = true` + `A(x, u, out, P)` (likewise B, lx(x, u, t, out, P), lu, lxx, luu, lux,
 *                   term_grad(x, out, P) + term_hess; column-major), function constraints (ocp.hpp:61-68) are NEQ / NIN + equality(x, u, out, P)
 *                   / inequality(...); input bounds kDefaultBounds + default_bounds(lb, ub).  What is not declared is finite-differenced / absent.
 *   struct_name     its name
 *   source_root     tree that holds multi_agent_solver_amd/csrc and include (NULL: the tree this library was built in)
 *   model_id        out: the id to put into mas_hip_problem_desc::model_id (>= 100)
 * The struct is compiled with the ROCm compiler of the machine (hipcc; ~1.5 minutes: the host driver and every kernel template for
 * this model, with the library's own options -- no contraction, IEEE division) into a plugin library, which is loaded into the
 * process.  Needs hipcc at run time; MAS_HIP_ERR_INVALID with the compiler's messages in mas_hip_last_error(NULL) when the source
 * does not compile.  Compiled plugins are kept (MAS_HIP_MODEL_CACHE, default /tmp/mas_hip_models_<uid>; the key is a hash of the struct, its
 * name and every header it is compiled against): registering the same workload again -- every start of an application -- loads the plugin
 * at once.  Process-wide, thread-safe (registrations from several threads compile side by side), not undone. */
int mas_hip_register_model(const char* struct_source, const char* struct_name, const char* source_root, int* model_id);
int mas_hip_model_default_params(int model_id, int horizon_steps, double* params16);
int mas_hip_create(const mas_hip_problem_desc* desc, mas_hip_solver** out);
int mas_hip_create_on_stream(const mas_hip_problem_desc* desc, void* hip_stream, mas_hip_solver** out);
int mas_hip_destroy(mas_hip_solver*);
const char* mas_hip_last_error(const mas_hip_solver*);   /* NULL handle: last create() error */

/* ---- solver parameters: iLQR::set_params, ilqr.hpp:39-55 ----
 * keys "max_iterations", "tolerance", "max_ms" (required by the reference: params.at()),
 * optional "debug" (the reference's per-solve and per-iteration lines, ilqr.hpp:79-80,262-267, prefixed "[agent i] "), "penalty", "penalty_increase", "constraint_tolerance",
 * "inequality_activation_tolerance" (augmented-Lagrangian branch, ilqr.hpp:47-54);
 * unknown keys are ignored like the reference. */
int mas_hip_set_param(mas_hip_solver*, const char* key, double value);
/* fails with MAS_HIP_ERR_INVALID if a required key was never set (mirrors std::out_of_range) */
int mas_hip_check_params(const mas_hip_solver*);

/* ---- kernel forms (tests / measurements) ----
 * Several launches exist in more than one BIT-IDENTICAL form (e.g. the two-kernel fused round against the five-kernel one, the
 * one-lane against the four-lane line search); the defaults are the measured best.  The parity tests hold the forms against each
 * other through this entry point.  Keys: "fused" (0/1), "forward_quad_max" (live agents), "backward_tile" (0/1),
 * "backward_list" (0/1), "backward_wide_max" / "backward_wave_max" (live agents up to which the six-state backward pass uses 32 / 64 lanes per agent), "central_pipeline" (0/1), "central_one_xcd" (0/1), "force_single_workgroup_central" (0/1),
 * "central_combine" (0/1), "central_peek" (0/1), "central_wpb" (1, 2, 4, 8, 16; clamped to what the CU's LDS holds for the model), "central_fold" (0/1: centralised
 * iterations in four launches -- the sweep commits, the select publishes -- or with a commit kernel and a memset), "fused_trust" (0/1: the chained trust-region
 * epilogue inside the fused kernel), "trust_fused_max" (live agents up to which the five-kernel round's epilogue is one kernel), "trust_reroll_levels" (0/1),
 * "forward_staged" (0/1: models with kStagedLineSearch roll out the first step size alone and the others only where it failed), "dense_chip" (0/1: the literal
 * centralised solve as chip-wide launches or in one workgroup), "dense_ladder_blocked" (0/1).  Unknown key or value: MAS_HIP_ERR_INVALID.  Results never depend on them. */
int mas_hip_set_option(mas_hip_solver*, const char* key, int64_t value);

/* ---- problem data: OCP::initial_state, OCP::best_controls (warm start, ilqr.hpp:72-75) ---- */
int mas_hip_set_initial_states_host(mas_hip_solver*, const double* x0 /* [N][nx] */);
int mas_hip_set_controls_host(mas_hip_solver*, const double* U /* [N][T][nu] or NULL = zeros */);
int mas_hip_set_initial_states_device(mas_hip_solver*, const void* x0_dev);
int mas_hip_set_controls_device(mas_hip_solver*, const void* U_dev);
/* OCP::initialize_problem (ocp.hpp:102-183): roll out the current controls, set best_cost. */
int mas_hip_initialize_problem(mas_hip_solver*);
/* snapshot / restore of (x0, controls) on the device, so a benchmark can repeat a run with
 * its inputs already resident in HBM */
int mas_hip_save_initial(mas_hip_solver*);
int mas_hip_restore_initial(mas_hip_solver*);   /* enqueued on the handle's stream (returns before the copies have run) */

/* ---- the hot path ---- */
/* N x iLQR::solve (ilqr.hpp:58-273) == first loop of detail::sequential_solve, nash.hpp:59-64 */
int mas_hip_solve(mas_hip_solver*);
/* detail::run_sequential, nash.hpp:74-90 */
int mas_hip_run_sequential(mas_hip_solver*, int max_outer);
/* detail::run_trust_region, nash.hpp:182-248 */
int mas_hip_run_trust_region(mas_hip_solver*, int max_outer);
/* detail::run_line_search, nash.hpp:92-180 */
int mas_hip_run_line_search(mas_hip_solver*, int max_outer);
/* CentralizedStrategy::operator(), centralized.hpp:18-38, under the exact block-diagonal
 * reading of the stacked problem: shared alpha / acceptance / termination / regularisation */
int mas_hip_run_centralized(mas_hip_solver*);
/* CentralizedStrategy::operator() in the reference's LITERAL reading: one iLQR on the stacked OCP of build_global_ocp
 * (multi_agent_problem.hpp:52-127) -- finite differences of the SUMMED cost, dense stacked Hessians and gains, one shared
 * alpha / termination / regularisation by construction.  4 (N nx)^2 evaluations of the N-agent cost per time step and a dense
 * (N nx) x (N nx) Riccati recursion; computed as chip-wide launches with exact re-use only (csrc/mas_dense_chip.hpp: the prefix of the
 * summed cost's own chain of additions, the exact zeros of the stacked A and B) up to 256 agents and 1024 stacked controls
 * (MAS_HIP_ERR_INVALID beyond): single-track, H = 100: 0.1 s at 16 agents, 0.4 s at 64, 2.7 s at BASELINE config 2's 256.  Bit-equal to
 * the CPU oracle's restatement.  mas_hip_set_option("dense_chip", 0) selects the one-workgroup form of rounds 2-3 (<= 64 agents, 56 s
 * there).  Precondition of the re-use: finite value function (a diverged solve may differ in WHICH non-finite values it holds).
 * Per-agent costs are the agents' own objectives of
 * the stacked solution and mas_hip_total_cost the stacked best_cost (centralized.hpp:26-35); iterations are the stacked solve's. */
int mas_hip_run_centralized_dense(mas_hip_solver*);

/* ---- agents sharded over ranks (one handle per GPU): the one exchange the Nash strategies have ----
 * detail::compute_total_cost (nash.hpp:45-49) sums the cost of ALL agents; run_line_search accepts or rejects a
 * joint step on that sum (nash.hpp:119,134-143).  When the id-sorted block list is sharded over ranks, the caller
 * installs this callback: it receives this handle's per-agent costs (host pointer, n_local doubles, block order)
 * and must return in *global_total the sum over every rank's agents accumulated serially in global block order
 * (rank 0's agents first) -- the value a single handle holding all agents computes, so that every rank takes the
 * same decisions.  Non-zero return aborts the strategy call with MAS_HIP_ERR_RUNTIME.  NULL (default): local sum.
 * The sequential and trust-region strategies exchange nothing (nash.hpp:59-64,200-212: agents are decoupled). */
typedef int (*mas_hip_cost_exchange_fn)(void* ctx, const double* local_costs, int32_t n_local, double* global_total);
int mas_hip_set_cost_exchange(mas_hip_solver*, mas_hip_cost_exchange_fn fn, void* ctx);

/* ---- agents sharded over the GPUs of a node: RCCL inside the library ----
 * Replaces the OpenMP fan-out over agents of strategies/nash.hpp:59-71 across devices: the id-sorted block list
 * (multi_agent_problem.hpp:37-50) is cut into contiguous ranges (mas_hip_shard_range), one handle per GPU -- one process per GPU
 * (mas_hip_comm_create, ids handed around by the launcher) or one process driving several GPUs from one thread each
 * (mas_hip_comm_create_all).  librccl.so.1 is opened on first use; nothing else in this header needs it.  The sequential and
 * trust-region strategies exchange nothing (agents are decoupled in every shipped example); the two exchanges that exist:
 *   mas_hip_all_gather_solution     the publish step (update_initial_with_best for coupled costs, nash.hpp:66-71): every rank
 *                                   receives every agent's trajectory, [n_total][T+1][nx] / [n_total][T][nu] / [n_total] DEVICE buffers in
 *                                   the reference layout; the shard is converted on the device straight to its displacement and travels
 *                                   as one collective per array (ncclAllGather; uneven shards: grouped broadcasts, nothing padded)
 *   mas_hip_set_comm_cost_exchange  run_line_search's total cost over ALL agents (nash.hpp:45-49,119,134-143): the shards' costs are
 *                                   all-gathered and summed by one wavefront in global block order on every rank -- bit for bit
 *                                   the total one handle holding all agents forms, so every rank takes the same decisions, and no
 *                                   cost leaves the device.  Takes precedence over nothing: clear the host callback first.
 * A communicator is bound to the device it was created on and must outlive the handles that use it. */
typedef struct mas_hip_comm mas_hip_comm;
#define MAS_HIP_COMM_ID_BYTES 128
int mas_hip_comm_unique_id(void* id_out /* MAS_HIP_COMM_ID_BYTES */);                               /* ncclGetUniqueId: one rank, then to the others by any channel */
int mas_hip_comm_create(const void* id, int n_ranks, int rank, int device, mas_hip_comm** out);     /* ncclCommInitRank (collective over the ranks) */
int mas_hip_comm_create_all(int n_devices, const int* devices, mas_hip_comm** out /* [n_devices] */);   /* ncclCommInitAll: one process, several GPUs */
int mas_hip_comm_destroy(mas_hip_comm*);
/* a rank that cannot go on (upload failed, handle refused, solve gave up) releases every peer that waits for it inside a collective
 * (ncclCommAbort).  May be called from another thread than the one driving the rank; every later collective on the communicator
 * fails at once; the object stays valid until mas_hip_comm_destroy.  mas_amd's sharded strategies abort ALL communicators of the
 * group from the failing shard's thread. */
int mas_hip_comm_abort(mas_hip_comm*);
const char* mas_hip_comm_last_error(void);
/* Communicators made AFTER this call use the in-process LOOPBACK transport instead of librccl (0: back to librccl): ranks are host
 * threads of this process, collectives a rendezvous and copies between the ranks' buffers.  Not a product path -- it exists so that
 * the multi-rank logic of this section (shard displacements, even and uneven; grouped broadcasts; the line search's total over all
 * ranks) runs with several ranks where RCCL cannot: without a second GPU, with several shards on one GPU (tests/test_loopback_ranks.py,
 * tests/test_gpu_rccl.py).  The multi-rank entry points have run over librccl itself with ONE rank only so far. */
int mas_hip_comm_use_loopback(int enabled);
void mas_hip_shard_range(int n_total, int n_ranks, int rank, int* lo, int* hi);                       /* contiguous id ranges, sizes differ by at most one */
int mas_hip_all_gather_solution(mas_hip_solver*, mas_hip_comm*, int n_total, void* X_all_dev, void* U_all_dev, void* cost_all_dev);   /* any buffer may be NULL */
int mas_hip_set_comm_cost_exchange(mas_hip_solver*, mas_hip_comm* /* NULL: local sum */, int n_total);

/* ---- results: OCP::best_states/best_controls/best_cost; Solution (solution.hpp:9-15) ---- */
int mas_hip_get_solution_host(mas_hip_solver*, double* X /* [N][T+1][nx] */, double* U /* [N][T][nu] */,
                              double* cost /* [N] */, int32_t* iterations /* [N] */, int32_t* exit_status /* [N] */);
int mas_hip_get_solution_device(mas_hip_solver*, void* X_dev, void* U_dev, void* cost_dev);
int mas_hip_total_cost(mas_hip_solver*, double* total);             /* Solution.total_cost */
int mas_hip_get_trust_radii_host(mas_hip_solver*, double* radii /* [N] */);

/* ---- staged access for parity tests: one linearisation sweep + one backward pass on the
 * current trajectory.  derivs [N][T][ND] with ND = A,B,l_x,l_u,l_xx,l_uu,l_ux (column-major
 * blocks), k [N][T][nu], K [N][T][nx][nu] (column-major nu x nx per step). Any may be NULL. */
int mas_hip_debug_backward_host(mas_hip_solver*, double* derivs, double* k, double* K, int64_t* reg_events);

/* ---- measurement ---- */
typedef struct mas_hip_stats {
  int64_t agent_iterations;      /* sum over agents of iLQR iterations since reset          */
  int64_t converged_solves;      /* solves that exited through improvement < tolerance      */
  int64_t solves;                /* agent-solves started                                    */
  int64_t reg_events;            /* times reg*I was added (ilqr.hpp:180)                    */
  int64_t batch_iterations;      /* host-loop iterations (kernel rounds)                    */
  double  wall_ms;               /* host wall time inside run_* / solve since reset         */
} mas_hip_stats;
int mas_hip_get_stats(mas_hip_solver*, mas_hip_stats* out);
int mas_hip_reset_stats(mas_hip_solver*);

/* per-kernel HIP-event timing on the handle's stream (off by default) */
typedef struct mas_hip_kernel_time {
  char    name[48];
  int64_t launches;
  double  total_ms;
  double  algorithmic_bytes;     /* summed over launches (DESIGN.md defines the per-unit figures) */
} mas_hip_kernel_time;
int mas_hip_set_profiling(mas_hip_solver*, int enabled);
int mas_hip_get_kernel_times(mas_hip_solver*, mas_hip_kernel_time* out, int capacity, int* count);

#ifdef __cplusplus
}
#endif
#endif /* MAS_HIP_H */
