/*
 * mas_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the iLQR hot path of markomiz/multi_agent_solver
 * (reference snapshot 2026-01-09).  Every function cites the reference
 * file:line it follows (paths relative to the reference root).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library -- and only as the checker / CPU baseline.  The product
 * (multi_agent_solver_amd/) never links, imports or calls it.
 *
 * PARITY STATUS
 *   The reference is header-only C++ on Eigen 3.4, and Eigen is absent from
 *   this image, so the reference cannot be compiled here (no oracle/_ref).
 *   - rollout / cost / FD / problem-assembly pieces (a1..a5, a10, a12's
 *     build_global_ocp) are PINNED by the reference's own gtest known answers
 *     (tests/ocp_tests.cpp), restated in oracle/ref_unit_cases.c and stored
 *     as data in tests/golden/reference_unit_tests.json.
 *   - iLQR::solve itself (a6..a9) is exercised by no reference test: for it
 *     this oracle is "PARITY UNPINNED" except for the two costs the
 *     reference publishes in its README.md:160-177 (see tests/golden/
 *     reference_readme_costs.json), and closed-form LQR known answers.
 *
 * All matrices are column-major (Eigen default).  A trajectory is
 * nx x (T+1) with column t contiguous: X[t*nx + i].
 */
#ifndef MAS_ORACLE_H
#define MAS_ORACLE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* arithmetic variants of the restatement (population-tolerance tests only; parity always uses (0, 0)): see mas_oracle.c */
void mas_oracle_set_arith_variant(int sum_order /* 0 index order, 1 packet-pairwise */, int symmetrise /* 0 aliased, 1 mean of old values */);

/* ------------------------------------------------------------------ OCP -- */
/* Mirrors struct OCP, include/multi_agent_solver/ocp.hpp:30-100.            */
struct mas_ocp;
typedef void   (*mas_dynamics_fn)(const struct mas_ocp*, const double* x, const double* u, double* dxdt);
typedef double (*mas_stage_cost_fn)(const struct mas_ocp*, const double* x, const double* u, size_t t);
typedef double (*mas_terminal_cost_fn)(const struct mas_ocp*, const double* x);
/* derivative hooks: out is column-major; NULL => finite differences (ocp.hpp:117-135) */
typedef void (*mas_xu_mat_fn)(const struct mas_ocp*, const double* x, const double* u, double* out);
typedef void (*mas_xut_mat_fn)(const struct mas_ocp*, const double* x, const double* u, size_t t, double* out);
typedef void (*mas_x_mat_fn)(const struct mas_ocp*, const double* x, double* out);

typedef struct mas_ocp {
  int    state_dim, control_dim, horizon_steps;
  double dt;
  double* initial_state;     /* nx            */
  double* initial_states;    /* nx*(T+1)      */
  double* initial_controls;  /* nu*T          */
  int     initial_controls_set; /* reference: initial_controls pre-sized by caller (ocp.hpp:106-109) */
  double* best_states;
  double* best_controls;
  double  best_cost;
  int     has_input_lower, has_input_upper;
  double* input_lower_bounds;
  double* input_upper_bounds;

  mas_dynamics_fn      dynamics;
  mas_stage_cost_fn    stage_cost;
  mas_terminal_cost_fn terminal_cost;

  mas_xu_mat_fn  dynamics_state_jacobian;    /* nx*nx */
  mas_xu_mat_fn  dynamics_control_jacobian;  /* nx*nu */
  mas_xut_mat_fn cost_state_gradient;        /* nx    */
  mas_xut_mat_fn cost_control_gradient;      /* nu    */
  mas_xut_mat_fn cost_state_hessian;         /* nx*nx */
  mas_xut_mat_fn cost_control_hessian;       /* nu*nu */
  mas_xut_mat_fn cost_cross_term;            /* nu*nx */
  mas_x_mat_fn   terminal_cost_gradient;     /* nx    */
  mas_x_mat_fn   terminal_cost_hessian;      /* nx*nx */

  /* function constraints (ocp.hpp:61-68): h(x,u) = 0, g(x,u) <= 0; NULL = none.  *_dim is what the
   * reference obtains by evaluating the function once (ilqr.hpp:318-323).  Jacobian hooks: NULL => FD
   * (ocp.hpp:137-171, finite_differences.hpp:289-345); column-major m x nx / m x nu. */
  void (*equality_constraints)(const struct mas_ocp*, const double* x, const double* u, double* out);
  void (*inequality_constraints)(const struct mas_ocp*, const double* x, const double* u, double* out);
  int  eq_dim, ineq_dim;
  mas_xu_mat_fn equality_constraints_state_jacobian, equality_constraints_control_jacobian;
  mas_xu_mat_fn inequality_constraints_state_jacobian, inequality_constraints_control_jacobian;

  size_t id;
  int    model_id;       /* registry id, 0 for ad-hoc callback problems */
  double params[16];     /* model constants (meaning per model)          */
  void*  ctx;            /* free for ad-hoc problems (e.g. stacked global OCP) */
} mas_ocp;

mas_ocp* mas_ocp_new(int nx, int nu, int T, double dt);
void     mas_ocp_free(mas_ocp*);
void     mas_ocp_set_input_bounds(mas_ocp*, const double* lb, const double* ub);
void     mas_ocp_set_initial_controls(mas_ocp*, const double* U);
void     mas_ocp_initialize_problem(mas_ocp*);        /* ocp.hpp:102-183 */
void     mas_ocp_reset(mas_ocp*);                     /* ocp.hpp:83-93   */
void     mas_ocp_update_initial_with_best(mas_ocp*);  /* ocp.hpp:95-100  */
double   mas_ocp_objective(const mas_ocp*, const double* X, const double* U); /* ocp.hpp:14-28,173-181 */

/* ----------------------------------------------------------- numerics -- */
void mas_integrate_rk4(const mas_ocp*, const double* x, const double* u, double dt, double* out);  /* integrator.hpp:19-28 */
void mas_integrate_horizon(const mas_ocp*, const double* x0, const double* U, double dt, double* X); /* integrator.hpp:31-48 */

void mas_fd_dynamics_state_jacobian(const mas_ocp*, const double* x, const double* u, double* A);     /* finite_differences.hpp:53-71  */
void mas_fd_dynamics_control_jacobian(const mas_ocp*, const double* x, const double* u, double* B);   /* :73-92   */
void mas_fd_cost_state_gradient(const mas_ocp*, const double* x, const double* u, size_t t, double* g);   /* :110-122 */
void mas_fd_cost_control_gradient(const mas_ocp*, const double* x, const double* u, size_t t, double* g); /* :124-136 */
void mas_fd_cost_state_hessian(const mas_ocp*, const double* x, const double* u, size_t t, double* H);    /* :138-173 */
void mas_fd_cost_control_hessian(const mas_ocp*, const double* x, const double* u, size_t t, double* H);  /* :175-210 */
void mas_fd_terminal_cost_gradient(const mas_ocp*, const double* x, double* g);                           /* :212-224 */
void mas_fd_terminal_cost_hessian(const mas_ocp*, const double* x, double* H);                            /* :226-261 */
void mas_fd_cost_cross_term(const mas_ocp*, const double* x, const double* u, size_t t, double* H);       /* :263-287 */
/* whole-trajectory FD gradient used by CGD only (finite_differences.hpp:17-48): kept
 * because the reference's unit test 4 pins it, and it exercises rollout + objective. */
void mas_finite_differences_gradient(const mas_ocp*, const double* x0, const double* U, double dt, double* G);
void mas_clamp_controls_column(const mas_ocp*, double* u);  /* constraint_helpers.hpp:107-114, one column */

/* -------------------------------------------------------------- iLQR -- */
enum {
  MAS_EXIT_CONVERGED = 0,   /* improvement < tolerance    (ilqr.hpp:269-271) */
  MAS_EXIT_MAX_ITER  = 1,   /* iter == max_iterations     (ilqr.hpp:82)      */
  MAS_EXIT_TIME      = 2,   /* elapsed_ms > max_ms        (ilqr.hpp:84-90)   */
  MAS_EXIT_REG_LIMIT = 3    /* oracle-only guard, never reached (see llt_ladder) */
};

typedef struct mas_ilqr {
  /* parameters: defaults ilqr.hpp:26-37 */
  int    max_iterations;
  double tolerance;
  double max_ms;
  int    debug;
  /* augmented-Lagrangian state, ilqr.hpp:31-36,415-421: persists across solve() calls like the members of
   * an iLQR object (multipliers are re-sized, not reset, by resize_buffers :331-338,354-362) */
  double penalty_parameter;                 /* 10.0 */
  double penalty_increase;                  /* 5.0  */
  double constraint_tolerance;              /* 1e-4 */
  double inequality_activation_tolerance;   /* 1e-6 */
  int    mult_T, mult_eq_dim, mult_ineq_dim;
  double* eq_multipliers;                   /* [T][eq_dim]   */
  double* ineq_multipliers;                 /* [T][ineq_dim] */
  double last_eq_violation, last_ineq_violation;
  /* statistics of the most recent solve (extension; reference exposes none) */
  int    iterations;        /* loop bodies executed                     */
  int    exit_reason;
  long   reg_events;        /* times reg*I was added (ilqr.hpp:180)      */
  long   trials;            /* forward rollouts tried in line searches   */
  int    alpha_hist[12];    /* [j]: accepted alpha = 2^-j ; [10]: none   */
  /* optional per-iteration trace (caller-owned, may be NULL) */
  int     trace_cap;
  double* trace_cost;       /* cost after iteration i                    */
  double* trace_alpha;      /* accepted alpha (0 if none)                */
  /* workspace */
  int    ws_nx, ws_nu, ws_T;
  double* ws;
} mas_ilqr;

void mas_ilqr_init(mas_ilqr*);                 /* defaults: 50 / 1e-6 / inf */
void mas_ilqr_free(mas_ilqr*);
/* set_params (ilqr.hpp:39-55): the three keys are required by the reference
 * (params.at throws); here they are plain arguments. */
void mas_ilqr_set_params(mas_ilqr*, int max_iterations, double tolerance, double max_ms, int debug);
void mas_ilqr_set_al_params(mas_ilqr*, double penalty, double penalty_increase, double constraint_tolerance,
                            double inequality_activation_tolerance);   /* optional keys of ilqr.hpp:47-54 */
void mas_ilqr_solve(mas_ilqr*, mas_ocp*);      /* ilqr.hpp:58-273 incl. the augmented-Lagrangian branch */
double mas_ilqr_compute_merit(const mas_ilqr*, const mas_ocp*, const double* X, const double* U);   /* ilqr.hpp:379-407 */
void mas_fd_constraints_state_jacobian(const mas_ocp*, int which /*0 eq, 1 ineq*/, const double* x, const double* u, double* J);   /* finite_differences.hpp:289-316 */
void mas_fd_constraints_control_jacobian(const mas_ocp*, int which, const double* x, const double* u, double* J);                  /* :318-345 */

/* one backward pass on a given trajectory; exposed so tests can compare the
 * linearisation + Riccati outputs of the HIP kernels stage by stage.
 * Outputs (any may be NULL): k [T*nu], K [T*nu*nx col-major nu x nx],
 * derivs [T*(nx*nx + nx*nu + nx + nu + nx*nx + nu*nu + nu*nx)] in the order
 * A,B,l_x,l_u,l_xx,l_uu,l_ux per step. Returns number of reg events. */
long mas_ilqr_backward_pass(mas_ocp*, const double* X, const double* U, double* k, double* K, double* derivs);

/* --------------------------------------------------------- strategies -- */
typedef struct mas_strategy_stats {
  long   total_iterations;     /* sum over agents and outer rounds of iLQR iterations */
  long   converged_solves;     /* solves that exited with MAS_EXIT_CONVERGED          */
  long   reg_events;
  long   trials;
  int    alpha_hist[12];
} mas_strategy_stats;

/* nash.hpp:53-90 */
double mas_run_sequential(mas_ocp** ocps, int n, int max_outer, int max_iterations, double tolerance, double max_ms,
                          mas_strategy_stats* stats, int* iters_per_agent);
/* nash.hpp:182-248; radii_out (n) optional */
double mas_run_trust_region(mas_ocp** ocps, int n, int max_outer, int max_iterations, double tolerance, double max_ms,
                            mas_strategy_stats* stats, double* radii_out, int* iters_per_agent);
/* nash.hpp:92-180 */
double mas_run_line_search(mas_ocp** ocps, int n, int max_outer, int max_iterations, double tolerance, double max_ms,
                           mas_strategy_stats* stats, int* iters_per_agent);
/* centralized.hpp:18-38 + multi_agent_problem.hpp:52-127: dense stacked problem.
 * O(N^3) work with FD -- small N only. */
double mas_run_centralized_dense(mas_ocp** ocps, int n, int max_iterations, double tolerance, double max_ms,
                                 mas_strategy_stats* stats);
/* The same dense solve, the same numbers, with exact re-use (prefix sums of the summed cost's chain, the block structure of A and B)
 * and OpenMP over independent output entries, so that it reaches config 2's 256 agents: mas_oracle_dense_impl.h. */
double mas_run_centralized_dense_structured(mas_ocp** ocps, int n, int max_iterations, double tolerance, double max_ms,
                                            mas_strategy_stats* stats);
/* Same algorithm under the exact-block-diagonal reading of the stacked problem
 * (SURVEY.md section 7): per-agent blocks, but alpha / acceptance / termination
 * (ilqr.hpp:206-228,269) and the regularisation ladder (ilqr.hpp:172-182) are
 * shared by all agents.  This is what the HIP "centralized" path implements. */
double mas_run_centralized_blocked(mas_ocp** ocps, int n, int max_iterations, double tolerance, double max_ms,
                                   mas_strategy_stats* stats);

mas_ocp* mas_build_global_ocp(mas_ocp** ocps, int n);  /* multi_agent_problem.hpp:52-127 */
void     mas_global_ocp_free(mas_ocp*);

/* ------------------------------------------------------ model registry -- */
enum {
  MAS_MODEL_SINGLE_TRACK      = 1, /* examples/multi_agent_single_track.cpp:32-72 + models/single_track_model.hpp:23-44 */
  MAS_MODEL_LQR_IDENTITY      = 2, /* examples/multi_agent_lqr.cpp:21-76   */
  MAS_MODEL_PENDULUM          = 3, /* examples/pendulum_swing_up.cpp:29-117 + models/pendulum_model.hpp:8-20 */
  MAS_MODEL_SINGLE_TRACK_LANE = 4, /* examples/single_track_ocp.cpp:14-116 (analytic derivs, FD cross term)  */
  MAS_MODEL_BICYCLE6          = 5, /* this repo's own 6-state model (no reference counterpart)                */
  MAS_MODEL_SINGLE_TRACK_LIMITED = 6 /* this repo's own: single track + function constraints (speed limit, v >= 0,
                                        steering/acceleration coupling) to exercise the augmented-Lagrangian branch */
};

/* external models (ids >= 100): dynamics / stage cost / terminal cost as plain C functions over the problem's 16 constants; every derivative is
 * finite-differenced.  oracle.register_model() compiles a user-defined workload of the product for the CPU and enters it here. */
typedef void   (*mas_ext_dynamics_fn)(const double* x, const double* u, double* dxdt, const double* params16);
typedef double (*mas_ext_stage_fn)(const double* x, const double* u, int t, const double* params16);
typedef double (*mas_ext_terminal_fn)(const double* x, const double* params16);
int mas_oracle_register_external_model(int nx, int nu, mas_ext_dynamics_fn, mas_ext_stage_fn, mas_ext_terminal_fn,
                                       const double* lower_bounds /* NULL: none */, const double* upper_bounds);
/* optional analytic derivative hooks of an external model (types.hpp:30-50; column-major matrices as everywhere here): a NULL entry
 * stays finite-differenced.  term_grad and term_hess come as a pair, like the product's kAnalyticTerm. */
typedef void (*mas_ext_xu_fn)(const double* x, const double* u, double* out, const double* params16);
typedef void (*mas_ext_xut_fn)(const double* x, const double* u, int t, double* out, const double* params16);
typedef void (*mas_ext_x_fn)(const double* x, double* out, const double* params16);
typedef struct { mas_ext_xu_fn A, B; mas_ext_xut_fn lx, lu, lxx, luu, lux; mas_ext_x_fn term_grad, term_hess; } mas_ext_hooks;
int mas_oracle_external_set_hooks(int model_id, const mas_ext_hooks* hooks);
/* optional function constraints h(x,u) = 0, g(x,u) <= 0 (ocp.hpp:61-68) */
int mas_oracle_external_set_constraints(int model_id, int neq, mas_ext_xu_fn equality, int nin, mas_ext_xu_fn inequality);

/* Build an initialised OCP of a registered model.  x0: nx values.  U0: NULL => zeros.
 * params: NULL => the example's constants.  Bounds come from the model definition. */
mas_ocp* mas_model_make_ocp(int model_id, int T, double dt, const double* x0, const double* U0, const double* params, size_t id);
int      mas_model_dims(int model_id, int* nx, int* nu, int* n_params);
void     mas_model_default_params(int model_id, int T, double* params);

/* Batched convenience driver used by tests / bench (python ctypes):
 * strategy: 0 sequential, 1 trustregion, 2 linesearch, 3 centralized dense, 4 centralized blocked,
 *           5 = single solve per agent (no strategy layer; max_outer ignored), 6 centralized dense with exact re-use (structured)
 * x0 [N*nx], U [N*nu*T] in/out (reference layout per agent), X [N*nx*(T+1)] out, cost [N] out,
 * iters [N] out (total iLQR iterations per agent; centralized: same value for all), stats optional.
 * Returns total cost (Solution.total_cost).  threads <= 0 => omp default. */
double mas_oracle_solve_batch(int model_id, const double* params, int N, int T, double dt,
                              const double* x0, double* U, double* X, double* cost, int* iters,
                              int strategy, int max_outer, int max_iterations, double tolerance, double max_ms,
                              int threads, mas_strategy_stats* stats);
int mas_oracle_max_threads(void);
/* AL parameters given to every solver the strategy drivers create (defaults 10, 5, 1e-4, 1e-6) */
/* buf [N] or NULL: receives enum mas_exit_reason of every agent's last solve (Nash strategies) */
void mas_oracle_set_exit_buffer(int* buf);
void mas_oracle_set_radii_buffer(double* buf);   /* optional [N]: final trust radii of the trust-region strategy (mas_oracle_solve_batch) */
void mas_oracle_set_al_params(double penalty, double penalty_increase, double constraint_tolerance, double inequality_activation_tolerance);
/* 0 = platform libm (faithful to a glibc-linked reference binary), 1 = include/mas_dettrig.h */
void mas_oracle_set_trig_mode(int mode);
int  mas_oracle_get_trig_mode(void);

#ifdef __cplusplus
}
#endif
#endif
