/*
 * ref_unit_cases.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * The reference's own four gtest cases (tests/ocp_tests.cpp:21-177) restated on the
 * oracle's C API.  Each function rebuilds the test's inline OCP and writes the
 * quantities the gtest asserts on into `out`; tests/test_oracle_reference_pins.py
 * compares them with the expected values stored as data in
 * tests/golden/reference_unit_tests.json.  This is how the oracle's rollout,
 * cost-summation, finite-difference and stacked-problem pieces are pinned.
 */
#include "mas_oracle.h"

#include <stdlib.h>
#include <string.h>

/* create_integrator(), tests/ocp_tests.cpp:13-17:  f(x,u) = u + 0*x */
static void integrator_dyn(const mas_ocp* o, const double* x, const double* u, double* d) {
  for (int i = 0; i < o->state_dim; ++i) d[i] = u[i] + x[i] * 0.0;
}
static double sqnorm_stage(const mas_ocp* o, const double* x, const double* u, size_t t) {
  (void)t; double a = 0.0, b = 0.0;
  for (int i = 0; i < o->state_dim; ++i) a += x[i] * x[i];
  for (int i = 0; i < o->control_dim; ++i) b += u[i] * u[i];
  return a + b;
}
static double sqnorm_terminal(const mas_ocp* o, const double* x) {
  double a = 0.0; for (int i = 0; i < o->state_dim; ++i) a += x[i] * x[i]; return a;
}
static double zero_stage(const mas_ocp* o, const double* x, const double* u, size_t t) { (void)o; (void)x; (void)u; (void)t; return 0.0; }
static double zero_term(const mas_ocp* o, const double* x) { (void)o; (void)x; return 0.0; }

/* TEST(OCPTest, InitializeProblemSetsDefaultsAndBestCost), :21-54
 * out: [0] best_cost  [1] state_grad(0)  [2] control_grad(0)  [3..6] X(0..3)  */
int mas_ref_case_initialize(double* out) {
  mas_ocp* o = mas_ocp_new(1, 1, 3, 0.1);
  o->dynamics = integrator_dyn; o->stage_cost = sqnorm_stage; o->terminal_cost = sqnorm_terminal;
  mas_ocp_initialize_problem(o);
  out[0] = o->best_cost;
  mas_fd_cost_state_gradient(o, o->best_states, o->best_controls, 0, &out[1]);
  mas_fd_cost_control_gradient(o, o->best_states, o->best_controls, 0, &out[2]);
  for (int t = 0; t <= 3; ++t) out[3 + t] = o->best_states[t];
  mas_ocp_free(o);
  return 7;
}

/* TEST(OCPTest, UpdateInitialWithBestCopiesTrajectories), :56-74
 * out: initial_controls (4) then initial_states (6) after update_initial_with_best */
int mas_ref_case_update_initial(double* out) {
  mas_ocp* o = mas_ocp_new(2, 2, 2, 1.0);
  o->dynamics = integrator_dyn; o->stage_cost = zero_stage; o->terminal_cost = zero_term;
  mas_ocp_initialize_problem(o);
  for (int i = 0; i < 4; ++i) o->best_controls[i] = 1.0;
  for (int i = 0; i < 6; ++i) o->best_states[i] = 1.0;
  mas_ocp_update_initial_with_best(o);
  memcpy(out, o->initial_controls, sizeof(double) * 4);
  memcpy(out + 4, o->initial_states, sizeof(double) * 6);
  mas_ocp_free(o);
  return 10;
}

/* TEST(MultiAgentProblemTest, BuildGlobalProblemMergesAgents), :76-154 */
static void dyn_a(const mas_ocp* o, const double* x, const double* u, double* d) { for (int i = 0; i < o->state_dim; ++i) d[i] = x[i] + u[0]; }
static double stage_a(const mas_ocp* o, const double* x, const double* u, size_t t) { (void)t; double s = 0.0; for (int i = 0; i < o->state_dim; ++i) s += x[i]; double c = 0.0; for (int i = 0; i < o->control_dim; ++i) c += u[i]; return s + c; }
static double term_a(const mas_ocp* o, const double* x) { double s = 0.0; for (int i = 0; i < o->state_dim; ++i) s += x[i]; return 2.0 * s; }
static void dyn_b(const mas_ocp* o, const double* x, const double* u, double* d) { double c = 0.0; for (int i = 0; i < o->control_dim; ++i) c += u[i]; for (int i = 0; i < o->state_dim; ++i) d[i] = x[i] + 2.0 * c; }
static double stage_b(const mas_ocp* o, const double* x, const double* u, size_t t) { (void)t; double s = 0.0; for (int i = 0; i < o->state_dim; ++i) s += x[i]; double c = 0.0; for (int i = 0; i < o->control_dim; ++i) c += u[i]; return 2.0 * s + 3.0 * c; }
static double term_b(const mas_ocp* o, const double* x) { double s = 0.0; for (int i = 0; i < o->state_dim; ++i) s += x[i]; return s; }

static int by_id(const void* p, const void* q) {
  const mas_ocp* a = *(mas_ocp* const*)p; const mas_ocp* b = *(mas_ocp* const*)q;
  return (a->id > b->id) - (a->id < b->id);
}
/* compute_offsets, multi_agent_problem.hpp:37-50: blocks sorted by agent id */
void mas_sort_by_id(mas_ocp** ocps, int n) { qsort(ocps, (size_t)n, sizeof(mas_ocp*), by_id); }

/* out: [0] first block id [1] second block id [2] state_dim [3] control_dim [4] horizon [5] dt
 *      [6..8] lower bounds [9..11] initial state [12..14] dynamics(state,control)
 *      [15] stage_cost [16] terminal_cost [17] has bounds */
int mas_ref_case_global(double* out) {
  mas_ocp* a = mas_ocp_new(2, 1, 2, 0.5);
  a->initial_state[0] = a->initial_state[1] = 1.0;
  a->dynamics = dyn_a; a->stage_cost = stage_a; a->terminal_cost = term_a;
  { const double lb[1] = {-1.0}, ub[1] = {1.0}; mas_ocp_set_input_bounds(a, lb, ub); }
  a->id = 1; mas_ocp_initialize_problem(a);
  mas_ocp* b = mas_ocp_new(1, 2, 2, 0.5);
  b->initial_state[0] = 3.0;
  b->dynamics = dyn_b; b->stage_cost = stage_b; b->terminal_cost = term_b;
  { const double lb[2] = {-2.0, -2.0}, ub[2] = {2.0, 2.0}; mas_ocp_set_input_bounds(b, lb, ub); }
  b->id = 2; mas_ocp_initialize_problem(b);

  mas_ocp* list[2] = {b, a};             /* added as (2, b) then (1, a) */
  mas_sort_by_id(list, 2);
  out[0] = (double)list[0]->id; out[1] = (double)list[1]->id;
  mas_ocp* g = mas_build_global_ocp(list, 2);
  out[2] = g->state_dim; out[3] = g->control_dim; out[4] = g->horizon_steps; out[5] = g->dt;
  for (int i = 0; i < 3; ++i) out[6 + i] = g->input_lower_bounds[i];
  for (int i = 0; i < 3; ++i) out[9 + i] = g->initial_state[i];
  const double state[3] = {1.0, 2.0, 3.0};      /* LinSpaced(3, 1, 3)  */
  const double control[3] = {-1.0, 0.0, 1.0};   /* LinSpaced(3, -1, 1) */
  g->dynamics(g, state, control, &out[12]);
  out[15] = g->stage_cost(g, state, control, 0);
  out[16] = g->terminal_cost(g, state);
  out[17] = (double)(g->has_input_lower && g->has_input_upper);
  mas_global_ocp_free(g); mas_ocp_free(a); mas_ocp_free(b);
  return 18;
}

/* TEST(FiniteDifferencesTest, GradientMatchesAnalyticalForQuadraticObjective), :156-177 */
static void dyn_u(const mas_ocp* o, const double* x, const double* u, double* d) { (void)x; for (int i = 0; i < o->state_dim; ++i) d[i] = u[i]; }
int mas_ref_case_fd_gradient(double* out) {
  mas_ocp* o = mas_ocp_new(1, 1, 2, 1.0);
  o->dynamics = dyn_u; o->stage_cost = sqnorm_stage; o->terminal_cost = sqnorm_terminal;  /* = sum states^2 + sum controls^2 */
  const double U[2] = {1.0, -1.0};
  mas_finite_differences_gradient(o, o->initial_state, U, 1.0, out);
  /* also the rollout itself: RK4 of x' = u is exactly x + dt*u */
  double X[3]; mas_integrate_horizon(o, o->initial_state, U, 1.0, X);
  out[2] = X[0]; out[3] = X[1]; out[4] = X[2];
  mas_ocp_free(o);
  return 5;
}
