/*
 * mas_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See mas_oracle.h for the parity status and the usage rule.
 *
 * Plain C restatement of the reference's iLQR hot path.  Build with
 *   gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp
 * (the reference's CMake sets no -march, i.e. baseline x86-64 without FMA,
 * CMakeLists.txt:1-45; -ffp-contract=off keeps that true on any host).
 *
 * Where Eigen leaves the order of a small reduction unspecified the natural
 * (increasing index) order is used; this is the only liberty taken and it is
 * a rounding-level (1e-16 relative) one.
 */
#include "mas_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Transcendental functions of the workload models.  mode 0 (default): the platform libm, as
 * the reference binary would call it (std::sin/cos/tan).  mode 1: the portable bit-reproducible
 * implementation of include/mas_dettrig.h, which the HIP path always uses; in this mode the
 * oracle and the GPU agree to the last bit.  See mas_dettrig.h for why this matters (the
 * algorithm is chaotic: last-bit differences in sin/cos change converged costs by whole per cents). */
#include "../include/mas_dettrig.h"
static int g_trig_mode = 0;
void mas_oracle_set_trig_mode(int mode) { g_trig_mode = mode; }
int  mas_oracle_get_trig_mode(void) { return g_trig_mode; }
static double o_sin(double x) { return g_trig_mode ? mas_dt_sin(x) : sin(x); }
static double o_cos(double x) { return g_trig_mode ? mas_dt_cos(x) : cos(x); }
static double o_tan(double x) { return g_trig_mode ? mas_dt_tan(x) : tan(x); }

static double* dalloc(size_t n) {
  double* p = (double*)calloc(n ? n : 1, sizeof(double));
  if (!p) { fprintf(stderr, "mas_oracle: out of memory\n"); abort(); }
  return p;
}

/* ===================================================================== OCP */

mas_ocp* mas_ocp_new(int nx, int nu, int T, double dt) {
  mas_ocp* o = (mas_ocp*)calloc(1, sizeof(mas_ocp));
  o->state_dim = nx; o->control_dim = nu; o->horizon_steps = T; o->dt = dt;
  o->initial_state    = dalloc((size_t)nx);
  o->initial_states   = dalloc((size_t)nx * (T + 1));
  o->initial_controls = dalloc((size_t)nu * T);
  o->best_states      = dalloc((size_t)nx * (T + 1));
  o->best_controls    = dalloc((size_t)nu * T);
  o->input_lower_bounds = dalloc((size_t)nu);
  o->input_upper_bounds = dalloc((size_t)nu);
  o->best_cost = 1.7976931348623157e308; /* numeric_limits<double>::max(), ocp.hpp:39 */
  return o;
}

void mas_ocp_free(mas_ocp* o) {
  if (!o) return;
  free(o->initial_state); free(o->initial_states); free(o->initial_controls);
  free(o->best_states); free(o->best_controls);
  free(o->input_lower_bounds); free(o->input_upper_bounds);
  free(o);
}

void mas_ocp_set_input_bounds(mas_ocp* o, const double* lb, const double* ub) {
  if (lb) { memcpy(o->input_lower_bounds, lb, sizeof(double) * o->control_dim); o->has_input_lower = 1; }
  if (ub) { memcpy(o->input_upper_bounds, ub, sizeof(double) * o->control_dim); o->has_input_upper = 1; }
}

void mas_ocp_set_initial_controls(mas_ocp* o, const double* U) {
  memcpy(o->initial_controls, U, sizeof(double) * o->control_dim * o->horizon_steps);
  o->initial_controls_set = 1;
}

/* compute_trajectory_cost, ocp.hpp:14-28: stages 0..T-1 in order from 0.0, then terminal. */
double mas_ocp_objective(const mas_ocp* o, const double* X, const double* U) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  double cost = 0.0;
  for (int t = 0; t < T; ++t) cost += o->stage_cost(o, X + (size_t)t * nx, U + (size_t)t * nu, (size_t)t);
  cost += o->terminal_cost(o, X + (size_t)T * nx);
  return cost;
}

/* integrate_rk4, integrator.hpp:19-28 */
void mas_integrate_rk4(const mas_ocp* o, const double* x, const double* u, double dt, double* out) {
  const int nx = o->state_dim;
  double stackbuf[5 * 16];
  double* buf = nx <= 16 ? stackbuf : dalloc((size_t)5 * nx);
  double *k1 = buf, *k2 = buf + nx, *k3 = buf + 2 * nx, *k4 = buf + 3 * nx, *xs = buf + 4 * nx;
  o->dynamics(o, x, u, k1);
  for (int i = 0; i < nx; ++i) xs[i] = x[i] + (0.5 * dt) * k1[i];
  o->dynamics(o, xs, u, k2);
  for (int i = 0; i < nx; ++i) xs[i] = x[i] + (0.5 * dt) * k2[i];
  o->dynamics(o, xs, u, k3);
  for (int i = 0; i < nx; ++i) xs[i] = x[i] + dt * k3[i];
  o->dynamics(o, xs, u, k4);
  for (int i = 0; i < nx; ++i) out[i] = x[i] + (dt / 6.0) * (((k1[i] + 2 * k2[i]) + 2 * k3[i]) + k4[i]);
  if (buf != stackbuf) free(buf);
}

/* integrate_horizon, integrator.hpp:31-48 */
void mas_integrate_horizon(const mas_ocp* o, const double* x0, const double* U, double dt, double* X) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  memcpy(X, x0, sizeof(double) * nx);
  for (int t = 0; t < T; ++t) mas_integrate_rk4(o, X + (size_t)t * nx, U + (size_t)t * nu, dt, X + (size_t)(t + 1) * nx);
}

/* initialize_problem, ocp.hpp:102-183 (FD hooks are represented by NULL hooks) */
void mas_ocp_initialize_problem(mas_ocp* o) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  if (!o->initial_controls_set) memset(o->initial_controls, 0, sizeof(double) * nu * T);
  mas_integrate_horizon(o, o->initial_state, o->initial_controls, o->dt, o->initial_states);
  memcpy(o->best_states, o->initial_states, sizeof(double) * nx * (T + 1));
  memcpy(o->best_controls, o->initial_controls, sizeof(double) * nu * T);
  o->best_cost = mas_ocp_objective(o, o->initial_states, o->initial_controls);
}

/* reset, ocp.hpp:83-93 */
void mas_ocp_reset(mas_ocp* o) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  memset(o->initial_controls, 0, sizeof(double) * nu * T);
  mas_integrate_horizon(o, o->initial_state, o->initial_controls, o->dt, o->initial_states);
  memcpy(o->best_states, o->initial_states, sizeof(double) * nx * (T + 1));
  memcpy(o->best_controls, o->initial_controls, sizeof(double) * nu * T);
  o->best_cost = mas_ocp_objective(o, o->initial_states, o->initial_controls);
}

/* update_initial_with_best, ocp.hpp:95-100 */
void mas_ocp_update_initial_with_best(mas_ocp* o) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  memcpy(o->initial_controls, o->best_controls, sizeof(double) * nu * T);
  memcpy(o->initial_states, o->best_states, sizeof(double) * nx * (T + 1));
}

/* clamp_controls, constraint_helpers.hpp:107-114: cwiseMin(upper) then cwiseMax(lower).
 * The reference re-clamps the whole matrix at every step (ilqr.hpp:213-214); all other
 * columns are either already clamped or overwritten before use, so clamping the current
 * column is equivalent. */
void mas_clamp_controls_column(const mas_ocp* o, double* u) {
  for (int i = 0; i < o->control_dim; ++i) {
    double v = u[i];
    v = (o->input_upper_bounds[i] < v) ? o->input_upper_bounds[i] : v; /* std::min(v, ub) */
    v = (v < o->input_lower_bounds[i]) ? o->input_lower_bounds[i] : v; /* std::max(v, lb) */
    u[i] = v;
  }
}

/* ====================================================== finite differences */

static double safe_stage(const mas_ocp* o, const double* x, const double* u, size_t t) {
  double v = o->stage_cost(o, x, u, t);          /* finite_differences.hpp:95-100 */
  return isfinite(v) ? v : 0.0;
}
static double safe_terminal(const mas_ocp* o, const double* x) {
  double v = o->terminal_cost(o, x);             /* finite_differences.hpp:102-107 */
  return isfinite(v) ? v : 0.0;
}

/* v + s1*e_i*eps (+ s2*e_j*eps): mirrors Eigen's  x + dx  /  x - dx  on whole vectors,
 * including the "+ 0.0" on untouched components. */
static void perturb1(const double* v, int n, int i, double se, double* out) {
  for (int k = 0; k < n; ++k) {
    double d = (k == i) ? fabs(se) : 0.0;
    out[k] = se >= 0 ? v[k] + d : v[k] - d;
  }
}
static void perturb2(const double* v, int n, int i, double si, int j, double sj, double* out) {
  for (int k = 0; k < n; ++k) {
    double di = (k == i) ? fabs(si) : 0.0, dj = (k == j) ? fabs(sj) : 0.0;
    double w = si >= 0 ? v[k] + di : v[k] - di;
    out[k]   = sj >= 0 ? w + dj : w - dj;
  }
}

void mas_fd_dynamics_state_jacobian(const mas_ocp* o, const double* x, const double* u, double* A) {
  const int nx = o->state_dim; const double eps = 1e-6;
  double* w = dalloc((size_t)3 * nx); double *xp = w, *fp = w + nx, *fm = w + 2 * nx;
  for (int i = 0; i < nx; ++i) {
    perturb1(x, nx, i, +eps, xp); o->dynamics(o, xp, u, fp);
    perturb1(x, nx, i, -eps, xp); o->dynamics(o, xp, u, fm);
    for (int r = 0; r < nx; ++r) A[(size_t)i * nx + r] = (fp[r] - fm[r]) / (2 * eps);
  }
  free(w);
}

void mas_fd_dynamics_control_jacobian(const mas_ocp* o, const double* x, const double* u, double* B) {
  const int nx = o->state_dim, nu = o->control_dim; const double eps = 1e-6;
  double* w = dalloc((size_t)2 * nx + nu); double *fp = w, *fm = w + nx, *up = w + 2 * nx;
  for (int i = 0; i < nu; ++i) {
    perturb1(u, nu, i, +eps, up); o->dynamics(o, x, up, fp);
    perturb1(u, nu, i, -eps, up); o->dynamics(o, x, up, fm);
    for (int r = 0; r < nx; ++r) B[(size_t)i * nx + r] = (fp[r] - fm[r]) / (2 * eps);
  }
  free(w);
}

void mas_fd_cost_state_gradient(const mas_ocp* o, const double* x, const double* u, size_t t, double* g) {
  const int nx = o->state_dim; const double eps = 1e-6;
  double* xp = dalloc((size_t)nx);
  for (int i = 0; i < nx; ++i) {
    perturb1(x, nx, i, +eps, xp); double fp = o->stage_cost(o, xp, u, t);
    perturb1(x, nx, i, -eps, xp); double fm = o->stage_cost(o, xp, u, t);
    g[i] = (fp - fm) / (2 * eps);
  }
  free(xp);
}

void mas_fd_cost_control_gradient(const mas_ocp* o, const double* x, const double* u, size_t t, double* g) {
  const int nu = o->control_dim; const double eps = 1e-6;
  double* up = dalloc((size_t)nu);
  for (int i = 0; i < nu; ++i) {
    perturb1(u, nu, i, +eps, up); double fp = o->stage_cost(o, x, up, t);
    perturb1(u, nu, i, -eps, up); double fm = o->stage_cost(o, x, up, t);
    g[i] = (fp - fm) / (2 * eps);
  }
  free(up);
}

void mas_fd_cost_state_hessian(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) {
  const int n = o->state_dim; const double eps = 1e-5;
  double* xp = dalloc((size_t)n);
  for (int i = 0; i < n; ++i) {
    perturb1(x, n, i, +eps, xp); double fp = safe_stage(o, xp, u, t);
    double f = safe_stage(o, x, u, t);
    perturb1(x, n, i, -eps, xp); double fm = safe_stage(o, xp, u, t);
    H[(size_t)i * n + i] = (fp - 2 * f + fm) / (eps * eps);
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      if (i != j) {
        perturb2(x, n, i, +eps, j, +eps, xp); double fpp = safe_stage(o, xp, u, t);
        perturb2(x, n, i, +eps, j, -eps, xp); double fpm = safe_stage(o, xp, u, t);
        perturb2(x, n, i, -eps, j, +eps, xp); double fmp = safe_stage(o, xp, u, t);
        perturb2(x, n, i, -eps, j, -eps, xp); double fmm = safe_stage(o, xp, u, t);
        H[(size_t)j * n + i] = (fpp - fpm - fmp + fmm) / (4 * eps * eps);   /* H(i,j) */
      }
  free(xp);
}

void mas_fd_cost_control_hessian(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) {
  const int m = o->control_dim; const double eps = 1e-5;
  double* up = dalloc((size_t)m);
  for (int i = 0; i < m; ++i) {
    perturb1(u, m, i, +eps, up); double fp = safe_stage(o, x, up, t);
    double f = safe_stage(o, x, u, t);
    perturb1(u, m, i, -eps, up); double fm = safe_stage(o, x, up, t);
    H[(size_t)i * m + i] = (fp - 2 * f + fm) / (eps * eps);
  }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j)
      if (i != j) {
        perturb2(u, m, i, +eps, j, +eps, up); double fpp = safe_stage(o, x, up, t);
        perturb2(u, m, i, +eps, j, -eps, up); double fpm = safe_stage(o, x, up, t);
        perturb2(u, m, i, -eps, j, +eps, up); double fmp = safe_stage(o, x, up, t);
        perturb2(u, m, i, -eps, j, -eps, up); double fmm = safe_stage(o, x, up, t);
        H[(size_t)j * m + i] = (fpp - fpm - fmp + fmm) / (4 * eps * eps);
      }
  free(up);
}

void mas_fd_terminal_cost_gradient(const mas_ocp* o, const double* x, double* g) {
  const int nx = o->state_dim; const double eps = 1e-6;
  double* xp = dalloc((size_t)nx);
  for (int i = 0; i < nx; ++i) {
    perturb1(x, nx, i, +eps, xp); double fp = o->terminal_cost(o, xp);
    perturb1(x, nx, i, -eps, xp); double fm = o->terminal_cost(o, xp);
    g[i] = (fp - fm) / (2 * eps);
  }
  free(xp);
}

void mas_fd_terminal_cost_hessian(const mas_ocp* o, const double* x, double* H) {
  const int n = o->state_dim; const double eps = 1e-5;
  double* xp = dalloc((size_t)n);
  for (int i = 0; i < n; ++i) {
    perturb1(x, n, i, +eps, xp); double fp = safe_terminal(o, xp);
    double f = safe_terminal(o, x);
    perturb1(x, n, i, -eps, xp); double fm = safe_terminal(o, xp);
    H[(size_t)i * n + i] = (fp - 2 * f + fm) / (eps * eps);
  }
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j)
      if (i != j) {
        perturb2(x, n, i, +eps, j, +eps, xp); double fpp = safe_terminal(o, xp);
        perturb2(x, n, i, +eps, j, -eps, xp); double fpm = safe_terminal(o, xp);
        perturb2(x, n, i, -eps, j, +eps, xp); double fmp = safe_terminal(o, xp);
        perturb2(x, n, i, -eps, j, -eps, xp); double fmm = safe_terminal(o, xp);
        H[(size_t)j * n + i] = (fpp - fpm - fmp + fmm) / (4 * eps * eps);
      }
  free(xp);
}

/* compute_cost_cross_term, finite_differences.hpp:263-287: H is nu x nx, eps=1e-6,
 * f_pm = f(x-dx, u+du), f_mp = f(x+dx, u-du). */
void mas_fd_cost_cross_term(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) {
  const int m = o->control_dim, n = o->state_dim; const double eps = 1e-6;
  double* w = dalloc((size_t)2 * n + 2 * m);
  double *xp = w, *xm = w + n, *up = w + 2 * n, *um = w + 2 * n + m;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < n; ++j) {
      perturb1(u, m, i, +eps, up); perturb1(u, m, i, -eps, um);
      perturb1(x, n, j, +eps, xp); perturb1(x, n, j, -eps, xm);
      double fpp = safe_stage(o, xp, up, t);
      double fpm = safe_stage(o, xm, up, t);
      double fmp = safe_stage(o, xp, um, t);
      double fmm = safe_stage(o, xm, um, t);
      H[(size_t)j * m + i] = (fpp - fpm - fmp + fmm) / (4 * eps * eps);
    }
  free(w);
}

/* finite_differences_gradient, finite_differences.hpp:17-48 (CGD's routine; pinned by
 * the reference's unit test 4). */
void mas_finite_differences_gradient(const mas_ocp* o, const double* x0, const double* U, double dt, double* G) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  double* Up = dalloc((size_t)nu * T); double* Xp = dalloc((size_t)nx * (T + 1));
  for (int t = 0; t < T; ++t)
    for (int i = 0; i < nu; ++i) {
      double c = U[(size_t)t * nu + i];
      double eps = fmax(1e-6, 1e-8 * fabs(c));
      memcpy(Up, U, sizeof(double) * nu * T); Up[(size_t)t * nu + i] += eps;
      mas_integrate_horizon(o, x0, Up, dt, Xp); double cp = mas_ocp_objective(o, Xp, Up);
      memcpy(Up, U, sizeof(double) * nu * T); Up[(size_t)t * nu + i] -= eps;
      mas_integrate_horizon(o, x0, Up, dt, Xp); double cm = mas_ocp_objective(o, Xp, Up);
      G[(size_t)t * nu + i] = (cp - cm) / (2 * eps);
    }
  free(Up); free(Xp);
}

/* compute_constraints_state_jacobian / _control_jacobian, finite_differences.hpp:289-345: central
 * differences with eps = 1e-6 (the reference first evaluates the constraint once for its size). */
static void eval_constraint(const mas_ocp* o, int which, const double* x, const double* u, double* out) {
  if (which == 0) o->equality_constraints(o, x, u, out); else o->inequality_constraints(o, x, u, out);
}
void mas_fd_constraints_state_jacobian(const mas_ocp* o, int which, const double* x, const double* u, double* J) {
  const int n = o->state_dim, m = which == 0 ? o->eq_dim : o->ineq_dim; const double eps = 1e-6;
  double* w = dalloc((size_t)n + 2 * m); double *xp = w, *fp = w + n, *fm = w + n + m;
  for (int i = 0; i < n; ++i) {
    perturb1(x, n, i, +eps, xp); eval_constraint(o, which, xp, u, fp);
    perturb1(x, n, i, -eps, xp); eval_constraint(o, which, xp, u, fm);
    for (int r = 0; r < m; ++r) J[(size_t)i * m + r] = (fp[r] - fm[r]) / (2 * eps);
  }
  free(w);
}
void mas_fd_constraints_control_jacobian(const mas_ocp* o, int which, const double* x, const double* u, double* J) {
  const int p = o->control_dim, m = which == 0 ? o->eq_dim : o->ineq_dim; const double eps = 1e-6;
  double* w = dalloc((size_t)p + 2 * m); double *up = w, *fp = w + p, *fm = w + p + m;
  for (int i = 0; i < p; ++i) {
    perturb1(u, p, i, +eps, up); eval_constraint(o, which, x, up, fp);
    perturb1(u, p, i, -eps, up); eval_constraint(o, which, x, up, fm);
    for (int r = 0; r < m; ++r) J[(size_t)i * m + r] = (fp[r] - fm[r]) / (2 * eps);
  }
  free(w);
}

/* ==================================================================== iLQR */

void mas_ilqr_init(mas_ilqr* s) {
  memset(s, 0, sizeof(*s));
  s->max_iterations = 50; s->tolerance = 1e-6; s->max_ms = INFINITY; s->debug = 0;  /* ilqr.hpp:26-37 */
  s->penalty_parameter = 10.0; s->penalty_increase = 5.0; s->constraint_tolerance = 1e-4; s->inequality_activation_tolerance = 1e-6;
}
void mas_ilqr_free(mas_ilqr* s) { free(s->ws); s->ws = NULL; free(s->eq_multipliers); free(s->ineq_multipliers); s->eq_multipliers = s->ineq_multipliers = NULL; }
void mas_ilqr_set_al_params(mas_ilqr* s, double penalty, double penalty_increase, double constraint_tolerance, double inequality_activation_tolerance) {
  s->penalty_parameter = penalty; s->penalty_increase = penalty_increase; s->constraint_tolerance = constraint_tolerance;
  s->inequality_activation_tolerance = inequality_activation_tolerance;
}
void mas_ilqr_set_params(mas_ilqr* s, int max_iterations, double tolerance, double max_ms, int debug) {
  s->max_iterations = max_iterations; s->tolerance = tolerance; s->max_ms = max_ms; s->debug = debug;
}

/* ---- arithmetic variants (test infrastructure for the POPULATION tolerance) ----
 * Two readings of Eigen this restatement cannot check against a real Eigen build:
 *   sum order     0 = products summed in index order from 0.0 (the default: coefficient-based lazy product)
 *                 1 = SSE2 packet-pairwise: even and odd terms summed separately, then added -- (a0b0+a2b2)+(a1b1+a3b3)
 *   symmetrise    0 = the aliased in-place statement below (no temporary)
 *                 1 = the mathematical mean 0.5 (V + V^T) of the OLD values (as if Eigen evaluated into a temporary)
 * The parity tests always run variant (0, 0); tests/test_population_tolerance.py runs the others to bound what the unverifiable
 * readings, the trig library and one-ulp input changes can move on the POPULATION level. */
static int g_sum_order = 0, g_symmetrise = 0;
void mas_oracle_set_arith_variant(int sum_order, int symmetrise) { g_sum_order = sum_order; g_symmetrise = symmetrise; }

/* sum_k a[k*sa] * b[k*sb], k < p, in the order of the selected variant */
static double dot_strided(const double* a, size_t sa, const double* b, size_t sb, int p) {
  if (g_sum_order == 0 || p < 4) {
    double s = 0.0;
    for (int k = 0; k < p; ++k) s += a[(size_t)k * sa] * b[(size_t)k * sb];
    return s;
  }
  double even = a[0] * b[0], odd = a[sa] * b[sb];
  int k = 2;
  for (; k + 1 < p; k += 2) { even += a[(size_t)k * sa] * b[(size_t)k * sb]; odd += a[(size_t)(k + 1) * sa] * b[(size_t)(k + 1) * sb]; }
  double s = even + odd;
  if (k < p) s += a[(size_t)k * sa] * b[(size_t)k * sb];
  return s;
}

/* in-place  V = 0.5 * (V + V^T)  exactly as Eigen executes the reference's aliased
 * statement (ilqr.hpp:102,192): no temporary is made for a cwise expression, the
 * mixed storage orders force the scalar column-major traversal, so entry (i,j), i>j,
 * is averaged from the old values and entry (j,i) then sees the already-updated (i,j). */
static void symmetrise_inplace(double* V, int n) {
  if (g_symmetrise == 1) {                       /* variant: every entry from the old values */
    for (int j = 0; j < n; ++j)
      for (int i = j + 1; i < n; ++i) {
        const double m = 0.5 * (V[(size_t)j * n + i] + V[(size_t)i * n + j]);
        const double m2 = 0.5 * (V[(size_t)i * n + j] + V[(size_t)j * n + i]);
        V[(size_t)j * n + i] = m; V[(size_t)i * n + j] = m2;
      }
    for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 0.5 * (V[(size_t)i * n + i] + V[(size_t)i * n + i]);
    return;
  }
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i)
      V[(size_t)j * n + i] = 0.5 * (V[(size_t)j * n + i] + V[(size_t)i * n + j]);
}

/* Eigen::LLT<MatrixXd>::compute (unblocked, lower), Eigen 3.4 LLT.h llt_inplace::unblocked.
 * Returns 0 on success, 1 if a pivot x <= 0 was met (NaN pivots pass, as in Eigen). */
static int llt_compute(double* M, int n) {
  for (int k = 0; k < n; ++k) {
    double x = M[(size_t)k * n + k];
    if (k > 0) { double s = 0.0; for (int j = 0; j < k; ++j) s += M[(size_t)j * n + k] * M[(size_t)j * n + k]; x = M[(size_t)k * n + k] - s; }
    if (x <= 0.0) return 1;
    x = sqrt(x);
    M[(size_t)k * n + k] = x;
    for (int r = k + 1; r < n; ++r) {
      double a = M[(size_t)k * n + r];
      for (int j = 0; j < k; ++j) a += (-M[(size_t)j * n + k]) * M[(size_t)j * n + r]; /* A21 -= A20*A10^T, column by column */
      M[(size_t)k * n + r] = a / x;
    }
  }
  return 0;
}

/* LLT::solve(Identity): L y = I then L^T x = y, following Eigen's small triangular
 * solver (reciprocal of the diagonal, then multiply). */
static void llt_solve_identity(const double* L, int n, double* Inv) {
  for (int c = 0; c < n; ++c) {
    double* o = Inv + (size_t)c * n;
    for (int i = 0; i < n; ++i) o[i] = (i == c) ? 1.0 : 0.0;
    for (int i = 0; i < n; ++i) {                 /* forward, column-major L */
      double a = 1.0 / L[(size_t)i * n + i];
      o[i] *= a;
      double b = o[i];
      for (int r = i + 1; r < n; ++r) o[r] -= b * L[(size_t)i * n + r];
    }
    for (int i = n - 1; i >= 0; --i) {            /* backward with U = L^T */
      double a = 1.0 / L[(size_t)i * n + i];
      double b = 0.0;
      for (int r = i + 1; r < n; ++r) b += L[(size_t)i * n + r] * o[r];
      o[i] = (o[i] - b) * a;
    }
  }
}

/* C(m x n) = A^T(m x p from A: p x m) * B(p x n) */
static void mat_AtB(const double* A, int p, int m, const double* B, int n, double* C) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) C[(size_t)j * m + i] = dot_strided(A + (size_t)i * p, 1, B + (size_t)j * p, 1, p);
}
/* C(m x n) = A(m x p) * B(p x n) */
static void mat_AB(const double* A, int m, int p, const double* B, int n, double* C) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) C[(size_t)j * m + i] = dot_strided(A + i, (size_t)m, B + (size_t)j * p, 1, p);
}

typedef struct {
  int nx, nu;
  double *A, *B, *lx, *lu, *lxx, *luu, *lux;            /* linearisation at one step */
  double *qx, *qu, *qxx, *qux, *quu, *quu_reg, *quu_inv;
  double *tmp_xx, *tmp_ux, *tmp_x, *tmp_u, *tmp_uu;
} step_ws;

static size_t step_ws_size(int nx, int nu) {
  return (size_t)nx * nx + nx * nu + nx + nu + nx * nx + nu * nu + nu * nx   /* A..lux */
       + nx + nu + nx * nx + nu * nx + nu * nu * 3                           /* q's    */
       + nx * nx + nu * nx + nx + nu + nu * nu + nu * nx;                    /* tmps   */
}
static void step_ws_bind(step_ws* w, double* p, int nx, int nu) {
  w->nx = nx; w->nu = nu;
  w->A = p; p += nx * nx; w->B = p; p += nx * nu; w->lx = p; p += nx; w->lu = p; p += nu;
  w->lxx = p; p += nx * nx; w->luu = p; p += nu * nu; w->lux = p; p += nu * nx;
  w->qx = p; p += nx; w->qu = p; p += nu; w->qxx = p; p += nx * nx; w->qux = p; p += nu * nx;
  w->quu = p; p += nu * nu; w->quu_reg = p; p += nu * nu; w->quu_inv = p; p += nu * nu;
  w->tmp_xx = p; p += nx * nx; w->tmp_ux = p; p += nu * nx; w->tmp_x = p; p += nx; w->tmp_u = p; p += nu;
  w->tmp_uu = p; p += nu * nu + nu * nx;
}

/* linearise one step: ilqr.hpp:106-113 through the OCP hooks (NULL => FD, ocp.hpp:117-135) */
static void linearise_step(const mas_ocp* o, const double* x, const double* u, size_t t, step_ws* w) {
  if (o->dynamics_state_jacobian) o->dynamics_state_jacobian(o, x, u, w->A); else mas_fd_dynamics_state_jacobian(o, x, u, w->A);
  if (o->dynamics_control_jacobian) o->dynamics_control_jacobian(o, x, u, w->B); else mas_fd_dynamics_control_jacobian(o, x, u, w->B);
  if (o->cost_state_gradient) o->cost_state_gradient(o, x, u, t, w->lx); else mas_fd_cost_state_gradient(o, x, u, t, w->lx);
  if (o->cost_control_gradient) o->cost_control_gradient(o, x, u, t, w->lu); else mas_fd_cost_control_gradient(o, x, u, t, w->lu);
  if (o->cost_state_hessian) o->cost_state_hessian(o, x, u, t, w->lxx); else mas_fd_cost_state_hessian(o, x, u, t, w->lxx);
  if (o->cost_control_hessian) o->cost_control_hessian(o, x, u, t, w->luu); else mas_fd_cost_control_hessian(o, x, u, t, w->luu);
  if (o->cost_cross_term) o->cost_cross_term(o, x, u, t, w->lux); else mas_fd_cost_cross_term(o, x, u, t, w->lux);
}

/* Q-function assembly, ilqr.hpp:115-119 */
static void assemble_q(step_ws* w, const double* vx, const double* vxx) {
  const int nx = w->nx, nu = w->nu;
  mat_AtB(w->A, nx, nx, vx, 1, w->tmp_x);   for (int i = 0; i < nx; ++i) w->qx[i] = w->lx[i] + w->tmp_x[i];
  mat_AtB(w->B, nx, nu, vx, 1, w->tmp_u);   for (int i = 0; i < nu; ++i) w->qu[i] = w->lu[i] + w->tmp_u[i];
  mat_AtB(w->A, nx, nx, vxx, nx, w->tmp_xx);                 /* (A^T V)       */
  mat_AB(w->tmp_xx, nx, nx, w->A, nx, w->qxx);               /* (A^T V) A     */
  for (int i = 0; i < nx * nx; ++i) w->qxx[i] = w->lxx[i] + w->qxx[i];
  mat_AtB(w->B, nx, nu, vxx, nx, w->tmp_ux);                 /* (B^T V) nu x nx */
  mat_AB(w->tmp_ux, nu, nx, w->A, nx, w->qux);
  for (int i = 0; i < nu * nx; ++i) w->qux[i] = w->lux[i] + w->qux[i];
  mat_AB(w->tmp_ux, nu, nx, w->B, nu, w->quu);
  for (int i = 0; i < nu * nu; ++i) w->quu[i] = w->luu[i] + w->quu[i];
}

/* gains + value update, ilqr.hpp:183-192.  quu_reg holds the factor L on entry. */
static void gains_and_value(step_ws* w, double* k, double* K, double* vx, double* vxx) {
  const int nx = w->nx, nu = w->nu;
  llt_solve_identity(w->quu_reg, nu, w->quu_inv);
  for (int i = 0; i < nu * nu; ++i) w->tmp_uu[i] = -w->quu_inv[i];          /* unary minus first */
  mat_AB(w->tmp_uu, nu, nu, w->qu, 1, k);                                   /* k = -Quu^-1 q_u   */
  mat_AB(w->tmp_uu, nu, nu, w->qux, nx, K);                                 /* K = -Quu^-1 q_ux  */
  /* v_x = q_x + K^T q_u + q_ux^T k + (K^T q_uu) k   -- unregularised q_uu */
  double* KtQuu = w->tmp_uu + nu * nu;                                       /* nx x nu */
  mat_AtB(K, nu, nx, w->quu, nu, KtQuu);
  mat_AtB(K, nu, nx, w->qu, 1, w->tmp_x);
  for (int i = 0; i < nx; ++i) vx[i] = w->qx[i] + w->tmp_x[i];
  mat_AtB(w->qux, nu, nx, k, 1, w->tmp_x);
  for (int i = 0; i < nx; ++i) vx[i] = vx[i] + w->tmp_x[i];
  mat_AB(KtQuu, nx, nu, k, 1, w->tmp_x);
  for (int i = 0; i < nx; ++i) vx[i] = vx[i] + w->tmp_x[i];
  /* v_xx = q_xx + K^T q_ux + q_ux^T K + (K^T q_uu) K */
  mat_AtB(K, nu, nx, w->qux, nx, w->tmp_xx);
  for (int i = 0; i < nx * nx; ++i) vxx[i] = w->qxx[i] + w->tmp_xx[i];
  mat_AtB(w->qux, nu, nx, K, nx, w->tmp_xx);
  for (int i = 0; i < nx * nx; ++i) vxx[i] = vxx[i] + w->tmp_xx[i];
  mat_AB(KtQuu, nx, nu, K, nx, w->tmp_xx);
  for (int i = 0; i < nx * nx; ++i) vxx[i] = vxx[i] + w->tmp_xx[i];
  symmetrise_inplace(vxx, nx);
}

/* The reference's ladder is `while(true)` (ilqr.hpp:175-182) but it always ends: reg is
 * multiplied by 10 per rung, overflows to +inf after ~315 rungs, and a pivot of +inf or
 * NaN is not `<= 0`, so Eigen's LLT then reports Success.  400 is therefore never reached;
 * it only guards the oracle against a logic error. */
#define MAS_REG_LADDER_MAX 400

/* diagnostic: how many rungs each (solve, time step) needed -- index min(rungs, 63); not thread-exact (plain increments from the
 * OpenMP loop), good enough for a histogram; read and cleared through mas_oracle_rung_histogram */
static long g_rung_hist[64];
void mas_oracle_rung_histogram(long* out, int clear) { for (int i = 0; i < 64; ++i) { if (out) out[i] = g_rung_hist[i]; if (clear) g_rung_hist[i] = 0; } }

/* LLT with the x10 regularisation ladder, ilqr.hpp:172-182.  Returns rungs used, or -1. */
static int llt_ladder(step_ws* w) {
  const int nu = w->nu;
  double acc[64]; /* running q_uu_reg (pre-factorisation) */
  double* cur = nu * nu <= 64 ? acc : dalloc((size_t)nu * nu);
  memcpy(cur, w->quu, sizeof(double) * nu * nu);
  double reg = 1e-6; int rungs = 0;
  for (;;) {
    memcpy(w->quu_reg, cur, sizeof(double) * nu * nu);
    if (llt_compute(w->quu_reg, nu) == 0) break;
    if (rungs >= MAS_REG_LADDER_MAX) { rungs = -1; break; }
    for (int i = 0; i < nu; ++i) cur[(size_t)i * nu + i] += reg;
    reg *= 10.0; ++rungs;
  }
  if (cur != acc) free(cur);
  if (rungs >= 0) g_rung_hist[rungs < 63 ? rungs : 63]++;
  return rungs;
}

static void terminal_boundary(const mas_ocp* o, const double* xT, double* vx, double* vxx) {
  if (o->terminal_cost_gradient) o->terminal_cost_gradient(o, xT, vx); else mas_fd_terminal_cost_gradient(o, xT, vx);
  if (o->terminal_cost_hessian) o->terminal_cost_hessian(o, xT, vxx); else mas_fd_terminal_cost_hessian(o, xT, vxx);
  symmetrise_inplace(vxx, o->state_dim);   /* ilqr.hpp:102 */
}

long mas_ilqr_backward_pass(mas_ocp* o, const double* X, const double* U, double* k, double* K, double* derivs) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  double* buf = dalloc(step_ws_size(nx, nu) + nx + (size_t)nx * nx + nu + (size_t)nu * nx);
  step_ws w; step_ws_bind(&w, buf, nx, nu);
  double* vx = buf + step_ws_size(nx, nu); double* vxx = vx + nx; double* kt = vxx + (size_t)nx * nx; double* Kt = kt + nu;
  const size_t dstride = (size_t)nx * nx + nx * nu + nx + nu + nx * nx + nu * nu + nu * nx;
  long regs = 0;
  terminal_boundary(o, X + (size_t)T * nx, vx, vxx);
  for (int t = T - 1; t >= 0; --t) {
    linearise_step(o, X + (size_t)t * nx, U + (size_t)t * nu, (size_t)t, &w);
    if (derivs) memcpy(derivs + dstride * t, w.A, sizeof(double) * dstride);
    assemble_q(&w, vx, vxx);
    int r = llt_ladder(&w); if (r > 0) regs += r;
    gains_and_value(&w, kt, Kt, vx, vxx);
    if (k) memcpy(k + (size_t)t * nu, kt, sizeof(double) * nu);
    if (K) memcpy(K + (size_t)t * nu * nx, Kt, sizeof(double) * nu * nx);
  }
  free(buf);
  return regs;
}

static double now_ms(void) {
  struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* compute_merit, ilqr.hpp:379-407 */
double mas_ilqr_compute_merit(const mas_ilqr* s, const mas_ocp* o, const double* X, const double* U) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  const int me = o->equality_constraints ? o->eq_dim : 0, mi = o->inequality_constraints ? o->ineq_dim : 0;
  double merit = mas_ocp_objective(o, X, U);
  if (me == 0 && mi == 0) return merit;
  double r[16];
  for (int t = 0; t < T; ++t) {
    if (me > 0) {
      o->equality_constraints(o, X + (size_t)t * nx, U + (size_t)t * nu, r);
      double dot = 0.0, sq = 0.0;
      for (int k = 0; k < me; ++k) { dot += s->eq_multipliers[(size_t)t * me + k] * r[k]; sq += r[k] * r[k]; }
      merit += dot + 0.5 * s->penalty_parameter * sq;
    }
    if (mi > 0) {
      o->inequality_constraints(o, X + (size_t)t * nx, U + (size_t)t * nu, r);
      double dot = 0.0, sq = 0.0;
      for (int k = 0; k < mi; ++k) {
        const double slack = r[k] > 0.0 ? r[k] : 0.0;                                        /* cwiseMax(0) */
        const double active = (r[k] > -s->inequality_activation_tolerance) ? 1.0 : 0.0;
        const double active_slack = slack * active, weighted = s->ineq_multipliers[(size_t)t * mi + k] * active;
        dot += weighted * active_slack; sq += active_slack * active_slack;
      }
      merit += dot;
      merit += 0.5 * s->penalty_parameter * sq;
    }
  }
  return merit;
}

/* resize of the multiplier storage, ilqr.hpp:325-369: kept when the shape is unchanged, zero otherwise */
static void al_resize(mas_ilqr* s, int T, int me, int mi) {
  if (s->mult_T != T || s->mult_eq_dim != me) { free(s->eq_multipliers); s->eq_multipliers = me > 0 ? dalloc((size_t)T * me) : NULL; }
  if (s->mult_T != T || s->mult_ineq_dim != mi) { free(s->ineq_multipliers); s->ineq_multipliers = mi > 0 ? dalloc((size_t)T * mi) : NULL; }
  s->mult_T = T; s->mult_eq_dim = me; s->mult_ineq_dim = mi;
}

/* augmented-Lagrangian terms of the Q-function at one step, ilqr.hpp:121-170 */
static void al_terms(const mas_ilqr* s, const mas_ocp* o, const double* x, const double* u, int t, step_ws* w) {
  const int nx = w->nx, nu = w->nu;
  const int me = o->equality_constraints ? o->eq_dim : 0, mi = o->inequality_constraints ? o->ineq_dim : 0;
  const double rho = s->penalty_parameter;
  double res[16], dual[16], act[16], Jx[16 * 16], Ju[16 * 16];
  for (int which = 0; which < 2; ++which) {
    const int m = which == 0 ? me : mi;
    if (m == 0) continue;
    eval_constraint(o, which, x, u, res);
    mas_xu_mat_fn jx = which == 0 ? o->equality_constraints_state_jacobian : o->inequality_constraints_state_jacobian;
    mas_xu_mat_fn ju = which == 0 ? o->equality_constraints_control_jacobian : o->inequality_constraints_control_jacobian;
    if (jx) jx(o, x, u, Jx); else mas_fd_constraints_state_jacobian(o, which, x, u, Jx);
    if (ju) ju(o, x, u, Ju); else mas_fd_constraints_control_jacobian(o, which, x, u, Ju);
    int any_active = 0;
    for (int k = 0; k < m; ++k) {
      if (which == 0) { dual[k] = s->eq_multipliers[(size_t)t * m + k] + rho * res[k]; act[k] = 1.0; any_active = 1; }
      else {
        const double slack = res[k] > 0.0 ? res[k] : 0.0;
        act[k] = (res[k] > -s->inequality_activation_tolerance) ? 1.0 : 0.0;
        dual[k] = s->ineq_multipliers[(size_t)t * m + k] * act[k] + rho * slack * act[k];
        if (act[k] != 0.0) any_active = 1;
      }
    }
    /* q_x += Jx^T dual ; q_u += Ju^T dual */
    for (int i = 0; i < nx; ++i) { double a = 0.0; for (int k = 0; k < m; ++k) a += Jx[(size_t)i * m + k] * dual[k]; w->qx[i] += a; }
    for (int i = 0; i < nu; ++i) { double a = 0.0; for (int k = 0; k < m; ++k) a += Ju[(size_t)i * m + k] * dual[k]; w->qu[i] += a; }
    if (!any_active) continue;                       /* :163 (equalities always contribute) */
    /* q_xx += (rho Jx^T) [diag(active)] Jx  etc.; for equalities there is no diag factor */
    for (int j = 0; j < nx; ++j) for (int i = 0; i < nx; ++i) {
      double a = 0.0;
      for (int k = 0; k < m; ++k) { double l = rho * Jx[(size_t)i * m + k]; if (which == 1) l = l * act[k]; a += l * Jx[(size_t)j * m + k]; }
      w->qxx[(size_t)j * nx + i] += a;
    }
    for (int j = 0; j < nx; ++j) for (int i = 0; i < nu; ++i) {
      double a = 0.0;
      for (int k = 0; k < m; ++k) { double l = rho * Ju[(size_t)i * m + k]; if (which == 1) l = l * act[k]; a += l * Jx[(size_t)j * m + k]; }
      w->qux[(size_t)j * nu + i] += a;
    }
    for (int j = 0; j < nu; ++j) for (int i = 0; i < nu; ++i) {
      double a = 0.0;
      for (int k = 0; k < m; ++k) { double l = rho * Ju[(size_t)i * m + k]; if (which == 1) l = l * act[k]; a += l * Ju[(size_t)j * m + k]; }
      w->quu[(size_t)j * nu + i] += a;
    }
  }
}

/* multiplier and penalty updates after the line search, ilqr.hpp:236-260 */
static void al_update(mas_ilqr* s, const mas_ocp* o, const double* X, const double* U) {
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  const int me = o->equality_constraints ? o->eq_dim : 0, mi = o->inequality_constraints ? o->ineq_dim : 0;
  double eqv = 0.0, inv = 0.0, r[16];
  for (int t = 0; t < T; ++t) {
    if (me > 0) {
      o->equality_constraints(o, X + (size_t)t * nx, U + (size_t)t * nu, r);
      double sq = 0.0;
      for (int k = 0; k < me; ++k) { s->eq_multipliers[(size_t)t * me + k] += s->penalty_parameter * r[k]; sq += r[k] * r[k]; }
      eqv += sq;
    }
    if (mi > 0) {
      o->inequality_constraints(o, X + (size_t)t * nx, U + (size_t)t * nu, r);
      double sq = 0.0;
      for (int k = 0; k < mi; ++k) {
        const double positive = r[k] > 0.0 ? r[k] : 0.0;
        const double v = s->ineq_multipliers[(size_t)t * mi + k] + s->penalty_parameter * positive;
        s->ineq_multipliers[(size_t)t * mi + k] = v > 0.0 ? v : 0.0;
        sq += positive * positive;
      }
      inv += sq;
    }
  }
  s->last_eq_violation = sqrt(eqv); s->last_ineq_violation = sqrt(inv);
  if (s->last_eq_violation > s->constraint_tolerance || s->last_ineq_violation > s->constraint_tolerance)
    s->penalty_parameter *= s->penalty_increase;
}

/* iLQR::solve, ilqr.hpp:58-273.  Without constraint functions (every shipped example) merit == objective
 * (ilqr.hpp:379-407) and the multiplier/penalty updates are no-ops. */
void mas_ilqr_solve(mas_ilqr* s, mas_ocp* o) {
  const double start = now_ms();
  const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
  const double dt = o->dt;
  const size_t nX = (size_t)nx * (T + 1), nU = (size_t)nu * T;
  const size_t need = step_ws_size(nx, nu) + nx + (size_t)nx * nx + (size_t)T * nu + (size_t)T * nu * nx + 2 * nX + 2 * nU + nx;
  if (!s->ws || s->ws_nx != nx || s->ws_nu != nu || s->ws_T != T) {
    free(s->ws); s->ws = dalloc(need); s->ws_nx = nx; s->ws_nu = nu; s->ws_T = T;
  }
  double* p = s->ws;
  step_ws w; step_ws_bind(&w, p, nx, nu); p += step_ws_size(nx, nu);
  double* vx = p; p += nx; double* vxx = p; p += (size_t)nx * nx;
  double* k = p; p += (size_t)T * nu; double* K = p; p += (size_t)T * nu * nx;
  double* x_trial = p; p += nX; double* u_trial = p; p += nU;
  double* best_x = p; p += nX; double* best_u = p; p += nU; double* dx = p; p += nx;
  memset(k, 0, sizeof(double) * T * nu); memset(K, 0, sizeof(double) * T * nu * nx);   /* resize_buffers :292-293 */

  double* x = o->best_states; double* u = o->best_controls;
  s->iterations = 0; s->exit_reason = MAS_EXIT_MAX_ITER; s->reg_events = 0; s->trials = 0;
  memset(s->alpha_hist, 0, sizeof(s->alpha_hist));

  const int has_al = (o->equality_constraints && o->eq_dim > 0) || (o->inequality_constraints && o->ineq_dim > 0);
  al_resize(s, T, o->equality_constraints ? o->eq_dim : 0, o->inequality_constraints ? o->ineq_dim : 0);
  s->last_eq_violation = s->last_ineq_violation = 0.0;
  mas_integrate_horizon(o, o->initial_state, u, dt, x);      /* :75 warm start = best_controls */
  o->best_cost = mas_ocp_objective(o, x, u);                 /* :76 */
  double current_merit = mas_ilqr_compute_merit(s, o, x, u); /* :78 */
  if (s->debug) printf("iLQR initial cost=%.17g merit=%.17g\n", o->best_cost, current_merit);

  for (int iter = 0; iter < s->max_iterations; ++iter) {
    const double elapsed_ms = (double)(long long)(now_ms() - start);   /* duration_cast<milliseconds> truncates */
    if (elapsed_ms > s->max_ms) { s->exit_reason = MAS_EXIT_TIME; break; }
    s->iterations = iter + 1;

    terminal_boundary(o, x + (size_t)T * nx, vx, vxx);       /* :92-102 */
    int reg_fail = 0;
    for (int t = T - 1; t >= 0; --t) {                       /* :104-193 */
      linearise_step(o, x + (size_t)t * nx, u + (size_t)t * nu, (size_t)t, &w);
      assemble_q(&w, vx, vxx);
      if (has_al) al_terms(s, o, x + (size_t)t * nx, u + (size_t)t * nu, t, &w);
      int r = llt_ladder(&w);
      if (r < 0) { reg_fail = 1; break; }
      s->reg_events += r;
      gains_and_value(&w, k + (size_t)t * nu, K + (size_t)t * nu * nx, vx, vxx);
    }
    if (reg_fail) { s->exit_reason = MAS_EXIT_REG_LIMIT; break; }

    memset(x_trial, 0, sizeof(double) * nX); memset(u_trial, 0, sizeof(double) * nU);   /* :195-197 */
    memcpy(x_trial, o->initial_state, sizeof(double) * nx);
    const double amin = 1e-3; double alpha = 1.0;
    double best_merit = current_merit;
    memcpy(best_x, x, sizeof(double) * nX); memcpy(best_u, u, sizeof(double) * nU);
    int accepted_j = 10, j = 0;
    while (alpha >= amin) {                                  /* :206-228 */
      for (int t = 0; t < T; ++t) {
        const double* xt = x_trial + (size_t)t * nx; double* ut = u_trial + (size_t)t * nu;
        for (int i = 0; i < nx; ++i) dx[i] = xt[i] - x[(size_t)t * nx + i];
        const double* Kt = K + (size_t)t * nu * nx; const double* kt = k + (size_t)t * nu;
        for (int i = 0; i < nu; ++i) {
          double kd = 0.0; for (int c = 0; c < nx; ++c) kd += Kt[(size_t)c * nu + i] * dx[c];
          ut[i] = (u[(size_t)t * nu + i] + alpha * kt[i]) + kd;
        }
        if (o->has_input_lower && o->has_input_upper) mas_clamp_controls_column(o, ut);
        mas_integrate_rk4(o, xt, ut, dt, x_trial + (size_t)(t + 1) * nx);
      }
      ++s->trials;
      const double trial_merit = mas_ilqr_compute_merit(s, o, x_trial, u_trial);
      if (trial_merit < best_merit) {
        best_merit = trial_merit;
        memcpy(best_x, x_trial, sizeof(double) * nX); memcpy(best_u, u_trial, sizeof(double) * nU);
        accepted_j = j;
        break;
      }
      alpha *= 0.5; ++j;
    }
    s->alpha_hist[accepted_j]++;
    const double improvement = current_merit - best_merit;   /* :230-234 */
    memcpy(x, best_x, sizeof(double) * nX); memcpy(u, best_u, sizeof(double) * nU);
    o->best_cost = mas_ocp_objective(o, x, u);
    current_merit = best_merit;
    if (has_al) al_update(s, o, x, u);                        /* :236-260; current_merit is NOT recomputed */
    if (iter < s->trace_cap) {
      if (s->trace_cost) s->trace_cost[iter] = o->best_cost;
      if (s->trace_alpha) s->trace_alpha[iter] = accepted_j < 10 ? ldexp(1.0, -accepted_j) : 0.0;
    }
    if (s->debug)
      printf("iLQR iter %d: cost=%.17g merit=%.17g d_merit=%.17g\n", iter, o->best_cost, current_merit, improvement);
    if (improvement < s->tolerance && s->last_eq_violation < s->constraint_tolerance &&
        s->last_ineq_violation < s->constraint_tolerance) { s->exit_reason = MAS_EXIT_CONVERGED; break; }   /* :269-271 */
  }
}

/* ============================================================== strategies */

static void stats_add(mas_strategy_stats* st, const mas_ilqr* s) {
  if (!st) return;
  st->total_iterations += s->iterations;
  st->converged_solves += (s->exit_reason == MAS_EXIT_CONVERGED);
  st->reg_events += s->reg_events; st->trials += s->trials;
  for (int i = 0; i < 12; ++i) st->alpha_hist[i] += s->alpha_hist[i];
}

static double collect_total(mas_ocp** ocps, int n) {   /* collect_solution, nash.hpp:23-37 */
  double c = 0.0; for (int i = 0; i < n; ++i) c += ocps[i]->best_cost; return c;
}

/* optional per-agent exit reason of each agent's LAST solve (enum mas_exit_reason); the reference's solve() returns
 * nothing (ilqr.hpp:58-273), the product reports it as an extension and the tests check it against this */
static int* g_exit_out = NULL;
void mas_oracle_set_exit_buffer(int* buf) { g_exit_out = buf; }
/* the trust radii the trust-region strategy ends with (nash.hpp:184,231,241), handed to mas_run_trust_region by mas_oracle_solve_batch */
static double* g_radii_out = NULL;
void mas_oracle_set_radii_buffer(double* buf) { g_radii_out = buf; }
static void report_exits(const mas_ilqr* sv, int n) { if (g_exit_out) for (int i = 0; i < n; ++i) g_exit_out[i] = sv[i].exit_reason; }

static double g_al_params[4] = {10.0, 5.0, 1e-4, 1e-6};
void mas_oracle_set_al_params(double penalty, double penalty_increase, double constraint_tolerance, double inequality_activation_tolerance) {
  g_al_params[0] = penalty; g_al_params[1] = penalty_increase; g_al_params[2] = constraint_tolerance; g_al_params[3] = inequality_activation_tolerance;
}
static mas_ilqr* make_solvers(int n, int max_iterations, double tolerance, double max_ms) {
  mas_ilqr* sv = (mas_ilqr*)calloc((size_t)n, sizeof(mas_ilqr));   /* nash.hpp:78-84: fresh solver per agent */
  for (int i = 0; i < n; ++i) {
    mas_ilqr_init(&sv[i]); mas_ilqr_set_params(&sv[i], max_iterations, tolerance, max_ms, 0);
    mas_ilqr_set_al_params(&sv[i], g_al_params[0], g_al_params[1], g_al_params[2], g_al_params[3]);
  }
  return sv;
}
static void free_solvers(mas_ilqr* sv, int n) { for (int i = 0; i < n; ++i) mas_ilqr_free(&sv[i]); free(sv); }

/* run_sequential, nash.hpp:74-90 with sequential_solve :53-72 */
double mas_run_sequential(mas_ocp** ocps, int n, int max_outer, int max_iterations, double tolerance, double max_ms,
                          mas_strategy_stats* stats, int* iters_per_agent) {
  mas_ilqr* sv = make_solvers(n, max_iterations, tolerance, max_ms);
  mas_strategy_stats* per = (mas_strategy_stats*)calloc((size_t)n, sizeof(mas_strategy_stats));
  if (iters_per_agent) memset(iters_per_agent, 0, sizeof(int) * n);
  for (int outer = 0; outer < max_outer; ++outer) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) { mas_ilqr_solve(&sv[i], ocps[i]); stats_add(&per[i], &sv[i]); }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) mas_ocp_update_initial_with_best(ocps[i]);
  }
  for (int i = 0; i < n; ++i) {
    if (iters_per_agent) iters_per_agent[i] = (int)per[i].total_iterations;
    if (stats) { stats->total_iterations += per[i].total_iterations; stats->converged_solves += per[i].converged_solves;
                 stats->reg_events += per[i].reg_events; stats->trials += per[i].trials;
                 for (int j = 0; j < 12; ++j) stats->alpha_hist[j] += per[i].alpha_hist[j]; }
  }
  report_exits(sv, n);
  free(per); free_solvers(sv, n);
  return collect_total(ocps, n);
}

/* run_trust_region, nash.hpp:182-248 */
double mas_run_trust_region(mas_ocp** ocps, int n, int max_outer, int max_iterations, double tolerance, double max_ms,
                            mas_strategy_stats* stats, double* radii_out, int* iters_per_agent) {
  mas_ilqr* sv = make_solvers(n, max_iterations, tolerance, max_ms);
  mas_strategy_stats* per = (mas_strategy_stats*)calloc((size_t)n, sizeof(mas_strategy_stats));
  double* radii = dalloc((size_t)n); for (int i = 0; i < n; ++i) radii[i] = 1.0;
  for (int outer = 0; outer < max_outer; ++outer) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      mas_ocp* o = ocps[i];
      const int T = o->horizon_steps, nx = o->state_dim, nu = o->control_dim;
      const size_t nX = (size_t)nx * (T + 1), nU = (size_t)nu * T;
      double* old_u = dalloc(nU); double* old_x = dalloc(nX); double* delta = dalloc(nU);
      memcpy(old_u, o->best_controls, sizeof(double) * nU); memcpy(old_x, o->best_states, sizeof(double) * nX);
      const double old_cost = o->best_cost;
      mas_ilqr_solve(&sv[i], o); stats_add(&per[i], &sv[i]);
      double cand_cost = o->best_cost;                   /* cand_u / cand_x live in o->best_* */
      double ss = 0.0;
      for (size_t q = 0; q < nU; ++q) { delta[q] = o->best_controls[q] - old_u[q]; ss += delta[q] * delta[q]; }
      const double norm = sqrt(ss);
      if (norm > radii[i]) {
        const double scale = radii[i] / norm;
        for (size_t q = 0; q < nU; ++q) o->best_controls[q] = old_u[q] + scale * delta[q];
        mas_integrate_horizon(o, o->initial_state, o->best_controls, o->dt, o->best_states);
        cand_cost = mas_ocp_objective(o, o->best_states, o->best_controls);
      }
      if (cand_cost < old_cost) {
        o->best_cost = cand_cost; mas_ocp_update_initial_with_best(o); radii[i] *= 1.5;
      } else {
        memcpy(o->best_controls, old_u, sizeof(double) * nU); memcpy(o->best_states, old_x, sizeof(double) * nX);
        o->best_cost = old_cost; mas_ocp_update_initial_with_best(o); radii[i] *= 0.5;
      }
      free(old_u); free(old_x); free(delta);
    }
  }
  for (int i = 0; i < n; ++i) {
    if (iters_per_agent) iters_per_agent[i] = (int)per[i].total_iterations;
    if (radii_out) radii_out[i] = radii[i];
    if (stats) { stats->total_iterations += per[i].total_iterations; stats->converged_solves += per[i].converged_solves;
                 stats->reg_events += per[i].reg_events; stats->trials += per[i].trials;
                 for (int j = 0; j < 12; ++j) stats->alpha_hist[j] += per[i].alpha_hist[j]; }
  }
  report_exits(sv, n);
  free(per); free(radii); free_solvers(sv, n);
  return collect_total(ocps, n);
}

/* run_line_search, nash.hpp:92-180 */
double mas_run_line_search(mas_ocp** ocps, int n, int max_outer, int max_iterations, double tolerance, double max_ms,
                           mas_strategy_stats* stats, int* iters_per_agent) {
  mas_ilqr* sv = make_solvers(n, max_iterations, tolerance, max_ms);
  mas_strategy_stats* per = (mas_strategy_stats*)calloc((size_t)n, sizeof(mas_strategy_stats));
  double** old_u = (double**)calloc((size_t)n, sizeof(double*)); double** old_x = (double**)calloc((size_t)n, sizeof(double*));
  double** cand_u = (double**)calloc((size_t)n, sizeof(double*));
  double** tr_u = (double**)calloc((size_t)n, sizeof(double*)); double** tr_x = (double**)calloc((size_t)n, sizeof(double*));
  for (int i = 0; i < n; ++i) {
    const size_t nX = (size_t)ocps[i]->state_dim * (ocps[i]->horizon_steps + 1), nU = (size_t)ocps[i]->control_dim * ocps[i]->horizon_steps;
    old_u[i] = dalloc(nU); cand_u[i] = dalloc(nU); tr_u[i] = dalloc(nU); old_x[i] = dalloc(nX); tr_x[i] = dalloc(nX);
  }
  double base_cost = collect_total(ocps, n);              /* total_cost, nash.hpp:39-51 */
  for (int outer = 0; outer < max_outer; ++outer) {
    for (int i = 0; i < n; ++i) {
      const size_t nX = (size_t)ocps[i]->state_dim * (ocps[i]->horizon_steps + 1), nU = (size_t)ocps[i]->control_dim * ocps[i]->horizon_steps;
      memcpy(old_u[i], ocps[i]->best_controls, sizeof(double) * nU); memcpy(old_x[i], ocps[i]->best_states, sizeof(double) * nX);
    }
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) { mas_ilqr_solve(&sv[i], ocps[i]); stats_add(&per[i], &sv[i]); }
    for (int i = 0; i < n; ++i) mas_ocp_update_initial_with_best(ocps[i]);
    const double new_cost = collect_total(ocps, n);
    if (new_cost >= base_cost) {
      for (int i = 0; i < n; ++i) memcpy(cand_u[i], ocps[i]->best_controls, sizeof(double) * ocps[i]->control_dim * ocps[i]->horizon_steps);
      double alpha = 0.5; int accepted = 0;
      while (alpha > 1e-3 && !accepted) {
        double trial_cost = 0.0;
        for (int i = 0; i < n; ++i) {     /* serial sum: the reference's omp reduction order is unspecified */
          mas_ocp* o = ocps[i]; const size_t nU = (size_t)o->control_dim * o->horizon_steps;
          for (size_t q = 0; q < nU; ++q) tr_u[i][q] = old_u[i][q] + alpha * (cand_u[i][q] - old_u[i][q]);
          mas_integrate_horizon(o, o->initial_state, tr_u[i], o->dt, tr_x[i]);
          trial_cost += mas_ocp_objective(o, tr_x[i], tr_u[i]);
        }
        if (trial_cost < base_cost) {
          for (int i = 0; i < n; ++i) {
            mas_ocp* o = ocps[i]; const size_t nX = (size_t)o->state_dim * (o->horizon_steps + 1), nU = (size_t)o->control_dim * o->horizon_steps;
            memcpy(o->best_controls, tr_u[i], sizeof(double) * nU); memcpy(o->best_states, tr_x[i], sizeof(double) * nX);
            o->best_cost = mas_ocp_objective(o, tr_x[i], tr_u[i]); mas_ocp_update_initial_with_best(o);
          }
          base_cost = trial_cost; accepted = 1;
        } else alpha *= 0.5;
      }
      if (!accepted)
        for (int i = 0; i < n; ++i) {
          mas_ocp* o = ocps[i]; const size_t nX = (size_t)o->state_dim * (o->horizon_steps + 1), nU = (size_t)o->control_dim * o->horizon_steps;
          memcpy(o->best_controls, old_u[i], sizeof(double) * nU); memcpy(o->best_states, old_x[i], sizeof(double) * nX);
          o->best_cost = mas_ocp_objective(o, old_x[i], old_u[i]); mas_ocp_update_initial_with_best(o);
        }
    } else base_cost = new_cost;
  }
  for (int i = 0; i < n; ++i) {
    if (iters_per_agent) iters_per_agent[i] = (int)per[i].total_iterations;
    if (stats) { stats->total_iterations += per[i].total_iterations; stats->converged_solves += per[i].converged_solves;
                 stats->reg_events += per[i].reg_events; stats->trials += per[i].trials;
                 for (int j = 0; j < 12; ++j) stats->alpha_hist[j] += per[i].alpha_hist[j]; }
    free(old_u[i]); free(old_x[i]); free(cand_u[i]); free(tr_u[i]); free(tr_x[i]);
  }
  free(old_u); free(old_x); free(cand_u); free(tr_u); free(tr_x); report_exits(sv, n); free(per); free_solvers(sv, n);
  return collect_total(ocps, n);
}

/* ---- stacked global OCP, multi_agent_problem.hpp:52-127 ---- */
typedef struct { mas_ocp** ocps; int n; int* xoff; int* uoff; } global_ctx;

static void global_dynamics(const mas_ocp* g, const double* X, const double* U, double* out) {
  const global_ctx* c = (const global_ctx*)g->ctx;
  for (int b = 0; b < c->n; ++b) c->ocps[b]->dynamics(c->ocps[b], X + c->xoff[b], U + c->uoff[b], out + c->xoff[b]);
}
static double global_stage(const mas_ocp* g, const double* X, const double* U, size_t t) {
  const global_ctx* c = (const global_ctx*)g->ctx; double cost = 0.0;
  for (int b = 0; b < c->n; ++b) cost += c->ocps[b]->stage_cost(c->ocps[b], X + c->xoff[b], U + c->uoff[b], t);
  return cost;
}
static double global_terminal(const mas_ocp* g, const double* X) {
  const global_ctx* c = (const global_ctx*)g->ctx; double cost = 0.0;
  for (int b = 0; b < c->n; ++b) cost += c->ocps[b]->terminal_cost(c->ocps[b], X + c->xoff[b]);
  return cost;
}

/* ocps must already be in id order (compute_offsets sorts by id, multi_agent_problem.hpp:37-50) */
mas_ocp* mas_build_global_ocp(mas_ocp** ocps, int n) {
  int tx = 0, tu = 0;
  global_ctx* c = (global_ctx*)calloc(1, sizeof(global_ctx));
  c->ocps = (mas_ocp**)calloc((size_t)n, sizeof(mas_ocp*)); c->n = n;
  c->xoff = (int*)calloc((size_t)n, sizeof(int)); c->uoff = (int*)calloc((size_t)n, sizeof(int));
  for (int b = 0; b < n; ++b) { c->ocps[b] = ocps[b]; c->xoff[b] = tx; c->uoff[b] = tu; tx += ocps[b]->state_dim; tu += ocps[b]->control_dim; }
  mas_ocp* g = mas_ocp_new(tx, tu, n ? ocps[0]->horizon_steps : 0, n ? ocps[0]->dt : 0.0);
  g->ctx = c;
  int all_bounds = 1;
  for (int b = 0; b < n; ++b) {
    memcpy(g->initial_state + c->xoff[b], ocps[b]->initial_state, sizeof(double) * ocps[b]->state_dim);
    all_bounds &= ocps[b]->has_input_lower && ocps[b]->has_input_upper;
  }
  if (all_bounds) {
    for (int b = 0; b < n; ++b) {
      memcpy(g->input_lower_bounds + c->uoff[b], ocps[b]->input_lower_bounds, sizeof(double) * ocps[b]->control_dim);
      memcpy(g->input_upper_bounds + c->uoff[b], ocps[b]->input_upper_bounds, sizeof(double) * ocps[b]->control_dim);
    }
    g->has_input_lower = g->has_input_upper = 1;
  }
  g->dynamics = global_dynamics; g->stage_cost = global_stage; g->terminal_cost = global_terminal;
  mas_ocp_initialize_problem(g);
  return g;
}
void mas_global_ocp_free(mas_ocp* g) {
  if (!g) return;
  global_ctx* c = (global_ctx*)g->ctx; free(c->ocps); free(c->xoff); free(c->uoff); free(c); mas_ocp_free(g);
}

/* CentralizedStrategy::operator(), centralized.hpp:18-38 */
double mas_run_centralized_dense(mas_ocp** ocps, int n, int max_iterations, double tolerance, double max_ms,
                                 mas_strategy_stats* stats) {
  mas_ocp* g = mas_build_global_ocp(ocps, n);
  const global_ctx* c = (const global_ctx*)g->ctx;
  mas_ilqr s; mas_ilqr_init(&s); mas_ilqr_set_params(&s, max_iterations, tolerance, max_ms, 0);
  mas_ilqr_solve(&s, g); stats_add(stats, &s);
  const int T = g->horizon_steps;
  for (int b = 0; b < n; ++b) {
    mas_ocp* o = ocps[b];
    for (int t = 0; t <= T; ++t) memcpy(o->best_states + (size_t)t * o->state_dim, g->best_states + (size_t)t * g->state_dim + c->xoff[b], sizeof(double) * o->state_dim);
    for (int t = 0; t < T; ++t) memcpy(o->best_controls + (size_t)t * o->control_dim, g->best_controls + (size_t)t * g->control_dim + c->uoff[b], sizeof(double) * o->control_dim);
    o->best_cost = mas_ocp_objective(o, o->best_states, o->best_controls);
  }
  const double total = g->best_cost;
  mas_ilqr_free(&s); mas_global_ocp_free(g);
  return total;
}

#include "mas_oracle_dense_impl.h"   /* the same solve with exact re-use + OpenMP: mas_run_centralized_dense_structured */

/* sum over blocks of the stacked objective, in the stacked problem's order:
 * for each t the block costs are added in block order, then the stage sums in t order
 * (global_stage inside compute_trajectory_cost). */
static double blocked_objective(mas_ocp** ocps, int n, double** X, double** U) {
  const int T = ocps[0]->horizon_steps; double cost = 0.0;
  for (int t = 0; t < T; ++t) {
    double st = 0.0;
    for (int b = 0; b < n; ++b) st += ocps[b]->stage_cost(ocps[b], X[b] + (size_t)t * ocps[b]->state_dim, U[b] + (size_t)t * ocps[b]->control_dim, (size_t)t);
    cost += st;
  }
  double tc = 0.0;
  for (int b = 0; b < n; ++b) tc += ocps[b]->terminal_cost(ocps[b], X[b] + (size_t)T * ocps[b]->state_dim);
  return cost + tc;
}

/* Block-diagonal reading of the centralised solve (see header). */
double mas_run_centralized_blocked(mas_ocp** ocps, int n, int max_iterations, double tolerance, double max_ms,
                                   mas_strategy_stats* stats) {
  (void)max_ms;
  const int T = ocps[0]->horizon_steps;
  step_ws* w = (step_ws*)calloc((size_t)n, sizeof(step_ws));
  double **wbuf = (double**)calloc((size_t)n, sizeof(double*)), **vx = (double**)calloc((size_t)n, sizeof(double*)), **vxx = (double**)calloc((size_t)n, sizeof(double*));
  double **k = (double**)calloc((size_t)n, sizeof(double*)), **K = (double**)calloc((size_t)n, sizeof(double*));
  double **xt = (double**)calloc((size_t)n, sizeof(double*)), **ut = (double**)calloc((size_t)n, sizeof(double*));
  double **x = (double**)calloc((size_t)n, sizeof(double*)), **u = (double**)calloc((size_t)n, sizeof(double*));
  double **cur = (double**)calloc((size_t)n, sizeof(double*));
  for (int b = 0; b < n; ++b) {
    mas_ocp* o = ocps[b]; const int nx = o->state_dim, nu = o->control_dim;
    wbuf[b] = dalloc(step_ws_size(nx, nu)); step_ws_bind(&w[b], wbuf[b], nx, nu);
    vx[b] = dalloc((size_t)nx); vxx[b] = dalloc((size_t)nx * nx); k[b] = dalloc((size_t)T * nu); K[b] = dalloc((size_t)T * nu * nx);
    xt[b] = dalloc((size_t)nx * (T + 1)); ut[b] = dalloc((size_t)nu * T); cur[b] = dalloc((size_t)nu * nu);
    x[b] = o->best_states; u[b] = o->best_controls;
    memset(u[b], 0, sizeof(double) * nu * T);   /* the stacked OCP starts from zero controls (multi_agent_problem.hpp:124 -> ocp.hpp:106-109) */
    mas_integrate_horizon(o, o->initial_state, u[b], o->dt, x[b]);
  }
  double current_merit = blocked_objective(ocps, n, x, u);
  long iterations = 0, regs = 0, trials = 0; int exit_reason = MAS_EXIT_MAX_ITER; int hist[12] = {0};
  for (int iter = 0; iter < max_iterations; ++iter) {
    iterations = iter + 1;
    for (int b = 0; b < n; ++b) terminal_boundary(ocps[b], x[b] + (size_t)T * ocps[b]->state_dim, vx[b], vxx[b]);
    int fail_all = 0;
    for (int t = T - 1; t >= 0 && !fail_all; --t) {
      for (int b = 0; b < n; ++b) {
        mas_ocp* o = ocps[b];
        linearise_step(o, x[b] + (size_t)t * o->state_dim, u[b] + (size_t)t * o->control_dim, (size_t)t, &w[b]);
        assemble_q(&w[b], vx[b], vxx[b]);
        memcpy(cur[b], w[b].quu, sizeof(double) * o->control_dim * o->control_dim);
      }
      /* shared ladder: one LLT over the block-diagonal q_uu fails if any block fails; reg*I then
       * goes onto EVERY block (ilqr.hpp:180) */
      double reg = 1e-6; int rungs = 0;
      for (;;) {
        int any_fail = 0;
        for (int b = 0; b < n; ++b) {
          const int nu = ocps[b]->control_dim;
          memcpy(w[b].quu_reg, cur[b], sizeof(double) * nu * nu);
          if (llt_compute(w[b].quu_reg, nu)) any_fail = 1;
        }
        if (!any_fail) break;
        if (rungs >= MAS_REG_LADDER_MAX) { fail_all = 1; break; }
        for (int b = 0; b < n; ++b) { const int nu = ocps[b]->control_dim; for (int i = 0; i < nu; ++i) cur[b][(size_t)i * nu + i] += reg; }
        reg *= 10.0; ++rungs;
      }
      regs += rungs;
      if (fail_all) break;
      for (int b = 0; b < n; ++b)
        gains_and_value(&w[b], k[b] + (size_t)t * ocps[b]->control_dim, K[b] + (size_t)t * ocps[b]->control_dim * ocps[b]->state_dim, vx[b], vxx[b]);
    }
    if (fail_all) { exit_reason = MAS_EXIT_REG_LIMIT; break; }

    double alpha = 1.0, best_merit = current_merit; int accepted_j = 10, j = 0;
    while (alpha >= 1e-3) {
      for (int b = 0; b < n; ++b) {
        mas_ocp* o = ocps[b]; const int nx = o->state_dim, nu = o->control_dim; double dx[16];
        memcpy(xt[b], o->initial_state, sizeof(double) * nx);
        for (int t = 0; t < T; ++t) {
          for (int i = 0; i < nx; ++i) dx[i] = xt[b][(size_t)t * nx + i] - x[b][(size_t)t * nx + i];
          for (int i = 0; i < nu; ++i) {
            double kd = 0.0; for (int c = 0; c < nx; ++c) kd += K[b][(size_t)t * nu * nx + (size_t)c * nu + i] * dx[c];
            ut[b][(size_t)t * nu + i] = (u[b][(size_t)t * nu + i] + alpha * k[b][(size_t)t * nu + i]) + kd;
          }
          if (o->has_input_lower && o->has_input_upper) mas_clamp_controls_column(o, ut[b] + (size_t)t * nu);
          mas_integrate_rk4(o, xt[b] + (size_t)t * nx, ut[b] + (size_t)t * nu, o->dt, xt[b] + (size_t)(t + 1) * nx);
        }
      }
      ++trials;
      const double trial_merit = blocked_objective(ocps, n, xt, ut);
      if (trial_merit < best_merit) {
        best_merit = trial_merit; accepted_j = j;
        for (int b = 0; b < n; ++b) {
          memcpy(x[b], xt[b], sizeof(double) * ocps[b]->state_dim * (T + 1)); memcpy(u[b], ut[b], sizeof(double) * ocps[b]->control_dim * T);
        }
        break;
      }
      alpha *= 0.5; ++j;
    }
    hist[accepted_j]++;
    const double improvement = current_merit - best_merit;
    current_merit = best_merit;
    if (improvement < tolerance) { exit_reason = MAS_EXIT_CONVERGED; break; }
  }
  for (int b = 0; b < n; ++b) {
    ocps[b]->best_cost = mas_ocp_objective(ocps[b], x[b], u[b]);   /* centralized.hpp:32 */
    free(wbuf[b]); free(vx[b]); free(vxx[b]); free(k[b]); free(K[b]); free(xt[b]); free(ut[b]); free(cur[b]);
  }
  const double total = blocked_objective(ocps, n, x, u);              /* global.best_cost */
  if (stats) {
    stats->total_iterations += iterations; stats->converged_solves += (exit_reason == MAS_EXIT_CONVERGED);
    stats->reg_events += regs; stats->trials += trials; for (int i = 0; i < 12; ++i) stats->alpha_hist[i] += hist[i];
  }
  free(w); free(wbuf); free(vx); free(vxx); free(k); free(K); free(xt); free(ut); free(x); free(u); free(cur);
  return total;
}

/* ========================================================== model registry */

/* --- single track: examples/models/single_track_model.hpp:23-44 ---
 * params: [0] track_radius [1] target_velocity [2] w_track [3] w_speed [4] w_delta [5] w_acc [6] L */
static void st_dynamics(const mas_ocp* o, const double* x, const double* u, double* d) {
  const double psi = x[2], v = x[3], delta = u[0], a = u[1], L = o->params[6];
  d[0] = v * o_cos(psi); d[1] = v * o_sin(psi); d[2] = v * o_tan(delta) / L; d[3] = a;
}
/* examples/multi_agent_single_track.cpp:54-64 */
static double st_stage(const mas_ocp* o, const double* s, const double* c, size_t t) {
  (void)t;
  const double x = s[0], y = s[1], vx = s[3], delta = c[0], a_cmd = c[1];
  const double* p = o->params;
  const double dist = fabs(sqrt(x * x + y * y) - p[0]);
  const double se = vx - p[1];
  return p[2] * dist * dist + p[3] * se * se + p[4] * delta * delta + p[5] * a_cmd * a_cmd;
}
static double zero_terminal(const mas_ocp* o, const double* x) { (void)o; (void)x; return 0.0; }

/* --- identity LQR: examples/multi_agent_lqr.cpp:21-76 (A=B=Q=R=Qf=I, n_x = n_u) --- */
static void lqr_dynamics(const mas_ocp* o, const double* x, const double* u, double* d) {
  for (int i = 0; i < o->state_dim; ++i) d[i] = x[i] + u[i];         /* A*x + B*u with identities */
}
static double lqr_stage(const mas_ocp* o, const double* x, const double* u, size_t t) {
  (void)t; double a = 0.0, b = 0.0;
  for (int i = 0; i < o->state_dim; ++i) a += x[i] * x[i];
  for (int i = 0; i < o->control_dim; ++i) b += u[i] * u[i];
  return a + b;
}
static double lqr_terminal(const mas_ocp* o, const double* x) { double a = 0.0; for (int i = 0; i < o->state_dim; ++i) a += x[i] * x[i]; return a; }
static void lqr_A(const mas_ocp* o, const double* x, const double* u, double* A) { (void)x; (void)u; const int n = o->state_dim; for (int i = 0; i < n * n; ++i) A[i] = 0.0; for (int i = 0; i < n; ++i) A[(size_t)i * n + i] = 1.0; }
static void lqr_B(const mas_ocp* o, const double* x, const double* u, double* B) { (void)x; (void)u; const int n = o->state_dim, m = o->control_dim; for (int i = 0; i < n * m; ++i) B[i] = 0.0; for (int i = 0; i < (n < m ? n : m); ++i) B[(size_t)i * n + i] = 1.0; }
static void lqr_lx(const mas_ocp* o, const double* x, const double* u, size_t t, double* g) { (void)u; (void)t; for (int i = 0; i < o->state_dim; ++i) g[i] = 2.0 * x[i]; }
static void lqr_lu(const mas_ocp* o, const double* x, const double* u, size_t t, double* g) { (void)x; (void)t; for (int i = 0; i < o->control_dim; ++i) g[i] = 2.0 * u[i]; }
static void lqr_lxx(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) { (void)x; (void)u; (void)t; const int n = o->state_dim; for (int i = 0; i < n * n; ++i) H[i] = 0.0; for (int i = 0; i < n; ++i) H[(size_t)i * n + i] = 2.0; }
static void lqr_luu(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) { (void)x; (void)u; (void)t; const int n = o->control_dim; for (int i = 0; i < n * n; ++i) H[i] = 0.0; for (int i = 0; i < n; ++i) H[(size_t)i * n + i] = 2.0; }
static void lqr_lux(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) { (void)x; (void)u; (void)t; for (int i = 0; i < o->control_dim * o->state_dim; ++i) H[i] = 0.0; }
static void lqr_vx(const mas_ocp* o, const double* x, double* g) { for (int i = 0; i < o->state_dim; ++i) g[i] = 2.0 * x[i]; }
static void lqr_vxx(const mas_ocp* o, const double* x, double* H) { (void)x; const int n = o->state_dim; for (int i = 0; i < n * n; ++i) H[i] = 0.0; for (int i = 0; i < n; ++i) H[(size_t)i * n + i] = 2.0; }

/* --- pendulum: examples/models/pendulum_model.hpp:8-20, examples/pendulum_swing_up.cpp:29-117 ---
 * params: [0] horizon_d (the T inside the cost's  s = k/(T-1)) */
static void pend_dynamics(const mas_ocp* o, const double* x, const double* u, double* d) {
  (void)o; const double g = 9.81, l = 1.0, m = 1.0, b = 0.1;
  d[0] = x[1];
  d[1] = (g / l) * o_sin(x[0]) + u[0] / (m * l * l) - (b / (m * l * l)) * x[1];
}
static double pend_stage(const mas_ocp* o, const double* x, const double* u, size_t k) {
  const double g = 9.81, l = 1.0, m = 1.0, mgl = m * g * l, E_des = mgl;
  const double w_energy = 2.0, w_u = 0.05, w_shape = 2.0, w_omega = 0.05;
  const double horizon_d = o->params[0];
  const double theta = x[0], omega = x[1], torque = u[0];
  const double s = (double)k / (horizon_d - 1.0);
  const double late = s * s, early = 1.0 - late;
  const double w_energy_k = w_energy * (0.2 + 0.8 * early);
  const double w_shape_k = w_shape * (0.2 + 0.8 * late);
  const double w_omega_k = w_omega * (0.2 + 0.8 * late);
  const double Tk = 0.5 * m * l * l * omega * omega;
  const double V = mgl * o_cos(theta);
  const double E = Tk + V;
  const double energy_error = (E - E_des) / mgl;
  const double upright_error = 1.0 - o_cos(theta);
  return w_energy_k * energy_error * energy_error + w_shape_k * upright_error + w_omega_k * omega * omega + w_u * torque * torque;
}
static double pend_terminal(const mas_ocp* o, const double* x) {
  (void)o; const double wT_pos = 500.0, wT_vel = 100.0;
  const double upright_error = 1.0 - o_cos(x[0]);
  return wT_pos * upright_error + wT_vel * x[1] * x[1];
}

/* --- single-track lane following: examples/single_track_ocp.cpp:14-116 ---
 * analytic A,B (models/single_track_model.hpp:49-82), analytic l_x,l_u,l_xx,l_uu; the cross
 * term and the terminal derivatives stay FD (never set by the example).
 * params: [0] desired_velocity [1] w_lane [2] w_speed [3] w_delta [4] w_acc [6] L */
static double lane_stage(const mas_ocp* o, const double* s, const double* c, size_t t) {
  (void)t; const double* p = o->params;
  const double y = s[1], vx = s[3], delta = c[0], a_cmd = c[1];
  const double lane_error = y, speed_error = vx - p[0];
  return p[1] * (lane_error * lane_error) + p[2] * (speed_error * speed_error) + p[3] * (delta * delta) + p[4] * (a_cmd * a_cmd);
}
static void lane_A(const mas_ocp* o, const double* x, const double* u, double* A) {
  const double psi = x[2], v = x[3], delta = u[0], L = o->params[6];
  for (int i = 0; i < 16; ++i) A[i] = 0.0;
  A[2 * 4 + 0] = -v * o_sin(psi); A[3 * 4 + 0] = o_cos(psi);
  A[2 * 4 + 1] = v * o_cos(psi);  A[3 * 4 + 1] = o_sin(psi);
  A[3 * 4 + 2] = o_tan(delta) / L;
}
static void lane_B(const mas_ocp* o, const double* x, const double* u, double* B) {
  const double v = x[3], delta = u[0], L = o->params[6];
  for (int i = 0; i < 8; ++i) B[i] = 0.0;
  B[0 * 4 + 2] = v / (L * o_cos(delta) * o_cos(delta));
  B[1 * 4 + 3] = 1.0;
}
static void lane_lx(const mas_ocp* o, const double* x, const double* u, size_t t, double* g) { (void)u; (void)t; const double* p = o->params; g[0] = 0.0; g[1] = 2.0 * p[1] * x[1]; g[2] = 0.0; g[3] = 2.0 * p[2] * (x[3] - p[0]); }
static void lane_lu(const mas_ocp* o, const double* x, const double* u, size_t t, double* g) { (void)x; (void)t; const double* p = o->params; g[0] = 2.0 * p[3] * u[0]; g[1] = 2.0 * p[4] * u[1]; }
static void lane_lxx(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) { (void)x; (void)u; (void)t; const double* p = o->params; for (int i = 0; i < 16; ++i) H[i] = 0.0; H[1 * 4 + 1] = 2.0 * p[1]; H[3 * 4 + 3] = 2.0 * p[2]; }
static void lane_luu(const mas_ocp* o, const double* x, const double* u, size_t t, double* H) { (void)x; (void)u; (void)t; const double* p = o->params; H[0] = 2.0 * p[3]; H[1] = 0.0; H[2] = 0.0; H[3] = 2.0 * p[4]; }

/* --- bicycle6: THIS REPO'S OWN model (BASELINE.json config 5; no reference counterpart) ---
 * state [X, Y, psi, v, delta, a], control [delta_rate, jerk]:
 *   X' = v cos psi, Y' = v sin psi, psi' = v tan(delta)/L, v' = a, delta' = u0, a' = u1
 * cost = w_track*(|r - R|)^2 + w_speed*(v - vt)^2 + w_delta*delta^2 + w_acc*a^2 + w_rate*(u0^2 + u1^2)
 * params: [0] R [1] vt [2] w_track [3] w_speed [4] w_delta [5] w_acc [6] L [7] w_rate */
static void b6_dynamics(const mas_ocp* o, const double* x, const double* u, double* d) {
  const double psi = x[2], v = x[3], delta = x[4], a = x[5], L = o->params[6];
  d[0] = v * o_cos(psi); d[1] = v * o_sin(psi); d[2] = v * o_tan(delta) / L; d[3] = a; d[4] = u[0]; d[5] = u[1];
}
static double b6_stage(const mas_ocp* o, const double* s, const double* c, size_t t) {
  (void)t; const double* p = o->params;
  const double x = s[0], y = s[1], vx = s[3], delta = s[4], a = s[5];
  const double dist = fabs(sqrt(x * x + y * y) - p[0]);
  const double se = vx - p[1];
  return p[2] * dist * dist + p[3] * se * se + p[4] * delta * delta + p[5] * a * a + p[7] * c[0] * c[0] + p[7] * c[1] * c[1];
}

/* --- single track with function constraints: THIS REPO'S OWN test workload for the augmented-Lagrangian
 * branch (no shipped reference example sets a constraint function).  Dynamics, cost and bounds of the
 * single-track workload plus
 *   equality    h(x,u) = [ delta - p[9] * a ]                (steering coupled to acceleration)
 *   inequality  g(x,u) = [ v - p[8] ,  -v ]                  (speed limit, no reversing)
 * params: single-track's [0..6], [8] v_max, [9] coupling */
static void stl_eq(const mas_ocp* o, const double* x, const double* u, double* out) { (void)x; out[0] = u[0] - o->params[9] * u[1]; }
static void stl_ineq(const mas_ocp* o, const double* x, const double* u, double* out) { (void)u; out[0] = x[3] - o->params[8]; out[1] = -x[3]; }

/* ---- external models (ids >= 100): the three callbacks of an OCP (types.hpp:21-50) as plain C functions, e.g. a user-defined workload of the
 * product (mas_hip_register_model) compiled for the CPU by oracle.register_model().  Every derivative hook stays empty => finite differences. */
#define MAS_EXTERNAL_FIRST 100
#define MAS_EXTERNAL_MAX 64
typedef struct { int nx, nu, has_bounds; double lb[8], ub[8]; mas_ext_dynamics_fn dyn; mas_ext_stage_fn stage; mas_ext_terminal_fn term; mas_ext_hooks hooks;
                 int neq, nin; mas_ext_xu_fn eq, in; } ext_model;
static ext_model g_ext[MAS_EXTERNAL_MAX];
static int g_ext_count = 0;
int mas_oracle_register_external_model(int nx, int nu, mas_ext_dynamics_fn dyn, mas_ext_stage_fn stage, mas_ext_terminal_fn term,
                                       const double* lb, const double* ub) {
  if (g_ext_count >= MAS_EXTERNAL_MAX || nx < 1 || nx > 16 || nu < 1 || nu > 8 || !dyn || !stage || !term) return -1;
  ext_model* e = &g_ext[g_ext_count];
  memset(&e->hooks, 0, sizeof(e->hooks));
  e->neq = e->nin = 0; e->eq = e->in = NULL;
  e->nx = nx; e->nu = nu; e->dyn = dyn; e->stage = stage; e->term = term; e->has_bounds = (lb && ub);
  if (e->has_bounds) for (int i = 0; i < nu; ++i) { e->lb[i] = lb[i]; e->ub[i] = ub[i]; }
  return MAS_EXTERNAL_FIRST + g_ext_count++;
}
static const ext_model* ext_of(int model_id) { const int k = model_id - MAS_EXTERNAL_FIRST; return (k >= 0 && k < g_ext_count) ? &g_ext[k] : NULL; }
static void ext_dynamics(const mas_ocp* o, const double* x, const double* u, double* d) { ext_of(o->model_id)->dyn(x, u, d, o->params); }
static double ext_stage(const mas_ocp* o, const double* x, const double* u, size_t t) { return ext_of(o->model_id)->stage(x, u, (int)t, o->params); }
static double ext_terminal(const mas_ocp* o, const double* x) { return ext_of(o->model_id)->term(x, o->params); }
/* the derivative hooks an external model provides (types.hpp:30-50); the others stay empty => finite differences (ocp.hpp:117-135) */
int mas_oracle_external_set_hooks(int model_id, const mas_ext_hooks* hooks) {
  const int k = model_id - MAS_EXTERNAL_FIRST;
  if (k < 0 || k >= g_ext_count || !hooks) return -1;
  g_ext[k].hooks = *hooks;
  return 0;
}
/* function constraints h(x,u) = 0 (neq rows), g(x,u) <= 0 (nin rows) of an external model (ocp.hpp:61-68); their Jacobians stay FD (ocp.hpp:137-171) */
int mas_oracle_external_set_constraints(int model_id, int neq, mas_ext_xu_fn equality, int nin, mas_ext_xu_fn inequality) {
  const int k = model_id - MAS_EXTERNAL_FIRST;
  if (k < 0 || k >= g_ext_count || neq < 0 || nin < 0 || (neq > 0 && !equality) || (nin > 0 && !inequality)) return -1;
  g_ext[k].neq = neq; g_ext[k].eq = equality; g_ext[k].nin = nin; g_ext[k].in = inequality;
  return 0;
}
static void ext_eq(const mas_ocp* o, const double* x, const double* u, double* out) { ext_of(o->model_id)->eq(x, u, out, o->params); }
static void ext_in(const mas_ocp* o, const double* x, const double* u, double* out) { ext_of(o->model_id)->in(x, u, out, o->params); }
static void ext_A(const mas_ocp* o, const double* x, const double* u, double* out) { ext_of(o->model_id)->hooks.A(x, u, out, o->params); }
static void ext_B(const mas_ocp* o, const double* x, const double* u, double* out) { ext_of(o->model_id)->hooks.B(x, u, out, o->params); }
static void ext_lx(const mas_ocp* o, const double* x, const double* u, size_t t, double* out) { ext_of(o->model_id)->hooks.lx(x, u, (int)t, out, o->params); }
static void ext_lu(const mas_ocp* o, const double* x, const double* u, size_t t, double* out) { ext_of(o->model_id)->hooks.lu(x, u, (int)t, out, o->params); }
static void ext_lxx(const mas_ocp* o, const double* x, const double* u, size_t t, double* out) { ext_of(o->model_id)->hooks.lxx(x, u, (int)t, out, o->params); }
static void ext_luu(const mas_ocp* o, const double* x, const double* u, size_t t, double* out) { ext_of(o->model_id)->hooks.luu(x, u, (int)t, out, o->params); }
static void ext_lux(const mas_ocp* o, const double* x, const double* u, size_t t, double* out) { ext_of(o->model_id)->hooks.lux(x, u, (int)t, out, o->params); }
static void ext_tg(const mas_ocp* o, const double* x, double* out) { ext_of(o->model_id)->hooks.term_grad(x, out, o->params); }
static void ext_th(const mas_ocp* o, const double* x, double* out) { ext_of(o->model_id)->hooks.term_hess(x, out, o->params); }

int mas_model_dims(int model_id, int* nx, int* nu, int* n_params) {
  int a = 0, b = 0, c = 0;
  if (ext_of(model_id)) { if (nx) *nx = ext_of(model_id)->nx; if (nu) *nu = ext_of(model_id)->nu; if (n_params) *n_params = 16; return 0; }
  switch (model_id) {
    case MAS_MODEL_SINGLE_TRACK: a = 4; b = 2; c = 7; break;
    case MAS_MODEL_LQR_IDENTITY: a = 4; b = 4; c = 0; break;
    case MAS_MODEL_PENDULUM: a = 2; b = 1; c = 1; break;
    case MAS_MODEL_SINGLE_TRACK_LANE: a = 4; b = 2; c = 7; break;
    case MAS_MODEL_BICYCLE6: a = 6; b = 2; c = 8; break;
    case MAS_MODEL_SINGLE_TRACK_LIMITED: a = 4; b = 2; c = 10; break;
    default: return -1;
  }
  if (nx) *nx = a;
  if (nu) *nu = b;
  if (n_params) *n_params = c;
  return 0;
}

void mas_model_default_params(int model_id, int T, double* p) {
  for (int i = 0; i < 16; ++i) p[i] = 0.0;
  switch (model_id) {
    case MAS_MODEL_SINGLE_TRACK:      /* multi_agent_single_track.cpp:49-52,109-110 */
      p[0] = 20.0; p[1] = 5.0; p[2] = 1.0; p[3] = 1.0; p[4] = 0.001; p[5] = 0.001; p[6] = 2.5; break;
    case MAS_MODEL_PENDULUM: p[0] = (double)T; break;   /* pendulum_swing_up.cpp:62 */
    case MAS_MODEL_SINGLE_TRACK_LANE: /* single_track_ocp.cpp:34-40 */
      p[0] = 1.0; p[1] = 10.0; p[2] = 1.0; p[3] = 0.1; p[4] = 0.1; p[6] = 2.5; break;
    case MAS_MODEL_BICYCLE6:
      p[0] = 20.0; p[1] = 5.0; p[2] = 1.0; p[3] = 1.0; p[4] = 0.001; p[5] = 0.001; p[6] = 2.5; p[7] = 0.001; break;
    case MAS_MODEL_SINGLE_TRACK_LIMITED:
      p[0] = 20.0; p[1] = 5.0; p[2] = 1.0; p[3] = 1.0; p[4] = 0.001; p[5] = 0.001; p[6] = 2.5; p[8] = 4.5; p[9] = 0.1; break;
    default: break;
  }
}

mas_ocp* mas_model_make_ocp(int model_id, int T, double dt, const double* x0, const double* U0, const double* params, size_t id) {
  int nx, nu, np;
  if (mas_model_dims(model_id, &nx, &nu, &np)) return NULL;
  mas_ocp* o = mas_ocp_new(nx, nu, T, dt);
  o->model_id = model_id; o->id = id;
  if (params) memcpy(o->params, params, sizeof(double) * 16); else mas_model_default_params(model_id, T, o->params);
  memcpy(o->initial_state, x0, sizeof(double) * nx);
  o->terminal_cost = zero_terminal;
  if (ext_of(model_id)) {
    const ext_model* e = ext_of(model_id);
    o->dynamics = ext_dynamics; o->stage_cost = ext_stage; o->terminal_cost = ext_terminal;
    if (e->hooks.A) o->dynamics_state_jacobian = ext_A;
    if (e->hooks.B) o->dynamics_control_jacobian = ext_B;
    if (e->hooks.lx) o->cost_state_gradient = ext_lx;
    if (e->hooks.lu) o->cost_control_gradient = ext_lu;
    if (e->hooks.lxx) o->cost_state_hessian = ext_lxx;
    if (e->hooks.luu) o->cost_control_hessian = ext_luu;
    if (e->hooks.lux) o->cost_cross_term = ext_lux;
    if (e->neq > 0) { o->equality_constraints = ext_eq; o->eq_dim = e->neq; }
    if (e->nin > 0) { o->inequality_constraints = ext_in; o->ineq_dim = e->nin; }
    if (e->hooks.term_grad && e->hooks.term_hess) { o->terminal_cost_gradient = ext_tg; o->terminal_cost_hessian = ext_th; }
    if (e->has_bounds) mas_ocp_set_input_bounds(o, e->lb, e->ub);
  }
  switch (model_id) {
    case MAS_MODEL_SINGLE_TRACK: {
      o->dynamics = st_dynamics; o->stage_cost = st_stage;
      const double lb[2] = {-0.5, -0.5}, ub[2] = {0.5, 0.5};   /* multi_agent_single_track.cpp:66-67 */
      mas_ocp_set_input_bounds(o, lb, ub);
    } break;
    case MAS_MODEL_LQR_IDENTITY:
      o->dynamics = lqr_dynamics; o->stage_cost = lqr_stage; o->terminal_cost = lqr_terminal;
      o->dynamics_state_jacobian = lqr_A; o->dynamics_control_jacobian = lqr_B;
      o->cost_state_gradient = lqr_lx; o->cost_control_gradient = lqr_lu;
      o->cost_state_hessian = lqr_lxx; o->cost_control_hessian = lqr_luu; o->cost_cross_term = lqr_lux;
      o->terminal_cost_gradient = lqr_vx; o->terminal_cost_hessian = lqr_vxx;
      break;
    case MAS_MODEL_PENDULUM: {
      o->dynamics = pend_dynamics; o->stage_cost = pend_stage; o->terminal_cost = pend_terminal;
      const double lb[1] = {-5.0}, ub[1] = {5.0};               /* pendulum_swing_up.cpp:100-103 */
      mas_ocp_set_input_bounds(o, lb, ub);
    } break;
    case MAS_MODEL_SINGLE_TRACK_LANE: {
      o->dynamics = st_dynamics; o->stage_cost = lane_stage;
      o->dynamics_state_jacobian = lane_A; o->dynamics_control_jacobian = lane_B;
      o->cost_state_gradient = lane_lx; o->cost_control_gradient = lane_lu;
      o->cost_state_hessian = lane_lxx; o->cost_control_hessian = lane_luu;
      const double lb[2] = {-0.7, -1.0}, ub[2] = {0.7, 1.0};    /* single_track_ocp.cpp:105-109 */
      mas_ocp_set_input_bounds(o, lb, ub);
    } break;
    case MAS_MODEL_BICYCLE6: {
      o->dynamics = b6_dynamics; o->stage_cost = b6_stage;
      const double lb[2] = {-0.5, -0.5}, ub[2] = {0.5, 0.5};
      mas_ocp_set_input_bounds(o, lb, ub);
    } break;
    case MAS_MODEL_SINGLE_TRACK_LIMITED: {
      o->dynamics = st_dynamics; o->stage_cost = st_stage;
      o->equality_constraints = stl_eq; o->eq_dim = 1;
      o->inequality_constraints = stl_ineq; o->ineq_dim = 2;
      const double lb[2] = {-0.5, -0.5}, ub[2] = {0.5, 0.5};
      mas_ocp_set_input_bounds(o, lb, ub);
    } break;
  }
  if (U0) mas_ocp_set_initial_controls(o, U0);
  mas_ocp_initialize_problem(o);
  return o;
}

int mas_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

double mas_oracle_solve_batch(int model_id, const double* params, int N, int T, double dt,
                              const double* x0, double* U, double* X, double* cost, int* iters,
                              int strategy, int max_outer, int max_iterations, double tolerance, double max_ms,
                              int threads, mas_strategy_stats* stats) {
  int nx, nu, np;
  if (mas_model_dims(model_id, &nx, &nu, &np)) return NAN;
#ifdef _OPENMP
  const int prev_threads = omp_get_max_threads();
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
  if (stats) memset(stats, 0, sizeof(*stats));
  mas_ocp** ocps = (mas_ocp**)calloc((size_t)N, sizeof(mas_ocp*));
  const size_t nX = (size_t)nx * (T + 1), nU = (size_t)nu * T;
  for (int i = 0; i < N; ++i) ocps[i] = mas_model_make_ocp(model_id, T, dt, x0 + (size_t)i * nx, U + (size_t)i * nU, params, (size_t)i);
  double total = NAN;
  switch (strategy) {
    case 0: total = mas_run_sequential(ocps, N, max_outer, max_iterations, tolerance, max_ms, stats, iters); break;
    case 1: total = mas_run_trust_region(ocps, N, max_outer, max_iterations, tolerance, max_ms, stats, g_radii_out, iters); break;
    case 2: total = mas_run_line_search(ocps, N, max_outer, max_iterations, tolerance, max_ms, stats, iters); break;
    case 3: { mas_strategy_stats st; memset(&st, 0, sizeof(st));
              total = mas_run_centralized_dense(ocps, N, max_iterations, tolerance, max_ms, &st);
              if (iters) for (int i = 0; i < N; ++i) iters[i] = (int)st.total_iterations;
              if (stats) *stats = st; } break;
    case 4: { mas_strategy_stats st; memset(&st, 0, sizeof(st));
              total = mas_run_centralized_blocked(ocps, N, max_iterations, tolerance, max_ms, &st);
              if (iters) for (int i = 0; i < N; ++i) iters[i] = (int)st.total_iterations;
              if (stats) *stats = st; } break;
    case 5: total = mas_run_sequential(ocps, N, 1, max_iterations, tolerance, max_ms, stats, iters); break;
    case 6: { mas_strategy_stats st; memset(&st, 0, sizeof(st));
              total = mas_run_centralized_dense_structured(ocps, N, max_iterations, tolerance, max_ms, &st);
              if (iters) for (int i = 0; i < N; ++i) iters[i] = (int)st.total_iterations;
              if (stats) *stats = st; } break;
    default: break;
  }
  for (int i = 0; i < N; ++i) {
    memcpy(U + (size_t)i * nU, ocps[i]->best_controls, sizeof(double) * nU);
    if (X) memcpy(X + (size_t)i * nX, ocps[i]->best_states, sizeof(double) * nX);
    if (cost) cost[i] = ocps[i]->best_cost;
    mas_ocp_free(ocps[i]);
  }
  free(ocps);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(prev_threads);
#endif
  return total;
}
