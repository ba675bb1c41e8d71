/* mas_oracle_dense_impl.h -- TEST INFRASTRUCTURE (part of the CPU oracle; included once, at the end of mas_oracle.c).
 *
 * The centralised strategy in the reference's LITERAL reading (centralized.hpp:18-38 on build_global_ocp,
 * multi_agent_problem.hpp:52-127) at sizes the plain restatement (mas_run_centralized_dense: mas_ilqr_solve on the stacked OCP
 * through the generic FD hooks) cannot reach: 4 (N nx)^2 evaluations of the N-agent cost per time step.
 *
 * mas_run_centralized_dense_structured computes THE SAME NUMBERS with two kinds of exact re-use, nothing else:
 *   1. the stacked stage cost is  c = 0; c += cost_b  in block order (multi_agent_problem.hpp:94-105).  A finite-difference
 *      evaluation perturbs one or two coordinates, i.e. the costs of at most two agents; every other agent's cost is the number the
 *      unperturbed evaluation produced (x + 0.0 on untouched coordinates is the same value).  So the additions up to the first
 *      perturbed agent are the stored prefix pre[b] = c_0 + .. + c_{b-1} of that very chain, and from there the chain is redone with
 *      the perturbed agents' costs substituted: the same additions on the same operands in the same order;
 *   2. A and B of the stacked problem are block-diagonal with exact +0.0 outside the blocks ((f - f) / (2 eps)), so in a product sum
 *      over k the terms outside the block are (+0.0) * v = +-0.0 and leave a finite sum's VALUE unchanged (at most the sign of an
 *      all-zero sum differs, which no operation of the path tells apart -- DESIGN.md section 2.1); they are skipped.  Precondition:
 *      finite V (0 * inf would be NaN in the literal product).
 * Dense parts (LLT + ladder, LLT::solve(I), gains, value update, feedback law) are the statements of mas_oracle.c with OpenMP over
 * independent OUTPUT ENTRIES; every entry keeps its own sum over k from 0.0 in index order.
 * tests/test_centralized_readings.py holds this against mas_run_centralized_dense (np.array_equal) at the sizes both can run. */

/* OpenMP only where a loop is worth a parallel region (stacked state dimension >= 64), and never more than 16 threads: a solve enters
 * tens of thousands of regions, and a box that shows more cores than it grants turns every region's barrier into a time slice */
static int g_dense_par = 0;

typedef struct {
  int N, T, nx, nu, n, m;
  mas_ocp** o;
  double *c, *pre;              /* [N] base cost of every agent at the current (t) point; [N+1] prefix sums of the chain */
  double *cx5, *cu5, *cx6, *cu6;/* [N][nx|nu][2] the agent's cost with one coordinate moved by +-1e-5 / +-1e-6 (index 0: +, 1: -) */
} dstack;

/* the chain  c = 0; c += cost_b (b = 0..N-1)  with agent b1's cost replaced by v1 and, if b2 >= 0 (b2 > b1), agent b2's by v2 */
static inline double dchain(const dstack* s, int b1, double v1, int b2, double v2) {
  const int N = s->N; const double* c = s->c;
  double a = s->pre[b1];
  a += v1;
  if (b2 < 0) { for (int b = b1 + 1; b < N; ++b) a += c[b]; return a; }
  for (int b = b1 + 1; b < b2; ++b) a += c[b];
  a += v2;
  for (int b = b2 + 1; b < N; ++b) a += c[b];
  return a;
}
static inline double dsafe(double v) { return isfinite(v) ? v : 0.0; }       /* safe_eval, finite_differences.hpp:95-107 */
/* two substitutions given in any order */
static inline double dchain2(const dstack* s, int ba, double va, int bb, double vb) {
  return ba < bb ? dchain(s, ba, va, bb, vb) : dchain(s, bb, vb, ba, va);
}

/* cost tables of one time step (terminal: t == T, controls unused) */
static void dtables(dstack* s, const double* xg, const double* ug, int t) {
  const int N = s->N, nx = s->nx, nu = s->nu, term = t == s->T;
#pragma omp parallel for schedule(static) if (g_dense_par)
  for (int b = 0; b < N; ++b) {
    const mas_ocp* o = s->o[b];
    const double* x = xg + (size_t)b * nx; const double* u = term ? NULL : ug + (size_t)b * nu;
    double xp[16], up[16];
    s->c[b] = term ? o->terminal_cost(o, x) : o->stage_cost(o, x, u, (size_t)t);
    for (int k = 0; k < nx; ++k)
      for (int sg = 0; sg < 2; ++sg) {
        perturb1(x, nx, k, sg ? -1e-5 : +1e-5, xp);
        s->cx5[((size_t)b * nx + k) * 2 + sg] = term ? o->terminal_cost(o, xp) : o->stage_cost(o, xp, u, (size_t)t);
        perturb1(x, nx, k, sg ? -1e-6 : +1e-6, xp);
        s->cx6[((size_t)b * nx + k) * 2 + sg] = term ? o->terminal_cost(o, xp) : o->stage_cost(o, xp, u, (size_t)t);
      }
    if (!term)
      for (int k = 0; k < nu; ++k)
        for (int sg = 0; sg < 2; ++sg) {
          perturb1(u, nu, k, sg ? -1e-5 : +1e-5, up);
          s->cu5[((size_t)b * nu + k) * 2 + sg] = o->stage_cost(o, x, up, (size_t)t);
          perturb1(u, nu, k, sg ? -1e-6 : +1e-6, up);
          s->cu6[((size_t)b * nu + k) * 2 + sg] = o->stage_cost(o, x, up, (size_t)t);
        }
  }
  s->pre[0] = 0.0;
  for (int b = 0; b < N; ++b) s->pre[b + 1] = s->pre[b] + s->c[b];
}

/* Hessian of the summed cost in the stacked STATE (which = 0) or CONTROL (which = 1) vector, eps = 1e-5 through safe_eval
 * (finite_differences.hpp:138-210, 226-261): H is dim x dim column-major, dim = N * per */
static void dhessian(const dstack* s, const double* xg, const double* ug, int t, int which, double* H) {
  const int N = s->N, nx = s->nx, nu = s->nu, per = which ? nu : nx, dim = N * per, term = t == s->T;
  const double eps = 1e-5;
  const double* tab = which ? s->cu5 : s->cx5;
  const double f0 = dsafe(s->pre[N]);
#pragma omp parallel for schedule(dynamic, 1) if (g_dense_par)
  for (int bi = 0; bi < N; ++bi) {
    const mas_ocp* o = s->o[bi];
    const double* x = xg + (size_t)bi * nx; const double* u = term ? NULL : ug + (size_t)bi * nu;
    double vp[16];
    for (int ki = 0; ki < per; ++ki) {
      const int i = bi * per + ki;
      {                                                                  /* diagonal entry */
        const double fp = dsafe(dchain(s, bi, tab[((size_t)bi * per + ki) * 2 + 0], -1, 0.0));
        const double fm = dsafe(dchain(s, bi, tab[((size_t)bi * per + ki) * 2 + 1], -1, 0.0));
        H[(size_t)i * dim + i] = (fp - 2 * f0 + fm) / (eps * eps);
      }
      for (int kj = 0; kj < per; ++kj) {                                 /* the agent's own block: both coordinates move inside one agent */
        if (kj == ki) continue;
        const int j = bi * per + kj;
        double f[4];
        for (int q = 0; q < 4; ++q) {
          const double si = (q & 2) ? -eps : +eps, sj = (q & 1) ? -eps : +eps;     /* pp pm mp mm */
          perturb2(which ? u : x, per, ki, si, kj, sj, vp);
          const double cb = term ? o->terminal_cost(o, vp) : (which ? o->stage_cost(o, x, vp, (size_t)t) : o->stage_cost(o, vp, u, (size_t)t));
          f[q] = dsafe(dchain(s, bi, cb, -1, 0.0));
        }
        H[(size_t)j * dim + i] = (f[0] - f[1] - f[2] + f[3]) / (4 * eps * eps);
      }
      for (int bj = 0; bj < N; ++bj) {                                   /* other agents' blocks: one table value each */
        if (bj == bi) continue;
        for (int kj = 0; kj < per; ++kj) {
          const int j = bj * per + kj;
          double f[4];
          for (int q = 0; q < 4; ++q)
            f[q] = dsafe(dchain2(s, bi, tab[((size_t)bi * per + ki) * 2 + ((q >> 1) & 1)], bj, tab[((size_t)bj * per + kj) * 2 + (q & 1)]));
          H[(size_t)j * dim + i] = (f[0] - f[1] - f[2] + f[3]) / (4 * eps * eps);
        }
      }
    }
  }
}

/* cross term m x n of the summed cost, eps = 1e-6, f_pm = f(x - dx, u + du), f_mp = f(x + dx, u - du) (finite_differences.hpp:263-287) */
static void dcross(const dstack* s, const double* xg, const double* ug, int t, double* H) {
  const int N = s->N, nx = s->nx, nu = s->nu, n = s->n, m = s->m;
  const double eps = 1e-6;
  (void)n;
#pragma omp parallel for schedule(dynamic, 1) if (g_dense_par)
  for (int bi = 0; bi < N; ++bi) {                                       /* control coordinate i = (bi, ki) */
    const mas_ocp* o = s->o[bi];
    double xp[16], up[16];
    for (int ki = 0; ki < nu; ++ki) {
      const int i = bi * nu + ki;
      for (int bj = 0; bj < N; ++bj)
        for (int kj = 0; kj < nx; ++kj) {                                /* state coordinate j = (bj, kj) */
          const int j = bj * nx + kj;
          double f[4];                                                   /* pp pm mp mm: first sign = x, second = u, then fpm = (x-, u+), fmp = (x+, u-) */
          for (int q = 0; q < 4; ++q) {
            const int sx = (q == 1 || q == 3) ? 1 : 0, su = (q >= 2) ? 1 : 0;        /* fpp (x+,u+)  fpm (x-,u+)  fmp (x+,u-)  fmm (x-,u-) */
            if (bi == bj) {
              perturb1(xg + (size_t)bi * nx, nx, kj, sx ? -eps : +eps, xp);
              perturb1(ug + (size_t)bi * nu, nu, ki, su ? -eps : +eps, up);
              f[q] = dsafe(dchain(s, bi, o->stage_cost(o, xp, up, (size_t)t), -1, 0.0));
            } else {
              f[q] = dsafe(dchain2(s, bj, s->cx6[((size_t)bj * nx + kj) * 2 + sx], bi, s->cu6[((size_t)bi * nu + ki) * 2 + su]));
            }
          }
          H[(size_t)j * m + i] = (f[0] - f[1] - f[2] + f[3]) / (4 * eps * eps);
        }
    }
  }
}

/* gradient of the summed cost, eps = 1e-6, no safe_eval (finite_differences.hpp:110-136, 212-224) */
static void dgradient(const dstack* s, int which, double* g) {
  const int N = s->N, per = which ? s->nu : s->nx;
  const double eps = 1e-6; const double* tab = which ? s->cu6 : s->cx6;
#pragma omp parallel for schedule(static) if (g_dense_par)
  for (int b = 0; b < N; ++b)
    for (int k = 0; k < per; ++k) {
      const double fp = dchain(s, b, tab[((size_t)b * per + k) * 2 + 0], -1, 0.0), fm = dchain(s, b, tab[((size_t)b * per + k) * 2 + 1], -1, 0.0);
      g[(size_t)b * per + k] = (fp - fm) / (2 * eps);
    }
}

/* ---- dense helpers with OpenMP over output entries; every entry: s = 0; s += a * b over k in index order ---- */
static void par_AtB(const double* A, int p, int m, const double* B, int n, double* C) {     /* C(m x n) = A^T B, A stored p x m */
#pragma omp parallel for schedule(static) collapse(2) if (g_dense_par)
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      const double* a = A + (size_t)i * p; const double* b = B + (size_t)j * p;
      double s = 0.0;
      for (int k = 0; k < p; ++k) s += a[k] * b[k];
      C[(size_t)j * m + i] = s;
    }
}
static void par_AB(const double* A, int m, int p, const double* B, int n, double* C) {      /* C(m x n) = A B, A stored m x p */
#pragma omp parallel for schedule(static) if (g_dense_par)
  for (int j = 0; j < n; ++j) {
    double* c = C + (size_t)j * m; const double* b = B + (size_t)j * p;
    for (int i = 0; i < m; ++i) c[i] = 0.0;
    for (int k = 0; k < p; ++k) { const double bk = b[k]; const double* a = A + (size_t)k * m; for (int i = 0; i < m; ++i) c[i] += a[i] * bk; }
  }
}
/* llt_compute with the rows of a column shared out (same per-entry sums: j ascending) */
static int par_llt_compute(double* M, int n, double* acc) {
  for (int k = 0; k < n; ++k) {
    double x = M[(size_t)k * n + k];
    if (k > 0) { double s = 0.0; for (int j = 0; j < k; ++j) s += M[(size_t)j * n + k] * M[(size_t)j * n + k]; x = M[(size_t)k * n + k] - s; }
    if (x <= 0.0) return 1;
    x = sqrt(x);
    M[(size_t)k * n + k] = x;
    const int rows = n - k - 1;
    if (rows <= 0) continue;
    double* col = M + (size_t)k * n;
#pragma omp parallel if (g_dense_par && (long)rows * k > 20000)
    {
#ifdef _OPENMP
      const int nt = omp_get_num_threads(), id = omp_get_thread_num();
#else
      const int nt = 1, id = 0;
#endif
      const int lo = k + 1 + (int)((long)rows * id / nt), hi = k + 1 + (int)((long)rows * (id + 1) / nt);
      for (int r = lo; r < hi; ++r) acc[r] = col[r];
      for (int j = 0; j < k; ++j) { const double lkj = -M[(size_t)j * n + k]; const double* cj = M + (size_t)j * n; for (int r = lo; r < hi; ++r) acc[r] += lkj * cj[r]; }
      for (int r = lo; r < hi; ++r) col[r] = acc[r] / x;
    }
  }
  return 0;
}
static void par_llt_solve_identity(const double* L, int n, double* Inv) {
#pragma omp parallel for schedule(dynamic, 4) if (g_dense_par)
  for (int c = 0; c < n; ++c) {
    double* o = Inv + (size_t)c * n;
    for (int i = 0; i < n; ++i) o[i] = (i == c) ? 1.0 : 0.0;
    for (int i = 0; i < n; ++i) {
      double a = 1.0 / L[(size_t)i * n + i];
      o[i] *= a;
      double b = o[i];
      const double* li = L + (size_t)i * n;
      for (int r = i + 1; r < n; ++r) o[r] -= b * li[r];
    }
    for (int i = n - 1; i >= 0; --i) {
      double a = 1.0 / L[(size_t)i * n + i];
      double b = 0.0;
      const double* li = L + (size_t)i * n;
      for (int r = i + 1; r < n; ++r) b += li[r] * o[r];
      o[i] = (o[i] - b) * a;
    }
  }
}
static void par_symmetrise(double* V, int n) {           /* symmetrise_inplace, default variant: two phases give the serial loop's numbers */
#pragma omp parallel for schedule(static) if (g_dense_par)
  for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) V[(size_t)j * n + i] = 0.5 * (V[(size_t)j * n + i] + V[(size_t)i * n + j]);
#pragma omp parallel for schedule(static) if (g_dense_par)
  for (int j = 0; j < n; ++j) for (int i = 0; i <= j; ++i) V[(size_t)j * n + i] = 0.5 * (V[(size_t)j * n + i] + V[(size_t)i * n + j]);
}

double mas_run_centralized_dense_structured(mas_ocp** ocps, int N, int max_iterations, double tolerance, double max_ms,
                                            mas_strategy_stats* stats) {
  const double start = now_ms();
  const int T = ocps[0]->horizon_steps, nx = ocps[0]->state_dim, nu = ocps[0]->control_dim, n = N * nx, m = N * nu;
  const double dt = ocps[0]->dt;
  const size_t n2 = (size_t)n * n, m2 = (size_t)m * m, mn = (size_t)m * n;
  dstack s; memset(&s, 0, sizeof(s));
  g_dense_par = n >= 64;
#ifdef _OPENMP
  const int threads_before = omp_get_max_threads();
  if (threads_before > 16) omp_set_num_threads(16);
#endif
  s.N = N; s.T = T; s.nx = nx; s.nu = nu; s.n = n; s.m = m; s.o = ocps;
  s.c = dalloc((size_t)N); s.pre = dalloc((size_t)N + 1);
  s.cx5 = dalloc((size_t)N * nx * 2); s.cx6 = dalloc((size_t)N * nx * 2); s.cu5 = dalloc((size_t)N * nu * 2); s.cu6 = dalloc((size_t)N * nu * 2);
  double *x = dalloc((size_t)(T + 1) * n), *u = dalloc((size_t)T * m), *xt = dalloc((size_t)(T + 1) * n), *ut = dalloc((size_t)T * m);
  double *Ab = dalloc((size_t)N * nx * nx), *Bb = dalloc((size_t)N * nx * nu);
  double *lx = dalloc(n), *lu = dalloc(m), *lxx = dalloc(n2), *luu = dalloc(m2), *lux = dalloc(mn);
  double *k = dalloc((size_t)T * m), *K = dalloc((size_t)T * mn);
  double *V = dalloc(n2), *vx = dalloc(n), *qx = dalloc(n), *qu = dalloc(m), *qxx = dalloc(n2), *qux = dalloc(mn), *quu = dalloc(m2);
  double *quu_reg = dalloc(m2), *quu_inv = dalloc(m2), *cur = dalloc(m2), *tmp_xx = dalloc(n2), *tmp_ux = dalloc(mn), *tmp_x = dalloc(n);
  double *tmp_uu = dalloc(m2), *KtQuu = dalloc(mn), *dx = dalloc(n), *acc = dalloc(m);
  const int all_bounds = ({ int ab = 1; for (int b = 0; b < N; ++b) ab &= ocps[b]->has_input_lower && ocps[b]->has_input_upper; ab; });

  /* build_global_ocp + initialize_problem: stacked x0, zero controls, rollout, objective (multi_agent_problem.hpp:52-127, ocp.hpp:102-183) */
#define D_ROLLOUT(Xs, Us) do { _Pragma("omp parallel for schedule(static) if (g_dense_par)") for (int b = 0; b < N; ++b) \
    for (int t = 0; t < T; ++t) mas_integrate_rk4(ocps[b], (Xs) + (size_t)t * n + (size_t)b * nx, (Us) + (size_t)t * m + (size_t)b * nu, dt, (Xs) + (size_t)(t + 1) * n + (size_t)b * nx); } while (0)
  for (int b = 0; b < N; ++b) memcpy(x + (size_t)b * nx, ocps[b]->initial_state, sizeof(double) * nx);
  memset(u, 0, sizeof(double) * T * m);
  D_ROLLOUT(x, u);
  /* compute_trajectory_cost on the stacked problem: stage sums (block order) added in t order, then the terminal sum */
  double* gs = dalloc((size_t)T + 1);
#define D_OBJECTIVE(Xs, Us, out) do { _Pragma("omp parallel for schedule(static) if (g_dense_par)") for (int t = 0; t <= T; ++t) { double cc = 0.0; \
      for (int b = 0; b < N; ++b) cc += t < T ? ocps[b]->stage_cost(ocps[b], (Xs) + (size_t)t * n + (size_t)b * nx, (Us) + (size_t)t * m + (size_t)b * nu, (size_t)t) \
                                              : ocps[b]->terminal_cost(ocps[b], (Xs) + (size_t)T * n + (size_t)b * nx); gs[t] = cc; } \
    double tot = 0.0; for (int t = 0; t < T; ++t) tot += gs[t]; tot += gs[T]; (out) = tot; } while (0)
  double best_cost; D_OBJECTIVE(x, u, best_cost);
  double current_merit = best_cost;
  int iterations = 0, exit_reason = MAS_EXIT_MAX_ITER; long reg_events = 0, trials = 0;
  int alpha_hist[12]; memset(alpha_hist, 0, sizeof(alpha_hist));

  for (int iter = 0; iter < max_iterations; ++iter) {
    const double elapsed_ms = (double)(long long)(now_ms() - start);
    if (elapsed_ms > max_ms) { exit_reason = MAS_EXIT_TIME; break; }
    iterations = iter + 1;

    /* terminal boundary (ilqr.hpp:92-102) */
    dtables(&s, x + (size_t)T * n, NULL, T);
    dgradient(&s, 0, vx);
    dhessian(&s, x + (size_t)T * n, NULL, T, 0, V);
    par_symmetrise(V, n);

    int reg_fail = 0;
    for (int t = T - 1; t >= 0; --t) {
      const double* xs = x + (size_t)t * n; const double* us = u + (size_t)t * m;
      /* linearisation of the stacked problem through the FD hooks (ilqr.hpp:106-113) */
#pragma omp parallel for schedule(static) if (g_dense_par)
      for (int b = 0; b < N; ++b) {
        mas_fd_dynamics_state_jacobian(ocps[b], xs + (size_t)b * nx, us + (size_t)b * nu, Ab + (size_t)b * nx * nx);
        mas_fd_dynamics_control_jacobian(ocps[b], xs + (size_t)b * nx, us + (size_t)b * nu, Bb + (size_t)b * nx * nu);
      }
      dtables(&s, xs, us, t);
      dgradient(&s, 0, lx); dgradient(&s, 1, lu);
      dhessian(&s, xs, us, t, 0, lxx); dhessian(&s, xs, us, t, 1, luu);
      dcross(&s, xs, us, t, lux);

      /* assemble_q (ilqr.hpp:115-119) with the block structure of A and B */
#pragma omp parallel for schedule(static) if (g_dense_par)
      for (int i = 0; i < n; ++i) {
        const int b = i / nx, ki = i % nx; const double* a = Ab + (size_t)b * nx * nx + (size_t)ki * nx;
        double sx = 0.0; for (int r = 0; r < nx; ++r) sx += a[r] * vx[(size_t)b * nx + r];
        qx[i] = lx[i] + sx;
      }
#pragma omp parallel for schedule(static) if (g_dense_par)
      for (int i = 0; i < m; ++i) {
        const int b = i / nu, ki = i % nu; const double* bc = Bb + (size_t)b * nx * nu + (size_t)ki * nx;
        double su = 0.0; for (int r = 0; r < nx; ++r) su += bc[r] * vx[(size_t)b * nx + r];
        qu[i] = lu[i] + su;
      }
#pragma omp parallel for schedule(static) if (g_dense_par)
      for (int j = 0; j < n; ++j) {
        for (int i = 0; i < n; ++i) {                                    /* tmp_xx = A^T V */
          const int b = i / nx, ki = i % nx; const double* a = Ab + (size_t)b * nx * nx + (size_t)ki * nx;
          double sv = 0.0; for (int r = 0; r < nx; ++r) sv += a[r] * V[(size_t)j * n + (size_t)b * nx + r];
          tmp_xx[(size_t)j * n + i] = sv;
        }
        for (int i = 0; i < m; ++i) {                                    /* tmp_ux = B^T V */
          const int b = i / nu, ki = i % nu; const double* bc = Bb + (size_t)b * nx * nu + (size_t)ki * nx;
          double sv = 0.0; for (int r = 0; r < nx; ++r) sv += bc[r] * V[(size_t)j * n + (size_t)b * nx + r];
          tmp_ux[(size_t)j * m + i] = sv;
        }
      }
#pragma omp parallel for schedule(static) if (g_dense_par)
      for (int j = 0; j < n; ++j) {
        const int bj = j / nx, kj = j % nx; const double* a = Ab + (size_t)bj * nx * nx + (size_t)kj * nx;
        for (int i = 0; i < n; ++i) {                                    /* q_xx = l_xx + (A^T V) A */
          double sv = 0.0; for (int r = 0; r < nx; ++r) sv += tmp_xx[((size_t)bj * nx + r) * n + i] * a[r];
          qxx[(size_t)j * n + i] = lxx[(size_t)j * n + i] + sv;
        }
        for (int i = 0; i < m; ++i) {                                    /* q_ux = l_ux + (B^T V) A */
          double sv = 0.0; for (int r = 0; r < nx; ++r) sv += tmp_ux[((size_t)bj * nx + r) * m + i] * a[r];
          qux[(size_t)j * m + i] = lux[(size_t)j * m + i] + sv;
        }
      }
#pragma omp parallel for schedule(static) if (g_dense_par)
      for (int j = 0; j < m; ++j) {
        const int bj = j / nu, kj = j % nu; const double* bc = Bb + (size_t)bj * nx * nu + (size_t)kj * nx;
        for (int i = 0; i < m; ++i) {                                    /* q_uu = l_uu + (B^T V) B */
          double sv = 0.0; for (int r = 0; r < nx; ++r) sv += tmp_ux[((size_t)bj * nx + r) * m + i] * bc[r];
          quu[(size_t)j * m + i] = luu[(size_t)j * m + i] + sv;
        }
      }

      /* LLT + x10 ladder (ilqr.hpp:172-182) */
      memcpy(cur, quu, sizeof(double) * m2);
      double reg = 1e-6; int rungs = 0;
      for (;;) {
        memcpy(quu_reg, cur, sizeof(double) * m2);
        if (par_llt_compute(quu_reg, m, acc) == 0) break;
        if (rungs >= MAS_REG_LADDER_MAX) { rungs = -1; break; }
        for (int i = 0; i < m; ++i) cur[(size_t)i * m + i] += reg;
        reg *= 10.0; ++rungs;
      }
      if (rungs < 0) { reg_fail = 1; break; }
      reg_events += rungs;

      /* gains_and_value (ilqr.hpp:183-192) */
      double* kt = k + (size_t)t * m; double* Kt = K + (size_t)t * mn;
      par_llt_solve_identity(quu_reg, m, quu_inv);
      for (size_t e = 0; e < m2; ++e) tmp_uu[e] = -quu_inv[e];
      par_AB(tmp_uu, m, m, qu, 1, kt);
      par_AB(tmp_uu, m, m, qux, n, Kt);
      par_AtB(Kt, m, n, quu, m, KtQuu);
      par_AtB(Kt, m, n, qu, 1, tmp_x);  for (int i = 0; i < n; ++i) vx[i] = qx[i] + tmp_x[i];
      par_AtB(qux, m, n, kt, 1, tmp_x); for (int i = 0; i < n; ++i) vx[i] = vx[i] + tmp_x[i];
      par_AB(KtQuu, n, m, kt, 1, tmp_x); for (int i = 0; i < n; ++i) vx[i] = vx[i] + tmp_x[i];
      par_AtB(Kt, m, n, qux, n, tmp_xx);  for (size_t e = 0; e < n2; ++e) V[e] = qxx[e] + tmp_xx[e];
      par_AtB(qux, m, n, Kt, n, tmp_xx);  for (size_t e = 0; e < n2; ++e) V[e] = V[e] + tmp_xx[e];
      par_AB(KtQuu, n, m, Kt, n, tmp_xx); for (size_t e = 0; e < n2; ++e) V[e] = V[e] + tmp_xx[e];
      par_symmetrise(V, n);
    }
    if (reg_fail) { exit_reason = MAS_EXIT_REG_LIMIT; break; }

    /* forward pass + line search (ilqr.hpp:195-228) */
    double alpha = 1.0, best_merit = current_merit; int accepted = 0, j = 0;
    while (alpha >= 1e-3) {
      memcpy(xt, x, sizeof(double) * n);
      for (int t = 0; t < T; ++t) {
        const double* xtt = xt + (size_t)t * n; double* utt = ut + (size_t)t * m;
        for (int i = 0; i < n; ++i) dx[i] = xtt[i] - x[(size_t)t * n + i];
        const double* Kt = K + (size_t)t * mn; const double* kt = k + (size_t)t * m;
#pragma omp parallel for schedule(static) if (g_dense_par)
        for (int i = 0; i < m; ++i) {
          double kd = 0.0; for (int c = 0; c < n; ++c) kd += Kt[(size_t)c * m + i] * dx[c];
          utt[i] = (u[(size_t)t * m + i] + alpha * kt[i]) + kd;
        }
        if (all_bounds) for (int b = 0; b < N; ++b) mas_clamp_controls_column(ocps[b], utt + (size_t)b * nu);
#pragma omp parallel for schedule(static) if (g_dense_par)
        for (int b = 0; b < N; ++b) mas_integrate_rk4(ocps[b], xtt + (size_t)b * nx, utt + (size_t)b * nu, dt, xt + (size_t)(t + 1) * n + (size_t)b * nx);
      }
      ++trials;
      double trial_merit; D_OBJECTIVE(xt, ut, trial_merit);
      if (trial_merit < best_merit) { best_merit = trial_merit; accepted = 1; break; }
      alpha *= 0.5; ++j;
    }
    alpha_hist[accepted ? j : 10]++;
    const double improvement = current_merit - best_merit;
    if (accepted) { memcpy(x, xt, sizeof(double) * (T + 1) * n); memcpy(u, ut, sizeof(double) * T * m); best_cost = best_merit; }
    current_merit = best_merit;
    if (improvement < tolerance) { exit_reason = MAS_EXIT_CONVERGED; break; }
  }

  /* hand the stacked solution back to the agents (centralized.hpp:28-35) */
  for (int b = 0; b < N; ++b) {
    mas_ocp* o = ocps[b];
    for (int t = 0; t <= T; ++t) memcpy(o->best_states + (size_t)t * nx, x + (size_t)t * n + (size_t)b * nx, sizeof(double) * nx);
    for (int t = 0; t < T; ++t) memcpy(o->best_controls + (size_t)t * nu, u + (size_t)t * m + (size_t)b * nu, sizeof(double) * nu);
    o->best_cost = mas_ocp_objective(o, o->best_states, o->best_controls);
  }
  if (stats) {
    stats->total_iterations += iterations; stats->converged_solves += (exit_reason == MAS_EXIT_CONVERGED);
    stats->reg_events += reg_events; stats->trials += trials;
    for (int q = 0; q < 12; ++q) stats->alpha_hist[q] += alpha_hist[q];
  }
#ifdef _OPENMP
  omp_set_num_threads(threads_before);
#endif
#undef D_ROLLOUT
#undef D_OBJECTIVE
  double* all[] = {s.c, s.pre, s.cx5, s.cx6, s.cu5, s.cu6, x, u, xt, ut, Ab, Bb, lx, lu, lxx, luu, lux, k, K, V, vx, qx, qu, qxx, qux, quu, quu_reg,
                   quu_inv, cur, tmp_xx, tmp_ux, tmp_x, tmp_uu, KtQuu, dx, acc, gs};
  for (size_t q = 0; q < sizeof(all) / sizeof(all[0]); ++q) free(all[q]);
  return best_cost;
}
