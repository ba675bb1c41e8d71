// mas_dense.hpp -- the centralised strategy in the reference's LITERAL reading, for small problems.
//
// CentralizedStrategy (strategies/centralized.hpp:18-38) stacks all agents into ONE OCP (build_global_ocp,
// multi_agent_problem.hpp:52-127: stacked dynamics, summed stage / terminal cost, stacked bounds, no analytic
// hooks) and runs ONE iLQR on it.  Every cost derivative is then a finite difference of the SUMMED cost, so the
// stacked Hessians are dense n x n / m x n / m x m matrices (n = N*nx, m = N*nu) whose cross-agent blocks hold
// rounding noise divided by eps^2, and the Riccati recursion, the LLT and the feedback gains are dense too.
// mas_central.hpp solves the block-diagonal reading of the same problem, which is what scales (the literal solve costs
// 4 (N nx)^2 evaluations of the N-agent cost per time step); this file is the literal solve itself, for the problem sizes
// at which the reference can run it at all (N <= kDenseMaxAgents), so that the two readings can be held against each
// other on the device and `--strategy centralized` has a drop-in that computes the reference's own number.
//
// One workgroup walks the whole solve (ilqr.hpp:58-273 on the stacked problem): every phase is a loop over independent
// output entries shared out over the threads (pfor) with a barrier behind it; every entry is computed by the same
// sequence of operations, in the same order, as the scalar statement of the reference (sums over k from 0.0, the summed
// cost added up agent by agent in block order), so the result does not depend on the thread count -- the host emulation
// runs the same code with one "thread".  Performance is not a goal here.
#pragma once

#include "mas_central.hpp"

namespace mas {

constexpr int kDenseMaxAgents = 64;     // (single-track, H = 100, one workgroup of 1024 lanes: 1.1 s at 16 agents, 7 s at 32, 56 s at 64)

template <class M>
struct DenseCentralBody {
  static constexpr int NX = M::NX, NU = M::NU;
  Dims d; double dt; ModelParams P;
  bool clamp; double lb[NU], ub[NU];
  double tolerance; int max_iterations; double max_ms;
  const double* x0;          // [NX][Np] SoA
  double* X; double* U;      // SoA trajectories in / out ([(T+1)*NX][Np], [T*NU][Np]); U is the zero start (ocp.hpp:106-109)
  double* cost;              // [Np] per-agent objective of the final trajectories (centralized.hpp:28-35)
  int* iters_total; int* exit_code; long long* reg_events;
  CentralState* st;
  double* ws;                // workspace, layout below
  int* flags;                // [4] workgroup-wide decisions (written by thread 0, read by all after a barrier)

  // ------------------------------------------------------------------ workspace layout ------
  struct Layout {
    int N, T, n, m;
    size_t Xg, Ug, Xt, Ut, A, B, lx, lu, lxx, luu, lux, k, K, V, vx, qx, qu, qxx, qux, quu, quu_reg, quu_inv, cur, tmp_xx, tmp_ux, tmp_x,
        tmp_u, tmp_uu, KtQuu, dx, gs, total;
    MAS_HD Layout(int N_, int T_) : N(N_), T(T_), n(N_ * NX), m(N_ * NU) {
      size_t o = 0;
      auto take = [&](size_t c) { const size_t at = o; o += c; return at; };
      const size_t n2 = static_cast<size_t>(n) * n, m2 = static_cast<size_t>(m) * m, mn = static_cast<size_t>(m) * n;
      Xg = take(static_cast<size_t>(T + 1) * n); Ug = take(static_cast<size_t>(T) * m);
      Xt = take(static_cast<size_t>(T + 1) * n); Ut = take(static_cast<size_t>(T) * m);
      A = take(T * n2); B = take(T * mn); lx = take(static_cast<size_t>(T) * n); lu = take(static_cast<size_t>(T) * m);
      lxx = take(T * n2); luu = take(T * m2); lux = take(T * mn);
      k = take(static_cast<size_t>(T) * m); K = take(T * mn);
      V = take(n2); vx = take(n); qx = take(n); qu = take(m); qxx = take(n2); qux = take(mn); quu = take(m2); quu_reg = take(m2);
      quu_inv = take(m2); cur = take(m2); tmp_xx = take(n2); tmp_ux = take(mn); tmp_x = take(n); tmp_u = take(m); tmp_uu = take(m2);
      KtQuu = take(mn); dx = take(n); gs = take(static_cast<size_t>(T) + 1);
      total = o;
    }
  };
  static size_t workspace_doubles(int N, int T) { return Layout(N, T).total; }

  // ------------------------------------------------------------------ threads ------
#if defined(__HIP_DEVICE_COMPILE__)
  template <class F> __device__ void pfor(int count, F f) const { for (int e = static_cast<int>(threadIdx.x); e < count; e += static_cast<int>(blockDim.x)) f(e); __syncthreads(); }
  template <class F> __device__ void once(F f) const { if (threadIdx.x == 0) f(); __syncthreads(); }
#else
  template <class F> void pfor(int count, F f) const { for (int e = 0; e < count; ++e) f(e); }
  template <class F> void once(F f) const { f(); }
#endif

  // ------------------------------------------------------------------ the stacked callbacks ------
  // which coordinates of the stacked state / control vector are perturbed, and by how much (applied in the order
  // first, second -- Eigen's x + dx_i + dx_j)
  struct Pert { int xi = -1; double dxi = 0.0; int xj = -1; double dxj = 0.0; int ui = -1; double dui = 0.0; int uj = -1; double duj = 0.0; };
  MAS_HD static void local_x(const double* Xs, int b, const Pert& p, double* x) {
    MAS_UNROLL for (int k = 0; k < NX; ++k) {
      const int g = b * NX + k; double v = Xs[g];
      if (g == p.xi) v = v + p.dxi;
      if (g == p.xj) v = v + p.dxj;
      x[k] = v;
    }
  }
  MAS_HD static void local_u(const double* Us, int b, const Pert& p, double* u) {
    MAS_UNROLL for (int k = 0; k < NU; ++k) {
      const int g = b * NU + k; double v = Us[g];
      if (g == p.ui) v = v + p.dui;
      if (g == p.uj) v = v + p.duj;
      u[k] = v;
    }
  }
  // stacked stage cost (multi_agent_problem.hpp:94-105): the agents' costs added up in block order from 0.0
  MAS_HD double gstage(const double* Xs, const double* Us, int t, const Pert& p, int N) const {
    double c = 0.0;
    for (int b = 0; b < N; ++b) { double x[NX], u[NU]; local_x(Xs, b, p, x); local_u(Us, b, p, u); c += M::stage_cost(x, u, t, P); }
    return c;
  }
  MAS_HD double gterminal(const double* Xs, const Pert& p, int N) const {                      // :107-113
    double c = 0.0;
    for (int b = 0; b < N; ++b) { double x[NX]; local_x(Xs, b, p, x); c += M::terminal_cost(x, P); }
    return c;
  }

  // ------------------------------------------------------------------ one solve ------
  MAS_HD void run(int N) const {
    const Layout L(N, d.T);
    const int T = d.T, n = L.n, m = L.m;
    double* Xg = ws + L.Xg; double* Ug = ws + L.Ug; double* Xt = ws + L.Xt; double* Ut = ws + L.Ut;
    const Pert none{};

    // stacked copies of the start: x0, zero controls, rollout, objective (build_global_ocp + initialize_problem + ilqr.hpp:75-78)
    pfor(T * m, [&](int e) { Ug[e] = 0.0; });
    pfor(n, [&](int e) { Xg[e] = MAS_AT(x0, e % NX, e / NX); });
    rollout(Xg, Ug, N);
    double current_merit = 0.0;
    objective(Xg, Ug, N, L);
    current_merit = ws[L.gs + T];                      // (the accumulated objective is left behind the stage sums)
    double best_cost = current_merit;
    long long regs = 0;
    int iterations = 0, exit_reason = kExitMaxIter;
    const unsigned long long start = MAS_NOW_TICKS();

    for (int iter = 0; iter < max_iterations; ++iter) {
      // elapsed_ms > max_ms at the top of the iteration (ilqr.hpp:84-90); one clock for the one stacked solve
      once([&]() {
        const double elapsed_ms = static_cast<double>(static_cast<long long>(static_cast<double>(MAS_NOW_TICKS() - start) / kTicksPerMs));
        flags[0] = (max_ms < 1e300 && elapsed_ms > max_ms) ? 1 : 0;
      });
      if (flags[0]) { exit_reason = kExitTime; break; }
      iterations = iter + 1;

      linearise_all(Xg, Ug, N, L);
      terminal_boundary_dense(Xg + static_cast<size_t>(T) * n, N, L);
      for (int t = T - 1; t >= 0; --t) regs += backward_step(t, L);

      // line search (ilqr.hpp:195-228): alpha = 1, 1/2, ... while alpha >= 1e-3, first strict decrease of the merit
      double best_merit = current_merit, alpha = 1.0;
      bool accepted = false;
      while (alpha >= 1e-3) {
        trial_rollout(Xg, Ug, Xt, Ut, alpha, N, L);
        objective(Xt, Ut, N, L);
        const double trial_merit = ws[L.gs + T];
        if (trial_merit < best_merit) { best_merit = trial_merit; accepted = true; break; }
        alpha *= 0.5;
      }
      const double improvement = current_merit - best_merit;
      if (accepted) {
        pfor((T + 1) * n, [&](int e) { Xg[e] = Xt[e]; });
        pfor(T * m, [&](int e) { Ug[e] = Ut[e]; });
        best_cost = best_merit;                        // cost = objective(best_x, best_u) == the accepted trial's merit
      }
      current_merit = best_merit;
      if (improvement < tolerance) { exit_reason = kExitConverged; break; }
    }

    // hand the stacked solution back to the agents (centralized.hpp:28-35): trajectories, per-agent objective, total = stacked best_cost
    pfor(N * (T + 1), [&](int e) {
      const int b = e / (T + 1), t = e % (T + 1);
      MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(X, t * NX + i, b) = Xg[static_cast<size_t>(t) * n + b * NX + i];
      if (t < T) { MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(U, t * NU + i, b) = Ug[static_cast<size_t>(t) * m + b * NU + i]; }
    });
    pfor(N, [&](int b) {
      double c = 0.0, x[NX], u[NU];
      for (int t = 0; t < T; ++t) { local_x(Xg + static_cast<size_t>(t) * n, b, none, x); local_u(Ug + static_cast<size_t>(t) * m, b, none, u); c += M::stage_cost(x, u, t, P); }
      local_x(Xg + static_cast<size_t>(T) * n, b, none, x);
      c += M::terminal_cost(x, P);
      cost[b] = c; iters_total[b] += iterations; exit_code[b] = exit_reason;
    });
    once([&]() {
      st->cur_total = best_cost; st->iterations = iterations; st->done = exit_reason == kExitConverged ? 1 : 0; st->j_sel = -1; st->improvement = 0.0;
      reg_events[0] += regs;
    });
  }

  // integrate_horizon on the stacked problem: every agent's own RK4 (the stacked dynamics is the agents' dynamics side by side)
  MAS_HD void rollout(double* Xs, const double* Us, int N) const {
    const int n = N * NX, m = N * NU;
    for (int t = 0; t < d.T; ++t)
      pfor(N, [&](int b) {
        double x[NX], u[NU], xn[NX];
        MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = Xs[static_cast<size_t>(t) * n + b * NX + i];
        MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = Us[static_cast<size_t>(t) * m + b * NU + i];
        rk4_step<M>(x, u, dt, xn, P);
        MAS_UNROLL for (int i = 0; i < NX; ++i) Xs[static_cast<size_t>(t + 1) * n + b * NX + i] = xn[i];
      });
  }
  // compute_trajectory_cost (ocp.hpp:14-28) of the stacked problem: stage sums per t, then added up in t order, then the terminal cost
  MAS_HD void objective(const double* Xs, const double* Us, int N, const Layout& L) const {
    const int T = d.T, n = L.n, m = L.m;
    const Pert none{};
    pfor(T + 1, [&](int t) { ws[L.gs + t] = t < T ? gstage(Xs + static_cast<size_t>(t) * n, Us + static_cast<size_t>(t) * m, t, none, N) : gterminal(Xs + static_cast<size_t>(T) * n, none, N); });
    once([&]() { double c = 0.0; for (int t = 0; t < T; ++t) c += ws[L.gs + t]; c += ws[L.gs + T]; ws[L.gs + T] = c; });
  }

  // ilqr.hpp:106-113 through the FD hooks of the stacked OCP (finite_differences.hpp:53-287), all time steps
  MAS_HD void linearise_all(const double* Xg, const double* Ug, int N, const Layout& L) const {
    const int T = d.T, n = L.n, m = L.m;
    const size_t n2 = static_cast<size_t>(n) * n, m2 = static_cast<size_t>(m) * m, mn = static_cast<size_t>(m) * n;
    // A, B: column i = (f(+eps e_i) - f(-eps e_i)) / (2 eps); rows of other agents are (f - f) / (2 eps) = 0
    pfor(T * n, [&](int e) {
      const int t = e / n, i = e % n, b = i / NX;
      const double eps = 1e-6;
      double* col = ws + L.A + t * n2 + static_cast<size_t>(i) * n;
      for (int r = 0; r < n; ++r) col[r] = (0.0 - 0.0) / (2 * eps);
      double xp[NX], xm[NX], u[NU], fp[NX], fm[NX];
      Pert pp; pp.xi = i; pp.dxi = +eps; Pert pm; pm.xi = i; pm.dxi = -eps;
      local_x(Xg + static_cast<size_t>(t) * n, b, pp, xp); local_x(Xg + static_cast<size_t>(t) * n, b, pm, xm); local_u(Ug + static_cast<size_t>(t) * m, b, pp, u);
      M::dynamics(xp, u, fp, P); M::dynamics(xm, u, fm, P);
      MAS_UNROLL for (int r = 0; r < NX; ++r) col[b * NX + r] = (fp[r] - fm[r]) / (2 * eps);
    });
    pfor(T * m, [&](int e) {
      const int t = e / m, i = e % m, b = i / NU;
      const double eps = 1e-6;
      double* col = ws + L.B + t * mn + static_cast<size_t>(i) * n;
      for (int r = 0; r < n; ++r) col[r] = (0.0 - 0.0) / (2 * eps);
      double x[NX], up[NU], um[NU], fp[NX], fm[NX];
      Pert pp; pp.ui = i; pp.dui = +eps; Pert pm; pm.ui = i; pm.dui = -eps;
      local_x(Xg + static_cast<size_t>(t) * n, b, pp, x); local_u(Ug + static_cast<size_t>(t) * m, b, pp, up); local_u(Ug + static_cast<size_t>(t) * m, b, pm, um);
      M::dynamics(x, up, fp, P); M::dynamics(x, um, fm, P);
      MAS_UNROLL for (int r = 0; r < NX; ++r) col[b * NX + r] = (fp[r] - fm[r]) / (2 * eps);
    });
    // gradients, eps = 1e-6, no safe_eval (:110-136)
    pfor(T * n, [&](int e) {
      const int t = e / n, i = e % n; const double eps = 1e-6;
      const double* xs = Xg + static_cast<size_t>(t) * n; const double* us = Ug + static_cast<size_t>(t) * m;
      Pert pp; pp.xi = i; pp.dxi = +eps; Pert pm; pm.xi = i; pm.dxi = -eps;
      ws[L.lx + static_cast<size_t>(t) * n + i] = (gstage(xs, us, t, pp, N) - gstage(xs, us, t, pm, N)) / (2 * eps);
    });
    pfor(T * m, [&](int e) {
      const int t = e / m, i = e % m; const double eps = 1e-6;
      const double* xs = Xg + static_cast<size_t>(t) * n; const double* us = Ug + static_cast<size_t>(t) * m;
      Pert pp; pp.ui = i; pp.dui = +eps; Pert pm; pm.ui = i; pm.dui = -eps;
      ws[L.lu + static_cast<size_t>(t) * m + i] = (gstage(xs, us, t, pp, N) - gstage(xs, us, t, pm, N)) / (2 * eps);
    });
    // Hessians, eps = 1e-5, through safe_eval; (i,j) and (j,i) separately (:138-210)
    pfor(static_cast<int>(T * n2), [&](int e) {
      const int t = static_cast<int>(e / n2), ij = static_cast<int>(e % n2), i = ij % n, j = ij / n; const double eps = 1e-5;
      const double* xs = Xg + static_cast<size_t>(t) * n; const double* us = Ug + static_cast<size_t>(t) * m;
      double h;
      if (i == j) {
        Pert pp; pp.xi = i; pp.dxi = +eps; Pert pm; pm.xi = i; pm.dxi = -eps;
        const double f_plus = safe_value(gstage(xs, us, t, pp, N)), f = safe_value(gstage(xs, us, t, Pert{}, N)), f_minus = safe_value(gstage(xs, us, t, pm, N));
        h = (f_plus - 2 * f + f_minus) / (eps * eps);
      } else {
        Pert a; a.xi = i; a.dxi = +eps; a.xj = j; a.dxj = +eps; Pert b2 = a; b2.dxj = -eps; Pert c = a; c.dxi = -eps; Pert dd = c; dd.dxj = -eps;
        const double f_pp = safe_value(gstage(xs, us, t, a, N)), f_pm = safe_value(gstage(xs, us, t, b2, N));
        const double f_mp = safe_value(gstage(xs, us, t, c, N)), f_mm = safe_value(gstage(xs, us, t, dd, N));
        h = (f_pp - f_pm - f_mp + f_mm) / (4 * eps * eps);
      }
      ws[L.lxx + t * n2 + static_cast<size_t>(j) * n + i] = h;
    });
    pfor(static_cast<int>(T * m2), [&](int e) {
      const int t = static_cast<int>(e / m2), ij = static_cast<int>(e % m2), i = ij % m, j = ij / m; const double eps = 1e-5;
      const double* xs = Xg + static_cast<size_t>(t) * n; const double* us = Ug + static_cast<size_t>(t) * m;
      double h;
      if (i == j) {
        Pert pp; pp.ui = i; pp.dui = +eps; Pert pm; pm.ui = i; pm.dui = -eps;
        const double f_plus = safe_value(gstage(xs, us, t, pp, N)), f = safe_value(gstage(xs, us, t, Pert{}, N)), f_minus = safe_value(gstage(xs, us, t, pm, N));
        h = (f_plus - 2 * f + f_minus) / (eps * eps);
      } else {
        Pert a; a.ui = i; a.dui = +eps; a.uj = j; a.duj = +eps; Pert b2 = a; b2.duj = -eps; Pert c = a; c.dui = -eps; Pert dd = c; dd.duj = -eps;
        const double f_pp = safe_value(gstage(xs, us, t, a, N)), f_pm = safe_value(gstage(xs, us, t, b2, N));
        const double f_mp = safe_value(gstage(xs, us, t, c, N)), f_mm = safe_value(gstage(xs, us, t, dd, N));
        h = (f_pp - f_pm - f_mp + f_mm) / (4 * eps * eps);
      }
      ws[L.luu + t * m2 + static_cast<size_t>(j) * m + i] = h;
    });
    // cross term m x n, eps = 1e-6: f_pm = f(x - dx, u + du), f_mp = f(x + dx, u - du) (:263-287)
    pfor(static_cast<int>(T * mn), [&](int e) {
      const int t = static_cast<int>(e / mn), ij = static_cast<int>(e % mn), i = ij % m, j = ij / m; const double eps = 1e-6;
      const double* xs = Xg + static_cast<size_t>(t) * n; const double* us = Ug + static_cast<size_t>(t) * m;
      Pert a; a.ui = i; a.dui = +eps; a.xi = j; a.dxi = +eps; Pert b2 = a; b2.dxi = -eps; Pert c = a; c.dui = -eps; Pert dd = c; dd.dxi = -eps;
      const double f_pp = safe_value(gstage(xs, us, t, a, N)), f_pm = safe_value(gstage(xs, us, t, b2, N));
      const double f_mp = safe_value(gstage(xs, us, t, c, N)), f_mm = safe_value(gstage(xs, us, t, dd, N));
      ws[L.lux + t * mn + static_cast<size_t>(j) * m + i] = (f_pp - f_pm - f_mp + f_mm) / (4 * eps * eps);
    });
  }

  // v_x, v_xx at the horizon (ilqr.hpp:92-102; finite_differences.hpp:212-261) and the aliased symmetrisation
  MAS_HD void terminal_boundary_dense(const double* xT, int N, const Layout& L) const {
    const int n = L.n;
    pfor(n, [&](int i) {
      const double eps = 1e-6; Pert pp; pp.xi = i; pp.dxi = +eps; Pert pm; pm.xi = i; pm.dxi = -eps;
      ws[L.vx + i] = (gterminal(xT, pp, N) - gterminal(xT, pm, N)) / (2 * eps);
    });
    pfor(n * n, [&](int ij) {
      const int i = ij % n, j = ij / n; const double eps = 1e-5;
      double h;
      if (i == j) {
        Pert pp; pp.xi = i; pp.dxi = +eps; Pert pm; pm.xi = i; pm.dxi = -eps;
        const double f_plus = safe_value(gterminal(xT, pp, N)), f = safe_value(gterminal(xT, Pert{}, N)), f_minus = safe_value(gterminal(xT, pm, N));
        h = (f_plus - 2 * f + f_minus) / (eps * eps);
      } else {
        Pert a; a.xi = i; a.dxi = +eps; a.xj = j; a.dxj = +eps; Pert b2 = a; b2.dxj = -eps; Pert c = a; c.dxi = -eps; Pert dd = c; dd.dxj = -eps;
        const double f_pp = safe_value(gterminal(xT, a, N)), f_pm = safe_value(gterminal(xT, b2, N));
        const double f_mp = safe_value(gterminal(xT, c, N)), f_mm = safe_value(gterminal(xT, dd, N));
        h = (f_pp - f_pm - f_mp + f_mm) / (4 * eps * eps);
      }
      ws[L.V + static_cast<size_t>(j) * n + i] = h;
    });
    symmetrise_dense(ws + L.V, n);
  }
  // v_xx = 0.5 (v_xx + v_xx^T) evaluated in place in column-major order: an entry at or below the diagonal
  // averages the old values; an entry above it sees its already rewritten mirror -- two phases, the same numbers as the serial loop
  MAS_HD void symmetrise_dense(double* V, int n) const {
    pfor(n * n, [&](int ij) { const int i = ij % n, j = ij / n; if (i > j) V[static_cast<size_t>(j) * n + i] = 0.5 * (V[static_cast<size_t>(j) * n + i] + V[static_cast<size_t>(i) * n + j]); });
    pfor(n * n, [&](int ij) {
      const int i = ij % n, j = ij / n;
      if (i == j) V[static_cast<size_t>(j) * n + i] = 0.5 * (V[static_cast<size_t>(j) * n + i] + V[static_cast<size_t>(i) * n + j]);
      else if (i < j) V[static_cast<size_t>(j) * n + i] = 0.5 * (V[static_cast<size_t>(j) * n + i] + V[static_cast<size_t>(i) * n + j]);   // mirror (j,i) is below the diagonal: already new
    });
  }

  // C(mm x nn) = A^T B with A stored p x mm, B stored p x nn; C(mm x nn) = A B with A stored mm x p (column-major, sums over k from 0.0)
  MAS_HD void mat_AtB_d(const double* A, int p, int mm, const double* B, int nn, double* C) const {
    pfor(mm * nn, [&](int e) { const int i = e % mm, j = e / mm; double s = 0.0; for (int k2 = 0; k2 < p; ++k2) s += A[static_cast<size_t>(i) * p + k2] * B[static_cast<size_t>(j) * p + k2]; C[static_cast<size_t>(j) * mm + i] = s; });
  }
  MAS_HD void mat_AB_d(const double* A, int mm, int p, const double* B, int nn, double* C) const {
    pfor(mm * nn, [&](int e) { const int i = e % mm, j = e / mm; double s = 0.0; for (int k2 = 0; k2 < p; ++k2) s += A[static_cast<size_t>(k2) * mm + i] * B[static_cast<size_t>(j) * p + k2]; C[static_cast<size_t>(j) * mm + i] = s; });
  }
  MAS_HD void add_into(double* dst, const double* a, const double* b, int count) const { pfor(count, [&](int e) { dst[e] = a[e] + b[e]; }); }

  // one step of the backward recursion on the dense stacked problem (ilqr.hpp:115-119,172-192); returns the rungs used
  MAS_HD int backward_step(int t, const Layout& L) const {
    const int n = L.n, m = L.m;
    const size_t n2 = static_cast<size_t>(n) * n, m2 = static_cast<size_t>(m) * m, mn = static_cast<size_t>(m) * n;
    const double* A = ws + L.A + t * n2; const double* B = ws + L.B + t * mn;
    double* V = ws + L.V; double* vx = ws + L.vx;
    double *qx = ws + L.qx, *qu = ws + L.qu, *qxx = ws + L.qxx, *qux = ws + L.qux, *quu = ws + L.quu, *quu_reg = ws + L.quu_reg, *quu_inv = ws + L.quu_inv;
    double *cur = ws + L.cur, *tmp_xx = ws + L.tmp_xx, *tmp_ux = ws + L.tmp_ux, *tmp_x = ws + L.tmp_x, *tmp_u = ws + L.tmp_u, *tmp_uu = ws + L.tmp_uu, *KtQuu = ws + L.KtQuu;
    double* kt = ws + L.k + static_cast<size_t>(t) * m; double* Kt = ws + L.K + t * mn;

    mat_AtB_d(A, n, n, vx, 1, tmp_x); add_into(qx, ws + L.lx + static_cast<size_t>(t) * n, tmp_x, n);
    mat_AtB_d(B, n, m, vx, 1, tmp_u); add_into(qu, ws + L.lu + static_cast<size_t>(t) * m, tmp_u, m);
    mat_AtB_d(A, n, n, V, n, tmp_xx); mat_AB_d(tmp_xx, n, n, A, n, qxx); add_into(qxx, ws + L.lxx + t * n2, qxx, static_cast<int>(n2));
    mat_AtB_d(B, n, m, V, n, tmp_ux); mat_AB_d(tmp_ux, m, n, A, n, qux); add_into(qux, ws + L.lux + t * mn, qux, static_cast<int>(mn));
    mat_AB_d(tmp_ux, m, n, B, m, quu); add_into(quu, ws + L.luu + t * m2, quu, static_cast<int>(m2));

    // LLT + x10 ladder (:172-182): one thread factorises (Eigen's unblocked lower LLT), everybody reads the verdict
    pfor(static_cast<int>(m2), [&](int e) { cur[e] = quu[e]; });
    double reg = 1e-6; int rungs = 0;
    for (;;) {
      pfor(static_cast<int>(m2), [&](int e) { quu_reg[e] = cur[e]; });
      once([&]() { flags[1] = llt_compute_dyn(quu_reg, m); });
      if (flags[1] == 0 || rungs >= kRegLadderMax) break;
      pfor(m, [&](int i) { cur[static_cast<size_t>(i) * m + i] += reg; });
      reg *= 10.0; ++rungs;
    }
    // Quu_reg^-1 = LLT::solve(I), one identity column per thread (:183)
    pfor(m, [&](int c) {
      double* o = quu_inv + static_cast<size_t>(c) * m;
      for (int i = 0; i < m; ++i) o[i] = (i == c) ? 1.0 : 0.0;
      for (int i = 0; i < m; ++i) {
        const double a = 1.0 / quu_reg[static_cast<size_t>(i) * m + i];
        o[i] *= a;
        const double b2 = o[i];
        for (int r = i + 1; r < m; ++r) o[r] -= b2 * quu_reg[static_cast<size_t>(i) * m + r];
      }
      for (int i = m - 1; i >= 0; --i) {
        const double a = 1.0 / quu_reg[static_cast<size_t>(i) * m + i];
        double b2 = 0.0;
        for (int r = i + 1; r < m; ++r) b2 += quu_reg[static_cast<size_t>(i) * m + r] * o[r];
        o[i] = (o[i] - b2) * a;
      }
    });
    pfor(static_cast<int>(m2), [&](int e) { tmp_uu[e] = -quu_inv[e]; });
    mat_AB_d(tmp_uu, m, m, qu, 1, kt);                                 // k = -Quu^-1 q_u   (:185)
    mat_AB_d(tmp_uu, m, m, qux, n, Kt);                                // K = -Quu^-1 q_ux  (:186)
    mat_AtB_d(Kt, m, n, quu, m, KtQuu);                                // K^T q_uu (unregularised)
    mat_AtB_d(Kt, m, n, qu, 1, tmp_x); add_into(vx, qx, tmp_x, n);     // v_x (:188-189)
    mat_AtB_d(qux, m, n, kt, 1, tmp_x); add_into(vx, vx, tmp_x, n);
    mat_AB_d(KtQuu, n, m, kt, 1, tmp_x); add_into(vx, vx, tmp_x, n);
    mat_AtB_d(Kt, m, n, qux, n, tmp_xx); add_into(V, qxx, tmp_xx, static_cast<int>(n2));   // v_xx (:190-191)
    mat_AtB_d(qux, m, n, Kt, n, tmp_xx); add_into(V, V, tmp_xx, static_cast<int>(n2));
    mat_AB_d(KtQuu, n, m, Kt, n, tmp_xx); add_into(V, V, tmp_xx, static_cast<int>(n2));
    symmetrise_dense(V, n);                                            // :192
    return rungs;
  }
  // Eigen::LLT (unblocked, lower) on an mm x mm column-major matrix; 1 if a pivot is <= 0 (NaN pivots pass, like Eigen)
  MAS_HD static int llt_compute_dyn(double* Mx, int mm) {
    for (int k2 = 0; k2 < mm; ++k2) {
      double x = Mx[static_cast<size_t>(k2) * mm + k2];
      if (k2 > 0) { double s = 0.0; for (int j = 0; j < k2; ++j) s += Mx[static_cast<size_t>(j) * mm + k2] * Mx[static_cast<size_t>(j) * mm + k2]; x = Mx[static_cast<size_t>(k2) * mm + k2] - s; }
      if (x <= 0.0) return 1;
      x = sqrt(x);
      Mx[static_cast<size_t>(k2) * mm + k2] = x;
      for (int r = k2 + 1; r < mm; ++r) {
        double a = Mx[static_cast<size_t>(k2) * mm + r];
        for (int j = 0; j < k2; ++j) a += (-Mx[static_cast<size_t>(j) * mm + k2]) * Mx[static_cast<size_t>(j) * mm + r];
        Mx[static_cast<size_t>(k2) * mm + r] = a / x;
      }
    }
    return 0;
  }

  // one line-search trial on the stacked problem (ilqr.hpp:208-217): dense feedback, clamp, every agent's RK4
  MAS_HD void trial_rollout(const double* Xg, const double* Ug, double* Xt, double* Ut, double alpha, int N, const Layout& L) const {
    const int T = d.T, n = L.n, m = L.m;
    const size_t mn = static_cast<size_t>(m) * n;
    double* dx = ws + L.dx;
    pfor(n, [&](int e) { Xt[e] = Xg[e]; });                            // x_trial.col(0) = initial_state
    for (int t = 0; t < T; ++t) {
      const double* kt = ws + L.k + static_cast<size_t>(t) * m; const double* Kt = ws + L.K + t * mn;
      pfor(n, [&](int c) { dx[c] = Xt[static_cast<size_t>(t) * n + c] - Xg[static_cast<size_t>(t) * n + c]; });
      pfor(m, [&](int i) {
        double kd = 0.0;
        for (int c = 0; c < n; ++c) kd += Kt[static_cast<size_t>(c) * m + i] * dx[c];
        double v = (Ug[static_cast<size_t>(t) * m + i] + alpha * kt[i]) + kd;
        if (clamp) { const int q = i % NU; v = (ub[q] < v) ? ub[q] : v; v = (v < lb[q]) ? lb[q] : v; }
        Ut[static_cast<size_t>(t) * m + i] = v;
      });
      pfor(N, [&](int b) {
        double x[NX], u[NU], xn[NX];
        MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = Xt[static_cast<size_t>(t) * n + b * NX + i];
        MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = Ut[static_cast<size_t>(t) * m + b * NU + i];
        rk4_step<M>(x, u, dt, xn, P);
        MAS_UNROLL for (int i = 0; i < NX; ++i) Xt[static_cast<size_t>(t + 1) * n + b * NX + i] = xn[i];
      });
    }
  }

#if defined(MAS_HOST_EMULATION)
  void run_serial(int N) const { run(N); }
#else
  __device__ void run_block(int N) const { run(N); }
#endif
};

}  // namespace mas
