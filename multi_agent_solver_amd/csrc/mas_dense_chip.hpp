// mas_dense_chip.hpp -- the centralised strategy in the reference's LITERAL reading, spread over the chip (HIP build only).
//
// mas_dense.hpp walks CentralizedStrategy (strategies/centralized.hpp:18-38: ONE iLQR, ilqr.hpp:58-273, on the stacked OCP of
// build_global_ocp, multi_agent_problem.hpp:52-127) with ONE workgroup: 1.1 / 7 / 56 s at 16 / 32 / 64 agents (H = 100).  This file is
// the same solve -- every number it stores is the value the reference's scalar statement produces -- as a sequence of chip-wide
// launches, so that BASELINE config 2's 256 agents are computable (n = 1024 stacked states, m = 512 stacked controls).
//
// What makes it affordable is exact re-use, nothing else (the CPU checker's structured restatement of the solve re-uses the
// same two facts and is held against its plain restatement bit for bit, tests/test_centralized_structured.py):
//   * the stacked stage cost is  c = 0; c += cost_b  in block order (multi_agent_problem.hpp:94-105).  A finite difference moves one
//     or two coordinates, i.e. the costs of at most two agents: every evaluation is the stored prefix of that very chain up to the
//     first moved agent and the same additions from there on, with one or two table values substituted (DenseTables);
//   * A and B of the stacked problem are block-diagonal with exact zeros outside the blocks ((f - f) / (2 eps)): the terms a
//     product sum takes from there are +-0.0 and leave a finite sum's value unchanged; they are skipped.  Precondition: finite V.
// The dense part stays dense and keeps every entry's sum over k from 0.0 in index order (products rounded before they are added:
// no FMA, no MFMA): tiled GEMMs (dc_gemm), the x10 ladder with its rungs tried SIDE BY SIDE (one workgroup per rung, left-looking
// LLT by rows, a rung beyond the lowest success is abandoned), LLT::solve(I) with the columns dealt over workgroups.
//
// Kernel boundaries are the grid barriers (1.5-1.9 us each on this part against 4-9 us for an in-kernel barrier, no residency
// condition, coherence for free); control flow that depends on results (which rung, which step size, converged?) lives in a small
// state block on the device that every kernel reads, and the host reads ONE pinned word per iteration.
#pragma once

#include <climits>

#include "mas_dense.hpp"

namespace mas {

constexpr int kDenseChipRungs = 16;          // rungs of the x10 ladder tried side by side (rung r + 16 q by workgroup r)
constexpr int kDenseChipMaxM = 1024;         // stacked controls (two rows per thread in the LLT, one LDS row of the factor)
constexpr int kDenseChipMaxAgents = 256;     // (cost tables of a time step in LDS; workspace ~3.5 GB at (4,2), H = 100)

struct DenseChipState {                      // control block of one stacked solve
  int t;                                     // time step the backward pass stands at (T: terminal boundary)
  int stop;                                  // the solve has ended: every later kernel returns at once
  int iterations, exit_reason;
  int rung_best;                             // lowest rung of the ladder whose LLT succeeded (INT_MAX: none yet)
  int accepted_j;                            // line search: first step size with a strict decrease of the merit, -1: none
  int failed, pad;
  long long regs;
  unsigned long long rung_failed;            // ladder: bit r set once rung r < 64 has failed (a later batch starts when all rungs below it have)
  double current_merit, best_cost, improvement;
  unsigned long long start_ticks;
};

struct DenseChipWs {                         // device arrays of the stacked problem (column-major like the reference's Eigen matrices)
  DenseChipState* S;
  double *Xg, *Ug, *Xt, *Ut;                 // [(T+1) n], [T m]; trials: kNumAlphas of each
  double *Ab, *Bb;                           // [N][nx nx], [N][nx nu]: the blocks of A and B at the current time step
  double *c, *pre, *cx5, *cu5, *cx6, *cu6;   // DenseTables of the current time step
  double *lx, *lu, *lxx, *luu, *lux;
  double *k, *K;                             // [T m], [T m n]
  double *V, *vx, *qx, *qu, *qxx, *qux, *quu, *Rr, *Lt, *SD, *inv, *tmp_xx, *tmp_ux, *V3, *KtQuu, *gs;
  int* rung_state;                           // [16] blocked ladder: 0 = being factorised, 1 = failed, 2 = factorised
  long long* host_word;                      // pinned: (iterations << 8) | stop, stored once per iteration
};

// ---- cost tables of one time step in LDS: c[N], pre[N+1], cx5/cx6 [N][NX][2], cu5/cu6 [N][NU][2] (index 0: +eps, 1: -eps) ----
template <int NX, int NU>
struct DenseTables {
  const double *c, *pre, *cx5, *cu5, *cx6, *cu6; int N;
  __host__ __device__ static size_t doubles(int N) { return static_cast<size_t>(N) * (2 + 4 * NX + 4 * NU) + 1; }
  __device__ void load(double* lds, const DenseChipWs& w, int N_) {
    N = N_;
    double* p = lds;
    auto take = [&](const double* src, int count) { for (int e = static_cast<int>(threadIdx.x); e < count; e += static_cast<int>(blockDim.x)) p[e] = src[e]; const double* at = p; p += count; return at; };
    c = take(w.c, N); pre = take(w.pre, N + 1); cx5 = take(w.cx5, N * NX * 2); cu5 = take(w.cu5, N * NU * 2); cx6 = take(w.cx6, N * NX * 2); cu6 = take(w.cu6, N * NU * 2);
    __syncthreads();
  }
  // K evaluations of the chain  s = 0; s += cost_b  side by side (the K differences of one finite-difference stencil): agent b1's cost
  // replaced by v1[s] and, if b2 >= 0, agent b2's by v2[s] (b1 != b2, any order); one walk over the agents, every c[b] read once
  template <int K>
  __device__ void chains(int b1, const double* v1, int b2, const double* v2, double* out) const {
    int lo = b1, hi = b2; const double* vlo = v1; const double* vhi = b2 >= 0 ? v2 : v1;
    if (b2 >= 0 && b2 < b1) { lo = b2; hi = b1; vlo = v2; vhi = v1; }
    double a[K];
    const double p0 = pre[lo];
    MAS_UNROLL for (int s = 0; s < K; ++s) a[s] = p0 + vlo[s];
    for (int b = lo + 1; b < N; ++b) {
      const double cb = c[b];
      MAS_UNROLL for (int s = 0; s < K; ++s) { const double v = (b == hi) ? vhi[s] : cb; a[s] += v; }
    }
    MAS_UNROLL for (int s = 0; s < K; ++s) out[s] = a[s];
  }
};

// sum_k a[k sa] * b[k sb], k < p, from 0.0 in index order, the operands requested eight products ahead of the additions (a lane walking
// such a chain alone otherwise pays a memory round trip per term)
__device__ inline double dc_dot(const double* __restrict__ a, size_t sa, const double* __restrict__ b, size_t sb, int p) {
  double s = 0.0;
  int k = 0;
  for (; k + 8 <= p; k += 8) {
    double x[8], y[8];
    MAS_UNROLL for (int q = 0; q < 8; ++q) { x[q] = a[static_cast<size_t>(k + q) * sa]; y[q] = b[static_cast<size_t>(k + q) * sb]; }
    MAS_UNROLL for (int q = 0; q < 8; ++q) { const double pr = x[q] * y[q]; s += pr; }
  }
  for (; k < p; ++k) { const double pr = a[static_cast<size_t>(k) * sa] * b[static_cast<size_t>(k) * sb]; s += pr; }
  return s;
}

// ------------------------------------------------------------------ start of a solve ------
template <class M>
__global__ void dc_init(DenseChipWs w, Dims d, const double* __restrict__ x0) {
  constexpr int NX = M::NX, NU = M::NU;
  const int n = d.N * NX, m = d.N * NU;
  const int total = d.T * m + n;
  for (int e = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x); e < total; e += static_cast<int>(gridDim.x * blockDim.x)) {
    if (e < d.T * m) w.Ug[e] = 0.0;                                       // the stacked OCP starts from zero controls (ocp.hpp:106-109)
    else { const int q = e - d.T * m; w.Xg[q] = MAS_AT(x0, q % NX, q / NX); }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    DenseChipState* S = w.S;
    S->t = d.T; S->stop = 0; S->iterations = 0; S->exit_reason = kExitMaxIter; S->rung_best = INT_MAX; S->rung_failed = 0ull; S->accepted_j = -1; S->failed = 0;
    S->regs = 0; S->improvement = 0.0; S->start_ticks = MAS_NOW_TICKS();
  }
}
// integrate_horizon on the stacked problem = every agent's own RK4 (one lane per agent)
template <class M>
__global__ void dc_rollout(DenseChipWs w, Dims d, double dt, ModelParams P) {
  constexpr int NX = M::NX, NU = M::NU;
  const int b = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (b >= d.N) return;
  const int n = d.N * NX, m = d.N * NU;
  double x[NX], u[NU], xn[NX];
  MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = w.Xg[b * NX + i];
  for (int t = 0; t < d.T; ++t) {
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = w.Ug[static_cast<size_t>(t) * m + b * NU + i];
    rk4_step<M>(x, u, dt, xn, P);
    MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = xn[i]; w.Xg[static_cast<size_t>(t + 1) * n + b * NX + i] = xn[i]; }
  }
}
// stage sums of compute_trajectory_cost on the stacked problem: gs[j][t] = sum over the agents in block order (t = T: terminal)
template <class M>
__global__ void dc_stage_sums(DenseChipWs w, Dims d, ModelParams P, const double* __restrict__ Xs, const double* __restrict__ Us, int planes, bool ignore_stop) {
  constexpr int NX = M::NX, NU = M::NU;
  if (!ignore_stop && w.S->stop) return;
  const int n = d.N * NX, m = d.N * NU, T = d.T;
  const int e = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= planes * (T + 1)) return;
  const int j = e / (T + 1), t = e % (T + 1);
  const double* X = Xs + static_cast<size_t>(j) * (T + 1) * n + static_cast<size_t>(t) * n;
  const double* U = Us + static_cast<size_t>(j) * T * m + static_cast<size_t>(t) * m;
  double cst = 0.0;
  for (int b = 0; b < d.N; ++b) {
    double x[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = X[b * NX + i];
    if (t < T) { MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = U[b * NU + i]; cst += M::stage_cost(x, u, t, P); }
    else cst += M::terminal_cost(x, P);
  }
  w.gs[static_cast<size_t>(j) * (T + 1) + t] = cst;
}
__device__ inline double dc_total(const double* gs, int T) { double c = 0.0; for (int t = 0; t < T; ++t) c += gs[t]; c += gs[T]; return c; }
static __global__ void dc_initial_merit(DenseChipWs w, int T) {
  if (blockIdx.x == 0 && threadIdx.x == 0) { const double c = dc_total(w.gs, T); w.S->current_merit = c; w.S->best_cost = c; }
}

// ------------------------------------------------------------------ one iteration ------
// elapsed_ms > max_ms at the top of the iteration (ilqr.hpp:84-90): one clock for the one stacked solve
static __global__ void dc_begin_iteration(DenseChipWs w, int T, int iter, double max_ms) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  DenseChipState* S = w.S;
  S->accepted_j = -1;
  if (S->stop) return;
  const double elapsed_ms = static_cast<double>(static_cast<long long>(static_cast<double>(MAS_NOW_TICKS() - S->start_ticks) / kTicksPerMs));
  if (max_ms < 1e300 && elapsed_ms > max_ms) {
    S->exit_reason = kExitTime; S->stop = 1;
    __hip_atomic_store(w.host_word, (static_cast<long long>(S->iterations) << 8) | 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  S->iterations = iter + 1; S->t = T; S->rung_best = INT_MAX; S->rung_failed = 0ull;
}

// cost tables and the blocks of A, B at time step S->t (terminal boundary: S->t == T, costs of x_T only)
template <class M>
__global__ void dc_tables(DenseChipWs w, Dims d, ModelParams P) {
  constexpr int NX = M::NX, NU = M::NU, kItems = 1 + 4 * NX + 4 * NU + NX + NU;
  if (w.S->stop) return;
  const int t = w.S->t, T = d.T, n = d.N * NX, m = d.N * NU;
  const bool term = t == T;
  const int e = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= d.N * kItems) return;
  const int b = e / kItems, it = e % kItems;
  double x[NX], u[NU], xp[NX], up[NU];
  MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = w.Xg[static_cast<size_t>(t) * n + b * NX + i];
  MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = term ? 0.0 : w.Ug[static_cast<size_t>(t) * m + b * NU + i];
  auto cost_of = [&](const double* xx, const double* uu) { return term ? M::terminal_cost(xx, P) : M::stage_cost(xx, uu, t, P); };
  if (it == 0) { w.c[b] = cost_of(x, u); return; }
  int q = it - 1;
  if (q < 4 * NX) {                                                      // x moved by +-1e-5 / +-1e-6
    const int k = q >> 2, sg = q & 1, six = (q >> 1) & 1;
    const double eps = six ? 1e-6 : 1e-5;
    perturb1<NX>(x, k, sg ? -eps : +eps, xp);
    (six ? w.cx6 : w.cx5)[(static_cast<size_t>(b) * NX + k) * 2 + sg] = cost_of(xp, u);
    return;
  }
  q -= 4 * NX;
  if (q < 4 * NU) {
    if (term) return;
    const int k = q >> 2, sg = q & 1, six = (q >> 1) & 1;
    const double eps = six ? 1e-6 : 1e-5;
    perturb1<NU>(u, k, sg ? -eps : +eps, up);
    (six ? w.cu6 : w.cu5)[(static_cast<size_t>(b) * NU + k) * 2 + sg] = cost_of(x, up);
    return;
  }
  if (term) return;
  q -= 4 * NU;
  const double eps = 1e-6;
  double fp[NX], fm[NX];
  if (q < NX) {                                                          // column q of the agent's block of A (finite_differences.hpp:53-71)
    perturb1<NX>(x, q, +eps, xp); M::dynamics(xp, u, fp, P);
    perturb1<NX>(x, q, -eps, xp); M::dynamics(xp, u, fm, P);
    MAS_UNROLL for (int r = 0; r < NX; ++r) w.Ab[(static_cast<size_t>(b) * NX + q) * NX + r] = (fp[r] - fm[r]) / (2 * eps);
  } else {                                                               // column of B (:73-92)
    q -= NX;
    perturb1<NU>(u, q, +eps, up); M::dynamics(x, up, fp, P);
    perturb1<NU>(u, q, -eps, up); M::dynamics(x, up, fm, P);
    MAS_UNROLL for (int r = 0; r < NX; ++r) w.Bb[(static_cast<size_t>(b) * NU + q) * NX + r] = (fp[r] - fm[r]) / (2 * eps);
  }
}
// pre[b] = c_0 + .. + c_{b-1}: the chain of the summed cost itself, by ONE wavefront -- 64 costs per coalesced load, added up lane by
// lane through v_readlane (a thread walking the array alone pays a memory round trip per addition)
static __global__ void __launch_bounds__(64) dc_prefix(DenseChipWs w, int N) {
  if (w.S->stop || blockIdx.x != 0) return;
  const int lane = static_cast<int>(threadIdx.x);
  double a = 0.0;
  for (int base = 0; base < N; base += 64) {
    const double v = (base + lane < N) ? w.c[base + lane] : 0.0;
    double mine = 0.0;
    MAS_UNROLL for (int l = 0; l < 64; ++l) {
      if (lane == l) mine = a;
      const int lo = __builtin_amdgcn_readlane(static_cast<int>(__double_as_longlong(v) & 0xffffffffll), l), hi = __builtin_amdgcn_readlane(static_cast<int>(__double_as_longlong(v) >> 32), l);
      const double vl = __longlong_as_double((static_cast<long long>(hi) << 32) | static_cast<unsigned int>(lo));
      if (base + l < N) a += vl;
    }
    if (base + lane < N) w.pre[base + lane] = mine;
  }
  if (lane == 0) w.pre[N] = a;
}

// every finite difference of the SUMMED cost at time step S->t, one lane per output entry:
//   stage:    l_x (n), l_u (m), l_xx (n^2), l_uu (m^2), l_ux (m n)        finite_differences.hpp:110-210, 263-287
//   terminal: v_x (n), v_xx (n^2)                                           :212-261
template <class M>
__global__ void __launch_bounds__(256) dc_fd(DenseChipWs w, Dims d, ModelParams P) {
  constexpr int NX = M::NX, NU = M::NU;
  extern __shared__ double dc_lds[];
  if (w.S->stop) return;
  const int t = w.S->t, T = d.T, N = d.N, n = N * NX, m = N * NU;
  const bool term = t == T;
  DenseTables<NX, NU> tb;
  tb.load(dc_lds, w, N);
  const long long n2 = static_cast<long long>(n) * n, m2 = static_cast<long long>(m) * m, mn = static_cast<long long>(m) * n;
  const long long total = term ? n + n2 : n + m + n2 + m2 + mn;
  const double* xs = w.Xg + static_cast<size_t>(t) * n; const double* us = w.Ug + static_cast<size_t>(term ? 0 : t) * m;
  double* gx = term ? w.vx : w.lx; double* Hx = term ? w.V : w.lxx;
  auto cost_of = [&](const double* xx, const double* uu) { return term ? M::terminal_cost(xx, P) : M::stage_cost(xx, uu, t, P); };
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    long long q = e;
    if (q < n) {                                                         // gradient in the state, eps = 1e-6, no safe_eval
      const int b = static_cast<int>(q) / NX, k = static_cast<int>(q) % NX; const double eps = 1e-6;
      double f[2]; tb.template chains<2>(b, tb.cx6 + (b * NX + k) * 2, -1, nullptr, f);
      gx[q] = (f[0] - f[1]) / (2 * eps);
      continue;
    }
    q -= n;
    if (!term) {
      if (q < m) {
        const int b = static_cast<int>(q) / NU, k = static_cast<int>(q) % NU; const double eps = 1e-6;
        double f[2]; tb.template chains<2>(b, tb.cu6 + (b * NU + k) * 2, -1, nullptr, f);
        w.lu[q] = (f[0] - f[1]) / (2 * eps);
        continue;
      }
      q -= m;
    }
    if (q < n2) {                                                        // Hessian in the state, eps = 1e-5, through safe_eval; (i,j) and (j,i) separately
      const int i = static_cast<int>(q % n), j = static_cast<int>(q / n), bi = i / NX, ki = i % NX, bj = j / NX, kj = j % NX; const double eps = 1e-5;
      double h;
      if (i == j) {
        double f[2]; tb.template chains<2>(bi, tb.cx5 + (bi * NX + ki) * 2, -1, nullptr, f);
        const double fp = safe_value(f[0]), f0 = safe_value(tb.pre[N]), fm = safe_value(f[1]);
        h = (fp - 2 * f0 + fm) / (eps * eps);
      } else {
        double f[4];
        if (bi == bj) {                                                  // both coordinates inside one agent: its cost evaluated with both moved
          double x[NX], u[NU], xp[NX];
          MAS_UNROLL for (int r = 0; r < NX; ++r) x[r] = xs[bi * NX + r];
          MAS_UNROLL for (int r = 0; r < NU; ++r) u[r] = term ? 0.0 : us[bi * NU + r];
          double cb[4];
          MAS_UNROLL for (int s = 0; s < 4; ++s) { perturb2<NX>(x, ki, (s & 2) ? -eps : +eps, kj, (s & 1) ? -eps : +eps, xp); cb[s] = cost_of(xp, u); }
          tb.template chains<4>(bi, cb, -1, nullptr, f);
        } else {
          double vi[4], vj[4];
          MAS_UNROLL for (int s = 0; s < 4; ++s) { vi[s] = tb.cx5[(bi * NX + ki) * 2 + ((s >> 1) & 1)]; vj[s] = tb.cx5[(bj * NX + kj) * 2 + (s & 1)]; }
          tb.template chains<4>(bi, vi, bj, vj, f);
        }
        MAS_UNROLL for (int s = 0; s < 4; ++s) f[s] = safe_value(f[s]);
        h = (f[0] - f[1] - f[2] + f[3]) / (4 * eps * eps);
      }
      Hx[q] = h;                                                         // H(i, j) at [j n + i]
      continue;
    }
    q -= n2;
    if (q < m2) {                                                        // Hessian in the control
      const int i = static_cast<int>(q % m), j = static_cast<int>(q / m), bi = i / NU, ki = i % NU, bj = j / NU, kj = j % NU; const double eps = 1e-5;
      double h;
      if (i == j) {
        double f[2]; tb.template chains<2>(bi, tb.cu5 + (bi * NU + ki) * 2, -1, nullptr, f);
        const double fp = safe_value(f[0]), f0 = safe_value(tb.pre[N]), fm = safe_value(f[1]);
        h = (fp - 2 * f0 + fm) / (eps * eps);
      } else {
        double f[4];
        if (bi == bj) {
          double x[NX], u[NU], up[NU];
          MAS_UNROLL for (int r = 0; r < NX; ++r) x[r] = xs[bi * NX + r];
          MAS_UNROLL for (int r = 0; r < NU; ++r) u[r] = us[bi * NU + r];
          double cb[4];
          MAS_UNROLL for (int s = 0; s < 4; ++s) { perturb2<NU>(u, ki, (s & 2) ? -eps : +eps, kj, (s & 1) ? -eps : +eps, up); cb[s] = M::stage_cost(x, up, t, P); }
          tb.template chains<4>(bi, cb, -1, nullptr, f);
        } else {
          double vi[4], vj[4];
          MAS_UNROLL for (int s = 0; s < 4; ++s) { vi[s] = tb.cu5[(bi * NU + ki) * 2 + ((s >> 1) & 1)]; vj[s] = tb.cu5[(bj * NU + kj) * 2 + (s & 1)]; }
          tb.template chains<4>(bi, vi, bj, vj, f);
        }
        MAS_UNROLL for (int s = 0; s < 4; ++s) f[s] = safe_value(f[s]);
        h = (f[0] - f[1] - f[2] + f[3]) / (4 * eps * eps);
      }
      w.luu[q] = h;
      continue;
    }
    q -= m2;
    {                                                                    // cross term m x n, eps = 1e-6: f_pp (x+,u+)  f_pm (x-,u+)  f_mp (x+,u-)  f_mm (x-,u-)
      const int i = static_cast<int>(q % m), j = static_cast<int>(q / m), bi = i / NU, ki = i % NU, bj = j / NX, kj = j % NX; const double eps = 1e-6;
      double f[4];
      if (bi == bj) {
        double x[NX], u[NU], xp[NX], up[NU];
        MAS_UNROLL for (int r = 0; r < NX; ++r) x[r] = xs[bi * NX + r];
        MAS_UNROLL for (int r = 0; r < NU; ++r) u[r] = us[bi * NU + r];
        double cb[4];
        MAS_UNROLL for (int s = 0; s < 4; ++s) { perturb1<NX>(x, kj, (s & 1) ? -eps : +eps, xp); perturb1<NU>(u, ki, (s & 2) ? -eps : +eps, up); cb[s] = M::stage_cost(xp, up, t, P); }
        tb.template chains<4>(bi, cb, -1, nullptr, f);
      } else {
        double vx_[4], vu_[4];
        MAS_UNROLL for (int s = 0; s < 4; ++s) { vx_[s] = tb.cx6[(bj * NX + kj) * 2 + (s & 1)]; vu_[s] = tb.cu6[(bi * NU + ki) * 2 + ((s >> 1) & 1)]; }
        tb.template chains<4>(bj, vx_, bi, vu_, f);
      }
      MAS_UNROLL for (int s = 0; s < 4; ++s) f[s] = safe_value(f[s]);
      w.lux[q] = (f[0] - f[1] - f[2] + f[3]) / (4 * eps * eps);          // H(i, j) at [j m + i]
    }
  }
}

// v_xx = 0.5 (v_xx + v_xx^T) evaluated in place in column-major order: an entry below the diagonal averages the old
// values, an entry on or above it sees its already rewritten mirror -- two launches give the serial loop's numbers.
// The second one ends a step of the backward pass: the rungs the step needed are counted and S->t moves on.
static __global__ void dc_symmetrise(DenseChipWs w, int n, int phase) {
  if (w.S->stop) return;
  double* V = w.V;
  const long long n2 = static_cast<long long>(n) * n;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < n2; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int i = static_cast<int>(e % n), j = static_cast<int>(e / n);
    if ((phase == 0) == (i > j)) V[e] = 0.5 * (V[e] + V[static_cast<size_t>(i) * n + j]);
  }
}
static __global__ void dc_step_end(DenseChipWs w, int T) {
  if (blockIdx.x != 0 || threadIdx.x != 0 || w.S->stop) return;
  DenseChipState* S = w.S;
  if (S->t < T) S->regs += S->rung_best;
  S->t -= 1; S->rung_best = INT_MAX; S->rung_failed = 0ull;
}

// assemble_q (ilqr.hpp:115-119) with the block structure of A and B.  Launch 1: q_x, q_u, A^T V, B^T V; launch 2: q_xx, q_ux, q_uu.
template <int NX, int NU>
__global__ void dc_assemble1(DenseChipWs w, int N) {
  if (w.S->stop) return;
  const int n = N * NX, m = N * NU;
  const long long n2 = static_cast<long long>(n) * n, mn = static_cast<long long>(m) * n, total = n + m + n2 + mn;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    long long q = e;
    if (q < n) { const int b = static_cast<int>(q) / NX, k = static_cast<int>(q) % NX; const double* a = w.Ab + (static_cast<size_t>(b) * NX + k) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += a[r] * w.vx[b * NX + r];
      w.qx[q] = w.lx[q] + s; continue; }
    q -= n;
    if (q < m) { const int b = static_cast<int>(q) / NU, k = static_cast<int>(q) % NU; const double* bc = w.Bb + (static_cast<size_t>(b) * NU + k) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += bc[r] * w.vx[b * NX + r];
      w.qu[q] = w.lu[q] + s; continue; }
    q -= m;
    if (q < n2) { const int i = static_cast<int>(q % n), j = static_cast<int>(q / n), b = i / NX, k = i % NX; const double* a = w.Ab + (static_cast<size_t>(b) * NX + k) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += a[r] * w.V[static_cast<size_t>(j) * n + b * NX + r];
      w.tmp_xx[q] = s; continue; }
    q -= n2;
    { const int i = static_cast<int>(q % m), j = static_cast<int>(q / m), b = i / NU, k = i % NU; const double* bc = w.Bb + (static_cast<size_t>(b) * NU + k) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += bc[r] * w.V[static_cast<size_t>(j) * n + b * NX + r];
      w.tmp_ux[q] = s; }
  }
}
template <int NX, int NU>
__global__ void dc_assemble2(DenseChipWs w, int N) {
  if (w.S->stop) return;
  const int n = N * NX, m = N * NU;
  const long long n2 = static_cast<long long>(n) * n, mn = static_cast<long long>(m) * n, m2 = static_cast<long long>(m) * m, total = n2 + mn + m2;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    long long q = e;
    if (q < n2) { const int i = static_cast<int>(q % n), j = static_cast<int>(q / n), bj = j / NX, kj = j % NX; const double* a = w.Ab + (static_cast<size_t>(bj) * NX + kj) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += w.tmp_xx[(static_cast<size_t>(bj) * NX + r) * n + i] * a[r];
      w.qxx[q] = w.lxx[q] + s; continue; }
    q -= n2;
    if (q < mn) { const int i = static_cast<int>(q % m), j = static_cast<int>(q / m), bj = j / NX, kj = j % NX; const double* a = w.Ab + (static_cast<size_t>(bj) * NX + kj) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += w.tmp_ux[(static_cast<size_t>(bj) * NX + r) * m + i] * a[r];
      w.qux[q] = w.lux[q] + s; continue; }
    q -= mn;
    { const int i = static_cast<int>(q % m), j = static_cast<int>(q / m), bj = j / NU, kj = j % NU; const double* bc = w.Bb + (static_cast<size_t>(bj) * NU + kj) * NX;
      double s = 0.0; MAS_UNROLL for (int r = 0; r < NX; ++r) s += w.tmp_ux[(static_cast<size_t>(bj) * NX + r) * m + i] * bc[r];
      w.quu[q] = w.luu[q] + s; }
  }
}

// LLT + x10 ladder (ilqr.hpp:172-182), the rungs SIDE BY SIDE: workgroup g tries rungs g, g + 16, ... -- rung r is q_uu with
// reg = 1e-6, 1e-5, ... added to the diagonal r times, cumulatively, like the loop -- and the lowest success (atomicMin) is the rung the
// sequential ladder stops at.  A rung is dropped once a lower one has succeeded.  Before a workgroup starts a rung of a LATER batch
// (>= 16) it looks at the rungs below it: while one of them is still being tried the later rung may turn out unnecessary, so it polls
// (bounded: the sixteen workgroups of this launch are resident together on any device this library runs on, but nothing depends on
// it -- after the bound it simply tries the rung).
// One LLT (Eigen's unblocked lower LLT, llt_compute_dyn), left-looking by rows in panels of eight columns: thread r owns row r and forms
//   a(r,k) = M(r,k) + sum_{j<k} (-L(k,j)) L(r,j),  j ascending,
// from ITS OWN earlier results L(r,j) (one load serves the eight columns of the panel) and the panel's rows of the factor, staged in
// LDS from a row-major copy; inside the panel the columns follow one another (pivot, division, the later columns' next term), two
// barriers a column.  Every thread keeps the running sum of squares of its row for its own pivot.
constexpr int kDcPanel = 8;
static __global__ void __launch_bounds__(512) dc_ladder(DenseChipWs w, int m, int first_rung) {
  extern __shared__ double dc_lds[];
  __shared__ double sh_x[2], tri[kDcPanel * kDcPanel]; __shared__ int sh_fail, sh_go;
  DenseChipState* S = w.S;
  if (S->stop) return;
  unsigned long long* rung_failed = &S->rung_failed;
  double* lrows = dc_lds;                                                // [j][8]: -L(k0 + q, j), j < k0
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const size_t m2 = static_cast<size_t>(m) * m;
  double* R = w.Rr + static_cast<size_t>(blockIdx.x) * m2; double* LT = w.Lt + static_cast<size_t>(blockIdx.x) * m2;
  for (int rung = first_rung + static_cast<int>(blockIdx.x); rung <= kRegLadderMax; rung += kDenseChipRungs) {
    if (tid == 0) {
      int go = 1;
      const unsigned long long below = (rung >= 64) ? ~0ull : ((1ull << rung) - 1ull);
      for (int spins = 0;; ++spins) {
        if (__hip_atomic_load(&S->rung_best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < rung) { go = 0; break; }
        if (rung < first_rung + kDenseChipRungs || rung >= 64 || spins > (1 << 16)) break;
        if ((__hip_atomic_load(rung_failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & below) == below) break;     // every lower rung has failed: this one is needed
        __builtin_amdgcn_s_sleep(8);
      }
      sh_go = go; sh_fail = 0;
    }
    __syncthreads();
    if (!sh_go) return;
    for (size_t e = tid; e < m2; e += nt) {
      double v = w.quu[e];
      if (e % (static_cast<size_t>(m) + 1) == 0) { double reg = 1e-6; for (int q = 0; q < rung; ++q) { v += reg; reg *= 10.0; } }
      R[e] = v;
    }
    __syncthreads();
    double sq[2] = {0.0, 0.0};                                           // running sum of squares of the thread's rows tid, tid + nt
    bool failed = false;
    for (int k0 = 0; k0 < m && !failed; k0 += kDcPanel) {
      const int pw = min(kDcPanel, m - k0);
      for (int e = tid; e < k0 * pw; e += nt) { const int q = e / k0, j = e % k0; lrows[j * kDcPanel + q] = -LT[static_cast<size_t>(k0 + q) * m + j]; }
      __syncthreads();
      double a[2][kDcPanel];
      MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
        const int r = tid + slot * nt;
        MAS_UNROLL for (int q = 0; q < kDcPanel; ++q) a[slot][q] = 0.0;
        if (r > k0 && r < m) {
          MAS_UNROLL for (int q = 0; q < kDcPanel; ++q) a[slot][q] = (q < pw) ? R[static_cast<size_t>(k0 + q) * m + r] : 0.0;
          const double* Rrow = R + r;
          int j = 0;
          for (; j + 4 <= k0; j += 4) {
            double l[4];
            MAS_UNROLL for (int u = 0; u < 4; ++u) l[u] = Rrow[static_cast<size_t>(j + u) * m];
            MAS_UNROLL for (int u = 0; u < 4; ++u) { MAS_UNROLL for (int q = 0; q < kDcPanel; ++q) { const double pr = lrows[(j + u) * kDcPanel + q] * l[u]; a[slot][q] += pr; } }
          }
          for (; j < k0; ++j) { const double l = Rrow[static_cast<size_t>(j) * m]; MAS_UNROLL for (int q = 0; q < kDcPanel; ++q) { const double pr = lrows[j * kDcPanel + q] * l; a[slot][q] += pr; } }
        }
      }
      MAS_UNROLL for (int q = 0; q < kDcPanel; ++q) {
        const int k = k0 + q;
        if (k < m && !failed) {                                           // (uniform)
          if (tid == k % nt) {
            double x = R[static_cast<size_t>(k) * m + k];
            if (k > 0) x = x - ((k / nt) ? sq[1] : sq[0]);
            if (x <= 0.0) sh_fail = 1;                                    // (NaN pivots pass, like Eigen)
            else { x = sqrt(x); R[static_cast<size_t>(k) * m + k] = x; sh_x[q & 1] = x; }
          }
          __syncthreads();
          if (sh_fail) { failed = true; }
          if (!failed) {
            const double x = sh_x[q & 1];
            double v[2] = {0.0, 0.0};
            MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
              const int r = tid + slot * nt;
              if (r > k && r < m) {
                v[slot] = a[slot][q] / x;
                R[static_cast<size_t>(k) * m + r] = v[slot]; LT[static_cast<size_t>(r) * m + k] = v[slot];
                sq[slot] += v[slot] * v[slot];
                if (r < k0 + kDcPanel) tri[(r - k0) * kDcPanel + q] = -v[slot];
              }
            }
            __syncthreads();
            MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
              const int r = tid + slot * nt;
              if (r > k && r < m) {
                MAS_UNROLL for (int q2 = q + 1; q2 < kDcPanel; ++q2) { if (k0 + q2 < r) { const double pr = tri[q2 * kDcPanel + q] * v[slot]; a[slot][q2] += pr; } }
              }
            }
          }
        }
      }
    }
    __syncthreads();
    if (!failed) { if (tid == 0) atomicMin(&S->rung_best, rung); return; }
    if (tid == 0 && rung < 64) __hip_atomic_fetch_or(rung_failed, 1ull << rung, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
  }
}

// ---- the same sixteen factorisations spread over the CHIP: blocked by panels of kDcBlock columns ----
// For an entry (r, k) of the factor the reference forms  a = M(r,k) + sum_{j<k} (-L(k,j)) L(r,j)  with j ascending.  The sum is taken panel by
// panel: once panel p (columns pB .. pB+B-1) is final, EVERY entry to its right adds that panel's B terms, in j order, to its running value
// (dc_llt_update: 64 x 64 tiles of the trailing triangle of all sixteen rungs, 4 x 4 entries per thread) -- the terms reach every entry
// in the order of the serial loop -- and only the panel itself (dc_llt_panel: one workgroup per rung, thread = row, pivot / division /
// next column's term, two barriers a column) is a chain.  m / B panel steps instead of m column steps (B = 32; 16 beyond 512 rows: the panel
// lives in LDS).
// The running sum of squares of row r (for its own pivot: x = M(r,r) - s, s from 0.0 in j order) is kept in SD by the panel kernel.
constexpr int kDcBlock = 32;
static __global__ void dc_llt_init(DenseChipWs w, int m) {
  if (w.S->stop) return;
  const size_t m2 = static_cast<size_t>(m) * m, total = m2 * kDenseChipRungs;
  for (size_t e = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; e < total; e += static_cast<size_t>(gridDim.x) * blockDim.x) {
    const int rung = static_cast<int>(e / m2); const size_t q = e % m2;
    double v = w.quu[q];
    if (q % (static_cast<size_t>(m) + 1) == 0) { double reg = 1e-6; for (int i = 0; i < rung; ++i) { v += reg; reg *= 10.0; } }
    w.Rr[e] = v;
    if (q < static_cast<size_t>(m)) w.SD[static_cast<size_t>(rung) * m + q] = 0.0;
    if (q == 0) w.rung_state[rung] = 0;
  }
}
static __global__ void __launch_bounds__(512) dc_llt_panel(DenseChipWs w, int m, int p, int bw) {
  extern __shared__ double dc_lds[];                                    // ap[bw][m]: the panel's columns (entry (r, k0 + q) at ap[q m + r])
  __shared__ double sh_x[2], tri[kDcBlock * kDcBlock], diag[kDcBlock]; __shared__ int sh_fail;
  DenseChipState* S = w.S;
  const int rung = static_cast<int>(blockIdx.x);
  if (S->stop || w.rung_state[rung] != 0) return;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  double* R = w.Rr + static_cast<size_t>(rung) * m * m; double* SD = w.SD + static_cast<size_t>(rung) * m;
  double* ap = dc_lds;
  const int k0 = p * bw, pw = min(bw, m - k0);
  double sq[2];
  MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
    const int r = tid + slot * nt;
    sq[slot] = (r < m) ? SD[r] : 0.0;
    if (r < m) for (int q = 0; q < pw; ++q) ap[q * m + r] = (r > k0 + q) ? R[static_cast<size_t>(k0 + q) * m + r] : 0.0;
  }
  if (tid < pw) diag[tid] = R[static_cast<size_t>(k0 + tid) * m + k0 + tid];      // M(k,k) of the panel's columns, fetched ahead of the chain
  if (tid == 0) sh_fail = 0;
  __syncthreads();
  // The column loop is a LOOP, the panel lives in LDS: unrolled over registers (the first form) its 32 x 31 / 2 update statements were 50 KB of
  // code that every launch fetched cold -- 3 us a column, most of it instruction fetch.  Nothing inside the loop touches global memory.
  bool failed = false;
#pragma nounroll
  for (int q = 0; q < pw; ++q) {
    const int k = k0 + q;
    if (tid == k % nt) {
      double x = diag[q];
      if (k > 0) x = x - ((k / nt) ? sq[1] : sq[0]);
      if (x <= 0.0) sh_fail = 1;                                        // (NaN pivots pass, like Eigen)
      else { x = sqrt(x); diag[q] = x; sh_x[q & 1] = x; }
    }
    __syncthreads();
    if (sh_fail) { failed = true; break; }                              // (uniform)
    const double x = sh_x[q & 1];
    double v[2] = {0.0, 0.0};
    MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
      const int r = tid + slot * nt;
      if (r > k && r < m) {
        v[slot] = ap[q * m + r] / x;
        ap[q * m + r] = v[slot];
        sq[slot] += v[slot] * v[slot];
        if (r < k0 + pw) tri[q * kDcBlock + (r - k0)] = -v[slot];       // [column][row of the panel]: -L(k0 + q2, k)
      }
    }
    __syncthreads();
    MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
      const int r = tid + slot * nt;
      if (r > k && r < m) {
        // every later column of the panel takes its next term (-L(k0 + q2, k)) L(r, k); entries on or above the diagonal (k0 + q2 >= r) are
        // never read again, so nothing is tested here
        // (eight columns at a time: their operands requested together, then the eight results stored -- an LDS round trip per chunk, not per column)
        int q2 = q + 1;
        for (; q2 + 8 <= pw; q2 += 8) {
          double tq[8], aq[8];
          MAS_UNROLL for (int u = 0; u < 8; ++u) { tq[u] = tri[q * kDcBlock + q2 + u]; aq[u] = ap[(q2 + u) * m + r]; }
          MAS_UNROLL for (int u = 0; u < 8; ++u) { const double pr = tq[u] * v[slot]; ap[(q2 + u) * m + r] = aq[u] + pr; }
        }
        for (; q2 < pw; ++q2) { const double pr = tri[q * kDcBlock + q2] * v[slot]; ap[q2 * m + r] += pr; }
      }
    }
  }
  __syncthreads();
  if (failed) { if (tid == 0) { w.rung_state[rung] = 1; __hip_atomic_fetch_or(&S->rung_failed, 1ull << rung, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return; }
  MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
    const int r = tid + slot * nt;
    if (r < m) {
      SD[r] = sq[slot];
      for (int q = 0; q < pw; ++q) { if (r > k0 + q) R[static_cast<size_t>(k0 + q) * m + r] = ap[q * m + r]; }
    }
  }
  if (tid < pw) R[static_cast<size_t>(k0 + tid) * m + k0 + tid] = diag[tid];
  if (k0 + pw >= m && tid == 0) { w.rung_state[rung] = 2; atomicMin(&S->rung_best, rung); }
}
static __global__ void __launch_bounds__(256) dc_llt_update(DenseChipWs w, int m, int p, int bw) {
  __shared__ double Lr[kDcBlock][64 + 4], Lk[kDcBlock][64 + 4];
  const int rung = static_cast<int>(blockIdx.y);
  if (w.S->stop || w.rung_state[rung] != 0) return;
  double* R = w.Rr + static_cast<size_t>(rung) * m * m;
  const int k0p = p * bw, c0 = k0p + bw;                                 // the panel's first column; first column of the trailing part
  if (c0 >= m) return;
  const int nb = (m - c0 + 63) / 64;                                    // 64-wide blocks of the trailing part; tiles (ti >= tk)
  int tile = static_cast<int>(blockIdx.x), ti = 0;
  while (tile > ti) { tile -= ti + 1; ++ti; }                           // tile number -> (ti, tk) of the lower triangle of blocks
  const int tk = tile;
  if (ti >= nb) return;
  const int r0 = c0 + 64 * ti, kk0 = c0 + 64 * tk;
  const int tid = static_cast<int>(threadIdx.x), tx = tid & 15, ty = tid >> 4;
  for (int e = tid; e < bw * 64; e += 256) {
    const int j = e >> 6, c = e & 63;
    Lr[j][c] = (r0 + c < m) ? R[static_cast<size_t>(k0p + j) * m + r0 + c] : 0.0;
    Lk[j][c] = (kk0 + c < m) ? -R[static_cast<size_t>(k0p + j) * m + kk0 + c] : 0.0;
  }
  __syncthreads();
  double acc[4][4];
  MAS_UNROLL for (int b = 0; b < 4; ++b) { MAS_UNROLL for (int a = 0; a < 4; ++a) { const int r = r0 + tx * 4 + a, k = kk0 + ty * 4 + b; acc[a][b] = (r > k && r < m) ? R[static_cast<size_t>(k) * m + r] : 0.0; } }
  for (int j = 0; j < bw; ++j) {
    double lr[4], lk[4];
    MAS_UNROLL for (int a = 0; a < 4; ++a) lr[a] = Lr[j][tx * 4 + a];
    MAS_UNROLL for (int b = 0; b < 4; ++b) lk[b] = Lk[j][ty * 4 + b];
    MAS_UNROLL for (int b = 0; b < 4; ++b) { MAS_UNROLL for (int a = 0; a < 4; ++a) { const double pr = lk[b] * lr[a]; acc[a][b] += pr; } }
  }
  MAS_UNROLL for (int b = 0; b < 4; ++b) { MAS_UNROLL for (int a = 0; a < 4; ++a) { const int r = r0 + tx * 4 + a, k = kk0 + ty * 4 + b; if (r > k && r < m) R[static_cast<size_t>(k) * m + r] = acc[a][b]; } }
}

// Quu_reg^-1 = LLT::solve(I) (ilqr.hpp:183; llt_solve_identity): 2^cshift identity columns per workgroup.
// Forward substitution right-looking with the rows dealt over the threads: thread r keeps ITS row of the columns' vectors in registers; at
// step i the owner of row i scales it (o[i] *= 1 / L(i,i)) and publishes it through LDS, every row below subtracts b L(r,i) -- the
// subtractions reach every entry in the order i = 0, 1, ... of the serial loop; steps before the workgroup's first column subtract b = 0
// and are skipped.  One barrier a step (the published row alternates between two LDS rows), L(r,i) requested a step ahead.
// Backward substitution: o[i] = (o[i] - sum_{r>i} L(r,i) o[r]) / L(i,i) with the sum from 0.0 over r ascending -- its FIRST term is the row
// that has only just become final, so the sum is a serial chain by construction (m^2 / 2 dependent additions per column).  One lane per
// column walks it; the products of step i - 1 (all but that first term) are formed by the OTHER wavefronts while the chain of step i runs.
// workgroup barrier for data exchanged through LDS only: waits for this wavefront's LDS operations, not for its vector memory -- __syncthreads()
// is also a fence for global memory, i.e. it drains every load a step has requested AHEAD for the next one (a memory round trip per barrier)
__device__ __forceinline__ void dc_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int kDcSolveMaxCols = 16, kDcSolvePasses = 12;
static __global__ void __launch_bounds__(512) dc_solve(DenseChipWs w, int m, int cshift) {
  extern __shared__ double dc_lds[];
  DenseChipState* S = w.S;
  if (S->stop) return;
  const int best = S->rung_best;
  if (best == INT_MAX) { if (blockIdx.x == 0 && threadIdx.x == 0) S->failed = 1; return; }
  const double* R = w.Rr + static_cast<size_t>(best % kDenseChipRungs) * m * m;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  const int cols = 1 << cshift, ld = cols + 1;
  const int c0 = static_cast<int>(blockIdx.x) * cols, nc = min(cols, m - c0);
  if (nc <= 0) return;
  double* o = dc_lds;                                                   // [m][ld] the columns' vectors
  // products of the backward substitution, two steps: [column][row], columns pl doubles apart -- a column's terms are CONTIGUOUS, so the lane
  // that adds them up reads two per LDS instruction at immediate offsets (the chain is a lone wavefront: it is bound by instructions per
  // term, measured 5 -> 38 cycles per term with the terms a row apart); pl = m + 2 keeps the 8-16 columns of a read on distinct banks
  const int pl = m + 2;
  double* prA = o + static_cast<size_t>(m) * ld; double* prB = prA + static_cast<size_t>(cols) * pl;
  double* ainv = prB + static_cast<size_t>(cols) * pl;                  // [m] 1 / L(i,i)
  double* brow = ainv + m;                                              // [2][cols] the scaled row of the current step
  for (int i = tid; i < m; i += nt) ainv[i] = 1.0 / R[static_cast<size_t>(i) * m + i];
  double acc[2][kDcSolveMaxCols], lcur[2], lnext[2] = {0.0, 0.0};
  MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
    const int r = tid + slot * nt;
    MAS_UNROLL for (int c = 0; c < kDcSolveMaxCols; ++c) acc[slot][c] = (r == c0 + c && c < nc) ? 1.0 : 0.0;
    lcur[slot] = (r > c0 && r < m) ? R[static_cast<size_t>(c0) * m + r] : 0.0;
  }
  __syncthreads();
  for (int i = c0; i < m; ++i) {
    MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
      const int r = tid + slot * nt;
      if (r == i) { const double a = ainv[i]; MAS_UNROLL for (int c = 0; c < kDcSolveMaxCols; ++c) { if (c < nc) { const double v = acc[slot][c] * a; acc[slot][c] = v; brow[(i & 1) * cols + c] = v; } } }
      lnext[slot] = (i + 1 < m && r > i + 1 && r < m) ? R[static_cast<size_t>(i + 1) * m + r] : 0.0;
    }
    dc_lds_barrier();
    MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
      const int r = tid + slot * nt;
      if (r > i && r < m) { MAS_UNROLL for (int c = 0; c < kDcSolveMaxCols; ++c) { if (c < nc) { const double pr = brow[(i & 1) * cols + c] * lcur[slot]; acc[slot][c] -= pr; } } }
      lcur[slot] = lnext[slot];
    }
  }
  MAS_UNROLL for (int slot = 0; slot < 2; ++slot) {
    const int r = tid + slot * nt;
    if (r < m) { MAS_UNROLL for (int c = 0; c < kDcSolveMaxCols; ++c) { if (c < nc) o[r * ld + c] = acc[slot][c]; } }
  }
  dc_lds_barrier();
  const bool split = nt > 64;                                           // wavefront 0 walks the chains, the others form the next step's products
  const int lanes = split ? nt - 64 : nt, me = split ? tid - 64 : tid;
  const int rpp = max(1, lanes >> cshift), passes = (m + rpp - 1) / rpp, my_r = me >> cshift, my_c = me & (cols - 1);      // rows per pass of the product lanes
  double lreg[kDcSolvePasses];                                          // L(r, i - 1) of this lane's rows for the products formed in step i
  MAS_UNROLL for (int k2 = 0; k2 < kDcSolvePasses; ++k2) { const int r = my_r + k2 * rpp; lreg[k2] = (m >= 2 && k2 < passes && r < m && me >= 0) ? R[static_cast<size_t>(m - 2) * m + r] : 0.0; }
  double lfirst = 0.0;                                                  // L(i+1, i): the factor of the chain's first term, requested a step ahead
  for (int i = m - 1; i >= 0; --i) {
    const double* cur = (i & 1) ? prA : prB; double* nxt = (i & 1) ? prB : prA;
    const double lf = lfirst;                                           // (of this step)
    if (i >= 1) lfirst = R[static_cast<size_t>(i - 1) * m + i];          // (of the next one)
    if (tid < nc) {
      double b = 0.0;
      if (i + 1 < m) { const double p = lf * o[(i + 1) * ld + tid]; b += p; }
      const double* col = cur + static_cast<size_t>(tid) * pl;
      int r = i + 2;
      if ((m & 1) == 0) {                                               // (16-byte aligned pairs)
        if ((r & 1) && r < m) { b += col[r]; ++r; }
        for (; r + 32 <= m; r += 32) {                                  // 32 terms requested at once (two register sets, nothing is copied), then added in order
          double2 ta[8], tb[8];
          MAS_UNROLL for (int q = 0; q < 8; ++q) ta[q] = *reinterpret_cast<const double2*>(col + r + 2 * q);
          MAS_UNROLL for (int q = 0; q < 8; ++q) tb[q] = *reinterpret_cast<const double2*>(col + r + 16 + 2 * q);
          MAS_UNROLL for (int q = 0; q < 8; ++q) { b += ta[q].x; b += ta[q].y; }
          MAS_UNROLL for (int q = 0; q < 8; ++q) { b += tb[q].x; b += tb[q].y; }
        }
        for (; r + 8 <= m; r += 8) {
          double2 t4[4];
          MAS_UNROLL for (int q = 0; q < 4; ++q) t4[q] = *reinterpret_cast<const double2*>(col + r + 2 * q);
          MAS_UNROLL for (int q = 0; q < 4; ++q) { b += t4[q].x; b += t4[q].y; }
        }
      }
      for (; r < m; ++r) b += col[r];
      o[i * ld + tid] = (o[i * ld + tid] - b) * ainv[i];
    }
    if (i >= 1 && (!split || tid >= 64)) {                              // products of step i - 1: rows r >= i + 1 (final), L(r, i - 1)
      if (passes <= kDcSolvePasses) {                                   // every lane owns fixed rows; their factor entries were requested a step ago
        MAS_UNROLL for (int k2 = 0; k2 < kDcSolvePasses; ++k2) {
          const int r = my_r + k2 * rpp;
          if (k2 < passes && r >= i + 1 && r < m && my_c < nc) nxt[static_cast<size_t>(my_c) * pl + r] = lreg[k2] * o[r * ld + my_c];
          if (k2 < passes && i >= 2 && r >= i && r < m) lreg[k2] = R[static_cast<size_t>(i - 2) * m + r];      // for the products of the next step
        }
      } else {
        const double* Lp = R + static_cast<size_t>(i - 1) * m;
        for (int e = me; e < ((m - 1 - i) << cshift); e += lanes) {
          const int r = i + 1 + (e >> cshift), c = e & (cols - 1);
          if (c < nc) nxt[static_cast<size_t>(c) * pl + r] = Lp[r] * o[r * ld + c];
        }
      }
    }
    dc_lds_barrier();
  }
  __syncthreads();
  for (int e = tid; e < m * nc; e += nt) { const int c = e / m, r = e % m; w.inv[static_cast<size_t>(c0 + c) * m + r] = o[r * ld + c]; }
}

// C(mm x nn) = op(A) B, every entry  s = 0; s += a * b  over k in index order (the products rounded before they are added).
//   TA = false: A stored mm x p (mat_AB);  TA = true: A stored p x mm, op(A) = A^T (mat_AtB);  NEG: op(A) = -A (the unary minus of
//   ilqr.hpp:185-186 comes first);  B stored p x nn;  all column-major.  64 x 64 tiles of C, 4 x 4 entries per thread, k in slabs of 16.
template <bool TA, bool NEG>
__global__ void __launch_bounds__(256) dc_gemm(const DenseChipState* S, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, int mm, int p, int nn) {
  constexpr int TS = 64, KS = 16;
  __shared__ double As[KS][TS + 4], Bs[KS][TS + 4];
  if (S->stop) return;
  const int tiles_i = (mm + TS - 1) / TS, tiles_j = (nn + TS - 1) / TS;
  const int tid = static_cast<int>(threadIdx.x), tx = tid & 15, ty = tid >> 4;
  for (int tile = static_cast<int>(blockIdx.x); tile < tiles_i * tiles_j; tile += static_cast<int>(gridDim.x)) {
    const int i0 = (tile % tiles_i) * TS, j0 = (tile / tiles_i) * TS;
    double acc[4][4];
    MAS_UNROLL for (int a = 0; a < 4; ++a) { MAS_UNROLL for (int b = 0; b < 4; ++b) acc[a][b] = 0.0; }
    for (int k0 = 0; k0 < p; k0 += KS) {
      const int kmax = min(KS, p - k0);
      __syncthreads();
      if (TA) {
        MAS_UNROLL for (int q = 0; q < 4; ++q) { const int kk = tid & 15, ii = (tid >> 4) + 16 * q; const int i = i0 + ii, k = k0 + kk;
          double v = (i < mm && k < p) ? A[static_cast<size_t>(i) * p + k] : 0.0; As[kk][ii] = NEG ? -v : v; }
      } else {
        MAS_UNROLL for (int q = 0; q < 4; ++q) { const int ii = tid & 63, kk = (tid >> 6) + 4 * q; const int i = i0 + ii, k = k0 + kk;
          double v = (i < mm && k < p) ? A[static_cast<size_t>(k) * mm + i] : 0.0; As[kk][ii] = NEG ? -v : v; }
      }
      MAS_UNROLL for (int q = 0; q < 4; ++q) { const int kk = tid & 15, jj = (tid >> 4) + 16 * q; const int j = j0 + jj, k = k0 + kk;
        Bs[kk][jj] = (j < nn && k < p) ? B[static_cast<size_t>(j) * p + k] : 0.0; }
      __syncthreads();
      for (int kk = 0; kk < kmax; ++kk) {
        double av[4], bv[4];
        MAS_UNROLL for (int a = 0; a < 4; ++a) av[a] = As[kk][tx * 4 + a];
        MAS_UNROLL for (int b = 0; b < 4; ++b) bv[b] = Bs[kk][ty * 4 + b];
        MAS_UNROLL for (int a = 0; a < 4; ++a) { MAS_UNROLL for (int b = 0; b < 4; ++b) { const double pr = av[a] * bv[b]; acc[a][b] = acc[a][b] + pr; } }
      }
    }
    MAS_UNROLL for (int b = 0; b < 4; ++b) { const int j = j0 + ty * 4 + b;
      MAS_UNROLL for (int a = 0; a < 4; ++a) { const int i = i0 + tx * 4 + a; if (i < mm && j < nn) C[static_cast<size_t>(j) * mm + i] = acc[a][b]; } }
  }
}

// k = -Quu^-1 q_u (ilqr.hpp:185): one lane per control, the m-term sum in index order
static __global__ void dc_gain_k(DenseChipWs w, int m) {
  if (w.S->stop) return;
  const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= m) return;
  (w.k + static_cast<size_t>(w.S->t) * m)[i] = -dc_dot(w.inv + i, static_cast<size_t>(m), w.qu, 1, m);     // sum of (-a) b == -(sum of a b): rounding is symmetric in the sign
}
// v_x = q_x + K^T q_u + q_ux^T k + (K^T q_uu) k (ilqr.hpp:188-189), the three products one after the other as in the reference
static __global__ void dc_vx(DenseChipWs w, int n, int m) {
  if (w.S->stop) return;
  const int i = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  const double* Kt = w.K + static_cast<size_t>(w.S->t) * m * n; const double* kt = w.k + static_cast<size_t>(w.S->t) * m;
  double v = w.qx[i] + dc_dot(Kt + static_cast<size_t>(i) * m, 1, w.qu, 1, m);
  v = v + dc_dot(w.qux + static_cast<size_t>(i) * m, 1, kt, 1, m);
  w.vx[i] = v + dc_dot(w.KtQuu + i, static_cast<size_t>(n), kt, 1, m);
}
// v_xx = q_xx + K^T q_ux + q_ux^T K + (K^T q_uu) K (ilqr.hpp:190-191); q_ux^T K is the transpose of K^T q_ux entry by entry (the same
// products in the same order), so it is read from tmp_xx transposed
static __global__ void dc_vxx(DenseChipWs w, int n) {
  if (w.S->stop) return;
  const long long n2 = static_cast<long long>(n) * n;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < n2; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    const int i = static_cast<int>(e % n), j = static_cast<int>(e / n);
    double v = w.qxx[e] + w.tmp_xx[e];
    v = v + w.tmp_xx[static_cast<size_t>(i) * n + j];
    w.V[e] = v + w.V3[e];
  }
}

// the ten line-search trials (ilqr.hpp:195-217) side by side, one workgroup each: dense feedback, clamp, every agent's RK4
template <class M>
__global__ void __launch_bounds__(512) dc_trials(DenseChipWs w, Dims d, double dt, ModelParams P, bool clamp, const double* lbp, const double* ubp) {
  constexpr int NX = M::NX, NU = M::NU;
  extern __shared__ double dc_lds[];
  if (w.S->stop) return;
  const int T = d.T, N = d.N, n = N * NX, m = N * NU, j = static_cast<int>(blockIdx.x);
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x);
  double alpha = 1.0; for (int q = 0; q < j; ++q) alpha *= 0.5;
  double* Xt = w.Xt + static_cast<size_t>(j) * (T + 1) * n; double* Ut = w.Ut + static_cast<size_t>(j) * T * m;
  double* dx = dc_lds; double* xc = dx + n; double* uc = xc + n;
  double lb[NU], ub[NU];
  MAS_UNROLL for (int i = 0; i < NU; ++i) { lb[i] = clamp ? lbp[i] : 0.0; ub[i] = clamp ? ubp[i] : 0.0; }
  for (int e = tid; e < n; e += nt) { const double v = w.Xg[e]; xc[e] = v; Xt[e] = v; }       // x_trial.col(0) = initial_state
  __syncthreads();
  for (int t = 0; t < T; ++t) {
    const double* kt = w.k + static_cast<size_t>(t) * m; const double* Kt = w.K + static_cast<size_t>(t) * m * n;
    for (int e = tid; e < n; e += nt) dx[e] = xc[e] - w.Xg[static_cast<size_t>(t) * n + e];
    __syncthreads();
    for (int i = tid; i < m; i += nt) {
      const double kd = dc_dot(Kt + i, static_cast<size_t>(m), dx, 1, n);
      double v = (w.Ug[static_cast<size_t>(t) * m + i] + alpha * kt[i]) + kd;
      if (clamp) { const int q = i % NU; v = (ub[q] < v) ? ub[q] : v; v = (v < lb[q]) ? lb[q] : v; }
      uc[i] = v; Ut[static_cast<size_t>(t) * m + i] = v;
    }
    __syncthreads();
    for (int b = tid; b < N; b += nt) {
      double x[NX], u[NU], xn[NX];
      MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = xc[b * NX + i];
      MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = uc[b * NU + i];
      rk4_step<M>(x, u, dt, xn, P);
      MAS_UNROLL for (int i = 0; i < NX; ++i) { xc[b * NX + i] = xn[i]; Xt[static_cast<size_t>(t + 1) * n + b * NX + i] = xn[i]; }
    }
    __syncthreads();
  }
}
// first strict decrease of the merit, termination (ilqr.hpp:218-234, 269-271); one word for the host
static __global__ void dc_accept(DenseChipWs w, int T, double tolerance, int max_iterations) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  DenseChipState* S = w.S;
  if (S->stop) return;
  if (S->failed) { S->exit_reason = kExitMaxIter | kExitNonFinite; S->stop = 1; }
  else {
    double best_merit = S->current_merit; int sel = -1;
    for (int j = 0; j < kNumAlphas; ++j) { const double tm = dc_total(w.gs + static_cast<size_t>(j) * (T + 1), T); if (tm < best_merit) { best_merit = tm; sel = j; break; } }
    const double improvement = S->current_merit - best_merit;
    S->accepted_j = sel; S->improvement = improvement;
    if (sel >= 0) S->best_cost = best_merit;                             // cost = objective(best_x, best_u) == the accepted trial's merit
    S->current_merit = best_merit;
    if (improvement < tolerance) { S->exit_reason = kExitConverged; S->stop = 1; }
    else if (S->iterations >= max_iterations) S->stop = 1;
  }
  __hip_atomic_store(w.host_word, (static_cast<long long>(S->iterations) << 8) | (S->stop ? 1 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
static __global__ void dc_commit(DenseChipWs w, int T, int n, int m) {
  const int j = w.S->accepted_j;
  if (j < 0) return;
  const long long nx_ = static_cast<long long>(T + 1) * n, nu_ = static_cast<long long>(T) * m;
  const double* Xt = w.Xt + static_cast<size_t>(j) * nx_; const double* Ut = w.Ut + static_cast<size_t>(j) * nu_;
  for (long long e = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; e < nx_ + nu_; e += static_cast<long long>(gridDim.x) * blockDim.x) {
    if (e < nx_) w.Xg[e] = Xt[e]; else w.Ug[e - nx_] = Ut[e - nx_];
  }
}
// hand the stacked solution back to the agents (centralized.hpp:28-35): trajectories, per-agent objective, total = stacked best_cost
template <class M>
__global__ void dc_handback(DenseChipWs w, Dims d, ModelParams P, double* __restrict__ X, double* __restrict__ U, double* __restrict__ cost,
                            int* __restrict__ iters_total, int* __restrict__ exit_code, long long* __restrict__ reg_events, CentralState* st) {
  constexpr int NX = M::NX, NU = M::NU;
  const int b = static_cast<int>(blockIdx.x * blockDim.x + threadIdx.x);
  const int T = d.T, n = d.N * NX, m = d.N * NU;
  const DenseChipState* S = w.S;
  if (b < d.N) {
    double c = 0.0, x[NX], u[NU];
    for (int t = 0; t <= T; ++t) {
      MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = w.Xg[static_cast<size_t>(t) * n + b * NX + i]; MAS_AT(X, t * NX + i, b) = x[i]; }
      if (t < T) { MAS_UNROLL for (int i = 0; i < NU; ++i) { u[i] = w.Ug[static_cast<size_t>(t) * m + b * NU + i]; MAS_AT(U, t * NU + i, b) = u[i]; } c += M::stage_cost(x, u, t, P); }
      else c += M::terminal_cost(x, P);
    }
    cost[b] = c; iters_total[b] += S->iterations; exit_code[b] = S->exit_reason;
  }
  if (b == 0) {
    st->cur_total = S->best_cost; st->iterations = S->iterations; st->done = S->exit_reason == kExitConverged ? 1 : 0; st->j_sel = -1; st->improvement = 0.0;
    reg_events[0] += S->regs;
  }
}

}  // namespace mas
