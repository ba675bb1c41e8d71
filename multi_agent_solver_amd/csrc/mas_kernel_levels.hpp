// mas_kernel_levels.hpp -- open-loop rollouts evaluated by DEPENDENCY LEVELS, and the trust-region epilogue built on them (HIP build only)
#pragma once

#include <utility>

#include "mas_bodies.hpp"

namespace mas {

// An open-loop rollout x_{t+1} = integrate_rk4(x_t, u_t) with KNOWN controls (the trust-region re-roll, nash.hpp:219-226) is a chain of
// T steps of ~300-700 dependent FP64 instructions each when one lane (or one quad) walks it: 105-170 us at T = 200 for a handful of
// agents, on the critical path of every round of a chained trust-region run.  For the vehicle models it does not have to be a chain
// of whole steps.  Their dynamics are triangular (M::state_level): the RK4 increment (dt/6)(k1 + 2 k2 + 2 k3 + k4) of a component
// depends only on the control and on components of LOWER levels -- bicycle6: delta, a <- u;  v <- a;  psi <- v, delta;  X, Y <- v, psi --
// so level by level
//   * the increments of ALL time steps are evaluated side by side (one lane per time step: rk4_increment on the known components;
//     what a level does not need is dead code), and
//   * x_{t+1} = x_t + inc_t is added up serially by one lane per component: T dependent additions, not T dependent RK4 steps.
// Stage costs are evaluated side by side as well and summed serially from 0.0.  Every operation is the one rk4_step / stage_cost
// perform for that step, on the same operands (nothing is contracted; the sums run in the reference's order), so the trajectory and
// the cost are bit-identical to the serial rollout: ~10 us instead of 105-170.
//
// LDS of one workgroup (doubles):  xs[NX][T+1]  states;  inc[NX][T]  increments / squared differences / stage costs;  uc[T*NU]  controls;
// uo[T*NU]  old controls;  misc[8].
template <class M>
struct LevelsLayout {
  static constexpr int NX = M::NX, NU = M::NU, kThreads = 256;
  __host__ __device__ static size_t doubles(int T) { return static_cast<size_t>(NX) * (T + 1) + static_cast<size_t>(NX) * T + 2 * static_cast<size_t>(NU) * T + 8; }
  __host__ __device__ static size_t bytes(int T) { return doubles(T) * sizeof(double); }
};

// level LEV of the rollout in LDS: increments of its components for every t (all threads), then their serial sums (one lane per component)
template <class M, int LEV>
__device__ __forceinline__ void levels_pass(double* xs, double* inc, const double* uc, int T, double dt, const ModelParams& P, int tid, int nthreads) {
  constexpr int NX = M::NX, NU = M::NU;
  for (int t = tid; t < T; t += nthreads) {
    double x[NX], u[NU], in[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = (M::state_level(i) < LEV) ? xs[i * (T + 1) + t] : 0.0;     // (components not known yet: operands of dead code)
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = uc[t * NU + i];
    rk4_increment<M>(x, u, dt, in, P);
    MAS_UNROLL for (int i = 0; i < NX; ++i) { if (M::state_level(i) == LEV) inc[i * T + t] = in[i]; }
  }
  __syncthreads();
  if (tid < NX && M::state_level(tid) == LEV) {
    double* row = xs + tid * (T + 1);
    const double* add = inc + tid * T;
    double v = row[0];
    int t = 0;
    for (; t + 8 <= T; t += 8) {                       // eight operands in flight; the additions stay in order
      double a[8];
      MAS_UNROLL for (int q = 0; q < 8; ++q) a[q] = add[t + q];
      MAS_UNROLL for (int q = 0; q < 8; ++q) { v = v + a[q]; row[t + q + 1] = v; }
    }
    for (; t < T; ++t) { v = v + add[t]; row[t + 1] = v; }
  }
  __syncthreads();
}
template <class M, int... LEV>
__device__ __forceinline__ void levels_all(double* xs, double* inc, const double* uc, int T, double dt, const ModelParams& P, int tid, int nthreads, std::integer_sequence<int, LEV...>) {
  (levels_pass<M, LEV>(xs, inc, uc, T, dt, P, tid, nthreads), ...);
}
// serial sum of v[0..n-1] from 0.0 in index order by the calling lane (operands in LDS, eight in flight)
__device__ __forceinline__ double lds_serial_sum(const double* v, int n) {
  double s = 0.0;
  int r = 0;
  for (; r + 8 <= n; r += 8) {
    double a[8];
    MAS_UNROLL for (int q = 0; q < 8; ++q) a[q] = v[r + q];
    MAS_UNROLL for (int q = 0; q < 8; ++q) s += a[q];
  }
  for (; r < n; ++r) s += v[r];
  return s;
}
// integrate_horizon + objective of the controls in uc[] from xs[.][0]: xs filled, the cost returned to every thread.
// Called by all threads of the workgroup (uniform); uc and xs[.][0] must be visible (a barrier behind their writes).
template <class M>
__device__ __forceinline__ double levels_rollout_cost(double* xs, double* inc, const double* uc, double* misc, int T, double dt, const ModelParams& P, int tid, int nthreads) {
  constexpr int NX = M::NX, NU = M::NU;
  static_assert(M::kRolloutLevels > 0, "model without dependency levels");
  levels_all<M>(xs, inc, uc, T, dt, P, tid, nthreads, std::make_integer_sequence<int, M::kRolloutLevels>{});
  for (int t = tid; t < T; t += nthreads) {
    double x[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = xs[i * (T + 1) + t];
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = uc[t * NU + i];
    inc[t] = M::stage_cost(x, u, t, P);
  }
  __syncthreads();
  if (tid == 0) {
    double c = lds_serial_sum(inc, T);                 // c = 0; c += stage(x_t, u_t) for t = 0 .. T-1 (ocp.hpp:14-28)
    double xT[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) xT[i] = xs[i * (T + 1) + T];
    c += M::terminal_cost(xT, P);
    misc[3] = c;
  }
  __syncthreads();
  return misc[3];
}

// The same for a GROUP of G agents at once (the trust-region epilogue inside the fused round, mas_fused_hip.hpp: up to 16 agents of a
// workgroup re-roll in the same round): agent g's arrays lie g * stride doubles further on (xs | inc | uc | misc, as above); the items
// of a pass are (agent, time step), the serial sums run on one lane per (agent, component).  Same statements, same operands, same
// order per agent: bit-identical to levels_rollout_cost.  Every agent's cost is left in its misc[3].
template <class M, int LEV>
__device__ __forceinline__ void levels_pass_group(double* base, size_t stride, int G, int T, double dt, const ModelParams& P, int tid, int nthreads) {
  constexpr int NX = M::NX, NU = M::NU;
  for (int e = tid; e < G * T; e += nthreads) {
    const int g = e / T, t = e - g * T;
    double* xs = base + g * stride; double* inc = xs + NX * (T + 1); const double* uc = inc + NX * T;
    double x[NX], u[NU], in[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = (M::state_level(i) < LEV) ? xs[i * (T + 1) + t] : 0.0;
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = uc[t * NU + i];
    rk4_increment<M>(x, u, dt, in, P);
    MAS_UNROLL for (int i = 0; i < NX; ++i) { if (M::state_level(i) == LEV) inc[i * T + t] = in[i]; }
  }
  __syncthreads();
  if (tid < G * NX) {
    const int g = tid / NX, comp = tid - g * NX;
    if (M::state_level(comp) == LEV) {
      double* row = base + g * stride + comp * (T + 1);
      const double* add = base + g * stride + NX * (T + 1) + comp * T;
      double v = row[0];
      int t = 0;
      for (; t + 8 <= T; t += 8) {
        double a[8];
        MAS_UNROLL for (int q = 0; q < 8; ++q) a[q] = add[t + q];
        MAS_UNROLL for (int q = 0; q < 8; ++q) { v = v + a[q]; row[t + q + 1] = v; }
      }
      for (; t < T; ++t) { v = v + add[t]; row[t + 1] = v; }
    }
  }
  __syncthreads();
}
template <class M, int... LEV>
__device__ __forceinline__ void levels_all_group(double* base, size_t stride, int G, int T, double dt, const ModelParams& P, int tid, int nthreads, std::integer_sequence<int, LEV...>) {
  (levels_pass_group<M, LEV>(base, stride, G, T, dt, P, tid, nthreads), ...);
}
template <class M>
__device__ __forceinline__ void levels_rollout_cost_group(double* base, size_t stride, int G, int T, double dt, const ModelParams& P, int tid, int nthreads) {
  constexpr int NX = M::NX, NU = M::NU;
  static_assert(M::kRolloutLevels > 0, "model without dependency levels");
  levels_all_group<M>(base, stride, G, T, dt, P, tid, nthreads, std::make_integer_sequence<int, M::kRolloutLevels>{});
  for (int e = tid; e < G * T; e += nthreads) {
    const int g = e / T, t = e - g * T;
    const double* xs = base + g * stride; double* inc = base + g * stride + NX * (T + 1); const double* uc = inc + NX * T;
    double x[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = xs[i * (T + 1) + t];
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = uc[t * NU + i];
    inc[t] = M::stage_cost(x, u, t, P);
  }
  __syncthreads();
  if (tid < G) {
    const double* xs = base + tid * stride; const double* inc = xs + NX * (T + 1); double* misc = base + tid * stride + NX * (T + 1) + NX * T + T * NU;
    double c = lds_serial_sum(inc, T);
    double xT[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) xT[i] = xs[i * (T + 1) + T];
    c += M::terminal_cost(xT, P);
    misc[3] = c;
  }
  __syncthreads();
}

// TrustRerollBody (nash.hpp:219-226: scaled step, rollout, cost, accept / revert) by dependency levels, one workgroup per queued agent:
// the re-roll of the three-kernel epilogue in the rounds in which thousands of solves end (norm and copy stay coalesced kernels over
// all agents there) -- ~25 us for a few hundred queued agents where the four-lanes-per-rollout chain (mas_trust_reroll_quad_kernel) takes 150.
template <class M>
__global__ void __launch_bounds__(LevelsLayout<M>::kThreads) mas_trust_reroll_levels_kernel(const TrustRerollBody<M> f) {
  extern __shared__ double lv_sh[];
  constexpr int NX = M::NX, NU = M::NU, NT = LevelsLayout<M>::kThreads;
  const Dims d = f.d;
  const int T = d.T, tid = static_cast<int>(threadIdx.x), R = T * NU, RX = (T + 1) * NX;
  if (static_cast<int>(blockIdx.x) >= *f.roll_count) return;           // uniform
  const int a = f.roll_list[blockIdx.x];
  const double sc = f.scale[a];
  double* xs = lv_sh; double* inc = xs + NX * (T + 1); double* uc = inc + NX * T; double* misc = uc + 2 * R;
  for (int r = tid; r < R; r += NT) {
    const double o = MAS_AT(f.U_old, r, a);
    const double u = o + sc * (MAS_AT(f.U, r, a) - o);                  // cand_u = old_u + scale * delta
    uc[r] = u; MAS_AT(f.U, r, a) = u;
  }
  if (tid < NX) xs[tid * (T + 1)] = MAS_AT(f.x0, tid, a);
  __syncthreads();
  const double c = levels_rollout_cost<M>(xs, inc, uc, misc, T, f.dt, f.P, tid, NT);
  for (int q = tid; q < RX; q += NT) MAS_AT(f.X, q, a) = xs[(q % NX) * (T + 1) + q / NX];
  if (tid == 0) f.ts.finish(a, c, f.ts.radii[a]);
}

// Trust-region epilogue of chained solves (TrustNormBody -> TrustRerollBody -> TrustCopyBody, mas_bodies.hpp; nash.hpp:214-243) as ONE
// kernel, one workgroup per agent whose solve ended in this round (SelectBody::ended_list): step norm (squared differences side by
// side, their sum serially in the reference's column-major order), the re-roll by dependency levels when the step leaves the radius,
// accept (radius x1.5; the new trajectory becomes the old one) or revert (radius x0.5; the old trajectory is restored).  The
// trajectory rows of ONE agent are 8-byte elements Np doubles apart, so this form is for the rounds in which few agents end (the
// host bounds the grid by the number of live agents); the three-kernel form serves the rounds in which thousands do.
template <class M>
__global__ void __launch_bounds__(LevelsLayout<M>::kThreads) mas_trust_epilogue_kernel(Dims d, double dt, ModelParams P, const double* __restrict__ x0,
                                                                                     double* __restrict__ X, double* __restrict__ U, double* __restrict__ X_old,
                                                                                     double* __restrict__ U_old, TrustState ts, const int* __restrict__ ended_list,
                                                                                     const int* __restrict__ ended_count) {
  extern __shared__ double lv_sh[];
  constexpr int NX = M::NX, NU = M::NU, NT = LevelsLayout<M>::kThreads;
  const int T = d.T, tid = static_cast<int>(threadIdx.x), R = T * NU, RX = (T + 1) * NX;
  if (static_cast<int>(blockIdx.x) >= *ended_count) return;            // uniform
  const int a = ended_list[blockIdx.x];
  double* xs = lv_sh; double* inc = xs + NX * (T + 1); double* uc = inc + NX * T; double* uo = uc + R; double* misc = uo + R;
  static_assert(NU <= NX, "the squared differences of the controls live in the increments' space");
  for (int r = tid; r < R; r += NT) {
    const double c = MAS_AT(U, r, a), o = MAS_AT(U_old, r, a);
    uc[r] = c; uo[r] = o;
    const double dl = c - o;
    inc[r] = dl * dl;
  }
  __syncthreads();
  if (tid == 0) {
    const double norm = sqrt(lds_serial_sum(inc, R));                  // delta.norm(): Frobenius, column-major order (nash.hpp:216-217)
    const double radius = ts.radii[a];
    misc[0] = (norm > radius) ? 1.0 : 0.0; misc[1] = radius / norm; misc[2] = radius;
  }
  __syncthreads();
  const bool reroll = misc[0] != 0.0;                                   // uniform
  const double radius = misc[2];
  double cand = 0.0;
  if (reroll) {                                                         // cand_u = old_u + scale * delta; cand_x = rollout; cand_cost = objective (nash.hpp:219-226)
    const double sc = misc[1];
    for (int r = tid; r < R; r += NT) { const double o = uo[r]; uc[r] = o + sc * (uc[r] - o); }
    if (tid < NX) xs[tid * (T + 1)] = MAS_AT(x0, tid, a);
    __syncthreads();
    cand = levels_rollout_cost<M>(xs, inc, uc, misc, T, dt, P, tid, NT);
  }
  if (tid == 0) {
    if (!reroll) cand = ts.cost[a];
    misc[4] = (cand < ts.cost_old[a]) ? 1.0 : 0.0;
    ts.finish(a, cand, radius);                                         // accept / revert, nash.hpp:228-243
  }
  __syncthreads();
  const bool accept = misc[4] != 0.0;                                   // uniform
  if (accept) {                         // the accepted trajectory is what the next outer iteration measures its step against
    if (reroll) {
      for (int q = tid; q < RX; q += NT) { const double v = xs[(q % NX) * (T + 1) + q / NX]; MAS_AT(X, q, a) = v; MAS_AT(X_old, q, a) = v; }
      for (int r = tid; r < R; r += NT) { const double v = uc[r]; MAS_AT(U, r, a) = v; MAS_AT(U_old, r, a) = v; }
    } else {
      for (int q = tid; q < RX; q += NT) MAS_AT(X_old, q, a) = MAS_AT(X, q, a);
      for (int r = tid; r < R; r += NT) MAS_AT(U_old, r, a) = uc[r];
    }
  } else {
    for (int q = tid; q < RX; q += NT) MAS_AT(X, q, a) = MAS_AT(X_old, q, a);
    for (int r = tid; r < R; r += NT) MAS_AT(U, r, a) = uo[r];
  }
}

}  // namespace mas
