// mas_kernel_quad.hpp -- line search / rollout / trust-region re-roll with four lanes per rollout (HIP build only; see mas_kernels_hip.hpp)
#pragma once

#include <type_traits>
#include <utility>

#include "mas_bodies.hpp"

namespace mas {

// ------------------------------------------------------- line-search rollouts, 4 lanes/trial --
// ForwardBody gives every (agent, alpha) rollout one lane, and a lone wavefront issues one FP64 instruction every
// four cycles: 4096 agents x 10 trials are 640 wavefronts whose 100-step chains of ~490 instructions ARE the launch
// time, however few agents are still active.  This kernel spends FOUR lanes on a rollout -- one per RK4 stage -- for
// the rounds (and problems) in which the live rollouts no longer fill the chip: the four sin/cos evaluations of a
// step, ~45 % of its instructions, run side by side.  That is possible where the stage ARGUMENTS of the expensive part
// do not depend on its results (QuadRk4 below: in the single-track model psi and v advance without sin/cos).
//   lane (slot, s): slot = position in the compacted list of live agents, s = RK4 stage 0..3
//   - control i = s & 1 of the feedback law on this lane, the pair exchanged inside the quad (DPP quad_perm)
//   - every lane keeps the whole state (tan, cost and the final combination are evaluated redundantly: SIMD)
//   - trial planes are indexed by SLOT, not by agent: the 16 slots of a wavefront are 16 consecutive doubles of a row,
//     lane s stores state component s, so a store instruction writes four whole 128-byte lines (ForwardBody's
//     whole-line rule); CommitSlotBody maps slot -> agent when it copies the accepted trial
// Every arithmetic operation is the one rk4_step / forward_control / stage_cost perform for that rollout, on the same
// operands in the same order -- only the lane that executes it changes -- so results are bit-identical to ForwardBody.
template <int CTRL>
__device__ __forceinline__ double quad_perm(double v) {      // quad_perm:[CTRL & 3, (CTRL >> 2) & 3, ...] on both halves
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) { return quad_perm<SRC * 0x55>(v); }      // every lane of the quad reads lane SRC
__device__ __forceinline__ double quad_from_prev(double v) { return quad_perm<0x90>(v); }        // lane s reads lane s-1 (lane 0 itself)

// integrate_rk4 (integrator.hpp:19-28) with lane s of a quad evaluating stage s.  Specialised per dynamics.
template <class M> struct QuadRk4;
// single-track (examples/models/single_track_model.hpp:23-44): d(psi)/dt = v tan(delta) / L and dv/dt = a do not
// involve X, Y, sin or cos, so lane s forms ITS stage's v and psi from x, u alone (one psi-rate from the lane below),
// evaluates sin/cos there, and the quad exchanges the four stages' X- and Y-rates.
template <>
struct QuadRk4<SingleTrack> {
  __device__ static void step(const double* x, const double* u, double dt, int s, double* out, const ModelParams& P) {
    const SingleTrack::Ctl c = SingleTrack::prepare(u, P);
    const double a = u[1], L = P.p[6];
    const double h = (s == 3) ? dt : (0.5 * dt);                       // x + (0.5*dt)*k1, x + (0.5*dt)*k2, x + dt*k3
    const double hv = x[3] + h * a;
    const double vs = (s == 0) ? x[3] : hv;                             // stage 0 evaluates f at x itself
    const double kpsi = div_by(vs * c.tan_delta, make_inv_div(L));     // dynamics_c: d[2]
    const double hp = x[2] + h * quad_from_prev(kpsi);
    const double psis = (s == 0) ? x[2] : hp;
    double sn, cs;
    mas_dt_sincos(psis, &sn, &cs);
    const double kx = vs * cs, ky = vs * sn;                            // d[0], d[1]
    const double k1x = quad_bcast<0>(kx), k2x = quad_bcast<1>(kx), k3x = quad_bcast<2>(kx), k4x = quad_bcast<3>(kx);
    const double k1y = quad_bcast<0>(ky), k2y = quad_bcast<1>(ky), k3y = quad_bcast<2>(ky), k4y = quad_bcast<3>(ky);
    const double k1p = quad_bcast<0>(kpsi), k2p = quad_bcast<1>(kpsi), k3p = quad_bcast<2>(kpsi), k4p = quad_bcast<3>(kpsi);
    out[0] = x[0] + (dt / 6.0) * (((k1x + 2.0 * k2x) + 2.0 * k3x) + k4x);
    out[1] = x[1] + (dt / 6.0) * (((k1y + 2.0 * k2y) + 2.0 * k3y) + k4y);
    out[2] = x[2] + (dt / 6.0) * (((k1p + 2.0 * k2p) + 2.0 * k3p) + k4p);
    out[3] = x[3] + (dt / 6.0) * (((a + 2.0 * a) + 2.0 * a) + a);
  }
};
template <> struct QuadRk4<SingleTrackLane> : QuadRk4<SingleTrack> {};
// bicycle6 (mas_models.hpp): steering angle and acceleration are states driven by the controls, so a, delta, v and psi of every
// stage follow from x and u without sin/cos -- and here tan(delta_s) differs from stage to stage too: the four tan AND the four
// sin/cos evaluations of a step run side by side.
template <>
struct QuadRk4<Bicycle6> {
  __device__ static void step(const double* x, const double* u, double dt, int s, double* out, const ModelParams& P) {
    const double L = P.p[6];
    const double h = (s == 3) ? dt : (0.5 * dt);
    const double ha = x[5] + h * u[1], hd = x[4] + h * u[0];
    const double as = (s == 0) ? x[5] : ha;                             // stage state: a, delta
    const double ds = (s == 0) ? x[4] : hd;
    const double hv = x[3] + h * quad_from_prev(as);                    // v + h * (dv/dt of the stage below) = v + h * a_{s-1}
    const double vs = (s == 0) ? x[3] : hv;
    const double kpsi = div_by(vs * mas_dt_tan_onediv(ds), make_inv_div(L));   // dynamics: d[2]
    const double hp = x[2] + h * quad_from_prev(kpsi);
    const double psis = (s == 0) ? x[2] : hp;
    double sn, cs;
    mas_dt_sincos(psis, &sn, &cs);
    const double kx = vs * cs, ky = vs * sn;
    const double k1x = quad_bcast<0>(kx), k2x = quad_bcast<1>(kx), k3x = quad_bcast<2>(kx), k4x = quad_bcast<3>(kx);
    const double k1y = quad_bcast<0>(ky), k2y = quad_bcast<1>(ky), k3y = quad_bcast<2>(ky), k4y = quad_bcast<3>(ky);
    const double k1p = quad_bcast<0>(kpsi), k2p = quad_bcast<1>(kpsi), k3p = quad_bcast<2>(kpsi), k4p = quad_bcast<3>(kpsi);
    const double k1v = quad_bcast<0>(as), k2v = quad_bcast<1>(as), k3v = quad_bcast<2>(as), k4v = quad_bcast<3>(as);
    out[0] = x[0] + (dt / 6.0) * (((k1x + 2.0 * k2x) + 2.0 * k3x) + k4x);
    out[1] = x[1] + (dt / 6.0) * (((k1y + 2.0 * k2y) + 2.0 * k3y) + k4y);
    out[2] = x[2] + (dt / 6.0) * (((k1p + 2.0 * k2p) + 2.0 * k3p) + k4p);
    out[3] = x[3] + (dt / 6.0) * (((k1v + 2.0 * k2v) + 2.0 * k3v) + k4v);
    out[4] = x[4] + (dt / 6.0) * (((u[0] + 2.0 * u[0]) + 2.0 * u[0]) + u[0]);
    out[5] = x[5] + (dt / 6.0) * (((u[1] + 2.0 * u[1]) + 2.0 * u[1]) + u[1]);
  }
};

// which state rows a lane of the quad stores: component s, and for NX > 4 also component 4 + (s & 1) (lanes 2, 3 repeat lanes 0, 1)
template <int NX>
struct QuadStore {
  static_assert(NX == 4 || NX == 6, "state components per quad");
  __device__ static double first(const double* v, int s) { const double lo2 = (s & 1) ? v[1] : v[0], hi2 = (s & 1) ? v[3] : v[2]; return (s & 2) ? hi2 : lo2; }
  __device__ static double second(const double* v, int s) { if constexpr (NX > 4) return (s & 1) ? v[5] : v[4]; else return 0.0; }
};

template <class M>
__global__ void __launch_bounds__(256) mas_forward_quad_kernel(const ForwardBody<M> f, const int* __restrict__ live_list, const int* __restrict__ live_count) {
  constexpr int NX = M::NX, NU = M::NU;
  static_assert((NX == 4 || NX == 6) && NU == 2, "state components per lane of the quad (QuadStore), one control per lane pair");
  const Dims d = f.d;
  const int tid = static_cast<int>(threadIdx.x), s = tid & 3, ci = s & 1;
  if (f.publish.host && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) f.publish();   // fused rounds: counters of the select that ran in the kernel before
  const int n_live = live_count ? *live_count : d.N;                   // uniform
  const int slot0 = (static_cast<int>(blockIdx.x) * 64 + (tid >> 6) * 16);
  if (slot0 >= n_live) return;                                         // wave-uniform
  const int slot = slot0 + ((tid & 63) >> 2);
  // a surplus quad of the last wavefront repeats the wavefront's first rollout (finite operands, in-range addresses); it
  // stores into its own, unused slot columns, and its merit is not stored
  const bool act = slot < n_live;
  const int src = act ? slot : slot0;
  const int a = live_list ? live_list[src] : src;
  if (__ballot(f.active[a] != 0) == 0ull) return;                      // listed agents are active; without a list this is the "solve has ended" exit
  const int j = static_cast<int>(blockIdx.y);
  double alpha = 1.0;
  for (int q = 0; q < j; ++q) alpha *= 0.5;
  double* Xj = f.Xs + static_cast<size_t>(j) * (d.T + 1) * NX * d.Np;
  double* Uj = f.Us + static_cast<size_t>(j) * d.T * NU * d.Np;

  // loads: uniform per-step base pointers + loop-invariant 32-bit byte offsets (row_in_step * Np + agent) * 8
  unsigned offx[NX], offu, offK[NX];
  MAS_UNROLL for (int i = 0; i < NX; ++i) offx[i] = static_cast<unsigned>((static_cast<size_t>(i) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  offu = static_cast<unsigned>((static_cast<size_t>(ci) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  MAS_UNROLL for (int c = 0; c < NX; ++c) offK[c] = static_cast<unsigned>((static_cast<size_t>(c * NU + ci) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  // stores: lane s owns state row s and control row s & 1 of its slot column
  const unsigned sto_x = static_cast<unsigned>((static_cast<size_t>(s) * d.Np + static_cast<size_t>(slot)) * sizeof(double));
  const unsigned sto_x2 = static_cast<unsigned>((static_cast<size_t>(4 + ci) * d.Np + static_cast<size_t>(slot)) * sizeof(double));   // NX > 4
  const unsigned sto_u = static_cast<unsigned>((static_cast<size_t>(ci) * d.Np + static_cast<size_t>(slot)) * sizeof(double));
  const size_t sX = static_cast<size_t>(NX) * d.Np, sU = static_cast<size_t>(NU) * d.Np, sK = static_cast<size_t>(NU) * NX * d.Np;
  const double *Xt = f.X, *Ut = f.U, *kt = f.kff, *Kt = f.Kfb;
  double *Xjt = Xj, *Ujt = Uj;
  struct Operands { double xr[NX], ur, k, K[NX]; };
  Operands opA, opB;
  auto request = [&](Operands& o) {
    MAS_UNROLL for (int i = 0; i < NX; ++i) o.xr[i] = ld_off(Xt, offx[i]);
    o.ur = ld_off(Ut, offu);
    o.k = ld_off(kt, offu);
    MAS_UNROLL for (int c = 0; c < NX; ++c) o.K[c] = ld_off(Kt, offK[c]);
  };
  double x[NX], xn[NX], u[NU];
  MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = ld_off(f.x0, offx[i]);
  auto store_x = [&]() {
    { unsigned o = sto_x; MAS_OPAQUE32(o); st_off(Xjt, o, QuadStore<NX>::first(x, s)); }
    if constexpr (NX > 4) { unsigned o = sto_x2; MAS_OPAQUE32(o); st_off(Xjt, o, QuadStore<NX>::second(x, s)); }
  };
  store_x();
  request(opA);
  double lbv = f.clamp ? f.lb[ci] : -INFINITY, ubv = f.clamp ? f.ub[ci] : INFINITY;
  MAS_VGPR64(lbv); MAS_VGPR64(ubv);
  double c = 0.0;
  auto step = [&](int t, const Operands& cur, Operands& next) {
    if (t + 1 < d.T) { Xt += sX; Ut += sU; kt += sU; Kt += sK; }     // uniform
    MAS_UNROLL for (int i = 0; i < NX; ++i) { MAS_OPAQUE32(offx[i]); MAS_OPAQUE32(offK[i]); }
    MAS_OPAQUE32(offu);
    request(next);
    // forward_control (mas_lane.hpp), control ci on this lane
    double ui;
    {
      double kd = 0.0;
      MAS_UNROLL for (int q = 0; q < NX; ++q) kd += cur.K[q] * (x[q] - cur.xr[q]);
      ui = (cur.ur + alpha * cur.k) + kd;
      ui = (ubv < ui) ? ubv : ui; ui = (ui < lbv) ? lbv : ui;     // (-inf / +inf without input bounds: ForwardBody)
    }
    u[0] = quad_bcast<0>(ui); u[1] = quad_bcast<1>(ui);
    { unsigned o = sto_u; MAS_OPAQUE32(o); st_off(Ujt, o, ui); }
    c += M::stage_cost(x, u, t, f.P);
    QuadRk4<M>::step(x, u, f.dt, s, xn, f.P);
    Xjt += sX; Ujt += sU;
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    store_x();
  };
  int t = 0;
  for (; t + 1 < d.T; t += 2) { step(t, opA, opB); step(t + 1, opB, opA); }
  if (t < d.T) step(t, opA, opB);
  c += M::terminal_cost(x, f.P);
  if (act && s == 0) f.trial_merit[static_cast<size_t>(j) * d.Np + a] = c;
}

// integrate_horizon + objective (RolloutCostBody) with four lanes per agent: the rollout every solve starts from (restore,
// initialize_problem) is one 64-agents-per-wavefront chain of T steps -- at 4096 agents 64 wavefronts on 1024 SIMDs.
// lane (agent, s): RK4 stage s; lane s stores state component s (16 consecutive agents = one 128-byte line per row).
template <class M>
__global__ void __launch_bounds__(256) mas_rollout_quad_kernel(const RolloutCostBody<M> f) {
  constexpr int NX = M::NX, NU = M::NU;
  const Dims d = f.d;
  const int tid = static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x), s = tid & 3;
  const int a_raw = tid >> 2;
  if ((a_raw & ~15) >= d.N) return;                                   // wave-uniform: 16 agents per wavefront
  const bool act = a_raw < d.N;
  const int a = act ? a_raw : d.N - 1;                                // a surplus quad repeats the last agent (its stores rewrite the same values)
  unsigned offx[NX], offu[NU];
  MAS_UNROLL for (int i = 0; i < NX; ++i) offx[i] = static_cast<unsigned>((static_cast<size_t>(i) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  MAS_UNROLL for (int i = 0; i < NU; ++i) offu[i] = static_cast<unsigned>((static_cast<size_t>(i) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  const unsigned sto_x = static_cast<unsigned>((static_cast<size_t>(s) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  const unsigned sto_x2 = static_cast<unsigned>((static_cast<size_t>(4 + (s & 1)) * d.Np + static_cast<size_t>(a)) * sizeof(double));   // NX > 4
  const size_t sX = static_cast<size_t>(NX) * d.Np, sU = static_cast<size_t>(NU) * d.Np;
  const double* Ut = f.U; double* Xt = f.X;
  double x[NX], xn[NX], uA[NU], uB[NU];
  MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = ld_off(f.x0, offx[i]);
  auto store_x = [&]() {
    { unsigned o = sto_x; MAS_OPAQUE32(o); st_off(Xt, o, QuadStore<NX>::first(x, s)); }
    if constexpr (NX > 4) { unsigned o = sto_x2; MAS_OPAQUE32(o); st_off(Xt, o, QuadStore<NX>::second(x, s)); }
  };
  store_x();
  MAS_UNROLL for (int i = 0; i < NU; ++i) uA[i] = ld_off(Ut, offu[i]);
  double c = 0.0;
  auto step = [&](int t, const double* u, double* un) {
    if (t + 1 < d.T) Ut += sU;                                        // uniform
    MAS_UNROLL for (int i = 0; i < NU; ++i) { MAS_OPAQUE32(offu[i]); un[i] = ld_off(Ut, offu[i]); }
    c += M::stage_cost(x, u, t, f.P);
    QuadRk4<M>::step(x, u, f.dt, s, xn, f.P);
    Xt += sX;
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    store_x();
  };
  int t = 0;
  for (; t + 1 < d.T; t += 2) { step(t, uA, uB); step(t + 1, uB, uA); }
  if (t < d.T) step(t, uA, uB);
  c += M::terminal_cost(x, f.P);
  if (act && s == 0) { f.cost[a] = c; if (f.merit) f.merit[a] = c; }
}

// TrustRerollBody (nash.hpp:219-226: scaled step, rollout, cost) with four lanes per queued agent: the re-roll is a single T-step
// chain for a handful of agents in most rounds of a chained trust-region run -- pure latency.
template <class M>
__global__ void __launch_bounds__(256) mas_trust_reroll_quad_kernel(const TrustRerollBody<M> f) {
  constexpr int NX = M::NX, NU = M::NU;
  static_assert(NU == 2, "one control per lane pair");
  const Dims d = f.d;
  const int tid = static_cast<int>(blockIdx.x) * 256 + static_cast<int>(threadIdx.x), s = tid & 3, ci = s & 1;
  const int n_roll = *f.roll_count;                                    // uniform
  const int slot_raw = tid >> 2;
  if ((slot_raw & ~15) >= n_roll) return;                              // wave-uniform
  const bool act = slot_raw < n_roll;
  const int a = f.roll_list[act ? slot_raw : (slot_raw & ~15)];        // a surplus quad repeats the wavefront's first agent (same values to the same addresses)
  const double sc = f.scale[a];
  unsigned offx[NX];
  MAS_UNROLL for (int i = 0; i < NX; ++i) offx[i] = static_cast<unsigned>((static_cast<size_t>(i) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  const unsigned offu = static_cast<unsigned>((static_cast<size_t>(ci) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  const unsigned sto_x = static_cast<unsigned>((static_cast<size_t>(s) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  const unsigned sto_x2 = static_cast<unsigned>((static_cast<size_t>(4 + ci) * d.Np + static_cast<size_t>(a)) * sizeof(double));   // NX > 4
  const size_t sX = static_cast<size_t>(NX) * d.Np, sU = static_cast<size_t>(NU) * d.Np;
  double* Ut = f.U; const double* Uo = f.U_old; double* Xt = f.X;
  double x[NX], xn[NX], u[NU];
  MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = ld_off(f.x0, offx[i]);
  auto store_x = [&]() {
    { unsigned o = sto_x; MAS_OPAQUE32(o); st_off(Xt, o, QuadStore<NX>::first(x, s)); }
    if constexpr (NX > 4) { unsigned o = sto_x2; MAS_OPAQUE32(o); st_off(Xt, o, QuadStore<NX>::second(x, s)); }
  };
  store_x();
  double curA = ld_off(Ut, offu), oldA = ld_off(Uo, offu), curB, oldB;
  double c = 0.0;
  auto step = [&](int t, double ucur, double uold, double& ncur, double& nold) {
    double* Uthis = Ut;
    if (t + 1 < d.T) { Ut += sU; Uo += sU; }                           // uniform
    { unsigned o = offu; MAS_OPAQUE32(o); ncur = ld_off(Ut, o); nold = ld_off(Uo, o); }
    const double ui = uold + sc * (ucur - uold);                        // cand_u = old_u + scale * delta, control ci on this lane
    { unsigned o = offu; MAS_OPAQUE32(o); st_off(Uthis, o, ui); }
    u[0] = quad_bcast<0>(ui); u[1] = quad_bcast<1>(ui);
    c += M::stage_cost(x, u, t, f.P);
    QuadRk4<M>::step(x, u, f.dt, s, xn, f.P);
    Xt += sX;
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = xn[i];
    store_x();
  };
  int t = 0;
  for (; t + 1 < d.T; t += 2) { step(t, curA, oldA, curB, oldB); step(t + 1, curB, oldB, curA, oldA); }
  if (t < d.T) step(t, curA, oldA, curB, oldB);
  c += M::terminal_cost(x, f.P);
  if (act && s == 0) f.ts.finish(a, c, f.ts.radii[a]);
}

}  // namespace mas
