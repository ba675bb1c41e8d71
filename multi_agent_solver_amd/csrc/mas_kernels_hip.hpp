// mas_kernels_hip.hpp -- the hand-shaped gfx950 kernels of the iteration (HIP build only).
//
//  * mas_sweep_kernel            linearisation sweep, one lane per (agent, t).  Each lane produces one derivative tile (ND doubles)
//                                in registers; the wavefront stages the tiles through LDS in 16-element chunks so that every global
//                                store instruction writes contiguous 128-byte runs of the tile array.  Inside a solve the wavefront
//                                slots walk the compacted list of still-active agents.
//  * mas_forward_quad_kernel,    line search / initial rollout / trust-region re-roll with FOUR lanes per rollout, one per RK4 stage
//    mas_rollout_quad_kernel,    (QuadRk4): for launches whose rollouts do not fill the chip, where the launch time is one wavefront's
//    mas_trust_reroll_quad_kernel chain.  The one-lane forms are ForwardBody / RolloutCostBody / TrustRerollBody (mas_bodies.hpp).
//  * mas_backward_row_kernel     backward Riccati recursion for NX == 4: the 16 lanes of an agent are one DPP row holding the 4 x 4
//                                value Hessian; V and the tiles never touch LDS (quad_perm / row_newbcast / direct loads), one LDS
//                                round trip per step.
//  * mas_backward_row_pipelined_kernel  the same with the shared ladder of the centralised strategy, the cross-workgroup rung
//                                agreement hidden behind a speculative step.
//  * mas_backward_tile_kernel    backward recursion for any NX with SIXTEEN lanes per agent over an LDS tile per agent: the small
//                                dense products of ilqr.hpp:115-119,183-192 are spread element-wise over the 16 lanes.
// All backward kernels compute, element for element, the arithmetic of BackwardBody (the one-lane form the host emulation
// replays): every output element is the same sequence of multiplies and adds, sums in k order (sum_terms: from the first term).
// Their time-step loops are written against the gfx950 ISA they compile to: a lone wavefront per SIMD pays 5.5-7 clock ticks per
// instruction whatever its instruction-level parallelism, so the loops avoid selects (loop-invariant -0.0 addends), predication
// (clamped prefetch lanes), vector address arithmetic (scalar base + 32-bit lane offsets) and LDS round trips.
//
// No MFMA: v_mfma_f64_4x4x4 fuses the multiply-adds of a dot product, the reference (SSE2, no FMA;
// CMakeLists.txt sets no -march) rounds after every multiply -- and the algorithm amplifies last-bit
// differences chaotically (include/mas_dettrig.h), so a fused product would break parity.
#pragma once

#include <type_traits>
#include <utility>

#include "mas_bodies.hpp"

namespace mas {

// ------------------------------------------------------------------------- sweep --------
// live_list / live_count (both null or both set): the agents still active, compacted by the previous iteration's
// select (SelectBody::next_list).  Wavefront slot s then linearises agent live_list[s], so the work of a launch
// shrinks with the number of active agents instead of leaving the lanes of retired agents idle.  A wavefront whose
// 64 slots hold one aligned block of 64 consecutive agents (every wavefront while all agents are active) keeps
// the contiguous full-line store path; any other wavefront scatters whole tile chunks (128 B per agent and chunk).
template <class M>
__global__ void __launch_bounds__(256, 2) mas_sweep_kernel(Dims d, ModelParams P, const double* __restrict__ X, const double* __restrict__ U,
                                                        double* __restrict__ D, const int* __restrict__ active, int lo, int n,
                                                        unsigned long long* __restrict__ stamp,
                                                        const int* __restrict__ live_list, const int* __restrict__ live_count) {
  using L = DerivLayout<M>;
  constexpr int NX = M::NX, NU = M::NU, ND = L::ND;
  constexpr int CH = 16;              // tile elements staged per pass
  // profiling only (stamp == nullptr otherwise): every workgroup stores its first and last 100-MHz REFCLK
  // tick in its own slot (no atomics, no contention); the host takes min / max = the kernel's execution
  // span, free of the dispatch overhead that host-side events include
  unsigned long long* my_stamp = stamp ? stamp + 2 * (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) : nullptr;
  if (my_stamp && threadIdx.x == 0 && threadIdx.y == 0) my_stamp[0] = __builtin_amdgcn_s_memrealtime();
  struct StampEnd {
    unsigned long long* p;
    __device__ ~StampEnd() { if (p && threadIdx.x == 0 && threadIdx.y == 0) p[1] = __builtin_amdgcn_s_memrealtime(); }
  } stamp_end{my_stamp};
  constexpr int PITCH = CH + 1;       // odd pitch: the 16 lanes of a ds_write_b64 group hit distinct banks
  __shared__ double stage[4][64 * PITCH];
  __shared__ int slot_agent[4][64];

  const int lane = threadIdx.x, w = threadIdx.y;
  const int t = blockIdx.y * 4 + w;
  if (t >= d.T) return;                                    // wave-uniform
  int a; bool act;
  if (live_list != nullptr) {
    const int n_live = *live_count;                        // uniform
    const int s = blockIdx.x * 64 + lane;
    if (static_cast<int>(blockIdx.x) * 64 >= n_live) return;
    act = s < n_live;
    a = act ? live_list[s] : -1;
  } else {
    a = lo + blockIdx.x * 64 + lane;                       // lo is a multiple of 64, so the wavefront covers one 64-agent block
    act = (a < n) && active[a] != 0;
  }
  const unsigned long long mask = __ballot(act);
  if (mask == 0ull) return;                                // wave-uniform: nothing to linearise in this group
  const int a_first = __builtin_amdgcn_readfirstlane(a);   // agent of the first active lane
  const bool one_block = (live_list == nullptr) ||
                         (mask == ~0ull && (a_first & 63) == 0 && __ballot(a == a_first + lane) == ~0ull);

  double Dl[ND];
  if (act) {
    double x[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = MAS_AT(X, t * NX + i, a);
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = MAS_AT(U, t * NU + i, a);
    linearise<M>(x, u, t, Dl, P);
  } else {
    MAS_UNROLL for (int i = 0; i < ND; ++i) Dl[i] = 0.0;
  }

  double* st = stage[w];
  double* step_tiles = D + static_cast<size_t>(t) * (d.Np / 64) * 64 * ND;     // tiles of time step t
  if (one_block) {
    // the 64 tiles of this wavefront: 4 (5, 7 ...) chunk runs of 64*cw doubles each, contiguous and 64-B aligned
    const int a0 = (live_list == nullptr) ? (lo + static_cast<int>(blockIdx.x) * 64) : a_first;
    double* block = step_tiles + static_cast<size_t>(a0 / 64) * 64 * ND;
    // two copies of the store loop, chosen once per wavefront: all 64 agents active (plain stores) or not
    // (each store predicated on its tile's agent)
    auto flush = [&](auto masked_tag) {
      constexpr bool kMasked = decltype(masked_tag)::value;
      MAS_UNROLL for (int c0 = 0; c0 < ND; c0 += CH) {
        const int cw = (ND - c0 < CH) ? (ND - c0) : CH;              // compile-time after unrolling
        double* run = block + static_cast<size_t>(c0) * 64;
        __builtin_amdgcn_wave_barrier();
        MAS_UNROLL for (int e = 0; e < CH; ++e) { if (e < cw) st[lane * PITCH + e] = Dl[c0 + e]; }
        __builtin_amdgcn_wave_barrier();                              // LDS ops of one wave complete in order
        MAS_UNROLL for (int k = 0; k < CH; ++k) {
          if (k < cw) {
            const int idx = k * 64 + lane;                            // linear index into the run: [agent][cw]
            const int ag = idx / cw, e = idx - ag * cw;
            const double v = st[ag * PITCH + e];
            if constexpr (kMasked) { if ((mask >> ag) & 1ull) run[idx] = v; }
            else run[idx] = v;                                        // 64 lanes -> 512 contiguous bytes
          }
        }
      }
    };
    if (mask == ~0ull) flush(std::false_type{}); else flush(std::true_type{});
    return;
  }

  // compacted slots: slot ag holds agent slot_agent[ag]; store k of a chunk moves element e of slot ag's tile to
  //   step_tiles[(A/64)*64*ND + c0*64 + (A%64)*cw + e]   (tile_offset), -1 = empty slot
  // The (slot, element) a lane serves in store k depends only on k and the chunk width, so the destinations are
  // computed once per wavefront: 16 for the full-width chunks, ND % 16 for the short last one.
  slot_agent[w][lane] = a;
  __builtin_amdgcn_wave_barrier();
  constexpr int NFULL = ND / CH, REM = ND % CH;
  int dst_full[CH], dst_rem[REM > 0 ? REM : 1];
  MAS_UNROLL for (int k = 0; k < CH; ++k) {
    const int A = slot_agent[w][k * (64 / CH) + lane / CH];
    dst_full[k] = (A < 0) ? -1 : (A >> 6) * 64 * ND + (A & 63) * CH + (lane & (CH - 1));
  }
  MAS_UNROLL for (int k = 0; k < REM; ++k) {
    constexpr int REMD = REM > 0 ? REM : 1;                          // (the loop has no trips when REM == 0)
    const int idx = k * 64 + lane, ag = idx / REMD, e = idx - ag * REMD;
    const int A = slot_agent[w][ag];
    dst_rem[k] = (A < 0) ? -1 : (A >> 6) * 64 * ND + NFULL * CH * 64 + (A & 63) * REM + e;
  }
  MAS_UNROLL for (int c0 = 0; c0 < ND; c0 += CH) {
    const int cw = (ND - c0 < CH) ? (ND - c0) : CH;                  // compile-time after unrolling
    __builtin_amdgcn_wave_barrier();
    MAS_UNROLL for (int e = 0; e < CH; ++e) { if (e < cw) st[lane * PITCH + e] = Dl[c0 + e]; }
    __builtin_amdgcn_wave_barrier();
    MAS_UNROLL for (int k = 0; k < CH; ++k) {
      if (k < cw) {
        const int idx = k * 64 + lane;
        const int ag = idx / cw, e = idx - ag * cw;
        const double v = st[ag * PITCH + e];
        if (cw == CH) { if (dst_full[k] >= 0) step_tiles[static_cast<size_t>(dst_full[k]) + static_cast<size_t>(c0) * 64] = v; }
        else { if (dst_rem[k] >= 0) step_tiles[dst_rem[k]] = v; }
      }
    }
  }
}

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

template <int SRC>
__device__ __forceinline__ double row_bcast(double v) {      // lane SRC of the 16-lane row -> every lane of the row
  return __longlong_as_double(__builtin_amdgcn_update_dpp(0ll, __double_as_longlong(v), 0x150 + SRC, 0xF, 0xF, true));
}
// LLT::solve(Identity) (llt_solve_identity, mas_lane.hpp) with its NU columns spread over the lanes of an agent's DPP row: lane l
// solves for column l % 4 -- the same sequence of operations for every column, the unit vector's entries as data -- and the
// columns are gathered by row broadcasts (lane c of the row holds column c).  Each entry is exactly what llt_solve_identity
// computes; NU - 1 column solves less per lane (at NU = 4: ~75 of ~500 instructions of a backward step).
template <int N, int... E>
__device__ __forceinline__ void gather_inverse(const double* col, double* Inv, std::integer_sequence<int, E...>) {
  ((Inv[E] = row_bcast<E / N>(col[E % N])), ...);                     // Inv[c * N + i] = column c, entry i
}
template <int N>
__device__ __forceinline__ void llt_solve_identity_by_columns(const double* Lm, int c, double* Inv) {
  double o[N];
  MAS_UNROLL for (int i = 0; i < N; ++i) o[i] = (i == c) ? 1.0 : 0.0;
  MAS_UNROLL for (int i = 0; i < N; ++i) {
    const double a = 1.0 / Lm[i * N + i];
    o[i] *= a;
    const double b = o[i];
    MAS_UNROLL for (int r = i + 1; r < N; ++r) o[r] -= b * Lm[i * N + r];
  }
  MAS_UNROLL for (int i = N - 1; i >= 0; --i) {
    const double a = 1.0 / Lm[i * N + i];
    double b = 0.0;
    MAS_UNROLL for (int r = i + 1; r < N; ++r) b += Lm[i * N + r] * o[r];
    o[i] = (o[i] - b) * a;
  }
  gather_inverse<N>(o, Inv, std::make_integer_sequence<int, N * N>{});
}
// q_uu entries from the pass-2 results of the LDS-tile kernel: entry e is item W0 + e, formed in round (W0 + e) / 16 by lane (W0 + e) % 16
template <int W0, int R2, int... E>
__device__ __forceinline__ void tile_gather_quu(const double (&res)[R2], double* quu, std::integer_sequence<int, E...>) {
  ((quu[E] = row_bcast<(W0 + E) % 16>(res[(W0 + E) / 16])), ...);
}
// sum of N terms in index order, starting FROM THE FIRST TERM -- as Eigen's redux does for the reference's small products.  The
// CPU oracle and the one-lane bodies start from +0.0 (s = 0.0; s += term_k); the two agree in every bit except that an
// all-zero sum can come out as -0.0 here and +0.0 there.  No operation of the path tells the two zeros apart (it never
// divides by, takes the square root or the sign of such a value: LLT rejects a pivot <= 0 before its sqrt), so every result the
// ABI returns is the same IEEE value; per backward step of the (4, 2) models this is 26 dependent additions less.
template <int N, class F>
__device__ __forceinline__ double sum_terms(F term) {
  double s = term(0);
  MAS_UNROLL for (int k = 1; k < N; ++k) s += term(k);
  return s;
}

// ---------------------------------------------------------------- backward, 16 lanes/agent --
template <class M>
struct BackwardTileLayout {
  using L = DerivLayout<M>;
  static constexpr int NX = M::NX, NU = M::NU, ND = L::ND;
  // LDS tile of one agent, in doubles.  Two rules shape it (MI355X_MICROARCH.md, LDS): a ds_write_b64 is served in four
  // groups of 16 consecutive lanes -- one agent's 16 lanes -- over 32 dword banks, so the 16 lanes must hit 16 different
  // doubles modulo 16; and lanes that have nothing to write must NOT share one dummy address (sixteen lanes on one
  // address cost fifteen extra cycles: that, not the operand reads, was the 64 conflict cycles per time step of round 1).
  // Hence a PAD region of 16 doubles at an offset that is a multiple of 16: an idle lane writes pad_at(x) where x is the
  // address it would have had if the item list went on -- the residues modulo 16 of one instruction stay a permutation.
  static constexpr int oD = 0, oV = oD + ND, oVx = oV + NX * NX, oAtV = oVx + NX, oBtV = oAtV + NX * NX, oQx = oBtV + NU * NX,
                       oQu = oQx + NX, oQxx = oQu + NU, oQux = oQxx + NX * NX, oQuu = oQux + NU * NX, oL = oQuu + NU * NU, oNegZero = oL + NU * NU,
                       kUsed = oNegZero + 1, oPad = (kUsed + 15) / 16 * 16, kEnd = oPad + 16;
  // agents of one 32-lane read group (two tiles) kStrideMod doubles apart modulo 32
#if defined(MAS_BWD_STRIDE_MOD32)
  static constexpr int kStrideMod = MAS_BWD_STRIDE_MOD32;
#else
  static constexpr int kStrideMod = 16;
#endif
  static constexpr int STRIDE = kEnd + ((kStrideMod - kEnd % 32) % 32 + 32) % 32;
  static_assert(STRIDE >= kEnd, "tile stride");
  __host__ __device__ static constexpr int pad_at(int x) { return oPad + (x & 15); }
};

// Per time step the wavefront makes three passes over its four LDS tiles, every lane running the
// same instruction stream on its own operand addresses (no divergence inside a pass):
//   pass 1   A^T V, B^T V, q_x, q_u                       (length-NX dot products)
//   pass 2   q_xx, q_ux, q_uu                             (length-NX dot products)
//   pass 3   LLT + regularisation ladder, Quu^-1, gains, value update, symmetrisation -- all in
//            registers: each lane rebuilds the two gain columns and the two rows of K^T q_uu it needs,
//            evaluates its own and its mirror entry of v_xx and applies the reference's aliased
//            in-place symmetrisation rule, so no further exchange is needed.
// Rung agreement of the centralised strategy (SharedLadder): the stacked q_uu is block-diagonal, its LLT fails if
// ANY block fails and reg*I is then added to EVERY block (ilqr.hpp:172-182), so all agents use the first rung at
// which all of them succeed.  Per (time step, chunk of 16 rungs) there is ONE 64-bit word: every wavefront ORs the rungs
// at which one of its agents failed into bits 0..15 and then adds 1 to the arrival counter in bits 32.., both
// fire-and-forget, and polls the word until the counter shows all wavefronts: a rung is agreed on when its bit is clear.
// The OR precedes the arrival in the word's modification order because both come from one lane (coherence of a single
// location follows program order), so a complete counter implies a complete mask.  tools/ubench/xcd_barrier.hip measures the
// bare arrive-and-wait of 16 workgroups at 0.87 us (0.72 us when they share an XCD); round 2's three words with 9-bit failure
// counters, behind a workgroup-level combination with two barriers, cost ~2 us per time step.
// The launch is at most 256 workgroups, all resident at once (checked on the host), so the wait always ends; it is bounded
// anyway (a wavefront that gives up raises *error, every later wait returns at once, and the host reports it).
struct LadderSync {
  unsigned long long* words = nullptr;   // [T * kChunks][4] (one used), preset to zero
  int* error = nullptr;
  // (not part of the agreement; carried here because every backward kernel shares one signature)  Per-agent LDS-tile kernel: the
  // round's compacted list of live agents -- wavefront slot s works on agent live_list[s], so the launch shrinks with the live
  // agents.  With two wavefronts per SIMD (8192 six-state agents = 2048 wavefronts) that halves the time of the tail rounds,
  // where almost every group of four consecutive agents still holds a live one.  null: slot s = agent lo + s, `active` decides.
  const int* live_list = nullptr; const int* live_count = nullptr;
  int xcd_stride = 1;                    // pipelined kernel: 8 = only every eighth workgroup works (the others exit at once), which puts all working workgroups on ONE XCD (workgroups are dealt round-robin over the 8 XCDs; tools/ubench/xcd_barrier.hip checks that through HW_REG_XCC_ID): their arrive-and-wait then stays inside one L2 (0.72 against 0.87 us bare).  Placement only -- the agreement is correct wherever the workgroups run
  int combine = 1;                       // workgroups of several wavefronts: combine their masks through LDS and arrive once per workgroup (measured on C2, us per pass: 245 against 546 when each of the 64 wavefronts arrives itself; single-wavefront workgroups: 253)
  static constexpr int kChunks = 32, kArrivalShift = 32;
};
__device__ inline void ladder_post(const LadderSync& sy, int slot, unsigned wave_ok_mask) {
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* w = sy.words + static_cast<size_t>(slot) * 4;
    const unsigned long long fail = ~wave_ok_mask & 0xFFFFu;
    if (fail != 0ull) __hip_atomic_fetch_or(w, fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(w, 1ull << LadderSync::kArrivalShift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ inline unsigned ladder_wait(const LadderSync& sy, int slot, int n_waves) {
  unsigned agreed = 0u;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* w = sy.words + static_cast<size_t>(slot) * 4;
    unsigned long long v = 0ull;
    int spins = 0;
    for (;;) {
      v = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (static_cast<int>(v >> LadderSync::kArrivalShift) == n_waves) break;
      if (++spins > (1 << 18) || __hip_atomic_load(sy.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { atomicExch(sy.error, 1); break; }   // ~0.3 s, once per launch
      __builtin_amdgcn_s_sleep(1);
    }
    agreed = static_cast<unsigned>(~v) & 0xFFFFu;
  }
  return static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(agreed)));
}
__device__ inline unsigned ladder_agree(const LadderSync& sy, int slot, unsigned wave_ok_mask, int n_waves) {
  ladder_post(sy, slot, wave_ok_mask);
  return ladder_wait(sy, slot, n_waves);
}

// WPB = wavefronts per workgroup; the wavefronts of a workgroup are independent (each owns its four LDS tiles) except in the
// shared ladder.  Two reasons for WPB = 4:
//  * placement (per-agent Nash path): a workgroup's four wavefronts go to the four SIMDs of one CU, so 4096 agents = 256
//    workgroups = exactly one wavefront per SIMD.  As 1024 single-wavefront workgroups the dispatcher doubles some SIMDs up
//    while others stay empty, and the launch takes as long as its slowest SIMD: measured 163 us against 124 us per launch.
//  * the shared ladder of the centralised strategy: the rung masks of a workgroup's wavefronts are first combined through
//    LDS (one s_barrier), and only ONE wavefront per workgroup takes part in the cross-workgroup agreement, so 256 agents
//    are 16 arrivals per time step instead of 64 -- and a problem that fits one workgroup needs no global agreement at all.
template <class M, int LPA, bool SharedLadder = false, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB) mas_backward_tile_kernel(Dims d, ModelParams P, const double* __restrict__ X, const double* __restrict__ D,
                                                               double* __restrict__ kff, double* __restrict__ Kfb,
                                                               long long* __restrict__ reg_events, const int* __restrict__ active, int lo, int n,
                                                               LadderSync sync) {
  using TL = BackwardTileLayout<M>;
  using L = DerivLayout<M>;
  constexpr int NX = M::NX, NU = M::NU, ND = L::ND;
  constexpr int APW = 64 / LPA;                     // agents per wavefront (LPA lanes per agent)
  constexpr int LL = 16;                            // ladder width: rungs tried at once (lanes beyond LL repeat rung lane % LL)
  static_assert(LPA >= LL && 64 % LPA == 0, "lanes per agent");
  // kLocalQuu (experiment, off): every lane forms q_uu itself right after pass 1, so that the LLT / ladder chain starts one
  // LDS round trip earlier and pass 2 runs in its shadow.  Measured on C3: 142 us per launch against 132 us -- the 36 extra
  // FP64 instructions per lane cost more than the round trip saves; the kernel is bound by dependent-issue latency
  // (7 cycles per dependent FP64 instruction, ~255 of them per step), not by LDS.
#if defined(MAS_BWD_LOCAL_QUU)
  constexpr bool kLocalQuu = (MAS_BWD_LOCAL_QUU != 0) && NU <= 2;
#else
  constexpr bool kLocalQuu = false;
#endif
  __shared__ double lds[WPB * APW * TL::STRIDE];
  __shared__ unsigned blk_field[2][WPB > 1 ? WPB : 1];   // rung masks of the workgroup's wavefronts, double-buffered by agreement parity
  __shared__ unsigned blk_agreed[2];
  unsigned agreement_parity = 0;

  const int wave = static_cast<int>(threadIdx.x) >> 6;    // 0 when WPB == 1
  const int tid = static_cast<int>(threadIdx.x) & 63, lane = tid & (LPA - 1), slot = tid / LPA, rung_lane = lane & (LL - 1);
  const int slot0 = (static_cast<int>(blockIdx.x) * WPB + wave) * APW;
  int a = lo + slot0 + slot;
  bool valid, counted;
  if (!SharedLadder && sync.live_list != nullptr) {
    const int n_live = *sync.live_count;                    // uniform
    if (slot0 >= n_live) return;
    counted = slot0 + slot < n_live;
    a = sync.live_list[counted ? slot0 + slot : slot0];     // a surplus slot of the last wavefront repeats its first agent: same values to the same addresses
    valid = true;
  } else {
    valid = (a < n) && active[a] != 0;
    counted = valid;
  }
  if (!SharedLadder && __ballot(valid) == 0ull) return;   // uniform for the (single-wave) block; a shared ladder needs every wavefront
  if (SharedLadder && active[lo] == 0) return;             // the stacked solve has ended (all flags fall together): uniform over the whole launch
  const int tile = wave * APW + slot;
  double* T = lds + tile * TL::STRIDE;
  // the tiles of a wavefront are private to it: what separates its LDS writes from its own later reads is the order in which
  // one wavefront's LDS instructions execute, not a workgroup barrier (with one wavefront per workgroup __syncthreads()
  // compiles to exactly that wait; with several it would be a real s_barrier nobody needs here)
  auto wave_sync = [] {
    if constexpr (WPB == 1) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  if (!valid) {                                     // idle slot: finite operands (its lanes run the same code, results unused)
    for (int e = lane; e < TL::STRIDE; e += LPA) T[e] = 0.0;
  }
  if (lane == 0) T[TL::oNegZero] = -0.0;            // additive identity that keeps every bit of the other operand (x + -0.0 == x)
  if (valid && lane == 0) {                         // terminal boundary condition, ilqr.hpp:92-102
    double xT[NX], vx[NX], vxx[NX * NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) xT[i] = MAS_AT(X, d.T * NX + i, a);
    terminal_boundary<M>(xT, vx, vxx, P);
    MAS_UNROLL for (int i = 0; i < NX; ++i) T[TL::oVx + i] = vx[i];
    MAS_UNROLL for (int i = 0; i < NX * NX; ++i) T[TL::oV + i] = vxx[i];
  }

  // ---- loop-invariant operand addresses of this lane's work items (tile-relative, in doubles)
  constexpr int W1 = NX * NX + NU * NX + NX + NU, R1 = (W1 + LPA - 1) / LPA;
  constexpr int W2 = NX * NX + NU * NX + (kLocalQuu ? 0 : NU * NU), R2 = (W2 + LPA - 1) / LPA;
  int p1[R1], q1[R1], b1[R1], o1[R1];              // pass 1: out = [bias +] sum_k T[p+k] * T[q+k]
  MAS_UNROLL for (int r = 0; r < R1; ++r) {
    const int w = lane + r * LPA;
    int pa = 0, qa = 0, ba = TL::oNegZero, oa = TL::pad_at(TL::oAtV + w);     // idle lanes of the last round: own pad slot
    if (w < NX * NX) { const int i = w % NX, j = w / NX; pa = TL::oD + L::oA + i * NX; qa = TL::oV + j * NX; oa = TL::oAtV + w; }
    else if (w < NX * NX + NU * NX) { const int e = w - NX * NX, i = e % NU, j = e / NU; pa = TL::oD + L::oB + i * NX; qa = TL::oV + j * NX; oa = TL::oBtV + e; }
    else if (w < NX * NX + NU * NX + NX) { const int i = w - NX * NX - NU * NX; pa = TL::oD + L::oA + i * NX; qa = TL::oVx; ba = TL::oD + L::oLx + i; oa = TL::oQx + i; }
    else if (w < W1) { const int i = w - NX * NX - NU * NX - NX; pa = TL::oD + L::oB + i * NX; qa = TL::oVx; ba = TL::oD + L::oLu + i; oa = TL::oQu + i; }
    p1[r] = pa; q1[r] = qa; b1[r] = ba; o1[r] = oa;
  }
  int p2[R2], ps2[R2], q2[R2], b2[R2], o2[R2];     // pass 2: out = bias + sum_k T[p + k*ps] * T[q+k]
  MAS_UNROLL for (int r = 0; r < R2; ++r) {
    const int w = lane + r * LPA;
    int pa = 0, ps = 1, qa = 0, ba = 0, oa = TL::pad_at(TL::oQxx + w);
    if (w < NX * NX) { const int i = w % NX, j = w / NX; pa = TL::oAtV + i; ps = NX; qa = TL::oD + L::oA + j * NX; ba = TL::oD + L::oLxx + w; oa = TL::oQxx + w; }
    else if (w < NX * NX + NU * NX) { const int e = w - NX * NX, i = e % NU, j = e / NU; pa = TL::oBtV + i; ps = NU; qa = TL::oD + L::oA + j * NX; ba = TL::oD + L::oLux + e; oa = TL::oQux + e; }
    else if (w < W2) { const int e = w - NX * NX - NU * NX, i = e % NU, j = e / NU; pa = TL::oBtV + i; ps = NU; qa = TL::oD + L::oB + j * NX; ba = TL::oD + L::oLuu + e; oa = TL::oQuu + e; }
    p2[r] = pa; ps2[r] = ps; q2[r] = qa; b2[r] = ba; o2[r] = oa;
  }

  // software pipeline: the tiles of step t-1 travel HBM -> registers while step t is computed.  Every lane works for
  // its OWN agent: lane l fetches element c*16 + l of its agent's tile (chunk c holds the 4 agents' cw-element pieces back
  // to back, tile_offset) and publishes it to its agent's LDS tile, so each 16-lane write group fills 16 consecutive
  // doubles.  No predication: a surplus lane of the short last chunk re-reads the piece's last element and publishes it to
  // its pad slot; fetch and publish are NCH loads and NCH LDS writes with loop-invariant per-lane offsets and one uniform
  // base pointer that steps back by Np*ND doubles per time step.
  constexpr int NCH = (ND + kTileChunk - 1) / kTileChunk;
  static_assert(kTileChunk <= LPA, "one lane per chunk element");
  double pre[NCH];
  unsigned fetch_off[NCH]; int publish_at[NCH];
  MAS_UNROLL for (int c = 0; c < NCH; ++c) {
    const int cw = (ND - c * kTileChunk < kTileChunk) ? (ND - c * kTileChunk) : kTileChunk;
    const bool mine = lane < cw;
    const int e = mine ? lane : cw - 1;
    fetch_off[c] = static_cast<unsigned>((static_cast<size_t>(a / 64) * 64 * ND + static_cast<size_t>(c) * kTileChunk * 64 + static_cast<size_t>(a % 64) * cw + e) * sizeof(double));
    publish_at[c] = tile * TL::STRIDE + (mine ? TL::oD + c * kTileChunk + lane : TL::pad_at(TL::oD + c * kTileChunk + lane));
  }
  const size_t step_doubles = static_cast<size_t>(d.Np) * ND;
  // (uniform base = the tiles of a time step; the agent's 64-block is part of the lane offset, so that listed agents may come from anywhere)
  const double* tiles_t = D + static_cast<size_t>(d.T - 1) * step_doubles;
  auto fetch = [&]() { MAS_UNROLL for (int c = 0; c < NCH; ++c) pre[c] = ld_off(tiles_t, fetch_off[c]); };
  long long regs = 0;
  // ladder addends of this lane, loop-invariant: reg_q = 1e-6 * 10^q (by repeated multiplication, exactly the
  // reference's `reg *= 10.0`) for the q < lane rungs it has to apply, -0.0 (adds nothing, changes no bit) after
  double rung_add[LL - 1];
  {
    double reg = 1e-6;
    MAS_UNROLL for (int q = 0; q < LL - 1; ++q) { rung_add[q] = (q < rung_lane) ? reg : -0.0; reg *= 10.0; }
  }
  // gain stores: uniform per-step base pointers + loop-invariant 32-bit byte offsets (row * Np + agent) * 8
  constexpr int RMX = (NX * NX + LPA - 1) / LPA;
  unsigned goff[NU], koff[RMX][NU];                  // k rows; this lane's gain column (rows i*NU + m) per round
  MAS_UNROLL for (int m = 0; m < NU; ++m) goff[m] = static_cast<unsigned>((static_cast<size_t>(m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  MAS_UNROLL for (int r = 0; r < RMX; ++r) {
    const int w = lane + r * LPA, i = (w < NX * NX) ? (w % NX) : 0;
    MAS_UNROLL for (int m = 0; m < NU; ++m) koff[r][m] = static_cast<unsigned>((static_cast<size_t>(i * NU + m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  }
  double* kff_t = kff + static_cast<size_t>(d.T - 1) * NU * d.Np;
  double* Kfb_t = Kfb + static_cast<size_t>(d.T - 1) * NU * NX * d.Np;
  fetch();

  // one time step.  The first step is peeled off the loop so that the loop is entered in the state its back edge
  // has -- tile loads issued, then the gain stores of a step -- and the wait for the tile at the head of the loop can
  // let the younger stores stay in flight (s_waitcnt vmcnt(#stores)); entered from straight-line code the compiler has
  // to wait for vmcnt(0) there, which puts every step's store acknowledgements on the serial chain.
  auto step = [&](int t) {
    MAS_UNROLL for (int c = 0; c < NCH; ++c) lds[publish_at[c]] = pre[c];
    if (t > 0) tiles_t -= step_doubles;              // uniform; at t == 0 the (unused) prefetch re-reads step 0
    MAS_UNROLL for (int c = 0; c < NCH; ++c) MAS_OPAQUE32(fetch_off[c]);
    MAS_UNROLL for (int m = 0; m < NU; ++m) MAS_OPAQUE32(goff[m]);
    MAS_UNROLL for (int r = 0; r < RMX; ++r) { MAS_UNROLL for (int m = 0; m < NU; ++m) MAS_OPAQUE32(koff[r][m]); }
    fetch();
    wave_sync();

    // ---- pass 1 (ilqr.hpp:115-116 and the first factors of :117-119)
    {
      double res[R1];
      MAS_UNROLL for (int r = 0; r < R1; ++r) {
        const double s = sum_terms<NX>([&](int k) { return T[p1[r] + k] * T[q1[r] + k]; });
        res[r] = T[b1[r]] + s;                                        // items without a bias term add -0.0
      }
      MAS_UNROLL for (int r = 0; r < R1; ++r) T[o1[r]] = res[r];      // idle lanes write their own pad slot
    }
    wave_sync();
    double quu[NU * NU], qu[NU], Lfac[NU * NU], ninv[NU * NU], kf[NU];
    if constexpr (kLocalQuu) {                                        // q_uu = l_uu + (B^T V) B, :119, every lane for itself
      MAS_UNROLL for (int e = 0; e < NU * NU; ++e) {
        const int i = e % NU, j = e / NU;
        double s = 0.0;
        MAS_UNROLL for (int k = 0; k < NX; ++k) s += T[TL::oBtV + i + k * NU] * T[TL::oD + L::oB + j * NX + k];
        quu[e] = T[TL::oD + L::oLuu + e] + s;
      }
    }
    // ---- pass 2 (second factors of ilqr.hpp:117-119)
    {
      double res[R2];
      MAS_UNROLL for (int r = 0; r < R2; ++r) {
        const double s = sum_terms<NX>([&](int k) { return T[p2[r] + k * ps2[r]] * T[q2[r] + k]; });
        res[r] = T[b2[r]] + s;
      }
      MAS_UNROLL for (int r = 0; r < R2; ++r) T[o2[r]] = res[r];
      if constexpr (!kLocalQuu && LPA == 16) {
        // q_uu straight from the lanes that formed it (item NX^2 + NU NX + e sits in round (..) / 16 on lane (..) % 16, and an agent's 16
        // lanes are one DPP row): the LLT / ladder chain starts without waiting for the LDS round trip of the other pass-2 results
        tile_gather_quu<NX * NX + NU * NX, R2>(res, quu, std::make_integer_sequence<int, NU * NU>{});
      }
    }
    if constexpr (!kLocalQuu && LPA != 16) {
      wave_sync();
      MAS_UNROLL for (int i = 0; i < NU * NU; ++i) quu[i] = T[TL::oQuu + i];
    }
    // ---- pass 3 (ilqr.hpp:172-192)
    {
      // LLT + x10 regularisation ladder (:172-182), the 16 lanes of an agent trying 16 consecutive
      // rungs at once: lane r adds reg_0 .. reg_{r-1} to the diagonal in the reference's order (so
      // its matrix is bit-identical to what the sequential ladder holds at rung r) and factorises it;
      // the first lane that succeeds is the rung the sequential loop would have stopped at.
      {
        double cur[NU * NU];
        MAS_UNROLL for (int i = 0; i < NU * NU; ++i) cur[i] = quu[i];
        // the additions stay sequential (floating-point addition is not associative): straight-line code
        // over the lane's precomputed addends, lane r applying the first r of them
        MAS_UNROLL for (int q = 0; q < LL - 1; ++q) {
          MAS_UNROLL for (int i = 0; i < NU; ++i) cur[i * NU + i] += rung_add[q];
        }
        int win = -1;
        if constexpr (SharedLadder) {
          // all agents of the launch move through the ladder together, 16 rungs per agreement
          double reg_next = 0.0;
          for (int base = 0, chunk = 0;; base += LL, ++chunk) {
            if (chunk > 0) {
              if (chunk == 1) { reg_next = 1e-6; for (int q = 0; q < rung_lane; ++q) reg_next *= 10.0; }
              for (int q = 0; q < LL; ++q) { MAS_UNROLL for (int i = 0; i < NU; ++i) cur[i * NU + i] += reg_next; reg_next *= 10.0; }
            }
            MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Lfac[i] = cur[i];
            const int ok = llt_compute<NU>(Lfac) == 0;
            unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
            if (!valid) field = 0xFFFFu;                                   // an idle slot never holds the others back
            MAS_UNROLL for (int sh = LPA; sh < 64; sh <<= 1) field &= static_cast<unsigned>(__shfl_xor(static_cast<int>(field), sh));
            unsigned agreed;
            if constexpr (WPB == 1) {
              agreed = ladder_agree(sync, t * LadderSync::kChunks + chunk, field, static_cast<int>(gridDim.x));
            } else if (gridDim.x == 1) {
              // one workgroup holds the whole problem: LDS and one barrier
              const unsigned par = agreement_parity; agreement_parity ^= 1u;
              if (tid == 0) blk_field[par][wave] = field;
              __syncthreads();
              unsigned bf = 0xFFFFu;
              MAS_UNROLL for (int wv = 0; wv < WPB; ++wv) bf &= blk_field[par][wv];
              agreed = bf;
            } else if (sync.combine != 0) {
              // workgroup first (LDS + one barrier), then one wavefront per workgroup across workgroups
              const unsigned par = agreement_parity; agreement_parity ^= 1u;
              if (tid == 0) blk_field[par][wave] = field;
              __syncthreads();
              if (wave == 0) {
                unsigned bf = 0xFFFFu;
                MAS_UNROLL for (int wv = 0; wv < WPB; ++wv) bf &= blk_field[par][wv];
                const unsigned g = ladder_agree(sync, t * LadderSync::kChunks + chunk, bf, static_cast<int>(gridDim.x));
                if (tid == 0) blk_agreed[par] = g;
              }
              __syncthreads();
              agreed = blk_agreed[par];
            } else {
              // every wavefront of the launch arrives on its own: no workgroup barrier on the way in or out
              agreed = ladder_agree(sync, t * LadderSync::kChunks + chunk, field, static_cast<int>(gridDim.x) * WPB);
            }
            if (agreed != 0u) { win = base + __builtin_ctz(agreed); break; }
            if (chunk + 1 >= LadderSync::kChunks || base + LL > kRegLadderMax) { win = base + LL - 1; break; }   // guard, never reached (see kRegLadderMax)
          }
        } else {
        // rungs 0..15 first, as straight-line code (three out of four steps need no regularisation at all and
        // the rest settle within 15 rungs on every workload of this repository) ...
        {
          MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Lfac[i] = cur[i];
          const int ok = llt_compute_straight<NU>(Lfac) == 0;
          const unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
          if (field != 0u) win = __builtin_ctz(field);
        }
        // ... and the general ladder only if some agent of the wavefront is still unsettled: its lanes go on
        // with rungs base+16 .. base+31, and so on
        if (__ballot(win < 0) != 0ull) {
          double reg_next = 0.0;                                       // this lane's next rung value
          int base = 0;
          for (;;) {
            if (win < 0 && base + LL > kRegLadderMax) win = base + LL - 1;     // guard, never reached (see kRegLadderMax)
            if (__ballot(win < 0) == 0ull) break;                              // every agent of the wavefront is settled
            if (win < 0) {
              if (base == 0) { reg_next = 1e-6; for (int q = 0; q < rung_lane; ++q) reg_next *= 10.0; }
              for (int q = 0; q < LL; ++q) { MAS_UNROLL for (int i = 0; i < NU; ++i) cur[i * NU + i] += reg_next; reg_next *= 10.0; }
              base += LL;
            }
            double Ltry[NU * NU];
            MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Ltry[i] = cur[i];
            const int ok = llt_compute<NU>(Ltry) == 0;
            const unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
            if (win < 0) {
              MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Lfac[i] = Ltry[i];
              if (field != 0u) win = base + __builtin_ctz(field);
            }
          }
        }
        }
        if (lane == 0 && (!SharedLadder || a == 0)) regs += win;         // the stacked solve counts its rungs once (agent 0)
        if constexpr (NU <= 2 && LPA == 16) {
          // the factor of the winning rung straight from its lane (ds_bpermute), see mas_backward_row_kernel
          const int src = ((tid & ~(LPA - 1)) + (win & (LL - 1))) * 4;
          MAS_UNROLL for (int e = 0; e < NU * NU; ++e) {
            if (e % NU >= e / NU) {
              const int lo32 = __builtin_amdgcn_ds_bpermute(src, __double2loint(Lfac[e])), hi32 = __builtin_amdgcn_ds_bpermute(src, __double2hiint(Lfac[e]));
              Lfac[e] = __hiloint2double(hi32, lo32);
            }
          }
          wave_sync();                                               // the pass-2 outputs before the reads below
        } else {
          if (lane == (win & (LL - 1))) { MAS_UNROLL for (int i = 0; i < NU * NU; ++i) T[TL::oL + i] = Lfac[i]; }
          wave_sync();                                               // also orders the pass-2 outputs before the reads below
          MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Lfac[i] = T[TL::oL + i];
        }
      }
      MAS_UNROLL for (int i = 0; i < NU; ++i) qu[i] = T[TL::oQu + i];
      llt_solve_identity<NU>(Lfac, ninv);                           // Quu_reg^-1, :183
      MAS_UNROLL for (int i = 0; i < NU * NU; ++i) ninv[i] = -ninv[i];
      MAS_UNROLL for (int m = 0; m < NU; ++m) kf[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * qu[q]; });   // k = -Quu^-1 q_u, :185
      // gain stores are unconditional straight-line code: every lane of an agent holds the same k (and every lane
      // with the same i the same K column), so the duplicates write identical bits to identical addresses, and an
      // idle slot writes into the unused gain rows of its retired agent: no divergent branch, no exec juggling.
      MAS_UNROLL for (int m = 0; m < NU; ++m) st_off(kff_t, goff[m], kf[m]);

      constexpr int RM = (NX * NX + LPA - 1) / LPA;                  // rounds over the v_xx entries
      MAS_UNROLL for (int r = 0; r < RM; ++r) {
        const int w = lane + r * LPA;
        const bool live = valid && (w < NX * NX);
        const int i = (w < NX * NX) ? (w % NX) : 0;
        const int j = (w < NX * NX) ? (w / NX) : 0;
        double quxi[NU], quxj[NU], Ki[NU], Kj[NU], ktqi[NU], ktqj[NU];
        MAS_UNROLL for (int m = 0; m < NU; ++m) { quxi[m] = T[TL::oQux + i * NU + m]; quxj[m] = T[TL::oQux + j * NU + m]; }
        const double qxx_own = T[TL::oQxx + j * NX + i], qxx_mir = T[TL::oQxx + i * NX + j], qx_i = T[TL::oQx + i];
        MAS_UNROLL for (int m = 0; m < NU; ++m) {                   // columns i and j of K = -Quu^-1 q_ux, :186
          Ki[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * quxi[q]; });
          Kj[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * quxj[q]; });
        }
        MAS_UNROLL for (int m = 0; m < NU; ++m) {                   // rows i and j of K^T q_uu (unregularised)
          ktqi[m] = sum_terms<NU>([&](int q) { return Ki[q] * quu[m * NU + q]; });
          ktqj[m] = sum_terms<NU>([&](int q) { return Kj[q] * quu[m * NU + q]; });
        }
        // own entry (i,j) and mirror entry (j,i) of v_xx, :190-191
        double s, own, mir;
        s = sum_terms<NU>([&](int m) { return Ki[m] * quxj[m]; });
        own = qxx_own + s;
        s = sum_terms<NU>([&](int m) { return quxi[m] * Kj[m]; });
        own = own + s;
        s = sum_terms<NU>([&](int m) { return ktqi[m] * Kj[m]; });
        own = own + s;
        s = sum_terms<NU>([&](int m) { return Kj[m] * quxi[m]; });
        mir = qxx_mir + s;
        s = sum_terms<NU>([&](int m) { return quxj[m] * Ki[m]; });
        mir = mir + s;
        s = sum_terms<NU>([&](int m) { return ktqj[m] * Ki[m]; });
        mir = mir + s;
        // v_xx = 0.5 (v_xx + v_xx^T) in the reference's aliased in-place order, :192: a lower or diagonal
        // entry averages the old values, an upper entry sees its (already rewritten) mirror
        const double sym = (i >= j) ? 0.5 * (own + mir) : 0.5 * (own + 0.5 * (mir + own));
        T[live ? (TL::oV + w) : TL::pad_at(TL::oV + w)] = sym;
        // the lanes of the first tile column (j == 0) also own gain column i and v_x entry i, :186,188-189
        {
          double vxi;
          s = sum_terms<NU>([&](int m) { return Ki[m] * qu[m]; });
          vxi = qx_i + s;
          s = sum_terms<NU>([&](int m) { return quxi[m] * kf[m]; });
          vxi = vxi + s;
          s = sum_terms<NU>([&](int m) { return ktqi[m] * kf[m]; });
          vxi = vxi + s;
          const bool owner = live && j == 0;
          T[owner ? (TL::oVx + i) : TL::pad_at(TL::oVx + w)] = vxi;
          MAS_UNROLL for (int m = 0; m < NU; ++m) st_off(Kfb_t, koff[r][m], Ki[m]);
        }
      }
    }
    // the barrier after the next publish orders these writes before pass 1 reads them
    kff_t -= static_cast<size_t>(NU) * d.Np; Kfb_t -= static_cast<size_t>(NU) * NX * d.Np;
  };
  step(d.T - 1);
  for (int t = d.T - 2; t >= 0; --t) step(t);
  if (counted && lane == 0) reg_events[a] += regs;
}

// ------------------------------------------------ backward, 16 lanes/agent, NX == 4: registers + DPP --
// The tile kernel above exchanges everything through LDS: four write -> wait -> read round trips per time step, ~40 % of a
// lone wavefront's step (SQ_WAIT_INST_LDS), and with four wavefronts per CU the LDS pipe itself is half busy.  For NX == 4
// the 16 lanes of an agent are exactly one DPP row holding the 4 x 4 value Hessian, lane (i, j) = i + 4 j owning V(i, j):
//   * V never touches LDS: column j of V is the lane's own quad (quad_perm broadcasts), v_x(k) lives in lane k of the row
//     (v_mov_b64_dpp row_newbcast: one instruction hands lane n of a row to all 16 lanes);
//   * the derivative tile never touches LDS either: every lane loads the two columns of A and B and the five scalars ITS
//     items need straight from HBM / L2 into registers, one step ahead (the 16 lanes of an agent hit the same 464-byte tile);
//   * q_uu and q_u reach all lanes by row_newbcast, so the LLT / ladder chain starts right after pass 2 without an LDS
//     round trip, and the pass-2 results that do cross quads (q_ux columns, the mirror entry of q_xx) travel through LDS in
//     the shadow of that chain.
// Two LDS round trips are left on the chain of a step: the rows of A^T V / B^T V (a transposition across quads, which no DPP
// row operation expresses) and the factor of the winning rung.  Item -> lane assignment (all items of a lane use ITS i, j):
//   pass 1   (A^T V)(i,j), (B^T V)(i,j) [i < NU], q_x(i), q_u(i) [i < NU]
//   pass 2   q_xx(i,j), q_ux(i,j) [i < NU], q_uu(i,j) [i, j < NU]
//   pass 3   as in the tile kernel
// Every output element is the same sequence of multiplies and adds as in the tile kernel and in BackwardBody.
__device__ __forceinline__ double2 ld2_off(const double* base, unsigned byte_off) { return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + byte_off); }
// out[e] = the value lane (I0 + e % NU, e / NU) of the row holds, e < NU * NU (q_uu); out[m] = lane m's (q_u); out[m] = lane m of the quad's (q_ux column)
template <int NU, int I0, int... E>
__device__ __forceinline__ void row_gather_block_at(double v, double* out, std::integer_sequence<int, E...>) { ((out[E] = row_bcast<(I0 + E % NU) + 4 * (E / NU)>(v)), ...); }
template <int... E>
__device__ __forceinline__ void row_gather_first(double v, double* out, std::integer_sequence<int, E...>) { ((out[E] = row_bcast<E>(v)), ...); }
template <int... E>
__device__ __forceinline__ void quad_gather(double v, double* out, std::integer_sequence<int, E...>) { ((out[E] = quad_bcast<E>(v)), ...); }

template <class M, bool SharedLadder = false, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB) mas_backward_row_kernel(Dims d, ModelParams P, const double* __restrict__ X, const double* __restrict__ D,
                                                              double* __restrict__ kff, double* __restrict__ Kfb,
                                                              long long* __restrict__ reg_events, const int* __restrict__ active, int lo, int n,
                                                              LadderSync sync) {
  using L = DerivLayout<M>;
  constexpr int NX = 4, NU = M::NU, ND = L::ND;
  static_assert(M::NX == 4 && NU >= 1 && NU <= 4, "one DPP row = the 4 x 4 value Hessian");
  static_assert(L::oB + NX * NU <= 2 * kTileChunk && ND >= 2 * kTileChunk, "the columns of A and B lie in full-width chunks (16-byte aligned pairs)");
  constexpr int LPA = 16, APW = 4, LL = 16;
  // LDS per agent (doubles): rows of A^T V [i*4 + k], rows of B^T V [m*4 + k], q_xx [i + 4 j], q_ux columns [j*NU + m], the
  // winning factor, 16 pad slots for lanes without an item (own slot per lane: no write conflicts); stride = 16 (mod 32)
  constexpr int oAtV = 0, oBtV = 16, oQxx = 32, oQux = 48, oL = 64, oVx = 80, oQx = 84, oPad = 96, STRIDE = 112;
  // NU <= 2 leaves lanes without an item in the second round of pass 2 (q_ux has 4 NU items): they take q_uu and q_x, which
  // saves a dot product in each of the two passes.  Second-round item of lane (i, j), all of the form bias + row . column with
  // the row read from LDS and the column loaded from the tile:
  //    i < NU           q_ux(i, j)      = l_ux + (B^T V)(i, :) . A(:, j)
  //    i >= 2, j < NU   q_uu(i-2, j)    = l_uu + (B^T V)(i-2, :) . B(:, j)          [i - 2 < NU]
  //    i >= 2, j >= 2   q_x(c)          = l_x  + v_x . A(:, c),  c = (i-2) + 2 (j-2)   (v_x kept in LDS for this)
  constexpr bool kMerged = NU <= 2;
  // Models whose control Hessian is analytic (the identity LQR: l_uu = 2 I, q_uu positive definite by construction) never climb the
  // ladder: trying rung 0 on every lane first saves them the rung additions and the exchange of the factor.  With finite-
  // difference Hessians a quarter of the steps need rungs 5-12 and a wavefront of four agents rarely passes as a whole: measured
  // on C3 the attempt costs as much as it saves (91.3 -> 93 us), so those models go straight to the 16-rung form.
  constexpr bool kRung0First = M::kAnalyticLuu && !SharedLadder;
#if defined(MAS_BWD_FACTOR_LDS)
  constexpr bool kFactorByPermute = false;
#else
  constexpr bool kFactorByPermute = NU <= 2;
#endif
  __shared__ double lds[WPB * APW * STRIDE];
  __shared__ unsigned blk_field[2][WPB > 1 ? WPB : 1];
  __shared__ unsigned blk_agreed[2];
  unsigned agreement_parity = 0;

  const int wave = static_cast<int>(threadIdx.x) >> 6;
  const int tid = static_cast<int>(threadIdx.x) & 63, lane = tid & (LPA - 1), slot = tid / LPA, rung_lane = lane;
  const int i = lane & 3, j = lane >> 2, iu = (i < NU) ? i : NU - 1, ju = (j < NU) ? j : NU - 1;
  const int a0 = lo + (static_cast<int>(blockIdx.x) * WPB + wave) * APW, a = a0 + slot;
  const bool valid = (a < n) && active[a] != 0;
  if (!SharedLadder && __ballot(valid) == 0ull) return;
  if (SharedLadder && active[lo] == 0) return;             // the stacked solve has ended (all flags fall together): uniform over the whole launch
  double* T = lds + (wave * APW + slot) * STRIDE;
  auto wave_sync = [] {
    if constexpr (WPB == 1) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  // terminal boundary condition, ilqr.hpp:92-102: lane 0 evaluates it, every lane picks up its own entries
  if (lane == 0) {
    double xT[NX], vx[NX], vxx[NX * NX];
    MAS_UNROLL for (int q = 0; q < NX; ++q) xT[q] = MAS_AT(X, d.T * NX + q, a);
    terminal_boundary<M>(xT, vx, vxx, P);
    MAS_UNROLL for (int q = 0; q < NX; ++q) T[oVx + q] = vx[q];
    MAS_UNROLL for (int q = 0; q < NX * NX; ++q) T[oQxx + q] = vxx[q];
  }
  wave_sync();
  double Vown = T[oQxx + lane], vxi = T[oVx + i];
  wave_sync();

  // this lane's operands of one tile: loop-invariant 32-bit byte offsets under a uniform base that steps back one time step
  auto elem_off = [&](int e) {
    const int c = e / kTileChunk, cw = (ND - c * kTileChunk < kTileChunk) ? (ND - c * kTileChunk) : kTileChunk;
    return static_cast<unsigned>((static_cast<size_t>(c) * kTileChunk * 64 + static_cast<size_t>(a % 64) * cw + static_cast<size_t>(e % kTileChunk)) * sizeof(double));
  };
  unsigned oAi = elem_off(L::oA + 4 * i), oAj = elem_off(L::oA + 4 * j), oBi = elem_off(L::oB + 4 * iu), oBj = elem_off(L::oB + 4 * ju);
  unsigned olx = elem_off(L::oLx + i), olu = elem_off(L::oLu + iu), olxx = elem_off(L::oLxx + lane), olux = elem_off(L::oLux + iu + NU * j),
           oluu = elem_off(L::oLuu + iu + NU * ju);
  // merged second round: its column (in the Bj slot) and bias (in the lux slot) are this lane's item's; its LDS row address
  const int item2 = (i < NU) ? 0 : ((i >= 2 && j < NU && i - 2 < NU) ? 1 : ((i >= 2 && j >= 2) ? 2 : 3));   // q_ux, q_uu, q_x, none
  const int c2 = (i - 2) + 2 * (j - 2);
  int r_row2 = oBtV + iu * 4;
  if constexpr (kMerged) {
    if (item2 == 0) { oBj = oAj; }
    else if (item2 == 1) { oBj = elem_off(L::oB + 4 * j); olux = elem_off(L::oLuu + (i - 2) + NU * j); r_row2 = oBtV + (i - 2) * 4; }
    else if (item2 == 2) { oBj = elem_off(L::oA + 4 * c2); olux = elem_off(L::oLx + c2); r_row2 = oVx; }
    else { oBj = oAj; r_row2 = oBtV; }
  }
  struct Tile { double2 Ai[2], Aj[2], Bi[2], Bj[2]; double lx, lu, lxx, lux, luu; };
  const size_t step_doubles = static_cast<size_t>(d.Np) * ND;
  // (the 64-agent block of a wavefront is uniform: said explicitly, so that the base pointer lives in scalar registers)
  const double* tiles_t = D + (static_cast<size_t>(d.T - 1) * (d.Np / 64) + static_cast<size_t>(__builtin_amdgcn_readfirstlane(a0 / 64))) * 64 * ND;
  auto fetch = [&](Tile& o) {
    o.Ai[0] = ld2_off(tiles_t, oAi); o.Ai[1] = ld2_off(tiles_t, oAi + 16);
    o.Aj[0] = ld2_off(tiles_t, oAj); o.Aj[1] = ld2_off(tiles_t, oAj + 16);
    o.Bi[0] = ld2_off(tiles_t, oBi); o.Bi[1] = ld2_off(tiles_t, oBi + 16);
    o.Bj[0] = ld2_off(tiles_t, oBj); o.Bj[1] = ld2_off(tiles_t, oBj + 16);
    o.lu = ld_off(tiles_t, olu); o.lxx = ld_off(tiles_t, olxx); o.lux = ld_off(tiles_t, olux);
    if constexpr (!kMerged) { o.lx = ld_off(tiles_t, olx); o.luu = ld_off(tiles_t, oluu); }
  };
  long long regs = 0;
  double rung_add[LL - 1];
  {
    double reg = 1e-6;
    MAS_UNROLL for (int q = 0; q < LL - 1; ++q) { rung_add[q] = (q < rung_lane) ? reg : -0.0; reg *= 10.0; }
  }
  unsigned goff[NU], koff[NU];
  MAS_UNROLL for (int m = 0; m < NU; ++m) goff[m] = static_cast<unsigned>((static_cast<size_t>(m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  MAS_UNROLL for (int m = 0; m < NU; ++m) koff[m] = static_cast<unsigned>((static_cast<size_t>(i * NU + m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  double* kff_t = kff + static_cast<size_t>(d.T - 1) * NU * d.Np;
  double* Kfb_t = Kfb + static_cast<size_t>(d.T - 1) * NU * NX * d.Np;
  // LDS addresses of this lane (a lane without an item writes its own pad slot)
  const int w_atv = oAtV + i * 4 + j, w_btv = (i < NU) ? (oBtV + i * 4 + j) : (oPad + lane);
  const int w_qux = (i < NU) ? (oQux + j * NU + i) : ((kMerged && item2 == 2) ? (oQx + c2) : (oPad + lane));
  const int w_vx = (j == 0) ? (oVx + i) : (oPad + lane);

  Tile tA, tB;
  fetch(tA);
  auto step = [&](int t, const Tile& c, Tile& nx) {
    if (t > 0) tiles_t -= step_doubles;              // uniform; at t == 0 the (unused) prefetch re-reads step 0
    MAS_OPAQUE32(oAi); MAS_OPAQUE32(oAj); MAS_OPAQUE32(oBi); MAS_OPAQUE32(oBj);
    MAS_OPAQUE32(olx); MAS_OPAQUE32(olu); MAS_OPAQUE32(olxx); MAS_OPAQUE32(olux); MAS_OPAQUE32(oluu);
    MAS_UNROLL for (int m = 0; m < NU; ++m) { MAS_OPAQUE32(goff[m]); MAS_OPAQUE32(koff[m]); }
    fetch(nx);
    const double Ai[NX] = {c.Ai[0].x, c.Ai[0].y, c.Ai[1].x, c.Ai[1].y}, Aj[NX] = {c.Aj[0].x, c.Aj[0].y, c.Aj[1].x, c.Aj[1].y};
    const double Bi[NX] = {c.Bi[0].x, c.Bi[0].y, c.Bi[1].x, c.Bi[1].y}, Bj[NX] = {c.Bj[0].x, c.Bj[0].y, c.Bj[1].x, c.Bj[1].y};

    // ---- pass 1 (ilqr.hpp:115-116 and the first factors of :117-119)
    double qx_i, qu_own;
    {
      const double Vc[NX] = {quad_bcast<0>(Vown), quad_bcast<1>(Vown), quad_bcast<2>(Vown), quad_bcast<3>(Vown)};   // V(k, j)
      const double vxk[NX] = {row_bcast<0>(vxi), row_bcast<1>(vxi), row_bcast<2>(vxi), row_bcast<3>(vxi)};         // v_x(k)
      double s;
      s = sum_terms<NX>([&](int k) { return Ai[k] * Vc[k]; });
      T[w_atv] = s;
      s = sum_terms<NX>([&](int k) { return Bi[k] * Vc[k]; });
      T[w_btv] = s;
      if constexpr (!kMerged) {
        s = sum_terms<NX>([&](int k) { return Ai[k] * vxk[k]; });
        qx_i = c.lx + s;
      }
      s = sum_terms<NX>([&](int k) { return Bi[k] * vxk[k]; });
      qu_own = c.lu + s;
    }
    wave_sync();
    // ---- pass 2 (second factors of ilqr.hpp:117-119)
    double quu[NU * NU], qu[NU], quxj[NU], Lfac[NU * NU], ninv[NU * NU], kf[NU], qxx_own;
    {
      double ra[NX], rb[NX];
      MAS_UNROLL for (int k = 0; k < NX; ++k) { ra[k] = T[oAtV + i * 4 + k]; rb[k] = T[r_row2 + k]; }
      double s;
      s = sum_terms<NX>([&](int k) { return ra[k] * Aj[k]; });
      qxx_own = c.lxx + s;
      T[oQxx + lane] = qxx_own;
      if constexpr (kMerged) {
        s = sum_terms<NX>([&](int k) { return rb[k] * Bj[k]; });          // this lane's second-round item (see kMerged)
        const double item = c.lux + s;
        T[w_qux] = item;
        row_gather_block_at<NU, 2>(item, quu, std::make_integer_sequence<int, NU * NU>{});   // q_uu(m, n) from lane (2 + m, n)
        quad_gather(item, quxj, std::make_integer_sequence<int, NU>{});
      } else {
        s = sum_terms<NX>([&](int k) { return rb[k] * Aj[k]; });
        const double qux_own = c.lux + s;
        s = sum_terms<NX>([&](int k) { return rb[k] * Bj[k]; });
        const double quu_own = c.luu + s;
        T[w_qux] = qux_own;
        row_gather_block_at<NU, 0>(quu_own, quu, std::make_integer_sequence<int, NU * NU>{});
        quad_gather(qux_own, quxj, std::make_integer_sequence<int, NU>{});
      }
      row_gather_first(qu_own, qu, std::make_integer_sequence<int, NU>{});
    }
    // ---- pass 3 (ilqr.hpp:172-192)
    {
      {
        double cur[NU * NU];
        MAS_UNROLL for (int e = 0; e < NU * NU; ++e) cur[e] = quu[e];
        int win = -1;
        bool rung0_everywhere = false;
        if constexpr (kRung0First) {
          // every lane factorises q_uu as it is; if that succeeds for every agent of the wavefront -- a wave-uniform outcome --
          // each lane already HOLDS the winning factor: no rung additions (2 x 15 per diagonal entry), no exchange of NU^2 values
          MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = cur[e];
          const int ok0 = llt_compute_straight<NU>(Lfac) == 0;
          rung0_everywhere = __ballot(ok0 || !valid) == ~0ull;
          if (rung0_everywhere) win = 0;
        }
        if (!rung0_everywhere) {
        MAS_UNROLL for (int q = 0; q < LL - 1; ++q) {
          MAS_UNROLL for (int e = 0; e < NU; ++e) cur[e * NU + e] += rung_add[q];
        }
        if constexpr (SharedLadder) {
          double reg_next = 0.0;
          for (int base = 0, chunk = 0;; base += LL, ++chunk) {
            if (chunk > 0) {
              if (chunk == 1) { reg_next = 1e-6; for (int q = 0; q < rung_lane; ++q) reg_next *= 10.0; }
              for (int q = 0; q < LL; ++q) { MAS_UNROLL for (int e = 0; e < NU; ++e) cur[e * NU + e] += reg_next; reg_next *= 10.0; }
            }
            MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = cur[e];
            const int ok = llt_compute<NU>(Lfac) == 0;
            unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
            if (!valid) field = 0xFFFFu;
            MAS_UNROLL for (int sh = LPA; sh < 64; sh <<= 1) field &= static_cast<unsigned>(__shfl_xor(static_cast<int>(field), sh));
            unsigned agreed;
            if constexpr (WPB == 1) {
              agreed = ladder_agree(sync, t * LadderSync::kChunks + chunk, field, static_cast<int>(gridDim.x));
            } else if (gridDim.x == 1) {
              // one workgroup holds the whole problem: LDS and one barrier
              const unsigned par = agreement_parity; agreement_parity ^= 1u;
              if (tid == 0) blk_field[par][wave] = field;
              __syncthreads();
              unsigned bf = 0xFFFFu;
              MAS_UNROLL for (int wv = 0; wv < WPB; ++wv) bf &= blk_field[par][wv];
              agreed = bf;
            } else if (sync.combine != 0) {
              // workgroup first (LDS + one barrier), then one wavefront per workgroup across workgroups
              const unsigned par = agreement_parity; agreement_parity ^= 1u;
              if (tid == 0) blk_field[par][wave] = field;
              __syncthreads();
              if (wave == 0) {
                unsigned bf = 0xFFFFu;
                MAS_UNROLL for (int wv = 0; wv < WPB; ++wv) bf &= blk_field[par][wv];
                const unsigned g = ladder_agree(sync, t * LadderSync::kChunks + chunk, bf, static_cast<int>(gridDim.x));
                if (tid == 0) blk_agreed[par] = g;
              }
              __syncthreads();
              agreed = blk_agreed[par];
            } else {
              // every wavefront of the launch arrives on its own: no workgroup barrier on the way in or out
              agreed = ladder_agree(sync, t * LadderSync::kChunks + chunk, field, static_cast<int>(gridDim.x) * WPB);
            }
            if (agreed != 0u) { win = base + __builtin_ctz(agreed); break; }
            if (chunk + 1 >= LadderSync::kChunks || base + LL > kRegLadderMax) { win = base + LL - 1; break; }
          }
        } else {
          {
            MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = cur[e];
            const int ok = llt_compute_straight<NU>(Lfac) == 0;
            const unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
            if (field != 0u) win = __builtin_ctz(field);
          }
          if (__ballot(win < 0) != 0ull) {
            double reg_next = 0.0;
            int base = 0;
            for (;;) {
              if (win < 0 && base + LL > kRegLadderMax) win = base + LL - 1;
              if (__ballot(win < 0) == 0ull) break;
              if (win < 0) {
                if (base == 0) { reg_next = 1e-6; for (int q = 0; q < rung_lane; ++q) reg_next *= 10.0; }
                for (int q = 0; q < LL; ++q) { MAS_UNROLL for (int e = 0; e < NU; ++e) cur[e * NU + e] += reg_next; reg_next *= 10.0; }
                base += LL;
              }
              double Ltry[NU * NU];
              MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Ltry[e] = cur[e];
              const int ok = llt_compute<NU>(Ltry) == 0;
              const unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
              if (win < 0) {
                MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = Ltry[e];
                if (field != 0u) win = base + __builtin_ctz(field);
              }
            }
          }
        }
        if (lane == 0 && (!SharedLadder || a == 0)) regs += win;
        if constexpr (kFactorByPermute) {
          // the factor of the winning rung straight from its lane (ds_bpermute: the LDS crossbar without a write -> wait -> read):
          // three entries of a 2 x 2 factor (the strictly upper one is never read), six dword permutes
          const int src = ((tid & ~(LPA - 1)) + (win & (LL - 1))) * 4;
          MAS_UNROLL for (int e = 0; e < NU * NU; ++e) {
            if (e % NU >= e / NU) {                                  // column-major: row >= column
              const int lo32 = __builtin_amdgcn_ds_bpermute(src, __double2loint(Lfac[e])), hi32 = __builtin_amdgcn_ds_bpermute(src, __double2hiint(Lfac[e]));
              Lfac[e] = __hiloint2double(hi32, lo32);
            }
          }
          wave_sync();                                               // the pass-2 outputs before the reads below
        } else {
        if (lane == (win & (LL - 1))) { MAS_UNROLL for (int e = 0; e < NU * NU; ++e) T[oL + e] = Lfac[e]; }
        wave_sync();                                                 // also orders the pass-2 outputs before the reads below
        MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = T[oL + e];
        }
        } else {
          wave_sync();                                               // the pass-2 outputs before the reads below
        }
      }
      if constexpr (NU > 2) llt_solve_identity_by_columns<NU>(Lfac, i < NU ? i : NU - 1, ninv);   // lane (i, j) solves column i
      else llt_solve_identity<NU>(Lfac, ninv);
      MAS_UNROLL for (int e = 0; e < NU * NU; ++e) ninv[e] = -ninv[e];
      MAS_UNROLL for (int m = 0; m < NU; ++m) kf[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * qu[q]; });
      MAS_UNROLL for (int m = 0; m < NU; ++m) st_off(kff_t, goff[m], kf[m]);
      {
        double quxi[NU], Ki[NU], Kj[NU], ktqi[NU], ktqj[NU];
        MAS_UNROLL for (int m = 0; m < NU; ++m) quxi[m] = T[oQux + i * NU + m];
        const double qxx_mir = T[oQxx + j + 4 * i];
        if constexpr (kMerged) qx_i = T[oQx + i];
        MAS_UNROLL for (int m = 0; m < NU; ++m) {
          Ki[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * quxi[q]; });
          Kj[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * quxj[q]; });
        }
        MAS_UNROLL for (int m = 0; m < NU; ++m) {
          ktqi[m] = sum_terms<NU>([&](int q) { return Ki[q] * quu[m * NU + q]; });
          ktqj[m] = sum_terms<NU>([&](int q) { return Kj[q] * quu[m * NU + q]; });
        }
        double s, own, mir;
        s = sum_terms<NU>([&](int m) { return Ki[m] * quxj[m]; });
        own = qxx_own + s;
        s = sum_terms<NU>([&](int m) { return quxi[m] * Kj[m]; });
        own = own + s;
        s = sum_terms<NU>([&](int m) { return ktqi[m] * Kj[m]; });
        own = own + s;
        s = sum_terms<NU>([&](int m) { return Kj[m] * quxi[m]; });
        mir = qxx_mir + s;
        s = sum_terms<NU>([&](int m) { return quxj[m] * Ki[m]; });
        mir = mir + s;
        s = sum_terms<NU>([&](int m) { return ktqj[m] * Ki[m]; });
        mir = mir + s;
        Vown = (i >= j) ? 0.5 * (own + mir) : 0.5 * (own + 0.5 * (mir + own));      // aliased in-place symmetrisation, :192
        double vx_new;
        s = sum_terms<NU>([&](int m) { return Ki[m] * qu[m]; });
        vx_new = qx_i + s;
        s = sum_terms<NU>([&](int m) { return quxi[m] * kf[m]; });
        vx_new = vx_new + s;
        s = sum_terms<NU>([&](int m) { return ktqi[m] * kf[m]; });
        vxi = vx_new + s;
        if constexpr (kMerged) T[w_vx] = vxi;                       // pass 2 of the next step reads v_x as an LDS row
        MAS_UNROLL for (int m = 0; m < NU; ++m) st_off(Kfb_t, koff[m], Ki[m]);
      }
    }
    // (the LDS instructions of one wavefront execute in order: the next step's pass-1 writes cannot overtake these reads)
    kff_t -= static_cast<size_t>(NU) * d.Np; Kfb_t -= static_cast<size_t>(NU) * NX * d.Np;
  };
  int t = d.T - 1;
  for (; t >= 1; t -= 2) { step(t, tA, tB); step(t - 1, tB, tA); }
  if (t == 0) step(0, tA, tB);
  if (valid && lane == 0) reg_events[a] += regs;
}

// ------------------------------------- centralised backward pass, NX == 4: the agreement's latency behind a speculative step --
// mas_backward_row_kernel<M, true, WPB> does, per time step, 0.9 us of work and then waits ~1.5 us for the rung every agent of the
// stacked problem agrees on (an arrive-and-wait among the launch's workgroups costs 0.9 us bare, xcd_barrier ubench).  On C2 three
// steps out of four agree on rung 0 (oracle: 296 of 400).  This kernel therefore posts its rung mask for step t and, if rung 0
// works for all ITS agents, goes on at once as if rung 0 had been agreed: pass 3 of step t with rung 0, then passes 1-2 and the
// ladder of step t-1 -- the work of one full step -- and only then reads the agreement of step t.  If it is rung 0 (the common
// case) the speculative work stands and the mask of step t-1 is posted; otherwise pass 3 of step t and the front of step t-1 are
// redone with the agreed rung.  Nothing speculative is ever published: a wavefront posts the mask of step t-1 only after step t
// is settled, so every word of LadderSync is written once per wavefront exactly as in the plain kernel, and the gains of a
// step are simply stored again when it is redone.  Cost per step ~ max(agreement, work) instead of their sum.
// A step is split in front(t) = passes 1, 2 + the 16 rung candidates of the ladder (everything that does not depend on the
// agreed rung of step t) and back(t, rung) = factor exchange + pass 3.  front(t-1) must not destroy what a redo of back(t)
// reads: its register outputs ping-pong between two sets and its LDS outputs (q_xx, q_ux, q_x) between two regions by parity of t.
template <class M, int WPB>
__global__ void __launch_bounds__(64 * WPB) __attribute__((amdgpu_waves_per_eu(1, 2)))   // two Front sets + two tiles in registers: ~250 VGPRs, never more than one wavefront per SIMD anyway
mas_backward_row_pipelined_kernel(Dims d, ModelParams P, const double* __restrict__ X, const double* __restrict__ D,
                                                                        double* __restrict__ kff, double* __restrict__ Kfb,
                                                                        long long* __restrict__ reg_events, const int* __restrict__ active, int lo, int n,
                                                                        LadderSync sync) {
  using L = DerivLayout<M>;
  constexpr int NX = 4, NU = M::NU, ND = L::ND;
  static_assert(M::NX == 4 && NU >= 1 && NU <= 4, "one DPP row = the 4 x 4 value Hessian");
  static_assert(L::oB + NX * NU <= 2 * kTileChunk && ND >= 2 * kTileChunk, "the columns of A and B lie in full-width chunks (16-byte aligned pairs)");
  constexpr int LPA = 16, APW = 4, LL = 16;
  // LDS per agent (doubles): as in mas_backward_row_kernel, with the pass-2 outputs twice (parity of t)
  constexpr int oAtV = 0, oBtV = 16, oR2 = 32, kR2 = 40 /* q_xx 16, q_ux 16, q_x 4 (+4) */, rQxx = 0, rQux = 16, rQx = 32,
                oL = oR2 + 2 * kR2, oVx = oL + 16, oPad = oVx + 16, STRIDE = 176;
  static_assert(oPad + 16 <= STRIDE && STRIDE % 32 == 16, "tile stride");
  constexpr bool kMerged = NU <= 2;
  __shared__ double lds[WPB * APW * STRIDE];
  __shared__ unsigned blk_field[2][WPB > 1 ? WPB : 1];
  __shared__ unsigned blk_agreed[2];
  unsigned agreement_parity = 0;

  const int wave = static_cast<int>(threadIdx.x) >> 6;
  const int tid = static_cast<int>(threadIdx.x) & 63, lane = tid & (LPA - 1), slot = tid / LPA, rung_lane = lane;
  const int i = lane & 3, j = lane >> 2, iu = (i < NU) ? i : NU - 1, ju = (j < NU) ? j : NU - 1;
  if (static_cast<int>(blockIdx.x) % sync.xcd_stride != 0) return;    // (one-XCD placement: see LadderSync::xcd_stride)
  const int group = static_cast<int>(blockIdx.x) / sync.xcd_stride, n_groups = static_cast<int>(gridDim.x) / sync.xcd_stride;
  const int a0 = lo + (group * WPB + wave) * APW, a = a0 + slot;
  const bool valid = (a < n) && active[a] != 0;
  if (active[lo] == 0) return;                               // the stacked solve has ended (all flags fall together): uniform over the whole launch
  double* T = lds + (wave * APW + slot) * STRIDE;
  auto wave_sync = [] {
    if constexpr (WPB == 1) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };

  if (lane == 0) {                                           // terminal boundary condition, ilqr.hpp:92-102
    double xT[NX], vx[NX], vxx[NX * NX];
    MAS_UNROLL for (int q = 0; q < NX; ++q) xT[q] = MAS_AT(X, d.T * NX + q, a);
    terminal_boundary<M>(xT, vx, vxx, P);
    MAS_UNROLL for (int q = 0; q < NX; ++q) T[oVx + q] = vx[q];
    MAS_UNROLL for (int q = 0; q < NX * NX; ++q) T[oR2 + rQxx + q] = vxx[q];
  }
  wave_sync();
  double Vown = T[oR2 + rQxx + lane], vxi = T[oVx + i];
  wave_sync();

  auto elem_off = [&](int e) {
    const int c = e / kTileChunk, cw = (ND - c * kTileChunk < kTileChunk) ? (ND - c * kTileChunk) : kTileChunk;
    return static_cast<unsigned>((static_cast<size_t>(c) * kTileChunk * 64 + static_cast<size_t>(a % 64) * cw + static_cast<size_t>(e % kTileChunk)) * sizeof(double));
  };
  unsigned oAi = elem_off(L::oA + 4 * i), oAj = elem_off(L::oA + 4 * j), oBi = elem_off(L::oB + 4 * iu), oBj = elem_off(L::oB + 4 * ju);
  unsigned olx = elem_off(L::oLx + i), olu = elem_off(L::oLu + iu), olxx = elem_off(L::oLxx + lane), olux = elem_off(L::oLux + iu + NU * j),
           oluu = elem_off(L::oLuu + iu + NU * ju);
  const int item2 = (i < NU) ? 0 : ((i >= 2 && j < NU && i - 2 < NU) ? 1 : ((i >= 2 && j >= 2) ? 2 : 3));   // q_ux, q_uu, q_x, none (see mas_backward_row_kernel)
  const int c2 = (i - 2) + 2 * (j - 2);
  int r_row2 = oBtV + iu * 4;
  if constexpr (kMerged) {
    if (item2 == 0) { oBj = oAj; }
    else if (item2 == 1) { oBj = elem_off(L::oB + 4 * j); olux = elem_off(L::oLuu + (i - 2) + NU * j); r_row2 = oBtV + (i - 2) * 4; }
    else if (item2 == 2) { oBj = elem_off(L::oA + 4 * c2); olux = elem_off(L::oLx + c2); r_row2 = oVx; }
    else { oBj = oAj; r_row2 = oBtV; }
  }
  struct Tile { double2 Ai[2], Aj[2], Bi[2], Bj[2]; double lx, lu, lxx, lux, luu; };
  const size_t step_doubles = static_cast<size_t>(d.Np) * ND;
  const double* tiles_0 = D + static_cast<size_t>(__builtin_amdgcn_readfirstlane(a0 / 64)) * 64 * ND;       // this wavefront's block of step 0 (uniform)
  auto fetch = [&](Tile& o, int t) {
    const double* tiles_t = tiles_0 + static_cast<size_t>(t) * step_doubles;
    o.Ai[0] = ld2_off(tiles_t, oAi); o.Ai[1] = ld2_off(tiles_t, oAi + 16);
    o.Aj[0] = ld2_off(tiles_t, oAj); o.Aj[1] = ld2_off(tiles_t, oAj + 16);
    o.Bi[0] = ld2_off(tiles_t, oBi); o.Bi[1] = ld2_off(tiles_t, oBi + 16);
    o.Bj[0] = ld2_off(tiles_t, oBj); o.Bj[1] = ld2_off(tiles_t, oBj + 16);
    o.lu = ld_off(tiles_t, olu); o.lxx = ld_off(tiles_t, olxx); o.lux = ld_off(tiles_t, olux);
    if constexpr (!kMerged) { o.lx = ld_off(tiles_t, olx); o.luu = ld_off(tiles_t, oluu); }
  };
  long long regs = 0;
  double rung_add[LL - 1];
  {
    double reg = 1e-6;
    MAS_UNROLL for (int q = 0; q < LL - 1; ++q) { rung_add[q] = (q < rung_lane) ? reg : -0.0; reg *= 10.0; }
  }
  unsigned goff[NU], koff[NU];
  MAS_UNROLL for (int m = 0; m < NU; ++m) goff[m] = static_cast<unsigned>((static_cast<size_t>(m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  MAS_UNROLL for (int m = 0; m < NU; ++m) koff[m] = static_cast<unsigned>((static_cast<size_t>(i * NU + m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  const int w_atv = oAtV + i * 4 + j, w_btv = (i < NU) ? (oBtV + i * 4 + j) : (oPad + lane);
  const int w_qux_rel = (i < NU) ? (rQux + j * NU + i) : ((kMerged && item2 == 2) ? (rQx + c2) : -1);     // relative to the parity's region; -1: pad
  const int w_vx = (j == 0) ? (oVx + i) : (oPad + lane);

  // what front(t) hands to back(t): this lane's rung candidate (matrix and factor), the broadcast pass-2 results, the mask
  struct Front { double quu[NU * NU], qu[NU], quxj[NU], qxx_own, qx_i, cur[NU * NU], Lcand[NU * NU]; unsigned field; };

  auto front = [&](const Tile& c, int par, Front& o) {
    const int R2 = oR2 + par * kR2;
    const double Ai[NX] = {c.Ai[0].x, c.Ai[0].y, c.Ai[1].x, c.Ai[1].y}, Aj[NX] = {c.Aj[0].x, c.Aj[0].y, c.Aj[1].x, c.Aj[1].y};
    const double Bi[NX] = {c.Bi[0].x, c.Bi[0].y, c.Bi[1].x, c.Bi[1].y}, Bj[NX] = {c.Bj[0].x, c.Bj[0].y, c.Bj[1].x, c.Bj[1].y};
    double qu_own;
    {
      const double Vc[NX] = {quad_bcast<0>(Vown), quad_bcast<1>(Vown), quad_bcast<2>(Vown), quad_bcast<3>(Vown)};
      const double vxk[NX] = {row_bcast<0>(vxi), row_bcast<1>(vxi), row_bcast<2>(vxi), row_bcast<3>(vxi)};
      T[w_atv] = sum_terms<NX>([&](int k) { return Ai[k] * Vc[k]; });
      T[w_btv] = sum_terms<NX>([&](int k) { return Bi[k] * Vc[k]; });
      if constexpr (!kMerged) o.qx_i = c.lx + sum_terms<NX>([&](int k) { return Ai[k] * vxk[k]; });
      qu_own = c.lu + sum_terms<NX>([&](int k) { return Bi[k] * vxk[k]; });
    }
    wave_sync();
    {
      double ra[NX], rb[NX];
      MAS_UNROLL for (int k = 0; k < NX; ++k) { ra[k] = T[oAtV + i * 4 + k]; rb[k] = T[r_row2 + k]; }
      o.qxx_own = c.lxx + sum_terms<NX>([&](int k) { return ra[k] * Aj[k]; });
      T[R2 + rQxx + lane] = o.qxx_own;
      if constexpr (kMerged) {
        const double item = c.lux + sum_terms<NX>([&](int k) { return rb[k] * Bj[k]; });
        T[w_qux_rel >= 0 ? R2 + w_qux_rel : oPad + lane] = item;
        row_gather_block_at<NU, 2>(item, o.quu, std::make_integer_sequence<int, NU * NU>{});
        quad_gather(item, o.quxj, std::make_integer_sequence<int, NU>{});
      } else {
        const double qux_own = c.lux + sum_terms<NX>([&](int k) { return rb[k] * Aj[k]; });
        const double quu_own = c.luu + sum_terms<NX>([&](int k) { return rb[k] * Bj[k]; });
        T[w_qux_rel >= 0 ? R2 + w_qux_rel : oPad + lane] = qux_own;
        row_gather_block_at<NU, 0>(quu_own, o.quu, std::make_integer_sequence<int, NU * NU>{});
        quad_gather(qux_own, o.quxj, std::make_integer_sequence<int, NU>{});
      }
      row_gather_first(qu_own, o.qu, std::make_integer_sequence<int, NU>{});
    }
    // the 16 rung candidates of the ladder (ilqr.hpp:172-182), lane r holding rung r
    MAS_UNROLL for (int e = 0; e < NU * NU; ++e) o.cur[e] = o.quu[e];
    MAS_UNROLL for (int q = 0; q < LL - 1; ++q) {
      MAS_UNROLL for (int e = 0; e < NU; ++e) o.cur[e * NU + e] += rung_add[q];
    }
    MAS_UNROLL for (int e = 0; e < NU * NU; ++e) o.Lcand[e] = o.cur[e];
    const int ok = llt_compute<NU>(o.Lcand) == 0;
    unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
    if (!valid) field = 0xFFFFu;
    MAS_UNROLL for (int sh = LPA; sh < 64; sh <<= 1) field &= static_cast<unsigned>(__shfl_xor(static_cast<int>(field), sh));
    o.field = field;
  };

  // pass 3 of step t with the factor of lane `src` (rung src of the current chunk): gains, value function
  auto back = [&](int t, int par, const Front& f, const double* Lsrc, int src) {
    const int R2 = oR2 + par * kR2;
    double Lfac[NU * NU], ninv[NU * NU], kf[NU];
    if constexpr (NU <= 2) {                                     // the factor straight from its lane (see mas_backward_row_kernel)
      const int srcaddr = ((tid & ~(LPA - 1)) + src) * 4;
      MAS_UNROLL for (int e = 0; e < NU * NU; ++e) {
        Lfac[e] = Lsrc[e];
        if (e % NU >= e / NU) {
          const int lo32 = __builtin_amdgcn_ds_bpermute(srcaddr, __double2loint(Lsrc[e])), hi32 = __builtin_amdgcn_ds_bpermute(srcaddr, __double2hiint(Lsrc[e]));
          Lfac[e] = __hiloint2double(hi32, lo32);
        }
      }
      wave_sync();                                               // the pass-2 outputs before the reads below
    } else {
      if (lane == src) { MAS_UNROLL for (int e = 0; e < NU * NU; ++e) T[oL + e] = Lsrc[e]; }
      wave_sync();                                               // also orders the pass-2 outputs before the reads below
      MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = T[oL + e];
    }
    double* kff_t = kff + static_cast<size_t>(t) * NU * d.Np;
    double* Kfb_t = Kfb + static_cast<size_t>(t) * NU * NX * d.Np;
    if constexpr (NU > 2) llt_solve_identity_by_columns<NU>(Lfac, i < NU ? i : NU - 1, ninv);   // lane (i, j) solves column i
    else llt_solve_identity<NU>(Lfac, ninv);
    MAS_UNROLL for (int e = 0; e < NU * NU; ++e) ninv[e] = -ninv[e];
    MAS_UNROLL for (int m = 0; m < NU; ++m) kf[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * f.qu[q]; });
    MAS_UNROLL for (int m = 0; m < NU; ++m) { unsigned o = goff[m]; MAS_OPAQUE32(o); st_off(kff_t, o, kf[m]); }
    double quxi[NU], Ki[NU], Kj[NU], ktqi[NU], ktqj[NU];
    MAS_UNROLL for (int m = 0; m < NU; ++m) quxi[m] = T[R2 + rQux + i * NU + m];
    const double qxx_mir = T[R2 + rQxx + j + 4 * i];
    double qx_i = f.qx_i;
    if constexpr (kMerged) qx_i = T[R2 + rQx + i];
    MAS_UNROLL for (int m = 0; m < NU; ++m) {
      Ki[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * quxi[q]; });
      Kj[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * f.quxj[q]; });
    }
    MAS_UNROLL for (int m = 0; m < NU; ++m) {
      ktqi[m] = sum_terms<NU>([&](int q) { return Ki[q] * f.quu[m * NU + q]; });
      ktqj[m] = sum_terms<NU>([&](int q) { return Kj[q] * f.quu[m * NU + q]; });
    }
    double own, mir;
    own = f.qxx_own + sum_terms<NU>([&](int m) { return Ki[m] * f.quxj[m]; });
    own = own + sum_terms<NU>([&](int m) { return quxi[m] * Kj[m]; });
    own = own + sum_terms<NU>([&](int m) { return ktqi[m] * Kj[m]; });
    mir = qxx_mir + sum_terms<NU>([&](int m) { return Kj[m] * quxi[m]; });
    mir = mir + sum_terms<NU>([&](int m) { return f.quxj[m] * Ki[m]; });
    mir = mir + sum_terms<NU>([&](int m) { return ktqj[m] * Ki[m]; });
    Vown = (i >= j) ? 0.5 * (own + mir) : 0.5 * (own + 0.5 * (mir + own));      // aliased in-place symmetrisation, :192
    double vx_new = qx_i + sum_terms<NU>([&](int m) { return Ki[m] * f.qu[m]; });
    vx_new = vx_new + sum_terms<NU>([&](int m) { return quxi[m] * kf[m]; });
    vxi = vx_new + sum_terms<NU>([&](int m) { return ktqi[m] * kf[m]; });
    if constexpr (kMerged) T[w_vx] = vxi;
    MAS_UNROLL for (int m = 0; m < NU; ++m) { unsigned o = koff[m]; MAS_OPAQUE32(o); st_off(Kfb_t, o, Ki[m]); }
  };

  // the agreement, split in two: arrive (every wavefront / one per workgroup) and wait
  const bool one_group = n_groups == 1;
  auto post = [&](int word, unsigned field) {
    if constexpr (WPB > 1) {
      if (tid == 0) blk_field[agreement_parity][wave] = field;
      __syncthreads();
      if (wave == 0) {
        unsigned bf = 0xFFFFu;
        MAS_UNROLL for (int wv = 0; wv < WPB; ++wv) bf &= blk_field[agreement_parity][wv];
        if (one_group) { if (tid == 0) blk_agreed[agreement_parity] = bf; }
        else ladder_post(sync, word, bf);
      }
    } else {
      ladder_post(sync, word, field);
    }
  };
  auto wait = [&](int word) -> unsigned {
    unsigned agreed;
    if constexpr (WPB > 1) {
      if (wave == 0 && !one_group) { const unsigned g = ladder_wait(sync, word, n_groups); if (tid == 0) blk_agreed[agreement_parity] = g; }
      __syncthreads();
      agreed = blk_agreed[agreement_parity];
      agreement_parity ^= 1u;
    } else {
      agreed = ladder_wait(sync, word, n_groups);
    }
    return agreed;
  };
  // rungs beyond the first sixteen: all agents move on together, 16 rungs per agreement, synchronously (never on the shipped workloads)
  auto settle_beyond = [&](int t, Front& f, int& win) {
    double reg_next = 1e-6;
    for (int q = 0; q < rung_lane; ++q) reg_next *= 10.0;
    for (int base = LL, chunk = 1;; base += LL, ++chunk) {
      for (int q = 0; q < LL; ++q) { MAS_UNROLL for (int e = 0; e < NU; ++e) f.cur[e * NU + e] += reg_next; reg_next *= 10.0; }
      MAS_UNROLL for (int e = 0; e < NU * NU; ++e) f.Lcand[e] = f.cur[e];
      const int ok = llt_compute<NU>(f.Lcand) == 0;
      unsigned field = static_cast<unsigned>(__ballot(ok) >> (slot * LPA)) & 0xFFFFu;
      if (!valid) field = 0xFFFFu;
      MAS_UNROLL for (int sh = LPA; sh < 64; sh <<= 1) field &= static_cast<unsigned>(__shfl_xor(static_cast<int>(field), sh));
      post(t * LadderSync::kChunks + chunk, field);
      const unsigned agreed = wait(t * LadderSync::kChunks + chunk);
      if (agreed != 0u) { win = base + __builtin_ctz(agreed); return; }
      if (chunk + 1 >= LadderSync::kChunks || base + LL > kRegLadderMax) { win = base + LL - 1; return; }
    }
  };

  // two register sets (Front, Tile) alternate from step to step; written out by name -- indexing them with the parity of t would
  // put them in scratch memory
  Tile tileA, tileB;
  Front frA, frB;
  // one step: `cur` = front(t) (settled, posted), `tile_next` holds tile t-1, `tile_refill` held tile t and is refilled with t-2
  auto iterate = [&](int t, int par, Front& cur, Front& nxt, Tile& tile_next, Tile& tile_refill) {
    if (t > 1) fetch(tile_refill, t - 2);                       // tile t is no longer needed (front(t) is settled): a whole step ahead of its use
    const bool speculate = (cur.field & 1u) != 0u;             // rung 0 works for every agent of this wavefront
    if (speculate) {
      back(t, par, cur, cur.Lcand, 0);
      if (t > 0) front(tile_next, par ^ 1, nxt);
    }
    const unsigned agreed = wait(t * LadderSync::kChunks);
    int win;
    if (agreed != 0u) win = __builtin_ctz(agreed); else settle_beyond(t, cur, win);
    if (!(speculate && win == 0)) {                             // not speculated, or speculated wrongly: the step with the agreed rung
      back(t, par, cur, cur.Lcand, win & (LL - 1));
      if (t > 0) front(tile_next, par ^ 1, nxt);
    }
    if (lane == 0 && a == 0) regs += win;                       // the stacked solve counts its rungs once (agent 0)
    if (t > 0) post((t - 1) * LadderSync::kChunks, nxt.field);
  };
  int t = d.T - 1;
  fetch(tileA, t);
  if (t > 0) fetch(tileB, t - 1);
  front(tileA, 0, frA);
  post(t * LadderSync::kChunks, frA.field);
  for (; t >= 1; t -= 2) { iterate(t, 0, frA, frB, tileB, tileA); iterate(t - 1, 1, frB, frA, tileA, tileB); }
  if (t == 0) iterate(0, 0, frA, frB, tileB, tileA);
  if (valid && lane == 0) reg_events[a] += regs;
}

}  // namespace mas
