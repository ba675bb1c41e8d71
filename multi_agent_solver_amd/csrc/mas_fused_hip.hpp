// mas_fused_hip.hpp -- one iLQR round in TWO kernels for the four-state models (HIP build only).
//
// The unfused round is five launches on every agent's critical path:
//     sweep (27 us) -> backward (86) -> line search (96) -> select (5) -> commit (9)            [C3, DESIGN.md section 4]
// although only the two chains in the middle are inherently serial.  Here a round is
//     mas_linearise_riccati_kernel  ->  line search
// and the first kernel does, per workgroup of 16 consecutive agents,
//   phase 0   select of the PREVIOUS round for its 16 agents (SelectBody: acceptance, termination, counters, next live list)
//   helpers   HELPERS wavefronts linearise the workgroup's (agent, t) items in REVERSE time, four time steps of the 16 agents per
//             wavefront pass (lane = (agent, dt); the first chunk shared out among three of them by blocks of the tile, so that the
//             recursion starts sooner), and hand the derivative tiles to the backward wavefronts through a ring of
//             RING chunk slots in LDS -- the tiles (46 KB per agent and iteration, 90 % of the unfused sweep's HBM traffic and all
//             of the backward pass's reads) never leave the CU.  The same lanes commit the trial the select accepted: they read
//             x_t, u_t from the trial plane and store them to X, U (ilqr.hpp:231-232) while they linearise there.
//   backward  four wavefronts (4 agents each, 16 lanes per agent = one DPP row = the 4 x 4 value Hessian) walk the Riccati
//             recursion of mas_backward_row_kernel, tile operands from the LDS ring instead of HBM.
// The linearisation of step t only needs the committed trajectory, the recursion consumes tiles last step first, and a lone
// backward wavefront leaves two thirds of its SIMD's issue slots empty (profiles/r2_sq_counters_c3.json): the helpers'
// ~1,500 FP64 instructions per item run in those slots (s_setprio keeps the chain ahead of them).
// Synchronisation is inside ONE workgroup: LDS words `ready[slot]` (helper -> backward) and `consumed[wave]` (backward ->
// helper), written behind s_waitcnt lgkmcnt(0); no global flags, no cross-CU visibility rules.  Every wait is bounded.
// Arithmetic: linearise<M>, SelectBody::decide and the recursion are the very functions / statements of the unfused kernels, so
// every result is bit-identical (tests/test_gpu_parity.py compares both forms with each other and with the oracle).
#pragma once

#include "mas_kernels_hip.hpp"

namespace mas {

struct FusedRound {
  SelectBody sel;                 // the previous round's select (do_select != 0)
  int do_select = 0;
  const double* Xs = nullptr;     // trial planes of the previous round's line search
  const double* Us = nullptr;
  int planes_by_slot = 0;         // those planes are indexed by the agent's slot in that round's live list (mas_forward_quad_kernel)
  const int* slot_prev = nullptr; // [Np] that slot
  double* D_dump = nullptr;       // staged access (parity tests): the tiles also go to the global tile array
  int* error = nullptr;           // raised when a bounded wait gives up
  // chained trust-region strategy (TRUST instantiation): the epilogue of every agent whose solve ended in the select above runs in this
  // kernel, before anything is linearised (fused_trust_phase)
  TrustState ts{};
  double* X_old = nullptr; double* U_old = nullptr; const double* x0 = nullptr; double dt = 0.0;
};

constexpr int kFusedAgents = 16;            // agents per workgroup: four backward wavefronts of four agents
constexpr int kFusedSpinLimit = 1 << 22;

// LDS control words of a workgroup
struct FusedCtl {
  int ready[4];                  // ready[s] = n + 1 once chunk n lies complete in ring slot s = n % RING
  int consumed[4];               // chunks backward wavefront w has finished reading (its data has arrived)
  int act[kFusedAgents], selj[kFusedAgents], col[kFusedAgents];
  int any_act, any_pend;
  int pad0;
  int parts_done;                // helpers that have written their blocks of chunk 0
  // trust-region epilogue (TRUST): the agent's solve ended in this round's select; its trajectory comes back from the OLD planes (revert);
  // what the helpers store to X, U also goes to the old planes (accept); its step leaves the radius (re-roll), scaled by scale[]
  int ended[kFusedAgents], from_old[kFusedAgents], dup[kFusedAgents], reroll[kFusedAgents];
  int any_ended, any_dup, rcount, rnext;   // rlist[0 .. rcount-1]: the group of re-rolling agents at work; rnext: where the next group starts
  int rlist[kFusedAgents];
  double scale[kFusedAgents];
};

template <class M, int HELPERS, int RING>
struct FusedLayout {
  using L = DerivLayout<M>;
  static constexpr int ND = L::ND;
  static constexpr int TS = ND | 1;                               // tile stride in the ring (odd: the 16 lanes of a ds_write_b64 group hit distinct banks)
  static constexpr int kBwdStride = 112;                          // per-agent scratch of the recursion (mas_backward_row_kernel)
  static constexpr int oBwd = 0, oRing = oBwd + kFusedAgents * kBwdStride, oCtl = oRing + RING * 64 * TS;
  static constexpr size_t kBytes = sizeof(double) * oCtl + sizeof(FusedCtl);
  static constexpr int kThreads = 64 * (4 + HELPERS);
  static_assert(RING > 0 ? HELPERS >= 1 : HELPERS <= 1, "RING == 0: no linearising helpers (at most the one wavefront that commits)");
};

// The hand-shake words between the wavefronts of a workgroup (ready[], consumed[], parts_done) are written with RELEASE and read with
// ACQUIRE semantics at workgroup scope, so that the order "tile stores -> flag" / "flag -> tile loads" is the memory model's and does not
// rest on the instruction order the scheduler happens to keep (on LDS both compile to the plain ds instruction plus the s_waitcnt
// that was already there).
__device__ __forceinline__ void lds_publish(int* word, int value) { __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// one LDS word, re-read until `done(value)`; uniform over the wavefront
template <class Pred>
__device__ __forceinline__ bool lds_wait(int* word, Pred done, int* error) {
  int spins = 0;
  for (;;) {
    const int v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (done(v)) return true;
    ++spins;
    if (spins > kFusedSpinLimit) { if ((threadIdx.x & 63) == 0 && error) atomicExch(error, 1); return false; }   // ~0.2 s
    if ((spins & 4095) == 0 && error && __hip_atomic_load(error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;   // somebody gave up already
    __builtin_amdgcn_s_sleep(2);
  }
}

// this lane's operands of one time step of the recursion (mas_backward_row_kernel): two columns of A and B and five scalars
struct RowTile { double Ai[4], Aj[4], Bi[4], Bj[4]; double lx, lu, lxx, lux, luu; };
__device__ __forceinline__ double pick4(int idx, double a0, double a1, double a2, double a3) {
  double r = a0; r = (idx == 1) ? a1 : r; r = (idx == 2) ? a2 : r; r = (idx == 3) ? a3 : r; return r;
}

// Where the recursion's tiles come from.  A source provides
//   Raw                      what travels one step ahead of its use
//   begin()                  before the first request
//   request(raw, t)          ask for the operands of step t (t >= 0); kRequestLate: issued behind the operand reads of pass 2
//   make(raw, t, tile)       at the start of step t: this lane's RowTile
//   arrived(t)               every LDS operation of the step that requested step t's operands has completed
// ---- tiles produced by the helper wavefronts, in the LDS ring
template <class M, int HELPERS, int RING>
struct RingSource {
  using FL = FusedLayout<M, HELPERS, RING>;
  using L = DerivLayout<M>;
  static constexpr int NX = 4, NU = M::NU, TS = FL::TS;
  static constexpr bool kMerged = NU <= 2, kRequestLate = true;
  using Raw = RowTile;
  const double* ring; FusedCtl* ctl; int* error; int wave, al, C, tid;
  int eAi, eAj, eBi, eBj, elx, elu, elxx, elux, eluu;
  __device__ RingSource(double* lds, FusedCtl* ctl_, int* error_, int wave_, int al_, int T, int lane, int i, int j, int iu, int ju)
      : ring(lds + FL::oRing), ctl(ctl_), error(error_), wave(wave_), al(al_), C((T + 3) >> 2), tid(static_cast<int>(threadIdx.x) & 63) {
    // element indices of this lane's operands (the ring holds plain tiles, TS doubles apart)
    eAi = L::oA + 4 * i; eAj = L::oA + 4 * j; eBi = L::oB + 4 * iu; eBj = L::oB + 4 * ju;
    elx = L::oLx + i; elu = L::oLu + iu; elxx = L::oLxx + lane; elux = L::oLux + iu + NU * j; eluu = L::oLuu + iu + NU * ju;
    if constexpr (kMerged) {                                           // merged second round of pass 2, see mas_backward_row_kernel
      const int item2 = (i < NU) ? 0 : ((i >= 2 && j < NU && i - 2 < NU) ? 1 : ((i >= 2 && j >= 2) ? 2 : 3));
      const int c2 = (i - 2) + 2 * (j - 2);
      if (item2 == 0) { eBj = eAj; }
      else if (item2 == 1) { eBj = L::oB + 4 * j; elux = L::oLuu + (i - 2) + NU * j; }
      else if (item2 == 2) { eBj = L::oA + 4 * c2; elux = L::oLx + c2; }
      else { eBj = eAj; }
    }
  }
  // tile of step t: chunk n = C-1 - t/4 (the order the helpers produce them in), ring slot n % RING, tile (t & 3) * 16 + agent
  __device__ __forceinline__ void acquire(int t) const {
    const int n = C - 1 - (t >> 2);
    lds_wait(&ctl->ready[n % RING], [n](int v) { return v >= n + 1; }, error);
  }
  __device__ __forceinline__ void begin() const {}
  __device__ __forceinline__ void request(Raw& o, int t, bool first) const {
    if (first || (t & 3) == 3) acquire(t);                             // uniform: this tile opens a new chunk
    const int n = C - 1 - (t >> 2);
    const double* tile = ring + (static_cast<size_t>(n % RING) * 64 + static_cast<size_t>(t & 3) * 16 + al) * TS;
    MAS_UNROLL for (int k = 0; k < NX; ++k) { o.Ai[k] = tile[eAi + k]; o.Aj[k] = tile[eAj + k]; o.Bi[k] = tile[eBi + k]; o.Bj[k] = tile[eBj + k]; }
    o.lu = tile[elu]; o.lxx = tile[elxx]; o.lux = tile[elux];
    if constexpr (!kMerged) { o.lx = tile[elx]; o.luu = tile[eluu]; }
  }
  __device__ __forceinline__ const RowTile& make(const Raw& r, int, RowTile&) const { return r; }
  // step t's tile has arrived: if it was the last of its chunk, the helpers may reuse the slot
  __device__ __forceinline__ void arrived(int t) const { if ((t & 3) == 0 && tid == 0) lds_publish(&ctl->consumed[wave], C - (t >> 2)); }
};

// ---- models whose derivatives are all analytic (the identity LQR): every lane evaluates them in registers from x_t, u_t.  No
// linearising helpers, no ring, no sweep: per step eight doubles are read instead of an 88-double tile.  The accepted trial is
// committed by a FIFTH wavefront of the workgroup (fused_commit_wave), not by these lanes: their stores would be four-lane
// partial-line writes, whose slow acknowledgements hold back the in-order counter the next step's operand loads retire
// through (measured: 73 -> 111 us per launch on the identity LQR).
template <class M>
struct AnalyticSource {
  using L = DerivLayout<M>;
  static constexpr int NX = 4, NU = M::NU;
  static constexpr bool kMerged = NU <= 2, kRequestLate = false;
  static_assert(M::kAnalyticA && M::kAnalyticB && M::kAnalyticLx && M::kAnalyticLu && M::kAnalyticLxx && M::kAnalyticLuu && M::kAnalyticLux && !kMerged,
                "every derivative analytic; the merged second round of NU <= 2 is not laid out here");
  struct Raw { double x[NX], u[NU]; };
  Dims d; const ModelParams* P; const double* Xsrc; const double* Usrc;
  unsigned col; int a, lane, i, j, iu, ju;
  __device__ __forceinline__ void begin() const {}
  __device__ __forceinline__ void request(Raw& o, int t, bool) const {
    MAS_UNROLL for (int q = 0; q < NX; ++q) o.x[q] = (Xsrc + (static_cast<size_t>(t) * NX + q) * d.Np)[col];
    MAS_UNROLL for (int q = 0; q < NU; ++q) o.u[q] = (Usrc + (static_cast<size_t>(t) * NU + q) * d.Np)[col];
  }
  __device__ __forceinline__ RowTile make(const Raw& r, int t, RowTile&) const {
    double Dl[L::ND];
    linearise<M>(r.x, r.u, t, Dl, *P);
    static_assert(NU == 4, "the picks below are written for 4 x 4 blocks");
    auto at = [&](int base, int r0, int c) {                           // entry (r0, c) of a column-major 4 x 4 block of Dl, r0 and c lane-dependent
      return pick4(c, pick4(r0, Dl[base], Dl[base + 1], Dl[base + 2], Dl[base + 3]), pick4(r0, Dl[base + 4], Dl[base + 5], Dl[base + 6], Dl[base + 7]),
                   pick4(r0, Dl[base + 8], Dl[base + 9], Dl[base + 10], Dl[base + 11]), pick4(r0, Dl[base + 12], Dl[base + 13], Dl[base + 14], Dl[base + 15]));
    };
    RowTile o;
    MAS_UNROLL for (int k = 0; k < NX; ++k) {                          // columns i, j of A (entries A(k, i)) and iu, ju of B
      o.Ai[k] = pick4(i, Dl[L::oA + k], Dl[L::oA + 4 + k], Dl[L::oA + 8 + k], Dl[L::oA + 12 + k]);
      o.Aj[k] = pick4(j, Dl[L::oA + k], Dl[L::oA + 4 + k], Dl[L::oA + 8 + k], Dl[L::oA + 12 + k]);
      o.Bi[k] = pick4(iu, Dl[L::oB + k], Dl[L::oB + 4 + k], Dl[L::oB + 8 + k], Dl[L::oB + 12 + k]);
      o.Bj[k] = pick4(ju, Dl[L::oB + k], Dl[L::oB + 4 + k], Dl[L::oB + 8 + k], Dl[L::oB + 12 + k]);
    }
    o.lx = pick4(i, Dl[L::oLx], Dl[L::oLx + 1], Dl[L::oLx + 2], Dl[L::oLx + 3]);
    o.lu = pick4(iu, Dl[L::oLu], Dl[L::oLu + 1], Dl[L::oLu + 2], Dl[L::oLu + 3]);
    o.lxx = at(L::oLxx, i, j);                                         // l_xx(i, j): element i + 4 j = this lane's
    o.lux = at(L::oLux, iu, j);                                        // l_ux(iu, j)
    o.luu = at(L::oLuu, iu, ju);                                       // l_uu(iu, ju)
    return o;
  }
  __device__ __forceinline__ void arrived(int) const {}
};

// ---- the recursion of one backward wavefront (4 agents, 16 lanes each): mas_backward_row_kernel's step, operands from the LDS ring --
template <class M, int BWD_BASE, int BWD_STRIDE, class Src>
__device__ __forceinline__ void fused_backward_wave(const Dims& d, const ModelParams& P, double* __restrict__ lds, bool valid, Src& src,
                                                    int wave, int a_base, const double* __restrict__ xT_src, unsigned xT_col, double* __restrict__ X_commit,
                                                    double* __restrict__ kff, double* __restrict__ Kfb, long long* __restrict__ reg_events) {
  constexpr int NX = 4, NU = M::NU;
  static_assert(M::NX == 4 && NU >= 1 && NU <= 4, "one DPP row = the 4 x 4 value Hessian");
  constexpr int LPA = 16, LL = 16;
  constexpr int oAtV = 0, oBtV = 16, oQxx = 32, oQux = 48, oL = 64, oVx = 80, oQx = 84, oPad = 96;
  constexpr bool kMerged = NU <= 2;
  constexpr bool kRung0First = M::kAnalyticLuu;
  constexpr bool kFactorByPermute = NU <= 2;

  const int tid = static_cast<int>(threadIdx.x) & 63, lane = tid & (LPA - 1), slot = tid / LPA, rung_lane = lane;
  const int i = lane & 3, j = lane >> 2, iu = (i < NU) ? i : NU - 1, ju = (j < NU) ? j : NU - 1;
  const int al = wave * 4 + slot, a = a_base + al;
  double* T = lds + BWD_BASE + al * BWD_STRIDE;
  auto wave_sync = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };

  // terminal boundary condition, ilqr.hpp:92-102: lane 0 evaluates it, every lane picks up its own entries
  if (lane == 0) {
    double xT[NX], vx[NX], vxx[NX * NX];
    MAS_UNROLL for (int q = 0; q < NX; ++q) xT[q] = (valid || X_commit) ? (xT_src + (static_cast<size_t>(d.T) * NX + q) * d.Np)[xT_col] : 0.0;
    if (X_commit) { MAS_UNROLL for (int q = 0; q < NX; ++q) MAS_AT(X_commit, d.T * NX + q, a) = xT[q]; }      // (sources that commit while they read: x_T with the boundary condition)
    terminal_boundary<M>(xT, vx, vxx, P);
    MAS_UNROLL for (int q = 0; q < NX; ++q) T[oVx + q] = vx[q];
    MAS_UNROLL for (int q = 0; q < NX * NX; ++q) T[oQxx + q] = vxx[q];
  }
  wave_sync();
  double Vown = T[oQxx + lane], vxi = T[oVx + i];
  wave_sync();

  // LDS row the second-round item of pass 2 reads (merged second round, NU <= 2: see mas_backward_row_kernel)
  const int item2 = (i < NU) ? 0 : ((i >= 2 && j < NU && i - 2 < NU) ? 1 : ((i >= 2 && j >= 2) ? 2 : 3));
  const int c2 = (i - 2) + 2 * (j - 2);
  int r_row2 = oBtV + iu * 4;
  if constexpr (kMerged) {
    if (item2 == 1) r_row2 = oBtV + (i - 2) * 4;
    else if (item2 == 2) r_row2 = oVx;
    else if (item2 == 3) r_row2 = oBtV;
  }
  using Raw = typename Src::Raw;
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
  const int w_atv = oAtV + i * 4 + j, w_btv = (i < NU) ? (oBtV + i * 4 + j) : (oPad + lane);
  const int w_qux = (i < NU) ? (oQux + j * NU + i) : ((kMerged && item2 == 2) ? (oQx + c2) : (oPad + lane));
  const int w_vx = (j == 0) ? (oVx + i) : (oPad + lane);

  Raw tA, tB;
  src.begin();
  src.request(tA, d.T - 1, true);
  auto step = [&](int t, const Raw& raw, Raw& nx) {
    MAS_UNROLL for (int m = 0; m < NU; ++m) { MAS_OPAQUE32(goff[m]); MAS_OPAQUE32(koff[m]); }
    // UNCONDITIONAL (at t == 0 step 0 is requested again, unused): behind a branch the compiler must size the wait for these
    // operands for the path WITHOUT the request -- eight operations fewer in flight -- and on the taken path that wait then
    // includes the acknowledgement of the previous step's gain stores: measured +0.6 us per step on the identity LQR
    if constexpr (!Src::kRequestLate) src.request(nx, t > 0 ? t - 1 : 0, false);
    RowTile made;
    const RowTile& c = src.make(raw, t, made);
    const double* Ai = c.Ai; const double* Aj = c.Aj; const double* Bi = c.Bi; const double* Bj = c.Bj;

    // ---- pass 1 (ilqr.hpp:115-116 and the first factors of :117-119)
    double qx_i, qu_own;
    constexpr bool kIdentity = M::kIdentityJacobians && !kMerged && NU == NX;      // A = B = I: every product of passes 1 and 2 is V (or v_x) itself
    static_assert(!kIdentity || !Src::kRequestLate, "identity Jacobians: tiles come from the analytic source");
    if constexpr (kIdentity) {
      qx_i = c.lx + vxi;                                                // q_x = l_x + A^T v_x,  q_u = l_u + B^T v_x: lane (i, j) holds v_x(i), and iu == i
      qu_own = c.lu + vxi;
    } else {
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
    if constexpr (!kIdentity) wave_sync();
    // ---- pass 2 (second factors of ilqr.hpp:117-119)
    double quu[NU * NU], qu[NU], quxj[NU], Lfac[NU * NU], ninv[NU * NU], kf[NU], qxx_own;
    if constexpr (kIdentity) {
      // q_xx = l_xx + (A^T V) A,  q_ux = l_ux + (B^T V) A,  q_uu = l_uu + (B^T V) B  with A = B = I: entry (i, j) of each is l + V(i, j), this lane's own
      qxx_own = c.lxx + Vown;
      T[oQxx + lane] = qxx_own;
      const double qux_own = c.lux + Vown;
      const double quu_own = c.luu + Vown;
      T[w_qux] = qux_own;
      row_gather_block_at<NU, 0>(quu_own, quu, std::make_integer_sequence<int, NU * NU>{});
      quad_gather(qux_own, quxj, std::make_integer_sequence<int, NU>{});
      row_gather_first(qu_own, qu, std::make_integer_sequence<int, NU>{});
    } else {
      double ra[NX], rb[NX];
      MAS_UNROLL for (int k = 0; k < NX; ++k) { ra[k] = T[oAtV + i * 4 + k]; rb[k] = T[r_row2 + k]; }
      // the next step's tile is requested HERE, behind the operand reads of pass 2 (an LDS queue serves one wavefront's
      // instructions in order): its ~20 reads travel in the shadow of passes 2 and 3 instead of in front of the wait above
      if constexpr (Src::kRequestLate) src.request(nx, t > 0 ? t - 1 : 0, false);
      double s;
      s = sum_terms<NX>([&](int k) { return ra[k] * Aj[k]; });
      qxx_own = c.lxx + s;
      T[oQxx + lane] = qxx_own;
      if constexpr (kMerged) {
        s = sum_terms<NX>([&](int k) { return rb[k] * Bj[k]; });          // this lane's second-round item
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
          MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = cur[e];
          const int ok0 = llt_compute_straight<NU>(Lfac) == 0;
          rung0_everywhere = __ballot(ok0 || !valid) == ~0ull;
          if (rung0_everywhere) win = 0;
        }
        if (!rung0_everywhere) {
          MAS_UNROLL for (int q = 0; q < LL - 1; ++q) {
            MAS_UNROLL for (int e = 0; e < NU; ++e) cur[e * NU + e] += rung_add[q];
          }
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
          if (lane == 0) regs += win;
          if constexpr (kFactorByPermute) {
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
            wave_sync();                                               // also orders the pass-2 outputs before the reads below
            MAS_UNROLL for (int e = 0; e < NU * NU; ++e) Lfac[e] = T[oL + e];
          }
        } else {
          wave_sync();                                                 // the pass-2 outputs before the reads below
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
    // the tile requested in pass 2 has arrived (pass 3 waited for every LDS operation): if it was the last of its chunk, the
    // helpers may reuse the slot
    if (t > 0) src.arrived(t - 1);
    kff_t -= static_cast<size_t>(NU) * d.Np; Kfb_t -= static_cast<size_t>(NU) * NX * d.Np;
  };
  int t = d.T - 1;
  for (; t >= 1; t -= 2) { step(t, tA, tB); step(t - 1, tB, tA); }
  if (t == 0) step(0, tA, tB);
  if (valid && lane == 0) reg_events[a] += regs;
}

// ---- trust-region epilogue of the agents whose solve ended in this round's select (detail::run_trust_region, nash.hpp:214-243), all
// threads of the workgroup, before the helpers and the recursion start.  The same statements as mas_trust_epilogue_kernel /
// TrustNormBody / TrustRerollBody / TrustCopyBody of the five-kernel round:
//   * step norm: the squared differences of all 16 agents side by side, their sums serially in the reference's column-major order
//     (one lane per agent);
//   * a step inside the radius is accepted (cost < old cost: radius x1.5, the new trajectory becomes the old one) or reverted
//     (radius x0.5) at once -- and costs NO copy pass: the helpers, which store the accepted trial to X, U while they linearise, store
//     it to the old planes as well (dup), or take the agent's trajectory from the old planes instead of the trial (from_old);
//   * a step that leaves the radius is scaled back and rolled out by dependency levels (mas_kernel_levels.hpp, all threads of the
//     workgroup, ~10 us per agent), re-costed, accepted or reverted.
// LDS: the ring, idle until the helpers start.
template <class M, class FL>
__device__ __forceinline__ void fused_trust_phase(const Dims& d, const ModelParams& P, double* lds, FusedCtl* ctl, int a_base,
                                                  double* __restrict__ X, double* __restrict__ U, const FusedRound& fr) {
  constexpr int NX = M::NX, NU = M::NU;
  const int tid = static_cast<int>(threadIdx.x), nt = static_cast<int>(blockDim.x), T = d.T, R = T * NU, RS = R | 1;
  double* scratch = lds + FL::oRing;
  auto u_new = [&](int al, int r) {                                     // the controls the solve ended with: its last accepted trial, else U
    const int jsel = ctl->selj[al];
    return jsel >= 0 ? (fr.Us + (static_cast<size_t>(jsel) * T * NU + r) * d.Np)[static_cast<unsigned>(ctl->col[al])] : MAS_AT(U, r, a_base + al);
  };
  for (int e = tid; e < kFusedAgents * R; e += nt) {
    const int al = e & (kFusedAgents - 1), r = e >> 4;
    if (ctl->ended[al]) { const double dl = u_new(al, r) - MAS_AT(fr.U_old, r, a_base + al); scratch[al * RS + r] = dl * dl; }
  }
  __syncthreads();
  if (tid < kFusedAgents && ctl->ended[tid]) {
    const int a = a_base + tid;
    const double norm = sqrt(lds_serial_sum(scratch + tid * RS, R));    // delta.norm(): Frobenius, column-major order (nash.hpp:216-217)
    const double radius = fr.ts.radii[a];
    if (norm > radius) { ctl->reroll[tid] = 1; ctl->scale[tid] = radius / norm; }
    else {
      const bool accept = fr.ts.cost[a] < fr.ts.cost_old[a];
      fr.ts.finish(a, fr.ts.cost[a], radius);                           // accept / revert, nash.hpp:228-243
      if (accept) ctl->dup[tid] = 1; else ctl->from_old[tid] = 1;
    }
  }
  __syncthreads();
  // the agents whose step leaves the radius, in groups of as many as the ring holds (xs | inc | uc | misc per agent)
  const size_t per = static_cast<size_t>(NX) * (T + 1) + static_cast<size_t>(NX) * T + R + 8;
  const int gmax = static_cast<int>((static_cast<size_t>(FL::oCtl - FL::oRing)) / per);
  if (tid == 0) ctl->rnext = 0;
  for (;;) {
    __syncthreads();
    if (tid == 0) {
      int G0 = 0, al = ctl->rnext;
      for (; al < kFusedAgents && G0 < gmax; ++al) { if (ctl->reroll[al]) ctl->rlist[G0++] = al; }
      ctl->rcount = G0; ctl->rnext = al;
    }
    __syncthreads();
    const int G = ctl->rcount;                                          // (uniform)
    if (G == 0) break;
    const int* list = ctl->rlist;
    for (int e = tid; e < G * R; e += nt) {
      const int g = e / R, r = e - g * R, al = list[g], a = a_base + al;
      const double o = MAS_AT(fr.U_old, r, a);
      (scratch + g * per + NX * (T + 1) + NX * T)[r] = o + ctl->scale[al] * (u_new(al, r) - o);          // cand_u = old_u + scale * delta
    }
    if (tid < G * NX) { const int g = tid / NX, i = tid - g * NX; (scratch + g * per)[i * (T + 1)] = MAS_AT(fr.x0, i, a_base + list[g]); }
    __syncthreads();
    levels_rollout_cost_group<M>(scratch, per, G, T, fr.dt, P, tid, nt);                                   // cand_x = rollout; cand_cost = objective (nash.hpp:219-226)
    if (tid < G) {
      const int a = a_base + list[tid];
      double* misc = scratch + tid * per + NX * (T + 1) + NX * T + R;
      const double c = misc[3];
      misc[4] = (c < fr.ts.cost_old[a]) ? 1.0 : 0.0;
      fr.ts.finish(a, c, fr.ts.radii[a]);
      if (misc[4] != 0.0) { ctl->selj[list[tid]] = -1; ctl->col[list[tid]] = a; }      // accepted: nothing pending any more, the helpers read X, U
      else ctl->from_old[list[tid]] = 1;
    }
    __syncthreads();
    for (int e = tid; e < G * (T + 1) * NX; e += nt) {                  // accepted: the re-rolled trajectory is the current AND the old one
      const int g = e / ((T + 1) * NX), q = e - g * (T + 1) * NX, a = a_base + list[g];
      const double* xs = scratch + g * per;
      if (xs[NX * (T + 1) + NX * T + R + 4] != 0.0) { const double v = xs[(q % NX) * (T + 1) + q / NX]; MAS_AT(X, q, a) = v; MAS_AT(fr.X_old, q, a) = v; }
    }
    for (int e = tid; e < G * R; e += nt) {
      const int g = e / R, r = e - g * R, a = a_base + list[g];
      const double* xs = scratch + g * per;
      if (xs[NX * (T + 1) + NX * T + R + 4] != 0.0) { const double v = xs[NX * (T + 1) + NX * T + r]; MAS_AT(U, r, a) = v; MAS_AT(fr.U_old, r, a) = v; }
    }
    __syncthreads();
  }
  if (tid < 64) {
    const bool mine = tid < kFusedAgents;
    const unsigned long long pm = __ballot(mine && (ctl->selj[mine ? tid : 0] >= 0 || ctl->from_old[mine ? tid : 0] != 0)), dm = __ballot(mine && ctl->dup[mine ? tid : 0] != 0);
    if (tid == 0) { ctl->any_pend = pm != 0ull; ctl->any_dup = dm != 0ull; }
  }
  __syncthreads();
}

// ---- one helper wavefront: commit + linearisation of the workgroup's items, chunk by chunk in reverse time ----
// The recursion cannot start before chunk 0 (the last four time steps) lies in the ring, and one item is ~1,400 dependent FP64
// instructions (~6 us for a lone wavefront).  So chunk 0 is SHARED OUT BY BLOCKS of the tile: helper 0 evaluates A and B, helper 1
// l_xx, helper 2 l_x, l_u, l_uu, l_ux (552 / 457 / 465 instructions: ~2.4 us), each writes its elements of the 64 tiles, and the
// last of the three to finish publishes the chunk; the remaining helper starts on chunk 1 at once, the others follow with chunks 2, 3,
// 4 and from there every helper takes every fourth chunk.  Same functions, same operands: identical tiles.
template <int PARTS, class L>
struct PartRange {                 // the elements of a tile the blocks in PARTS cover: up to two runs [lo1, hi1) and [lo2, hi2) (tile order: A B l_x l_u l_xx l_uu l_ux)
  static constexpr bool kAB = (PARTS & kLinA) != 0, kXX = (PARTS & kLinLxx) != 0;
  static constexpr int lo1 = kAB ? L::oA : kXX ? L::oLxx : L::oLx, hi1 = kAB ? L::oLx : kXX ? L::oLuu : L::oLxx;
  static constexpr int lo2 = (kAB || kXX) ? 0 : L::oLuu, hi2 = (kAB || kXX) ? 0 : L::oCore;
  template <class F> __device__ static void for_each(F f) {
    MAS_UNROLL for (int e = lo1; e < hi1; ++e) f(e);
    MAS_UNROLL for (int e = lo2; e < hi2; ++e) f(e);
  }
};
template <class M, int HELPERS, int RING, bool TRUST>
__device__ __forceinline__ void fused_helper_wave(const Dims& d, const ModelParams& P, double* __restrict__ lds, FusedCtl* ctl, int h, int a_base,
                                                  double* __restrict__ X, double* __restrict__ U, const FusedRound& fr) {
  using FL = FusedLayout<M, HELPERS, RING>;
  using L = DerivLayout<M>;
  constexpr int NX = M::NX, NU = M::NU, ND = L::ND, TS = FL::TS;
  // the three groups of chunk 0 (helpers beyond the third take whole chunks from the start)
  constexpr int kGroup0 = kLinA | kLinB, kGroup1 = kLinLxx, kGroup2 = kLinLx | kLinLu | kLinLuu | kLinLux;
#if defined(MAS_FUSED_NO_SHARED_FIRST_CHUNK)
  constexpr bool kShareFirst = false;                                  // (A/B builds)
#else
  constexpr bool kShareFirst = HELPERS >= 4 && ND == L::oCore;         // (unconstrained tiles; with fewer helpers nobody would be left for chunk 1)
#endif
  static_assert((PartRange<kGroup0, L>::hi1 - PartRange<kGroup0, L>::lo1) + (PartRange<kGroup1, L>::hi1 - PartRange<kGroup1, L>::lo1) +
                (PartRange<kGroup2, L>::hi1 - PartRange<kGroup2, L>::lo1) + (PartRange<kGroup2, L>::hi2 - PartRange<kGroup2, L>::lo2) == L::oCore,
                "the three groups cover a tile exactly once");
  const int l = static_cast<int>(threadIdx.x) & 63, al = l & 15, dt = l >> 4, a = a_base + al;
  const bool act = ctl->act[al] != 0;
  const int jsel = ctl->selj[al];
  bool from_old = false, dup = false;                                  // trust-region epilogue: reverted to / accepted as the old trajectory
  if constexpr (TRUST) { from_old = ctl->from_old[al] != 0; dup = ctl->dup[al] != 0; }
  const bool pend = jsel >= 0 || from_old;
  const bool any_act = ctl->any_act != 0;
  // where this agent's trajectory lies: the old planes (reverted), the trial plane its select accepted (column = agent or its slot in
  // that round's list), else X, U
  const double* Xsrc = from_old ? fr.X_old : (jsel >= 0 ? fr.Xs + static_cast<size_t>(jsel) * (d.T + 1) * NX * d.Np : X);
  const double* Usrc = from_old ? fr.U_old : (jsel >= 0 ? fr.Us + static_cast<size_t>(jsel) * d.T * NU * d.Np : U);
  const unsigned col = from_old ? static_cast<unsigned>(a) : static_cast<unsigned>(ctl->col[al]);
  double* ring = lds + FL::oRing;
  const int C = (d.T + 3) >> 2;
  auto load_xu = [&](int t, bool wanted, double* x, double* u) {
    MAS_UNROLL for (int q = 0; q < NX; ++q) x[q] = 0.0;
    MAS_UNROLL for (int q = 0; q < NU; ++q) u[q] = 0.0;
    if (wanted) {
      MAS_UNROLL for (int q = 0; q < NX; ++q) x[q] = (Xsrc + (static_cast<size_t>(t) * NX + q) * d.Np)[col];
      MAS_UNROLL for (int q = 0; q < NU; ++q) u[q] = (Usrc + (static_cast<size_t>(t) * NU + q) * d.Np)[col];
    }
  };
  auto commit = [&](int t, const double* x, const double* u) {         // x = best_x; u = best_u, ilqr.hpp:231-232
    if (pend) {
      MAS_UNROLL for (int q = 0; q < NX; ++q) MAS_AT(X, t * NX + q, a) = x[q];
      MAS_UNROLL for (int q = 0; q < NU; ++q) MAS_AT(U, t * NU + q, a) = u[q];
    }
    if constexpr (TRUST) {                                              // old_x = best_x; old_u = best_u of an accepted outer iteration, nash.hpp:232-236
      if (dup) {
        MAS_UNROLL for (int q = 0; q < NX; ++q) MAS_AT(fr.X_old, t * NX + q, a) = x[q];
        MAS_UNROLL for (int q = 0; q < NU; ++q) MAS_AT(fr.U_old, t * NU + q, a) = u[q];
      }
    }
    if (t == d.T - 1) {
      MAS_UNROLL for (int q = 0; q < NX; ++q) {
        const double xT = (Xsrc + (static_cast<size_t>(d.T) * NX + q) * d.Np)[col];
        if (pend) MAS_AT(X, d.T * NX + q, a) = xT;
        if constexpr (TRUST) { if (dup) MAS_AT(fr.X_old, d.T * NX + q, a) = xT; }
      }
    }
  };
  // the blocks PARTS of chunk 0: elements [lo, hi) of this lane's tile
  auto share_of_first_chunk = [&](auto parts_tag, bool commits) {
    constexpr int PARTS = decltype(parts_tag)::value;
    using R = PartRange<PARTS, L>;
    const int t = 4 * (C - 1) + dt;
    const bool in = t < d.T;
    double x[NX], u[NU];
    load_xu(t, in && (act || pend || dup), x, u);
    if (commits && in && (pend || dup)) commit(t, x, u);
    if (!any_act) return;
    double Dl[ND];
    R::for_each([&](int e) { Dl[e] = 0.0; });
    if (in && act) linearise<M, PARTS>(x, u, t, Dl, P);
    double* tile = ring + static_cast<size_t>(l) * TS;                 // chunk 0 -> slot 0
    R::for_each([&](int e) { tile[e] = Dl[e]; });
    if (fr.D_dump && in && act) R::for_each([&](int e) { MAS_TILE_AT(fr.D_dump, ND, t, a, e) = Dl[e]; });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int before = 0;
    if (l == 0) before = __hip_atomic_fetch_add(&ctl->parts_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);   // LDS atomic: the last of the three publishes the chunk
    if (__builtin_amdgcn_readfirstlane(before) == 2 && l == 0) lds_publish(&ctl->ready[0], 1);
  };
  auto whole_chunk = [&](int n) {
    const int t = 4 * (C - 1 - n) + dt;
    const bool in = t < d.T;
    double x[NX], u[NU];
    load_xu(t, in && (act || pend || dup), x, u);
    if (in && (pend || dup)) commit(t, x, u);
    if (!any_act) return;                                              // uniform: a workgroup of retired agents only commits
    double Dl[ND];
    if (in && act) linearise<M>(x, u, t, Dl, P);
    else { MAS_UNROLL for (int e = 0; e < ND; ++e) Dl[e] = 0.0; }
    if (n >= RING) {                                                   // the slot's previous chunk has been read by all four backward wavefronts
      MAS_UNROLL for (int w = 0; w < 4; ++w) lds_wait(&ctl->consumed[w], [n](int v) { return v >= n - RING + 1; }, fr.error);
    }
    double* tile = ring + (static_cast<size_t>(n % RING) * 64 + l) * TS;
    MAS_UNROLL for (int e = 0; e < ND; ++e) tile[e] = Dl[e];
    if (fr.D_dump && in && act) { MAS_UNROLL for (int e = 0; e < ND; ++e) MAS_TILE_AT(fr.D_dump, ND, t, a, e) = Dl[e]; }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (l == 0) lds_publish(&ctl->ready[n % RING], n + 1);
  };
  if constexpr (kShareFirst) {
    if (h == 0) share_of_first_chunk(std::integral_constant<int, kGroup0>{}, true);
    else if (h == 1) share_of_first_chunk(std::integral_constant<int, kGroup1>{}, false);
    else if (h == 2) share_of_first_chunk(std::integral_constant<int, kGroup2>{}, false);
  }
  // whole chunks, statically dealt: with chunk 0 shared out, the helper that has no share of it starts on chunk 1 at once and the
  // others follow with chunks 2, 3, 4 (a queue in LDS -- atomicAdd per grab -- was tried first: the kernel never finished, cause
  // not found)
  const int first = kShareFirst ? ((h + 1) % HELPERS) + 1 : h;
  for (int n = first; n < C; n += HELPERS) whole_chunk(n);
}

// ---- commit of the accepted trials of a workgroup's 16 agents by one wavefront (models without linearising helpers) ----
// lane = (agent, row mod 4): a store instruction covers four rows of the 16 agents -- whole 128-byte lines when all of them commit
template <class M>
__device__ __forceinline__ void fused_commit_wave(const Dims& d, const FusedCtl* ctl, int a_base, double* __restrict__ X, double* __restrict__ U, const FusedRound& fr) {
  constexpr int NX = M::NX, NU = M::NU;
  const int l = static_cast<int>(threadIdx.x) & 63, al = l & 15, sub = l >> 4, a = a_base + al;
  const int jsel = ctl->selj[al];
  if (jsel < 0) return;                                                // (divergent: the other lanes go on)
  const unsigned col = static_cast<unsigned>(ctl->col[al]);
  auto copy_rows = [&](const double* __restrict__ src, double* __restrict__ dst, int rows) {
    for (int r0 = 0; r0 < rows; r0 += 32) {                            // eight row groups in flight
      double v[8];
      MAS_UNROLL for (int q = 0; q < 8; ++q) { const int r = r0 + 4 * q + sub; if (r < rows) v[q] = (src + static_cast<size_t>(r) * d.Np)[col]; }
      MAS_UNROLL for (int q = 0; q < 8; ++q) { const int r = r0 + 4 * q + sub; if (r < rows) MAS_AT(dst, r, a) = v[q]; }
    }
  };
  copy_rows(fr.Xs + static_cast<size_t>(jsel) * (d.T + 1) * NX * d.Np, X, (d.T + 1) * NX);       // x = best_x; u = best_u, ilqr.hpp:231-232
  copy_rows(fr.Us + static_cast<size_t>(jsel) * d.T * NU * d.Np, U, d.T * NU);
}

// grid = ceil(N / 16) workgroups of 64 * (4 + HELPERS) threads, FusedLayout::kBytes of dynamic LDS
template <class M, int HELPERS, int RING, bool TRUST = false>
__global__ void __launch_bounds__(64 * (4 + HELPERS), HELPERS == 0 ? 1 : 2)   // (with helpers two wavefronts share a SIMD)
mas_linearise_riccati_kernel(Dims d, ModelParams P, double* __restrict__ X, double* __restrict__ U, double* __restrict__ kff, double* __restrict__ Kfb,
                             long long* __restrict__ reg_events, FusedRound fr) {
  using FL = FusedLayout<M, HELPERS, RING>;
  static_assert((RING >= 2 && RING <= 4 && HELPERS >= 1) || (RING == 0 && HELPERS <= 1), "ring of chunk slots");
  static_assert(!TRUST || (RING > 0 && M::kRolloutLevels > 0), "the trust-region epilogue runs in the helpers' workgroups and re-rolls by dependency levels");
  extern __shared__ double lds[];
  FusedCtl* ctl = reinterpret_cast<FusedCtl*>(lds + FL::oCtl);
  const int tid = static_cast<int>(threadIdx.x), wave = tid >> 6, lane = tid & 63;
  const int a_base = static_cast<int>(blockIdx.x) * kFusedAgents;

  // ---- phase 0: the previous round's select for this workgroup's agents (wavefront 0, lanes 0..15)
  int sel_outcome = SelectBody::kIdle;
  if (wave == 0) {
    const int a = a_base + lane;
    const bool mine = lane < kFusedAgents && a < d.N;
    if (fr.do_select && mine) sel_outcome = fr.sel.decide(a);
    int act = 0, jsel = -1;
    if (mine) { act = fr.sel.active[a]; jsel = fr.sel.alpha_sel[a]; }
    if (lane < kFusedAgents) {
      ctl->act[lane] = act; ctl->selj[lane] = jsel;
      ctl->col[lane] = (jsel >= 0 && fr.planes_by_slot) ? fr.slot_prev[a] : a;
    }
    const unsigned long long am = __ballot(act != 0), pm = __ballot(jsel >= 0);
    if (lane == 0) { ctl->any_act = am != 0ull; ctl->any_pend = pm != 0ull; }
    if (lane < 4) { ctl->ready[lane] = 0; ctl->consumed[lane] = 0; }
    if (lane == 0) ctl->parts_done = 0;
    if constexpr (TRUST) {
      const int ended = (mine && (sel_outcome == SelectBody::kConverged || sel_outcome == SelectBody::kRetired || (sel_outcome & SelectBody::kChained) != 0)) ? 1 : 0;
      if (lane < kFusedAgents) { ctl->ended[lane] = ended; ctl->from_old[lane] = 0; ctl->dup[lane] = 0; ctl->reroll[lane] = 0; }
      const unsigned long long em = __ballot(ended != 0);
      if (lane == 0) { ctl->any_ended = em != 0ull; ctl->any_dup = 0; }
    }
  }
  __syncthreads();
  // counters and the next round's live list (two atomics, one of them returning) AFTER the barrier: the helpers are on their
  // first chunk by now, and this wavefront's recursion cannot start before that chunk is there
  if (wave == 0 && fr.do_select) fr.sel.count(sel_outcome, a_base + lane);
  if constexpr (TRUST) {
    if (ctl->any_ended) fused_trust_phase<M, FL>(d, P, lds, ctl, a_base, X, U, fr);       // (uniform)
  }
  const bool any_act = ctl->any_act != 0, any_pend = ctl->any_pend != 0;
  bool any_dup = false;
  if constexpr (TRUST) any_dup = ctl->any_dup != 0;
  if (!any_act && !any_pend && !any_dup) return;

  if constexpr (RING == 0) {
    // every derivative analytic: the backward wavefronts evaluate them (AnalyticSource); a fifth wavefront commits
    if (wave >= 4) {
      if (any_pend) fused_commit_wave<M>(d, ctl, a_base, X, U, fr);
      return;
    }
    if (!any_act) return;
    const int slot = lane >> 4, al0 = wave * 4 + slot, a = a_base + al0;
    const bool valid = ctl->act[al0] != 0;
    if (__ballot(valid) == 0ull) return;
    const int jsel = ctl->selj[al0];
    const bool pend = jsel >= 0;
    const double* Xsrc = pend ? fr.Xs + static_cast<size_t>(jsel) * (d.T + 1) * M::NX * d.Np : X;
    const double* Usrc = pend ? fr.Us + static_cast<size_t>(jsel) * d.T * M::NU * d.Np : U;
    const int l16 = lane & 15, i = l16 & 3, j = l16 >> 2;
    AnalyticSource<M> src{d, &P, Xsrc, Usrc, static_cast<unsigned>(ctl->col[al0]), a, l16, i, j, (i < M::NU) ? i : M::NU - 1, (j < M::NU) ? j : M::NU - 1};
    fused_backward_wave<M, FL::oBwd, FL::kBwdStride>(d, P, lds, valid, src, wave, a_base, Xsrc, src.col, nullptr, kff, Kfb, reg_events);
  } else {
  if (wave < 4) {
    if (!any_act) return;
    const int al0 = wave * 4 + (lane >> 4);
    if (__ballot(ctl->act[al0] != 0) == 0ull) {                       // none of this wavefront's four agents is active: the helpers never wait for it
      if (lane == 0) lds_publish(&ctl->consumed[wave], 0x7fffffff);
      return;
    }
#if !defined(MAS_FUSED_NO_PRIO)
    __builtin_amdgcn_s_setprio(3);                                     // the chain goes first; the helpers fill the issue slots it leaves
#endif
    const int jsel = ctl->selj[al0];
    const double* xT_src = (jsel >= 0) ? fr.Xs + static_cast<size_t>(jsel) * (d.T + 1) * M::NX * d.Np : X;
    unsigned xT_col = static_cast<unsigned>(ctl->col[al0]);
    if constexpr (TRUST) { if (ctl->from_old[al0]) { xT_src = fr.X_old; xT_col = static_cast<unsigned>(a_base + al0); } }   // (reverted: x_T of the old trajectory)
    const int l16 = lane & 15, i = l16 & 3, j = l16 >> 2;
    RingSource<M, HELPERS, RING> src(lds, ctl, fr.error, wave, al0, d.T, l16, i, j, (i < M::NU) ? i : M::NU - 1, (j < M::NU) ? j : M::NU - 1);
    fused_backward_wave<M, FL::oBwd, FL::kBwdStride>(d, P, lds, ctl->act[al0] != 0, src, wave, a_base, xT_src, xT_col, nullptr, kff, Kfb, reg_events);
  } else {
    __builtin_amdgcn_s_setprio(0);
    fused_helper_wave<M, HELPERS, RING, TRUST>(d, P, lds, ctl, wave - 4, a_base, X, U, fr);
  }
  }
}

}  // namespace mas
