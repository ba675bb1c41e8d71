// mas_kernel_backward_tile.hpp -- backward Riccati recursion for any NX over an LDS tile per agent (HIP build only; see mas_kernels_hip.hpp)
#pragma once

#include <type_traits>
#include <utility>

#include "mas_kernel_backward_common.hpp"

namespace mas {

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
                       kOwn = oNegZero + 1;
  // round 3: [K | k] (NU x (NX+1), column-major) and K^T q_uu (NU x NX), formed once per step by NX+1 lanes (stage A of pass 3) instead
  // of by every lane of every round.  They live where A^T V and B^T V were: those are dead once pass 2 has read them, and one wavefront's
  // LDS instructions execute in order (a tile is private to its 16 lanes).  Models whose pass-1 outputs are too small get a region of their own.
  static constexpr int kGainCols = NU * (NX + 1) + NU * NX;
  static constexpr bool kGainsAliased = kGainCols + (oAtV & 1) <= NX * NX + NU * NX;
  static constexpr int oK = kGainsAliased ? oAtV + (oAtV & 1) : kOwn + (kOwn & 1),     // even: a column is one ds_read_b128 for NU = 2
                       oKtQ = oK + NU * (NX + 1),
                       kUsed = kGainsAliased ? kOwn : oKtQ + NU * NX, oPad = (kUsed + 15) / 16 * 16, kEnd = oPad + 16;
  // agents of one 32-lane read group (two tiles) kStrideMod doubles apart modulo 32
  static constexpr int kStrideMod = 16;       // measured in round 2 (conflict cycles per wavefront for strides = 16 / 10 / 2 / 0 mod 32: 3,843 / 4,117 / 5,215 / 7,411; launch time within 1 %)
  static constexpr int STRIDE = kEnd + ((kStrideMod - kEnd % 32) % 32 + 32) % 32;
  static_assert(STRIDE >= kEnd, "tile stride");
  __host__ __device__ static constexpr int pad_at(int x) { return oPad + (x & 15); }
};

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
  // (Measured and rejected in round 2: every lane forming q_uu itself right after pass 1 so that the LLT / ladder chain starts one LDS
  // round trip earlier -- C3 142 us per launch against 132: the 36 extra FP64 instructions per lane cost more than the round trip saves.)
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
  // Passes 1 and 2 are length-NX dot products row . column; an item is a BR x BC BLOCK of them -- adjacent row vectors against
  // adjacent column vectors.  16 lanes per agent: 2 x 2, 4 NX doubles read for four results where one result per item read 2 NX each
  // (round 3: with two wavefronts per SIMD -- 8192 bicycle6 agents -- the launch follows the CU's LDS bandwidth, not the instruction
  // count).  32 lanes per agent (the tail rounds' form, where a launch is as long as ONE wavefront's chain of instructions): 2 x 1,
  // half the instructions per lane; 64 lanes: 1 x 1.
  //   pass 1   rows: the NX + NU vectors A_i, B_i of the tile (contiguous, NX apart);  columns: V_0 .. V_{NX-1}, v_x (contiguous, NX apart)
  //            -> A^T V, B^T V, and in the v_x column q_x = l_x + A^T v_x, q_u = l_u + B^T v_x (:115-117).  NX = 6, NU = 2, 2 x 2: 4 x 4 = 16 items.
  //   pass 2   rows: those of A^T V (1 apart, stride NX) or of B^T V (1 apart, stride NU);  columns: A_j or B_j
  //            -> q_xx, q_ux, q_uu = l + row . column (:117-119).  9 + 3 + 1 = 13 items.
  // An odd count's last block overlaps its neighbour (columns of pass 1) or runs one vector past its group (the rest): such a result is
  // formed from finite tile contents and written to a pad slot.
  constexpr int BR = (LPA >= 64) ? 1 : 2, BC = (LPA >= 32) ? 1 : 2, BN = BR * BC;
  constexpr int RP1 = (NX + NU + BR - 1) / BR, CN1 = NX + 1, CP1 = (CN1 + BC - 1) / BC, W1 = RP1 * CP1, R1 = (W1 + LPA - 1) / LPA;
  int p1[R1], q1[R1], b1[R1][BR], o1[R1][BN];     // result k: row r0 + k % BR, column c0 + k / BR; only the block's LAST column can be v_x, i.e. carry a bias
  MAS_UNROLL for (int r = 0; r < R1; ++r) {
    const int w = lane + r * LPA;
    const bool item = w < W1;
    const int rp = item ? w % RP1 : 0, cp = item ? w / RP1 : 0;
    const int r0 = BR * rp, c0 = (BC * cp + BC - 1 < CN1) ? BC * cp : CN1 - BC;
    p1[r] = TL::oD + L::oA + r0 * NX; q1[r] = TL::oV + c0 * NX;
    MAS_UNROLL for (int k = 0; k < BN; ++k) {
      const int rr = r0 + k % BR, cc = c0 + k / BR;
      const bool fresh = item && rr < NX + NU && cc >= BC * cp;                        // (an overlapping column was its neighbour's)
      const int dst = (cc < NX) ? (rr < NX ? TL::oAtV + cc * NX + rr : TL::oBtV + cc * NU + (rr - NX)) : (rr < NX ? TL::oQx + rr : TL::oQu + (rr - NX));
      o1[r][k] = fresh ? dst : TL::pad_at(w + 4 * k);
      if (k / BR == BC - 1) b1[r][k % BR] = (fresh && cc == NX) ? TL::oD + L::oLx + rr : TL::oNegZero;     // (l_u follows l_x in the tile)
    }
  }
  static_assert(L::oLu == L::oLx + NX && L::oB == L::oA + NX * NX, "tile layout: B after A, l_u after l_x");
  constexpr int XP = (NX + BR - 1) / BR, UP = (NU + BR - 1) / BR, XC = (NX + BC - 1) / BC, UC = (NU + BC - 1) / BC;
  constexpr int W2xx = XP * XC, W2ux = W2xx + UP * XC, W2 = W2ux + UP * UC, R2 = (W2 + LPA - 1) / LPA;
  int p2[R2], ps2[R2], q2[R2], b2[R2][BN], o2[R2][BN];   // result k = T[b] + sum_m T[p + k % BR + m*ps] * T[q + (k / BR) NX + m]
  MAS_UNROLL for (int r = 0; r < R2; ++r) {
    const int w = lane + r * LPA;
    const bool item = w < W2;
    const int kind = !item ? 0 : (w < W2xx) ? 0 : (w < W2ux) ? 1 : 2;                 // q_xx / q_ux / q_uu
    const int e = (kind == 0) ? (item ? w : 0) : (kind == 1) ? w - W2xx : w - W2ux;
    const int rows = (kind == 0) ? NX : NU, cols = (kind == 2) ? NU : NX, rpn = (kind == 0) ? XP : UP;
    const int r0 = BR * (e % rpn), c0 = BC * (e / rpn);
    p2[r] = ((kind == 0) ? TL::oAtV : TL::oBtV) + r0; ps2[r] = rows;
    q2[r] = TL::oD + ((kind == 2) ? L::oB : L::oA) + c0 * NX;
    const int out0 = (kind == 0) ? TL::oQxx : (kind == 1) ? TL::oQux : TL::oQuu, bias0 = TL::oD + ((kind == 0) ? L::oLxx : (kind == 1) ? L::oLux : L::oLuu);
    MAS_UNROLL for (int k = 0; k < BN; ++k) {
      const int rr = r0 + k % BR, cc = c0 + k / BR;
      const bool fresh = item && rr < rows && cc < cols;
      o2[r][k] = fresh ? out0 + cc * rows + rr : TL::pad_at(w + 4 * k);
      b2[r][k] = fresh ? bias0 + cc * rows + rr : TL::oNegZero;
    }
  }

  // software pipeline: the tiles of step t-1 travel HBM -> registers while step t is computed.  Every lane works for
  // its OWN agent: lane l fetches element c*16 + l of its agent's tile (chunk c holds the 4 agents' cw-element pieces back
  // to back, tile_offset) and publishes it to its agent's LDS tile, so each 16-lane write group fills 16 consecutive
  // doubles.  No predication: a surplus lane of the short last chunk re-reads the piece's last element and publishes it to
  // its pad slot; fetch and publish are NCH loads and NCH LDS writes with loop-invariant per-lane offsets and one uniform
  // base pointer that steps back by Np*ND doubles per time step.
  // With 32 / 64 lanes per agent, 2 / 4 chunks travel per load.
  constexpr int NCHUNK = (ND + kTileChunk - 1) / kTileChunk, SUB = LPA / kTileChunk, NCH = (NCHUNK + SUB - 1) / SUB;
  static_assert(LPA % kTileChunk == 0, "whole chunks per load");
  double pre[NCH];
  unsigned fetch_off[NCH]; int publish_at[NCH];
  MAS_UNROLL for (int cr = 0; cr < NCH; ++cr) {
    const int c_raw = cr * SUB + lane / kTileChunk, l16 = lane % kTileChunk;
    const int c = c_raw < NCHUNK ? c_raw : NCHUNK - 1;                 // (a lane group beyond the last chunk repeats it into pad slots)
    const int cw = (ND - c * kTileChunk < kTileChunk) ? (ND - c * kTileChunk) : kTileChunk;
    const bool mine = c_raw < NCHUNK && l16 < cw;
    const int e = l16 < cw ? l16 : cw - 1;
    fetch_off[cr] = static_cast<unsigned>((static_cast<size_t>(a / 64) * 64 * ND + static_cast<size_t>(c) * kTileChunk * 64 + static_cast<size_t>(a % 64) * cw + e) * sizeof(double));
    publish_at[cr] = tile * TL::STRIDE + (mine ? TL::oD + c * kTileChunk + l16 : TL::pad_at(TL::oD + c * kTileChunk + l16));
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
  unsigned goff[NU], koff[NU];                       // k rows; gain column `lane` (rows lane*NU + m) of the stage-A lanes
  MAS_UNROLL for (int m = 0; m < NU; ++m) goff[m] = static_cast<unsigned>((static_cast<size_t>(m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  MAS_UNROLL for (int m = 0; m < NU; ++m) koff[m] = static_cast<unsigned>((static_cast<size_t>((lane < NX ? lane : 0) * NU + m) * d.Np + static_cast<size_t>(a)) * sizeof(double));
  // ---- pass 3, stage A: column c = lane of [q_ux | q_u] in, of [K | k] and [K | k]^T q_uu out (lanes beyond NX repeat column 0 into their pad slots)
  static_assert(NX + 1 <= LPA, "one lane per column of [K | k]");
  const int a_qc = (lane < NX) ? TL::oQux + lane * NU : (lane == NX) ? TL::oQu : TL::oQux;
  int w_kc[NU], w_tc[NU];                            // (element by element: a pad slot is ONE double; column NX of K^T q_uu is not used)
  MAS_UNROLL for (int m = 0; m < NU; ++m) {
    w_kc[m] = (lane <= NX) ? TL::oK + lane * NU + m : TL::pad_at(TL::oK + lane * NU + m);
    w_tc[m] = (lane < NX) ? TL::oKtQ + lane * NU + m : TL::pad_at(TL::oKtQ + lane * NU + m);
  }
  // ---- pass 3, stage B: items = the NX (NX + 1) / 2 pairs of mirrored v_xx entries (i >= j, column by column) + the NX entries of v_x
  // (column j = NX of the extended matrices).  An item forms BOTH entries of its pair -- they share all their operands -- so that 36 + 6
  // results take 27 items = 2 rounds of 16 lanes for NX = 6 where one entry per lane took 3 (each of them forming its mirror as well).
  constexpr int NPAIR = NX * (NX + 1) / 2, W3 = NPAIR + NX, R3 = (W3 + LPA - 1) / LPA;
  static_assert(R3 <= 32, "is_vx bit field");
  int a_Ki[R3], a_Kj[R3], a_qi[R3], a_qj[R3], a_ti[R3], a_tj[R3], a_own[R3], a_mir[R3], out_lo[R3], out_up[R3];
  unsigned is_vx = 0u;
  MAS_UNROLL for (int r = 0; r < R3; ++r) {
    const int w = lane + r * LPA;
    int i = 0, j = 0, rem = w;
    bool found = false;
    MAS_UNROLL for (int c = 0; c < NX; ++c) {
      if (!found && rem < NX - c) { j = c; i = c + rem; found = true; }
      if (!found) rem -= NX - c;
    }
    const bool vx = !found && rem < NX;               // rem = w - NPAIR here
    if (vx) { i = rem; j = NX; }
    if (!found && !vx) { i = 0; j = 0; }              // idle lane of the last round: reads item 0's operands, writes its pad slots
    a_Ki[r] = TL::oK + i * NU; a_Kj[r] = TL::oK + j * NU;
    a_qi[r] = TL::oQux + i * NU; a_qj[r] = (j < NX) ? TL::oQux + j * NU : TL::oQu;
    a_ti[r] = TL::oKtQ + i * NU; a_tj[r] = TL::oKtQ + (j < NX ? j : 0) * NU;
    a_own[r] = (j < NX) ? TL::oQxx + j * NX + i : TL::oQx + i;
    a_mir[r] = (j < NX) ? TL::oQxx + i * NX + j : TL::oQx + i;
    out_lo[r] = (valid && found) ? TL::oV + j * NX + i : (valid && vx) ? TL::oVx + i : TL::pad_at(TL::oV + w);
    out_up[r] = (valid && found && i != j) ? TL::oV + i * NX + j : TL::pad_at(TL::oV + w + 8);
    if (vx) is_vx |= 1u << r;
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
    MAS_UNROLL for (int m = 0; m < NU; ++m) MAS_OPAQUE32(koff[m]);
    fetch();
    wave_sync();

    // ---- pass 1 (ilqr.hpp:115-117: A^T V, B^T V, q_x, q_u)
    {
      double res[R1][BN];
      MAS_UNROLL for (int r = 0; r < R1; ++r) {
        double pr[BR][NX], qc[BC][NX];
        MAS_UNROLL for (int k = 0; k < NX; ++k) {
          MAS_UNROLL for (int br = 0; br < BR; ++br) pr[br][k] = T[p1[r] + br * NX + k];
          MAS_UNROLL for (int bc = 0; bc < BC; ++bc) qc[bc][k] = T[q1[r] + bc * NX + k];
        }
        MAS_UNROLL for (int k = 0; k < BN; ++k) {
          const double s = sum_terms<NX>([&](int m) { return pr[k % BR][m] * qc[k / BR][m]; });
          res[r][k] = (k / BR == BC - 1) ? T[b1[r][k % BR]] + s : s;    // l_x / l_u in the v_x column, -0.0 (adds nothing, keeps every bit) elsewhere
        }
      }
      MAS_UNROLL for (int r = 0; r < R1; ++r) { MAS_UNROLL for (int k = 0; k < BN; ++k) T[o1[r][k]] = res[r][k]; }      // idle results: own pad slots
    }
    wave_sync();
    double quu[NU * NU], Lfac[NU * NU], ninv[NU * NU];
    constexpr bool kLaneExchange = NU <= 2;     // q_uu and the winning factor travel lane to lane (DPP / ds_bpermute), not through the tile
    // ---- pass 2 (ilqr.hpp:117-119: q_xx, q_ux, q_uu)
    {
      double res[R2][BN];
      MAS_UNROLL for (int r = 0; r < R2; ++r) {
        double pr[BR][NX], qc[BC][NX];
        MAS_UNROLL for (int k = 0; k < NX; ++k) {
          MAS_UNROLL for (int br = 0; br < BR; ++br) pr[br][k] = T[p2[r] + k * ps2[r] + br];
          MAS_UNROLL for (int bc = 0; bc < BC; ++bc) qc[bc][k] = T[q2[r] + bc * NX + k];
        }
        MAS_UNROLL for (int k = 0; k < BN; ++k) {
          const double s = sum_terms<NX>([&](int m) { return pr[k % BR][m] * qc[k / BR][m]; });
          res[r][k] = T[b2[r][k]] + s;
        }
      }
      MAS_UNROLL for (int r = 0; r < R2; ++r) { MAS_UNROLL for (int k = 0; k < BN; ++k) T[o2[r][k]] = res[r][k]; }
      if constexpr (LPA == 16 || kLaneExchange) {
        // q_uu straight from the lanes that formed it: the LLT / ladder chain starts without waiting for the LDS round trip of the
        // other pass-2 results
        tile_gather_quu<NU, W2ux, UP, BR, BC, LPA>(res, quu, tid & ~(LPA - 1), std::make_integer_sequence<int, NU * NU>{});
      }
    }
    if constexpr (!(LPA == 16 || kLaneExchange)) {
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
        if constexpr (kLaneExchange) {
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
      llt_solve_identity<NU>(Lfac, ninv);                           // Quu_reg^-1, :183
      MAS_UNROLL for (int i = 0; i < NU * NU; ++i) ninv[i] = -ninv[i];
      // ---- stage A: lane c <= NX forms column c of [K | k] = -Quu^-1 [q_ux | q_u] (:185-186) and of its product with the
      // unregularised q_uu; the gains leave for HBM from here (K by lanes 0..NX-1, k by lane NX), once per step
      {
        double qc[NU], Kc[NU], tc[NU];
        MAS_UNROLL for (int m = 0; m < NU; ++m) qc[m] = T[a_qc + m];
        MAS_UNROLL for (int m = 0; m < NU; ++m) Kc[m] = sum_terms<NU>([&](int q) { return ninv[q * NU + m] * qc[q]; });
        MAS_UNROLL for (int m = 0; m < NU; ++m) tc[m] = sum_terms<NU>([&](int q) { return Kc[q] * quu[m * NU + q]; });
        MAS_UNROLL for (int m = 0; m < NU; ++m) { T[w_kc[m]] = Kc[m]; T[w_tc[m]] = tc[m]; }
        if (lane < NX) { MAS_UNROLL for (int m = 0; m < NU; ++m) st_off(Kfb_t, koff[m], Kc[m]); }
        if (lane == NX) { MAS_UNROLL for (int m = 0; m < NU; ++m) st_off(kff_t, goff[m], Kc[m]); }
      }
      wave_sync();
      // ---- stage B: v_xx and v_x (:188-192), one item per pair of mirrored entries (i >= j) and one per v_x entry
      MAS_UNROLL for (int r = 0; r < R3; ++r) {
        double Ki[NU], Kj[NU], qi[NU], qj[NU], ti[NU], tj[NU];
        MAS_UNROLL for (int m = 0; m < NU; ++m) {
          Ki[m] = T[a_Ki[r] + m]; Kj[m] = T[a_Kj[r] + m]; qi[m] = T[a_qi[r] + m]; qj[m] = T[a_qj[r] + m]; ti[m] = T[a_ti[r] + m]; tj[m] = T[a_tj[r] + m];
        }
        const double q_own = T[a_own[r]], q_mir = T[a_mir[r]];
        // entry (i,j) -- for a v_x item: q_x[i] + K_i.q_u + q_ux,i.k + (K^T q_uu)_i.k, the same expression with column NX of [K | k], [q_ux | q_u]
        double s, own, mir;
        s = sum_terms<NU>([&](int m) { return Ki[m] * qj[m]; });
        own = q_own + s;
        s = sum_terms<NU>([&](int m) { return qi[m] * Kj[m]; });
        own = own + s;
        s = sum_terms<NU>([&](int m) { return ti[m] * Kj[m]; });
        own = own + s;
        // entry (j,i)
        s = sum_terms<NU>([&](int m) { return Kj[m] * qi[m]; });
        mir = q_mir + s;
        s = sum_terms<NU>([&](int m) { return qj[m] * Ki[m]; });
        mir = mir + s;
        s = sum_terms<NU>([&](int m) { return tj[m] * Ki[m]; });
        mir = mir + s;
        // v_xx = 0.5 (v_xx + v_xx^T) in the reference's aliased in-place order, :192: the lower (or diagonal) entry averages the old
        // values, the upper entry sees its already rewritten mirror
        const double lower = 0.5 * (own + mir);
        const double upper = 0.5 * (mir + lower);
        T[out_lo[r]] = (is_vx >> r & 1u) ? own : lower;
        T[out_up[r]] = upper;
      }
    }
    // the barrier after the next publish orders these writes before pass 1 reads them
    kff_t -= static_cast<size_t>(NU) * d.Np; Kfb_t -= static_cast<size_t>(NU) * NX * d.Np;
  };
  step(d.T - 1);
  for (int t = d.T - 2; t >= 0; --t) step(t);
  if (counted && lane == 0) reg_events[a] += regs;
}

}  // namespace mas
