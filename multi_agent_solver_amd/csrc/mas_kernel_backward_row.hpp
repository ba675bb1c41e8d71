// mas_kernel_backward_row.hpp -- backward Riccati recursion for NX == 4 in registers + DPP, and its pipelined shared-ladder form (HIP build only; see mas_kernels_hip.hpp)
#pragma once

#include <type_traits>
#include <utility>

#include "mas_kernel_backward_common.hpp"

namespace mas {

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
  constexpr bool kFactorByPermute = NU <= 2;
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
    MAS_OPAQUE32(oAi); MAS_OPAQUE32(oAj); MAS_OPAQUE32(oBi); MAS_OPAQUE32(oBj);
    MAS_OPAQUE32(olx); MAS_OPAQUE32(olu); MAS_OPAQUE32(olxx); MAS_OPAQUE32(olux); MAS_OPAQUE32(oluu);
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

  // The agreement, split in two: arrive and wait.  In a workgroup of several wavefronts ONE wavefront (the last) posts the combined
  // mask and ANOTHER (the first) reads the word, so that neither carries both on top of the common work: measured per step on C2,
  // wave 0 spent 860 + 510 cycles around its atomics and its polls where the others spent 420 + 185, and everybody waits for the
  // slowest at the workgroup barrier.
  const bool one_group = n_groups == 1;
  const bool poster = (WPB == 1) || wave == WPB - 1, poller = ((WPB == 1) || wave == 0) && !one_group;
  // combine the masks the wavefronts left in blk_field[buf] and post them (behind a barrier that follows their writes)
  auto post_combined = [&](int word, unsigned buf) {
    if (poster) {
      unsigned bf = 0xFFFFu;
      MAS_UNROLL for (int wv = 0; wv < WPB; ++wv) bf &= blk_field[buf][wv];
      if (one_group) { if (tid == 0) blk_agreed[buf] = bf; }
      else ladder_post(sync, word, bf);
    }
  };
  auto post = [&](int word, unsigned field) {
    if constexpr (WPB > 1) {
      if (tid == 0) blk_field[agreement_parity][wave] = field;
      __syncthreads();
      post_combined(word, agreement_parity);
    } else {
      ladder_post(sync, word, field);
    }
  };
  // first: the word as ladder_peek returned it earlier (0: not requested)
  auto wait = [&](int word, unsigned long long first = 0ull) -> unsigned {
    unsigned agreed;
    if constexpr (WPB > 1) {
      if (poller) { const unsigned g = ladder_wait(sync, word, n_groups, first); if (tid == 0) blk_agreed[agreement_parity] = g; }
      __syncthreads();
      agreed = blk_agreed[agreement_parity];
      agreement_parity ^= 1u;
    } else {
      agreed = ladder_wait(sync, word, n_groups, first);
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
  unsigned long long seen = 0ull;                               // the agreement word as requested early (written by that load alone: see below)
#if defined(MAS_LADDER_STATS)
  unsigned long long cy_back = 0, cy_front = 0, cy_wait = 0, cy_redo = 0, cy_post = 0, cy_top = 0, c_prev = __builtin_readcyclecounter(), st_redo = 0;
#define MAS_LS(x) x
#else
#define MAS_LS(x)
#endif
  // one step: `cur` = front(t) (settled, posted), `tile_next` holds tile t-1, `tile_refill` held tile t and is refilled with t-2
  auto iterate = [&](int t, int par, Front& cur, Front& nxt, Tile& tile_next, Tile& tile_refill) {
    if (t > 1) fetch(tile_refill, t - 2);                       // tile t is no longer needed (front(t) is settled): a whole step ahead of its use
    const bool speculate = (cur.field & 1u) != 0u;             // rung 0 works for every agent of this wavefront
    MAS_LS(const unsigned long long c0 = __builtin_readcyclecounter(); unsigned long long c1 = c0; unsigned long long c2 = c0;)
    // The agreement word of step t is REQUESTED between the two halves of the speculative work and looked at behind them: the
    // other workgroups posted their masks of step t when this one did (~0.4 us to L2), so a load issued after back(t) usually
    // returns the complete word while front(t-1) runs (98 of 100 steps on C2), and the step costs max(work, post + read)
    // instead of work + read.  `seen` is written by that load alone and first touched behind front(t-1) (the empty asm pins
    // that point): a clear or a select in between makes the compiler wait for EVERY load and store in flight -- the tile prefetch,
    // the gain stores, and the atomics of the last post, whose acknowledgements then sit on the chain of every step.
    if (speculate) {
      back(t, par, cur, cur.Lcand, 0);
      MAS_LS(c1 = __builtin_readcyclecounter();)
      if (poller && sync.peek != 0) seen = ladder_peek(sync, t * LadderSync::kChunks);
      if (t > 0) {
        front(tile_next, par ^ 1, nxt);
        // the mask of step t-1 goes to LDS at once: if the agreement of step t is rung 0 -- the common case -- the barrier inside wait()
        // publishes it and the posting wavefront needs no barrier of its own
        if constexpr (WPB > 1) { if (tid == 0) blk_field[agreement_parity ^ 1u][wave] = nxt.field; }
      }
      MAS_LS(c2 = __builtin_readcyclecounter();)
    }
    asm volatile("" : "+v"(seen));
    const unsigned agreed = wait(t * LadderSync::kChunks, (speculate && sync.peek != 0) ? seen : 0ull);
    MAS_LS(const unsigned long long c3 = __builtin_readcyclecounter(); cy_back += c1 - c0; cy_front += c2 - c1; cy_wait += c3 - c2;)
    int win;
    if (agreed != 0u) win = __builtin_ctz(agreed); else settle_beyond(t, cur, win);
    if (!(speculate && win == 0)) {                             // not speculated, or speculated wrongly: the step with the agreed rung
      MAS_LS(st_redo++;)
      back(t, par, cur, cur.Lcand, win & (LL - 1));
      if (t > 0) {
        front(tile_next, par ^ 1, nxt);
        if constexpr (WPB > 1) { if (tid == 0) blk_field[agreement_parity][wave] = nxt.field; __syncthreads(); }
      }
    }
    if (lane == 0 && a == 0) regs += win;                       // the stacked solve counts its rungs once (agent 0)
    MAS_LS(const unsigned long long c4 = __builtin_readcyclecounter();)
    if (t > 0) {
      if constexpr (WPB > 1) post_combined((t - 1) * LadderSync::kChunks, agreement_parity);
      else ladder_post(sync, (t - 1) * LadderSync::kChunks, nxt.field);
    }
    MAS_LS(const unsigned long long c5 = __builtin_readcyclecounter(); cy_redo += c4 - c3; cy_post += c5 - c4; cy_top += c0 - c_prev; c_prev = c5;)
  };
  int t = d.T - 1;
  fetch(tileA, t);
  if (t > 0) fetch(tileB, t - 1);
  front(tileA, 0, frA);
  post(t * LadderSync::kChunks, frA.field);
  for (; t >= 1; t -= 2) { iterate(t, 0, frA, frB, tileB, tileA); iterate(t - 1, 1, frB, frA, tileA, tileB); }
  if (t == 0) iterate(0, 0, frA, frB, tileB, tileA);
  if (valid && lane == 0) reg_events[a] += regs;
#if defined(MAS_LADDER_STATS)
  if (tid == 0 && group == 0)
    printf("cycles wave %d: top(fetch) %llu back %llu front %llu wait %llu redo %llu post %llu; redone %llu\n", wave, cy_top, cy_back, cy_front, cy_wait, cy_redo, cy_post, st_redo);
#endif
#undef MAS_LS
}

}  // namespace mas
