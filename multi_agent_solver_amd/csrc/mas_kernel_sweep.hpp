// mas_kernel_sweep.hpp -- linearisation sweep as a kernel of its own (HIP build only; see mas_kernels_hip.hpp for the kernel map)
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
// COMMIT (the centralised strategy's iterations, no live list): the sweep also COMMITS the trial the previous iteration's select accepted
// (x = best_x; u = best_u, ilqr.hpp:231-232).  A lane whose agent has alpha_sel >= 0 takes x_t, u_t from that trial plane instead of X, U
// and stores them to X, U (the lanes of the last time step also x_T) -- whether or not the agent is still active --, so the copy kernel
// between select and the next sweep goes away: the values linearised are the values the copy would have put there.
struct SweepCommit { const double* Xs = nullptr; const double* Us = nullptr; const int* alpha_sel = nullptr; double* Xw = nullptr; double* Uw = nullptr; };
template <class M, bool COMMIT = false>
__global__ void __launch_bounds__(256, 2) mas_sweep_kernel(Dims d, ModelParams P, const double* __restrict__ X, const double* __restrict__ U,
                                                        double* __restrict__ D, const int* __restrict__ active, int lo, int n,
                                                        unsigned long long* __restrict__ stamp,
                                                        const int* __restrict__ live_list, const int* __restrict__ live_count, SweepCommit cm) {
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
  int jsel = -1;
  if constexpr (COMMIT) { if (a >= 0 && a < n) jsel = cm.alpha_sel[a]; }
  if constexpr (COMMIT) {
    if (mask == 0ull && __ballot(jsel >= 0) == 0ull) return;
  } else {
    if (mask == 0ull) return;                              // wave-uniform: nothing to linearise in this group
  }
  const int a_first = __builtin_amdgcn_readfirstlane(a);   // agent of the first active lane
  const bool one_block = (live_list == nullptr) ||
                         (mask == ~0ull && (a_first & 63) == 0 && __ballot(a == a_first + lane) == ~0ull);

  double Dl[ND];
  if constexpr (COMMIT) {
    double x[NX], u[NU];
    const bool pend = jsel >= 0;
    if (act || pend) {
      const double* Xr = pend ? cm.Xs + static_cast<size_t>(jsel) * (d.T + 1) * NX * d.Np : X;
      const double* Ur = pend ? cm.Us + static_cast<size_t>(jsel) * d.T * NU * d.Np : U;
      MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = MAS_AT(Xr, t * NX + i, a);
      MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = MAS_AT(Ur, t * NU + i, a);
      if (pend) {
        MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(cm.Xw, t * NX + i, a) = x[i];
        MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(cm.Uw, t * NU + i, a) = u[i];
        if (t == d.T - 1) { MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(cm.Xw, d.T * NX + i, a) = MAS_AT(Xr, d.T * NX + i, a); }
      }
    }
    if (mask == 0ull) return;                              // wave-uniform: only copies in this group
    if (act) linearise<M>(x, u, t, Dl, P);
    else { MAS_UNROLL for (int i = 0; i < ND; ++i) Dl[i] = 0.0; }
  } else {
  if (act) {
    double x[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = MAS_AT(X, t * NX + i, a);
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = MAS_AT(U, t * NU + i, a);
    linearise<M>(x, u, t, Dl, P);
  } else {
    MAS_UNROLL for (int i = 0; i < ND; ++i) Dl[i] = 0.0;
  }
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

}  // namespace mas
