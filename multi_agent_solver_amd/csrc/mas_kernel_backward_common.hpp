// mas_kernel_backward_common.hpp -- what the backward kernels share: DPP row helpers, column-wise LLT solve, sum_terms, the rung agreement (HIP build only)
#pragma once

#include <type_traits>
#include <utility>

#include "mas_kernel_quad.hpp"

namespace mas {

__device__ __forceinline__ double lane_read(double v, int src_lane) {      // v of lane src_lane of the wavefront (ds_bpermute, no memory)
  const int lo32 = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2loint(v)), hi32 = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2hiint(v));
  return __hiloint2double(hi32, lo32);
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
// q_uu entries from the pass-2 results of the LDS-tile kernel (items = BR x BC blocks of results, LPA lanes per agent): entry
// e = j*NU + i is result (i % BR) + BR (j % BC) of block (i / BR, j / BC), item W0 + (j / BC) * UP + i / BR, formed in round
// item / LPA by lane item % LPA of the agent.  16 lanes per agent are one DPP row (row broadcast); wider agents read the lane
// through ds_bpermute (agent_lane0 = first lane of the agent in the wavefront).
template <int NU, int W0, int UP, int BR, int BC, int LPA, int R2, int... E>
__device__ __forceinline__ void tile_gather_quu(const double (&res)[R2][BR * BC], double* quu, int agent_lane0, std::integer_sequence<int, E...>) {
  if constexpr (LPA == 16) {
    (void)agent_lane0;
    ((quu[E] = row_bcast<(W0 + ((E / NU) / BC) * UP + (E % NU) / BR) % 16>(res[(W0 + ((E / NU) / BC) * UP + (E % NU) / BR) / 16][((E % NU) % BR) + BR * ((E / NU) % BC)])), ...);
  } else {
    ((quu[E] = lane_read(res[(W0 + ((E / NU) / BC) * UP + (E % NU) / BR) / LPA][((E % NU) % BR) + BR * ((E / NU) % BC)],
                         agent_lane0 + (W0 + ((E / NU) / BC) * UP + (E % NU) / BR) % LPA)), ...);
  }
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


// Per time step the wavefront makes three passes over its LDS tiles (four agents x 16 lanes, two x 32 or one x 64), every lane
// running the same instruction stream on its own operand addresses (no divergence inside a pass):
//   pass 1   A^T V, B^T V, q_x, q_u                       (length-NX dot products, a block of them per lane)
//   pass 2   q_xx, q_ux, q_uu                             (length-NX dot products, a block of them per lane)
//   pass 3   LLT + regularisation ladder and Quu^-1 in registers (every lane: 16 rungs at once); stage A: lane c forms column c
//            of [K | k] and of its product with q_uu, once; stage B: one lane per PAIR of mirrored v_xx entries (and per v_x
//            entry) applies the value update and the reference's aliased in-place symmetrisation rule.
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
  int peek = 1;                          // pipelined kernel: request the agreement word between the halves of the speculative step (ladder_peek)
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
// first: a value of the word that was requested EARLIER (ladder_peek: its round trip to L2 has passed behind other work); 0 = none.
// A complete counter in it settles the agreement without a poll.
__device__ inline unsigned long long ladder_peek(const LadderSync& sy, int slot) {
  return __hip_atomic_load(sy.words + static_cast<size_t>(slot) * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline unsigned ladder_wait(const LadderSync& sy, int slot, int n_waves, unsigned long long first = 0ull) {
  unsigned agreed = 0u;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* w = sy.words + static_cast<size_t>(slot) * 4;
    unsigned long long v = first;
    int spins = 0;
    for (;;) {
      if (static_cast<int>(v >> LadderSync::kArrivalShift) == n_waves) break;
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

}  // namespace mas
