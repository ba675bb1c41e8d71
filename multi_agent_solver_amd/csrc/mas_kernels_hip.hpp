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
//  * mas_backward_tile_kernel    backward recursion for any NX over an LDS tile per agent, 16 lanes per agent (32 / 64 when few agents
//                                are live): the small dense products of ilqr.hpp:115-119,183-192 are dealt to the lanes as blocks of
//                                dot products, the gains are formed once per step, v_xx by pairs of mirrored entries.
// All backward kernels compute, element for element, the arithmetic of BackwardBody (the one-lane form the host emulation
// replays): every output element is the same sequence of multiplies and adds, sums in k order (sum_terms: from the first term).
// Their time-step loops are written against the gfx950 ISA they compile to: a lone wavefront per SIMD pays 5.5-7 clock ticks per
// instruction whatever its instruction-level parallelism, so the loops avoid selects (loop-invariant -0.0 addends), predication
// (clamped prefetch lanes), vector address arithmetic (scalar base + 32-bit lane offsets) and LDS round trips.
//
// No MFMA: v_mfma_f64_4x4x4 fuses the multiply-adds of a dot product, the reference (SSE2, no FMA;
// CMakeLists.txt sets no -march) rounds after every multiply -- and the algorithm amplifies last-bit
// differences chaotically (include/mas_dettrig.h), so a fused product would break parity.
//
// Files (round 3: one per kernel family):
//   mas_kernel_sweep.hpp             mas_sweep_kernel
//   mas_kernel_quad.hpp              quad_perm / QuadRk4, mas_forward_quad_kernel, mas_rollout_quad_kernel, mas_trust_reroll_quad_kernel
//   mas_kernel_backward_common.hpp   row_bcast, llt_solve_identity_by_columns, sum_terms, LadderSync (the rung agreement)
//   mas_kernel_backward_tile.hpp     BackwardTileLayout, mas_backward_tile_kernel
//   mas_kernel_backward_row.hpp      mas_backward_row_kernel, mas_backward_row_pipelined_kernel
//   mas_kernel_levels.hpp            open-loop rollouts by dependency levels (all time steps of a level side by side), mas_trust_epilogue_kernel
//   mas_fused_hip.hpp                mas_linearise_riccati_kernel (the two-kernel round of the four-state models; includes this file)
#pragma once

#include <type_traits>
#include <utility>

#include "mas_kernel_sweep.hpp"
#include "mas_kernel_quad.hpp"
#include "mas_kernel_backward_common.hpp"
#include "mas_kernel_backward_tile.hpp"
#include "mas_kernel_backward_row.hpp"
#include "mas_kernel_levels.hpp"
