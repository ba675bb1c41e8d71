// mas_bodies.hpp -- kernel bodies of the batched iLQR path over the SoA device layout.
//
// Device layout: trajectories, gains and per-agent scalars are stored
// "structure of arrays" with the AGENT index fastest:  element (row r, agent a) of an array with R
// rows lives at base[r * Np + a], Np = N rounded up to a multiple of 64.  A wavefront's 64 lanes
// are 64 consecutive agents, so each load/store instruction touches 64 consecutive doubles (512 B,
// fully coalesced).  Rows are (t, component) pairs, t-major.  The linearisation output is the
// exception: it is an array of per-(t, agent) tiles in a chunk-major block layout (tile_offset below).
//
// Each body is a functor called with (agent) or (agent, y); y is a time step (linearisation
// sweep, commit) or an alpha-trial index (speculative line search).
#pragma once

#include "mas_lane.hpp"

namespace mas {

constexpr int kNumAlphas = 10;   // alpha = 1, 1/2, ... while alpha >= 1e-3 (ilqr.hpp:199-206): 2^0 .. 2^-9

enum ExitCode : int { kExitConverged = 0, kExitMaxIter = 1, kExitTime = 2, kExitNonFinite = 4 };

struct Dims {
  int N;        // agents
  int T;        // horizon_steps
  size_t Np;    // padded agent count (row stride)
};

// Augmented-Lagrangian state of the solver objects (ilqr.hpp:415-421,445-453), one per agent and persistent
// across solves: multipliers lam[T*NEQ][Np], mu[T*NIN][Np], penalty rho[Np].  `on` is false for unconstrained
// models and for the centralised strategy (the stacked OCP carries no constraint functions,
// multi_agent_problem.hpp:52-127).
struct ALView {
  bool on = false;
  double* lam = nullptr; double* mu = nullptr; double* rho = nullptr;
  double act_tol = 1e-6;
};

// (uniform row pointer)[lane index]: lets the compiler address with a scalar base + 32-bit lane offset
#define MAS_AT(base, row, a) ((base) + static_cast<size_t>(row) * d.Np)[static_cast<unsigned>(a)]
// uniform (scalar) base pointer + 32-bit per-lane byte offset: the addressing form gfx950 executes with
// no vector address arithmetic (global_load/store v, v_off32, s[base])
MAS_HD double ld_off(const double* base, unsigned byte_off) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off); }
MAS_HD void st_off(double* base, unsigned byte_off, double v) { *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + byte_off) = v; }
// Derivative tiles D(t, agent) = ND doubles (A,B,l_x,l_u,l_xx,l_uu,l_ux).  Storage is chunk-major
// inside blocks of 64 agents:
//     D[t][agent/64][chunk c][agent%64][e%16],   chunk c = e/16, chunk width cw_c = min(16, ND - 16c)
// i.e. element e of tile (t,a) sits at  ((t*Np/64 + a/64)*64*ND) + c*16*64 + (a%64)*cw_c + e%16.
//  * the sweep (a wavefront = 64 agents of one step) stages 16 elements per lane through LDS and writes
//    each chunk as ONE contiguous, 64-byte-aligned run of 64*cw_c doubles (8 KiB): full-line stores only;
//  * the backward pass (a wavefront = 4 consecutive agents) reads 4*cw_c contiguous doubles per chunk.
constexpr int kTileChunk = 16;
// lanes that share one agent's backward recursion (mas_backward_tile_kernel): 16 or 32; 64 / this = agents per wavefront
constexpr int kBackwardLanesPerAgent = 16;   // measured on C3: 16 -> 145 us per launch, 32 -> 165 us
constexpr int kBackwardWavesPerBlock = 4;    // per-agent backward pass: wavefronts per workgroup (placement: one per SIMD of a CU; measured 1 / 2 / 8: 6.51 / 7.63 / 7.69 against 6.11 ms per C3 run)
constexpr int kForwardTrialsPerBlock = 1;    // line-search rollouts: alpha trials (wavefronts) per workgroup.  Measured on C3 (us per launch, full rounds): 1 -> 125, 2 -> 190 (two workgroups double up on the same SIMDs of a CU), 4 -> 140 (four issue-bound wavefronts share a CU's front end while 96 CUs idle), 3 -> 128
// centralised backward pass: wavefronts per workgroup.  Measured on C2 (256 agents, us per launch): 1 -> 416, 4 -> 293, 8 -> 330,
// 16 -> 485: fewer arrivals per agreement against more wavefronts sharing a SIMD
constexpr int kCentralWavesPerBlock = 4;
MAS_HD size_t tile_offset(int nd, size_t Np, int t, int a, int e) {
  const int c = e / kTileChunk, cw = (nd - c * kTileChunk < kTileChunk) ? (nd - c * kTileChunk) : kTileChunk;
  return (static_cast<size_t>(t) * (Np / 64) + static_cast<size_t>(a / 64)) * 64 * nd + static_cast<size_t>(c) * kTileChunk * 64 +
         static_cast<size_t>(a % 64) * cw + static_cast<size_t>(e % kTileChunk);
}
#define MAS_TILE_AT(base, ND_, t, a, e) (base)[tile_offset((ND_), d.Np, (t), (a), (e))]

// -------------------------------------------------------------------------------------------
// integrate_horizon + objective (ilqr.hpp:75-78; ocp.hpp:110,182): one lane per agent.
template <class M>
struct RolloutCostBody {
  Dims d; double dt; ModelParams P;
  const double* x0;      // [nx][Np]
  const double* U;       // [T*nu][Np]
  double* X;             // [(T+1)*nx][Np]
  double* cost;          // [Np]
  double* merit;         // [Np] (current merit == objective without constraint functions, ilqr.hpp:379-407)
  const int* active;     // optional mask
  ALView al;
  MAS_HD void operator()(int a) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (active && !active[a]) return;
    double x[NX], u[NU], xn[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = MAS_AT(x0, i, a); MAS_AT(X, i, a) = x[i]; }
    double c = 0.0;
    for (int t = 0; t < d.T; ++t) {
      MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = MAS_AT(U, t * NU + i, a);
      c += M::stage_cost(x, u, t, P);
      rk4_step<M>(x, u, dt, xn, P);
      MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = xn[i]; MAS_AT(X, (t + 1) * NX + i, a) = x[i]; }
    }
    c += M::terminal_cost(x, P);
    cost[a] = c;
    if (merit) {
      double m = c;
      if constexpr (M::NEQ + M::NIN > 0) {
        if (al.on) {                                   // compute_merit adds the constraint terms AFTER the whole objective, step by step
          const double rho = al.rho[a];
          for (int t = 0; t < d.T; ++t) {
            double xt[NX], ut[NU], lam[M::NEQ > 0 ? M::NEQ : 1], mu[M::NIN > 0 ? M::NIN : 1];
            MAS_UNROLL for (int i = 0; i < NX; ++i) xt[i] = MAS_AT(X, t * NX + i, a);
            MAS_UNROLL for (int i = 0; i < NU; ++i) ut[i] = MAS_AT(U, t * NU + i, a);
            MAS_UNROLL for (int k = 0; k < M::NEQ; ++k) lam[k] = MAS_AT(al.lam, t * M::NEQ + k, a);
            MAS_UNROLL for (int k = 0; k < M::NIN; ++k) mu[k] = MAS_AT(al.mu, t * M::NIN + k, a);
            m = al_merit_step<M>(m, xt, ut, lam, mu, rho, al.act_tol, P);
          }
        }
      }
      merit[a] = m;
    }
  }
};

// -------------------------------------------------------------------------------------------
// Linearisation sweep (ilqr.hpp:106-113), one lane per (agent, t): the time steps of one
// trajectory are independent, which the reference never exploits.  Streaming kernel:
// reads x_t,u_t, writes the ND derivative rows of step t.
template <class M>
struct SweepBody {
  Dims d; ModelParams P;
  const double* X; const double* U;
  double* D;             // derivative tiles (tile_offset)
  const int* active;
  MAS_HD void operator()(int a, int t) const {
    using L = DerivLayout<M>;
    constexpr int NX = M::NX, NU = M::NU;
    if (!active[a]) return;
    double x[NX], u[NU], Dl[L::ND];
    MAS_UNROLL for (int i = 0; i < NX; ++i) x[i] = MAS_AT(X, t * NX + i, a);
    MAS_UNROLL for (int i = 0; i < NU; ++i) u[i] = MAS_AT(U, t * NU + i, a);
    linearise<M>(x, u, t, Dl, P);
    MAS_UNROLL for (int i = 0; i < L::ND; ++i) MAS_TILE_AT(D, L::ND, t, a, i) = Dl[i];
  }
};

// -------------------------------------------------------------------------------------------
// Backward Riccati recursion (ilqr.hpp:92-193), one lane per agent, sequential in t.  This scalar
// form is what the host emulation replays; on the GPU the same recursion runs 16 lanes per agent
// over LDS tiles (mas_kernels_hip.hpp, BackwardTile), element for element the same arithmetic.
template <class M>
struct BackwardBody {
  Dims d; ModelParams P;
  const double* X; const double* D;
  double* kff;           // [T*nu][Np]
  double* Kfb;           // [T*nu*nx][Np], row (t, c*NU + i) = K_t(i, c)
  long long* reg_events; // [Np]
  const int* active;
  ALView al;
  MAS_HD void operator()(int a) const {
    using L = DerivLayout<M>;
    constexpr int NX = M::NX, NU = M::NU;
    if (!active[a]) return;
    double vx[NX], vxx[NX * NX], xT[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) xT[i] = MAS_AT(X, d.T * NX + i, a);
    terminal_boundary<M>(xT, vx, vxx, P);
    long long regs = 0;
    // software pipeline: the derivative rows of step t-1 are requested before step t is
    // computed, so their HBM latency hides behind ~1000 dependent FP64 instructions
    double Dl[L::ND], Dn[L::ND];
    MAS_UNROLL for (int i = 0; i < L::ND; ++i) Dl[i] = MAS_TILE_AT(D, L::ND, d.T - 1, a, i);
    for (int t = d.T - 1; t >= 0; --t) {
      const int tn = t > 0 ? t - 1 : 0;
      MAS_UNROLL for (int i = 0; i < L::ND; ++i) Dn[i] = MAS_TILE_AT(D, L::ND, tn, a, i);
      QFunction<M> q;
      assemble_q<M>(Dl, vx, vxx, q);
      if constexpr (M::NEQ + M::NIN > 0) {
        if (al.on) {
          double lam[M::NEQ > 0 ? M::NEQ : 1], mu[M::NIN > 0 ? M::NIN : 1];
          MAS_UNROLL for (int k2 = 0; k2 < M::NEQ; ++k2) lam[k2] = MAS_AT(al.lam, t * M::NEQ + k2, a);
          MAS_UNROLL for (int k2 = 0; k2 < M::NIN; ++k2) mu[k2] = MAS_AT(al.mu, t * M::NIN + k2, a);
          al_q_terms<M>(Dl, lam, mu, al.rho[a], al.act_tol, q);
        }
      }
      double Lfac[NU * NU], k[NU], K[NU * NX];
      regs += llt_ladder<NU>(q.quu, Lfac);
      gains_and_value<M>(q, Lfac, k, K, vx, vxx);
      MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(kff, t * NU + i, a) = k[i];
      MAS_UNROLL for (int i = 0; i < NU * NX; ++i) MAS_AT(Kfb, t * NU * NX + i, a) = K[i];
      MAS_UNROLL for (int i = 0; i < L::ND; ++i) Dl[i] = Dn[i];
    }
    reg_events[a] += regs;
  }
};

// Read-back of a round's counters: the commit is the last kernel of a round and runs after select has finished counting, so its
// first lane copies the counter pair (or the stacked solve's state) into host-visible pinned memory -- a copy engine / copy
// kernel per round (hipMemcpyAsync: ~4 us of kernel + ~6 us of gap in the trace) would sit on every round's critical path.
struct Publish {
  const long long* src = nullptr; long long* host = nullptr; int words = 0;
  const int* flag = nullptr;        // optional: one more word, host[words] = *flag (the centralised pass's "gave up waiting" flag)
  // tag != 0 (fused rounds, which the host POLLS for instead of waiting on an event): the counter pair {int active, long long
  // converged}, the flag and the tag travel as ONE 64-bit store -- [tag:15][flag:1][converged:24][active:24] -- so the host can
  // never see a tag without its counters
  int tag = 0;
  // centralised strategy (its host loop polls too): after the words, ONE more 64-bit store [poll_tag:32][..][flag:1][*poll_done != 0:1] to *poll_word
  long long* poll_word = nullptr; int poll_tag = 0; const int* poll_done = nullptr;
  static constexpr int kTagShift = 49, kFlagShift = 48, kConvShift = 24; static constexpr long long kMask24 = 0xFFFFFF;
  MAS_HD static long long pack(int tag_, int flag_, long long conv, long long act) {
    return (static_cast<long long>(tag_ & 0x7FFF) << kTagShift) | (static_cast<long long>(flag_ != 0) << kFlagShift) | ((conv & kMask24) << kConvShift) | (act & kMask24);
  }
  MAS_HD void operator()() const {
    if (tag != 0) { *reinterpret_cast<volatile long long*>(host) = pack(tag, flag ? *flag : 0, src[1], src[0] & 0xFFFFFFFFll); return; }
    for (int i = 0; i < words; ++i) host[i] = src[i];
    if (flag) host[words] = static_cast<long long>(*flag);
    if (poll_word) *reinterpret_cast<volatile long long*>(poll_word) = (static_cast<long long>(poll_tag) << 32) | ((flag && *flag != 0) ? 2ll : 0ll) | ((poll_done && *poll_done != 0) ? 1ll : 0ll);
  }
};

// -------------------------------------------------------------------------------------------
// Speculative line search (ilqr.hpp:195-228): one lane per (agent, alpha trial).  The
// reference tries alpha = 1, 1/2, ... one after the other and stops at the first strict
// decrease; here all ten trials roll out concurrently and SelectBody then picks the first
// (largest) alpha that decreases the merit -- the same decision, one sequential pass deep.
template <class M>
struct ForwardBody {
  Dims d; double dt; ModelParams P;
  bool clamp; double lb[M::NU], ub[M::NU];
  const double* x0; const double* X; const double* U; const double* kff; const double* Kfb;
  double* Xs;            // [NA][(T+1)*nx][Np] trial trajectories
  double* Us;            // [NA][T*nu][Np]
  double* trial_merit;   // [NA][Np]
  double* trial_obj;     // [NA][Np] objective of each trial (== merit without constraint functions; may be null then)
  const int* active;
  ALView al;
  Publish publish{};     // fused rounds (mas_fused_hip.hpp): the first lane publishes the counters of the select that ran in the kernel before
  // staged line search (M::kStagedLineSearch): this launch rolls out the step sizes j_base, j_base + 1, ...; with stage_first set, only for agents
  // whose FIRST trial (stage_first[a], written by the launch before) is no strict decrease of their merit stage_cur[a] -- SelectBody's comparison
  int j_base = 0; const double* stage_first = nullptr; const double* stage_cur = nullptr;
  MAS_HD void operator()(int a, int j) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (a == 0 && j == 0 && publish.host) publish();
    j += j_base;
#if defined(__HIP_DEVICE_COMPILE__)
    // A wavefront with at least one active agent rolls out ALL its 64 agents: a retired agent's lane costs nothing
    // extra (SIMD), its results go to its own unused trial slots and trial_merit entry (SelectBody ignores inactive
    // agents) -- but every store of the wavefront then covers whole 128-byte lines.  With the retired lanes masked
    // off the same stores are partial-line writes whose acknowledgements take so long that the in-order memory
    // counter throttles the wavefront: measured 373 us instead of 154 us per launch once few agents are active.
    if (__ballot(active[a] != 0 && !(stage_first != nullptr && stage_first[a] < stage_cur[a])) == 0ull) return;
#else
    if (!active[a]) return;
    if (stage_first != nullptr && stage_first[a] < stage_cur[a]) return;
#endif
    double alpha = 1.0;
    for (int q = 0; q < j; ++q) alpha *= 0.5;
    double* Xj = Xs + static_cast<size_t>(j) * (d.T + 1) * NX * d.Np;
    double* Uj = Us + static_cast<size_t>(j) * d.T * NU * d.Np;
    double x[NX], xn[NX], u[NU];
    // addressing: per step one uniform base pointer per array (advanced with scalar adds) plus loop-invariant
    // 32-bit byte offsets (row_in_step * Np + agent) * 8 per lane
    constexpr int NR = NU * NX;                                      // the widest per-step block (gain matrix rows)
    unsigned off[NR];
    MAS_UNROLL for (int i = 0; i < NR; ++i) off[i] = static_cast<unsigned>((static_cast<size_t>(i) * d.Np + static_cast<size_t>(a)) * sizeof(double));
    const size_t sX = static_cast<size_t>(NX) * d.Np, sU = static_cast<size_t>(NU) * d.Np, sK = static_cast<size_t>(NU) * NX * d.Np;
    const double *Xt = X, *Ut = U, *kt = kff, *Kt = Kfb;
    double *Xjt = Xj, *Ujt = Uj;
    // reference point and gains of step t+1 are requested before step t is integrated (software pipeline); the two
    // operand sets swap roles from one step to the next (loop unrolled by two), so nothing is copied between them
    struct Operands { double xr[NX], ur[NU], k[NU], K[NU * NX]; };
    Operands opA, opB;
    auto request = [&](Operands& o) {
      MAS_UNROLL for (int i = 0; i < NX; ++i) o.xr[i] = ld_off(Xt, off[i]);
      MAS_UNROLL for (int i = 0; i < NU; ++i) o.ur[i] = ld_off(Ut, off[i]);
      MAS_UNROLL for (int i = 0; i < NU; ++i) o.k[i] = ld_off(kt, off[i]);
      MAS_UNROLL for (int i = 0; i < NU * NX; ++i) o.K[i] = ld_off(Kt, off[i]);
    };
    MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = ld_off(x0, off[i]); st_off(Xjt, off[i], x[i]); }
    request(opA);
    double c = 0.0;
    double lbv[NU], ubv[NU];                                         // the bounds, kept in vector registers for the selects
    // models that ship with input bounds clamp unconditionally -- without bounds against -inf / +inf, never the smaller / larger
    // operand, so every control passes through bit for bit: no branch in the step, which the compiler then schedules as one block
    // (C3 forward 96 -> 94 us).  A model without default bounds (the identity LQR) keeps the branch: eight selects per step that
    // are never needed cost it more (42 -> 57 us) than the branch does.
    MAS_UNROLL for (int i = 0; i < NU; ++i) { lbv[i] = clamp ? lb[i] : -INFINITY; ubv[i] = clamp ? ub[i] : INFINITY; MAS_VGPR64(lbv[i]); MAS_VGPR64(ubv[i]); }
    auto step = [&](int t, const Operands& cur, Operands& next) {
      if (t + 1 < d.T) { Xt += sX; Ut += sU; kt += sU; Kt += sK; }   // uniform
      MAS_UNROLL for (int i = 0; i < NR; ++i) MAS_OPAQUE32(off[i]);
      request(next);
      forward_control<M>(x, cur.xr, cur.ur, cur.k, cur.K, alpha, M::kDefaultBounds ? true : clamp, lbv, ubv, u);
      MAS_UNROLL for (int i = 0; i < NU; ++i) { MAS_OPAQUE32(off[i]); st_off(Ujt, off[i], u[i]); }
      c += M::stage_cost(x, u, t, P);
      rk4_step<M>(x, u, dt, xn, P);
      Xjt += sX; Ujt += sU;
      MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = xn[i]; MAS_OPAQUE32(off[i]); st_off(Xjt, off[i], x[i]); }
    };
    int t = 0;
    for (; t + 1 < d.T; t += 2) { step(t, opA, opB); step(t + 1, opB, opA); }
    if (t < d.T) step(t, opA, opB);
    c += M::terminal_cost(x, P);
    double m = c;
    if constexpr (M::NEQ + M::NIN > 0) {
      if (al.on) {                                     // compute_merit(x_trial, u_trial), ilqr.hpp:219,379-407
        const double rho = al.rho[a];
        for (int t = 0; t < d.T; ++t) {
          double xt[NX], ut[NU], lam[M::NEQ > 0 ? M::NEQ : 1], mu[M::NIN > 0 ? M::NIN : 1];
          MAS_UNROLL for (int i = 0; i < NX; ++i) xt[i] = MAS_AT(Xj, t * NX + i, a);
          MAS_UNROLL for (int i = 0; i < NU; ++i) ut[i] = MAS_AT(Uj, t * NU + i, a);
          MAS_UNROLL for (int k2 = 0; k2 < M::NEQ; ++k2) lam[k2] = MAS_AT(al.lam, t * M::NEQ + k2, a);
          MAS_UNROLL for (int k2 = 0; k2 < M::NIN; ++k2) mu[k2] = MAS_AT(al.mu, t * M::NIN + k2, a);
          m = al_merit_step<M>(m, xt, ut, lam, mu, rho, al.act_tol, P);
        }
      }
    }
    trial_merit[static_cast<size_t>(j) * d.Np + a] = m;
    if (trial_obj) trial_obj[static_cast<size_t>(j) * d.Np + a] = c;
  }
};

// -------------------------------------------------------------------------------------------
// Per-agent acceptance + termination bookkeeping (ilqr.hpp:219-234,269-271).
struct SelectBody {
  Dims d; double tolerance; int max_iterations;
  const double* trial_merit;  // [NA][Np]
  double* cost; double* merit;
  int* active; int* alpha_sel; int* iters_solve; int* iters_total; int* exit_code;
  int* n_active;              // counter of agents still active after this iteration
  long long* n_converged;
  long long* clear;           // counter pair of the ring that the round after next will use: reset here
  const double* trial_obj = nullptr;   // objective of the trials when it differs from the merit (constraint functions)
  double* improvement_out = nullptr;   // non-null: termination is decided by ALUpdateBody after the multiplier update
  int* next_list = nullptr;            // non-null: the agents still active, compacted (slot = value of n_active when it was counted)
  int* solves_left = nullptr;          // non-null: an agent whose solve ends starts its next one at once (chained solves, see begin_chain)
  // per-solve wall-clock budget, ilqr.hpp:61,84-90: non-null = max_ms is finite; solve_start[a] is the tick at which agent
  // a's current solve began.  The reference checks `elapsed_ms > max_ms` (elapsed truncated to whole ms) at the top of every
  // iteration; here the check for iteration i+1 is made at the end of iteration i (select), a few microseconds earlier.
  unsigned long long* solve_start = nullptr; double max_ms = 0.0;
  int* dbg_iter = nullptr; double* dbg_improvement = nullptr;   // "debug" trace: iteration number (0 = idle) and d_merit per agent
  int* ended = nullptr;                // non-null: 1 for every agent whose solve ended in this iteration, else 0 (chained trust-region epilogue)
  int* slot_out = nullptr;             // non-null (with next_list): slot_out[a] = the agent's position in next_list (fused rounds: where its trial column lies)
  // non-null: the agents whose solve ended in this iteration, compacted (any order) -- one workgroup each in mas_trust_epilogue_kernel.
  // Two counters alternate by round: this one is counted up, ended_clear (the other one, read by the epilogue of the round before) reset.
  int* ended_list = nullptr; int* ended_count = nullptr; int* ended_clear = nullptr;
  MAS_HD bool out_of_time(int a) const {
    if (!solve_start) return false;
    const double elapsed_ms = static_cast<double>(static_cast<long long>(static_cast<double>(MAS_NOW_TICKS() - solve_start[a]) / kTicksPerMs));
    return elapsed_ms > max_ms;
  }
  // what became of agent a in this iteration; bit 2 = its solve ended but it goes straight on with its next solve
  enum Outcome { kIdle = 0, kStillActive = 1, kConverged = 2, kRetired = 3, kChained = 4 };
  // detail::run_sequential (nash.hpp:74-90) calls solve() on every agent max_outer times, and an agent's solve k+1
  // depends on nothing but its own solve k: the new solver state is iteration count 0, and trajectory, cost and
  // merit are what solve k left (its rollout of best_controls, bit for bit).  So the agent does not wait for the
  // others at a host-side barrier; it simply stays active.
  MAS_HD bool begin_chain(int a) const {
    if (!solves_left || solves_left[a] <= 0) return false;
    solves_left[a] -= 1; iters_solve[a] = 0;
    if (solve_start) solve_start[a] = MAS_NOW_TICKS();      // const auto start = clock::now(), ilqr.hpp:61
    return true;
  }
  MAS_HD int decide(int a) const {
    const int r = decide_(a);
    if (ended) ended[a] = (r == kConverged || r == kRetired || (r & kChained) != 0) ? 1 : 0;
    return r;
  }
  MAS_HD int decide_(int a) const {
    if (a == 0 && clear) { clear[0] = 0; clear[1] = 0; }
    if (a == 0 && ended_clear) *ended_clear = 0;
    if (!active[a]) { alpha_sel[a] = -1; if (dbg_iter) dbg_iter[a] = 0; return kIdle; }
    const double cur = merit[a];
    int sel = -1; double best = cur;
    double tm[kNumAlphas];                                  // all ten loads in flight at once: a load per loop trip would put up to ten
    MAS_UNROLL for (int j = 0; j < kNumAlphas; ++j) tm[j] = trial_merit[static_cast<size_t>(j) * d.Np + a];   // memory round trips on this kernel's critical path
    MAS_UNROLL for (int j = kNumAlphas - 1; j >= 0; --j) { if (tm[j] < cur) { best = tm[j]; sel = j; } }      // FIRST strict decrease (ilqr.hpp:220-226) = the lowest such j
    const double improvement = cur - best;                  // 0 when no trial was accepted
    alpha_sel[a] = sel;
    if (sel >= 0) { cost[a] = trial_obj ? trial_obj[static_cast<size_t>(sel) * d.Np + a] : best; merit[a] = best; }   // cost = objective(best_x,best_u); current_merit = best_merit
    const int it = iters_solve[a] + 1;
    iters_solve[a] = it; iters_total[a] += 1;
    if (dbg_iter) { dbg_iter[a] = it; dbg_improvement[a] = improvement; }
    if (improvement_out) { improvement_out[a] = improvement; return kIdle; }
    if (improvement < tolerance) {                          // ilqr.hpp:269-271
      exit_code[a] = kExitConverged | (std::isfinite(best) ? 0 : kExitNonFinite);
      if (begin_chain(a)) return kConverged | kChained;
      active[a] = 0;
      return kConverged;
    }
    if (it >= max_iterations) {                             // ilqr.hpp:82
      exit_code[a] = kExitMaxIter | (std::isfinite(best) ? 0 : kExitNonFinite);
      if (begin_chain(a)) return kRetired | kChained;
      active[a] = 0;
      return kRetired;
    }
    if (out_of_time(a)) {                                   // ilqr.hpp:84-90, top of the next iteration
      exit_code[a] = kExitTime | (std::isfinite(best) ? 0 : kExitNonFinite);
      if (begin_chain(a)) return kRetired | kChained;
      active[a] = 0;
      return kRetired;
    }
    return kStillActive;
  }
  MAS_HD void operator()(int a) const { count(decide(a), a); }
  // counters + compaction for outcome r of agent a (a whole wavefront calls this; lanes without an agent pass kIdle)
  MAS_HD void count(int r, int a) const {
#if defined(__HIP_DEVICE_COMPILE__)
    // one atomic per wavefront and counter; the returned base gives every still-active agent its slot in the
    // compacted list that the next iteration's sweep walks
    const bool goes_on = r == kStillActive || (r & kChained) != 0;
    const unsigned long long still = __ballot(goes_on), conv = __ballot((r & ~kChained) == kConverged);
    const int lane = static_cast<int>(threadIdx.x) & 63;
    if (conv != 0ull && lane == __builtin_ctzll(conv)) MAS_ATOMIC_ADD_I64(n_converged, static_cast<long long>(__popcll(conv)));
    if (still != 0ull) {
      const int leader = __builtin_ctzll(still);
      int base = 0;
      if (lane == leader) base = atomicAdd(n_active, __popcll(still));
      base = __shfl(base, leader);
      if (next_list && goes_on) {
        const int pos = base + __popcll(still & ((1ull << lane) - 1ull));
        next_list[pos] = a;
        if (slot_out) slot_out[a] = pos;
      }
    }
    if (ended_list) {
      const unsigned long long done = __ballot(r == kConverged || r == kRetired || (r & kChained) != 0);
      if (done != 0ull) {
        const int leader = __builtin_ctzll(done);
        int base = 0;
        if (lane == leader) base = atomicAdd(ended_count, __popcll(done));
        base = __shfl(base, leader);
        if ((done >> lane) & 1ull) ended_list[base + __popcll(done & ((1ull << lane) - 1ull))] = a;
      }
    }
#else
    if ((r & ~kChained) == kConverged) MAS_ATOMIC_ADD_I64(n_converged, 1LL);
    if (r == kStillActive || (r & kChained) != 0) { if (next_list) { next_list[*n_active] = a; if (slot_out) slot_out[a] = *n_active; } MAS_ATOMIC_ADD_INT(n_active, 1); }
    if (ended_list && (r == kConverged || r == kRetired || (r & kChained) != 0)) ended_list[(*ended_count)++] = a;
#endif
  }
};

// copy the accepted trial into the current trajectory (x = best_x; u = best_u, ilqr.hpp:231-232); Publish: see above ForwardBody
template <class M>
struct CommitBody {
  Dims d;
  const double* Xs; const double* Us; const int* alpha_sel;
  double* X; double* U;
  Publish publish{};
  MAS_HD void operator()(int a, int t) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (a == 0 && t == 0 && publish.host) publish();
    const int j = alpha_sel[a];
    if (j < 0) return;
    const double* Xj = Xs + static_cast<size_t>(j) * (d.T + 1) * NX * d.Np;
    const double* Uj = Us + static_cast<size_t>(j) * d.T * NU * d.Np;
    MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(X, t * NX + i, a) = MAS_AT(Xj, t * NX + i, a);
    if (t < d.T) { MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(U, t * NU + i, a) = MAS_AT(Uj, t * NU + i, a); }
  }
};

// the same copy after a four-lanes-per-rollout line search (mas_forward_quad_kernel): its trial planes are indexed by the
// SLOT of the agent in the round's list of live agents (null list: slot == agent)
template <class M>
struct CommitSlotBody {
  Dims d;
  const double* Xs; const double* Us; const int* alpha_sel;
  double* X; double* U;
  const int* live_list; const int* live_count;
  Publish publish{};
  MAS_HD void operator()(int slot, int t) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (slot == 0 && t == 0 && publish.host) publish();
    if (live_count && slot >= *live_count) return;
    const int a = live_list ? live_list[slot] : slot;
    const int j = alpha_sel[a];
    if (j < 0) return;
    const double* Xj = Xs + static_cast<size_t>(j) * (d.T + 1) * NX * d.Np;
    const double* Uj = Us + static_cast<size_t>(j) * d.T * NU * d.Np;
    MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(X, t * NX + i, a) = MAS_AT(Xj, t * NX + i, slot);
    if (t < d.T) { MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(U, t * NU + i, a) = MAS_AT(Uj, t * NU + i, slot); }
  }
};

// multiplier / penalty update and termination for constrained models, ilqr.hpp:236-260,269-271 (after commit)
template <class M>
struct ALUpdateBody {
  Dims d; ModelParams P; double tolerance; int max_iterations; ALParams ap;
  const double* X; const double* U; const double* improvement;
  ALView al;
  int* active; const int* iters_solve; int* exit_code; int* n_active; long long* n_converged;
  const unsigned long long* solve_start = nullptr; double max_ms = 0.0;     // finite per-solve budget (see SelectBody)
  MAS_HD void operator()(int a) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (!active[a]) return;
    const double rho = al.rho[a];
    double eqv = 0.0, inv = 0.0;
    for (int t = 0; t < d.T; ++t) {
      double xt[NX], ut[NU], lam[M::NEQ > 0 ? M::NEQ : 1], mu[M::NIN > 0 ? M::NIN : 1];
      MAS_UNROLL for (int i = 0; i < NX; ++i) xt[i] = MAS_AT(X, t * NX + i, a);
      MAS_UNROLL for (int i = 0; i < NU; ++i) ut[i] = MAS_AT(U, t * NU + i, a);
      MAS_UNROLL for (int k = 0; k < M::NEQ; ++k) lam[k] = MAS_AT(al.lam, t * M::NEQ + k, a);
      MAS_UNROLL for (int k = 0; k < M::NIN; ++k) mu[k] = MAS_AT(al.mu, t * M::NIN + k, a);
      al_update_step<M>(xt, ut, lam, mu, rho, eqv, inv, P);
      MAS_UNROLL for (int k = 0; k < M::NEQ; ++k) MAS_AT(al.lam, t * M::NEQ + k, a) = lam[k];
      MAS_UNROLL for (int k = 0; k < M::NIN; ++k) MAS_AT(al.mu, t * M::NIN + k, a) = mu[k];
    }
    const double eq_norm = sqrt(eqv), in_norm = sqrt(inv);
    if (eq_norm > ap.constraint_tolerance || in_norm > ap.constraint_tolerance) al.rho[a] = rho * ap.penalty_increase;
    const double imp = improvement[a];
    if (imp < tolerance && eq_norm < ap.constraint_tolerance && in_norm < ap.constraint_tolerance) {
      active[a] = 0; exit_code[a] = kExitConverged;
      MAS_ATOMIC_ADD_I64(n_converged, 1LL);
    } else if (iters_solve[a] >= max_iterations) {
      active[a] = 0; exit_code[a] = kExitMaxIter;
    } else if (solve_start && static_cast<double>(static_cast<long long>(static_cast<double>(MAS_NOW_TICKS() - solve_start[a]) / kTicksPerMs)) > max_ms) {
      active[a] = 0; exit_code[a] = kExitTime;
    } else {
      MAS_ATOMIC_ADD_INT(n_active, 1);
    }
  }
};

// start of a solve: every agent active, per-solve iteration counter cleared (resize_buffers
// zeroes the gains, ilqr.hpp:292-293; they are rewritten before use so nothing to clear here)
struct BeginSolveBody {
  int* active; int* iters_solve; int* exit_code; int* alpha_sel;
  const double* cost = nullptr; double* merit = nullptr;      // non-null: current_merit = objective (ilqr.hpp:78) when the rollout is already on the device
  long long* counters = nullptr; int n_counters = 0;           // the round counters of the solve, zeroed by agent 0
  int* live_list = nullptr; int* live_count = nullptr; int n_agents = 0;   // every agent is active: identity list
  int* solves_left = nullptr; int chained = 0;                 // further solves every agent runs back to back after this one
  unsigned long long* solve_start = nullptr;                   // non-null: finite max_ms, stamp the start of the solve (ilqr.hpp:61)
  int* slot_of = nullptr;                                      // non-null: position of every agent in the identity list (fused rounds)
  int* ended_counts = nullptr;                                 // non-null: the two alternating counters of SelectBody::ended_list, zeroed by agent 0
  MAS_HD void operator()(int a) const {
    active[a] = 1; iters_solve[a] = 0; exit_code[a] = kExitMaxIter; alpha_sel[a] = -1;
    if (slot_of) slot_of[a] = a;
    if (solve_start) solve_start[a] = MAS_NOW_TICKS();
    if (solves_left) solves_left[a] = chained;
    if (merit) merit[a] = cost[a];
    if (live_list) live_list[a] = a;
    if (a == 0) { for (int i = 0; i < n_counters; ++i) counters[i] = 0; if (live_count) *live_count = n_agents; if (ended_counts) { ended_counts[0] = 0; ended_counts[1] = 0; } }
  }
};

struct MarkTimeoutBody {
  int* active; int* exit_code;
  MAS_HD void operator()(int a) const { if (active[a]) { active[a] = 0; exit_code[a] = kExitTime; } }
};

// -------------------------------------------------------------------------------------------
// Trust-region epilogue, nash.hpp:214-243, one lane per agent.
template <class M>
struct TrustRegionBody {
  Dims d; double dt; ModelParams P;
  const double* x0;
  double* X; double* U; double* cost;               // candidate in, accepted/reverted out
  const double* X_old; const double* U_old; const double* cost_old;
  double* radii;
  MAS_HD void operator()(int a) const {
    constexpr int NX = M::NX, NU = M::NU;
    const double radius = radii[a];
    double ss = 0.0;                                   // delta.norm(): Frobenius, column-major order
    for (int r = 0; r < d.T * NU; ++r) { const double dl = MAS_AT(U, r, a) - MAS_AT(U_old, r, a); ss += dl * dl; }
    const double norm = sqrt(ss);
    double cand_cost = cost[a];
    if (norm > radius) {                               // scale the step back onto the trust region, re-roll, re-cost
      const double scale = radius / norm;
      double x[NX], xn[NX], u[NU];
      MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = MAS_AT(x0, i, a); MAS_AT(X, i, a) = x[i]; }
      double c = 0.0;
      for (int t = 0; t < d.T; ++t) {
        MAS_UNROLL for (int i = 0; i < NU; ++i) {
          const double uo = MAS_AT(U_old, t * NU + i, a);
          u[i] = uo + scale * (MAS_AT(U, t * NU + i, a) - uo);
          MAS_AT(U, t * NU + i, a) = u[i];
        }
        c += M::stage_cost(x, u, t, P);
        rk4_step<M>(x, u, dt, xn, P);
        MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = xn[i]; MAS_AT(X, (t + 1) * NX + i, a) = x[i]; }
      }
      c += M::terminal_cost(x, P);
      cand_cost = c;
    }
    if (cand_cost < cost_old[a]) {
      cost[a] = cand_cost; radii[a] = radius * 1.5;
    } else {
      for (int r = 0; r < d.T * NU; ++r) MAS_AT(U, r, a) = MAS_AT(U_old, r, a);
      for (int r = 0; r < (d.T + 1) * NX; ++r) MAS_AT(X, r, a) = MAS_AT(X_old, r, a);
      cost[a] = cost_old[a]; radii[a] = radius * 0.5;
    }
  }
};

// -------------------------------------------------------------------------------------------
// Trust-region epilogue for CHAINED solves (run_trust_region without a host barrier between outer iterations).
// detail::run_trust_region (nash.hpp:182-248) treats every agent on its own: solve, measure the control step against
// the agent's radius, scale it back and re-roll if it is too long, accept (radius x1.5) or revert (radius x0.5).  An
// agent's outer iteration k+1 therefore depends on nothing but its own iteration k, so an agent whose solve ends in
// some round runs its epilogue right there and goes on with its next solve in the following round.  Three kernels per
// round, all of them no-ops for agents whose solve did not end in it (SelectBody::ended):
//   TrustNormBody    lane = agent: step norm (serial sum in the reference's column-major order); a step inside the
//                    radius is accepted / reverted at once, a longer one is queued for the re-roll
//   TrustRerollBody  lane = queued agent: scaled step, rollout, cost, then accept / revert
//   TrustCopyBody    lane = (agent, t): revert = restore the old trajectory, accept = make the new one the old one
enum TrustVerdict : int { kTrustNone = 0, kTrustAccept = 1, kTrustRevert = 2, kTrustReroll = 3 };

struct TrustState {
  double* cost; double* merit; double* cost_old; double* radii;
  int* verdict;                     // [Np] TrustVerdict of this round
  // accept / revert, nash.hpp:228-243; the next solve's current_merit is the objective of the trajectory it starts from
  MAS_HD void finish(int a, double cand_cost, double radius) const {
    if (cand_cost < cost_old[a]) { cost[a] = cand_cost; merit[a] = cand_cost; cost_old[a] = cand_cost; radii[a] = radius * 1.5; verdict[a] = kTrustAccept; }
    else { cost[a] = cost_old[a]; merit[a] = cost_old[a]; radii[a] = radius * 0.5; verdict[a] = kTrustRevert; }
  }
};

template <class M>
struct TrustNormBody {
  Dims d; const int* ended; const double* U; const double* U_old; TrustState ts;
  double* scale_out;                // [Np] radius / norm for the queued agents
  int* roll_list; int* roll_count;  // agents whose step leaves the trust region, compacted
  MAS_HD void operator()(int a) const {
    constexpr int NU = M::NU;
    if (!ended[a]) { ts.verdict[a] = kTrustNone; return; }
    const double radius = ts.radii[a];
    double ss = 0.0;                                   // delta.norm(): Frobenius, column-major order (nash.hpp:216-217)
    const int R = d.T * NU;
    int r = 0;
    for (; r + 16 <= R; r += 16) {                     // 32 loads in flight (a lane walking the rows alone pays a memory round trip per batch), the additions stay in order
      double dl[16];
      MAS_UNROLL for (int q = 0; q < 16; ++q) dl[q] = MAS_AT(U, r + q, a) - MAS_AT(U_old, r + q, a);
      MAS_UNROLL for (int q = 0; q < 16; ++q) ss += dl[q] * dl[q];
    }
    for (; r < R; ++r) { const double dl = MAS_AT(U, r, a) - MAS_AT(U_old, r, a); ss += dl * dl; }
    const double norm = sqrt(ss);
    if (norm > radius) {
      scale_out[a] = radius / norm;
      ts.verdict[a] = kTrustReroll;
#if defined(__HIP_DEVICE_COMPILE__)
      roll_list[atomicAdd(roll_count, 1)] = a;
#else
      roll_list[(*roll_count)++] = a;
#endif
    } else {
      ts.finish(a, ts.cost[a], radius);
    }
  }
};

template <class M>
struct TrustRerollBody {
  Dims d; double dt; ModelParams P;
  const double* x0; double* X; double* U; const double* U_old; const double* scale; TrustState ts;
  const int* roll_list; const int* roll_count;
  MAS_HD void operator()(int slot) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (slot >= *roll_count) return;
    const int a = roll_list[slot];
    const double sc = scale[a];
    double x[NX], xn[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = MAS_AT(x0, i, a); MAS_AT(X, i, a) = x[i]; }
    double c = 0.0;
    for (int t = 0; t < d.T; ++t) {                    // cand_u = old_u + scale * delta; cand_x = rollout; cand_cost = objective (nash.hpp:219-226)
      MAS_UNROLL for (int i = 0; i < NU; ++i) {
        const double uo = MAS_AT(U_old, t * NU + i, a);
        u[i] = uo + sc * (MAS_AT(U, t * NU + i, a) - uo);
        MAS_AT(U, t * NU + i, a) = u[i];
      }
      c += M::stage_cost(x, u, t, P);
      rk4_step<M>(x, u, dt, xn, P);
      MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = xn[i]; MAS_AT(X, (t + 1) * NX + i, a) = x[i]; }
    }
    c += M::terminal_cost(x, P);
    ts.finish(a, c, ts.radii[a]);
  }
};

template <class M>
struct TrustCopyBody {
  Dims d; const int* verdict;
  double* X; double* U; double* X_old; double* U_old;
  int* roll_count;                  // reset here (last kernel of the round's epilogue) for the next round's TrustNormBody
  MAS_HD void operator()(int a, int t) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (a == 0 && t == 0) *roll_count = 0;
    const int v = verdict[a];
    if (v == kTrustAccept) {                           // the accepted trajectory is what the next outer iteration measures its step against
      MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(X_old, t * NX + i, a) = MAS_AT(X, t * NX + i, a);
      if (t < d.T) { MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(U_old, t * NU + i, a) = MAS_AT(U, t * NU + i, a); }
    } else if (v == kTrustRevert) {
      MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(X, t * NX + i, a) = MAS_AT(X_old, t * NX + i, a);
      if (t < d.T) { MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(U, t * NU + i, a) = MAS_AT(U_old, t * NU + i, a); }
    }
  }
};

// Joint step of detail::run_line_search (nash.hpp:92-180) decided ON THE DEVICE.  The reference compares sums of ALL agents' costs
// (compute_total_cost, nash.hpp:45-49): base_cost before, new_cost after the solves, trial_cost after every blended rollout.  The
// sums are formed by one wavefront in block order (wave_serial_sum, mas_central.hpp: the serial loop's rounding) and the decisions
// are kept in JointState, so the alpha ladder needs no read-back: every blend / sum pair is enqueued, and the pairs behind the
// accepted one find `accepted` set and do nothing.
struct JointState {
  double base;         // base_cost
  double last_total;   // the sum formed last
  int need_blend;      // new_cost >= base_cost: the alpha ladder runs (nash.hpp:121-123)
  int accepted;        // a blended step with trial_cost < base_cost was found (nash.hpp:144-158)
};

// Nash line-search blend, nash.hpp:134-142: U_trial = U_old + alpha (U_cand - U_old), roll out, cost
template <class M>
struct BlendRolloutBody {
  Dims d; double dt; double alpha; ModelParams P;
  const double* x0; const double* U_old; const double* U_cand;
  double* X_trial; double* U_trial; double* cost_trial;
  const JointState* st = nullptr;     // non-null: skip unless the ladder is running and nothing was accepted yet
  MAS_HD void operator()(int a) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (st && (!st->need_blend || st->accepted)) return;
    double x[NX], xn[NX], u[NU];
    MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = MAS_AT(x0, i, a); MAS_AT(X_trial, i, a) = x[i]; }
    double c = 0.0;
    for (int t = 0; t < d.T; ++t) {
      MAS_UNROLL for (int i = 0; i < NU; ++i) {
        const double uo = MAS_AT(U_old, t * NU + i, a);
        u[i] = uo + alpha * (MAS_AT(U_cand, t * NU + i, a) - uo);
        MAS_AT(U_trial, t * NU + i, a) = u[i];
      }
      c += M::stage_cost(x, u, t, P);
      rk4_step<M>(x, u, dt, xn, P);
      MAS_UNROLL for (int i = 0; i < NX; ++i) { x[i] = xn[i]; MAS_AT(X_trial, (t + 1) * NX + i, a) = x[i]; }
    }
    c += M::terminal_cost(x, P);
    cost_trial[a] = c;
  }
};

// no blended step was accepted: back to the trajectories of before the solves (nash.hpp:160-176)
template <class M>
struct RevertBody {
  Dims d; const JointState* st;
  double* X; double* U; double* cost; const double* X_old; const double* U_old; const double* cost_old;
  MAS_HD void operator()(int a, int t) const {
    constexpr int NX = M::NX, NU = M::NU;
    if (!st->need_blend || st->accepted) return;
    MAS_UNROLL for (int i = 0; i < NX; ++i) MAS_AT(X, t * NX + i, a) = MAS_AT(X_old, t * NX + i, a);
    if (t < d.T) { MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(U, t * NU + i, a) = MAS_AT(U_old, t * NU + i, a); }
    if (t == 0) cost[a] = cost_old[a];              // objective(old) == the stored old cost
  }
};

// -------------------------------------------------------------------------------------------
// layout conversion between the reference layout at the ABI ([N][cols][rows], one Eigen column
// per time step) and the SoA device layout ([cols*rows][Np]); y = column (time step)
struct AosToSoaBody {
  Dims d; int rows;
  const double* aos; double* soa; int cols;
  MAS_HD void operator()(int a, int c) const {
    for (int r = 0; r < rows; ++r) MAS_AT(soa, c * rows + r, a) = aos[(static_cast<size_t>(a) * cols + c) * rows + r];
  }
};
// derivative tiles [T][Np][ND] -> agent-fastest rows [T*ND][Np] (centralised backward pass: one lane per agent)
struct TilesToSoaBody {
  Dims d; int nd;
  const double* tiles; double* soa;
  MAS_HD void operator()(int a, int t) const {
    for (int e = 0; e < nd; ++e) MAS_AT(soa, t * nd + e, a) = MAS_TILE_AT(tiles, nd, t, a, e);
  }
};
// derivative tiles [T][Np][ND] -> host layout [N][T][ND] (parity tests only)
struct TilesToHostBody {
  Dims d; int nd;
  const double* tiles; double* out;
  MAS_HD void operator()(int a, int t) const {
    for (int e = 0; e < nd; ++e) out[(static_cast<size_t>(a) * d.T + t) * nd + e] = MAS_TILE_AT(tiles, nd, t, a, e);
  }
};
struct SoaToAosBody {
  Dims d; int rows;
  const double* soa; double* aos; int cols;
  MAS_HD void operator()(int a, int c) const {
    for (int r = 0; r < rows; ++r) aos[(static_cast<size_t>(a) * cols + c) * rows + r] = MAS_AT(soa, c * rows + r, a);
  }
};

}  // namespace mas
