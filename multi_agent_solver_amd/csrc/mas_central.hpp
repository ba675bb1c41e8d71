// mas_central.hpp -- the centralised strategy as block-diagonal batched work.
//
// CentralizedStrategy (strategies/centralized.hpp:18-38) stacks all agents into one OCP
// (multi_agent_problem.hpp:52-127) and runs ONE iLQR on it.  The stacked dynamics Jacobians
// are exactly block-diagonal and the stacked cost Hessians are block-diagonal up to finite-
// difference noise, so the solve is carried out per agent block -- but three things stay
// shared by all agents, as they are in the stacked solve:
//   * the step length alpha and its acceptance (merit = sum of all agents' costs),
//   * termination (improvement of that merit),
//   * the regularisation ladder: LLT of the block-diagonal q_uu fails if ANY block fails, and
//     reg*I is then added to EVERY block (ilqr.hpp:172-182).
// The ladder needs an all-agents agreement per time step.  On the GPU the backward pass is the 16-lane tile kernel
// (mas_kernels_hip.hpp, SharedLadder) with a counter-based agreement across its <= 256 resident wavefronts
// (LadderSync); CentralBackwardBody below -- ONE workgroup, one lane per agent, a barrier-reduction per rung -- is
// the statement the host emulation replays and the path of the constrained test models.
#pragma once

#include "mas_bodies.hpp"

namespace mas {

constexpr int kCentralMaxAgents = 1024;

struct CentralState {
  double cur_total;     // current merit of the stacked problem
  double improvement;
  int j_sel;            // accepted alpha index of the last iteration, -1 if none
  int done;             // improvement < tolerance
  int iterations;
  int pad;
};

#if !defined(MAS_HOST_EMULATION)
// sum of v[0..n-1] in index order from 0.0 -- the serial loop's rounding -- by ONE wavefront: 64 values per coalesced load,
// then added up lane by lane through v_readlane_b32 pairs (a thread walking the array alone pays a full memory round trip per value;
// __shfl with a constant lane compiles to ds_bpermute_b32 and a wait per element -- ~160 cycles per addition against ~20: the
// centralised select took 17 us for ten sums of 256 costs).
// Lanes beyond n contribute -0.0, which changes no bit of the running sum.  Every lane returns the sum.
__device__ inline double wave_serial_sum(const double* v, int n) {
  const int lane = static_cast<int>(threadIdx.x) & 63;
  double s = 0.0;
  for (int base = 0; base < n; base += 256) {                          // four loads in flight, then their 256 additions in order
    double x[4];
    MAS_UNROLL for (int q = 0; q < 4; ++q) { const int a = base + 64 * q + lane; x[q] = a < n ? v[a] : -0.0; }
    MAS_UNROLL for (int q = 0; q < 4; ++q) {
      if (base + 64 * q < n) {                                           // uniform
        const int lo32 = __double2loint(x[q]), hi32 = __double2hiint(x[q]);
        MAS_UNROLL for (int i = 0; i < 64; ++i) s += __hiloint2double(__builtin_amdgcn_readlane(hi32, i), __builtin_amdgcn_readlane(lo32, i));
      }
    }
  }
  return s;
}
#endif

// total cost of all agents + the decision of one stage of the Nash line search (JointState, mas_bodies.hpp); one wavefront
struct JointSumBody {
  enum Mode : int { kBase = 0, kAfterSolves = 1, kAfterTrial = 2, kFinal = 3 /* the Solution's total cost: the sum only, no decision */ };
  const double* costs; int n; JointState* st; int mode;
  MAS_HD void decide(double s) const {
    st->last_total = s;
    if (mode == kBase) { st->base = s; st->need_blend = 0; st->accepted = 0; }
    else if (mode == kAfterSolves) { st->accepted = 0; if (s >= st->base) st->need_blend = 1; else { st->base = s; st->need_blend = 0; } }   // nash.hpp:119-123,178
    else if (mode == kAfterTrial && st->need_blend && !st->accepted && s < st->base) { st->base = s; st->accepted = 1; }                       // nash.hpp:144-158
  }
#if defined(MAS_HOST_EMULATION)
  void run_serial(int) const {
    if (mode == kAfterTrial && (!st->need_blend || st->accepted)) return;
    double s = 0.0; for (int a = 0; a < n; ++a) s += costs[a];
    decide(s);
  }
#else
  __device__ void run_block(int) const {
    if (threadIdx.x < 64) {
      if (mode == kAfterTrial && (!st->need_blend || st->accepted)) return;      // uniform: the costs have not changed since the accepted trial
      const double s = wave_serial_sum(costs, n);
      if (threadIdx.x == 0) decide(s);
    }
  }
#endif
};

// serial sum in agent order (the stacked objective adds the blocks in block order)
struct CentralInitBody {
  Dims d; const double* cost; CentralState* st;
#if defined(MAS_HOST_EMULATION)
  void run_serial(int n) const {
    double s = 0.0; for (int a = 0; a < n; ++a) s += cost[a];
    st->cur_total = s; st->improvement = 0.0; st->j_sel = -1; st->done = 0; st->iterations = 0;
  }
#else
  __device__ void run_block(int n) const {
    if (threadIdx.x < 64) {
      const double s = wave_serial_sum(cost, n);
      if (threadIdx.x == 0) { st->cur_total = s; st->improvement = 0.0; st->j_sel = -1; st->done = 0; st->iterations = 0; }
    }
  }
#endif
};

template <class M>
struct CentralBackwardBody {
  Dims d; ModelParams P;
  const double* X; const double* D;
  double* kff; double* Kfb; long long* reg_events;
  const double* Dsoa;      // [T*ND][Np] agent-fastest copy of the tiles (GPU path); the emulation reads the tiles
  const int* active = nullptr;   // non-null: active[0] == 0 means the stacked solve has ended (an iteration enqueued behind the last one)

  struct AgentState { double vx[M::NX], vxx[M::NX * M::NX]; };

  MAS_HD void begin(int a, AgentState& s) const {
    constexpr int NX = M::NX;
    double xT[NX];
    MAS_UNROLL for (int i = 0; i < NX; ++i) xT[i] = MAS_AT(X, d.T * NX + i, a);
    terminal_boundary<M>(xT, s.vx, s.vxx, P);
  }
  MAS_HD void assemble(int a, int t, const AgentState& s, QFunction<M>& q) const {
    using L = DerivLayout<M>;
    double Dl[L::ND];
    MAS_UNROLL for (int i = 0; i < L::ND; ++i) Dl[i] = MAS_TILE_AT(D, L::ND, t, a, i);
    assemble_q<M>(Dl, s.vx, s.vxx, q);
  }
  MAS_HD void finish(int a, int t, const QFunction<M>& q, const double* Lfac, AgentState& s) const {
    constexpr int NX = M::NX, NU = M::NU;
    double k[NU], K[NU * NX];
    gains_and_value<M>(q, Lfac, k, K, s.vx, s.vxx);
    MAS_UNROLL for (int i = 0; i < NU; ++i) MAS_AT(kff, t * NU + i, a) = k[i];
    MAS_UNROLL for (int i = 0; i < NU * NX; ++i) MAS_AT(Kfb, t * NU * NX + i, a) = K[i];
  }

#if defined(MAS_HOST_EMULATION)
  void run_serial(int n) const {
    constexpr int NU = M::NU;
    std::vector<AgentState> st(n); std::vector<QFunction<M>> q(n);
    std::vector<double> cur(static_cast<size_t>(n) * NU * NU), Lf(static_cast<size_t>(n) * NU * NU);
    for (int a = 0; a < n; ++a) begin(a, st[a]);
    long long regs = 0;
    for (int t = d.T - 1; t >= 0; --t) {
      for (int a = 0; a < n; ++a) { assemble(a, t, st[a], q[a]); for (int i = 0; i < NU * NU; ++i) cur[a * NU * NU + i] = q[a].quu[i]; }
      double reg = 1e-6; int rungs = 0;
      for (;;) {
        int any = 0;
        for (int a = 0; a < n; ++a) { for (int i = 0; i < NU * NU; ++i) Lf[a * NU * NU + i] = cur[a * NU * NU + i]; any |= llt_compute<NU>(&Lf[a * NU * NU]); }
        if (!any || rungs >= kRegLadderMax) break;
        for (int a = 0; a < n; ++a) for (int i = 0; i < NU; ++i) cur[a * NU * NU + i * NU + i] += reg;
        reg *= 10.0; ++rungs;
      }
      regs += rungs;
      for (int a = 0; a < n; ++a) finish(a, t, q[a], &Lf[a * NU * NU], st[a]);
    }
    reg_events[0] += regs;
  }
#else
  // One lane per agent; the derivative rows are read from an agent-fastest (SoA) copy of the tiles
  // (Dsoa, made by TilesToSoaBody) so that the 64 lanes of a wavefront load 64 consecutive doubles,
  // and the rows of step t-1 are requested while step t is being computed.
  __device__ void run_block(int n) const {
    using L = DerivLayout<M>;
    constexpr int NU = M::NU, ND = L::ND;
    const int a = threadIdx.x;
    const bool valid = a < n;
    if (active && active[0] == 0) return;               // uniform
    AgentState s; QFunction<M> q;
    if (valid) begin(a, s);
    long long regs = 0;
    double Dl[ND], Dn[ND];
    if (valid) { MAS_UNROLL for (int i = 0; i < ND; ++i) Dl[i] = MAS_AT(Dsoa, (d.T - 1) * ND + i, a); }
    for (int t = d.T - 1; t >= 0; --t) {
      double cur[NU * NU], Lf[NU * NU];
      if (valid) {
        const int tn = t > 0 ? t - 1 : 0;
        MAS_UNROLL for (int i = 0; i < ND; ++i) Dn[i] = MAS_AT(Dsoa, tn * ND + i, a);
        assemble_q<M>(Dl, s.vx, s.vxx, q);
        MAS_UNROLL for (int i = 0; i < NU * NU; ++i) cur[i] = q.quu[i];
        MAS_UNROLL for (int i = 0; i < ND; ++i) Dl[i] = Dn[i];
      }
      double reg = 1e-6; int rungs = 0;
      for (;;) {                                        // every thread of the block takes the same trips
        int fail = 0;
        if (valid) { MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Lf[i] = cur[i]; fail = llt_compute<NU>(Lf); }
        const int any = __syncthreads_or(fail);
        if (!any || rungs >= kRegLadderMax) break;
        if (valid) { MAS_UNROLL for (int i = 0; i < NU; ++i) cur[i * NU + i] += reg; }
        reg *= 10.0; ++rungs;
      }
      regs += rungs;
      if (valid) finish(a, t, q, Lf, s);
    }
    if (a == 0) reg_events[0] += regs;
  }
#endif
};

// merit of every alpha trial = serial sum over agents; first strict decrease wins for ALL agents
// (ilqr.hpp:219-226 on the stacked problem); termination ilqr.hpp:269-271.
struct CentralSelectBody {
  Dims d; double tolerance;
  const double* trial_merit; double* cost; double* merit; int* alpha_sel; int* iters_total; CentralState* st;
  // non-null: the host enqueues iterations ahead of reading `done` back.  The iteration that terminates the solve clears every
  // agent's flag, which turns the kernels of an iteration enqueued behind it into no-ops (sweep, line search: no active agent;
  // backward pass: active[0]; this body: st->done; commit: alpha_sel = -1).
  int* active = nullptr;
  // folded iterations (GPU build, run_centralized): this kernel is also the iteration's PUBLISHER -- the state and the word the host polls,
  // stored by the thread that decided -- and it clears the words of the backward pass's rung agreement for the next iteration (the
  // "gave up" flag after it has been published), so neither a commit kernel nor a memset stands between two iterations
  Publish publish{};
  unsigned long long* clear_words = nullptr; int n_clear_words = 0, n_clear_tail = 0;

  MAS_HD static void decide(const double* sums, double tolerance, CentralState* st) {
    const double cur = st->cur_total;
    int sel = -1; double best = cur;
    for (int j = 0; j < kNumAlphas; ++j) if (sums[j] < best) { best = sums[j]; sel = j; break; }
    st->improvement = cur - best; st->cur_total = best; st->j_sel = sel; st->iterations += 1;
    st->done = (st->improvement < tolerance) ? 1 : 0;
  }
  MAS_HD void apply(int a, int sel) const {
    alpha_sel[a] = sel; iters_total[a] += 1;
    if (sel >= 0) { const double m = trial_merit[static_cast<size_t>(sel) * d.Np + a]; cost[a] = m; merit[a] = m; }
  }
#if defined(MAS_HOST_EMULATION)
  void run_serial(int n) const {
    double sums[kNumAlphas];
    for (int j = 0; j < kNumAlphas; ++j) { double s = 0.0; for (int a = 0; a < n; ++a) s += trial_merit[static_cast<size_t>(j) * d.Np + a]; sums[j] = s; }
    decide(sums, tolerance, st);
    for (int a = 0; a < n; ++a) apply(a, st->j_sel);
  }
#else
  __device__ void run_block(int n) const {
    __shared__ double sums[kNumAlphas];
    __shared__ int sel_sh;
    const int tid = threadIdx.x;
    if (st->done != 0) {                                                 // an iteration enqueued behind the last one (uniform: nobody has written st yet)
      if (tid < n) alpha_sel[tid] = -1;
      if (tid == 0 && publish.host) publish();
      return;
    }
    const int wave = tid >> 6, n_waves = (static_cast<int>(blockDim.x) + 63) >> 6;
    for (int j = wave; j < kNumAlphas; j += n_waves) {                  // one wavefront per alpha trial (several trials per wavefront if the block is small)
      const double s = wave_serial_sum(trial_merit + static_cast<size_t>(j) * d.Np, n);
      if ((tid & 63) == 0) sums[j] = s;
    }
    __syncthreads();
    __shared__ int done_sh;
    if (tid == 0) {
      decide(sums, tolerance, st); sel_sh = st->j_sel; done_sh = st->done;
      if (publish.host) publish();
      for (int q = 0; q < n_clear_tail; ++q) clear_words[n_clear_words + q] = 0ull;        // (the flag: published above)
    }
    for (int q = tid; q < n_clear_words; q += static_cast<int>(blockDim.x)) clear_words[q] = 0ull;
    __syncthreads();
    if (tid < n) { apply(tid, sel_sh); if (active && done_sh) active[tid] = 0; }
  }
#endif
};

}  // namespace mas
