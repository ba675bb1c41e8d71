// mas_solver.hpp -- host-side batch driver of the MI355X iLQR path.
//
// One BatchSolver owns the device-resident state of N independent agents of one workload
// model and runs the reference's strategies over them:
//   solve()            N x iLQR::solve                         ilqr.hpp:58-273 / nash.hpp:59-64
//   run_sequential()   detail::run_sequential                  nash.hpp:74-90
//   run_trust_region() detail::run_trust_region                nash.hpp:182-248
//   run_line_search()  detail::run_line_search                 nash.hpp:92-180
//   run_centralized()  CentralizedStrategy (block-diagonal)    centralized.hpp:18-38
// Per batch iteration it launches: linearisation sweep -> backward Riccati -> speculative
// line search -> select -> commit, and reads back one integer (agents still active).
#pragma once

#include <atomic>
#include <cmath>
#include <cstring>
#include <iostream>
#include <limits>
#include <memory>

#include "mas_backend.hpp"
#include "mas_bodies.hpp"
#include "mas_central.hpp"
#include "mas_dense.hpp"
#include "mas_rccl.hpp"
#include "mas_forms.hpp"      // the kernel forms this build can launch (product: hand-written gfx950 kernels; the test-only host emulation: none)

namespace mas {

struct ProblemDesc {
  int model_id = 0, n_agents = 0, horizon_steps = 0, device = 0;
  double dt = 0.0;
  bool has_params = false; double params[16] = {0};
  bool has_bounds = false, no_bounds = false; double lb[8] = {0}, ub[8] = {0};
};

struct HostStats {
  int64_t agent_iterations = 0, converged_solves = 0, solves = 0, reg_events = 0, batch_iterations = 0;
  double wall_ms = 0.0;
};

class SolverBase {
 public:
  virtual ~SolverBase() = default;
  virtual int nx() const = 0;
  virtual int nu() const = 0;
  virtual void set_initial_states(const double* x0, bool device_ptr) = 0;
  virtual void set_controls(const double* U, bool device_ptr) = 0;
  virtual void initialize_problem() = 0;
  virtual void save_initial() = 0;
  virtual void restore_initial() = 0;
  virtual void solve() = 0;
  virtual void run_sequential(int max_outer) = 0;
  virtual void run_trust_region(int max_outer) = 0;
  virtual void run_line_search(int max_outer) = 0;
  virtual void run_centralized() = 0;
  virtual void run_centralized_dense() = 0;
  virtual void get_solution(double* X, double* U, double* cost, int32_t* iters, int32_t* exit_code, bool device_ptr) = 0;
  virtual double total_cost() = 0;
  virtual void get_trust_radii(double* radii) = 0;
  virtual void debug_backward(double* derivs, double* k, double* K, int64_t* regs) = 0;
  virtual HostStats get_stats() = 0;
  virtual void fold_kernel_stamps() {}
  virtual void reset_stats() = 0;

  // iLQR::set_params, ilqr.hpp:39-55
  bool set_param(const std::string& key, double value) {
    if (key == "max_iterations") { max_iterations = static_cast<int>(value); has_max_iterations = true; }
    else if (key == "tolerance") { tolerance = value; has_tolerance = true; }
    else if (key == "max_ms") { max_ms = value; has_max_ms = true; }
    else if (key == "debug") { debug = value > 0.5; }
    else if (key == "penalty") { penalty = value; penalty_set = true; on_penalty_changed(); }                 // ilqr.hpp:47-54
    else if (key == "penalty_increase") { penalty_increase = value; }
    else if (key == "constraint_tolerance") { constraint_tolerance = value; }
    else if (key == "inequality_activation_tolerance") { inequality_activation_tolerance = value; }
    else return false;   // unknown keys: ignored like the reference
    return true;
  }
  bool params_complete() const { return has_max_iterations && has_tolerance && has_max_ms; }

  // total cost over all ranks' agents when the block list is sharded (mas_hip_set_cost_exchange); null = local
  using CostExchange = int (*)(void* ctx, const double* local_costs, int32_t n_local, double* global_total);
  CostExchange cost_exchange = nullptr;
  void* cost_exchange_ctx = nullptr;
  // agents sharded over GPUs with RCCL inside the library (mas_hip_set_comm_cost_exchange / mas_hip_all_gather_solution)
  virtual void set_comm_exchange(void* comm, int n_total) = 0;
  virtual void all_gather_solution(void* comm, int n_total, double* X_all, double* U_all, double* cost_all) = 0;

  // Which of several BIT-IDENTICAL kernel forms a launch uses (tests hold the forms against each other; tools/ measures them).
  // Defaults are the measured best.  Set through mas_hip_set_option; a build with -DMAS_HIP_ENV_SWITCHES (experiments only,
  // never the product build) also reads MAS_HIP_<KEY> from the environment, once, when the handle is created.
  struct KernelForms {
    bool fused = true;                      // two-kernel rounds for the four-state models (mas_fused_hip.hpp)
    long long forward_quad_max = 1600;      // four-lanes-per-rollout line search up to this many live agents (0: never)
    bool backward_tile = false;             // the LDS-tile backward kernel also for NX == 4
    bool backward_list = true;              // the LDS-tile kernel walks the round's list of live agents
    long long backward_wide_max = 4096;     // LDS-tile kernel with 32 lanes per agent up to this many live agents (0: never), NX > 4
    long long backward_wave_max = 1024;     // ... with a whole wavefront per agent up to this many
    bool trust_reroll_levels = true;        // the re-roll of the three-kernel epilogue by dependency levels (a workgroup per queued agent) instead of four lanes per rollout
    long long trust_fused_max = 4096;       // chained trust-region epilogue as ONE kernel (re-roll by dependency levels, mas_kernel_levels.hpp) up to this many live agents (0: never)
    bool central_pipeline = true;           // centralised backward pass with the speculative step
    bool central_one_xcd = true;            // its working workgroups on one XCD
    bool forward_staged = true;             // models with kStagedLineSearch (the identity LQR): first step size alone, the others only where it failed (false: all ten at once)
    bool central_fold = true;               // centralised iterations in four launches: the sweep commits, the select publishes and clears (false: commit kernel + memset)
    bool force_single_workgroup_central = false;   // the single-workgroup fallback of the shared ladder
    bool fused_trust = true;                // chained trust-region epilogue inside the fused kernel (four-state FD models); false: the five-kernel round
    bool dense_ladder_blocked = true;       // chip-wide literal solve: the ladder's sixteen LLTs blocked by panels across the chip; false: one workgroup per rung, column by column
    bool dense_chip = true;                 // the literal stacked solve as chip-wide launches (mas_dense_chip.hpp); false: one workgroup (mas_dense.hpp, <= 64 agents)
    int central_combine = 1, central_wpb = kCentralWavesPerBlock, central_peek = 1;
    bool set(const std::string& key, long long v) {
      if (key == "fused") fused = v != 0;
      else if (key == "forward_quad_max") forward_quad_max = v;
      else if (key == "backward_tile") backward_tile = v != 0;
      else if (key == "backward_list") backward_list = v != 0;
      else if (key == "backward_wide_max") backward_wide_max = v;
      else if (key == "backward_wave_max") backward_wave_max = v;
      else if (key == "trust_fused_max") trust_fused_max = v;
      else if (key == "trust_reroll_levels") trust_reroll_levels = v != 0;
      else if (key == "central_pipeline") central_pipeline = v != 0;
      else if (key == "central_one_xcd") central_one_xcd = v != 0;
      else if (key == "central_fold") central_fold = v != 0;
      else if (key == "forward_staged") forward_staged = v != 0;
      else if (key == "force_single_workgroup_central") force_single_workgroup_central = v != 0;
      else if (key == "dense_chip") dense_chip = v != 0;
      else if (key == "fused_trust") fused_trust = v != 0;
      else if (key == "dense_ladder_blocked") dense_ladder_blocked = v != 0;
      else if (key == "central_combine") central_combine = static_cast<int>(v);
      else if (key == "central_peek") central_peek = static_cast<int>(v);
      else if (key == "central_wpb") { if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) central_wpb = static_cast<int>(v); else return false; }
      else return false;
      return true;
    }
    void from_env() {
#if defined(MAS_HIP_ENV_SWITCHES)
      static const char* keys[] = {"fused", "forward_quad_max", "backward_tile", "backward_list", "backward_wide_max", "backward_wave_max", "trust_fused_max", "trust_reroll_levels", "central_pipeline", "central_one_xcd",
                                   "force_single_workgroup_central", "central_combine", "central_wpb", "central_peek", "dense_chip", "fused_trust", "dense_ladder_blocked", "central_fold", "forward_staged"};
      for (const char* k : keys) {
        std::string name = "MAS_HIP_";
        for (const char* c = k; *c; ++c) name += static_cast<char>(*c >= 'a' && *c <= 'z' ? *c - 32 : *c);
        if (const char* e = std::getenv(name.c_str())) set(k, std::atoll(e));
      }
#endif
    }
  };
  KernelForms switches;
  bool set_option(const std::string& key, long long v) { const bool ok = switches.set(key, v); on_option_changed(); return ok; }
  virtual void on_option_changed() {}

  Backend be;
  std::string last_error;
  // defaults of the iLQR constructor, ilqr.hpp:26-37
  int max_iterations = 50;
  double tolerance = 1e-6;
  double max_ms = std::numeric_limits<double>::infinity();
  bool debug = false;
  bool has_max_iterations = false, has_tolerance = false, has_max_ms = false;
  // augmented-Lagrangian parameters, defaults of ilqr.hpp:31-34
  double penalty = 10.0, penalty_increase = 5.0, constraint_tolerance = 1e-4, inequality_activation_tolerance = 1e-6;
  bool penalty_set = false;
  virtual void on_penalty_changed() {}
};

template <class M>
class SolverImpl final : public SolverBase {
  static constexpr int NX = M::NX, NU = M::NU;
  static constexpr bool kConstrained = (M::NEQ + M::NIN) > 0;
  // lanes that share one agent's backward recursion in the LDS-tile kernel
#if defined(MAS_LPA6)
  static constexpr int kLanes = (NX > 4) ? MAS_LPA6 : kBackwardLanesPerAgent;      // (A/B builds)
#else
  static constexpr int kLanes = kBackwardLanesPerAgent;
#endif
  // the two-kernel round of mas_fused_hip.hpp.  Finite-difference models: helper wavefronts linearise beside the recursion, NU <= 2
  // only (with NU = 4 the recursion alone fills the 256 VGPRs a wavefront may have when helpers share its SIMD: 54 spilled
  // registers, C4 17.3 -> 21.5 ms).  Models whose derivatives are all analytic: the recursion's lanes evaluate them in registers.
  static constexpr bool kAllAnalytic = M::kAnalyticA && M::kAnalyticB && M::kAnalyticLx && M::kAnalyticLu && M::kAnalyticLxx && M::kAnalyticLuu && M::kAnalyticLux;
  static constexpr bool kFusable = NX == 4 && !kConstrained && ((kAllAnalytic && NU == 4) || NU <= 2);
  using L = DerivLayout<M>;

 public:
  SolverImpl(const ProblemDesc& desc, void* external_stream) {
    d.N = desc.n_agents; d.T = desc.horizon_steps; d.Np = (static_cast<size_t>(desc.n_agents) + 63) / 64 * 64;
    dt = desc.dt;
    if (desc.has_params) for (int i = 0; i < 16; ++i) P.p[i] = desc.params[i];
    else model_default_params(desc.model_id, desc.horizon_steps, P.p);
    clamp = false;
    for (int i = 0; i < NU; ++i) { lb[i] = 0.0; ub[i] = 0.0; }
    if (!desc.no_bounds) {
      if (desc.has_bounds) { for (int i = 0; i < NU; ++i) { lb[i] = desc.lb[i]; ub[i] = desc.ub[i]; } clamp = true; }
      else if (M::kDefaultBounds) { M::default_bounds(lb, ub); clamp = true; }
    }
    be.init(desc.device, external_stream);
    const size_t Np = d.Np; const int T = d.T;
    x0 = dalloc(NX * Np); X = dalloc((T + 1) * NX * Np); U = dalloc(static_cast<size_t>(T) * NU * Np);
    D = dalloc(static_cast<size_t>(T) * L::ND * Np);
    kff = dalloc(static_cast<size_t>(T) * NU * Np); Kfb = dalloc(static_cast<size_t>(T) * NU * NX * Np);
    Xs = dalloc(static_cast<size_t>(kNumAlphas) * (T + 1) * NX * Np); Us = dalloc(static_cast<size_t>(kNumAlphas) * T * NU * Np);
    trial_merit = dalloc(static_cast<size_t>(kNumAlphas) * Np);
    cost = dalloc(Np); merit = dalloc(Np);
    x0_saved = dalloc(NX * Np); U_saved = dalloc(static_cast<size_t>(T) * NU * Np);
    live_lists[0] = ialloc(Np); live_lists[1] = ialloc(Np); live_count0 = ialloc(16); solves_left = ialloc(Np);
    slot_of[0] = ialloc(Np); slot_of[1] = ialloc(Np); fused_error = ialloc(16);
    active = ialloc(Np); alpha_sel = ialloc(Np); iters_solve = ialloc(Np); iters_total = ialloc(Np); exit_code = ialloc(Np);
    reg_events = static_cast<long long*>(raw_alloc(sizeof(long long) * Np));
    counters = static_cast<long long*>(raw_alloc(sizeof(long long) * 32));
    central = static_cast<CentralState*>(raw_alloc(sizeof(CentralState)));
    if constexpr (kConstrained) {
      if (M::NEQ > 0) al_lam = dalloc(static_cast<size_t>(T) * M::NEQ * Np);
      if (M::NIN > 0) al_mu = dalloc(static_cast<size_t>(T) * M::NIN * Np);
      al_rho = dalloc(Np); improvement = dalloc(Np); trial_obj = dalloc(static_cast<size_t>(kNumAlphas) * Np);
      al_rho_dirty = true;
    }
    const size_t stage_elems = static_cast<size_t>(d.N) * std::max<size_t>(static_cast<size_t>(T + 1) * NX, static_cast<size_t>(T) * std::max(L::ND, NU * NX));
    stage = dalloc(stage_elems);
    host_counters = static_cast<long long*>(be.alloc_pinned(sizeof(long long) * 32));
    host_central = static_cast<HostCentral*>(be.alloc_pinned(2 * sizeof(HostCentral)));   // two read-back slots (the centralised loop runs one iteration ahead)
    std::memset(host_counters, 0, sizeof(long long) * 32);      // hipHostMalloc does not zero: a word the device never writes
    std::memset(host_central, 0, 2 * sizeof(HostCentral));      // (the ladder's give-up flag without a ladder) must read 0
    switches.from_env();
    be.sync();
  }
  ~SolverImpl() override {
    try { be.sync(); } catch (...) {}
    if (round_events_made) for (auto& e : round_ev) be.drop_event(e);
    for (void* p : allocations) be.free(p);
    be.free_pinned(host_counters); be.free_pinned(host_central); if (host_joint) be.free_pinned(host_joint);
    free_chip_word();
    be.destroy();
  }

  int nx() const override { return NX; }
  int nu() const override { return NU; }

  void set_initial_states(const double* src, bool device_ptr) override {
    upload(src, static_cast<size_t>(d.N) * NX, device_ptr);
    be.template launch_2d<1>("aos_to_soa", AosToSoaBody{d, NX, stage, x0, 1}, d.N, 1);
    be.zero(iters_total, sizeof(int) * d.Np);            // new initial states = a new problem: per-agent iteration counts restart
    have_x0 = true; initialized = false; rollout_consistent = false;
  }
  void set_controls(const double* src, bool device_ptr) override {
    controls_zero = (src == nullptr);
    if (!src) { be.zero(U, sizeof(double) * d.T * NU * d.Np); }
    else {
      upload(src, static_cast<size_t>(d.N) * d.T * NU, device_ptr);
      be.template launch_2d<4>("aos_to_soa", AosToSoaBody{d, NU, stage, U, d.T}, d.N, d.T);
    }
    initialized = false; rollout_consistent = false;
  }
  // integrate_horizon + objective of every agent: four lanes per agent where the dynamics allow it (mas_rollout_quad_kernel)
  void launch_rollout_cost() {
    const RolloutCostBody<M> body{d, dt, P, x0, U, X, cost, merit, nullptr, al_view(al_enabled)};
    if (form_rollout_cost(body)) return;
    be.template launch_1d<1>("rollout_cost", body, d.N);
  }
  void launch_trust_reroll(const TrustRerollBody<M>& body) {
    if (form_trust_reroll(body)) return;
    be.template launch_1d<1>("trust_reroll", body, d.N);
  }
  void initialize_problem() override {
    require(have_x0, "initial states not set");
    launch_rollout_cost();
    if (!radii) radii = dalloc(d.Np);
    initialized = true; rollout_consistent = true;
  }
  void save_initial() override {
    be.d2d(x0_saved, x0, sizeof(double) * NX * d.Np); be.d2d(U_saved, U, sizeof(double) * d.T * NU * d.Np); saved = true;
    saved_controls_zero = controls_zero;
    // the rollout and cost of these controls (OCP::initialize_problem, ocp.hpp:102-183 -- problem set-up, which the reference's examples
    // do before they start their clock, examples/multi_agent_single_track.cpp:113-123) are saved with them, so that a restore is three
    // copies instead of a T-step rollout chain
    saved_rollout = initialized && rollout_consistent;
    if (saved_rollout) {
      if (!X_saved) { X_saved = dalloc(static_cast<size_t>(d.T + 1) * NX * d.Np); cost_saved = dalloc(d.Np); }
      be.d2d(X_saved, X, sizeof(double) * (d.T + 1) * NX * d.Np); be.d2d(cost_saved, cost, sizeof(double) * d.Np);
    }
  }
  void restore_initial() override {
    require(saved, "no saved initial state");
    be.d2d(x0, x0_saved, sizeof(double) * NX * d.Np); be.d2d(U, U_saved, sizeof(double) * d.T * NU * d.Np);
    controls_zero = saved_controls_zero;
    be.zero(iters_total, sizeof(int) * d.Np);   // per-agent iteration counts restart with the problem; reg_events keeps counting since reset_stats (no read-back here)
    have_x0 = true;
    if (saved_rollout && !(kConstrained && al_enabled)) {
      be.d2d(X, X_saved, sizeof(double) * (d.T + 1) * NX * d.Np); be.d2d(cost, cost_saved, sizeof(double) * d.Np);
      be.d2d(merit, cost_saved, sizeof(double) * d.Np);         // current merit == objective without constraint functions (ilqr.hpp:379-407)
      if (!radii) radii = dalloc(d.Np);
      initialized = true; rollout_consistent = true;
    } else {
      initialize_problem();
    }
  }

  // ---------------------------------------------------------------- N x iLQR::solve ------
  void solve() override { Timer tm(*this); solve_all(); }

  void run_sequential(int max_outer) override {
    reset_solver_state();   // the strategy creates fresh solver objects (nash.hpp:78-84)          // nash.hpp:74-90
    Timer tm(*this);
    ensure_initialized();
    // update_initial_with_best (nash.hpp:66-71) has no device-side effect, and an agent's solve k+1 depends only on
    // its own solve k: the max_outer solves of every agent run back to back in one stream of rounds (no host
    // barrier between outer iterations); a finite max_ms is honoured per solve by start stamps on the device.  The
    // multiplier update between solves and the per-agent debug trace need the host between them and keep the loop.
    const bool chain = max_outer > 1 && !(kConstrained && al_enabled) && !debug;
    if (chain) solve_all(max_outer);
    else for (int outer = 0; outer < max_outer; ++outer) solve_all();
    last_total_valid = false;
  }

  void run_trust_region(int max_outer) override {
    reset_solver_state();   // the strategy creates fresh solver objects (nash.hpp:78-84)        // nash.hpp:182-248
    Timer tm(*this);
    ensure_initialized();
    ensure_snapshots();
    fill(radii, 1.0);
    // An agent's outer iteration k+1 (solve, then the trust-region test on its own step) depends only on its own
    // iteration k, so the outer iterations are chained on the device like the solves of the sequential strategy: an agent
    // whose solve ends runs its epilogue in that round (TrustNormBody / TrustRerollBody / TrustCopyBody) and starts its
    // next solve in the following one -- no host barrier, no waiting for the slowest agent of every outer iteration.  The
    // multiplier update between solves and the per-agent debug trace need the host between solves and keep the loop.
    // (a solve that cannot run an iteration -- max_iterations 0 or a negative time budget -- enqueues no round and with it no
    // epilogue: those cases take the loop below, whose trust-region test follows every solve as a kernel of its own; found by
    // holding the radii against the oracle's in round 4: 0.5^max_outer there, 1.0 here before)
    const bool no_iteration = max_iterations <= 0 || (max_ms < std::numeric_limits<double>::infinity() && 0.0 > max_ms);
    const bool chain = max_outer > 1 && !(kConstrained && al_enabled) && !debug && !no_iteration;
    if (chain) {
      snapshot_old();
      solve_all(max_outer, /*trust_epilogue=*/true);
    } else {
      for (int outer = 0; outer < max_outer; ++outer) {
        snapshot_old();
        solve_all();
        be.template launch_1d<1>("trust_region", TrustRegionBody<M>{d, dt, P, x0, X, U, cost, X_old, U_old, cost_old, radii}, d.N);
      }
    }
    last_total_valid = false;
  }

  void run_line_search(int max_outer) override {
    reset_solver_state();   // the strategy creates fresh solver objects (nash.hpp:78-84)         // nash.hpp:92-180
    Timer tm(*this);
    ensure_initialized();
    ensure_snapshots();
    if (cost_exchange) { run_line_search_host_exchange(max_outer); return; }
    // The joint step is decided on the device (JointState): total costs by one wavefront in block order -- over every rank's agents
    // after an RCCL all-gather of the shards' costs when the block list is sharded (set_comm_exchange) -- and ONE read-back per
    // outer iteration (does the alpha ladder run at all?); the ladder's nine blend / sum pairs are enqueued without the host.
    if (!joint) { joint = static_cast<JointState*>(raw_alloc(sizeof(JointState))); host_joint = static_cast<JointState*>(be.alloc_pinned(sizeof(JointState))); }
    joint_sum_device(JointSumBody::kBase);
    for (int outer = 0; outer < max_outer; ++outer) {
      be.d2d(X_old, X, sizeof(double) * (d.T + 1) * NX * d.Np); be.d2d(U_old, U, sizeof(double) * d.T * NU * d.Np);
      be.d2d(cost_old, cost, sizeof(double) * d.Np);
      solve_all();
      joint_sum_device(JointSumBody::kAfterSolves);
      be.d2h(host_joint, joint, sizeof(JointState));
      if (!host_joint->need_blend) continue;
      be.d2d(U_cand, U, sizeof(double) * d.T * NU * d.Np);
      for (double alpha = 0.5; alpha > 1e-3; alpha *= 0.5) {             // nash.hpp:126-159
        be.template launch_1d<1>("blend_rollout", BlendRolloutBody<M>{d, dt, alpha, P, x0, U_old, U_cand, X, U, cost, joint}, d.N);
        joint_sum_device(JointSumBody::kAfterTrial);
      }
      be.template launch_2d<4>("revert", RevertBody<M>{d, joint, X, U, cost, X_old, U_old, cost_old}, d.N, d.T + 1);
    }
    // Solution.total_cost (collect_solution, nash.hpp:23-37): the sum over ALL agents in block order -- with the block list sharded over
    // ranks that is every rank's agents, formed like the totals the strategy decided on (one more all-gather + serial sum)
    joint_sum_device(JointSumBody::kFinal);
    be.d2h(host_joint, joint, sizeof(JointState));
    central_total = host_joint->last_total; last_total_valid = true;
  }
  // the same strategy with the total cost formed by the caller's host callback (mas_hip_set_cost_exchange: torch.distributed / gloo)
  void run_line_search_host_exchange(int max_outer) {
    double base_cost = joint_sum(cost);
    for (int outer = 0; outer < max_outer; ++outer) {
      be.d2d(X_old, X, sizeof(double) * (d.T + 1) * NX * d.Np); be.d2d(U_old, U, sizeof(double) * d.T * NU * d.Np);
      be.d2d(cost_old, cost, sizeof(double) * d.Np);
      solve_all();
      const double new_cost = joint_sum(cost);
      if (new_cost >= base_cost) {
        be.d2d(U_cand, U, sizeof(double) * d.T * NU * d.Np);
        double alpha = 0.5; bool accepted = false;
        while (alpha > 1e-3 && !accepted) {
          be.template launch_1d<1>("blend_rollout", BlendRolloutBody<M>{d, dt, alpha, P, x0, U_old, U_cand, X, U, cost}, d.N);
          const double trial_cost = joint_sum(cost);
          if (trial_cost < base_cost) { base_cost = trial_cost; accepted = true; }
          else alpha *= 0.5;
        }
        if (!accepted) {
          be.d2d(X, X_old, sizeof(double) * (d.T + 1) * NX * d.Np); be.d2d(U, U_old, sizeof(double) * d.T * NU * d.Np);
          be.d2d(cost, cost_old, sizeof(double) * d.Np);   // objective(old) == the stored old cost
        }
      } else base_cost = new_cost;
    }
    last_total_valid = false;
  }
  // sum of all agents' costs in global block order + the decision of `mode`, on the stream
  void joint_sum_device(int mode) {
    const double* costs = cost; int n = d.N;
    if (comm_exchange) {
      int lo = 0, hi = 0; shard_range(comm_total, comm_exchange->n_ranks, comm_exchange->rank, &lo, &hi);
      be.d2d(comm_costs + lo, cost, sizeof(double) * d.N);
      comm_exchange->all_gather_shards(comm_costs, comm_total, 1, be.cur);
      costs = comm_costs; n = comm_total;
    }
    be.launch_group("joint_sum", JointSumBody{costs, n, joint, mode}, 64);
  }
  void set_comm_exchange(void* comm, int n_total) override {
    comm_exchange = static_cast<Comm*>(comm);
    if (comm_exchange) {
      int lo = 0, hi = 0; shard_range(n_total, comm_exchange->n_ranks, comm_exchange->rank, &lo, &hi);
      if (hi - lo != d.N) { comm_exchange = nullptr; throw std::invalid_argument("this handle's agent count is not the shard size of its rank"); }
      if (comm_total != n_total || !comm_costs) comm_costs = dalloc(static_cast<size_t>(n_total));
      comm_total = n_total;
    }
  }
  // publish step of the Nash strategies (update_initial_with_best for coupled costs, nash.hpp:66-71): every rank receives every
  // agent's trajectory and cost in the reference layout, [n_total][T+1][nx], [n_total][T][nu], [n_total].  The shard is written
  // straight to its displacement in the caller's buffers (layout conversion on the device) and travels as one collective per array.
  void all_gather_solution(void* comm_, int n_total, double* X_all, double* U_all, double* cost_all) override {
    const Comm* c = static_cast<const Comm*>(comm_);
    if (!c) throw std::invalid_argument("null communicator");
    int lo = 0, hi = 0; shard_range(n_total, c->n_ranks, c->rank, &lo, &hi);
    if (hi - lo != d.N) throw std::invalid_argument("this handle's agent count is not the shard size of its rank");
    const size_t px = static_cast<size_t>(d.T + 1) * NX, pu = static_cast<size_t>(d.T) * NU;
    if (X_all) { be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NX, X, X_all + lo * px, d.T + 1}, d.N, d.T + 1); c->all_gather_shards(X_all, n_total, px, be.cur); }
    if (U_all) { be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NU, U, U_all + lo * pu, d.T}, d.N, d.T); c->all_gather_shards(U_all, n_total, pu, be.cur); }
    if (cost_all) { be.d2d(cost_all + lo, cost, sizeof(double) * d.N); c->all_gather_shards(cost_all, n_total, 1, be.cur); }
    be.sync();
  }

  void run_centralized() override {                      // centralized.hpp:18-38, block-diagonal reading
    Timer tm(*this);
    struct AlOff { bool& f; bool old; ~AlOff() { f = old; } } al_off{al_enabled, al_enabled};
    al_enabled = false;                                  // build_global_ocp stacks no constraint functions
    require(have_x0, "initial states not set");
    require(d.N <= kCentralMaxAgents, "centralized strategy supports at most 1024 agents per device");
    // the stacked OCP starts from zero controls (multi_agent_problem.hpp:124, ocp.hpp:106-109); if the device already holds zero
    // controls and their rollout (set_controls(null) / a restore of such a state, then initialize_problem), that is this very state
    if (!(controls_zero && initialized && rollout_consistent)) {
      be.zero(U, sizeof(double) * d.T * NU * d.Np);
      initialize_problem();
    }
    be.template launch_1d<1>("begin_solve", BeginSolveBody{active, iters_solve, exit_code, alpha_sel}, d.N);
    be.launch_group("central_init", CentralInitBody{d, cost, central}, d.N);
    stats.solves += 1;
    const auto t0 = std::chrono::steady_clock::now();
    int exit_reason = kExitMaxIter;
    // One iteration = sweep -> backward -> line search -> select -> commit + a read-back of the solve's state.  Like solve_all
    // the host runs one iteration AHEAD of the device: iteration r+1 is enqueued before the read-back of r is awaited; if r
    // turns out to have ended the solve, the kernels enqueued behind it find no active agent (CentralSelectBody::active).
    // A finite max_ms is checked at the top of every iteration (ilqr.hpp:84-90) and keeps the loop in step with the device.
    const bool run_ahead = kHostRunsAhead && std::isinf(max_ms) && max_ms > 0 && !debug;
    struct Marks { size_t p0, p1, s0, s1; };
    Marks marks[2];
    int enq = 0;
    bool ladder_gave_up = false;
    // the commit kernel of every iteration publishes the solve's state (host_central, read once when the loop has ended) and ONE word the
    // host polls: [iteration + 1 : 32][gave up : 1][done : 1] -- no event between iterations (an event record costs ~5.5 us of idle device per iteration)
    volatile long long* poll = host_counters + 24;
    poll[0] = 0; poll[1] = 0;
    auto enqueue = [&]() {
      Marks& mk = marks[enq & 1];
      mk.p0 = be.mark(); mk.s0 = stamp_launches.size();
      const double nact = static_cast<double>(d.N);
      // folded (GPU build): the sweep commits what the iteration before accepted, the select publishes -- no commit kernel in the loop
      const bool folded = form_central_folded();
      if (folded) form_sweep_commit(sweep_bytes() * nact); else launch_sweep(sweep_bytes() * nact);
      launch_central_backward(backward_bytes() * nact);
      launch_forward(make_forward(), forward_bytes() * nact * kNumAlphas, nullptr, nullptr, d.N);     // every agent active: slot == agent, CommitBody serves both forms
      // one wavefront per alpha trial for the ten serial sums (at least ten wavefronts, and a thread per agent for the bookkeeping)
      static_assert(sizeof(CentralState) % sizeof(long long) == 0, "published word by word");
      Publish pub{reinterpret_cast<const long long*>(central), reinterpret_cast<long long*>(host_central + (enq & 1)), static_cast<int>(sizeof(CentralState) / sizeof(long long)),
                  ladder_error_flag()};
      pub.poll_word = host_counters + 24 + (enq & 1); pub.poll_tag = enq + 1; pub.poll_done = &central->done;
      CentralSelectBody sel{d, tolerance, trial_merit, cost, merit, alpha_sel, iters_total, central, run_ahead ? active : nullptr};
      if (folded) { sel.publish = pub; form_central_select_extras(sel); }
      be.launch_group("central_select", sel, d.N, 0.0, 0, std::max(64 * kNumAlphas, static_cast<int>((d.N + 63) / 64 * 64)));
      if (!folded) be.template launch_2d<4>("commit", CommitBody<M>{d, Xs, Us, alpha_sel, X, U, pub}, d.N, d.T + 1);
      mk.p1 = be.mark(); mk.s1 = stamp_launches.size();
      ++enq;
    };
    auto await = [&](int r) -> long long {
      const auto w0 = std::chrono::steady_clock::now();
      for (long long spins = 1;; ++spins) {
        const long long w = poll[r & 1];
        if ((w >> 32) == r + 1) return w;
        if ((spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 30.0) {
          try { be.sync(); } catch (...) {}
          throw std::runtime_error("the device did not publish an iteration of the centralised solve within 30 s");
        }
      }
    };
    if (max_iterations > 0) { if (timed_out(t0)) exit_reason = kExitTime; else enqueue(); }
    for (int r = 0; r < enq; ++r) {
      if (run_ahead && enq == r + 1 && enq < max_iterations) enqueue();
      const long long w = await(r);
      stats.agent_iterations += d.N;
      stats.batch_iterations++;
      if (w & 2) ladder_gave_up = true;
      if (w & 1) {
        exit_reason = kExitConverged;
        if (enq > r + 1) {                               // the iteration enqueued ahead runs on no agent: nothing to account
          be.scale_bytes(marks[(r + 1) & 1].p0, marks[(r + 1) & 1].p1, 0.0);
          for (size_t q = marks[(r + 1) & 1].s0; q < marks[(r + 1) & 1].s1; ++q) stamp_launches[q].bytes = 0.0;
        }
        break;
      }
      if (enq == r + 1 && enq < max_iterations) { if (timed_out(t0)) { exit_reason = kExitTime; break; } enqueue(); }
    }
    // folded iterations: the trial the LAST select accepted has no sweep behind it to commit it (an iteration that ran ahead has done so and
    // reset alpha_sel: then this copies nothing)
    if (enq > 0 && form_central_folded()) be.template launch_2d<4>("commit", CommitBody<M>{d, Xs, Us, alpha_sel, X, U, Publish{}}, d.N, d.T + 1);
    be.sync();                                           // drains an iteration that ran ahead; the state below was published by the last commit / select
    if (ladder_gave_up) throw std::runtime_error("centralised backward pass: a wavefront gave up waiting for the rung agreement");   // (the flag is cleared with the words before every launch)
    // the state after the last executed iteration was published with it (an iteration enqueued behind the last one leaves it as it is);
    // only a solve without any iteration has to fetch what central_init wrote
    CentralState final_state{};
    if (enq > 0) final_state = host_central[(enq - 1) & 1].st;
    else { be.d2h(&host_central[0].st, central, sizeof(CentralState)); final_state = host_central[0].st; }
    central_total = final_state.cur_total; last_total_valid = true;
    if (exit_reason == kExitConverged) stats.converged_solves += 1;
    fill_int(exit_code, exit_reason);
    controls_zero = false;
  }

  // centralized.hpp:18-38 in the reference's literal reading: ONE iLQR on the stacked OCP with finite differences of the
  // summed cost (mas_dense.hpp).  Small problems only; one workgroup walks the whole solve.
  void run_centralized_dense() override {
    Timer tm(*this);
    require(have_x0, "initial states not set");
    if (form_centralized_dense()) return;      // chip-wide launches (mas_dense_chip.hpp); below: the one-workgroup form
    if (d.N > kDenseMaxAgents) throw std::invalid_argument("the literal stacked solve is limited to " + std::to_string(kDenseMaxAgents) +
                                                          " agents (4 (N nx)^2 evaluations of the N-agent cost per time step); use run_centralized");
    be.zero(U, sizeof(double) * d.T * NU * d.Np);       // the stacked OCP starts from zero controls (multi_agent_problem.hpp:124, ocp.hpp:106-109)
    const size_t need = DenseCentralBody<M>::workspace_doubles(d.N, d.T);
    if (!dense_ws || dense_ws_size < need) { dense_ws = dalloc(need); dense_ws_size = need; dense_flags = ialloc(16); }
    DenseCentralBody<M> body;
    body.d = d; body.dt = dt; body.P = P; body.clamp = clamp;
    for (int i = 0; i < NU; ++i) { body.lb[i] = lb[i]; body.ub[i] = ub[i]; }
    body.tolerance = tolerance; body.max_iterations = max_iterations; body.max_ms = max_ms;
    body.x0 = x0; body.X = X; body.U = U; body.cost = cost; body.iters_total = iters_total; body.exit_code = exit_code; body.reg_events = reg_events;
    body.st = central; body.ws = dense_ws; body.flags = dense_flags;
    be.launch_group("central_dense", body, d.N, 0.0, 0, /*threads=*/1024);
    be.d2h(&host_central[0].st, central, sizeof(CentralState));
    const CentralState& hc = host_central[0].st;
    central_total = hc.cur_total; last_total_valid = true;
    stats.solves += 1; stats.agent_iterations += static_cast<long long>(hc.iterations) * d.N; stats.batch_iterations += hc.iterations;
    if (hc.done) stats.converged_solves += 1;
    controls_zero = false;
    initialized = false; rollout_consistent = false;     // cost[] now holds per-agent objectives of the stacked solution, merit[] is stale
  }


  void get_solution(double* Xo, double* Uo, double* co, int32_t* it, int32_t* ex, bool device_ptr) override {
    if (device_ptr) {                  // the layout conversion writes the caller's device buffers directly
      if (Xo) be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NX, X, Xo, d.T + 1}, d.N, d.T + 1);
      if (Uo) be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NU, U, Uo, d.T}, d.N, d.T);
    } else {
      if (Xo) { be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NX, X, stage, d.T + 1}, d.N, d.T + 1); download(Xo, static_cast<size_t>(d.N) * (d.T + 1) * NX, false); }
      if (Uo) { be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NU, U, stage, d.T}, d.N, d.T); download(Uo, static_cast<size_t>(d.N) * d.T * NU, false); }
    }
    if (co) { if (device_ptr) be.d2d(co, cost, sizeof(double) * d.N); else be.d2h(co, cost, sizeof(double) * d.N); }
    if (it) be.d2h(it, iters_total, sizeof(int) * d.N);
    if (ex) be.d2h(ex, exit_code, sizeof(int) * d.N);
    be.sync();
  }
  double total_cost() override { return last_total_valid ? central_total : host_sum(cost); }
  void get_trust_radii(double* out) override { require(radii != nullptr, "no trust radii"); be.d2h(out, radii, sizeof(double) * d.N); }

  void debug_backward(double* derivs, double* ko, double* Ko, int64_t* regs) override {
    ensure_initialized();
    be.template launch_1d<1>("begin_solve", BeginSolveBody{active, iters_solve, exit_code, alpha_sel}, d.N);
    { std::vector<long long> rg(d.N); be.d2h(rg.data(), reg_events, sizeof(long long) * d.N); for (auto v : rg) reg_base += v; }   // keep the running statistics
    be.zero(reg_events, sizeof(long long) * d.Np);
    const bool staged_fused = form_staged_backward();
    if (!staged_fused) {
      launch_sweep(0.0);
      launch_backward(0.0);
    }
    if (derivs) { be.template launch_2d<4>("tiles_to_host", TilesToHostBody{d, L::ND, D, stage}, d.N, d.T); download(derivs, static_cast<size_t>(d.N) * d.T * L::ND, false); }
    if (ko) { be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NU, kff, stage, d.T}, d.N, d.T); download(ko, static_cast<size_t>(d.N) * d.T * NU, false); }
    if (Ko) { be.template launch_2d<4>("soa_to_aos", SoaToAosBody{d, NU * NX, Kfb, stage, d.T}, d.N, d.T); download(Ko, static_cast<size_t>(d.N) * d.T * NU * NX, false); }
    if (regs) { std::vector<long long> h(d.N); be.d2h(h.data(), reg_events, sizeof(long long) * d.N); int64_t s = 0; for (auto v : h) s += v; *regs = s; }
    be.zero(reg_events, sizeof(long long) * d.Np);
  }

  HostStats get_stats() override {
    std::vector<long long> rg(d.N);
    be.d2h(rg.data(), reg_events, sizeof(long long) * d.N);
    HostStats s = stats; s.reg_events = reg_base;
    for (auto v : rg) s.reg_events += v;
    return s;
  }
  // in-kernel timing of the sweep -> kernel statistics entry "sweep_inkernel" (100-MHz REFCLK ticks)
  void fold_kernel_stamps() override { form_fold_kernel_stamps(); }
  void reset_stats() override {
    stats = HostStats{}; reg_base = 0; stamp_used = 0; stamp_launches.clear();
    be.zero(reg_events, sizeof(long long) * d.Np);
  }

 private:
  struct Timer {
    SolverImpl& s; std::chrono::steady_clock::time_point t0;
    explicit Timer(SolverImpl& s_) : s(s_), t0(std::chrono::steady_clock::now()) {}
    ~Timer() {
      try { s.be.sync(); } catch (...) {}
      s.stats.wall_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
  };

  void require(bool ok, const char* what) const { if (!ok) throw std::logic_error(what); }
  void ensure_initialized() { if (!initialized) initialize_problem(); }
  void ensure_snapshots() {
    if (X_old) return;
    X_old = dalloc((d.T + 1) * NX * d.Np); U_old = dalloc(static_cast<size_t>(d.T) * NU * d.Np); cost_old = dalloc(d.Np);
    U_cand = dalloc(static_cast<size_t>(d.T) * NU * d.Np);
  }
  void snapshot_old() {
    be.d2d(X_old, X, sizeof(double) * (d.T + 1) * NX * d.Np); be.d2d(U_old, U, sizeof(double) * d.T * NU * d.Np);
    be.d2d(cost_old, cost, sizeof(double) * d.Np);
  }
  bool timed_out(std::chrono::steady_clock::time_point t0) const {
    // ilqr.hpp:84-90: elapsed truncated to whole milliseconds, compared with max_ms.  All solves of a
    // batch start together, so one host clock stands for every agent's.
    if (std::isinf(max_ms) && max_ms > 0) return false;
    const double elapsed_ms = static_cast<double>(std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count());
    return elapsed_ms > max_ms;
  }

  // algorithmic bytes per agent and launch
  // sweep: SURVEY.md section 8(d)'s figure, read 8[(nx+nu)T + nx] + write 8[T*ND + nx + nx^2] (51,392 B at (4,2,100)); this
  // kernel itself moves 8(nx + nx + nx^2) bytes less (x_T and the terminal v_x, v_xx are handled by the backward kernel)
  double sweep_bytes() const { return 8.0 * ((NX + NU) * d.T + NX) + 8.0 * (d.T * L::ND + NX + NX * NX); }
  double backward_bytes() const { return 8.0 * d.T * L::ND + 8.0 * NX + 8.0 * d.T * (NU + NU * NX); }
  double forward_bytes() const { return 8.0 * (d.T * (NX + NU + NU + NU * NX) + NX) + 8.0 * ((d.T + 1) * NX + d.T * NU); }

  // linearisation sweep: on the GPU a dedicated kernel that stages each wavefront's tiles through
  // LDS (mas_kernels_hip.hpp); the host emulation replays the plain body
  // live_count != nullptr: walk the compacted list of active agents (live_list) instead of the active flags
  void launch_sweep(double bytes) { launch_sweep(bytes, 0, d.N); }
  // slots: with a live list, an upper bound of its length (the grid covers that many list slots)
  void launch_sweep(double bytes, int lo, int hi, const int* live_count = nullptr, int slots = 0, const int* live_list = nullptr) {
    if (form_sweep(bytes, lo, hi, live_count, slots, live_list)) return;
    be.template launch_2d_range<4>("sweep", SweepBody<M>{d, P, X, U, D, active}, lo, hi, d.T, bytes);
  }
  // line-search rollouts.  Returns true when the trial planes were written slot-indexed (mas_forward_quad_kernel), i.e. the
  // commit has to map slots back to agents.  One lane per rollout (ForwardBody) while the rollouts fill the chip; four lanes
  // per rollout once they no longer do -- the tail rounds of a solve, small batches, the centralised strategy's 256 agents --
  // where the launch time is one wavefront's chain and the quad form makes that chain ~40 % shorter.
  // live_count: device-side length of live_list (null: every agent, slot == agent); slots: host-side upper bound of it
  // staged_ok: the caller's select decides PER AGENT (SelectBody), so a model with kStagedLineSearch may roll out the first step size alone and
  // the others only where it failed; the strategies that sum trial costs over agents (centralised, line-search Nash) need every trial
  bool launch_forward(const ForwardBody<M>& fwd, double bytes, const int* live_list, const int* live_count, long long slots, bool staged_ok = false) {
    if (form_forward_by_slot(fwd, bytes, live_list, live_count, slots)) return true;
    if constexpr (M::kStagedLineSearch && !kConstrained) {
      if (staged_ok && switches.forward_staged) {
        be.template launch_2d_range<kForwardTrialsPerBlock>("forward", fwd, 0, d.N, 1, bytes / kNumAlphas);
        ForwardBody<M> rest = fwd;
        rest.publish = Publish{}; rest.j_base = 1; rest.stage_first = trial_merit; rest.stage_cur = merit;
        be.template launch_2d_range<kForwardTrialsPerBlock>("forward_further", rest, 0, d.N, kNumAlphas - 1, 0.0);
        return false;
      }
    }
    be.template launch_2d_range<kForwardTrialsPerBlock>("forward", fwd, 0, d.N, kNumAlphas, bytes);
    return false;
  }
  // 1600 slots x 10 trials x 4 lanes = 1000 wavefronts: the most the quad form can spread one per SIMD (1024 on MI355X);
  // beyond that its wavefronts double up and the one-lane form (640 wavefronts at 4096 agents) is faster again
  long long quad_forward_max_slots() const {
    return switches.forward_quad_max;
  }
  // chained trust-region epilogue as one kernel (mas_trust_epilogue_kernel): vehicle models, rounds with at most trust_fused_max live
  // agents (the grid is bounded by the live count: at most that many solves can have ended).  false: the three-kernel form runs.
  bool launch_trust_epilogue(const int* ended_count) { return form_trust_epilogue(ended_count); }
  // backward pass of the stacked problem: the 16-lane tile kernel with the rung agreement across all wavefronts
  // (LadderSync); constrained models and the host emulation keep the one-workgroup body
  void launch_central_backward(double bytes) {
    if (form_central_backward(bytes)) return;
    if (!Dsoa) Dsoa = dalloc(static_cast<size_t>(d.T) * L::ND * d.Np);
    be.template launch_2d<4>("tiles_to_soa", TilesToSoaBody{d, L::ND, D, Dsoa}, d.N, d.T);
    be.launch_group("central_backward", CentralBackwardBody<M>{d, P, X, D, kff, Kfb, reg_events, Dsoa, active}, d.N, bytes);
  }
  // backward Riccati pass: 16 lanes per agent over LDS tiles on the GPU, scalar body in the emulation
  void launch_backward(double bytes) { launch_backward(bytes, 0, d.N); }
  // live_list / live_count / slots: the round's compacted list of live agents, its device-side length and a host-side upper bound
  // of it (LDS-tile kernel only: LadderSync::live_list); null = agents lo .. hi-1 by their flags
  void launch_backward(double bytes, int lo, int hi, const int* live_list = nullptr, const int* live_count = nullptr, long long slots = 0) {
    if (form_backward(bytes, lo, hi, live_list, live_count, slots)) return;
    be.template launch_1d_range<1>("backward", BackwardBody<M>{d, P, X, D, kff, Kfb, reg_events, active, al_view(al_enabled)}, lo, hi, bytes);
  }


  ALView al_view(bool on) const {
    ALView v;
    if constexpr (kConstrained) { v.on = on; v.lam = al_lam; v.mu = al_mu; v.rho = al_rho; v.act_tol = inequality_activation_tolerance; }
    return v;
  }
  // a freshly constructed iLQR: no multipliers, penalty_parameter = 10 or the "penalty" key (ilqr.hpp:31,47-48)
  void reset_solver_state() {
    if constexpr (kConstrained) {
      if (al_lam) be.zero(al_lam, sizeof(double) * d.T * M::NEQ * d.Np);
      if (al_mu) be.zero(al_mu, sizeof(double) * d.T * M::NIN * d.Np);
      al_rho_dirty = true;
    }
  }
  void on_penalty_changed() override { al_rho_dirty = true; }
  void sync_penalty() { if constexpr (kConstrained) { if (al_rho_dirty) { fill(al_rho, penalty); al_rho_dirty = false; } } }

  ForwardBody<M> make_forward() const {
    ForwardBody<M> f;
    f.d = d; f.dt = dt; f.P = P; f.clamp = clamp;
    for (int i = 0; i < NU; ++i) { f.lb[i] = lb[i]; f.ub[i] = ub[i]; }
    f.x0 = x0; f.X = X; f.U = U; f.kff = kff; f.Kfb = Kfb; f.Xs = Xs; f.Us = Us; f.trial_merit = trial_merit; f.active = active;
    f.trial_obj = (kConstrained && al_enabled) ? trial_obj : nullptr; f.al = al_view(al_enabled);
    return f;
  }

  // N x iLQR::solve with per-agent acceptance and termination.
  //
  // One round = sweep -> backward -> forward -> select -> commit (+ a 16-byte read-back of "agents still
  // active").  On the GPU the host runs ONE ROUND AHEAD: round r+1 is enqueued before the read-back of round
  // r is awaited, so the device never idles on the host.  If round r turns out to have retired the last
  // agent, the already-enqueued round finds no active agent and every kernel exits at its first ballot --
  // results are unaffected.  (Agent ranges on separate streams were also tried: no gain at 4096 agents, the
  // round is bound by each agent's own sequential chain -- DESIGN.md section 4.)
  // n_solves > 1: every agent performs that many consecutive solves (SelectBody::begin_chain)
  // trust_epilogue: every solve of the chain is followed by the agent's trust-region test (run_trust_region)
  void solve_all(int n_solves = 1, bool trust_epilogue = false) {
    ensure_initialized();
    controls_zero = false;
    if (trust_epilogue && !trust_verdict) {
      trust_verdict = ialloc(d.Np); trust_ended = ialloc(d.Np); trust_roll_list = ialloc(d.Np); trust_roll_count = ialloc(16); trust_scale = dalloc(d.Np);
      trust_ended_list = ialloc(d.Np); trust_ended_counts = ialloc(16);
    }
    // x = integrate_horizon(x0, best_controls); cost = objective; merit (ilqr.hpp:75-78).  After
    // initialize_problem / a previous solve / a strategy epilogue the device already holds exactly that
    // rollout and cost (same kernels, same arithmetic, bit for bit), so it is only redone when the
    // controls were replaced from outside.
    if (kConstrained && al_enabled) { sync_penalty(); rollout_consistent = false; }   // merit depends on the current multipliers
    // begin_solve also sets current_merit = compute_merit(x,u) == objective (ilqr.hpp:78) when no rollout follows,
    // and zeroes the round counters
    // max_ms (ilqr.hpp:61,84-90): every agent's solve has its own clock, started on the device when the solve begins
    const bool timed = max_ms < std::numeric_limits<double>::infinity();
    if (timed && !solve_start) solve_start = static_cast<unsigned long long*>(raw_alloc(sizeof(unsigned long long) * d.Np));
    if (debug && !dbg_iter) { dbg_iter = ialloc(d.Np); dbg_improvement = dalloc(d.Np); }
    be.template launch_1d<1>("begin_solve", BeginSolveBody{active, iters_solve, exit_code, alpha_sel, cost, rollout_consistent ? merit : nullptr,
                                                           counters, 2 * kCounterRing, live_lists[0], live_count0, d.N,
                                                           solves_left, n_solves - 1, timed ? solve_start : nullptr, slot_of[0],
                                                           trust_epilogue ? trust_ended_counts : nullptr}, d.N);
    if (!rollout_consistent) {
      launch_rollout_cost();
      rollout_consistent = true;
    }
    stats.solves += static_cast<long long>(d.N) * n_solves;
    const long long round_limit = static_cast<long long>(max_iterations) * n_solves;   // every solve ends after max_iterations rounds at the latest
    if (!round_events_made) { for (auto& e : round_ev) e = be.make_event(); round_events_made = true; }
    if (debug) debug_initial();
    const ForwardBody<M> fwd = make_forward();
    if (form_fused_rounds(n_solves, timed, fwd, trust_epilogue)) return;      // two-kernel rounds (mas_fused_hip.hpp)
    // a solve that follows a one-round solve (the re-solves of outer rounds 2..max_outer usually are) would
    // waste its run-ahead round, so run ahead only while the previous solve needed several rounds
    const bool run_ahead = kHostRunsAhead && !debug && (n_solves > 1 || last_solve_rounds != 1);
    // the host polls the round's packed word (no event between rounds: an event record costs ~6 us of device idle time per round);
    // rounds whose counters change after the commit (multiplier update) or that are inspected afterwards (debug) keep the event
    const bool poll = kHostPolls && !debug && !(kConstrained && al_enabled) && d.N < (1 << 24);
    struct Marks { size_t p0, p1, s0, s1; };
    Marks marks[2];
    auto enqueue = [&](int r) {
      Marks& mk = marks[r & 1];
      mk.p0 = be.mark(); mk.s0 = stamp_launches.size();
      // byte counts are per agent here and scaled by the round's active-agent count once it is known
      // the sweep walks the agents the previous round's select left active (all of them in round 0); with the
      // augmented-Lagrangian update deciding termination after select, the list is not built and the flags are used
      const bool al_round = kConstrained && al_enabled;
      const int* live = al_round ? nullptr : (r == 0 ? live_count0 : reinterpret_cast<const int*>(counters + 2 * ((r - 1) % kCounterRing)));
      // grid: the live list only shrinks during a chain of solves, so the last count the host has read back bounds it from
      // above -- workgroups beyond the list would each be dispatched only to exit, and 1,500 of those cost a tail-round sweep
      // more than its few live workgroups do
      // two lists: round r walks list r & 1 (sweep, quad line search, commit) while its select compacts the next round's
      const int* list_now = live_lists[r & 1]; int* list_next = live_lists[(r + 1) & 1];
      launch_sweep(sweep_bytes(), 0, d.N, live, live ? static_cast<int>(live_upper) : d.N, list_now);
      launch_backward(backward_bytes(), 0, d.N, live ? list_now : nullptr, live, live_upper);
      const bool by_slot = live ? launch_forward(fwd, forward_bytes() * kNumAlphas, list_now, live, live_upper, true)
                                : launch_forward(fwd, forward_bytes() * kNumAlphas, nullptr, nullptr, std::numeric_limits<long long>::max(), true);
      long long* ctr = counters + 2 * (r % kCounterRing);
      long long* ctr_clear = counters + 2 * ((r + 2) % kCounterRing);     // last read two rounds ago
      SelectBody sel{d, tolerance, max_iterations, trial_merit, cost, merit, active, alpha_sel, iters_solve,
                     iters_total, exit_code, reinterpret_cast<int*>(ctr), ctr + 1, ctr_clear,
                     al_round ? trial_obj : nullptr, al_round ? improvement : nullptr, al_round ? nullptr : list_next,
                     n_solves > 1 ? solves_left : nullptr, timed ? solve_start : nullptr, max_ms,
                     debug ? dbg_iter : nullptr, debug ? dbg_improvement : nullptr, trust_epilogue ? trust_ended : nullptr};
      if (trust_epilogue) { sel.ended_list = trust_ended_list; sel.ended_count = trust_ended_counts + (r & 1); sel.ended_clear = trust_ended_counts + ((r + 1) & 1); }
      be.template launch_1d<1>("select", sel, d.N);
      // the round's counters reach the host through the commit kernel (Publish); the multiplier update of constrained models
      // still changes them after the commit, so those rounds keep the copy
      Publish pub = al_round ? Publish{} : Publish{ctr, host_counters + 2 * (r % kCounterRing), 2};
      if (poll) { pub = Publish{ctr, host_counters + 16 + (r % kCounterRing), 2}; pub.tag = round_tag(r); }   // one packed word, polled by the host
      if (by_slot) be.template launch_2d<4>("commit", CommitSlotBody<M>{d, Xs, Us, alpha_sel, X, U, list_now, live, pub}, static_cast<int>(live_upper), d.T + 1);
      else be.template launch_2d<4>("commit", CommitBody<M>{d, Xs, Us, alpha_sel, X, U, pub}, d.N, d.T + 1);
      if (trust_epilogue && launch_trust_epilogue(trust_ended_counts + (r & 1))) {
        // (one kernel: norm, re-roll by dependency levels, accept / revert and the copies, a workgroup per agent whose solve ended)
      } else if (trust_epilogue) {
        const TrustState ts{cost, merit, cost_old, radii, trust_verdict};
        be.template launch_1d<1>("trust_norm", TrustNormBody<M>{d, trust_ended, U, U_old, ts, trust_scale, trust_roll_list, trust_roll_count}, d.N);
        launch_trust_reroll(TrustRerollBody<M>{d, dt, P, x0, X, U, U_old, trust_scale, ts, trust_roll_list, trust_roll_count});
        be.template launch_2d<4>("trust_copy", TrustCopyBody<M>{d, trust_verdict, X, U, X_old, U_old, trust_roll_count}, d.N, d.T + 1);
      }
      if constexpr (kConstrained) {
        if (al_round)
          be.template launch_1d<1>("al_update", ALUpdateBody<M>{d, P, tolerance, max_iterations, ALParams{penalty_increase, constraint_tolerance, inequality_activation_tolerance},
                                                               X, U, improvement, al_view(true), active, iters_solve, exit_code,
                                                               reinterpret_cast<int*>(ctr), ctr + 1, timed ? solve_start : nullptr, max_ms}, d.N);
      }
      if (al_round) be.d2h_async(host_counters + 2 * (r % kCounterRing), ctr, sizeof(long long) * 2);
      if (!poll) be.record(round_ev[r & 1]);
      mk.p1 = be.mark(); mk.s1 = stamp_launches.size();
    };
    long long n_before = d.N;
    live_upper = d.N;
    int enq = 0, rounds_this_solve = 0;
    if (poll) clear_round_words();
    // the first check of a solve sees elapsed_ms == 0, so only a negative budget stops it before its first iteration
    if (max_iterations > 0 && !(timed && 0.0 > max_ms)) { enqueue(0); enq = 1; }
    else if (max_iterations > 0) be.template launch_1d<1>("mark_timeout", MarkTimeoutBody{active, exit_code}, d.N);
    for (int r = 0; r < enq; ++r) {
      if (run_ahead && enq == r + 1 && r + 1 < round_limit) { enqueue(r + 1); ++enq; }
      long long n_after, n_conv;
      if (poll) {                                    // (the word is stored by the commit kernel's first lane, right after select has counted)
        const long long w = await_round_word(r);
        n_after = w & Publish::kMask24; n_conv = (w >> Publish::kConvShift) & Publish::kMask24;
      } else {
        be.wait(round_ev[r & 1]);
        const long long* hc = host_counters + 2 * (r % kCounterRing);
        n_after = static_cast<long long>(reinterpret_cast<const int*>(hc)[0]); n_conv = hc[1];
      }
      stats.agent_iterations += n_before;           // every agent active at the start of a round executes one iLQR iteration
      stats.converged_solves += n_conv;
      stats.batch_iterations++; ++rounds_this_solve;
      if (debug) debug_round();
      be.scale_bytes(marks[r & 1].p0, marks[r & 1].p1, static_cast<double>(n_before));
      for (size_t q = marks[r & 1].s0; q < marks[r & 1].s1; ++q) stamp_launches[q].bytes *= static_cast<double>(n_before);
      n_before = n_after;
      live_upper = std::max<long long>(n_after, 1);
      if (n_after == 0) {                            // a round enqueued ahead runs on zero agents: nothing to account
        if (enq > r + 1) { be.scale_bytes(marks[(r + 1) & 1].p0, marks[(r + 1) & 1].p1, 0.0); for (size_t q = marks[(r + 1) & 1].s0; q < marks[(r + 1) & 1].s1; ++q) stamp_launches[q].bytes = 0.0; }
        break;
      }
      if (enq == r + 1 && r + 1 < round_limit) { enqueue(r + 1); ++enq; }
    }
    if (poll) { if (enq > rounds_this_solve) be.sync(); }   // a round that ran ahead still publishes its word: drain it before the words are reused
    else if (enq > 0) be.wait(round_ev[(enq - 1) & 1]);   // drain a round that ran ahead
    last_solve_rounds = rounds_this_solve;
  }

  // the packed word of round r (Publish::pack) in the pinned ring at host_counters[16 ..]: polled, not waited for with an event
  // 15-bit tag of round r, never 0 (0 is the cleared word, and Publish takes tag 0 for "no packed word"): 1 .. 0x7FFF, round after round
  static int round_tag(long long r) { return 1 + static_cast<int>(r % 0x7FFF); }
  long long await_round_word(long long r) {
    volatile long long* words = host_counters + 16;
    const long long want = round_tag(r);
    const auto t0 = std::chrono::steady_clock::now();
    for (long long spins = 1;; ++spins) {
      const long long w = words[r % kCounterRing];
      if (((w >> Publish::kTagShift) & 0x7FFF) == want) return w;
      if ((spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
        try { be.sync(); } catch (...) {}            // nothing of this call may still publish when a later call clears and reuses the words
        throw std::runtime_error("the device did not publish a round's counters within 30 s");
      }
    }
  }
  void clear_round_words() const { volatile long long* words = host_counters + 16; for (int q = 0; q < kCounterRing; ++q) words[q] = 0; }


  // "debug" key (ilqr.hpp:45,79-80,262-267): the reference prints one line per solve start and one per iteration of
  // every agent (interleaved arbitrarily across OpenMP threads); here the same lines, agent by agent after each round,
  // prefixed with the agent's position in the batch
  void debug_initial() {
    std::vector<double> c(d.N), m(d.N);
    be.d2h(c.data(), cost, sizeof(double) * d.N); be.d2h(m.data(), merit, sizeof(double) * d.N);
    for (int a = 0; a < d.N; ++a) std::cout << "[agent " << a << "] iLQR initial cost=" << c[a] << " merit=" << m[a] << '\n';
    std::cout.flush();
  }
  void debug_round() {
    std::vector<double> c(d.N), m(d.N), imp(d.N); std::vector<int> it(d.N);
    be.d2h(c.data(), cost, sizeof(double) * d.N); be.d2h(m.data(), merit, sizeof(double) * d.N);
    be.d2h(imp.data(), dbg_improvement, sizeof(double) * d.N); be.d2h(it.data(), dbg_iter, sizeof(int) * d.N);
    for (int a = 0; a < d.N; ++a)
      if (it[a] > 0) std::cout << "[agent " << a << "] iLQR iter " << (it[a] - 1) << ": cost=" << c[a] << " merit=" << m[a] << " d_merit=" << imp[a]
                               << " eq_violation=0 ineq_violation=0\n";
    std::cout.flush();
  }

  // detail::compute_total_cost over ALL agents (nash.hpp:45-49): through the caller's exchange when the block
  // list is sharded over ranks, the local serial sum otherwise
  double joint_sum(const double* dev) {
    if (!cost_exchange) return host_sum(dev);
    std::vector<double> h(d.N); be.d2h(h.data(), dev, sizeof(double) * d.N);
    double total = 0.0;
    if (cost_exchange(cost_exchange_ctx, h.data(), static_cast<int32_t>(d.N), &total) != 0) throw std::runtime_error("cost exchange callback failed");
    return total;
  }
  double host_sum(const double* dev) {      // Solution.total_cost: serial sum in block order (nash.hpp:23-37)
    std::vector<double> h(d.N); be.d2h(h.data(), dev, sizeof(double) * d.N);
    double c = 0.0; for (double v : h) c += v; return c;
  }
  void fill(double* dev, double v) { std::vector<double> h(d.Np, v); be.h2d(dev, h.data(), sizeof(double) * d.Np); }
  void fill_int(int* dev, int v) { be.fill32(dev, v, d.Np); }       // on the stream, no host round trip
  // device pointers: the copy runs on the handle's stream and is complete on return, so the caller's buffer is no longer
  // referenced afterwards; the caller must have made its contents ready before the call (mas_hip.h, stream-ordering contract)
  void upload(const double* src, size_t n, bool device_ptr) { if (device_ptr) { be.d2d(stage, src, sizeof(double) * n); be.sync(); } else be.h2d(stage, src, sizeof(double) * n); }
  void download(double* dst, size_t n, bool device_ptr) { if (device_ptr) { be.d2d(dst, stage, sizeof(double) * n); be.sync(); } else be.d2h(dst, stage, sizeof(double) * n); }
  void* raw_alloc(size_t bytes) { void* p = be.alloc(bytes); allocations.push_back(p); return p; }   // freed by the destructor
  // device address of the "gave up waiting" flag of the rung agreement (null while the single-workgroup fallback is in use)
  const int* ladder_error_flag() const { return form_ladder_error_flag(); }
  double* dalloc(size_t n) { void* p = be.alloc(sizeof(double) * n); allocations.push_back(p); return static_cast<double*>(p); }
  int* ialloc(size_t n) { void* p = be.alloc(sizeof(int) * n); allocations.push_back(p); return static_cast<int*>(p); }

  // ---- the kernel forms of this build: form_*() launches a specialised kernel and returns true, or returns false and the generic body above runs
#include MAS_SOLVER_FORMS

  Dims d{}; double dt = 0.0; ModelParams P{}; bool clamp = false; double lb[NU], ub[NU];
  double *x0 = nullptr, *X = nullptr, *U = nullptr, *D = nullptr, *kff = nullptr, *Kfb = nullptr, *Xs = nullptr, *Us = nullptr;
  double *trial_merit = nullptr, *cost = nullptr, *merit = nullptr, *x0_saved = nullptr, *U_saved = nullptr, *stage = nullptr;
  double *Dsoa = nullptr;
  double *X_saved = nullptr, *cost_saved = nullptr; bool saved_rollout = false;   // rollout and cost of the saved initial controls
  double *dense_ws = nullptr; size_t dense_ws_size = 0; int *dense_flags = nullptr;   // literal stacked solve (mas_dense.hpp)
  double *al_lam = nullptr, *al_mu = nullptr, *al_rho = nullptr, *improvement = nullptr, *trial_obj = nullptr;
  bool al_enabled = true, al_rho_dirty = false;
  double *X_old = nullptr, *U_old = nullptr, *cost_old = nullptr, *U_cand = nullptr, *radii = nullptr;
  JointState *joint = nullptr, *host_joint = nullptr;     // Nash line search decided on the device; its pinned read-back copy
  Comm* comm_exchange = nullptr;                          // RCCL communicator of the sharded block list (owned by the caller)
  double* comm_costs = nullptr; int comm_total = 0;       // [comm_total] every rank's costs in global block order
  int *slot_of[2] = {nullptr, nullptr}, *fused_error = nullptr;   // fused rounds: every agent's position in live_lists[k]; raised when a bounded LDS wait gave up
  int *live_lists[2] = {nullptr, nullptr}, *live_count0 = nullptr;   // agents still active, compacted by select (round r reads list r & 1, writes the other); count of round 0 (= N)
  int *solves_left = nullptr;                         // chained solves an agent still has to run (run_sequential)
  int *trust_verdict = nullptr, *trust_ended = nullptr, *trust_roll_list = nullptr, *trust_roll_count = nullptr;   // chained trust-region epilogue
  double *trust_scale = nullptr;
  int *trust_ended_list = nullptr, *trust_ended_counts = nullptr;   // agents whose solve ended in the round, compacted by select; two counters alternating by round
  unsigned long long* solve_start = nullptr;          // tick at which each agent's current solve began (finite max_ms only)
  int* dbg_iter = nullptr; double* dbg_improvement = nullptr;   // per-agent trace of the last round ("debug")
  unsigned long long* ladder_sync = nullptr;   // rung agreement of the centralised backward pass (LadderSync)
  int *active = nullptr, *alpha_sel = nullptr, *iters_solve = nullptr, *iters_total = nullptr, *exit_code = nullptr;
  long long *reg_events = nullptr, *counters = nullptr, *host_counters = nullptr;
  struct HostCentral { CentralState st; long long ladder_error; };     // what the commit kernel publishes per iteration of the stacked solve
  CentralState* central = nullptr; HostCentral* host_central = nullptr;
  std::vector<void*> allocations;
  bool have_x0 = false, initialized = false, saved = false, last_total_valid = false, rollout_consistent = false;
  bool controls_zero = false, saved_controls_zero = false;     // U is known to be all zeros (as set, never changed by a solve since)
  double central_total = 0.0;
  long long reg_base = 0;
  static constexpr int kMaxStampedLaunches = 1024;
  struct StampedLaunch { size_t off, count; double bytes; };
  unsigned long long* sweep_stamps = nullptr;
  size_t stamp_capacity = 0, stamp_used = 0;
  std::vector<StampedLaunch> stamp_launches;
  static constexpr int kCounterRing = 4;
  typename Backend::Event round_ev[2]{};
  bool round_events_made = false;
  int last_solve_rounds = 0;
  long long live_upper = 0;        // upper bound of the number of live agents in the round being enqueued
  HostStats stats;
};

// The models are instantiated in TWO translation units that differ in ONE compiler option, the machine scheduler's strategy:
//   mas_capi.hip        default strategy     single-track family (C2, C3), pendulum
//   mas_models_ilp.hip  -mllvm -amdgpu-sched-strategy=max-ilp     identity LQR (C4), bicycle6 (C5)
// Every kernel here runs about one wavefront per SIMD, where hiding latency inside the wavefront is all there is; the max-ILP
// strategy does that better for the LDS-tile backward kernel and the six-state rollouts (C5 56.6 -> 54.1 ms, C4 18.5 -> 18.1 ms)
// and worse for the single-track kernels (C3 6.04 -> 6.24 ms), results identical bit for bit (tools/exp/ab_variants.py).
std::unique_ptr<SolverBase> make_solver_ilp(const ProblemDesc& desc, void* external_stream);      // mas_models_ilp.hip: models 2, 5
#if defined(MAS_TU_MODELS_ILP)
std::unique_ptr<SolverBase> make_solver_ilp(const ProblemDesc& desc, void* external_stream) {
  switch (desc.model_id) {
    case 2: return std::make_unique<SolverImpl<LqrIdentity>>(desc, external_stream);
    case 5: return std::make_unique<SolverImpl<Bicycle6>>(desc, external_stream);
    default: return nullptr;
  }
}
#else
// workloads registered at run time (mas_hip_register_model): a model struct compiled against these very templates into a plugin
// library whose factory is entered here; ids start at kFirstRegisteredModel
constexpr int kFirstRegisteredModel = 100;
struct RegisteredModel { int nx = 0, nu = 0; SolverBase* (*make)(const ProblemDesc*, void*) = nullptr; void* library = nullptr; std::string name; };
constexpr int kMaxRegisteredModels = 256;
// a fixed table filled front to back: an entry is complete before `count` (release) makes it visible, readers (acquire) never see a
// table that moves -- handles may be created on other threads while a model is being registered
struct ModelRegistry { RegisteredModel models[kMaxRegisteredModels]; std::atomic<int> count{0}; };
inline ModelRegistry& model_registry() { static ModelRegistry r; return r; }
inline bool registered_model(int model_id, RegisteredModel* out) {
  ModelRegistry& r = model_registry();
  const int k = model_id - kFirstRegisteredModel;
  if (k < 0 || k >= r.count.load(std::memory_order_acquire)) return false;
  *out = r.models[k];
  return true;
}
inline int register_model_entry(const RegisteredModel& rm) {      // (callers serialise: mas_hip_register_model holds its mutex)
  ModelRegistry& r = model_registry();
  const int k = r.count.load(std::memory_order_relaxed);
  if (k >= kMaxRegisteredModels) return -1;
  r.models[k] = rm;
  r.count.store(k + 1, std::memory_order_release);
  return kFirstRegisteredModel + k;
}
inline std::unique_ptr<SolverBase> make_solver(const ProblemDesc& desc, void* external_stream) {
  { RegisteredModel rm; if (registered_model(desc.model_id, &rm)) return std::unique_ptr<SolverBase>(rm.make(&desc, external_stream)); }
  switch (desc.model_id) {
    case 1: return std::make_unique<SolverImpl<SingleTrack>>(desc, external_stream);
    case 2: return make_solver_ilp(desc, external_stream);
    case 3: return std::make_unique<SolverImpl<Pendulum>>(desc, external_stream);
    case 4: return std::make_unique<SolverImpl<SingleTrackLane>>(desc, external_stream);
    case 5: return make_solver_ilp(desc, external_stream);
    case 6: return std::make_unique<SolverImpl<SingleTrackLimited>>(desc, external_stream);
    default: return nullptr;
  }
}
#endif

inline bool model_dims(int model_id, int* nx, int* nu) {
  int a = 0, b = 0;
#if !defined(MAS_TU_MODELS_ILP)
  { RegisteredModel rm; if (registered_model(model_id, &rm)) { if (nx) *nx = rm.nx; if (nu) *nu = rm.nu; return true; } }
#endif
  switch (model_id) {
    case 1: a = SingleTrack::NX; b = SingleTrack::NU; break;
    case 2: a = LqrIdentity::NX; b = LqrIdentity::NU; break;
    case 3: a = Pendulum::NX; b = Pendulum::NU; break;
    case 4: a = SingleTrackLane::NX; b = SingleTrackLane::NU; break;
    case 5: a = Bicycle6::NX; b = Bicycle6::NU; break;
    case 6: a = SingleTrackLimited::NX; b = SingleTrackLimited::NU; break;
    default: return false;
  }
  if (nx) *nx = a;
  if (nu) *nu = b;
  return true;
}

}  // namespace mas
