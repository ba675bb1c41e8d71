// mas_amd.hpp -- C++ host surface of the MI355X iLQR path, mirroring the reference's API for
// `--solver ilqr` so that code written against multi_agent_solver reads the same:
//
//   reference (include/multi_agent_solver/...)             here (namespace mas_amd)
//   types.hpp:14-57      State/Control/trajectories         Vector / Matrix (column-major, col(t) contiguous)
//   ocp.hpp:30-237       struct OCP                         struct OCP  (callbacks -> ModelDescriptor)
//   solvers/ilqr.hpp     class iLQR {set_params, solve}     class iLQR  (one-agent batch on the GPU)
//   solvers/solver.hpp   Solver variant, solve, set_params  same names
//   agent.hpp, solution.hpp, multi_agent_problem.hpp        Agent, AgentPtr, Solution, MultiAgentProblem
//   strategies/*.hpp     Centralized / SequentialNash / LineSearchNash / TrustRegionNash, Strategy, solve
//
// The reference's std::function hooks (types.hpp:21-50) cannot run on a GPU, so an OCP names a
// REGISTERED model (include/mas_hip.h, enum mas_hip_model) plus its constants instead.  Everything
// numerical happens behind the C ABI of libmas_hip.so; this header only moves data and keeps the
// reference's call protocol (who owns what, what solve() reads and writes, which errors are thrown).
// Header-only, C++17, no Eigen.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <limits>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <variant>
#include <vector>

#include "../../../include/mas_hip.h"

namespace mas_amd {

// ---------------------------------------------------------------------------- types ------
using Vector = std::vector<double>;
using State = Vector;
using Control = Vector;

// column-major dense matrix; a trajectory is rows x (T+1) with column t contiguous (types.hpp:14-19)
struct Matrix {
  int rows_ = 0, cols_ = 0;
  std::vector<double> data;
  Matrix() = default;
  Matrix(int r, int c, double v = 0.0) : rows_(r), cols_(c), data(static_cast<size_t>(r) * c, v) {}
  static Matrix Zero(int r, int c) { return Matrix(r, c, 0.0); }
  int rows() const { return rows_; }
  int cols() const { return cols_; }
  size_t size() const { return data.size(); }
  double& operator()(int r, int c) { return data[static_cast<size_t>(c) * rows_ + r]; }
  double operator()(int r, int c) const { return data[static_cast<size_t>(c) * rows_ + r]; }
  double* col(int c) { return data.data() + static_cast<size_t>(c) * rows_; }
  const double* col(int c) const { return data.data() + static_cast<size_t>(c) * rows_; }
};
using StateTrajectory = Matrix;
using ControlTrajectory = Matrix;
using SolverParams = std::unordered_map<std::string, double>;   // types.hpp:57

// replaces dynamics / stage_cost / terminal_cost and the optional derivative hooks of the reference OCP
struct ModelDescriptor {
  int model_id = 0;                       // enum mas_hip_model
  std::optional<std::vector<double>> params;   // 16 model constants; nullopt = the example's
  bool operator==(const ModelDescriptor& o) const { return model_id == o.model_id && params == o.params; }
};

class HipError : public std::runtime_error {
 public:
  HipError(int code, const std::string& what) : std::runtime_error(what), code_(code) {}
  int code() const { return code_; }
 private:
  int code_;
};

// A workload of the caller's own.  The reference takes any std::function callbacks (types.hpp:21-50); device code cannot, so the caller
// hands over the SOURCE of a struct with the same callbacks (mas::FdModel<Self, NX, NU>: dynamics, stage_cost, terminal_cost and whatever
// optional hooks it declares, include/mas_hip.h) and gets the descriptor to put into OCP::model.  The first registration of a workload
// compiles it against the kernel templates (about a minute of hipcc); later ones -- every further start of the program -- load the cached plugin.
inline ModelDescriptor register_model(const std::string& struct_source, const std::string& struct_name,
                                      std::optional<std::vector<double>> params = std::nullopt) {
  int id = 0;
  const int rc = mas_hip_register_model(struct_source.c_str(), struct_name.c_str(), nullptr, &id);
  if (rc != MAS_HIP_OK) throw HipError(rc, std::string("mas_hip_register_model: ") + mas_hip_last_error(nullptr));
  ModelDescriptor m;
  m.model_id = id; m.params = std::move(params);
  return m;
}

namespace detail {

// one mas_hip_solver handle for a homogeneous batch of agents
class Batch {
 public:
  Batch(const ModelDescriptor& model, int n_agents, int horizon, double dt, const std::optional<Control>& lb,
        const std::optional<Control>& ub, int device = 0) {
    mas_hip_problem_desc d{};
    d.model_id = model.model_id; d.n_agents = n_agents; d.horizon_steps = horizon; d.device = device; d.dt = dt;
    std::vector<double> p;
    if (model.params) { p = *model.params; p.resize(16, 0.0); d.params = p.data(); }
    if (lb && ub) { d.input_lower_bounds = lb->data(); d.input_upper_bounds = ub->data(); }
    else d.no_input_bounds = 1;             // clamping needs both bounds (ilqr.hpp:213)
    const int rc = mas_hip_create(&d, &h_);
    if (rc != MAS_HIP_OK) throw HipError(rc, std::string("mas_hip_create: ") + mas_hip_last_error(nullptr));
  }
  ~Batch() { if (h_) mas_hip_destroy(h_); }
  Batch(const Batch&) = delete;
  Batch& operator=(const Batch&) = delete;
  void check(int rc, const char* what) const { if (rc != MAS_HIP_OK) throw HipError(rc, std::string(what) + ": " + mas_hip_last_error(h_)); }
  void set_params(const SolverParams& params) {
    for (const auto& kv : params) check(mas_hip_set_param(h_, kv.first.c_str(), kv.second), "mas_hip_set_param");
  }
  mas_hip_solver* get() const { return h_; }
 private:
  mas_hip_solver* h_ = nullptr;
};

// What a device handle is made for: everything mas_hip_create takes.  Two OCPs with equal keys can use one handle
// (only x0, the controls and the solver parameters differ between calls).
struct BatchKey {
  ModelDescriptor model; int n_agents = 0, horizon = 0; double dt = 0.0;
  std::optional<Control> lb, ub;
  bool operator==(const BatchKey& o) const {
    return model == o.model && n_agents == o.n_agents && horizon == o.horizon && dt == o.dt && lb == o.lb && ub == o.ub;
  }
};

// One-entry handle cache.  A `mas_hip_solver` owns ~30 device allocations and a stream; creating and destroying one per
// solve(OCP&) made the N = 1 drop-in path (reference usage: nash.hpp:59-64 calls solve() on the same solver object once
// per outer iteration; examples/pendulum_swing_up.cpp:158) pay ~1 ms of hipMalloc / hipFree per call.  The cache keeps
// the last handle and reuses it while the key matches.  MAS_AMD_NO_HANDLE_CACHE=1 restores create-per-call (used to
// measure the difference).
class BatchCache {
 public:
  Batch& get(const BatchKey& key) {
    static const bool disabled = [] { const char* e = std::getenv("MAS_AMD_NO_HANDLE_CACHE"); return e && e[0] == '1'; }();
    if (disabled || !batch_ || !(key_ == key)) {
      batch_.reset();                                // free the old device memory before allocating the new handle
      batch_ = std::make_shared<Batch>(key.model, key.n_agents, key.horizon, key.dt, key.lb, key.ub);
      key_ = key;
    }
    return *batch_;
  }
  void clear() { batch_.reset(); }
 private:
  std::shared_ptr<Batch> batch_;
  BatchKey key_;
};

}  // namespace detail

// ------------------------------------------------------------------------------ OCP ------
// Mirrors struct OCP (ocp.hpp:30-100).  solve() reads initial_state, best_controls (warm start), dt,
// dims, input bounds; writes best_states, best_controls, best_cost; never touches initial_* (SURVEY 8(b)).
struct OCP {
  StateTrajectory initial_states;
  ControlTrajectory initial_controls;
  StateTrajectory best_states;
  ControlTrajectory best_controls;
  double best_cost = std::numeric_limits<double>::max();

  State initial_state;
  ModelDescriptor model;

  int control_dim = 0;
  int state_dim = 0;
  int horizon_steps = 0;
  double dt = 0.0;

  std::optional<Control> input_lower_bounds = std::nullopt;
  std::optional<Control> input_upper_bounds = std::nullopt;

  size_t id = 0;
  bool initialized = false;

  // ocp.hpp:102-183: size the guess, roll it out (RK4), set best_* and best_cost.  One-agent GPU call;
  // MultiAgentProblem::initialize_agents() does the same for all agents in one batch.
  void initialize_problem() {
    if (initial_controls.rows() != control_dim || initial_controls.cols() != horizon_steps)
      initial_controls = ControlTrajectory::Zero(control_dim, horizon_steps);
    // (called once per OCP: a handle of its own, released on return; repeated solves reuse iLQR's cached handle)
    detail::Batch b(model, 1, horizon_steps, dt, input_lower_bounds, input_upper_bounds);
    b.check(mas_hip_set_initial_states_host(b.get(), initial_state.data()), "set_initial_states");
    b.check(mas_hip_set_controls_host(b.get(), initial_controls.data.data()), "set_controls");
    b.check(mas_hip_initialize_problem(b.get()), "initialize_problem");
    initial_states = StateTrajectory(state_dim, horizon_steps + 1);
    b.check(mas_hip_get_solution_host(b.get(), initial_states.data.data(), nullptr, &best_cost, nullptr, nullptr), "get_solution");
    best_states = initial_states;
    best_controls = initial_controls;
    initialized = true;
  }
  void reset() {                                   // ocp.hpp:83-93
    initial_controls = ControlTrajectory::Zero(control_dim, horizon_steps);
    initialize_problem();
  }
  void update_initial_with_best() {                // ocp.hpp:95-100
    initial_controls = best_controls;
    initial_states = best_states;
  }
  bool verify_problem() const {                    // ocp.hpp:187-236 (asserts in the reference)
    int nx = 0, nu = 0;
    return state_dim != 0 && control_dim != 0 && horizon_steps != 0 && dt != 0.0 &&
           static_cast<int>(initial_state.size()) == state_dim && mas_hip_model_dims(model.model_id, &nx, &nu) == MAS_HIP_OK &&
           nx == state_dim && nu == control_dim &&
           (!input_lower_bounds || static_cast<int>(input_lower_bounds->size()) == control_dim) &&
           (!input_upper_bounds || static_cast<int>(input_upper_bounds->size()) == control_dim);
  }
};

// ----------------------------------------------------------------------------- iLQR ------
// class iLQR (solvers/ilqr.hpp:23-55): default-constructible, copyable, set_params + solve.
class iLQR {
 public:
  iLQR() = default;                                // defaults 50 / 1e-6 / inf live in the library (ilqr.hpp:26-37)
  // copies are independent solver objects (the reference stores them in std::vector<Solver>, nash.hpp:78-83, and calls
  // them from different OpenMP threads): a copy starts without a device handle of its own, it never shares one
  iLQR(const iLQR& o) : last_iterations(o.last_iterations), last_exit(o.last_exit), params_(o.params_) {}
  iLQR& operator=(const iLQR& o) { if (this != &o) { params_ = o.params_; last_iterations = o.last_iterations; last_exit = o.last_exit; cache_.clear(); pushed_to_ = nullptr; } return *this; }
  iLQR(iLQR&&) = default;
  iLQR& operator=(iLQR&&) = default;
  void set_params(const SolverParams& params) {
    params.at("max_iterations");                   // the reference uses params.at(): std::out_of_range if absent
    params.at("tolerance");
    params.at("max_ms");
    params_ = params;
  }
  const SolverParams& params() const { return params_; }

  // ilqr.hpp:58-273 on a one-agent batch.  The device handle is kept between calls (resize_buffers of the reference
  // likewise only reallocates when the problem's shape changes, ilqr.hpp:277-377); like the reference's solver object it
  // also keeps the augmented-Lagrangian multipliers and penalty across calls (ilqr.hpp:415-421).
  void solve(OCP& problem) {
    detail::Batch& b = cache_.get({problem.model, 1, problem.horizon_steps, problem.dt, problem.input_lower_bounds, problem.input_upper_bounds});
    // parameters go to the handle when it is new or they changed, not on every call: the reference sets penalty_parameter in
    // set_params ONCE and lets it grow across solves (ilqr.hpp:47-48,256-259); re-sending "penalty" would reset it
    if (pushed_to_ != b.get() || !(pushed_params_ == params_)) { b.set_params(params_); pushed_to_ = b.get(); pushed_params_ = params_; }
    if (problem.best_controls.rows() != problem.control_dim || problem.best_controls.cols() != problem.horizon_steps)
      problem.best_controls = ControlTrajectory::Zero(problem.control_dim, problem.horizon_steps);
    b.check(mas_hip_set_initial_states_host(b.get(), problem.initial_state.data()), "set_initial_states");
    b.check(mas_hip_set_controls_host(b.get(), problem.best_controls.data.data()), "set_controls");
    b.check(mas_hip_solve(b.get()), "solve");
    problem.best_states = StateTrajectory(problem.state_dim, problem.horizon_steps + 1);
    int32_t it = 0, ex = 0;
    b.check(mas_hip_get_solution_host(b.get(), problem.best_states.data.data(), problem.best_controls.data.data(), &problem.best_cost, &it, &ex),
            "get_solution");
    last_iterations = it; last_exit = ex;
  }
  int last_iterations = 0, last_exit = 0;          // extension: the reference exposes neither
 private:
  SolverParams params_, pushed_params_;
  const void* pushed_to_ = nullptr;               // the handle `pushed_params_` were last sent to
  detail::BatchCache cache_;
};

using Solver = std::variant<iLQR>;                 // solvers/solver.hpp:17-22 (cgd / osqp are out of scope)
inline void solve(Solver& solver, OCP& problem) { std::visit([&](auto& s) { s.solve(problem); }, solver); }
inline void set_params(Solver& solver, const SolverParams& params) { std::visit([&](auto& s) { s.set_params(params); }, solver); }
template <typename SolverT>
std::shared_ptr<Solver> create() { return std::make_shared<Solver>(std::in_place_type<SolverT>); }

// ------------------------------------------------------- agents, problem, solution ------
struct Agent {                                     // agent.hpp:9-42
  std::size_t id;
  std::shared_ptr<OCP> ocp;
  Agent(std::size_t id_, std::shared_ptr<OCP> ocp_) : id(id_), ocp(std::move(ocp_)) {}
  int state_dim() const { return ocp->state_dim; }
  int control_dim() const { return ocp->control_dim; }
  void reset() { ocp->reset(); }
  void update_initial_with_best() { ocp->update_initial_with_best(); }
};
using AgentPtr = std::shared_ptr<Agent>;

struct Solution {                                  // solution.hpp:9-15
  std::vector<StateTrajectory> states;
  std::vector<ControlTrajectory> controls;
  std::vector<double> costs;
  double total_cost = 0.0;
  std::vector<int> iterations;                     // extension: iLQR iterations per agent
};

struct AgentBlockInfo {                            // multi_agent_problem.hpp:14-22
  std::size_t agent_id;
  int state_offset, control_offset, state_dim, control_dim;
  AgentPtr agent;
};

class MultiAgentProblem {                          // multi_agent_problem.hpp:24-128
 public:
  std::vector<AgentPtr> agents;
  std::vector<AgentBlockInfo> blocks;
  void add_agent(const AgentPtr& a) { agents.push_back(a); }
  void compute_offsets() {                         // :37-50, blocks sorted by agent id
    blocks.clear();
    std::vector<AgentPtr> sorted = agents;
    std::sort(sorted.begin(), sorted.end(), [](const AgentPtr& a, const AgentPtr& b) { return a->id < b->id; });
    int s_off = 0, u_off = 0;
    for (auto& a : sorted) {
      blocks.push_back({a->id, s_off, u_off, a->state_dim(), a->control_dim(), a});
      s_off += a->state_dim(); u_off += a->control_dim();
    }
  }
};

namespace detail {

// The strategies that couple agents (line search, centralised) batch the id-sorted block list into ONE
// handle; that needs agents of one model, horizon, step and bound set (true for every reference example).
inline bool same_batch(const OCP& a, const OCP& b);
inline void require_homogeneous(const MultiAgentProblem& p) {
  if (p.blocks.empty()) return;
  const OCP& f = *p.blocks.front().agent->ocp;
  for (const auto& b : p.blocks)
    if (!same_batch(*b.agent->ocp, f))
      throw std::invalid_argument("this strategy couples all agents: the HIP iLQR path needs agents of one model / horizon / dt / bounds");
}

// one homogeneous group of blocks (indices into problem.blocks) resident in one handle
struct BatchedProblem {
  std::unique_ptr<Batch> batch;
  std::vector<int> members;
  int N = 0, T = 0, nx = 0, nu = 0;
};

inline bool same_batch(const OCP& a, const OCP& b) {
  return a.model == b.model && a.horizon_steps == b.horizon_steps && a.dt == b.dt && a.state_dim == b.state_dim &&
         a.control_dim == b.control_dim && a.input_lower_bounds == b.input_lower_bounds && a.input_upper_bounds == b.input_upper_bounds;
}

// Agents of one model / horizon / dt / bound set share a handle.  The per-agent strategies (sequential,
// trust-region) treat agents independently, so a mixed problem is simply split into several batches; the
// strategies that couple agents through the total cost (line search) or a stacked solve (centralised) need
// a single batch.
inline std::vector<std::vector<int>> group_blocks(const MultiAgentProblem& p) {
  std::vector<std::vector<int>> groups;
  for (int i = 0; i < static_cast<int>(p.blocks.size()); ++i) {
    bool placed = false;
    for (auto& g : groups)
      if (same_batch(*p.blocks[g.front()].agent->ocp, *p.blocks[i].agent->ocp)) { g.push_back(i); placed = true; break; }
    if (!placed) groups.push_back({i});
  }
  return groups;
}

inline BatchedProblem upload(MultiAgentProblem& problem, const std::vector<int>& members, const SolverParams& params, bool warm_from_best, int device = 0) {
  const OCP& f = *problem.blocks[members.front()].agent->ocp;
  BatchedProblem bp;
  bp.members = members;
  bp.N = static_cast<int>(members.size()); bp.T = f.horizon_steps; bp.nx = f.state_dim; bp.nu = f.control_dim;
  bp.batch = std::make_unique<Batch>(f.model, bp.N, bp.T, f.dt, f.input_lower_bounds, f.input_upper_bounds, device);
  bp.batch->set_params(params);
  std::vector<double> x0(static_cast<size_t>(bp.N) * bp.nx), U(static_cast<size_t>(bp.N) * bp.T * bp.nu, 0.0);
  for (int i = 0; i < bp.N; ++i) {
    const OCP& o = *problem.blocks[members[i]].agent->ocp;
    std::copy(o.initial_state.begin(), o.initial_state.end(), x0.begin() + static_cast<size_t>(i) * bp.nx);
    const ControlTrajectory& src = warm_from_best ? o.best_controls : o.initial_controls;
    if (src.rows() == bp.nu && src.cols() == bp.T) std::copy(src.data.begin(), src.data.end(), U.begin() + static_cast<size_t>(i) * bp.T * bp.nu);
  }
  bp.batch->check(mas_hip_set_initial_states_host(bp.batch->get(), x0.data()), "set_initial_states");
  bp.batch->check(mas_hip_set_controls_host(bp.batch->get(), U.data()), "set_controls");
  bp.batch->check(mas_hip_initialize_problem(bp.batch->get()), "initialize_problem");
  return bp;
}

inline BatchedProblem upload(MultiAgentProblem& problem, const SolverParams& params, bool warm_from_best) {
  require_homogeneous(problem);
  std::vector<int> all(problem.blocks.size());
  for (size_t i = 0; i < all.size(); ++i) all[i] = static_cast<int>(i);
  return upload(problem, all, params, warm_from_best);
}

// write the results of one batch back into its OCPs
inline void download_into(MultiAgentProblem& problem, BatchedProblem& bp, bool publish_initial, std::vector<int>& iterations) {
  const size_t nX = static_cast<size_t>(bp.nx) * (bp.T + 1), nU = static_cast<size_t>(bp.nu) * bp.T;
  std::vector<double> X(nX * bp.N), U(nU * bp.N), cost(bp.N);
  std::vector<int32_t> it(bp.N), ex(bp.N);
  bp.batch->check(mas_hip_get_solution_host(bp.batch->get(), X.data(), U.data(), cost.data(), it.data(), ex.data()), "get_solution");
  for (int i = 0; i < bp.N; ++i) {
    OCP& o = *problem.blocks[bp.members[i]].agent->ocp;
    o.best_states = StateTrajectory(bp.nx, bp.T + 1);
    o.best_controls = ControlTrajectory(bp.nu, bp.T);
    std::copy(X.begin() + nX * i, X.begin() + nX * (i + 1), o.best_states.data.begin());
    std::copy(U.begin() + nU * i, U.begin() + nU * (i + 1), o.best_controls.data.begin());
    o.best_cost = cost[i];
    if (publish_initial) o.update_initial_with_best();   // nash.hpp:66-71
    iterations[bp.members[i]] = it[i];
  }
}

// collect_solution (nash.hpp:23-37): blocks in id order, total = sum of best_cost in that order
inline Solution collect_solution(MultiAgentProblem& problem, const std::vector<int>& iterations) {
  Solution sol;
  for (size_t i = 0; i < problem.blocks.size(); ++i) {
    const OCP& o = *problem.blocks[i].agent->ocp;
    sol.states.push_back(o.best_states); sol.controls.push_back(o.best_controls); sol.costs.push_back(o.best_cost);
    sol.total_cost += o.best_cost;
  }
  sol.iterations = iterations;
  return sol;
}

inline Solution download(MultiAgentProblem& problem, BatchedProblem& bp, bool publish_initial) {
  std::vector<int> iterations(problem.blocks.size(), 0);
  download_into(problem, bp, publish_initial, iterations);
  Solution sol = collect_solution(problem, iterations);
  bp.batch->check(mas_hip_total_cost(bp.batch->get(), &sol.total_cost), "total_cost");   // centralised: the stacked merit
  return sol;
}

// ---- several GPUs in one process: the device plan ----
// The reference fans the agents out over OpenMP threads (nash.hpp:59-71); here the id-sorted block list is cut into contiguous
// ranges (mas_hip_shard_range), one per entry of the device plan, and every range is driven by its own host thread on its own
// handle.  Default plan: every visible device.  set_device_plan({0, 0}) puts two shards on one GPU (tests on a one-GPU box);
// environment MAS_AMD_DEVICES="0,1,2,3" overrides the default.
inline std::vector<int>& device_plan_storage() { static std::vector<int> plan; return plan; }
inline void set_device_plan(std::vector<int> devices) { device_plan_storage() = std::move(devices); }
inline std::vector<int> device_plan() {
  if (!device_plan_storage().empty()) return device_plan_storage();
  std::vector<int> plan;
  if (const char* e = std::getenv("MAS_AMD_DEVICES")) {
    for (const char* c = e; *c;) { char* end = nullptr; const long v = std::strtol(c, &end, 10); if (end == c) break; plan.push_back(static_cast<int>(v)); c = (*end == ',') ? end + 1 : end; }
  }
  if (plan.empty()) { const int n = std::max(1, mas_hip_device_count()); for (int i = 0; i < n; ++i) plan.push_back(i); }
  return plan;
}
// shards smaller than this stay on one device (a handle per device costs ~1 ms to set up)
constexpr int kMinAgentsPerShard = 32;

// run `run(batch)` on the members of one homogeneous group, sharded over the device plan; `coupled`: the shards exchange the total
// cost of all agents (line search) through mas_hip_set_comm_cost_exchange -- over RCCL when every shard has its own device, over the
// library's in-process loopback transport when two shards share one (RCCL refuses two ranks on a GPU; tests on a one-GPU box).
// Nothing that can fail is left for after the first collective: phase 1 creates every handle, uploads and installs the
// exchange on every shard, and only when ALL shards stand does phase 2 start the strategy; a shard that fails in phase 2 (a kernel
// gave up, a device error) aborts every communicator of the group, so that no peer waits for it inside a collective.
template <class Run>
inline void run_sharded(MultiAgentProblem& problem, const std::vector<int>& members, const SolverParams& params, bool publish_initial,
                        std::vector<int>& iterations, bool coupled, Run&& run) {
  std::vector<int> plan = device_plan();
  int shards = static_cast<int>(plan.size());
  while (shards > 1 && static_cast<int>(members.size()) / shards < kMinAgentsPerShard) --shards;
  if (shards <= 1) {
    auto bp = upload(problem, members, params, true, plan.empty() ? 0 : plan[0]);
    run(*bp.batch);
    download_into(problem, bp, publish_initial, iterations);
    return;
  }
  plan.resize(shards);
  const int n_total = static_cast<int>(members.size());
  bool distinct = true;
  for (int i = 0; i < shards; ++i) for (int j = i + 1; j < shards; ++j) if (plan[i] == plan[j]) distinct = false;
  std::vector<mas_hip_comm*> comms(shards, nullptr);
  struct CommGuard {
    std::vector<mas_hip_comm*>& c;
    ~CommGuard() { for (auto* p : c) if (p) mas_hip_comm_destroy(p); }
  } guard{comms};
  if (coupled) {
    if (!distinct) mas_hip_comm_use_loopback(1);          // (communicators keep the transport they were made by)
    const int rc = mas_hip_comm_create_all(shards, plan.data(), comms.data());
    if (!distinct) mas_hip_comm_use_loopback(0);
    if (rc != MAS_HIP_OK) throw HipError(MAS_HIP_ERR_RUNTIME, std::string("mas_hip_comm_create_all: ") + mas_hip_comm_last_error());
  }
  using Uploaded = BatchedProblem;
  std::vector<std::unique_ptr<Uploaded>> batches(shards);
  std::vector<std::exception_ptr> errors(shards);
  auto in_parallel = [&](auto&& body) {
    std::vector<std::thread> threads;
    for (int r = 0; r < shards; ++r) threads.emplace_back([&, r] { try { body(r); } catch (...) { errors[r] = std::current_exception(); } });
    for (auto& t : threads) t.join();
  };
  // phase 1: everything that can fail before a collective
  in_parallel([&](int r) {
    int lo = 0, hi = 0; mas_hip_shard_range(n_total, shards, r, &lo, &hi);
    const std::vector<int> mine(members.begin() + lo, members.begin() + hi);
    batches[r] = std::make_unique<Uploaded>(upload(problem, mine, params, true, plan[r]));
    if (coupled) batches[r]->batch->check(mas_hip_set_comm_cost_exchange(batches[r]->batch->get(), comms[r], n_total), "set_comm_cost_exchange");
  });
  for (auto& e : errors) if (e) std::rethrow_exception(e);        // no collective has started: nobody waits for anybody
  // phase 2: the strategy; a failing shard releases its peers
  std::once_flag abort_once;
  in_parallel([&](int r) {
    try {
      run(*batches[r]->batch);
      download_into(problem, *batches[r], publish_initial, iterations);     // every shard writes its own OCPs and its own entries of `iterations`
    } catch (...) {
      std::call_once(abort_once, [&] { for (auto* c : comms) if (c) mas_hip_comm_abort(c); });
      throw;
    }
  });
  batches.clear();                                                  // the handles go before their communicators (CommGuard)
  for (auto& e : errors) if (e) std::rethrow_exception(e);
}

// per-agent strategies over a possibly mixed agent list: one batch per homogeneous group
template <class Run>
inline Solution run_grouped(MultiAgentProblem& problem, const SolverParams& params, Run&& run) {
  if (problem.blocks.empty()) return Solution{};      // nothing to solve: empty Solution, total_cost 0 (nash.hpp:23-37)
  std::vector<int> iterations(problem.blocks.size(), 0);
  for (const auto& members : group_blocks(problem)) run_sharded(problem, members, params, true, iterations, /*coupled=*/false, run);
  return collect_solution(problem, iterations);
}

inline const SolverParams& params_of(const Solver& s) { return std::get<iLQR>(s).params(); }

}  // namespace detail

// OCP::initialize_problem for every agent in one GPU batch (what the examples do in a loop, e.g.
// examples/multi_agent_single_track.cpp:113-118)
inline void initialize_agents(MultiAgentProblem& problem) {
  problem.compute_offsets();
  if (problem.blocks.empty()) return;
  for (auto& b : problem.blocks) {
    OCP& o = *b.agent->ocp;
    if (o.initial_controls.rows() != o.control_dim || o.initial_controls.cols() != o.horizon_steps)
      o.initial_controls = ControlTrajectory::Zero(o.control_dim, o.horizon_steps);
  }
  SolverParams none;
  auto bp = detail::upload(problem, none, /*warm_from_best=*/false);
  Solution s = detail::download(problem, bp, false);
  for (size_t i = 0; i < problem.blocks.size(); ++i) {
    OCP& o = *problem.blocks[i].agent->ocp;
    o.initial_states = o.best_states; o.best_controls = o.initial_controls; o.initialized = true;
  }
}

// ----------------------------------------------------------------------- strategies ------
// strategies/centralized.hpp:10-39.  The reference stacks all agents into one OCP and finite-differences the SUMMED cost
// (dense cross-agent Hessians): `Literal` is that solve on the device (mas_hip_run_centralized_dense, <= 256 agents: chip-wide
// launches, 0.4 s at 64 agents and 2.7 s at 256 for H = 100); `Blocked` is the block-diagonal reading that scales
// (mas_hip_run_centralized, <= 1024 agents, a millisecond at 256; a different number).  `Auto` takes the
// literal solve -- the reference's own number -- up to 64 agents (the reference's example defaults to 10) and the reading that
// scales beyond.
struct CentralizedStrategy {
  enum class Reading { Auto, Literal, Blocked };
  static constexpr int kLiteralMaxAgents = 64;
  Solver solver;
  Reading reading = Reading::Auto;
  explicit CentralizedStrategy(Solver s, Reading r = Reading::Auto) : solver(std::move(s)), reading(r) {}
  Solution operator()(MultiAgentProblem& problem) {
    problem.compute_offsets();
    if (problem.blocks.empty()) return Solution{};
    auto bp = detail::upload(problem, detail::params_of(solver), false);
    const bool literal = reading == Reading::Literal || (reading == Reading::Auto && bp.N <= kLiteralMaxAgents);
    if (literal) bp.batch->check(mas_hip_run_centralized_dense(bp.batch->get()), "run_centralized_dense");
    else bp.batch->check(mas_hip_run_centralized(bp.batch->get()), "run_centralized");
    return detail::download(problem, bp, false);
  }
};

struct SequentialNashStrategy {                    // strategies/nash.hpp:250-268
  int max_outer; Solver solver_proto; SolverParams params;
  SequentialNashStrategy(int outer, Solver s, SolverParams p) : max_outer(outer), solver_proto(std::move(s)), params(std::move(p)) {}
  Solution operator()(MultiAgentProblem& problem) {
    problem.compute_offsets();
    const int outer = max_outer;
    return detail::run_grouped(problem, params, [outer](detail::Batch& b) { b.check(mas_hip_run_sequential(b.get(), outer), "run_sequential"); });
  }
};

struct LineSearchNashStrategy {                    // strategies/nash.hpp:270-288
  int max_outer; Solver solver_proto; SolverParams params;
  LineSearchNashStrategy(int outer, Solver s, SolverParams p) : max_outer(outer), solver_proto(std::move(s)), params(std::move(p)) {}
  Solution operator()(MultiAgentProblem& problem) {
    problem.compute_offsets();
    if (problem.blocks.empty()) return Solution{};
    detail::require_homogeneous(problem);
    std::vector<int> all(problem.blocks.size()), iterations(problem.blocks.size(), 0);
    for (size_t i = 0; i < all.size(); ++i) all[i] = static_cast<int>(i);
    const int outer = max_outer;
    detail::run_sharded(problem, all, params, true, iterations, /*coupled=*/true,
                        [outer](detail::Batch& b) { b.check(mas_hip_run_line_search(b.get(), outer), "run_line_search"); });
    return detail::collect_solution(problem, iterations);
  }
};

struct TrustRegionNashStrategy {                   // strategies/nash.hpp:290-308
  int max_outer; Solver solver_proto; SolverParams params;
  TrustRegionNashStrategy(int outer, Solver s, SolverParams p) : max_outer(outer), solver_proto(std::move(s)), params(std::move(p)) {}
  Solution operator()(MultiAgentProblem& problem) {
    problem.compute_offsets();
    const int outer = max_outer;
    return detail::run_grouped(problem, params, [outer](detail::Batch& b) { b.check(mas_hip_run_trust_region(b.get(), outer), "run_trust_region"); });
  }
};

using Strategy = std::variant<CentralizedStrategy, SequentialNashStrategy, LineSearchNashStrategy, TrustRegionNashStrategy>;   // strategy.hpp:13
inline Solution solve(Strategy& strategy, MultiAgentProblem& problem) { return std::visit([&](auto& s) { return s(problem); }, strategy); }

}  // namespace mas_amd
