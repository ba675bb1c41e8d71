// example_utils.hpp -- everything the example programs share: name tables, a small option parser, the result
// line / trajectory tables, and run_example(), the one driver all of them go through.
//
// What has to equal the reference (examples/example_utils.hpp:19-167, examples/cli.hpp:15-220), because
// scripts/compare_solvers.py:122-136 and the plotting scripts parse it, is the OBSERVABLE behaviour: accepted names and
// aliases, option spellings, error texts, the `solver= strategy= agents= cost= time_ms=` line and the
// `agent_<id>_states` / `_controls` tables.  How that behaviour is produced is this repository's own design: alias
// tables instead of if-chains, one declarative option list, and a Workload description per example instead of a main()
// per example.
#pragma once

#include <algorithm>
#include <array>
#include <cctype>
#include <charconv>
#include <chrono>
#include <functional>
#include <iomanip>
#include <iostream>
#include <limits>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>
#include <utility>

#include "../mas_amd/mas_amd.hpp"

namespace examples {

// ---------------------------------------------------------------------------------- names ------
// a name matches an alias when both agree after dropping everything but letters and digits, case-folded
inline std::string squash(std::string_view text) {
  std::string out;
  for (unsigned char ch : text)
    if (std::isalnum(ch)) out.push_back(static_cast<char>(std::tolower(ch)));
  return out;
}

struct Alias { std::string_view alias, canonical; };
inline constexpr std::array<Alias, 3> kSolverAliases{{{"ilqr", "ilqr"}, {"primaldualilqr", "ilqr"}, {"pdilqr", "ilqr"}}};   // cgd / osqp: other paths
inline constexpr std::array<Alias, 8> kStrategyAliases{{{"centralized", "centralized"}, {"centralised", "centralized"},
                                                        {"sequential", "sequential"}, {"sequentialnash", "sequential"},
                                                        {"linesearch", "linesearch"}, {"linesearchnash", "linesearch"},
                                                        {"trustregion", "trustregion"}, {"trustregionnash", "trustregion"}}};

template <size_t N>
std::string canonical_name(const std::array<Alias, N>& table, const char* kind, const std::string& name) {
  const std::string key = squash(name);
  for (const Alias& a : table)
    if (key == a.alias) return std::string(a.canonical);
  throw std::invalid_argument(std::string("Unknown ") + kind + " '" + name + "'.");
}
inline std::string canonical_solver_name(const std::string& name) { return canonical_name(kSolverAliases, "solver", name); }
inline std::string canonical_strategy_name(const std::string& name) { return canonical_name(kStrategyAliases, "strategy", name); }

inline mas_amd::Solver make_solver(const std::string& name) {
  canonical_solver_name(name);                               // throws for anything but the iLQR aliases
  return mas_amd::Solver{std::in_place_type<mas_amd::iLQR>};
}

inline mas_amd::Strategy make_strategy(const std::string& name, mas_amd::Solver solver, const mas_amd::SolverParams& params, int max_outer) {
  using namespace mas_amd;
  using Maker = Strategy (*)(Solver&&, const SolverParams&, int);
  static const std::pair<std::string_view, Maker> makers[] = {
      {"centralized", [](Solver&& s, const SolverParams& p, int) { set_params(s, p); return Strategy{CentralizedStrategy{std::move(s)}}; }},
      {"sequential", [](Solver&& s, const SolverParams& p, int outer) { return Strategy{SequentialNashStrategy{outer, std::move(s), p}}; }},
      {"linesearch", [](Solver&& s, const SolverParams& p, int outer) { return Strategy{LineSearchNashStrategy{outer, std::move(s), p}}; }},
      {"trustregion", [](Solver&& s, const SolverParams& p, int outer) { return Strategy{TrustRegionNashStrategy{outer, std::move(s), p}}; }}};
  const std::string canonical = canonical_strategy_name(name);
  for (const auto& m : makers)
    if (m.first == canonical) return m.second(std::move(solver), params, max_outer);
  throw std::logic_error("strategy table out of step with its alias table");
}

inline void print_available(std::ostream& os) {
  auto list = [&](const char* title, const auto& table) {
    os << "Available " << title << ": ";
    std::string_view last;
    bool first = true;
    for (const Alias& a : table)
      if (a.canonical != last) { os << (first ? "" : ", ") << a.canonical; last = a.canonical; first = false; }
    os << '\n';
  };
  list("solvers", kSolverAliases);
  list("strategies", kStrategyAliases);
}

// ------------------------------------------------------------------------------- printing ------
// `<label>_states` / `<label>_controls`, a `time,<v>0,<v>1,..` header, one row per column of the trajectory, blank line
inline void print_table(std::ostream& os, const mas_amd::Matrix& m, double dt, const std::string& title, char var) {
  if (m.size() == 0) return;
  os << title << "\ntime";
  for (int r = 0; r < m.rows(); ++r) os << ',' << var << r;
  os << '\n';
  for (int c = 0; c < m.cols(); ++c) {
    os << (dt > 0.0 ? c * dt : static_cast<double>(c));
    for (int r = 0; r < m.rows(); ++r) os << ',' << m(r, c);
    os << '\n';
  }
  os << '\n';
}
inline void print_state_trajectory(std::ostream& os, const mas_amd::Matrix& s, double dt, const std::string& label) { print_table(os, s, dt, label + "_states", 'x'); }
inline void print_control_trajectory(std::ostream& os, const mas_amd::Matrix& u, double dt, const std::string& label) { print_table(os, u, dt, label + "_controls", 'u'); }

// ---------------------------------------------------------------------------- command line ------
namespace cli {

struct Options {
  bool show_help = false;
  int agents = 10;                   // the reference's defaults (cli.hpp:163-167)
  int max_outer = 10;
  std::string solver = "ilqr";
  std::string strategy = "centralized";
  int horizon = 0;                   // extension (the reference hard-codes T): 0 = the example's own value
  bool quiet = false;                // extension: result line only
};

inline int parse_int(const std::string& label, const std::string& value) {
  int result = 0;
  const auto [ptr, ec] = std::from_chars(value.data(), value.data() + value.size(), result);
  if (ec != std::errc() || ptr != value.data() + value.size()) throw std::invalid_argument("Invalid value for " + label + ": '" + value + "'");
  return result;
}

// One declarative list of options.  An option name may be written with '_' for '-', and its value may follow as
// the next argument or after '='.
inline Options parse(int argc, char** argv, bool multi_agent) {
  Options o;
  struct Spec { const char* name; bool multi_only; bool takes_value; std::function<void(const std::string&)> apply; };
  const Spec specs[] = {
      {"--help", false, false, [&](const std::string&) { o.show_help = true; }},
      {"-h", false, false, [&](const std::string&) { o.show_help = true; }},
      {"--quiet", false, false, [&](const std::string&) { o.quiet = true; }},
      {"--solver", false, true, [&](const std::string& v) { o.solver = v; }},
      {"--horizon", false, true, [&](const std::string& v) { o.horizon = parse_int("--horizon", v); }},
      {"--agents", true, true, [&](const std::string& v) { o.agents = parse_int("--agents", v); }},
      {"--strategy", true, true, [&](const std::string& v) { o.strategy = v; }},
      {"--max-outer", true, true, [&](const std::string& v) { o.max_outer = parse_int("--max-outer", v); }}};
  bool positional_taken = false;
  for (int i = 1; i < argc; ++i) {
    const std::string raw = argv[i];
    std::string name = raw, value;
    bool has_inline_value = false;
    if (raw.rfind("--", 0) == 0) {
      const auto eq = raw.find('=');
      if (eq != std::string::npos) { name = raw.substr(0, eq); value = raw.substr(eq + 1); has_inline_value = true; }
      std::replace(name.begin() + 2, name.end(), '_', '-');
    }
    const Spec* hit = nullptr;
    for (const Spec& s : specs)
      if (name == s.name && (multi_agent || !s.multi_only) && (s.takes_value || !has_inline_value)) { hit = &s; break; }
    if (hit) {
      if (hit->takes_value && !has_inline_value) {
        if (i + 1 >= argc) throw std::invalid_argument(std::string("Missing value for option '") + hit->name + "'");
        value = argv[++i];
      }
      hit->apply(value);
    } else if (multi_agent && !positional_taken && !raw.empty() && raw.front() != '-') {
      o.agents = parse_int("agents", raw);          // `example N`
      positional_taken = true;
    } else {
      throw std::invalid_argument("Unknown argument '" + raw + "'");
    }
  }
  return o;
}

}  // namespace cli

// -------------------------------------------------------------------------------- driver ------
// What distinguishes one example from another.
struct Workload {
  const char* program;                          // name in the usage text
  bool multi_agent;                             // strategy run over N agents, or one direct solve
  int default_horizon;                          // the T the reference example hard-codes
  mas_amd::SolverParams params;                 // the example's solver parameters (max_ms is wall-clock in the reference)
  std::function<mas_amd::OCP(int index, int n_agents, int horizon)> make_ocp;
  const char* single_label = "agent";           // table label of a single-agent example
};

inline void print_usage(const Workload& w) {
  if (w.multi_agent)
    std::cout << "Usage: " << w.program << " [--agents N] [--solver NAME] [--strategy NAME] [--max-outer N] [--horizon T]\n"
              << "       " << w.program << " N\n\n";
  else
    std::cout << "Usage: " << w.program << " [--solver NAME] [--horizon T]\n\n";
  print_available(std::cout);
}

inline int run_example(const Workload& w, int argc, char** argv) {
  using namespace mas_amd;
  using clock = std::chrono::steady_clock;
  try {
    const cli::Options opt = cli::parse(argc, argv, w.multi_agent);
    if (opt.show_help) { print_usage(w); return 0; }
    const std::string solver_name = canonical_solver_name(opt.solver);          // unknown names fail before any GPU work
    const std::string strategy_name = w.multi_agent ? canonical_strategy_name(opt.strategy) : std::string();
    const int horizon = opt.horizon > 0 ? opt.horizon : w.default_horizon;
    SolverParams params = w.params;
    if (opt.horizon > 0) params["max_ms"] = std::numeric_limits<double>::infinity();   // benchmark configs bound work by iterations, not by the wall clock

    std::cout << std::fixed << std::setprecision(6);
    if (!w.multi_agent) {
      OCP problem = w.make_ocp(0, 1, horizon);
      problem.initialize_problem();
      Solver solver = make_solver(opt.solver);
      set_params(solver, params);
      const auto t0 = clock::now();
      solve(solver, problem);
      const double ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
      std::cout << "solver=" << solver_name << " cost=" << problem.best_cost << " time_ms=" << ms << '\n';
      if (!opt.quiet) {
        print_state_trajectory(std::cout, problem.best_states, problem.dt, w.single_label);
        print_control_trajectory(std::cout, problem.best_controls, problem.dt, w.single_label);
      }
      return 0;
    }

    MultiAgentProblem problem;
    for (int i = 0; i < opt.agents; ++i) {
      auto ocp = std::make_shared<OCP>(w.make_ocp(i, opt.agents, horizon));
      ocp->id = static_cast<size_t>(i);
      problem.add_agent(std::make_shared<Agent>(i, ocp));
    }
    initialize_agents(problem);                  // OCP::initialize_problem for all agents in one GPU batch
    Strategy strategy = make_strategy(opt.strategy, make_solver(opt.solver), params, opt.max_outer);
    const auto t0 = clock::now();
    const Solution solution = solve(strategy, problem);
    const double ms = std::chrono::duration<double, std::milli>(clock::now() - t0).count();
    std::cout << "solver=" << solver_name << " strategy=" << strategy_name << " agents=" << opt.agents << " cost=" << solution.total_cost
              << " time_ms=" << ms << '\n';
    if (!opt.quiet)
      for (std::size_t k = 0; k < solution.states.size() && k < problem.blocks.size(); ++k) {
        const auto& block = problem.blocks[k];
        const std::string label = "agent_" + std::to_string(block.agent_id);
        print_state_trajectory(std::cout, solution.states[k], block.agent->ocp->dt, label);
        print_control_trajectory(std::cout, solution.controls[k], block.agent->ocp->dt, label);
      }
    return 0;
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n" << "Use --help to see available options.\n";
    return 1;
  }
}

}  // namespace examples
