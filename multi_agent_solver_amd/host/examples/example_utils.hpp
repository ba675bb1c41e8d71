// example_utils.hpp -- name handling, factories and printers of the example programs, mirroring
// examples/example_utils.hpp:19-167 and examples/cli.hpp:15-220 of the reference so that
// scripts/compare_solvers.py:122-136 and the plotting scripts can drive these binaries unchanged.
#pragma once

#include <algorithm>
#include <cctype>
#include <charconv>
#include <ostream>
#include <stdexcept>
#include <string>
#include <string_view>

#include "../mas_amd/mas_amd.hpp"

namespace examples {

inline std::string normalize_key(const std::string& value) {          // example_utils.hpp:19-30
  std::string out;
  for (char ch : value)
    if (std::isalnum(static_cast<unsigned char>(ch))) out.push_back(static_cast<char>(std::tolower(static_cast<unsigned char>(ch))));
  return out;
}

inline std::string canonical_solver_name(const std::string& name) {   // :32-49; cgd / osqp are not part of this path
  const std::string key = normalize_key(name);
  if (key == "ilqr" || key == "primaldualilqr" || key == "pdilqr") return "ilqr";
  throw std::invalid_argument("Unknown solver '" + name + "'.");
}

inline std::string canonical_strategy_name(const std::string& name) { // :51-64
  const std::string key = normalize_key(name);
  if (key == "centralized" || key == "centralised") return "centralized";
  if (key == "sequential" || key == "sequentialnash") return "sequential";
  if (key == "linesearch" || key == "linesearchnash") return "linesearch";
  if (key == "trustregion" || key == "trustregionnash") return "trustregion";
  throw std::invalid_argument("Unknown strategy '" + name + "'.");
}

inline mas_amd::Solver make_solver(const std::string& name) {         // :77-93
  if (canonical_solver_name(name) == "ilqr") return mas_amd::Solver{std::in_place_type<mas_amd::iLQR>};
  throw std::invalid_argument("Unknown solver '" + name + "'.");
}

inline mas_amd::Strategy make_strategy(const std::string& name, mas_amd::Solver solver, const mas_amd::SolverParams& params, int max_outer) {  // :95-112
  const std::string canonical = canonical_strategy_name(name);
  if (canonical == "centralized") { mas_amd::set_params(solver, params); return mas_amd::Strategy{mas_amd::CentralizedStrategy{std::move(solver)}}; }
  if (canonical == "sequential") return mas_amd::Strategy{mas_amd::SequentialNashStrategy{max_outer, std::move(solver), params}};
  if (canonical == "linesearch") return mas_amd::Strategy{mas_amd::LineSearchNashStrategy{max_outer, std::move(solver), params}};
  return mas_amd::Strategy{mas_amd::TrustRegionNashStrategy{max_outer, std::move(solver), params}};
}

inline void print_available(std::ostream& os) {                        // :114-123
  os << "Available solvers: ilqr\n";
  os << "Available strategies: centralized, sequential, linesearch, trustregion\n";
}

inline void print_trajectory(std::ostream& os, const mas_amd::Matrix& m, double dt, const std::string& label, const char* suffix, char var) {
  if (m.size() == 0) return;                                           // :125-167
  os << label << suffix << "\ntime";
  for (int r = 0; r < m.rows(); ++r) os << ',' << var << r;
  os << '\n';
  for (int c = 0; c < m.cols(); ++c) {
    os << (dt > 0.0 ? static_cast<double>(c) * dt : static_cast<double>(c));
    for (int r = 0; r < m.rows(); ++r) os << ',' << m(r, c);
    os << '\n';
  }
  os << '\n';
}
inline void print_state_trajectory(std::ostream& os, const mas_amd::Matrix& s, double dt, const std::string& label) { print_trajectory(os, s, dt, label, "_states", 'x'); }
inline void print_control_trajectory(std::ostream& os, const mas_amd::Matrix& u, double dt, const std::string& label) { print_trajectory(os, u, dt, label, "_controls", 'u'); }

namespace cli {                                                        // examples/cli.hpp

inline std::string normalize_option(std::string option) {             // '_' and '-' interchangeable in option names
  if (option.rfind("--", 0) != 0) return option;
  const auto eq = option.find('=');
  std::replace(option.begin() + 2, option.begin() + static_cast<std::ptrdiff_t>(eq == std::string::npos ? option.size() : eq), '_', '-');
  return option;
}
inline int parse_int(const std::string& label, const std::string& value) {
  int result = 0;
  const auto [ptr, ec] = std::from_chars(value.data(), value.data() + value.size(), result);
  if (ec != std::errc() || ptr != value.data() + value.size()) throw std::invalid_argument("Invalid value for " + label + ": '" + value + "'");
  return result;
}

struct Options {
  bool show_help = false;
  int agents = 10;                   // cli.hpp:163-167 defaults
  int max_outer = 10;
  std::string solver = "ilqr";
  std::string strategy = "centralized";
  int horizon = 0;                   // extension (the reference hard-codes T): 0 = the example's own value
  bool quiet = false;                // extension: result line only
};

inline Options parse(int argc, char** argv, bool multi_agent) {
  Options o;
  bool positional_agents = false;
  for (int i = 1; i < argc; ++i) {
    const std::string raw = argv[i];
    const std::string cur = normalize_option(raw);
    auto value_of = [&](const std::string& name, std::string& out) -> bool {
      if (cur == name) { if (i + 1 >= argc) throw std::invalid_argument("Missing value for option '" + name + "'"); out = argv[++i]; return true; }
      if (cur.rfind(name + "=", 0) == 0) { out = cur.substr(name.size() + 1); return true; }
      return false;
    };
    std::string v;
    if (cur == "--help" || cur == "-h") { o.show_help = true; continue; }
    if (value_of("--solver", v)) { o.solver = v; continue; }
    if (value_of("--horizon", v)) { o.horizon = parse_int("--horizon", v); continue; }
    if (cur == "--quiet") { o.quiet = true; continue; }
    if (multi_agent) {
      if (value_of("--agents", v)) { o.agents = parse_int("--agents", v); continue; }
      if (value_of("--strategy", v)) { o.strategy = v; continue; }
      if (value_of("--max-outer", v)) { o.max_outer = parse_int("--max-outer", v); continue; }
      if (!raw.empty() && raw.front() != '-' && !positional_agents) { o.agents = parse_int("agents", raw); positional_agents = true; continue; }
    }
    throw std::invalid_argument("Unknown argument '" + raw + "'");
  }
  return o;
}

}  // namespace cli
}  // namespace examples
