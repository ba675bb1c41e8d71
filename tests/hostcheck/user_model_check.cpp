// TEST-ONLY (g++, -DMAS_HOST_EMULATION): a USER-DEFINED workload -- the kind of struct mas_hip_register_model compiles at run time
// (FdModel<Self, NX, NU>: dynamics, stage cost, terminal cost; every derivative finite-differenced) -- through the batch driver, lane
// bodies replayed on the CPU.  It is the single-track workload written again, so its results must be the oracle's (and the built-in
// model's) bit for bit: printed here in hex, compared by tests/test_hostcheck_parity.py.  usage: user_model_check <agents> <T> <strategy>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../multi_agent_solver_amd/csrc/mas_solver.hpp"

namespace mas {
struct UserSingleTrack : FdModel<UserSingleTrack, 4, 2> {
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = lb[1] = -0.5; ub[0] = ub[1] = 0.5; }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    const double psi = x[2], v = x[3], L = P.p[6];
    d[0] = v * mas_dt_cos(psi);
    d[1] = v * mas_dt_sin(psi);
    d[2] = (v * mas_dt_tan(u[0])) / L;
    d[3] = u[1];
  }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {
    const double dist = fabs(sqrt(s[0] * s[0] + s[1] * s[1]) - P.p[0]);
    const double se = s[3] - P.p[1];
    return P.p[2] * dist * dist + P.p[3] * se * se + P.p[4] * c[0] * c[0] + P.p[5] * c[1] * c[1];
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
};
}  // namespace mas

int main(int argc, char** argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 5, T = argc > 2 ? std::atoi(argv[2]) : 20;
  const char* strategy = argc > 3 ? argv[3] : "sequential";
  mas::ProblemDesc desc;
  desc.model_id = 100; desc.n_agents = N; desc.horizon_steps = T; desc.dt = 0.5;
  desc.has_params = true;
  const double p[7] = {20.0, 5.0, 1.0, 1.0, 0.001, 0.001, 2.5};          // examples/multi_agent_single_track.cpp:49-52,109-110
  for (int i = 0; i < 7; ++i) desc.params[i] = p[i];
  mas::SolverImpl<mas::UserSingleTrack> s(desc, nullptr);
  s.set_param("max_iterations", 100); s.set_param("tolerance", 1e-5); s.set_param("max_ms", std::numeric_limits<double>::infinity());
  std::vector<double> x0(static_cast<size_t>(N) * 4);
  for (int i = 0; i < N; ++i) {                                          // the reference's generator (agents on the circle), read from stdin as hex so that the inputs are the oracle's bits
    for (int k = 0; k < 4; ++k) { char buf[64]; if (std::scanf("%63s", buf) != 1) return 2; x0[static_cast<size_t>(i) * 4 + k] = std::strtod(buf, nullptr); }
  }
  s.set_initial_states(x0.data(), false); s.set_controls(nullptr, false); s.initialize_problem();
  if (!std::strcmp(strategy, "sequential")) s.run_sequential(3);
  else if (!std::strcmp(strategy, "trustregion")) s.run_trust_region(3);
  else if (!std::strcmp(strategy, "linesearch")) s.run_line_search(2);
  else if (!std::strcmp(strategy, "centralized")) s.run_centralized();
  else s.solve();
  std::vector<double> X(static_cast<size_t>(N) * (T + 1) * 4), U(static_cast<size_t>(N) * T * 2), cost(N);
  std::vector<int32_t> it(N), ex(N);
  s.get_solution(X.data(), U.data(), cost.data(), it.data(), ex.data(), false);
  for (int i = 0; i < N; ++i) std::printf("agent %d cost %a iters %d xT %a %a %a %a u0 %a %a\n", i, cost[i], it[i], X[(static_cast<size_t>(i) * (T + 1) + T) * 4 + 0],
                                          X[(static_cast<size_t>(i) * (T + 1) + T) * 4 + 1], X[(static_cast<size_t>(i) * (T + 1) + T) * 4 + 2],
                                          X[(static_cast<size_t>(i) * (T + 1) + T) * 4 + 3], U[static_cast<size_t>(i) * T * 2], U[static_cast<size_t>(i) * T * 2 + 1]);
  return 0;
}
