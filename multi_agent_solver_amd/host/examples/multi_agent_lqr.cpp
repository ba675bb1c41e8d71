// multi_agent_lqr -- examples/multi_agent_lqr.cpp on the MI355X iLQR path (identity LQR, n_x = n_u = 4).
#include <chrono>
#include <iomanip>
#include <iostream>
#include <limits>

#include "example_utils.hpp"

int main(int argc, char** argv) {
  using namespace mas_amd;
  try {
    const auto options = examples::cli::parse(argc, argv, true);
    if (options.show_help) {
      std::cout << "Usage: multi_agent_lqr [--agents N] [--solver NAME] [--strategy NAME] [--max-outer N] [--horizon T]\n"
                << "       multi_agent_lqr N\n\n";
      examples::print_available(std::cout);
      return 0;
    }
    examples::canonical_solver_name(options.solver);       // unknown names fail before any GPU work
    examples::canonical_strategy_name(options.strategy);
    constexpr int n_x = 4, n_u = 4;
    const int T = options.horizon > 0 ? options.horizon : 10;            // multi_agent_lqr.cpp:108
    constexpr double dt = 0.1;
    MultiAgentProblem problem;
    for (int i = 0; i < options.agents; ++i) {                           // create_linear_lqr_ocp, :21-76
      auto ocp = std::make_shared<OCP>();
      ocp->state_dim = n_x; ocp->control_dim = n_u; ocp->dt = dt; ocp->horizon_steps = T;
      ocp->initial_state = State(n_x, 0.0); ocp->initial_state[0] = 1.0;
      ocp->model.model_id = MAS_HIP_MODEL_LQR_IDENTITY;
      ocp->id = static_cast<size_t>(i);
      problem.add_agent(std::make_shared<Agent>(i, ocp));
    }
    initialize_agents(problem);
    SolverParams params{{"max_iterations", 100}, {"tolerance", 1e-5}, {"max_ms", 100}};
    if (options.horizon > 0) params["max_ms"] = std::numeric_limits<double>::infinity();

    auto solver = examples::make_solver(options.solver);
    Strategy strategy = examples::make_strategy(options.strategy, std::move(solver), params, options.max_outer);
    const auto start = std::chrono::steady_clock::now();
    const auto solution = mas_amd::solve(strategy, problem);
    const double elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count();
    std::cout << std::fixed << std::setprecision(6) << "solver=" << examples::canonical_solver_name(options.solver)
              << " strategy=" << examples::canonical_strategy_name(options.strategy) << " agents=" << options.agents
              << " cost=" << solution.total_cost << " time_ms=" << elapsed_ms << '\n';
    if (!options.quiet) {
      for (std::size_t idx = 0; idx < solution.states.size(); ++idx) {
        const std::string base = "agent_" + std::to_string(problem.blocks[idx].agent_id);
        examples::print_state_trajectory(std::cout, solution.states[idx], dt, base);
        examples::print_control_trajectory(std::cout, solution.controls[idx], dt, base);
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    std::cerr << "Use --help to see available options.\n";
    return 1;
  }
  return 0;
}
