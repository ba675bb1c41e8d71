// mixed_agents -- not a reference example: a MultiAgentProblem whose agents use DIFFERENT models (single-track
// cars and identity-LQR agents, interleaved by id) solved by the per-agent Nash strategies.  The reference
// allows this (every agent owns its OCP, nash.hpp:59-64); the GPU path splits the block list into one batch per
// homogeneous group.  Prints one line per agent: id, model, cost, iterations.
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>

#include "example_utils.hpp"

int main(int argc, char** argv) {
  using namespace mas_amd;
  try {
    const auto options = examples::cli::parse(argc, argv, true);
    const int T = options.horizon > 0 ? options.horizon : 20;
    MultiAgentProblem problem;
    for (int i = 0; i < options.agents; ++i) {
      auto ocp = std::make_shared<OCP>();
      ocp->horizon_steps = T; ocp->id = static_cast<size_t>(i);
      if (i % 2 == 0) {
        const double theta = 2.0 * M_PI * i / options.agents;
        ocp->state_dim = 4; ocp->control_dim = 2; ocp->dt = 0.5;
        ocp->initial_state = {20.0 * std::cos(theta), 20.0 * std::sin(theta), 1.57 + theta, 4.0};
        ocp->model.model_id = MAS_HIP_MODEL_SINGLE_TRACK;
        ocp->input_lower_bounds = Control(2, -0.5); ocp->input_upper_bounds = Control(2, 0.5);
      } else {
        ocp->state_dim = 4; ocp->control_dim = 4; ocp->dt = 0.1;
        ocp->initial_state = {1.0, 0.0, 0.0, 0.0};
        ocp->model.model_id = MAS_HIP_MODEL_LQR_IDENTITY;
      }
      problem.add_agent(std::make_shared<Agent>(i, ocp));
    }
    SolverParams params{{"max_iterations", 100}, {"tolerance", 1e-5}, {"max_ms", std::numeric_limits<double>::infinity()}};
    const std::string strat = examples::canonical_strategy_name(options.strategy == "centralized" ? "sequential" : options.strategy);
    Strategy strategy = examples::make_strategy(strat, examples::make_solver(options.solver), params, options.max_outer);
    const auto solution = mas_amd::solve(strategy, problem);
    std::cout << std::setprecision(17) << "strategy=" << strat << " agents=" << options.agents << " cost=" << solution.total_cost << '\n';
    for (std::size_t i = 0; i < problem.blocks.size(); ++i)
      std::cout << "agent " << problem.blocks[i].agent_id << " model " << problem.blocks[i].agent->ocp->model.model_id << " cost "
                << solution.costs[i] << " iterations " << solution.iterations[i] << '\n';
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
