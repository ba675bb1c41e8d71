// multi_agent_single_track -- the north-star workload of examples/multi_agent_single_track.cpp on the
// MI355X iLQR path: N single-track cars on a circular track, same flags, same result line.
//   multi_agent_single_track [--agents N] [--solver ilqr] [--strategy NAME] [--max-outer N] [--horizon T]
#include <chrono>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>

#include "example_utils.hpp"

// create_single_track_circular_ocp, examples/multi_agent_single_track.cpp:32-72 (dynamics, cost and bounds
// live in the registered device model MAS_HIP_MODEL_SINGLE_TRACK)
static mas_amd::OCP create_single_track_circular_ocp(double initial_theta, double track_radius, double target_velocity, int time_steps) {
  mas_amd::OCP p;
  p.state_dim = 4; p.control_dim = 2; p.horizon_steps = time_steps; p.dt = 0.5;
  p.initial_state = {track_radius * std::cos(initial_theta), track_radius * std::sin(initial_theta), 1.57 + initial_theta, 4.0};
  p.model.model_id = MAS_HIP_MODEL_SINGLE_TRACK;
  p.model.params = std::vector<double>{track_radius, target_velocity, 1.0, 1.0, 0.001, 0.001, 2.5};   // w_track, w_speed, w_delta, w_acc, L
  p.input_lower_bounds = mas_amd::Control(2, -0.5);
  p.input_upper_bounds = mas_amd::Control(2, 0.5);
  return p;
}

int main(int argc, char** argv) {
  using namespace mas_amd;
  try {
    const auto options = examples::cli::parse(argc, argv, true);
    if (options.show_help) {
      std::cout << "Usage: multi_agent_single_track [--agents N] [--solver NAME] [--strategy NAME] [--max-outer N] [--horizon T]\n"
                << "       multi_agent_single_track N\n\n";
      examples::print_available(std::cout);
      return 0;
    }
    examples::canonical_solver_name(options.solver);       // unknown names fail before any GPU work
    examples::canonical_strategy_name(options.strategy);
    SolverParams params{{"max_iterations", 100}, {"tolerance", 1e-5}, {"max_ms", 1000}};
    if (options.horizon > 0) params["max_ms"] = std::numeric_limits<double>::infinity();   // benchmark configs: bound work by iterations, not by the wall clock
    const int time_steps = options.horizon > 0 ? options.horizon : 10;                      // the reference hard-codes 10 (:108)
    constexpr double track_radius = 20.0, target_velocity = 5.0;

    MultiAgentProblem problem;
    for (int i = 0; i < options.agents; ++i) {
      const double theta = 2.0 * M_PI * i / options.agents;
      auto ocp = std::make_shared<OCP>(create_single_track_circular_ocp(theta, track_radius, target_velocity, time_steps));
      ocp->id = static_cast<size_t>(i);
      problem.add_agent(std::make_shared<Agent>(i, ocp));
    }
    initialize_agents(problem);            // OCP::initialize_problem for all agents, one GPU batch

    auto solver = examples::make_solver(options.solver);
    Strategy strategy = examples::make_strategy(options.strategy, std::move(solver), params, options.max_outer);
    const auto start = std::chrono::steady_clock::now();
    const auto solution = mas_amd::solve(strategy, problem);
    const auto end = std::chrono::steady_clock::now();
    const double elapsed_ms = std::chrono::duration<double, std::milli>(end - start).count();

    std::cout << std::fixed << std::setprecision(6) << "solver=" << examples::canonical_solver_name(options.solver)
              << " strategy=" << examples::canonical_strategy_name(options.strategy) << " agents=" << options.agents
              << " cost=" << solution.total_cost << " time_ms=" << elapsed_ms << '\n';
    if (!options.quiet) {
      for (std::size_t idx = 0; idx < solution.states.size() && idx < problem.blocks.size(); ++idx) {
        const auto& block = problem.blocks[idx];
        const std::string base = "agent_" + std::to_string(block.agent_id);
        examples::print_state_trajectory(std::cout, solution.states[idx], block.agent->ocp->dt, base);
        examples::print_control_trajectory(std::cout, solution.controls[idx], block.agent->ocp->dt, base);
      }
    }
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    std::cerr << "Use --help to see available options.\n";
    return 1;
  }
  return 0;
}
