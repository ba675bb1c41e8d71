// pendulum_swing_up -- examples/pendulum_swing_up.cpp on the MI355X iLQR path (single agent, direct solve).
#include <chrono>
#include <cmath>
#include <iomanip>
#include <iostream>
#include <limits>

#include "example_utils.hpp"

static mas_amd::OCP create_pendulum_swingup_ocp(int horizon) {           // pendulum_swing_up.cpp:29-117
  using namespace mas_amd;
  OCP p;
  p.state_dim = 2; p.control_dim = 1; p.horizon_steps = horizon; p.dt = 0.05;
  p.initial_state = {M_PI - 0.05, 0.0};
  p.model.model_id = MAS_HIP_MODEL_PENDULUM;                               // params default to horizon_d = T
  const double torque_max = 5.0;
  p.input_lower_bounds = Control(1, -torque_max);
  p.input_upper_bounds = Control(1, torque_max);
  p.initial_controls = ControlTrajectory::Zero(p.control_dim, p.horizon_steps);
  for (int k = 0; k < p.horizon_steps; ++k) {
    const double t = k * p.dt;
    p.initial_controls(0, k) = 0.2 * torque_max * std::sin(2.0 * M_PI * t);   // host-side input generation only
  }
  p.initialize_problem();
  return p;
}

int main(int argc, char** argv) {
  using namespace mas_amd;
  try {
    const auto options = examples::cli::parse(argc, argv, false);
    if (options.show_help) { std::cout << "Usage: pendulum_swing_up [--solver NAME] [--horizon T]\n\n"; examples::print_available(std::cout); return 0; }
    examples::canonical_solver_name(options.solver);
    OCP problem = create_pendulum_swingup_ocp(options.horizon > 0 ? options.horizon : 60);
    SolverParams params;
    params["max_iterations"] = 1000; params["tolerance"] = 1e-4; params["max_ms"] = 5000;
    if (options.horizon > 0) params["max_ms"] = std::numeric_limits<double>::infinity();
    auto solver = examples::make_solver(options.solver);
    mas_amd::set_params(solver, params);
    const auto start = std::chrono::steady_clock::now();
    mas_amd::solve(solver, problem);
    const double elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start).count();
    std::cout << std::fixed << std::setprecision(6) << "solver=" << examples::canonical_solver_name(options.solver) << " cost=" << problem.best_cost
              << " time_ms=" << elapsed_ms << '\n';
    if (!options.quiet) {
      examples::print_state_trajectory(std::cout, problem.best_states, problem.dt, "pendulum");
      examples::print_control_trajectory(std::cout, problem.best_controls, problem.dt, "pendulum");
    }
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    std::cerr << "Use --help to see available options.\n";
    return 1;
  }
  return 0;
}
