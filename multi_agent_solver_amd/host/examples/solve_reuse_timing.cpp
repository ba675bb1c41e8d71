// solve_reuse_timing -- not a reference example: times repeated iLQR::solve(OCP&) calls on ONE solver object, the way
// the reference's strategies use a solver (nash.hpp:59-64: the same solver object once per outer iteration).  Run it
// twice, with and without MAS_AMD_NO_HANDLE_CACHE=1, to see what keeping the device handle between calls saves
// (INTEGRATION.md section 3).  Prints: calls, mean ms per call, cost of the last call.
#include <chrono>
#include <cmath>
#include <iostream>
#include <limits>

#include "example_utils.hpp"

int main(int argc, char** argv) {
  using namespace mas_amd;
  try {
    const int calls = argc > 1 ? examples::cli::parse_int("calls", argv[1]) : 20;
    OCP p;                                                  // the lane-following car of examples/single_track_ocp.cpp, T = 80
    p.state_dim = 4; p.control_dim = 2; p.horizon_steps = 80; p.dt = 0.1;
    p.initial_state = {0.0, 1.0, 0.0, 0.0};
    p.model.model_id = MAS_HIP_MODEL_SINGLE_TRACK_LANE;
    p.input_lower_bounds = Control{-0.7, -1.0};
    p.input_upper_bounds = Control{0.7, 1.0};
    p.initialize_problem();
    iLQR solver;
    solver.set_params({{"max_iterations", 1}, {"tolerance", 1e-5}, {"max_ms", std::numeric_limits<double>::infinity()}});
    solver.solve(p);                                        // first call: library load, code object, first allocation
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < calls; ++i) solver.solve(p);        // one iLQR iteration per call, warm-started from best_controls
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::cout << "calls=" << calls << " ms_per_call=" << ms / calls << " cost=" << p.best_cost << " iterations_last_call=" << solver.last_iterations << '\n';
  } catch (const std::exception& e) {
    std::cerr << "Error: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
