// multi_agent_single_track -- the north-star workload: N single-track cars on a circular track
// (reference: examples/multi_agent_single_track.cpp; same flags, result line and tables -- example_utils.hpp).
#include <cmath>

#include "example_utils.hpp"

int main(int argc, char** argv) {
  examples::Workload w;
  w.program = "multi_agent_single_track";
  w.multi_agent = true;
  w.default_horizon = 10;                                                        // the reference hard-codes 10 steps (:108)
  w.params = {{"max_iterations", 100}, {"tolerance", 1e-5}, {"max_ms", 1000}};   // :105-107
  w.make_ocp = [](int index, int n_agents, int horizon) {
    // agent i starts at theta = 2 pi i / N on the circle, heading along it (:41-45,113-117); dynamics, cost and the
    // default bounds live in the registered device model MAS_HIP_MODEL_SINGLE_TRACK
    constexpr double track_radius = 20.0, target_velocity = 5.0;
    const double theta = 2.0 * M_PI * index / n_agents;
    mas_amd::OCP p;
    p.state_dim = 4; p.control_dim = 2; p.horizon_steps = horizon; p.dt = 0.5;
    p.initial_state = {track_radius * std::cos(theta), track_radius * std::sin(theta), 1.57 + theta, 4.0};
    p.model.model_id = MAS_HIP_MODEL_SINGLE_TRACK;
    p.model.params = std::vector<double>{track_radius, target_velocity, 1.0, 1.0, 0.001, 0.001, 2.5};   // w_track, w_speed, w_delta, w_acc, L
    p.input_lower_bounds = mas_amd::Control(2, -0.5);
    p.input_upper_bounds = mas_amd::Control(2, 0.5);
    return p;
  };
  return examples::run_example(w, argc, argv);
}
