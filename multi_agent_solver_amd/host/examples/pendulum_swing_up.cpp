// pendulum_swing_up -- one agent, direct solve(solver, ocp) (reference: examples/pendulum_swing_up.cpp).
#include <cmath>

#include "example_utils.hpp"

int main(int argc, char** argv) {
  examples::Workload w;
  w.program = "pendulum_swing_up";
  w.multi_agent = false;
  w.single_label = "pendulum";
  w.default_horizon = 60;
  w.params = {{"max_iterations", 1000}, {"tolerance", 1e-4}, {"max_ms", 5000}};   // :149-152
  w.make_ocp = [](int, int, int horizon) {
    mas_amd::OCP p;                                                               // :29-117
    constexpr double torque_max = 5.0;
    p.state_dim = 2; p.control_dim = 1; p.horizon_steps = horizon; p.dt = 0.05;
    p.initial_state = {M_PI - 0.05, 0.0};
    p.model.model_id = MAS_HIP_MODEL_PENDULUM;                                      // model constant horizon_d defaults to T
    p.input_lower_bounds = mas_amd::Control(1, -torque_max);
    p.input_upper_bounds = mas_amd::Control(1, torque_max);
    p.initial_controls = mas_amd::ControlTrajectory::Zero(1, horizon);             // warm start: a small sine torque (:105-113)
    for (int k = 0; k < horizon; ++k) p.initial_controls(0, k) = 0.2 * torque_max * std::sin(2.0 * M_PI * (k * p.dt));
    return p;
  };
  return examples::run_example(w, argc, argv);
}
