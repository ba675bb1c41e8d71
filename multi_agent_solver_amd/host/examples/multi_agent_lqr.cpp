// multi_agent_lqr -- identity LQR agents, n_x = n_u = 4 (reference: examples/multi_agent_lqr.cpp).
#include "example_utils.hpp"

int main(int argc, char** argv) {
  examples::Workload w;
  w.program = "multi_agent_lqr";
  w.multi_agent = true;
  w.default_horizon = 10;                                                       // :108
  w.params = {{"max_iterations", 100}, {"tolerance", 1e-5}, {"max_ms", 100}};   // :118-122
  w.make_ocp = [](int, int, int horizon) {
    mas_amd::OCP p;                                                             // x' = x + u, cost x'x + u'u, x0 = e1 (:21-76)
    p.state_dim = 4; p.control_dim = 4; p.dt = 0.1; p.horizon_steps = horizon;
    p.initial_state = mas_amd::State(4, 0.0);
    p.initial_state[0] = 1.0;
    p.model.model_id = MAS_HIP_MODEL_LQR_IDENTITY;
    return p;
  };
  return examples::run_example(w, argc, argv);
}
