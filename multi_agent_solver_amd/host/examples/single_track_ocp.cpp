// single_track_ocp -- one car following a lane, analytic dynamics Jacobians and cost gradients / Hessians, the cross term
// finite-differenced (reference: examples/single_track_ocp.cpp; the only workload with a published iLQR cost, README.md:167 --
// which today's sources do not reproduce: both CPU restatements give 508.593060, tests/golden/reference_readme_costs.json).
#include "example_utils.hpp"

int main(int argc, char** argv) {
  examples::Workload w;
  w.program = "single_track_ocp";
  w.multi_agent = false;
  w.single_label = "single_track";
  w.default_horizon = 80;                                                       // :24
  w.params = {{"max_iterations", 10}, {"tolerance", 1e-5}, {"max_ms", 100}};    // :147-150
  w.make_ocp = [](int, int, int horizon) {
    mas_amd::OCP p;                                                             // create_single_track_lane_following_ocp, :14-116
    p.state_dim = 4; p.control_dim = 2; p.horizon_steps = horizon; p.dt = 0.1;
    p.initial_state = {0.0, 1.0, 0.0, 0.0};                                     // one metre off the lane, at rest
    p.model.model_id = MAS_HIP_MODEL_SINGLE_TRACK_LANE;                         // desired velocity 1, weights 10 / 1 / 0.1 / 0.1: the model's defaults
    p.input_lower_bounds = mas_amd::Control{-0.7, -1.0};
    p.input_upper_bounds = mas_amd::Control{0.7, 1.0};
    return p;
  };
  return examples::run_example(w, argc, argv);
}
