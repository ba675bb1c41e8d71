// user_defined_model -- a workload that is NOT in the library: unicycles parking at the origin.  Not one of the reference's examples; it shows
// what a caller of the reference who writes an OCP of their own (dynamics / stage_cost / terminal_cost as lambdas, types.hpp:21-50) writes here:
// the same three callbacks as a struct, handed over as source (mas_amd::register_model -> mas_hip_register_model) -- every derivative is
// finite-differenced like the reference's empty hooks (ocp.hpp:117-135), strategies and flags as in the other examples.
#include <limits>

#include "example_utils.hpp"

static const char* kUnicycle = R"MODEL(
struct ParkingUnicycle : FdModel<ParkingUnicycle, 3, 2> {          // state: x, y, heading; control: speed, turn rate
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = -1.5; ub[0] = 1.5; lb[1] = -1.0; ub[1] = 1.0; }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams&) {
    d[0] = u[0] * mas_dt_cos(x[2]);
    d[1] = u[0] * mas_dt_sin(x[2]);
    d[2] = u[1];
  }
  MAS_HD static double stage_cost(const double* x, const double* u, int, const ModelParams& P) {
    return P.p[0] * (x[0] * x[0] + x[1] * x[1]) + P.p[1] * (u[0] * u[0] + u[1] * u[1]);
  }
  MAS_HD static double terminal_cost(const double* x, const ModelParams& P) { return P.p[2] * (x[0] * x[0] + x[1] * x[1]) + x[2] * x[2]; }
};
)MODEL";

int main(int argc, char** argv) {
  examples::Workload w;
  w.program = "user_defined_model";
  w.multi_agent = true;
  w.default_horizon = 30;
  w.params = {{"max_iterations", 100}, {"tolerance", 1e-5}, {"max_ms", std::numeric_limits<double>::infinity()}};
  w.make_ocp = [](int index, int, int horizon) {
    // registered at the first OCP (a minute of hipcc the first time, the cached plugin afterwards); --help and flag errors never get here
    static const mas_amd::ModelDescriptor model = mas_amd::register_model(kUnicycle, "ParkingUnicycle", std::vector<double>{1.0, 0.05, 20.0});
    mas_amd::OCP p;
    p.state_dim = 3; p.control_dim = 2; p.dt = 0.1; p.horizon_steps = horizon;
    p.initial_state = {2.0 + 0.5 * index, -1.0 + 0.25 * index * index, 0.125 * index};
    p.input_lower_bounds = mas_amd::Control{-1.5, -1.0};
    p.input_upper_bounds = mas_amd::Control{1.5, 1.0};
    p.model = model;
    return p;
  };
  return examples::run_example(w, argc, argv);
}
