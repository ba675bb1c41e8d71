// mas_models.hpp -- device-side workload models.
//
// The reference reaches problem-specific math through std::function callbacks on Eigen
// vectors (include/multi_agent_solver/types.hpp:21-50), which are opaque to a GPU.  Here a
// model is a POD parameter block plus static inline functions compiled into the kernels.
// Which derivative is finite-differenced and which is analytic follows the reference's
// example that defines the workload (OCP::initialize_problem installs FD defaults for every
// hook the example leaves empty, ocp.hpp:117-135).
//
// All arithmetic is FP64 and is compiled with -ffp-contract=off so that every expression
// rounds exactly like the scalar statement it restates.  sin/cos/tan come from
// include/mas_dettrig.h (pure IEEE + - * /), NOT from ocml: the algorithm is chaotic under
// last-bit differences (see that header), so the transcendental functions must be the same
// instruction sequence on the GPU and in the CPU oracle.
#pragma once

#include <cmath>

#include "mas_platform.hpp"
#include "../../include/mas_dettrig.h"   // bit-reproducible sin/cos/tan shared with the oracle's "det" mode

namespace mas {

struct ModelParams {
  double p[16];
};

// Function constraints h(x,u) = 0 (NEQ rows) and g(x,u) <= 0 (NIN rows), ocp.hpp:61-68.  No shipped reference
// example sets one; models without constraints inherit these empty hooks.
struct NoConstraints {
  static constexpr int NEQ = 0, NIN = 0;
  // true: the dynamics have a four-lanes-per-rollout RK4 (QuadRk4, mas_kernels_hip.hpp) for launches that do not fill the chip
  static constexpr bool kQuadRk4 = false;
  // > 0: the state components fall into that many DEPENDENCY LEVELS -- the RK4 increment of a component (rk4_increment, mas_lane.hpp)
  // depends only on the control and on components of LOWER levels (state_level(i)), never on the component itself or its peers --
  // so an open-loop rollout can be evaluated level by level, all time steps of a level side by side (mas_kernel_levels.hpp)
  static constexpr int kRolloutLevels = 0;
  MAS_HD static constexpr int state_level(int) { return 0; }
  // true: the model's analytic Jacobians are A = I, B = I at every point (NU == NX).  A product sum with a column of the identity is one factor
  // times 1 plus terms (+0.0) * v = +-0.0, i.e. the other factor's VALUE for finite operands, so A^T V, (A^T V) A, B^T V ... are V itself and a
  // kernel may skip the sums (the fused recursion does: mas_fused_hip.hpp; the other forms and the oracle form them, the results are held
  // against each other bit for bit)
  static constexpr bool kIdentityJacobians = false;
  // true: the per-agent line search rolls out the FIRST step size alone and the nine further ones only for agents whose first trial did not
  // decrease the merit (two launches; the select's decision is unchanged: it takes the first strict decrease in order).  For models whose full
  // step nearly always succeeds and whose rollout is bound by the bytes of its trial planes (the identity LQR: 131 MB per launch for ten trials)
  static constexpr bool kStagedLineSearch = false;
  MAS_HD static void equality(const double*, const double*, double*, const ModelParams&) {}
  MAS_HD static void inequality(const double*, const double*, double*, const ModelParams&) {}
};

// ---------------------------------------------------------------------------------------
// Base of a USER-DEFINED workload (mas_hip_register_model, include/mas_hip.h): like an OCP of the reference that sets only dynamics,
// stage_cost and terminal_cost (types.hpp:21-50) and leaves every derivative hook empty, so that OCP::initialize_problem installs the
// finite-difference defaults (ocp.hpp:117-135).  A model derives from FdModel<Self, NX, NU> and provides
//     MAS_HD static void   dynamics(const double* x, const double* u, double* dxdt, const ModelParams& P);
//     MAS_HD static double stage_cost(const double* x, const double* u, int t, const ModelParams& P);
//     MAS_HD static double terminal_cost(const double* x, const ModelParams& P);
// (P.p[0..15] are the problem's constants, mas_hip_problem_desc::params; sin / cos / tan as mas_dt_sin / mas_dt_cos / mas_dt_tan.)
// It may override what the built-in models set: default_bounds + kDefaultBounds, kZeroTerminal, Ctl / prepare / dynamics_c (the part of
// f(x, u) that depends on the zero-order-hold control alone, evaluated once per RK4 step), kRolloutLevels / state_level.
// The reference's OPTIONAL callbacks are members as well: a derivative hook (types.hpp:30-50) is `static constexpr bool kAnalyticA = true`
// (kAnalyticB, kAnalyticLx, kAnalyticLu, kAnalyticLxx, kAnalyticLuu, kAnalyticLux, kAnalyticTerm) with A(x, u, out, P), B, lx(x, u, t, out, P), lu,
// lxx, luu, lux, term_grad(x, out, P) + term_hess (column-major); function constraints (ocp.hpp:61-68) are NEQ / NIN with
// equality(x, u, out, P) and inequality(x, u, out, P).  Whatever is not declared stays finite-differenced / absent (ocp.hpp:117-171).
struct NoCtl {};
template <class Self, int NX_, int NU_>
struct FdModel : NoConstraints {
  static constexpr int NX = NX_, NU = NU_;
  static constexpr bool kAnalyticA = false, kAnalyticB = false, kAnalyticLx = false, kAnalyticLu = false,
                        kAnalyticLxx = false, kAnalyticLuu = false, kAnalyticLux = false, kAnalyticTerm = false;
  static constexpr bool kZeroTerminal = false;
  static constexpr bool kDefaultBounds = false;
  MAS_HD static void default_bounds(double*, double*) {}
  using Ctl = NoCtl;
  MAS_HD static Ctl prepare(const double*, const ModelParams&) { return Ctl{}; }
  MAS_HD static void dynamics_c(const double* x, const double* u, const Ctl&, double* d, const ModelParams& P) { Self::dynamics(x, u, d, P); }
};

// ---------------------------------------------------------------------------------------
// Single-track circular-track workload.
//   dynamics: examples/models/single_track_model.hpp:23-44
//   cost / bounds / constants: examples/multi_agent_single_track.cpp:46-67,103-110
//   no analytic hooks are set by that example (:46-69) => everything FD
// params: [0] track_radius [1] target_velocity [2] w_track [3] w_speed [4] w_delta [5] w_acc [6] L
struct SingleTrack : NoConstraints {
  static constexpr int NX = 4, NU = 2;
  static constexpr bool kAnalyticA = false, kAnalyticB = false, kAnalyticLx = false, kAnalyticLu = false,
                        kAnalyticLxx = false, kAnalyticLuu = false, kAnalyticLux = false, kAnalyticTerm = false;
  static constexpr bool kZeroTerminal = true;   // terminal_cost == 0 (:65): FD of it is exactly 0
  static constexpr bool kDefaultBounds = true;
  static constexpr bool kQuadRk4 = true;        // psi and v advance without sin/cos: the four RK4 stages' sin/cos can run side by side
  static constexpr int kRolloutLevels = 3;      // v <- a;  psi <- v, delta;  X, Y <- v, psi
  MAS_HD static constexpr int state_level(int i) { return i == 3 ? 0 : i == 2 ? 1 : 2; }
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = lb[1] = -0.5; ub[0] = ub[1] = 0.5; }

  // the part of f(x,u) that depends on u alone: RK4 evaluates f four times with the same zero-order-
  // hold control (integrator.hpp:22-25), so tan(delta) is computed once per step, not four times
  struct Ctl { double tan_delta; };
  MAS_HD static Ctl prepare(const double* u, const ModelParams&) { return Ctl{mas_dt_tan(u[0])}; }
  MAS_HD static void dynamics_c(const double* x, const double* u, const Ctl& c, double* d, const ModelParams& P) {
    const double psi = x[2], v = x[3], a = u[1], L = P.p[6];
    double sn, cs;
    mas_dt_sincos(psi, &sn, &cs);
    d[0] = v * cs;
    d[1] = v * sn;
    d[2] = div_by(v * c.tan_delta, make_inv_div(L));     // (v tan delta) / L; 1 / L is loop-invariant in every kernel
    d[3] = a;
  }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) { dynamics_c(x, u, prepare(u, P), d, P); }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {
    const double x = s[0], y = s[1], vx = s[3], delta = c[0], a_cmd = c[1];
    const double dist = fabs(sqrt(x * x + y * y) - P.p[0]);
    const double se = vx - P.p[1];
    return P.p[2] * dist * dist + P.p[3] * se * se + P.p[4] * delta * delta + P.p[5] * a_cmd * a_cmd;
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
};

// ---------------------------------------------------------------------------------------
// Identity LQR: examples/multi_agent_lqr.cpp:21-76 (A=B=Q=R=Qf=I, n_x=n_u=4), every
// derivative analytic (:35-41,46-71).

struct LqrIdentity : NoConstraints {
  static constexpr int NX = 4, NU = 4;
  static constexpr bool kIdentityJacobians = true;      // examples/multi_agent_lqr.cpp:35-41: A = B = I
  static constexpr bool kStagedLineSearch = true;
  using Ctl = NoCtl;
  MAS_HD static Ctl prepare(const double*, const ModelParams&) { return Ctl{}; }
  MAS_HD static void dynamics_c(const double* x, const double* u, const Ctl&, double* d, const ModelParams& P) { dynamics(x, u, d, P); }
  static constexpr bool kAnalyticA = true, kAnalyticB = true, kAnalyticLx = true, kAnalyticLu = true,
                        kAnalyticLxx = true, kAnalyticLuu = true, kAnalyticLux = true, kAnalyticTerm = true;
  static constexpr bool kZeroTerminal = false;
  static constexpr bool kDefaultBounds = false;
  MAS_HD static void default_bounds(double*, double*) {}

  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams&) {
    MAS_UNROLL for (int i = 0; i < NX; ++i) d[i] = x[i] + u[i];
  }
  MAS_HD static double stage_cost(const double* x, const double* u, int, const ModelParams&) {
    double a = 0.0, b = 0.0;
    MAS_UNROLL for (int i = 0; i < NX; ++i) a += x[i] * x[i];
    MAS_UNROLL for (int i = 0; i < NU; ++i) b += u[i] * u[i];
    return a + b;
  }
  MAS_HD static double terminal_cost(const double* x, const ModelParams&) {
    double a = 0.0;
    MAS_UNROLL for (int i = 0; i < NX; ++i) a += x[i] * x[i];
    return a;
  }
  MAS_HD static void A(const double*, const double*, double* M, const ModelParams&) {
    MAS_UNROLL for (int i = 0; i < NX * NX; ++i) M[i] = 0.0;
    MAS_UNROLL for (int i = 0; i < NX; ++i) M[i * NX + i] = 1.0;
  }
  MAS_HD static void B(const double*, const double*, double* M, const ModelParams&) {
    MAS_UNROLL for (int i = 0; i < NX * NU; ++i) M[i] = 0.0;
    MAS_UNROLL for (int i = 0; i < NU; ++i) M[i * NX + i] = 1.0;
  }
  MAS_HD static void lx(const double* x, const double*, int, double* g, const ModelParams&) { MAS_UNROLL for (int i = 0; i < NX; ++i) g[i] = 2.0 * x[i]; }
  MAS_HD static void lu(const double*, const double* u, int, double* g, const ModelParams&) { MAS_UNROLL for (int i = 0; i < NU; ++i) g[i] = 2.0 * u[i]; }
  MAS_HD static void lxx(const double*, const double*, int, double* H, const ModelParams&) {
    MAS_UNROLL for (int i = 0; i < NX * NX; ++i) H[i] = 0.0;
    MAS_UNROLL for (int i = 0; i < NX; ++i) H[i * NX + i] = 2.0;
  }
  MAS_HD static void luu(const double*, const double*, int, double* H, const ModelParams&) {
    MAS_UNROLL for (int i = 0; i < NU * NU; ++i) H[i] = 0.0;
    MAS_UNROLL for (int i = 0; i < NU; ++i) H[i * NU + i] = 2.0;
  }
  MAS_HD static void lux(const double*, const double*, int, double* H, const ModelParams&) { MAS_UNROLL for (int i = 0; i < NU * NX; ++i) H[i] = 0.0; }
  MAS_HD static void term_grad(const double* x, double* g, const ModelParams&) { MAS_UNROLL for (int i = 0; i < NX; ++i) g[i] = 2.0 * x[i]; }
  MAS_HD static void term_hess(const double*, double* H, const ModelParams&) {
    MAS_UNROLL for (int i = 0; i < NX * NX; ++i) H[i] = 0.0;
    MAS_UNROLL for (int i = 0; i < NX; ++i) H[i * NX + i] = 2.0;
  }
};

// ---------------------------------------------------------------------------------------
// Pendulum swing-up: examples/models/pendulum_model.hpp:8-20, examples/pendulum_swing_up.cpp:29-117.
// No analytic hooks are wired by the example => FD.  params: [0] horizon_d (the T in s = k/(T-1)).
struct Pendulum : NoConstraints {
  static constexpr int NX = 2, NU = 1;
  using Ctl = NoCtl;
  MAS_HD static Ctl prepare(const double*, const ModelParams&) { return Ctl{}; }
  MAS_HD static void dynamics_c(const double* x, const double* u, const Ctl&, double* d, const ModelParams& P) { dynamics(x, u, d, P); }
  static constexpr bool kAnalyticA = false, kAnalyticB = false, kAnalyticLx = false, kAnalyticLu = false,
                        kAnalyticLxx = false, kAnalyticLuu = false, kAnalyticLux = false, kAnalyticTerm = false;
  static constexpr bool kZeroTerminal = false;
  static constexpr bool kDefaultBounds = true;
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = -5.0; ub[0] = 5.0; }

  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams&) {
    const double g = 9.81, l = 1.0, m = 1.0, b = 0.1;
    d[0] = x[1];
    d[1] = (g / l) * mas_dt_sin(x[0]) + u[0] / (m * l * l) - (b / (m * l * l)) * x[1];
  }
  MAS_HD static double stage_cost(const double* x, const double* u, int k, const ModelParams& P) {
    const double g = 9.81, l = 1.0, m = 1.0, mgl = m * g * l, E_des = mgl;
    const double w_energy = 2.0, w_u = 0.05, w_shape = 2.0, w_omega = 0.05;
    const double horizon_d = P.p[0];
    const double theta = x[0], omega = x[1], torque = u[0];
    const double s = static_cast<double>(k) / (horizon_d - 1.0);
    const double late = s * s, early = 1.0 - late;
    const double w_energy_k = w_energy * (0.2 + 0.8 * early);
    const double w_shape_k = w_shape * (0.2 + 0.8 * late);
    const double w_omega_k = w_omega * (0.2 + 0.8 * late);
    const double Tk = 0.5 * m * l * l * omega * omega;
    const double V = mgl * mas_dt_cos(theta);
    const double E = Tk + V;
    const double energy_error = (E - E_des) / mgl;
    const double upright_error = 1.0 - mas_dt_cos(theta);
    return w_energy_k * energy_error * energy_error + w_shape_k * upright_error + w_omega_k * omega * omega + w_u * torque * torque;
  }
  MAS_HD static double terminal_cost(const double* x, const ModelParams&) {
    const double wT_pos = 500.0, wT_vel = 100.0;
    const double upright_error = 1.0 - mas_dt_cos(x[0]);
    return wT_pos * upright_error + wT_vel * x[1] * x[1];
  }
};

// ---------------------------------------------------------------------------------------
// Single-track lane following: examples/single_track_ocp.cpp:14-116.  Analytic A,B
// (models/single_track_model.hpp:49-82) and l_x,l_u,l_xx,l_uu (:62-94); the cross term and
// the (zero) terminal derivatives stay FD.
// params: [0] desired_velocity [1] w_lane [2] w_speed [3] w_delta [4] w_acc [6] L
struct SingleTrackLane : NoConstraints {
  static constexpr int NX = 4, NU = 2;
  static constexpr bool kAnalyticA = true, kAnalyticB = true, kAnalyticLx = true, kAnalyticLu = true,
                        kAnalyticLxx = true, kAnalyticLuu = true, kAnalyticLux = false, kAnalyticTerm = false;
  static constexpr bool kZeroTerminal = true;
  static constexpr bool kDefaultBounds = true;
  static constexpr bool kQuadRk4 = true;        // SingleTrack's dynamics
  static constexpr int kRolloutLevels = SingleTrack::kRolloutLevels;
  MAS_HD static constexpr int state_level(int i) { return SingleTrack::state_level(i); }
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = -0.7; lb[1] = -1.0; ub[0] = 0.7; ub[1] = 1.0; }

  using Ctl = SingleTrack::Ctl;
  MAS_HD static Ctl prepare(const double* u, const ModelParams& P) { return SingleTrack::prepare(u, P); }
  MAS_HD static void dynamics_c(const double* x, const double* u, const Ctl& c, double* d, const ModelParams& P) { SingleTrack::dynamics_c(x, u, c, d, P); }
  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) { SingleTrack::dynamics(x, u, d, P); }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {
    const double y = s[1], vx = s[3], delta = c[0], a_cmd = c[1];
    const double lane_error = y, speed_error = vx - P.p[0];
    return P.p[1] * (lane_error * lane_error) + P.p[2] * (speed_error * speed_error) + P.p[3] * (delta * delta) + P.p[4] * (a_cmd * a_cmd);
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
  MAS_HD static void A(const double* x, const double* u, double* M, const ModelParams& P) {
    const double psi = x[2], v = x[3], delta = u[0], L = P.p[6];
    MAS_UNROLL for (int i = 0; i < 16; ++i) M[i] = 0.0;
    M[2 * 4 + 0] = -v * mas_dt_sin(psi); M[3 * 4 + 0] = mas_dt_cos(psi);
    M[2 * 4 + 1] = v * mas_dt_cos(psi);  M[3 * 4 + 1] = mas_dt_sin(psi);
    M[3 * 4 + 2] = mas_dt_tan(delta) / L;
  }
  MAS_HD static void B(const double* x, const double* u, double* M, const ModelParams& P) {
    const double v = x[3], delta = u[0], L = P.p[6];
    MAS_UNROLL for (int i = 0; i < 8; ++i) M[i] = 0.0;
    M[0 * 4 + 2] = v / (L * mas_dt_cos(delta) * mas_dt_cos(delta));
    M[1 * 4 + 3] = 1.0;
  }
  MAS_HD static void lx(const double* x, const double*, int, double* g, const ModelParams& P) { g[0] = 0.0; g[1] = 2.0 * P.p[1] * x[1]; g[2] = 0.0; g[3] = 2.0 * P.p[2] * (x[3] - P.p[0]); }
  MAS_HD static void lu(const double*, const double* u, int, double* g, const ModelParams& P) { g[0] = 2.0 * P.p[3] * u[0]; g[1] = 2.0 * P.p[4] * u[1]; }
  MAS_HD static void lxx(const double*, const double*, int, double* H, const ModelParams& P) {
    MAS_UNROLL for (int i = 0; i < 16; ++i) H[i] = 0.0;
    H[1 * 4 + 1] = 2.0 * P.p[1]; H[3 * 4 + 3] = 2.0 * P.p[2];
  }
  MAS_HD static void luu(const double*, const double*, int, double* H, const ModelParams& P) { H[0] = 2.0 * P.p[3]; H[1] = 0.0; H[2] = 0.0; H[3] = 2.0 * P.p[4]; }
};

// ---------------------------------------------------------------------------------------
// bicycle6: THIS REPOSITORY'S OWN 6-state model for BASELINE config 5 (the reference has no
// nx=6 model).  Kinematic bicycle with steering angle and acceleration as
// states and their rates as controls; circular-track cost as in the single-track example.
//   state [X, Y, psi, v, delta, a], control [delta_rate, jerk]
// params: [0] R [1] vt [2] w_track [3] w_speed [4] w_delta [5] w_acc [6] L [7] w_rate
struct Bicycle6 : NoConstraints {
  static constexpr int NX = 6, NU = 2;
  using Ctl = NoCtl;
  MAS_HD static Ctl prepare(const double*, const ModelParams&) { return Ctl{}; }
  MAS_HD static void dynamics_c(const double* x, const double* u, const Ctl&, double* d, const ModelParams& P) { dynamics(x, u, d, P); }
  static constexpr bool kAnalyticA = false, kAnalyticB = false, kAnalyticLx = false, kAnalyticLu = false,
                        kAnalyticLxx = false, kAnalyticLuu = false, kAnalyticLux = false, kAnalyticTerm = false;
  static constexpr bool kZeroTerminal = true;
  static constexpr bool kDefaultBounds = true;
  static constexpr bool kQuadRk4 = true;        // a, delta, v, psi of every RK4 stage follow from x and u without sin/cos
  static constexpr int kRolloutLevels = 4;      // delta, a <- u;  v <- a;  psi <- v, delta;  X, Y <- v, psi
  MAS_HD static constexpr int state_level(int i) { return i >= 4 ? 0 : i == 3 ? 1 : i == 2 ? 2 : 3; }
  MAS_HD static void default_bounds(double* lb, double* ub) { lb[0] = lb[1] = -0.5; ub[0] = ub[1] = 0.5; }

  MAS_HD static void dynamics(const double* x, const double* u, double* d, const ModelParams& P) {
    const double psi = x[2], v = x[3], delta = x[4], a = x[5], L = P.p[6];
    double sn, cs;
    mas_dt_sincos(psi, &sn, &cs);
    d[0] = v * cs; d[1] = v * sn; d[2] = div_by(v * mas_dt_tan_onediv(delta), make_inv_div(L)); d[3] = a; d[4] = u[0]; d[5] = u[1];   // (mas_dt_tan_onediv == mas_dt_tan bit for bit, mas_dettrig.h)
  }
  MAS_HD static double stage_cost(const double* s, const double* c, int, const ModelParams& P) {
    const double x = s[0], y = s[1], vx = s[3], delta = s[4], a = s[5];
    const double dist = fabs(sqrt(x * x + y * y) - P.p[0]);
    const double se = vx - P.p[1];
    return P.p[2] * dist * dist + P.p[3] * se * se + P.p[4] * delta * delta + P.p[5] * a * a + P.p[7] * c[0] * c[0] + P.p[7] * c[1] * c[1];
  }
  MAS_HD static double terminal_cost(const double*, const ModelParams&) { return 0.0; }
};

// ---------------------------------------------------------------------------------------
// single track + function constraints: THIS REPOSITORY'S OWN workload for the augmented-Lagrangian branch of
// iLQR (ilqr.hpp:121-170,236-260,379-407).  Dynamics, cost, bounds of SingleTrack plus
//   h(x,u) = [ delta - p[9]*a ]        g(x,u) = [ v - p[8], -v ]
struct SingleTrackLimited : SingleTrack {
  static constexpr int NEQ = 1, NIN = 2;
  MAS_HD static void equality(const double*, const double* u, double* out, const ModelParams& P) { out[0] = u[0] - P.p[9] * u[1]; }
  MAS_HD static void inequality(const double* x, const double*, double* out, const ModelParams& P) { out[0] = x[3] - P.p[8]; out[1] = -x[3]; }
};

// number of doubles the linearisation of one (agent, t) produces: A,B,l_x,l_u,l_xx,l_uu,l_ux
template <class M>
struct DerivLayout {
  static constexpr int NX = M::NX, NU = M::NU;
  static constexpr int NEQ = M::NEQ, NIN = M::NIN;
  static constexpr int oA = 0, oB = oA + NX * NX, oLx = oB + NX * NU, oLu = oLx + NX, oLxx = oLu + NU,
                       oLuu = oLxx + NX * NX, oLux = oLuu + NU * NU, oCore = oLux + NU * NX,
                       // constraint residuals and Jacobians (column-major m x nx, m x nu), only for constrained models
                       oEq = oCore, oEqJx = oEq + NEQ, oEqJu = oEqJx + NEQ * NX, oIn = oEqJu + NEQ * NU, oInJx = oIn + NIN,
                       oInJu = oInJx + NIN * NX, ND = oInJu + NIN * NU;
};

inline void model_default_params(int model_id, int T, double* p) {
  for (int i = 0; i < 16; ++i) p[i] = 0.0;
  switch (model_id) {
    case 1: p[0] = 20.0; p[1] = 5.0; p[2] = 1.0; p[3] = 1.0; p[4] = 0.001; p[5] = 0.001; p[6] = 2.5; break;  // multi_agent_single_track.cpp:49-52,109-110
    case 3: p[0] = static_cast<double>(T); break;                                                       // pendulum_swing_up.cpp:62
    case 4: p[0] = 1.0; p[1] = 10.0; p[2] = 1.0; p[3] = 0.1; p[4] = 0.1; p[6] = 2.5; break;              // single_track_ocp.cpp:34-40
    case 5: p[0] = 20.0; p[1] = 5.0; p[2] = 1.0; p[3] = 1.0; p[4] = 0.001; p[5] = 0.001; p[6] = 2.5; p[7] = 0.001; break;
    case 6: p[0] = 20.0; p[1] = 5.0; p[2] = 1.0; p[3] = 1.0; p[4] = 0.001; p[5] = 0.001; p[6] = 2.5; p[8] = 4.5; p[9] = 0.1; break;
    default: break;
  }
}

}  // namespace mas
