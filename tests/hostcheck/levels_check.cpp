// TEST-ONLY (g++, -DMAS_HOST_EMULATION): the dependency levels the vehicle models declare (M::state_level, mas_models.hpp) against the
// serial rollout.  mas_kernel_levels.hpp evaluates an open-loop rollout level by level -- the RK4 increments of all time steps of a level
// from the components of lower levels only, then x_{t+1} = x_t + inc_t serially -- which is bit-identical to integrate_horizon exactly
// when the declared levels are true.  Checked here on random trajectories, with the unknown components replaced by GARBAGE (not zero).
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../multi_agent_solver_amd/csrc/mas_lane.hpp"

using namespace mas;

template <class M>
int check(const char* name, int model_id, int T, double dt, unsigned seed) {
  constexpr int NX = M::NX, NU = M::NU;
  static_assert(M::kRolloutLevels > 0, "model without levels");
  ModelParams P; model_default_params(model_id, T, P.p);
  std::mt19937_64 rng(seed);
  std::uniform_real_distribution<double> ux(-1.0, 1.0);
  int bad = 0;
  for (int rep = 0; rep < 50; ++rep) {
    std::vector<double> X((T + 1) * NX), Xl((T + 1) * NX), U(T * NU);
    for (int i = 0; i < NX; ++i) X[i] = Xl[i] = 5.0 * ux(rng);
    for (auto& u : U) u = 0.5 * ux(rng);
    double c = 0.0;
    for (int t = 0; t < T; ++t) {                       // integrate_horizon + objective, the serial statement
      c += M::stage_cost(&X[t * NX], &U[t * NU], t, P);
      rk4_step<M>(&X[t * NX], &U[t * NU], dt, &X[(t + 1) * NX], P);
    }
    c += M::terminal_cost(&X[T * NX], P);
    for (int lev = 0; lev < M::kRolloutLevels; ++lev) {
      std::vector<double> inc(T * NX);
      for (int t = 0; t < T; ++t) {
        double x[NX];
        for (int i = 0; i < NX; ++i) x[i] = (M::state_level(i) < lev) ? Xl[t * NX + i] : 1e3 * ux(rng);     // garbage where the level may not look
        rk4_increment<M>(x, &U[t * NU], dt, &inc[t * NX], P);
      }
      for (int i = 0; i < NX; ++i) {
        if (M::state_level(i) != lev) continue;
        for (int t = 0; t < T; ++t) Xl[(t + 1) * NX + i] = Xl[t * NX + i] + inc[t * NX + i];
      }
    }
    double cl = 0.0;
    for (int t = 0; t < T; ++t) cl += M::stage_cost(&Xl[t * NX], &U[t * NU], t, P);
    cl += M::terminal_cost(&Xl[T * NX], P);
    if (std::memcmp(X.data(), Xl.data(), sizeof(double) * X.size()) != 0 || std::memcmp(&c, &cl, sizeof(double)) != 0) ++bad;
  }
  std::printf("%s: %d levels, 50 random rollouts of %d steps, %d mismatches\n", name, M::kRolloutLevels, T, bad);
  return bad;
}

int main() {
  int bad = 0;
  bad += check<SingleTrack>("single_track", 1, 100, 0.5, 1u);
  bad += check<SingleTrackLane>("single_track_lane", 4, 60, 0.1, 2u);
  bad += check<Bicycle6>("bicycle6", 5, 200, 0.25, 3u);
  std::printf("levels_check: %d mismatches\n", bad);
  return bad == 0 ? 0 : 1;
}
