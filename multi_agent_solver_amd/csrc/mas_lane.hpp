// mas_lane.hpp -- per-lane FP64 math of the batched iLQR path.
//
// One "lane" owns one agent (or one agent x time step, or one agent x alpha trial); all state
// lives in registers as small fixed-size arrays indexed by compile-time constants after full
// unrolling.  Each function restates one piece of the reference and keeps the reference's
// operation order so results agree with a scalar CPU evaluation to the last bit wherever the
// math library agrees (sin/cos/tan differ by <= a few ulp between glibc and ROCm's ocml).
#pragma once

#include "mas_models.hpp"

namespace mas {

// ------------------------------------------------------------------ small helpers ------
// v with component i replaced by v[i] + se.  Eigen adds a whole vector (x + dx) so untouched
// components see "+ 0.0"; that only differs for -0.0, which none of the models can observe.
template <int N>
MAS_HD void perturb1(const double* v, int i, double se, double* out) {
  MAS_UNROLL for (int k = 0; k < N; ++k) out[k] = (k == i) ? v[k] + se : v[k];
}
template <int N>
MAS_HD void perturb2(const double* v, int i, double si, int j, double sj, double* out) {
  MAS_UNROLL for (int k = 0; k < N; ++k) {
    const double w = (k == i) ? v[k] + si : v[k];
    out[k] = (k == j) ? w + sj : w;
  }
}
MAS_HD double safe_value(double v) { return std::isfinite(v) ? v : 0.0; }   // finite_differences.hpp:95-107

// integrate_rk4, integrator.hpp:19-28, in two statements: the increment (dt/6)(k1 + 2 k2 + 2 k3 + k4) and its addition to x.  (Nothing
// is contracted, so `x + inc` is the reference's `x + (dt/6) * (...)` bit for bit.)  The increment alone is what the rollout by
// dependency levels needs (mas_kernel_levels.hpp): for the vehicle models the increment of a state component depends only on
// components of lower levels, never on the component itself.
template <class M>
MAS_HD void rk4_increment(const double* x, const double* u, double dt, double* inc, const ModelParams& P) {
  constexpr int NX = M::NX;
  double k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
  const typename M::Ctl c = M::prepare(u, P);      // u is held over the step: its own terms once
  M::dynamics_c(x, u, c, k1, P);
  MAS_UNROLL for (int i = 0; i < NX; ++i) xs[i] = x[i] + (0.5 * dt) * k1[i];
  M::dynamics_c(xs, u, c, k2, P);
  MAS_UNROLL for (int i = 0; i < NX; ++i) xs[i] = x[i] + (0.5 * dt) * k2[i];
  M::dynamics_c(xs, u, c, k3, P);
  MAS_UNROLL for (int i = 0; i < NX; ++i) xs[i] = x[i] + dt * k3[i];
  M::dynamics_c(xs, u, c, k4, P);
  MAS_UNROLL for (int i = 0; i < NX; ++i) inc[i] = (dt / 6.0) * (((k1[i] + 2.0 * k2[i]) + 2.0 * k3[i]) + k4[i]);
}
template <class M>
MAS_HD void rk4_step(const double* x, const double* u, double dt, double* out, const ModelParams& P) {
  constexpr int NX = M::NX;
  double inc[NX];
  rk4_increment<M>(x, u, dt, inc, P);
  MAS_UNROLL for (int i = 0; i < NX; ++i) out[i] = x[i] + inc[i];
}

// clamp_controls on one column, constraint_helpers.hpp:107-114: cwiseMin(upper).cwiseMax(lower)
template <int NU>
MAS_HD void clamp_column(double* u, const double* lb, const double* ub) {
  MAS_UNROLL for (int i = 0; i < NU; ++i) {
    double v = u[i];
    v = (ub[i] < v) ? ub[i] : v;
    v = (v < lb[i]) ? lb[i] : v;
    u[i] = v;
  }
}

// ------------------------------------------------------------------ linearisation ------
// One (agent, t): ilqr.hpp:106-113 through the hooks OCP::initialize_problem installs
// (ocp.hpp:117-135): central differences of finite_differences.hpp:53-287 unless the
// workload's example wires an analytic hook.  D is laid out as DerivLayout<M>.
// PARTS: which blocks of the tile to evaluate (the others are left untouched): the helper wavefronts of the fused kernel share the
// FIRST chunks of a launch out among themselves by block, so that the recursion can start sooner (mas_fused_hip.hpp)
enum LinPart : int { kLinA = 1, kLinB = 2, kLinLx = 4, kLinLu = 8, kLinLxx = 16, kLinLuu = 32, kLinLux = 64, kLinConstraints = 128, kLinAll = 255 };
template <class M, int PARTS = kLinAll>
MAS_HD void linearise(const double* x, const double* u, int t, double* D, const ModelParams& P) {
  using L = DerivLayout<M>;
  constexpr int NX = M::NX, NU = M::NU;
  double xp[NX], xm[NX], up[NU], um[NU], fp[NX], fm[NX];

  if constexpr (!(PARTS & kLinA)) {
  } else if constexpr (M::kAnalyticA) {
    M::A(x, u, D + L::oA, P);
  } else {                                              // finite_differences.hpp:53-71, eps = 1e-6
    const double eps = 1e-6;
    MAS_UNROLL for (int i = 0; i < NX; ++i) {
      perturb1<NX>(x, i, +eps, xp); M::dynamics(xp, u, fp, P);
      perturb1<NX>(x, i, -eps, xm); M::dynamics(xm, u, fm, P);
      MAS_UNROLL for (int r = 0; r < NX; ++r) D[L::oA + i * NX + r] = div_by(fp[r] - fm[r], make_inv_div(2 * eps));
    }
  }
  if constexpr (!(PARTS & kLinB)) {
  } else if constexpr (M::kAnalyticB) {
    M::B(x, u, D + L::oB, P);
  } else {                                              // :73-92
    const double eps = 1e-6;
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      perturb1<NU>(u, i, +eps, up); M::dynamics(x, up, fp, P);
      perturb1<NU>(u, i, -eps, um); M::dynamics(x, um, fm, P);
      MAS_UNROLL for (int r = 0; r < NX; ++r) D[L::oB + i * NX + r] = div_by(fp[r] - fm[r], make_inv_div(2 * eps));
    }
  }
  if constexpr (!(PARTS & kLinLx)) {
  } else if constexpr (M::kAnalyticLx) {
    M::lx(x, u, t, D + L::oLx, P);
  } else {                                              // :110-122 (no safe_eval on gradients)
    const double eps = 1e-6;
    MAS_UNROLL for (int i = 0; i < NX; ++i) {
      perturb1<NX>(x, i, +eps, xp); perturb1<NX>(x, i, -eps, xm);
      D[L::oLx + i] = div_by(M::stage_cost(xp, u, t, P) - M::stage_cost(xm, u, t, P), make_inv_div(2 * eps));
    }
  }
  if constexpr (!(PARTS & kLinLu)) {
  } else if constexpr (M::kAnalyticLu) {
    M::lu(x, u, t, D + L::oLu, P);
  } else {                                              // :124-136
    const double eps = 1e-6;
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      perturb1<NU>(u, i, +eps, up); perturb1<NU>(u, i, -eps, um);
      D[L::oLu + i] = div_by(M::stage_cost(x, up, t, P) - M::stage_cost(x, um, t, P), make_inv_div(2 * eps));
    }
  }
  if constexpr (!(PARTS & kLinLxx)) {
  } else if constexpr (M::kAnalyticLxx) {
    M::lxx(x, u, t, D + L::oLxx, P);
  } else {                                              // :138-173, eps = 1e-5, (i,j) and (j,i) separately
    const double eps = 1e-5;
    MAS_UNROLL for (int i = 0; i < NX; ++i) {
      perturb1<NX>(x, i, +eps, xp); perturb1<NX>(x, i, -eps, xm);
      const double f_plus = safe_value(M::stage_cost(xp, u, t, P));
      const double f = safe_value(M::stage_cost(x, u, t, P));
      const double f_minus = safe_value(M::stage_cost(xm, u, t, P));
      D[L::oLxx + i * NX + i] = div_by(f_plus - 2 * f + f_minus, make_inv_div(eps * eps));
    }
    MAS_UNROLL for (int i = 0; i < NX; ++i) {
      MAS_UNROLL for (int j = 0; j < NX; ++j) {
        if (i != j) {
          perturb2<NX>(x, i, +eps, j, +eps, xp); const double f_pp = safe_value(M::stage_cost(xp, u, t, P));
          perturb2<NX>(x, i, +eps, j, -eps, xp); const double f_pm = safe_value(M::stage_cost(xp, u, t, P));
          perturb2<NX>(x, i, -eps, j, +eps, xp); const double f_mp = safe_value(M::stage_cost(xp, u, t, P));
          perturb2<NX>(x, i, -eps, j, -eps, xp); const double f_mm = safe_value(M::stage_cost(xp, u, t, P));
          D[L::oLxx + j * NX + i] = div_by(f_pp - f_pm - f_mp + f_mm, make_inv_div(4 * eps * eps));
        }
      }
    }
  }
  if constexpr (!(PARTS & kLinLuu)) {
  } else if constexpr (M::kAnalyticLuu) {
    M::luu(x, u, t, D + L::oLuu, P);
  } else {                                              // :175-210
    const double eps = 1e-5;
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      perturb1<NU>(u, i, +eps, up); perturb1<NU>(u, i, -eps, um);
      const double f_plus = safe_value(M::stage_cost(x, up, t, P));
      const double f = safe_value(M::stage_cost(x, u, t, P));
      const double f_minus = safe_value(M::stage_cost(x, um, t, P));
      D[L::oLuu + i * NU + i] = div_by(f_plus - 2 * f + f_minus, make_inv_div(eps * eps));
    }
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      MAS_UNROLL for (int j = 0; j < NU; ++j) {
        if (i != j) {
          perturb2<NU>(u, i, +eps, j, +eps, up); const double f_pp = safe_value(M::stage_cost(x, up, t, P));
          perturb2<NU>(u, i, +eps, j, -eps, up); const double f_pm = safe_value(M::stage_cost(x, up, t, P));
          perturb2<NU>(u, i, -eps, j, +eps, up); const double f_mp = safe_value(M::stage_cost(x, up, t, P));
          perturb2<NU>(u, i, -eps, j, -eps, up); const double f_mm = safe_value(M::stage_cost(x, up, t, P));
          D[L::oLuu + j * NU + i] = div_by(f_pp - f_pm - f_mp + f_mm, make_inv_div(4 * eps * eps));
        }
      }
    }
  }
  if constexpr (!(PARTS & kLinLux)) {
  } else if constexpr (M::kAnalyticLux) {
    M::lux(x, u, t, D + L::oLux, P);
  } else {                                              // :263-287, eps = 1e-6, H is nu x nx
    const double eps = 1e-6;
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      MAS_UNROLL for (int j = 0; j < NX; ++j) {
        perturb1<NU>(u, i, +eps, up); perturb1<NU>(u, i, -eps, um);
        perturb1<NX>(x, j, +eps, xp); perturb1<NX>(x, j, -eps, xm);
        const double f_pp = safe_value(M::stage_cost(xp, up, t, P));
        const double f_pm = safe_value(M::stage_cost(xm, up, t, P));
        const double f_mp = safe_value(M::stage_cost(xp, um, t, P));
        const double f_mm = safe_value(M::stage_cost(xm, um, t, P));
        D[L::oLux + j * NU + i] = div_by(f_pp - f_pm - f_mp + f_mm, make_inv_div(4 * eps * eps));
      }
    }
  }
  // function constraints: residual and central-difference Jacobians (finite_differences.hpp:289-345, eps = 1e-6)
  if constexpr (M::NEQ > 0 && (PARTS & kLinConstraints)) {
    const double eps = 1e-6;
    double cp[M::NEQ], cm[M::NEQ];
    M::equality(x, u, D + L::oEq, P);
    MAS_UNROLL for (int i = 0; i < NX; ++i) {
      perturb1<NX>(x, i, +eps, xp); M::equality(xp, u, cp, P);
      perturb1<NX>(x, i, -eps, xm); M::equality(xm, u, cm, P);
      MAS_UNROLL for (int r = 0; r < M::NEQ; ++r) D[L::oEqJx + i * M::NEQ + r] = div_by(cp[r] - cm[r], make_inv_div(2 * eps));
    }
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      perturb1<NU>(u, i, +eps, up); M::equality(x, up, cp, P);
      perturb1<NU>(u, i, -eps, um); M::equality(x, um, cm, P);
      MAS_UNROLL for (int r = 0; r < M::NEQ; ++r) D[L::oEqJu + i * M::NEQ + r] = div_by(cp[r] - cm[r], make_inv_div(2 * eps));
    }
  }
  if constexpr (M::NIN > 0 && (PARTS & kLinConstraints)) {
    const double eps = 1e-6;
    double cp[M::NIN], cm[M::NIN];
    M::inequality(x, u, D + L::oIn, P);
    MAS_UNROLL for (int i = 0; i < NX; ++i) {
      perturb1<NX>(x, i, +eps, xp); M::inequality(xp, u, cp, P);
      perturb1<NX>(x, i, -eps, xm); M::inequality(xm, u, cm, P);
      MAS_UNROLL for (int r = 0; r < M::NIN; ++r) D[L::oInJx + i * M::NIN + r] = div_by(cp[r] - cm[r], make_inv_div(2 * eps));
    }
    MAS_UNROLL for (int i = 0; i < NU; ++i) {
      perturb1<NU>(u, i, +eps, up); M::inequality(x, up, cp, P);
      perturb1<NU>(u, i, -eps, um); M::inequality(x, um, cm, P);
      MAS_UNROLL for (int r = 0; r < M::NIN; ++r) D[L::oInJu + i * M::NIN + r] = div_by(cp[r] - cm[r], make_inv_div(2 * eps));
    }
  }
}

// in-place  V = 0.5 * (V + V^T)  in the order Eigen executes the reference's aliased
// statement (ilqr.hpp:102,192): scalar column-major traversal without a temporary, so the
// lower entry is averaged from old values and the mirrored upper entry then reads the
// already-updated lower one.
template <int N>
MAS_HD void symmetrise_inplace(double* V) {
  MAS_UNROLL for (int j = 0; j < N; ++j) {
    MAS_UNROLL for (int i = 0; i < N; ++i) V[j * N + i] = 0.5 * (V[j * N + i] + V[i * N + j]);
  }
}

// terminal boundary condition, ilqr.hpp:92-102
template <class M>
MAS_HD void terminal_boundary(const double* xT, double* vx, double* vxx, const ModelParams& P) {
  constexpr int NX = M::NX;
  if constexpr (M::kAnalyticTerm) {
    M::term_grad(xT, vx, P);
    M::term_hess(xT, vxx, P);
  } else if constexpr (M::kZeroTerminal) {
    // FD of an identically-zero terminal cost is exactly zero: (0-0)/(2 eps), (0-2*0+0)/eps^2
    MAS_UNROLL for (int i = 0; i < NX; ++i) vx[i] = 0.0;
    MAS_UNROLL for (int i = 0; i < NX * NX; ++i) vxx[i] = 0.0;
  } else {
    double xp[NX], xm[NX];
    {                                                   // finite_differences.hpp:212-224
      const double eps = 1e-6;
      MAS_UNROLL for (int i = 0; i < NX; ++i) {
        perturb1<NX>(xT, i, +eps, xp); perturb1<NX>(xT, i, -eps, xm);
        vx[i] = div_by(M::terminal_cost(xp, P) - M::terminal_cost(xm, P), make_inv_div(2 * eps));
      }
    }
    {                                                   // :226-261
      const double eps = 1e-5;
      MAS_UNROLL for (int i = 0; i < NX; ++i) {
        perturb1<NX>(xT, i, +eps, xp); perturb1<NX>(xT, i, -eps, xm);
        const double f_plus = safe_value(M::terminal_cost(xp, P));
        const double f = safe_value(M::terminal_cost(xT, P));
        const double f_minus = safe_value(M::terminal_cost(xm, P));
        vxx[i * NX + i] = div_by(f_plus - 2 * f + f_minus, make_inv_div(eps * eps));
      }
      MAS_UNROLL for (int i = 0; i < NX; ++i) {
        MAS_UNROLL for (int j = 0; j < NX; ++j) {
          if (i != j) {
            perturb2<NX>(xT, i, +eps, j, +eps, xp); const double f_pp = safe_value(M::terminal_cost(xp, P));
            perturb2<NX>(xT, i, +eps, j, -eps, xp); const double f_pm = safe_value(M::terminal_cost(xp, P));
            perturb2<NX>(xT, i, -eps, j, +eps, xp); const double f_mp = safe_value(M::terminal_cost(xp, P));
            perturb2<NX>(xT, i, -eps, j, -eps, xp); const double f_mm = safe_value(M::terminal_cost(xp, P));
            vxx[j * NX + i] = div_by(f_pp - f_pm - f_mp + f_mm, make_inv_div(4 * eps * eps));
          }
        }
      }
    }
  }
  symmetrise_inplace<NX>(vxx);
}

// ------------------------------------------------------------------ backward step ------
// C(m x n) = A^T B with A stored p x m, B stored p x n (column-major), sums in k order from 0.0
template <int P_, int M_, int N_>
MAS_HD void mat_AtB(const double* A, const double* B, double* C) {
  MAS_UNROLL for (int j = 0; j < N_; ++j) {
    MAS_UNROLL for (int i = 0; i < M_; ++i) {
      double s = 0.0;
      MAS_UNROLL for (int k = 0; k < P_; ++k) s += A[i * P_ + k] * B[j * P_ + k];
      C[j * M_ + i] = s;
    }
  }
}
// C(m x n) = A(m x p) B(p x n)
template <int M_, int P_, int N_>
MAS_HD void mat_AB(const double* A, const double* B, double* C) {
  MAS_UNROLL for (int j = 0; j < N_; ++j) {
    MAS_UNROLL for (int i = 0; i < M_; ++i) {
      double s = 0.0;
      MAS_UNROLL for (int k = 0; k < P_; ++k) s += A[k * M_ + i] * B[j * P_ + k];
      C[j * M_ + i] = s;
    }
  }
}

// Eigen::LLT (unblocked, lower) on an n x n column-major matrix; 1 if a pivot is <= 0.
// NaN pivots pass, exactly as Eigen's `if (x <= 0) return k;`.
template <int N>
MAS_HD int llt_compute(double* Mx) {
  int fail = 0;
  MAS_UNROLL for (int k = 0; k < N; ++k) {
    if (!fail) {
      double x = Mx[k * N + k];
      if (k > 0) {
        double s = 0.0;
        MAS_UNROLL for (int j = 0; j < k; ++j) s += Mx[j * N + k] * Mx[j * N + k];
        x = Mx[k * N + k] - s;
      }
      if (x <= 0.0) {
        fail = 1;
      } else {
        x = sqrt(x);
        Mx[k * N + k] = x;
        MAS_UNROLL for (int r = k + 1; r < N; ++r) {
          double a = Mx[k * N + r];
          MAS_UNROLL for (int j = 0; j < k; ++j) a += (-Mx[j * N + k]) * Mx[j * N + r];
          Mx[k * N + r] = a / x;
        }
      }
    }
  }
  return fail;
}

// The same factorisation as straight-line code: every pivot is processed, a failed one only raises the flag.
// When the result is 0 (success) every operation and value is the one llt_compute performs; after a failure the
// contents of Mx are meaningless (sqrt of a negative pivot is NaN) and the caller must not use them.
template <int N>
MAS_HD int llt_compute_straight(double* Mx) {
  int fail = 0;
  MAS_UNROLL for (int k = 0; k < N; ++k) {
    double x = Mx[k * N + k];
    if (k > 0) {
      double s = 0.0;
      MAS_UNROLL for (int j = 0; j < k; ++j) s += Mx[j * N + k] * Mx[j * N + k];
      x = Mx[k * N + k] - s;
    }
    fail |= (x <= 0.0) ? 1 : 0;
    x = sqrt(x);
    Mx[k * N + k] = x;
    MAS_UNROLL for (int r = k + 1; r < N; ++r) {
      double a = Mx[k * N + r];
      MAS_UNROLL for (int j = 0; j < k; ++j) a += (-Mx[j * N + k]) * Mx[j * N + r];
      Mx[k * N + r] = a / x;
    }
  }
  return fail;
}

// LLT::solve(Identity) following Eigen's small triangular solver (reciprocal then multiply)
template <int N>
MAS_HD void llt_solve_identity(const double* Lm, double* Inv) {
  MAS_UNROLL for (int c = 0; c < N; ++c) {
    double o[N];
    MAS_UNROLL for (int i = 0; i < N; ++i) o[i] = (i == c) ? 1.0 : 0.0;
    MAS_UNROLL for (int i = 0; i < N; ++i) {
      const double a = 1.0 / Lm[i * N + i];
      o[i] *= a;
      const double b = o[i];
      MAS_UNROLL for (int r = i + 1; r < N; ++r) o[r] -= b * Lm[i * N + r];
    }
    MAS_UNROLL for (int i = N - 1; i >= 0; --i) {
      const double a = 1.0 / Lm[i * N + i];
      double b = 0.0;
      MAS_UNROLL for (int r = i + 1; r < N; ++r) b += Lm[i * N + r] * o[r];
      o[i] = (o[i] - b) * a;
    }
    MAS_UNROLL for (int i = 0; i < N; ++i) Inv[c * N + i] = o[i];
  }
}

// Q-function of one step from its linearisation D and the value function (ilqr.hpp:115-119)
template <class M>
struct QFunction {
  static constexpr int NX = M::NX, NU = M::NU;
  double qx[NX], qu[NU], qxx[NX * NX], qux[NU * NX], quu[NU * NU];
};

template <class M>
MAS_HD void assemble_q(const double* D, const double* vx, const double* vxx, QFunction<M>& q) {
  using L = DerivLayout<M>;
  constexpr int NX = M::NX, NU = M::NU;
  double tx[NX], tu[NU], AtV[NX * NX], BtV[NU * NX], t2[NX * NX];
  mat_AtB<NX, NX, 1>(D + L::oA, vx, tx);
  MAS_UNROLL for (int i = 0; i < NX; ++i) q.qx[i] = D[L::oLx + i] + tx[i];
  mat_AtB<NX, NU, 1>(D + L::oB, vx, tu);
  MAS_UNROLL for (int i = 0; i < NU; ++i) q.qu[i] = D[L::oLu + i] + tu[i];
  mat_AtB<NX, NX, NX>(D + L::oA, vxx, AtV);
  mat_AB<NX, NX, NX>(AtV, D + L::oA, t2);
  MAS_UNROLL for (int i = 0; i < NX * NX; ++i) q.qxx[i] = D[L::oLxx + i] + t2[i];
  mat_AtB<NX, NU, NX>(D + L::oB, vxx, BtV);
  double t3[NU * NX];
  mat_AB<NU, NX, NX>(BtV, D + L::oA, t3);
  MAS_UNROLL for (int i = 0; i < NU * NX; ++i) q.qux[i] = D[L::oLux + i] + t3[i];
  double t4[NU * NU];
  mat_AB<NU, NX, NU>(BtV, D + L::oB, t4);
  MAS_UNROLL for (int i = 0; i < NU * NU; ++i) q.quu[i] = D[L::oLuu + i] + t4[i];
}

// The reference's ladder is `while(true)` (ilqr.hpp:175-182) yet always ends: reg grows x10
// per rung, overflows to +inf after ~315 rungs, and a +inf / NaN pivot is not `<= 0`.
constexpr int kRegLadderMax = 400;

// LLT with the cumulative x10 regularisation ladder.  `Lfac` receives the factor; `shift` is a
// diagonal offset applied before the first attempt (0 for the per-agent path; the centralised
// path passes the shared offset).  Returns rungs used.
template <int NU>
MAS_HD int llt_ladder(const double* quu, double* Lfac) {
  double cur[NU * NU];
  MAS_UNROLL for (int i = 0; i < NU * NU; ++i) cur[i] = quu[i];
  double reg = 1e-6;
  int rungs = 0;
  for (;;) {
    MAS_UNROLL for (int i = 0; i < NU * NU; ++i) Lfac[i] = cur[i];
    if (llt_compute<NU>(Lfac) == 0) break;
    if (rungs >= kRegLadderMax) break;
    MAS_UNROLL for (int i = 0; i < NU; ++i) cur[i * NU + i] += reg;
    reg *= 10.0;
    ++rungs;
  }
  return rungs;
}

// gains and value update, ilqr.hpp:183-192; Lfac = factor of the REGULARISED q_uu, the value
// update uses the UNregularised q.quu.
template <class M>
MAS_HD void gains_and_value(const QFunction<M>& q, const double* Lfac, double* k, double* K, double* vx, double* vxx) {
  constexpr int NX = M::NX, NU = M::NU;
  double inv[NU * NU], ninv[NU * NU], KtQuu[NX * NU], tx[NX], txx[NX * NX];
  llt_solve_identity<NU>(Lfac, inv);
  MAS_UNROLL for (int i = 0; i < NU * NU; ++i) ninv[i] = -inv[i];
  mat_AB<NU, NU, 1>(ninv, q.qu, k);
  mat_AB<NU, NU, NX>(ninv, q.qux, K);
  mat_AtB<NU, NX, NU>(K, q.quu, KtQuu);
  mat_AtB<NU, NX, 1>(K, q.qu, tx);
  MAS_UNROLL for (int i = 0; i < NX; ++i) vx[i] = q.qx[i] + tx[i];
  mat_AtB<NU, NX, 1>(q.qux, k, tx);
  MAS_UNROLL for (int i = 0; i < NX; ++i) vx[i] = vx[i] + tx[i];
  mat_AB<NX, NU, 1>(KtQuu, k, tx);
  MAS_UNROLL for (int i = 0; i < NX; ++i) vx[i] = vx[i] + tx[i];
  mat_AtB<NU, NX, NX>(K, q.qux, txx);
  MAS_UNROLL for (int i = 0; i < NX * NX; ++i) vxx[i] = q.qxx[i] + txx[i];
  mat_AtB<NU, NX, NX>(q.qux, K, txx);
  MAS_UNROLL for (int i = 0; i < NX * NX; ++i) vxx[i] = vxx[i] + txx[i];
  mat_AB<NX, NU, NX>(KtQuu, K, txx);
  MAS_UNROLL for (int i = 0; i < NX * NX; ++i) vxx[i] = vxx[i] + txx[i];
  symmetrise_inplace<NX>(vxx);
}

// ------------------------------------------------------- augmented Lagrangian (ilqr.hpp) ------
struct ALParams { double penalty_increase, constraint_tolerance, activation_tolerance; };

// Q-function terms of the constraints at one step, ilqr.hpp:121-170.  lam / mu: this step's multipliers.
template <class M>
MAS_HD void al_q_terms(const double* D, const double* lam, const double* mu, double rho, double act_tol, QFunction<M>& q) {
  using L = DerivLayout<M>;
  constexpr int NX = M::NX, NU = M::NU;
  if constexpr (M::NEQ > 0) {
    constexpr int ME = M::NEQ;
    const double* c = D + L::oEq; const double* Jx = D + L::oEqJx; const double* Ju = D + L::oEqJu;
    double dual[ME];
    MAS_UNROLL for (int k = 0; k < ME; ++k) dual[k] = lam[k] + rho * c[k];
    MAS_UNROLL for (int i = 0; i < NX; ++i) { double a = 0.0; MAS_UNROLL for (int k = 0; k < ME; ++k) a += Jx[i * ME + k] * dual[k]; q.qx[i] += a; }
    MAS_UNROLL for (int i = 0; i < NU; ++i) { double a = 0.0; MAS_UNROLL for (int k = 0; k < ME; ++k) a += Ju[i * ME + k] * dual[k]; q.qu[i] += a; }
    MAS_UNROLL for (int j = 0; j < NX; ++j) { MAS_UNROLL for (int i = 0; i < NX; ++i) {
      double a = 0.0; MAS_UNROLL for (int k = 0; k < ME; ++k) a += (rho * Jx[i * ME + k]) * Jx[j * ME + k]; q.qxx[j * NX + i] += a; } }
    MAS_UNROLL for (int j = 0; j < NX; ++j) { MAS_UNROLL for (int i = 0; i < NU; ++i) {
      double a = 0.0; MAS_UNROLL for (int k = 0; k < ME; ++k) a += (rho * Ju[i * ME + k]) * Jx[j * ME + k]; q.qux[j * NU + i] += a; } }
    MAS_UNROLL for (int j = 0; j < NU; ++j) { MAS_UNROLL for (int i = 0; i < NU; ++i) {
      double a = 0.0; MAS_UNROLL for (int k = 0; k < ME; ++k) a += (rho * Ju[i * ME + k]) * Ju[j * ME + k]; q.quu[j * NU + i] += a; } }
  }
  if constexpr (M::NIN > 0) {
    constexpr int MI = M::NIN;
    const double* g = D + L::oIn; const double* Jx = D + L::oInJx; const double* Ju = D + L::oInJu;
    double dual[MI], act[MI];
    bool any = false;
    MAS_UNROLL for (int k = 0; k < MI; ++k) {
      const double slack = g[k] > 0.0 ? g[k] : 0.0;
      act[k] = (g[k] > -act_tol) ? 1.0 : 0.0;
      dual[k] = mu[k] * act[k] + rho * slack * act[k];
      any = any || (act[k] != 0.0);
    }
    MAS_UNROLL for (int i = 0; i < NX; ++i) { double a = 0.0; MAS_UNROLL for (int k = 0; k < MI; ++k) a += Jx[i * MI + k] * dual[k]; q.qx[i] += a; }
    MAS_UNROLL for (int i = 0; i < NU; ++i) { double a = 0.0; MAS_UNROLL for (int k = 0; k < MI; ++k) a += Ju[i * MI + k] * dual[k]; q.qu[i] += a; }
    if (any) {                                          // ilqr.hpp:163
      MAS_UNROLL for (int j = 0; j < NX; ++j) { MAS_UNROLL for (int i = 0; i < NX; ++i) {
        double a = 0.0; MAS_UNROLL for (int k = 0; k < MI; ++k) a += ((rho * Jx[i * MI + k]) * act[k]) * Jx[j * MI + k]; q.qxx[j * NX + i] += a; } }
      MAS_UNROLL for (int j = 0; j < NX; ++j) { MAS_UNROLL for (int i = 0; i < NU; ++i) {
        double a = 0.0; MAS_UNROLL for (int k = 0; k < MI; ++k) a += ((rho * Ju[i * MI + k]) * act[k]) * Jx[j * MI + k]; q.qux[j * NU + i] += a; } }
      MAS_UNROLL for (int j = 0; j < NU; ++j) { MAS_UNROLL for (int i = 0; i < NU; ++i) {
        double a = 0.0; MAS_UNROLL for (int k = 0; k < MI; ++k) a += ((rho * Ju[i * MI + k]) * act[k]) * Ju[j * MI + k]; q.quu[j * NU + i] += a; } }
    }
  }
}

// merit contribution of one step added to `merit` in the reference's order, ilqr.hpp:386-404
template <class M>
MAS_HD double al_merit_step(double merit, const double* x, const double* u, const double* lam, const double* mu, double rho, double act_tol,
                            const ModelParams& P) {
  if constexpr (M::NEQ > 0) {
    double r[M::NEQ]; M::equality(x, u, r, P);
    double dot = 0.0, sq = 0.0;
    MAS_UNROLL for (int k = 0; k < M::NEQ; ++k) { dot += lam[k] * r[k]; sq += r[k] * r[k]; }
    merit += dot + 0.5 * rho * sq;
  }
  if constexpr (M::NIN > 0) {
    double r[M::NIN]; M::inequality(x, u, r, P);
    double dot = 0.0, sq = 0.0;
    MAS_UNROLL for (int k = 0; k < M::NIN; ++k) {
      const double slack = r[k] > 0.0 ? r[k] : 0.0;
      const double active = (r[k] > -act_tol) ? 1.0 : 0.0;
      const double active_slack = slack * active, weighted = mu[k] * active;
      dot += weighted * active_slack; sq += active_slack * active_slack;
    }
    merit += dot;
    merit += 0.5 * rho * sq;
  }
  return merit;
}

// multiplier update of one step, ilqr.hpp:239-254; returns the squared violations through eqv / inv
template <class M>
MAS_HD void al_update_step(const double* x, const double* u, double* lam, double* mu, double rho, double& eqv, double& inv, const ModelParams& P) {
  if constexpr (M::NEQ > 0) {
    double r[M::NEQ]; M::equality(x, u, r, P);
    double sq = 0.0;
    MAS_UNROLL for (int k = 0; k < M::NEQ; ++k) { lam[k] += rho * r[k]; sq += r[k] * r[k]; }
    eqv += sq;
  }
  if constexpr (M::NIN > 0) {
    double r[M::NIN]; M::inequality(x, u, r, P);
    double sq = 0.0;
    MAS_UNROLL for (int k = 0; k < M::NIN; ++k) {
      const double positive = r[k] > 0.0 ? r[k] : 0.0;
      const double v = mu[k] + rho * positive;
      mu[k] = v > 0.0 ? v : 0.0;
      sq += positive * positive;
    }
    inv += sq;
  }
}

// ------------------------------------------------------------------- forward step ------
// one column of the line-search rollout, ilqr.hpp:210-216
template <class M>
MAS_HD void forward_control(const double* x_trial, const double* x_ref, const double* u_ref, const double* k, const double* K,
                            double alpha, bool clamp, const double* lb, const double* ub, double* u_out) {
  constexpr int NX = M::NX, NU = M::NU;
  double dx[NX];
  MAS_UNROLL for (int i = 0; i < NX; ++i) dx[i] = x_trial[i] - x_ref[i];
  MAS_UNROLL for (int i = 0; i < NU; ++i) {
    double kd = 0.0;
    MAS_UNROLL for (int c = 0; c < NX; ++c) kd += K[c * NU + i] * dx[c];
    u_out[i] = (u_ref[i] + alpha * k[i]) + kd;
  }
  if (clamp) clamp_column<NU>(u_out, lb, ub);
}

}  // namespace mas
