// Embedded Runge-Kutta pairs for the adaptive solvers of the reference's `solver:` key (vihds/ode.py:79-81 hands any
// name other than modeuler / modeulerwhile to torchdiffeq==0.1, tests/test_ode_solvers.py:66-80 runs dopri5, dopri8 and
// the adjoint variants).  torchdiffeq is a third-party dependency that is absent from /root/reference: what follows is a
// restatement of its published algorithm (adaptive-step explicit RK with ONE step size for the whole batch, error ratio
// = mean over all state elements of (err / (atol + rtol max(|y0|, |y1|)))^2, step factor as in `_optimal_step_size`),
// parity unpinned.  Two deliberate differences, both documented in DESIGN.md: accepted steps are clipped so that every
// output time is a step end (torchdiffeq steps past them and interpolates), and the gradient is the discrete adjoint of
// the accepted steps (torchdiffeq differentiates through its python loop, or -- odeint_adjoint -- integrates the
// continuous adjoint backwards; `adjoint_solver: true` maps to the same discrete adjoint here).
//
// `dopri8` is served by a same-order stand-in (DOP853), see Tableau<VIHDS_SOLVER_DOPRI8> below.
//
// Structure: (1) the controller (vihds_ode_adaptive_grid, synchronous, host-driven) walks the batch through trial steps
// with `ode_trial_kernel` and returns the accepted time grid; (2) the ordinary fixed-grid forward / adjoint kernels then
// integrate on that grid with the pair's higher-order tableau (`rk_step_generic`, `rk_step_generic_vjp`).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"
#include "vihds_dop853_tableau.hpp"

namespace vihds {

template <int SOLVER>
struct Tableau;

// Dormand-Prince 5(4), FSAL; error coefficients as in torchdiffeq's dopri5 [recalled]
template <>
struct Tableau<VIHDS_SOLVER_DOPRI5> {
  static constexpr int NS = 6;      // stages of the propagated (5th order) solution
  static constexpr int ORDER = 5;
  static constexpr float c(int s) {
    const float v[7] = {0.f, 1.f / 5.f, 3.f / 10.f, 4.f / 5.f, 8.f / 9.f, 1.f, 1.f};
    return v[s];
  }
  static constexpr float a(int s, int r) {
    const float v[7][6] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f},
                           {1.f / 5.f, 0.f, 0.f, 0.f, 0.f, 0.f},
                           {3.f / 40.f, 9.f / 40.f, 0.f, 0.f, 0.f, 0.f},
                           {44.f / 45.f, -56.f / 15.f, 32.f / 9.f, 0.f, 0.f, 0.f},
                           {19372.f / 6561.f, -25360.f / 2187.f, 64448.f / 6561.f, -212.f / 729.f, 0.f, 0.f},
                           {9017.f / 3168.f, -355.f / 33.f, 46732.f / 5247.f, 49.f / 176.f, -5103.f / 18656.f, 0.f},
                           {35.f / 384.f, 0.f, 500.f / 1113.f, 125.f / 192.f, -2187.f / 6784.f, 11.f / 84.f}};
    return v[s][r];
  }
  static constexpr float b(int s) { return a(6, s); }
  static constexpr float e(int s) {  // error estimate = h sum_s e_s k_s over the 7 stages (k_7 = f(t + h, y'))
    const float v[7] = {35.f / 384.f - 1951.f / 21600.f, 0.f, 500.f / 1113.f - 22642.f / 50085.f,
                        125.f / 192.f - 451.f / 720.f, -2187.f / 6784.f + 12231.f / 42400.f,
                        11.f / 84.f - 649.f / 6300.f, -1.f / 60.f};
    return v[s];
  }
};
// Bogacki-Shampine 3(2), FSAL
template <>
struct Tableau<VIHDS_SOLVER_BOSH3> {
  static constexpr int NS = 3;
  static constexpr int ORDER = 3;
  static constexpr float c(int s) {
    const float v[4] = {0.f, 0.5f, 0.75f, 1.f};
    return v[s];
  }
  static constexpr float a(int s, int r) {
    const float v[4][3] = {{0.f, 0.f, 0.f}, {0.5f, 0.f, 0.f}, {0.f, 0.75f, 0.f}, {2.f / 9.f, 1.f / 3.f, 4.f / 9.f}};
    return v[s][r];
  }
  static constexpr float b(int s) { return a(3, s); }
  static constexpr float e(int s) {
    const float v[4] = {2.f / 9.f - 7.f / 24.f, 1.f / 3.f - 0.25f, 4.f / 9.f - 1.f / 3.f, -0.125f};
    return v[s];
  }
};
// Heun-Euler 2(1) ("adaptive_heun"); its error stage is stage 2 itself (no extra evaluation), kept in the FSAL form
template <>
struct Tableau<VIHDS_SOLVER_ADAPTIVE_HEUN> {
  static constexpr int NS = 2;
  static constexpr int ORDER = 2;
  static constexpr float c(int s) { return s == 0 ? 0.f : 1.f; }
  static constexpr float a(int s, int r) { return (s == 1 && r == 0) ? 1.f : ((s == 2) ? 0.5f : 0.f); }
  static constexpr float b(int) { return 0.5f; }
  static constexpr float e(int s) { return s == 0 ? -0.5f : (s == 1 ? 0.5f : 0.f); }
};

// `solver: dopri8`: the 12-stage Dormand-Prince 8(5,3) pair (DOP853; vihds_dop853_tableau.hpp says why not torchdiffeq's
// own 8(7) tableau), FSAL form like the others: evaluation 13 = f(t + h, y') carries the last error weight
template <>
struct Tableau<VIHDS_SOLVER_DOPRI8> {
  static constexpr int NS = Dop853Tab::NS;
  static constexpr int ORDER = 8;
  static constexpr float c(int s) { return Dop853Tab::c(s); }
  static constexpr float a(int s, int r) { return Dop853Tab::a(s, r); }
  static constexpr float b(int s) { return Dop853Tab::a(NS, s); }
  static constexpr float e(int s) { return Dop853Tab::e(s); }
};

constexpr bool solver_is_adaptive(int solver) { return solver >= VIHDS_SOLVER_DOPRI5 && solver <= VIHDS_SOLVER_DOPRI8; }
__host__ __device__ constexpr int adaptive_stages(int solver) {
  return solver == VIHDS_SOLVER_DOPRI8 ? 12 : (solver == VIHDS_SOLVER_DOPRI5 ? 6 : (solver == VIHDS_SOLVER_BOSH3 ? 3 : 2));
}

// one step y -> y' of the propagated solution; err (optional) = the embedded error estimate (one more evaluation,
// except where the pair's error weights on it vanish)
template <class M, class TB>
__device__ __forceinline__ void rk_step_generic(float t0, float h, float* y, const float* p, const float* wts,
                                                float* err) {
  constexpr int N = M::N, NS = TB::NS;
  float k[NS + 1][N], ya[N];
  VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float v = 0.f;
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (TB::a(s, r) != 0.f) v = fmaf(TB::a(s, r), k[r][j], v);
      ya[j] = fmaf(h, v, y[j]);
    }
    M::rhs(t0 + TB::c(s) * h, ya, p, wts, k[s]);
  }
  VIHDS_UNROLL for (int j = 0; j < N; ++j) {
    float v = 0.f;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (TB::b(s) != 0.f) v = fmaf(TB::b(s), k[s][j], v);
    ya[j] = fmaf(h, v, y[j]);
  }
  if (err) {
    if (TB::e(NS) != 0.f) M::rhs(t0 + h, ya, p, wts, k[NS]);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float v = 0.f;
      VIHDS_UNROLL for (int s = 0; s <= NS; ++s)
        if (TB::e(s) != 0.f) v = fmaf(TB::e(s), k[s][j], v);
      err[j] = h * v;
    }
  }
  VIHDS_UNROLL for (int j = 0; j < N; ++j) y[j] = ya[j];
}

// reverse of one step: lam (adjoint of y') -> adjoint of y; pb += parameter adjoint.  CALL(t, y, v, yb) applies the
// model's rhs_vjp (v = adjoint of the derivative, yb += adjoint of the state).
template <class M, class TB, class CALL>
__device__ __forceinline__ void rk_step_generic_vjp(float t0, float h, const float* y, const float* p, const float* wts,
                                                    float* lam, CALL&& call) {
  constexpr int N = M::N, NS = TB::NS;
  float k[NS][N], Y[NS][N], Yb[NS][N];
  VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float v = 0.f;
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (TB::a(s, r) != 0.f) v = fmaf(TB::a(s, r), k[r][j], v);
      Y[s][j] = fmaf(h, v, y[j]);
    }
    if (s + 1 < NS) M::rhs(t0 + TB::c(s) * h, Y[s], p, wts, k[s]);
  }
  float lam1[N];
  VIHDS_UNROLL for (int j = 0; j < N; ++j) lam1[j] = lam[j];
  VIHDS_UNROLL for (int s = NS - 1; s >= 0; --s) {
    float kb[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float v = TB::b(s) * lam1[j];
      VIHDS_UNROLL for (int r = s + 1; r < NS; ++r)
        if (TB::a(r, s) != 0.f) v = fmaf(TB::a(r, s), Yb[r][j], v);
      kb[j] = h * v;
      Yb[s][j] = 0.f;
    }
    call(t0 + TB::c(s) * h, Y[s], kb, Yb[s]);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) lam[j] += Yb[s][j];
  }
}

}  // namespace vihds
