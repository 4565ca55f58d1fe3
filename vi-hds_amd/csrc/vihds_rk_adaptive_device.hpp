// torchdiffeq==0.1's adaptive-step algorithm ON THE DEVICE (reference call site vihds/ode.py:79-81; the dependency is absent
// from the tree: restated from its published code -- AdaptiveStepsizeODESolver.advance, _select_initial_step,
// _compute_error_ratio, _optimal_step_size, interp._interp_fit / _interp_evaluate -- exactly as oracle.odeint_adaptive
// restates it; parity unpinned).  Round 4: the product's controller used to clip every accepted step to the
// output times (a different accepted grid than the dependency's) and ran from the host with one device round trip per
// trial step.  Here:
//
//   * ode_adaptive_fwd_kernel: ONE persistent launch, one thread per trajectory (<= 65 536 trajectories), state, step
//     size and accept / reject decisions in device memory and registers.  A trial step = the pair's stages + the FSAL
//     evaluation f(t + dt, y1), the error ratio summed per block, one grid barrier (a monotonic counter, agent-scope
//     release / acquire, every spin bounded), then every block adds the blocks' partial sums in block order -- the same
//     number in every block, so all take the same decision with no further exchange.  Steps are NOT shortened to hit an
//     output time: after an accepted step every output time inside it is evaluated from the step's quartic interpolant
//     (y0, y1, y_mid, f0, f1).  Accepted steps are logged (start time, size, state at the start: the "tape").
//   * ode_adaptive_bwd_kernel: the discrete adjoint of exactly that computation with the accepted step sizes held
//     constant: reverse walk over the tape; per step the adjoints of the interpolated outputs enter through y0, y1, y_mid,
//     f0, f1, then the stages are pulled back with the model's rhs_vjp.
//
// No host round trip, no hipStreamSynchronize: the pair of launches is hipGraph-capturable.  Round 5: the white-box models
// with neural precisions (*_precisions without a hidden layer: every spec the reference ships) run here too -- the network's
// weights are read as scalars like in the fixed-grid kernels, and the adjoint accumulates their gradient per thread (2 (4 NIN +
// 4) registers: WeightGradCtx), reduced per wavefront and added to g_weights.  dr_blackbox, a hidden precision layer and
// `dopri8` keep the clipped-grid controller (vihds_ode_adaptive_grid).
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_rk_adaptive.hpp"

namespace vihds {

// y_mid = y0 + dt sum_r mid(r) k_r over the NS stages and the FSAL evaluation (torchdiffeq: DPS_C_MID of dopri5.py; the
// `mid` rows of the Bogacki-Shampine and Heun-Euler solvers)
template <int SOLVER> struct MidWeights;
template <> struct MidWeights<VIHDS_SOLVER_DOPRI5> {
  static constexpr double w(int r) {
    const double v[7] = {6025192743.0 / 30085553152.0 / 2, 0.0, 51252292925.0 / 65400821598.0 / 2, -2691868925.0 / 45128329728.0 / 2,
                         187940372067.0 / 1594534317056.0 / 2, -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};
    return v[r];
  }
};
template <> struct MidWeights<VIHDS_SOLVER_BOSH3> {
  static constexpr double w(int r) { return r == 1 ? 0.5 : 0.0; }
};
template <> struct MidWeights<VIHDS_SOLVER_ADAPTIVE_HEUN> {
  static constexpr double w(int r) { return r == 0 ? 0.5 : 0.0; }
};
// tableau entries in double (the python side of torchdiffeq multiplies dt, a python float, with them before the product
// meets a float32 tensor)
template <int SOLVER> struct TabD;
template <> struct TabD<VIHDS_SOLVER_DOPRI5> {
  static constexpr double c(int s) { const double v[7] = {0, 1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1, 1}; return v[s]; }
  static constexpr double a(int s, int r) {
    const double v[7][6] = {{0, 0, 0, 0, 0, 0}, {1.0 / 5, 0, 0, 0, 0, 0}, {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
                            {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
                            {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
                            {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
                            {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
    return v[s][r];
  }
  static constexpr double e(int s) {
    const double v[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085, 125.0 / 192 - 451.0 / 720,
                         -2187.0 / 6784 + 12231.0 / 42400, 11.0 / 84 - 649.0 / 6300, -1.0 / 60};
    return v[s];
  }
};
template <> struct TabD<VIHDS_SOLVER_BOSH3> {
  static constexpr double c(int s) { const double v[4] = {0, 0.5, 0.75, 1}; return v[s]; }
  static constexpr double a(int s, int r) {
    const double v[4][3] = {{0, 0, 0}, {0.5, 0, 0}, {0, 0.75, 0}, {2.0 / 9, 1.0 / 3, 4.0 / 9}};
    return v[s][r];
  }
  static constexpr double e(int s) { const double v[4] = {2.0 / 9 - 7.0 / 24, 1.0 / 3 - 0.25, 4.0 / 9 - 1.0 / 3, -0.125}; return v[s]; }
};
template <> struct TabD<VIHDS_SOLVER_ADAPTIVE_HEUN> {
  static constexpr double c(int s) { return s == 0 ? 0.0 : 1.0; }
  static constexpr double a(int s, int r) { return (s == 1 && r == 0) ? 1.0 : (s == 2 ? 0.5 : 0.0); }
  static constexpr double e(int s) { return s == 0 ? -0.5 : (s == 1 ? 0.5 : 0.0); }
};

constexpr int ADP_BLOCK = 256;
constexpr int ADP_MAX_BLOCKS = 256;   // one block per CU at most: every block is resident, the grid barrier is safe
constexpr int ADP_CTRL = 16;          // control words at the head of the workspace (32-bit)
enum { ADP_COUNTER = 0, ADP_ERR = 1, ADP_NACC = 2, ADP_NREJ = 3 };
enum { ADP_E_OK = 0, ADP_E_TIMEOUT = 1, ADP_E_MAX_STEPS = 2, ADP_E_UNDERFLOW = 3, ADP_E_NAN = 4 };

// workspace layout (32-bit words): ctrl [16] | partial double [2 parities][2 sums][nblk] | t0 double [S+1] | dt double [S]
// | out_x double [T] | out_step int [T] (padded to even) | tape_y float [S+1][N][n]      (S = max_steps)
struct AdaptiveLayout {
  size_t partial, t0, dt, out_x, out_step, tape, total;
  __host__ __device__ AdaptiveLayout(int nblk, int max_steps, int T, int N, size_t n) {
    partial = ADP_CTRL;
    t0 = partial + 2 * (size_t)(2 * 2 * nblk);
    dt = t0 + 2 * (size_t)(max_steps + 1);
    out_x = dt + 2 * (size_t)max_steps;
    out_step = out_x + 2 * (size_t)T;
    tape = out_step + (size_t)((T + 1) & ~1);
    total = tape + (size_t)(max_steps + 1) * N * n;
  }
};
struct AdaptiveDev {
  float* ws;
  float rtol, atol;
  int max_steps;
};

// every spin is bounded (~1 s): a block that gives up sets the error word, the others time out after it
__device__ __forceinline__ bool adp_grid_barrier(unsigned int* ctrl, unsigned int nblocks, unsigned int& epoch) {
  __shared__ int ok_;
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    epoch += 1;
    atomicAdd(&ctrl[ADP_COUNTER], 1u);
    const unsigned int target = epoch * nblocks;
    int good = 1;
    long spins = 0;
    while (__hip_atomic_load(&ctrl[ADP_COUNTER], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(4);
      if (++spins > 2000000L || __hip_atomic_load(&ctrl[ADP_ERR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        atomicCAS(&ctrl[ADP_ERR], 0u, (unsigned int)ADP_E_TIMEOUT);
        good = 0;
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    ok_ = good;
  }
  __syncthreads();
  return ok_ != 0;
}
// sum of this block's per-thread doubles -> partial slot; barrier; every block adds all partials in block order
__device__ __forceinline__ bool adp_all_sum(double v0, double v1, double* partial, int parity, unsigned int* ctrl,
                                            unsigned int& epoch, double* out0, double* out1) {
  __shared__ double red[2][ADP_BLOCK];
  __shared__ double tot[2];
  const int nblk = gridDim.x;
  red[0][threadIdx.x] = v0;
  red[1][threadIdx.x] = v1;
  __syncthreads();
  for (int w = ADP_BLOCK / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  double* slot = partial + (size_t)parity * 2 * nblk;
  if (threadIdx.x == 0) {
    slot[blockIdx.x] = red[0][0];
    slot[nblk + blockIdx.x] = red[1][0];
  }
  if (!adp_grid_barrier(ctrl, (unsigned int)nblk, epoch)) return false;
  if (threadIdx.x == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int q = 0; q < nblk; ++q) {
      s0 += slot[q];
      s1 += slot[nblk + q];
    }
    tot[0] = s0;
    tot[1] = s1;
  }
  __syncthreads();
  *out0 = tot[0];
  *out1 = tot[1];
  __syncthreads();
  return true;
}

template <class M, int SOLVER>
__global__ void __launch_bounds__(ADP_BLOCK) ode_adaptive_fwd_kernel(OdeArgs a, AdaptiveDev d) {
  using TB = Tableau<SOLVER>;
  using TD = TabD<SOLVER>;
  using MW = MidWeights<SOLVER>;
  constexpr int N = M::N, NS = TB::NS;
  static_assert(!is_blackbox<M>::value, "the device-resident adaptive solver does not serve dr_blackbox");
  const AdaptiveLayout L(gridDim.x, d.max_steps, a.T, N, a.n);
  unsigned int* ctrl = reinterpret_cast<unsigned int*>(d.ws);
  double* partial = reinterpret_cast<double*>(d.ws + L.partial);
  double* t0s = reinterpret_cast<double*>(d.ws + L.t0);
  double* dts = reinterpret_cast<double*>(d.ws + L.dt);
  double* out_x = reinterpret_cast<double*>(d.ws + L.out_x);
  int* out_step = reinterpret_cast<int*>(d.ws + L.out_step);
  float* tape = d.ws + L.tape;
  const int i0 = blockIdx.x * ADP_BLOCK + threadIdx.x;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const size_t n = a.n;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  float th[M::NSLOT], prec[4], c[M::NC > 0 ? M::NC : 1], p[M::NP], y[N];
  load_theta<M>(a, i, b, th, prec, c);
  M::prepare(th, c, p);
  if constexpr (M::NEURAL_PREC) p[M::NP - 1] = __int_as_float(a.n_hidden_prec);
  M::init(th, c, y);
  const float* wts = a.weights;  // (NULL for the models without a network)
  unsigned int epoch = 0;
  const double cnt = (double)N * (double)n;
  const float rtol = d.rtol, atol = d.atol;
  double t = (double)a.times[0];
  float fcur[N];
  M::rhs((float)t, y, p, wts, fcur);
  if (live) {
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      tape[(size_t)j * n + i] = y[j];
      a.traj[(size_t)j * n + i] = y[j];
    }
  }
  // ---- _select_initial_step -----------------------------------------------------------------------------------------------
  double dt;
  {
    double s0 = 0.0, s1 = 0.0, q0, q1, q2, dummy;
    float sc[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      sc[j] = atol + rtol * fabsf(y[j]);
      const float u0 = y[j] / sc[j], u1 = fcur[j] / sc[j];
      if (live) { s0 += (double)u0 * (double)u0; s1 += (double)u1 * (double)u1; }
    }
    if (!adp_all_sum(s0, s1, partial, 0, ctrl, epoch, &q0, &q1)) return;
    const double d0 = sqrt(q0 / cnt), d1 = sqrt(q1 / cnt);
    const double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    float y1[N], f1[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y1[j] = y[j] + (float)h0 * fcur[j];
    M::rhs((float)(t + h0), y1, p, wts, f1);
    double s2 = 0.0;
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      const float u2 = (f1[j] - fcur[j]) / sc[j];
      if (live) s2 += (double)u2 * (double)u2;
    }
    if (!adp_all_sum(s2, 0.0, partial, 1, ctrl, epoch, &q2, &dummy)) return;
    const double d2 = sqrt(q2 / cnt) / h0;
    const double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? fmax(1e-6, h0 * 1e-3) : pow(0.01 / fmax(d1, d2), 1.0 / (TB::ORDER + 1));
    dt = fmin(100.0 * h0, h1);
  }
  // ---- AdaptiveStepsizeODESolver.advance for every output time ------------------------------------------------------------
  int g = 0, k_out = 1, n_rej = 0, parity = 0;
  int err_code = ADP_E_OK;
  while (k_out < a.T) {
    if (g >= d.max_steps) { err_code = ADP_E_MAX_STEPS; break; }
    if (!(t + dt > t)) { err_code = ADP_E_UNDERFLOW; break; }
    float k[NS + 1][N], y1[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) k[0][j] = fcur[j];
    VIHDS_UNROLL for (int s = 1; s < NS; ++s) {
      float ya[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        float v = y[j];
        VIHDS_UNROLL for (int r = 0; r < s; ++r)
          if (TD::a(s, r) != 0.0) v = v + (float)(dt * TD::a(s, r)) * k[r][j];
        ya[j] = v;
      }
      M::rhs((float)(t + TD::c(s) * dt), ya, p, wts, k[s]);
    }
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float v = y[j];
      VIHDS_UNROLL for (int r = 0; r < NS; ++r)
        if (TD::a(NS, r) != 0.0) v = v + (float)(dt * TD::a(NS, r)) * k[r][j];
      y1[j] = v;
    }
    M::rhs((float)(t + dt), y1, p, wts, k[NS]);
    double s0 = 0.0;
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float er = 0.f;
      bool first = true;
      VIHDS_UNROLL for (int r = 0; r <= NS; ++r)
        if (TD::e(r) != 0.0) {
          const float term = (float)(dt * TD::e(r)) * k[r][j];
          er = first ? term : er + term;
          first = false;
        }
      const float tol = atol + rtol * fmaxf(fabsf(y[j]), fabsf(y1[j]));
      const float rr = er / tol;
      if (live) s0 += (double)rr * (double)rr;
    }
    double e2, dummy;
    parity ^= 1;
    if (!adp_all_sum(s0, 0.0, partial, parity, ctrl, epoch, &e2, &dummy)) return;
    const double ratio = e2 / cnt;
    if (!(ratio == ratio)) { err_code = ADP_E_NAN; break; }
    double dt_next;
    if (ratio == 0.0) dt_next = dt * 10.0;
    else {
      const double dfactor = ratio < 1.0 ? 1.0 : 0.2;
      dt_next = dt / fmax(0.1, fmin(pow(ratio, 0.5 / TB::ORDER) / 0.9, 1.0 / dfactor));
    }
    if (ratio <= 1.0) {
      const double ta = t, tb = t + dt;
      // the quartic of this step (interp._interp_fit) at every output time it covers
      if (k_out < a.T && (double)a.times[k_out] <= tb) {
        float ym[N], ca[N], cb[N], cc[N];
        VIHDS_UNROLL for (int j = 0; j < N; ++j) {
          float v = y[j];
          VIHDS_UNROLL for (int r = 0; r <= NS; ++r)
            if (MW::w(r) != 0.0) v = v + (float)(dt * MW::w(r)) * k[r][j];
          ym[j] = v;
          const float f0 = k[0][j], f1 = k[NS][j], fdt = (float)dt;
          ca[j] = (float)(2.0 * dt) * (f1 - f0) - 8.f * (y1[j] + y[j]) + 16.f * ym[j];
          cb[j] = fdt * (5.f * f0 - 3.f * f1) + 18.f * y[j] + 14.f * y1[j] - 32.f * ym[j];
          cc[j] = fdt * (f1 - 4.f * f0) - 11.f * y[j] - 5.f * y1[j] + 16.f * ym[j];
        }
        while (k_out < a.T && (double)a.times[k_out] <= tb) {
          const double x = ((double)a.times[k_out] - ta) / (tb - ta);
          const float x1 = (float)x, x2 = (float)(x * x), x3 = (float)(x * x * x), x4 = (float)(x * x * x * x);
          if (live) {
            VIHDS_UNROLL for (int j = 0; j < N; ++j) {
              float tot = y[j];
              tot = tot + ((float)dt * k[0][j]) * x1;
              tot = tot + cc[j] * x2;
              tot = tot + cb[j] * x3;
              tot = tot + ca[j] * x4;
              a.traj[((size_t)k_out * N + j) * n + i] = tot;
            }
          }
          if (lead) { out_x[k_out] = x; out_step[k_out] = g; }
          ++k_out;
        }
      }
      if (lead) { t0s[g] = ta; dts[g] = dt; }
      ++g;
      t = tb;
      VIHDS_UNROLL for (int j = 0; j < N; ++j) { y[j] = y1[j]; fcur[j] = k[NS][j]; }
      if (live) {
        VIHDS_UNROLL for (int j = 0; j < N; ++j) tape[((size_t)g * N + j) * n + i] = y[j];
      }
    } else {
      ++n_rej;
    }
    dt = dt_next;
  }
  if (lead) {
    ctrl[ADP_NACC] = (unsigned int)g;
    ctrl[ADP_NREJ] = (unsigned int)n_rej;
    if (err_code != ADP_E_OK) atomicCAS(&ctrl[ADP_ERR], 0u, (unsigned int)err_code);
    out_x[0] = 0.0;
    out_step[0] = -1;
  }
}

// discrete adjoint over the tape.  g_traj [T][N][B][S]: upstream gradient of the solution at the output times.
template <class M, int SOLVER>
__global__ void __launch_bounds__(ADP_BLOCK) ode_adaptive_bwd_kernel(OdeArgs a, AdaptiveDev d, int nblk_fwd) {
  using TB = Tableau<SOLVER>;
  using TD = TabD<SOLVER>;
  using MW = MidWeights<SOLVER>;
  constexpr int N = M::N, NS = TB::NS;
  const AdaptiveLayout L(nblk_fwd, d.max_steps, a.T, N, a.n);
  const unsigned int* ctrl = reinterpret_cast<const unsigned int*>(d.ws);
  const double* t0s = reinterpret_cast<const double*>(d.ws + L.t0);
  const double* dts = reinterpret_cast<const double*>(d.ws + L.dt);
  const double* out_x = reinterpret_cast<const double*>(d.ws + L.out_x);
  const int* out_step = reinterpret_cast<const int*>(d.ws + L.out_step);
  const float* tape = d.ws + L.tape;
  const int i0 = blockIdx.x * ADP_BLOCK + threadIdx.x;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const size_t n = a.n;
  float th[M::NSLOT], prec[4], c[M::NC > 0 ? M::NC : 1], p[M::NP];
  load_theta<M>(a, i, b, th, prec, c);
  M::prepare(th, c, p);
  if constexpr (M::NEURAL_PREC) p[M::NP - 1] = __int_as_float(a.n_hidden_prec);
  const float* wts = a.weights;
  using CtxImpl = bwd_ctx<M>;  // NoCtx, or per-thread weight-gradient accumulators (white-box + neural precisions)
  typename CtxImpl::type ctx;
  CtxImpl::init(ctx, a, i);
  float lam[N], pb[M::NP];
  VIHDS_UNROLL for (int j = 0; j < N; ++j) lam[j] = 0.f;
  VIHDS_UNROLL for (int j = 0; j < M::NP; ++j) pb[j] = 0.f;
  const int G = (ctrl[ADP_ERR] == 0u) ? (int)ctrl[ADP_NACC] : 0;
  int kp = a.T - 1;
  for (int g = G - 1; g >= 0; --g) {
    const double t = t0s[g], dt = dts[g];
    float y[N], k[NS + 1][N], Ys[NS][N], y1[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) { y[j] = tape[((size_t)g * N + j) * n + i]; Ys[0][j] = y[j]; }
    M::rhs((float)t, y, p, wts, k[0]);
    VIHDS_UNROLL for (int s = 1; s < NS; ++s) {
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        float v = y[j];
        VIHDS_UNROLL for (int r = 0; r < s; ++r)
          if (TD::a(s, r) != 0.0) v = v + (float)(dt * TD::a(s, r)) * k[r][j];
        Ys[s][j] = v;
      }
      M::rhs((float)(t + TD::c(s) * dt), Ys[s], p, wts, k[s]);
    }
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      float v = y[j];
      VIHDS_UNROLL for (int r = 0; r < NS; ++r)
        if (TD::a(NS, r) != 0.0) v = v + (float)(dt * TD::a(NS, r)) * k[r][j];
      y1[j] = v;
    }
    float kb[NS + 1][N], y0b[N], y1b[N], ymb[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      y0b[j] = 0.f; y1b[j] = 0.f; ymb[j] = 0.f;
      VIHDS_UNROLL for (int r = 0; r <= NS; ++r) kb[r][j] = 0.f;
    }
    // adjoints of the outputs interpolated inside this step: total = y0 + (dt f0) x + c x^2 + b x^3 + a x^4
    bool any = false;
    while (kp >= 1 && out_step[kp] == g) {
      any = true;
      const double x = out_x[kp];
      const float x1 = (float)x, x2 = (float)(x * x), x3 = (float)(x * x * x), x4 = (float)(x * x * x * x);
      const float fdt = (float)dt, f2dt = (float)(2.0 * dt);
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        const float lo = a.g_traj ? a.g_traj[((size_t)kp * N + j) * n + i] : 0.f;
        const float ab = x4 * lo, bb = x3 * lo, cb = x2 * lo;
        y0b[j] += lo - 8.f * ab + 18.f * bb - 11.f * cb;
        y1b[j] += -8.f * ab + 14.f * bb - 5.f * cb;
        ymb[j] += 16.f * ab - 32.f * bb + 16.f * cb;
        kb[0][j] += fdt * x1 * lo - f2dt * ab + 5.f * fdt * bb - 4.f * fdt * cb;   // f0
        kb[NS][j] += f2dt * ab - 3.f * fdt * bb + fdt * cb;                       // f1
      }
      --kp;
    }
    float L1[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      if (any) {
        y0b[j] += ymb[j];
        VIHDS_UNROLL for (int r = 0; r <= NS; ++r)
          if (MW::w(r) != 0.0) kb[r][j] += (float)(dt * MW::w(r)) * ymb[j];
      }
      L1[j] = lam[j] + y1b[j];
    }
    if (any) {  // f1 = f(t + dt, y1): only the interpolant reads it
      float v[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) v[j] = kb[NS][j];
      call_vjp<M>((float)(t + dt), y1, p, wts, v, L1, pb, ctx);
    }
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      y0b[j] += L1[j];
      VIHDS_UNROLL for (int r = 0; r < NS; ++r)
        if (TD::a(NS, r) != 0.0) kb[r][j] += (float)(dt * TD::a(NS, r)) * L1[j];
    }
    VIHDS_UNROLL for (int s = NS - 1; s >= 0; --s) {
      float Yb[N], v[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) { Yb[j] = 0.f; v[j] = kb[s][j]; }
      call_vjp<M>((float)(t + TD::c(s) * dt), Ys[s], p, wts, v, Yb, pb, ctx);
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        y0b[j] += Yb[j];
        VIHDS_UNROLL for (int r = 0; r < s; ++r)
          if (TD::a(s, r) != 0.0) kb[r][j] += (float)(dt * TD::a(s, r)) * Yb[j];
      }
    }
    VIHDS_UNROLL for (int j = 0; j < N; ++j) lam[j] = y0b[j];
  }
  if (a.g_traj) {  // the first output is the initial state itself
    VIHDS_UNROLL for (int j = 0; j < N; ++j) lam[j] += a.g_traj[(size_t)j * n + i];
  }
  float thb[M::NSLOT];
  VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) thb[q] = 0.f;
  M::prepare_vjp(th, c, p, pb, thb);
  M::init_vjp(lam, thb);
  if (live) {
    VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) a.g_theta[(size_t)a.slot_row[q] * n + i] = thb[q];
    if (!M::NEURAL_PREC) {  // (the constant precisions enter the log-likelihood only, which the caller forms itself)
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) a.g_theta[(size_t)a.slot_row[M::NSLOT + j] * n + i] = 0.f;
    }
  }
  if constexpr (M::NW > 0) {  // the precision network's weight gradient: wavefront shuffle tree, one atomic per wavefront
    if (a.g_weights) {
      VIHDS_UNROLL for (int q = 0; q < M::NW; ++q) {
        float v = live ? ctx.wb[q] : 0.f;
        VIHDS_UNROLL for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&a.g_weights[q], v);
      }
    }
  }
}

// request block for the model launchers (see AdaptiveCtl): mode 1 = forward, 2 = adjoint
struct AdaptiveDevCtl {
  int mode;
  AdaptiveDev dev;
  int result;
};
extern thread_local AdaptiveDevCtl* g_adaptive_dev;

template <class M, int SOLVER>
inline int adaptive_device_s(const OdeArgs& a, const AdaptiveDevCtl& ctl, hipStream_t st) {
  if constexpr (is_blackbox<M>::value) {
    return VIHDS_E_UNSUPPORTED;
  } else {
    if (M::NW != 0 && a.n_hidden_prec >= 1) return VIHDS_E_UNSUPPORTED;  // (a hidden precision layer: the dump-mode adjoint only)
    const int nblk = (a.n + ADP_BLOCK - 1) / ADP_BLOCK;
    if (nblk > ADP_MAX_BLOCKS) return VIHDS_E_UNSUPPORTED;
    if (ctl.mode == 1) {
      if (hipMemsetAsync(ctl.dev.ws, 0, ADP_CTRL * sizeof(float), st) != hipSuccess) return VIHDS_E_HIP;
      hipLaunchKernelGGL((ode_adaptive_fwd_kernel<M, SOLVER>), dim3(nblk), dim3(ADP_BLOCK), 0, st, a, ctl.dev);
    } else {
      hipLaunchKernelGGL((ode_adaptive_bwd_kernel<M, SOLVER>), dim3(nblk), dim3(ADP_BLOCK), 0, st, a, ctl.dev, nblk);
    }
    return VIHDS_OK;
  }
}
template <class M, int ONLY>
inline int adaptive_device(int solver, const OdeArgs& a, const AdaptiveDevCtl& ctl, hipStream_t st) {
  switch (solver) {
    case VIHDS_SOLVER_DOPRI5:
      if constexpr (ONLY < 0 || ONLY == VIHDS_SOLVER_DOPRI5) return adaptive_device_s<M, VIHDS_SOLVER_DOPRI5>(a, ctl, st);
      break;
    case VIHDS_SOLVER_BOSH3:
      if constexpr (ONLY < 0 || ONLY == VIHDS_SOLVER_BOSH3) return adaptive_device_s<M, VIHDS_SOLVER_BOSH3>(a, ctl, st);
      break;
    case VIHDS_SOLVER_ADAPTIVE_HEUN:
      if constexpr (ONLY < 0 || ONLY == VIHDS_SOLVER_ADAPTIVE_HEUN) return adaptive_device_s<M, VIHDS_SOLVER_ADAPTIVE_HEUN>(a, ctl, st);
      break;
  }
  return VIHDS_E_UNSUPPORTED;
}

}  // namespace vihds
