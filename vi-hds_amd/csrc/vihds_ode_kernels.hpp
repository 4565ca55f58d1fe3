// Fused ODE-integration kernels: one thread per (row b, IWAE sample s) trajectory, the whole time loop inside
// the kernel, state and effective parameters in VGPRs, trajectory written [T][N][B][S] so that each store
// instruction of a wave covers 64 consecutive floats.
//
// Forward  = reference OdeModel.simulate (vihds/ode.py:66-82) + observe (:84-93) + expand_precisions
//            (vihds/precisions.py:31-35) + log_prob_observations (vihds/training.py:24-44), fused.
// Backward = discrete adjoint of the chosen scheme, i.e. exactly the gradient autograd produces for the
//            reference's python time loop: for each step (last to first) reload y_k from the stored
//            trajectory, recompute the stage states, and pull the adjoint back through the stages.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "../../include/vihds_hip.h"
#include "vihds_args.hpp"
#include "vihds_models.hpp"
#include "vihds_rk_adaptive.hpp"
#include "vihds_blackbox.hpp"
#include "vihds_iwae_inline.hpp"

namespace vihds {

constexpr float LOG2PI_F = 1.8378770664093453f;  // math.log(2*math.pi), vihds/training.py:43


template <int OBS>
__device__ __forceinline__ void observe(const float* y, float* xp) {
  xp[0] = y[0];
  xp[1] = y[0] * y[1];
  if (OBS == OBS_DEFAULT) {  // vihds/ode.py:84-93
    xp[2] = y[0] * (y[2] + y[4]);
    xp[3] = y[0] * (y[3] + y[5]);
  } else if (OBS == OBS_INDUCER) {  // models/inducer_constant.py:106-114: [OD, OD*RFP, OD*(YFP+F530), OD*F480]
    xp[2] = y[0] * (y[2] + y[3]);
    xp[3] = y[0] * y[4];
  } else {  // models/auto_constant.py:89-97, models/dr_blackbox.py:112-121
    xp[2] = y[0] * y[2];
    xp[3] = y[0] * y[3];
  }
}
template <int OBS>
__device__ __forceinline__ void observe_vjp(const float* y, const float* xpb, float* yb) {
  if (OBS == OBS_DEFAULT) {
    yb[0] += xpb[0] + xpb[1] * y[1] + xpb[2] * (y[2] + y[4]) + xpb[3] * (y[3] + y[5]);
    yb[1] += xpb[1] * y[0];
    yb[2] += xpb[2] * y[0];
    yb[4] += xpb[2] * y[0];
    yb[3] += xpb[3] * y[0];
    yb[5] += xpb[3] * y[0];
  } else if (OBS == OBS_INDUCER) {
    yb[0] += xpb[0] + xpb[1] * y[1] + xpb[2] * (y[2] + y[3]) + xpb[3] * y[4];
    yb[1] += xpb[1] * y[0];
    yb[2] += xpb[2] * y[0];
    yb[3] += xpb[2] * y[0];
    yb[4] += xpb[3] * y[0];
  } else {
    yb[0] += xpb[0] + xpb[1] * y[1] + xpb[2] * y[2] + xpb[3] * y[3];
    yb[1] += xpb[1] * y[0];
    yb[2] += xpb[2] * y[0];
    yb[3] += xpb[3] * y[0];
  }
}

// backward context per model kind
struct NoCtx {};
template <int NW>
struct WeightGradCtx {
  static constexpr bool DUMP = false;
  float wb[NW];
};
// white-box + neural precisions with an aux buffer: instead of 2*4*NIN register accumulators per thread (which spill:
// 0.4-0.5 KB of scratch per lane and a third of the adjoint's time for relay_constant_precisions) the adjoint stores,
// per RHS evaluation, the 8 pre-activation adjoints and the NIN layer inputs field-major [8+NIN][E][n]; the weight
// gradients are then two rectangles of vihds_gram_blocks over that dump.  Only the 8 bias sums stay in registers.
struct PrecDumpCtx {
  static constexpr bool DUMP = true;
  float* dump;     // &aux[i]
  size_t n;        // trajectories
  size_t fstride;  // floats between consecutive fields = E * n
  int e;           // evaluation counter
  float bsum[8];
};
__host__ __device__ inline int ode_stages(int solver) {
  if (solver_is_adaptive(solver)) return adaptive_stages(solver);
  return solver == VIHDS_SOLVER_EULER ? 1 : (solver == VIHDS_SOLVER_RK4 ? 4 : 2);
}
template <class M, bool BB = is_blackbox<M>::value, bool HAS_W = (M::NW > 0)>
struct bwd_ctx {  // white-box
  using type = NoCtx;
  __device__ static void init(type&, const OdeArgs&, int) {}
};
template <class M>
struct bwd_ctx<M, false, true> {  // white-box + neural precisions
  using type = WeightGradCtx<M::NW>;
  __device__ static void init(type& c, const OdeArgs&, int) {
    VIHDS_UNROLL for (int q = 0; q < M::NW; ++q) c.wb[q] = 0.f;
  }
};
template <class M>
struct bwd_ctx<M, true, true> {  // dr_blackbox
  using type = BlackboxCtx;
  __device__ static void init(type& c, const OdeArgs& a, int i) {
    c.dump = a.aux + i;
    c.n = (size_t)a.n;
    c.fstride = (size_t)(a.T - 1) * M::stages(a.solver) * a.n;
    c.e = 0;
    VIHDS_UNROLL for (int k = 0; k < 2 * M::NX + 8; ++k) c.bsum[k] = 0.f;
  }
};

template <class M, bool DUMP>
struct bwd_ctx_sel {
  using impl = bwd_ctx<M>;
};
template <class M>
struct bwd_ctx_sel<M, true> {
  struct impl {
    using type = PrecDumpCtx;
    __device__ static void init(type& c, const OdeArgs& a, int i) {
      c.dump = a.aux + i;
      c.n = (size_t)a.n;
      c.fstride = (size_t)(a.T - 1) * ode_stages(a.solver) * a.n;
      c.e = 0;
      VIHDS_UNROLL for (int k = 0; k < 8; ++k) c.bsum[k] = 0.f;
    }
  };
};

// ---- one step of each scheme ---------------------------------------------------------------------
template <class M, int SOLVER>
__device__ __forceinline__ void ode_step(float t0, float t1, float h0, float* y, const float* p, const float* wts) {
  constexpr int N = M::N;
  if constexpr (solver_is_adaptive(SOLVER)) {  // the accepted grid of an adaptive pair, its higher-order tableau
    rk_step_generic<M, Tableau<SOLVER>>(t0, t1 - t0, y, p, wts, nullptr);
    return;
  }
  float k1[N], k2[N], ya[N];
  if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
    // vihds/solvers.py:12-16 / :21-25
    const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + h * k1[j];
    M::rhs(t1, ya, p, wts, k2);
    const float hh = 0.5f * h;
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y[j] = y[j] + hh * (k1[j] + k2[j]);
  } else if (SOLVER == VIHDS_SOLVER_EULER) {
    const float dt = t1 - t0;
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y[j] = y[j] + dt * k1[j];
  } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
    // torchdiffeq 0.1 Midpoint.step_func: y_mid = y + f(t,y)*dt/2 ; dy = dt*f(t+dt/2, y_mid)
    const float dt = t1 - t0;
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + k1[j] * dt * 0.5f;
    M::rhs(t0 + dt * 0.5f, ya, p, wts, k2);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y[j] = y[j] + dt * k2[j];
  } else {
    // torchdiffeq 0.1 rk4_alt_step_func (3/8 rule)
    const float dt = t1 - t0;
    const float d3 = dt * (1.f / 3.f);  // the per-element k/3 of torchdiffeq become multiplies (1-ulp difference)
    float k3[N], k4[N];
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + d3 * k1[j];
    M::rhs(t0 + d3, ya, p, wts, k2);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + (dt * k2[j] - d3 * k1[j]);
    M::rhs(t0 + 2.f * d3, ya, p, wts, k3);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + dt * (k1[j] - k2[j] + k3[j]);
    M::rhs(t0 + dt, ya, p, wts, k4);
    const float d8 = dt * 0.125f;
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y[j] = y[j] + (k1[j] + 3.f * k2[j] + 3.f * k3[j] + k4[j]) * d8;
  }
}

template <class M, class Ctx>
__device__ __forceinline__ void call_vjp(float t, const float* y, const float* p, const float* w, const float* v,
                                         float* yb, float* pb, Ctx& ctx) {
  if constexpr (std::is_same<Ctx, NoCtx>::value) M::rhs_vjp(t, y, p, w, v, yb, pb);
  else M::rhs_vjp(t, y, p, w, v, yb, pb, ctx);
}

// reverse of one step: lam (adjoint of y_{k+1}) -> adjoint of y_k ; pb += parameter adjoint
template <class M, int SOLVER, class Ctx>
__device__ __forceinline__ void ode_step_vjp(float t0, float t1, float h0, const float* y, const float* p,
                                             const float* wts, float* lam, float* pb, Ctx& wtsb) {
  constexpr int N = M::N;
  if constexpr (solver_is_adaptive(SOLVER)) {
    rk_step_generic_vjp<M, Tableau<SOLVER>>(t0, t1 - t0, y, p, wts, lam, [&](float t, const float* ys, const float* v,
                                                                             float* yb) {
      call_vjp<M>(t, ys, p, wts, v, yb, pb, wtsb);
    });
    return;
  }
  float k1[N], ya[N], v[N], w[N];
  if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
    const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
    const float hh = 0.5f * h;
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + h * k1[j];
    // y' = y + hh*(f1 + f2)
    VIHDS_UNROLL for (int j = 0; j < N; ++j) { v[j] = hh * lam[j]; w[j] = 0.f; }
    call_vjp<M>(t1, ya, p, wts, v, w, pb, wtsb);  // w = ya_bar
    VIHDS_UNROLL for (int j = 0; j < N; ++j) { lam[j] += w[j]; v[j] += h * w[j]; }
    call_vjp<M>(t0, y, p, wts, v, lam, pb, wtsb);
  } else if (SOLVER == VIHDS_SOLVER_EULER) {
    const float dt = t1 - t0;
    VIHDS_UNROLL for (int j = 0; j < N; ++j) v[j] = dt * lam[j];
    call_vjp<M>(t0, y, p, wts, v, lam, pb, wtsb);
  } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
    const float dt = t1 - t0;
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + k1[j] * dt * 0.5f;
    VIHDS_UNROLL for (int j = 0; j < N; ++j) { v[j] = dt * lam[j]; w[j] = 0.f; }
    call_vjp<M>(t0 + dt * 0.5f, ya, p, wts, v, w, pb, wtsb);  // w = ymid_bar
    VIHDS_UNROLL for (int j = 0; j < N; ++j) { lam[j] += w[j]; v[j] = 0.5f * dt * w[j]; }
    call_vjp<M>(t0, y, p, wts, v, lam, pb, wtsb);
  } else {
    const float dt = t1 - t0;
    const float d3 = dt * (1.f / 3.f);
    float k2[N], k3[N], y2[N], y3[N];
    M::rhs(t0, y, p, wts, k1);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y2[j] = y[j] + d3 * k1[j];
    M::rhs(t0 + d3, y2, p, wts, k2);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y3[j] = y[j] + (dt * k2[j] - d3 * k1[j]);
    M::rhs(t0 + 2.f * d3, y3, p, wts, k3);
    VIHDS_UNROLL for (int j = 0; j < N; ++j) ya[j] = y[j] + dt * (k1[j] - k2[j] + k3[j]);  // y4
    const float d8 = dt * 0.125f;
    float k1b[N], k2b[N], k3b[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      v[j] = d8 * lam[j];  // k4_bar
      k1b[j] = v[j];
      k2b[j] = 3.f * v[j];
      k3b[j] = 3.f * v[j];
      w[j] = 0.f;
    }
    call_vjp<M>(t0 + dt, ya, p, wts, v, w, pb, wtsb);  // w = y4_bar
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      lam[j] += w[j];
      k1b[j] += dt * w[j];
      k2b[j] -= dt * w[j];
      k3b[j] += dt * w[j];
      w[j] = 0.f;
    }
    call_vjp<M>(t0 + 2.f * d3, y3, p, wts, k3b, w, pb, wtsb);  // w = y3_bar
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      lam[j] += w[j];
      k1b[j] -= d3 * w[j];
      k2b[j] += dt * w[j];
      w[j] = 0.f;
    }
    call_vjp<M>(t0 + d3, y2, p, wts, k2b, w, pb, wtsb);  // w = y2_bar
    VIHDS_UNROLL for (int j = 0; j < N; ++j) {
      lam[j] += w[j];
      k1b[j] += d3 * w[j];
    }
    call_vjp<M>(t0, y, p, wts, k1b, lam, pb, wtsb);
  }
}

template <class M>
__device__ __forceinline__ void load_theta(const OdeArgs& a, int i, int b, float* th, float* prec, float* c) {
  VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) th[q] = a.theta[(size_t)a.slot_row[q] * a.n + i];
  if (!M::NEURAL_PREC) {  // constant precisions: four more theta rows (reference precisions.py:31-35)
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) prec[j] = a.theta[(size_t)a.slot_row[M::NSLOT + j] * a.n + i];
  }
  VIHDS_UNROLL for (int q = 0; q < M::NC; ++q) c[q] = clampf(expf(a.cond[b * a.C + q]) - 1.f, 1e-12f, 1e6f);
}

// Shared neural weights: dr_blackbox stages them in LDS once per block (every lane reads the same address:
// broadcast); the white-box models with neural precisions read them as scalars straight from the caller's buffer
// (WithPrec::weights_ptr).
template <class M>
__device__ __forceinline__ const float* stage_weights(const OdeArgs& a, float* lds) {
  if constexpr (M::NW == 0) {
    return nullptr;
  } else if constexpr (is_blackbox<M>::value) {
    M::stage(a, lds);
    __syncthreads();
    return lds;
  } else {
    return a.weights;
  }
}

#ifndef VIHDS_NT_TRAJ
#define VIHDS_NT_TRAJ 1
#endif
// ---- forward -------------------------------------------------------------------------------------
// LDS_IN: the time grid and the observation rows of the data rows this block spans are staged in LDS once, so the
// time loop holds no vector-memory loads and its trajectory / x_predict stores are never waited on (see
// vihds_dr_lanes.hpp: with a global load in the loop every s_waitcnt vmcnt(0) for it also waits for the stores).
template <class M, int SOLVER, bool LDS_IN>
__global__ void __launch_bounds__(256) ode_fwd_kernel(OdeArgs a) {
  constexpr int N = M::N;
  __shared__ float wlds[M::NW > 0 ? M::NW : 1];
  extern __shared__ float in_lds[];  // [T] times | [nb][4][T] observations
  const float* wts = stage_weights<M>(a, wlds);
  int ob_off = 0;
  if (LDS_IN) {
    const int first = blockIdx.x * blockDim.x, last = min(first + (int)blockDim.x, a.n) - 1;
    const int b0 = first / a.S, nb = last / a.S - b0 + 1;
    for (int q = threadIdx.x; q < a.T; q += blockDim.x) in_lds[q] = a.times[q];
    const float* src = a.obs + (size_t)b0 * 4 * a.T;
    for (int q = threadIdx.x; q < nb * 4 * a.T; q += blockDim.x) in_lds[a.T + q] = src[q];
    __syncthreads();
    const int ii = min((int)(blockIdx.x * blockDim.x + threadIdx.x), a.n - 1);
    ob_off = a.T + (ii / a.S - b0) * 4 * a.T;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int b = i / a.S;
  float th[M::NSLOT], prec[4], c[M::NC > 0 ? M::NC : 1], p[M::NP], y[N];
  load_theta<M>(a, i, b, th, prec, c);
  if constexpr (is_blackbox<M>::value) {
    M::prepare_bb(th, a, b, p);
    M::init_bb(th, a, y);
  } else {
    M::prepare(th, c, p);
    if constexpr (M::NEURAL_PREC) p[M::NP - 1] = __int_as_float(a.n_hidden_prec);
    M::init(th, c, y);
  }

  float lc[4], lp[4];
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
    lc[j] = M::NEURAL_PREC ? 0.f : LOG2PI_F - logf(prec[j]);
    lp[j] = 0.f;
  }
  const float* ob = a.obs + (size_t)b * 4 * a.T;
  auto time_at = [&](int k) { return LDS_IN ? in_lds[k] : a.times[k]; };
  auto obs_at = [&](int j, int k) { return LDS_IN ? in_lds[ob_off + j * a.T + k] : ob[j * a.T + k]; };
  const float h0 = a.times[1] - a.times[0];
  const size_t n = a.n;

  // A trajectory larger than the chip's caches (the evaluation shape: 644 MB at B=234, S=1000) is written with streaming
  // (non-temporal) stores: its lines are not kept dirty in L2 / Infinity Cache, so the summaries launch that follows does not
  // share HBM with their write-back.  Training shapes keep ordinary stores (the adjoint re-reads them from cache).
  const bool nt_traj = VIHDS_NT_TRAJ && (size_t)a.n * N * a.T * sizeof(float) > ((size_t)256 << 20);
  // times / observations are fetched one step ahead so their latency is off the dependent chain
  float tA = time_at(0), tB = time_at(1);
  float obc[4], obn[4];
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) { obc[j] = a.logp ? obs_at(j, 0) : 0.f; obn[j] = 0.f; }
  for (int k = 0; k < a.T; ++k) {
    const float tC = (k + 1 < a.T) ? time_at(k + 1) : tB;
    if (a.logp && k + 1 < a.T) {
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) obn[j] = obs_at(j, k + 1);
    }
    if (k > 0) {
      ode_step<M, SOLVER>(tA, tB, h0, y, p, wts);
      tA = tB;
    }
    tB = tC;
    if (a.traj) {
      if (nt_traj) {  // (streaming stores at evaluation-sized launches: see nt_traj)
        VIHDS_UNROLL for (int j = 0; j < N; ++j) __builtin_nontemporal_store(y[j], &a.traj[((size_t)k * N + j) * n + i]);
      } else {
        VIHDS_UNROLL for (int j = 0; j < N; ++j) a.traj[((size_t)k * N + j) * n + i] = y[j];
      }
    }
    float xp[4];
    observe<M::OBS>(y, xp);
    if (a.xpred) {
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) a.xpred[((size_t)k * 4 + j) * n + i] = xp[j];
    }
    if (a.logp) {
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float e = xp[j] - obc[j];
        if (M::NEURAL_PREC) {  // precisions are ODE states (reference precisions.py:89-94)
          const float pr = y[(M::N - 4) + j];
          lp[j] += -0.5f * (LOG2PI_F - logf(pr) + pr * e * e);
        } else {
          lp[j] += -0.5f * (lc[j] + prec[j] * e * e);
        }
        obc[j] = obn[j];
      }
    }
  }
  if (a.logp) {
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) a.logp[(size_t)j * n + i] = lp[j];
  }
}

// ---- evaluation summaries without the trajectory's round trip through HBM (round 5) -------------------------------------
// Results.init (reference vihds/utils.py:79-99) needs, per data row and time point, sums over the row's samples weighted with
// the NORMALISED importance weights -- which exist only when every sample's last time point does.  The evaluation pass used
// to write the trajectory (644 MB at B = 234, S = 1 000) and stream it back through vihds_iw_summaries_states; now a first
// forward launch writes the log-likelihoods only (traj == NULL), the weights are formed, and THIS kernel integrates again
// and adds up on the way: per time point NV = n_species + 12 weighted values per lane, summed over the wavefront with
// gfx950's row swaps (2.5 instructions per value instead of 6 DPP additions), one partial row per wavefront and time point
// (26 MB at that size), which summ_finish_kernel adds up in a fixed order.  The integration is ode_fwd_kernel's (same
// ode_step, same operands): the trajectory summed here is bit for bit the one the weights were computed from.
// grid: (ceil(S / 256), B), 256 threads: every wavefront's 64 samples belong to one data row.
extern thread_local const SummArgs* g_summ;

// v[0 .. NVP): per-lane values -> u[p] (p < NVP / 4): in the lanes of row r (lanes 16 r .. 16 r + 15) the wavefront's total of
// value 4 p + {0, 2, 1, 3}[r]
// (NR <= NVP / 4 output registers: only the first 4 NR values are summed.  A swap costs the issuing wavefront 15 cycles, 23
// with a nop of its own -- tests/micro/permlane_rate.hip)
template <int NVP, int NR>
__device__ __forceinline__ void wave_sum_rows(const float (&v)[NVP], float (&u)[NVP / 4]) {
  static_assert(NVP % 4 == 0 && NR <= NVP / 4, "pad the values to a multiple of four");
  float a[2 * NR], b[2 * NR];
#pragma unroll
  for (int p = 0; p < 2 * NR; ++p) { a[p] = v[2 * p]; b[p] = v[2 * p + 1]; }
  // lanes l and l ^ 32: rows 0, 1 keep value 2 p, rows 2, 3 value 2 p + 1
  // (two swaps per asm block behind ONE s_nop: the compiler may put the copies that feed a swap right in front of its
  // block, and the swap reads a VALU result two wait states late at the earliest)
#pragma unroll
  for (int p = 0; p < 2 * NR; p += 2)
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3"
                 : "+v"(a[p]), "+v"(b[p]), "+v"(a[p + 1]), "+v"(b[p + 1]));
  float c[NR], d[NR];
#pragma unroll
  for (int p = 0; p < NR; ++p) { c[p] = a[2 * p] + b[2 * p]; d[p] = a[2 * p + 1] + b[2 * p + 1]; }
  // rows 0 + 1 and 2 + 3 of each: one value per row
#pragma unroll
  for (int p = 0; p + 1 < NR; p += 2)
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3"
                 : "+v"(c[p]), "+v"(d[p]), "+v"(c[p + 1]), "+v"(d[p + 1]));
  if (NR & 1) asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c[NR - 1]), "+v"(d[NR - 1]));
#pragma unroll
  for (int p = 0; p < NR; ++p) u[p] = c[p] + d[p];
  // all 16 lanes of a row: rotations by 8, 4, 2, 1 (every lane ends with the row's total), the registers side by side
#pragma unroll
  for (int p = 0; p < NR; ++p) u[p] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u[p]), 0x128, 0xf, 0xf, false));
#pragma unroll
  for (int p = 0; p < NR; ++p) u[p] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u[p]), 0x124, 0xf, 0xf, false));
#pragma unroll
  for (int p = 0; p < NR; ++p) u[p] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u[p]), 0x122, 0xf, 0xf, false));
#pragma unroll
  for (int p = 0; p < NR; ++p) u[p] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, u[p]), 0x121, 0xf, 0xf, false));
}

template <class M>
struct SummLayout {
  static constexpr int NS = M::NEURAL_PREC ? M::N - 4 : M::N;  // species (the last four states of a neural-precision model are the precisions)
  static constexpr int NV = NS + 12, NVP = (NV + 3) & ~3;      // states | w x | w (x^2 + 1 / prec) | w / prec
};

template <class M, int SOLVER>
__global__ void __launch_bounds__(256) ode_fwd_summ_kernel(OdeArgs a, SummArgs sa) {
  constexpr int N = M::N;
  using L = SummLayout<M>;
  constexpr int NS = L::NS, NVP = L::NVP;
  __shared__ float wlds[M::NW > 0 ? M::NW : 1];
  extern __shared__ float in_lds[];  // [T] times
  const float* wts = stage_weights<M>(a, wlds);
  for (int q = threadIdx.x; q < a.T; q += blockDim.x) in_lds[q] = a.times[q];
  __syncthreads();
  const int b = blockIdx.y;
  const int s0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = s0 < a.S;
  const int i = b * a.S + (live ? s0 : a.S - 1);  // (idle lanes shadow the row's last sample at weight 0)
  float th[M::NSLOT], prec[4], c[M::NC > 0 ? M::NC : 1], p[M::NP], y[N];
  load_theta<M>(a, i, b, th, prec, c);
  M::prepare(th, c, p);
  if constexpr (M::NEURAL_PREC) p[M::NP - 1] = __int_as_float(a.n_hidden_prec);
  M::init(th, c, y);
  const float w = live ? expf(sa.log_w[i] - sa.lse[b]) : 0.f;
  float ivc[4];  // 1 / precision (constant precisions)
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) ivc[j] = M::NEURAL_PREC ? 0.f : 1.f / prec[j];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the wavefront's partial rows [T][NVP]; lane 16 r writes the values of row r (wave_sum_rows)
  float* out = sa.partial + ((size_t)b * sa.nch + (size_t)blockIdx.x * (blockDim.x >> 6) + wave) * (size_t)a.T * NVP;
  const int row = lane >> 4;
  const int slot = row == 1 ? 2 : (row == 2 ? 1 : row);
  const bool writer = (lane & 15) == 0;
  float* p_out = out + slot;
  const float h0 = a.times[1] - a.times[0];
  float tA = in_lds[0], tB = in_lds[a.T > 1 ? 1 : 0];
  __builtin_amdgcn_s_waitcnt(0);  // (nothing loaded before the loop is waited for inside it, behind the loop's stores)
  for (int k = 0; k < a.T; ++k) {
    const float tC = (k + 1 < a.T) ? in_lds[k + 1] : tB;
    if (k > 0) {
      ode_step<M, SOLVER>(tA, tB, h0, y, p, wts);
      tA = tB;
    }
    tB = tC;
    float xp[4];
    observe<M::OBS>(y, xp);
    float v[NVP];
    VIHDS_UNROLL for (int j = 0; j < NVP; ++j) v[j] = 0.f;
    VIHDS_UNROLL for (int j = 0; j < NS; ++j) v[j] = w * y[j];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      const float iv = M::NEURAL_PREC ? 1.f / y[NS + j] : ivc[j];  // (per-sample terms as vihds_iw_summaries forms them)
      v[NS + j] = w * xp[j];
      v[NS + 4 + j] = w * (xp[j] * xp[j] + iv);
      v[NS + 8 + j] = w * iv;
    }
    // constant precisions: the four sums of w / precision do not depend on the time point -- they are taken at the first one
    // only (summ_finish_kernel hands them to every time point)
    constexpr int NR_ALL = NVP / 4, NR_STEP = M::NEURAL_PREC ? NR_ALL : (NS + 8 + 3) / 4;
    float u[NVP / 4];
    if (k == 0) {
      wave_sum_rows<NVP, NR_ALL>(v, u);
      if (writer) {
        VIHDS_UNROLL for (int q = 0; q < NR_ALL; ++q) p_out[4 * q] = u[q];
      }
    } else {
      wave_sum_rows<NVP, NR_STEP>(v, u);
      if (writer) {
        VIHDS_UNROLL for (int q = 0; q < NR_STEP; ++q) p_out[4 * q] = u[q];
      }
    }
    p_out += NVP;
  }
}
template <class M, int ONLY>
inline int launch_fwd_summ(int solver, const OdeArgs& a, const SummArgs& sa, hipStream_t st) {
  if constexpr (is_blackbox<M>::value) return VIHDS_E_UNSUPPORTED;
  else {
    if (sa.nvp != SummLayout<M>::NVP || sa.nch != ((a.S + 255) / 256) * 4) return VIHDS_E_BADARG;
    const dim3 grid((a.S + 255) / 256, a.B);
    const size_t lds = (size_t)a.T * sizeof(float);
#define VIHDS_SUMM_CASE(SV)                                                                                         \
  case SV:                                                                                                        \
    if constexpr (ONLY < 0 || SV == ONLY) {                                                                       \
      hipLaunchKernelGGL((ode_fwd_summ_kernel<M, SV>), grid, dim3(256), lds, st, a, sa);                           \
      return VIHDS_OK;                                                                                            \
    }                                                                                                             \
    break;
    switch (solver) {
      VIHDS_SUMM_CASE(VIHDS_SOLVER_MODEULER)
      VIHDS_SUMM_CASE(VIHDS_SOLVER_MODEULERWHILE)
      VIHDS_SUMM_CASE(VIHDS_SOLVER_EULER)
      VIHDS_SUMM_CASE(VIHDS_SOLVER_MIDPOINT)
      VIHDS_SUMM_CASE(VIHDS_SOLVER_RK4)
    }
#undef VIHDS_SUMM_CASE
    return VIHDS_E_UNSUPPORTED;  // (the adaptive pairs keep the two-kernel form)
  }
}

// ---- backward ------------------------------------------------------------------------------------
template <class M, int SOLVER, bool DUMP = false>
__global__ void __launch_bounds__(256) ode_bwd_kernel(OdeArgs a) {
  constexpr int N = M::N;
  __shared__ float wlds[M::NW > 0 ? M::NW : 1];
  const float* wts = stage_weights<M>(a, wlds);
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;  // tail lanes shadow the last trajectory (they take part in the reductions)
  const int b = i / a.S;
  float th[M::NSLOT], prec[4], c[M::NC > 0 ? M::NC : 1], p[M::NP];
  load_theta<M>(a, i, b, th, prec, c);
  if constexpr (is_blackbox<M>::value) {
    M::prepare_bb(th, a, b, p);
  } else {
    M::prepare(th, c, p);
    if constexpr (M::NEURAL_PREC) p[M::NP - 1] = __int_as_float(a.n_hidden_prec);
  }

  float lam[N], pb[M::NP], precb[4], glp[4];
  VIHDS_UNROLL for (int j = 0; j < N; ++j) lam[j] = 0.f;
  VIHDS_UNROLL for (int j = 0; j < M::NP; ++j) pb[j] = 0.f;
  // backward context: per-thread shared-weight gradient accumulators (white-box + neural precisions), or the
  // evaluation dump cursor (dr_blackbox)
  using CtxImpl = typename bwd_ctx_sel<M, DUMP>::impl;
  typename CtxImpl::type wtsb;
  CtxImpl::init(wtsb, a, i);
  const size_t n = a.n;
  const float w_iw = a.iw_logp ? iw_wave_weight(a, i, b) : 0.f;
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
    precb[j] = 0.f;
    glp[j] = ode_logp_grad(a, w_iw, i, j);
  }
  const float* ob = a.obs + (size_t)b * 4 * a.T;
  const float h0 = a.times[1] - a.times[0];

  // software pipeline: the stored state, observations and times of step k-1 are requested while step k computes
  float yn[N], obn[4];
  VIHDS_UNROLL for (int j = 0; j < N; ++j) yn[j] = a.traj_in[((size_t)(a.T - 1) * N + j) * n + i];
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) obn[j] = ob[j * a.T + a.T - 1];
  float tHi = a.times[a.T - 1], tLo = tHi;
  for (int k = a.T - 1; k >= 0; --k) {
    float y[N], obk[4];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) y[j] = yn[j];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) obk[j] = obn[j];
    const float tK = tLo;
    if (k > 0) {
      VIHDS_UNROLL for (int j = 0; j < N; ++j) yn[j] = a.traj_in[((size_t)(k - 1) * N + j) * n + i];
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) obn[j] = ob[j * a.T + k - 1];
      tLo = a.times[k - 1];
    }
    if (k < a.T - 1) ode_step_vjp<M, SOLVER>(tK, tHi, h0, y, p, wts, lam, pb, wtsb);
    tHi = tK;
    // gradient injected at time k: log-likelihood term, x_predict and trajectory upstream grads
    float xp[4], xpb[4];
    observe<M::OBS>(y, xp);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      const float e = xp[j] - obk[j];
      const float pr = M::NEURAL_PREC ? y[(M::N - 4) + j] : prec[j];
      xpb[j] = -glp[j] * pr * e;
      const float prb = glp[j] * (0.5f / pr - 0.5f * e * e);
      if (M::NEURAL_PREC) lam[(M::N - 4) + j] += prb;
      else precb[j] += prb;
      if (a.g_xpred) xpb[j] += a.g_xpred[((size_t)k * 4 + j) * n + i];
    }
    observe_vjp<M::OBS>(y, xpb, lam);
    if (a.g_traj) {
      VIHDS_UNROLL for (int j = 0; j < N; ++j) lam[j] += a.g_traj[((size_t)k * N + j) * n + i];
    }
  }
  float thb[M::NSLOT];
  VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) thb[q] = 0.f;
  if constexpr (is_blackbox<M>::value) {
    M::prepare_vjp_bb(th, a, b, pb, thb);
    if (live) M::store_delta(a, i, pb, wtsb);
  } else {
    M::prepare_vjp(th, c, p, pb, thb);
  }
  M::init_vjp(lam, thb);
  if (live) {
    VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) a.g_theta[(size_t)a.slot_row[q] * n + i] = thb[q];
    if (!M::NEURAL_PREC) {
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) a.g_theta[(size_t)a.slot_row[M::NSLOT + j] * n + i] = precb[j];
    }
  }
  if constexpr (!is_blackbox<M>::value && M::NW > 0) {
    if (a.g_weights) {
      if constexpr (DUMP) {
        // only the biases are accumulated here; the weight matrices come from the dump (vihds_gram_blocks)
        VIHDS_UNROLL for (int q = 0; q < 8; ++q) {
          float v = live ? wtsb.bsum[q] : 0.f;
          VIHDS_UNROLL for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
          if ((threadIdx.x & 63) == 0)
            atomicAdd(&a.g_weights[(q < 4 ? M::o_bp(a.n_hidden_prec) + q : M::o_bd(a.n_hidden_prec) + (q - 4))], v);
        }
      } else if (a.n_hidden_prec < 1) {
        // shared-weight gradient: per-thread register accumulators -> wave shuffle tree -> one atomic per wave
        VIHDS_UNROLL for (int q = 0; q < M::NW; ++q) {
          float v = live ? wtsb.wb[q] : 0.f;
          VIHDS_UNROLL for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
          if ((threadIdx.x & 63) == 0) atomicAdd(&a.g_weights[q], v);
        }
      }
    }
  }
}

inline int pick_block(int n) { return n >= (1 << 18) ? 256 : 64; }

template <class M, int SOLVER>
inline void launch_fwd_s(const OdeArgs& a, hipStream_t st) {
  const int blk = pick_block(a.n);
  const int nb = min(a.B, (blk - 1) / a.S + 2);
  const size_t lds = ((size_t)a.T + (size_t)nb * 4 * a.T) * sizeof(float);
  if (lds <= 32 * 1024)
    hipLaunchKernelGGL((ode_fwd_kernel<M, SOLVER, true>), dim3((a.n + blk - 1) / blk), dim3(blk), lds, st, a);
  else
    hipLaunchKernelGGL((ode_fwd_kernel<M, SOLVER, false>), dim3((a.n + blk - 1) / blk), dim3(blk), 0, st, a);
}
template <class M, int SOLVER>
inline void launch_bwd_s(const OdeArgs& a, hipStream_t st) {
  const int blk = pick_block(a.n);
  if constexpr (M::NEURAL_PREC && !is_blackbox<M>::value) {
    if (a.aux) {  // dump mode: the caller contracts the weight gradients (vihds_ode_bwd, include/vihds_hip.h)
      hipLaunchKernelGGL((ode_bwd_kernel<M, SOLVER, true>), dim3((a.n + blk - 1) / blk), dim3(blk), 0, st, a);
      return;
    }
  }
  hipLaunchKernelGGL((ode_bwd_kernel<M, SOLVER>), dim3((a.n + blk - 1) / blk), dim3(blk), 0, st, a);
}

// Request block of vihds_ode_adaptive_grid: when set (by the API, on the calling thread), the model's launcher runs the
// step-size controller instead of a forward / adjoint launch and leaves the grid length (or an error code) in `result`.
struct AdaptiveCtl {
  const float* times_host;
  float rtol, atol;
  float* workspace;
  float* grid_host;
  int max_grid;
  int* index_host;
  int result;
};
extern thread_local AdaptiveCtl* g_adaptive_ctl;
// vihds_theta_ode_fwd: the sampling stage to run in front of the forward launch (NULL: none); a launch function whose kernels
// have no such stage returns VIHDS_E_UNSUPPORTED when it is set
extern thread_local const ThetaStageArgs* g_theta_stage;
template <class M, int ONLY>
inline int adaptive_grid(int solver, const OdeArgs& a, const float* times_host, float rtol, float atol, float* workspace,
                         float* grid_host, int max_grid, int* index_host, hipStream_t st);

// -DVIHDS_ONLY_SOLVER=<id>: this translation unit holds the kernels of one solver only (the per-size dr_blackbox side
// libraries compile their eight solvers as eight objects in parallel: sized/ode_dr_blackbox_sized.hip)
// (ONLY is a template parameter of the dispatchers so that the eight objects hold eight DIFFERENT functions: the same
// inline function with different bodies would be folded into one by the linker)
#ifdef VIHDS_ONLY_SOLVER
constexpr int kOnlySolver = VIHDS_ONLY_SOLVER;
#else
constexpr int kOnlySolver = -1;
#endif
template <int ONLY>
constexpr bool solver_built_in(int sv) { return ONLY < 0 || sv == ONLY; }

}  // namespace vihds
#include "vihds_rk_adaptive_device.hpp"
namespace vihds {

template <class M, int ONLY = kOnlySolver>
inline int launch_ode(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  if (const SummArgs* sm = g_summ) {  // vihds_ode_fwd_summaries: the evaluation's second forward pass
    return backward ? VIHDS_E_BADARG : launch_fwd_summ<M, ONLY>(solver, a, *sm, st);
  }
  if (AdaptiveDevCtl* dc = g_adaptive_dev) {  // vihds_ode_adaptive_fwd / _bwd: the device-resident controller and its adjoint
    dc->result = adaptive_device<M, ONLY>(solver, a, *dc, st);
    return dc->result;
  }
  if (AdaptiveCtl* ctl = g_adaptive_ctl) {
    ctl->result = adaptive_grid<M, ONLY>(solver, a, ctl->times_host, ctl->rtol, ctl->atol, ctl->workspace, ctl->grid_host,
                                   ctl->max_grid, ctl->index_host, st);
    return ctl->result < 0 ? ctl->result : VIHDS_OK;
  }
#define VIHDS_CASE(SV)                                           \
  case SV:                                                       \
    if constexpr (solver_built_in<ONLY>(SV)) {                            \
      if (backward) launch_bwd_s<M, SV>(a, st);                  \
      else launch_fwd_s<M, SV>(a, st);                           \
      return VIHDS_OK;                                           \
    }                                                            \
    break;
  switch (solver) {
    VIHDS_CASE(VIHDS_SOLVER_MODEULER)
    VIHDS_CASE(VIHDS_SOLVER_MODEULERWHILE)
    VIHDS_CASE(VIHDS_SOLVER_EULER)
    VIHDS_CASE(VIHDS_SOLVER_MIDPOINT)
    VIHDS_CASE(VIHDS_SOLVER_RK4)
    VIHDS_CASE(VIHDS_SOLVER_DOPRI5)
    VIHDS_CASE(VIHDS_SOLVER_BOSH3)
    VIHDS_CASE(VIHDS_SOLVER_ADAPTIVE_HEUN)
    VIHDS_CASE(VIHDS_SOLVER_DOPRI8)
  }
#undef VIHDS_CASE
  return VIHDS_E_BADARG;
}

// ---- step-size controller of the adaptive pairs (vihds_rk_adaptive.hpp) ---------------------------------------------
// State of the whole batch in a caller-provided workspace Y [N][n]; every launch writes one float per block to
// `partial` (sums of squares), which the host adds up in block order (deterministic).
// MODE 0: Y <- initial state; partial[0][blk] = sum (y0 / scale)^2, partial[1][blk] = sum (f0 / scale)^2   (scale = atol + rtol |y0|)
// MODE 1: partial[0][blk] = sum ((f(t + h, y0 + h f0) - f0) / scale)^2                                     (initial step)
// MODE 2: trial step of size h from Yin at t: Yout <- y', partial[0][blk] = sum (err / (atol + rtol max(|y|, |y'|)))^2
template <class M, int SOLVER, int MODE>
__global__ void __launch_bounds__(256) ode_trial_kernel(OdeArgs a, const float* Yin, float* Yout, float t, float h,
                                                        float rtol, float atol, float* partial) {
  constexpr int N = M::N;
  __shared__ float wlds[M::NW > 0 ? M::NW : 1];
  __shared__ float red[2][256];
  const float* wts = stage_weights<M>(a, wlds);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  float s0 = 0.f, s1 = 0.f;
  if (i < a.n) {
    const int b = i / a.S;
    float th[M::NSLOT], prec[4], c[M::NC > 0 ? M::NC : 1], p[M::NP], y[N];
    load_theta<M>(a, i, b, th, prec, c);
    if constexpr (is_blackbox<M>::value) {
      M::prepare_bb(th, a, b, p);
      M::init_bb(th, a, y);
    } else {
      M::prepare(th, c, p);
      if constexpr (M::NEURAL_PREC) p[M::NP - 1] = __int_as_float(a.n_hidden_prec);
      M::init(th, c, y);
    }
    const size_t n = a.n;
    if (MODE == 0) {
      float f0[N];
      M::rhs(t, y, p, wts, f0);
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        Yout[(size_t)j * n + i] = y[j];
        const float sc = atol + rtol * fabsf(y[j]);
        s0 += (y[j] / sc) * (y[j] / sc);
        s1 += (f0[j] / sc) * (f0[j] / sc);
      }
    } else if (MODE == 1) {
      float f0[N], f1[N], y1[N];
      M::rhs(t, y, p, wts, f0);
      VIHDS_UNROLL for (int j = 0; j < N; ++j) y1[j] = y[j] + h * f0[j];
      M::rhs(t + h, y1, p, wts, f1);
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        const float sc = atol + rtol * fabsf(y[j]);
        const float d = (f1[j] - f0[j]) / sc;
        s0 += d * d;
      }
    } else {
      float err[N], y0[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) { y[j] = Yin[(size_t)j * n + i]; y0[j] = y[j]; }
      rk_step_generic<M, Tableau<SOLVER>>(t, h, y, p, wts, err);
      VIHDS_UNROLL for (int j = 0; j < N; ++j) {
        Yout[(size_t)j * n + i] = y[j];
        const float tol = atol + rtol * fmaxf(fabsf(y0[j]), fabsf(y[j]));
        const float r = err[j] / tol;
        s0 += r * r;
      }
    }
  }
  red[0][threadIdx.x] = s0;
  red[1][threadIdx.x] = s1;
  __syncthreads();
  for (int w = blockDim.x / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[0][threadIdx.x] += red[0][threadIdx.x + w];
      red[1][threadIdx.x] += red[1][threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0][0];
    partial[gridDim.x + blockIdx.x] = red[1][0];
  }
}

// Host-driven controller (synchronous on `st`).  Workspace: 2 N n floats of state + 2 blocks floats of partial sums.
// Returns the number of grid points written to grid_host (<= max_grid), or a negative error code.
template <class M, int SOLVER>
inline int adaptive_grid_s(const OdeArgs& a, const float* times_host, float rtol, float atol, float* workspace,
                           float* grid_host, int max_grid, int* index_host, hipStream_t st) {
  using TB = Tableau<SOLVER>;
  constexpr int N = M::N;
  const int blk = 256, nblk = (a.n + blk - 1) / blk;
  float* Y0 = workspace;
  float* Y1 = Y0 + (size_t)N * a.n;
  float* partial = Y1 + (size_t)N * a.n;
  std::vector<float> hp(2 * (size_t)nblk);
  auto sums = [&](double* s0, double* s1) -> bool {
    if (hipMemcpyAsync(hp.data(), partial, hp.size() * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
    if (hipStreamSynchronize(st) != hipSuccess) return false;
    double u = 0.0, v = 0.0;
    for (int q = 0; q < nblk; ++q) { u += hp[q]; v += hp[nblk + q]; }
    *s0 = u;
    if (s1) *s1 = v;
    return true;
  };
  const double cnt = (double)N * (double)a.n;
  const float t0 = times_host[0];
  // initial step (torchdiffeq 0.1 _select_initial_step) [recalled]
  double q0, q1, q2;
  hipLaunchKernelGGL((ode_trial_kernel<M, SOLVER, 0>), dim3(nblk), dim3(blk), 0, st, a, nullptr, Y0, t0, 0.f, rtol, atol, partial);
  if (!sums(&q0, &q1)) return VIHDS_E_HIP;
  const double d0 = std::sqrt(q0 / cnt), d1 = std::sqrt(q1 / cnt);
  double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
  hipLaunchKernelGGL((ode_trial_kernel<M, SOLVER, 1>), dim3(nblk), dim3(blk), 0, st, a, nullptr, Y0, t0, (float)h0, rtol, atol, partial);
  if (!sums(&q2, nullptr)) return VIHDS_E_HIP;
  const double d2 = std::sqrt(q2 / cnt) / h0;
  double h1 = (d1 <= 1e-15 && d2 <= 1e-15) ? std::max(1e-6, h0 * 1e-3) : std::pow(0.01 / std::max(d1, d2), 1.0 / (TB::ORDER + 1));
  double h = std::min(100.0 * h0, h1);
  // accepted steps, clipped to the output times
  int ng = 0;
  grid_host[ng++] = t0;
  index_host[0] = 0;
  float t = t0;
  for (int k = 1; k < a.T; ++k) {
    const float t_out = times_host[k];
    while (t < t_out) {
      if (ng >= max_grid) return VIHDS_E_UNSUPPORTED;
      const bool clip = t + (float)h >= t_out;
      const float t_next = clip ? t_out : t + (float)h;
      const float hs = t_next - t;
      if (!(hs > 0.f)) return VIHDS_E_BADARG;  // step size underflow
      hipLaunchKernelGGL((ode_trial_kernel<M, SOLVER, 2>), dim3(nblk), dim3(blk), 0, st, a, Y0, Y1, t, hs, rtol, atol, partial);
      double e2;
      if (!sums(&e2, nullptr)) return VIHDS_E_HIP;
      const double ratio = e2 / cnt;  // mean squared error ratio
      if (!(ratio == ratio)) return VIHDS_E_BADARG;
      const bool accept = ratio <= 1.0;
      // torchdiffeq 0.1 _optimal_step_size(last_step, mean_error_ratio, safety 0.9, ifactor 10, dfactor 0.2, order)
      double hn;
      if (ratio == 0.0) hn = (double)hs * 10.0;
      else {
        const double dfactor = ratio < 1.0 ? 1.0 : 0.2;
        const double er = std::sqrt(ratio);
        const double factor = std::max(1.0 / 10.0, std::min(std::pow(er, 1.0 / TB::ORDER) / 0.9, 1.0 / dfactor));
        hn = (double)hs / factor;
      }
      if (accept) {
        t = t_next;
        std::swap(Y0, Y1);
        grid_host[ng++] = t;
        // a clipped step says little about the step the controller wanted: keep the larger proposal
        h = clip ? std::max(h, hn) : hn;
      } else {
        h = hn;
      }
    }
    index_host[k] = ng - 1;
  }
  return ng;
}

template <class M, int ONLY>
inline int adaptive_grid(int solver, const OdeArgs& a, const float* times_host, float rtol, float atol, float* workspace,
                         float* grid_host, int max_grid, int* index_host, hipStream_t st) {
#define VIHDS_CASE(SV)       \
  case SV:                   \
    if constexpr (solver_built_in<ONLY>(SV)) return adaptive_grid_s<M, SV>(a, times_host, rtol, atol, workspace, grid_host, max_grid, index_host, st); \
    break;
  switch (solver) {
    VIHDS_CASE(VIHDS_SOLVER_DOPRI5)
    VIHDS_CASE(VIHDS_SOLVER_BOSH3)
    VIHDS_CASE(VIHDS_SOLVER_ADAPTIVE_HEUN)
    VIHDS_CASE(VIHDS_SOLVER_DOPRI8)
  }
#undef VIHDS_CASE
  return VIHDS_E_BADARG;
}

}  // namespace vihds
