// dr_blackbox on the matrix cores with the two networks of the right-hand side on two wavefronts.
//
// One wavefront per 16 trajectories (vihds_blackbox_mfma.hpp) walks 20 MFMAs per RHS evaluation forward and 76 per
// evaluation in the adjoint, one after the other on ONE matrix pipe, while 574 of the 1 024 SIMDs have nothing to do
// (450 groups at B=36, S=200): the launch lasts as long as that serial stream.  NeuralStates and NeuralPrecisions
// (reference vihds/ode.py:119-146, vihds/precisions.py:63-87) only meet at their inputs and outputs:
//   * the precision network reads the states (and t), the state network never reads a precision;
//   * in the adjoint the precision network's input adjoint (W1p^T gp, two numbers per lane) is added to the states' one.
// So a 16-trajectory group is a workgroup of cooperating wavefronts, each on its own SIMD:
//   wave A  states:      first + second layer of NeuralStates, its transposed layers in the adjoint; owns y_a, y_b, lambda_a,b
//   wave B  precisions:  the same for NeuralPrecisions; owns v, lambda_v, the log-likelihood (forward)
//   wave H1 / H2 (adjoint only): the Gram tiles of the state / precision network's weight gradients (32 MFMAs per
//           evaluation between them) from the tiles A and B leave in LDS
// Hand-overs go through LDS with workgroup barriers (s_barrier behind lgkmcnt(0) only; global prefetches stay in flight):
//   "in"  barrier: A has published the inputs (y_a, y_b | t) of an evaluation at a stage point; B reads them.  (At grid
//         points of the adjoint both wavefronts load the stored trajectory themselves: no hand-over.)
//   "out" barrier (adjoint): A and B have left their tiles and B its input adjoint; A adds it, H1 / H2 consume the tiles.
// Everything that is published is double-buffered by the index of the hand-over, so that a buffer is rewritten only
// behind the next barrier of the same kind, which its readers reach after they have read it.
// MFMA work per group and step (midpoint): A 63, B 45, H1 32, H2 32 instead of 172 on one pipe.
//
// Arithmetic: the same MFMAs on the same operands as the one-wavefront kernels; the only regrouping is that the input
// adjoint is (W1s^T gs) + (W1p^T gp) with the two products accumulated separately instead of in one chain.
#pragma once
#include "vihds_blackbox_mfma.hpp"

namespace vihds {

#ifdef VIHDS_BB_STAMPS
// profiling build (tests/probe/bb_stamps.py): the wavefronts of the first blocks write the 100 MHz wall clock at
// successive points of ONE step of the adjoint's time loop (k = T / 2), in order of execution: [block][role][32]
static __device__ unsigned long long* vihds_bb_stamp_buf = nullptr;
#define VIHDS_BB_STOP                                                                                      \
  if (vihds_bb_stamp_buf && stamp_on && lane == 0 && blockIdx.x < 64 && stamp_i < 32)                      \
    vihds_bb_stamp_buf[((size_t)blockIdx.x * 4 + role) * 32 + stamp_i++] = wall_clock64();
#else
#define VIHDS_BB_STOP
#endif

template <class KT>
struct BbSplitT {
  using K = KT;
  using BB = typename K::BB;
  using Weights = typename K::Weights;
  using WeightsT = typename K::WeightsT;
  static constexpr int MT = K::MT;
  // LDS of the adjoint (floats): tiles [2][GT_WAVE] | inputs [2][64][2] | input adjoints [2][64][2] | gc [64][4]
  static constexpr int O_IN = 2 * K::GT_WAVE, O_DY = O_IN + 256, O_GC = O_DY + 256, LDS_BWD = O_GC + 256;
  static constexpr int LDS_FWD = 256;  // inputs [2][64][2]

  __device__ __forceinline__ static void sync() { K::pair_sync(); }

  // hand-overs per step: stage points whose inputs A publishes, evaluations with an adjoint
  __host__ __device__ static constexpr int n_in(int solver) {
    return solver == VIHDS_SOLVER_EULER ? 0 : (solver == VIHDS_SOLVER_RK4 ? 3 : 1);
  }
  __host__ __device__ static constexpr int n_vjp(int solver) {
    return solver == VIHDS_SOLVER_EULER ? 1 : (solver == VIHDS_SOLVER_RK4 ? 4 : 2);
  }

  // ---- one network: first layer (two tiles, two K-steps) -> ReLU -> second layer -> pre-sigmoid outputs ---------------
  template <int NET>
  __device__ __forceinline__ static f32x4 net_eval(float b0, float b1, const Weights& W, const f32x4 hc[2][MT],
                                                   f32x4 h[NET == 0 ? K::MS : K::MP]) {
    constexpr int M = NET == 0 ? K::MS : K::MP, KN = NET == 0 ? K::KS : K::KP;
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (NET == 0) h[m] = K::mfma(W.w1s[m < K::MS ? m : 0][1], b1, K::mfma(W.w1s[m < K::MS ? m : 0][0], b0, hc[0][m]));
      else h[m] = K::mfma(W.w1p[m < K::MP ? m : 0][1], b1, K::mfma(W.w1p[m < K::MP ? m : 0][0], b0, hc[1][m]));
    }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[m][r] = __builtin_amdgcn_fmed3f(h[m][r], 0.f, __builtin_inff());  // ReLU as ONE v_med3 (fmaxf adds a canonicalising v_max)
    f32x4 z = NET == 0 ? W.b2s : W.b2p;
#pragma unroll
    for (int s = 0; s < KN; ++s) z = K::mfma(NET == 0 ? W.w2s[s < K::KS ? s : 0] : W.w2p[s < K::KP ? s : 0], h[K::step_m(s)][K::step_r(s)], z);
    return z;
  }
  // transposed layers: second-layer adjoint dz -> hidden pre-activation adjoints g (masked by the ReLU, added to delta)
  // -> input adjoint (rows 4q'+0 = state q', 4q'+1 = latent state 4+q')
  template <int NET>
  __device__ __forceinline__ static f32x4 net_vjp(const f32x4& dz, const f32x4* h, const WeightsT& WT, f32x4* g, f32x4* delta) {
    constexpr int M = NET == 0 ? K::MS : K::MP, KN = NET == 0 ? K::KS : K::KP, NR = NET == 0 ? 4 : 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < M; ++m) {
      f32x4 acc = zero;
#pragma unroll
      for (int r = 0; r < NR; ++r) acc = K::mfma(NET == 0 ? WT.w2sT[m < K::MS ? m : 0][r] : WT.w2pT[m < K::MP ? m : 0][r < 2 ? r : 0], dz[r], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g[m][r] = h[m][r] > 0.f ? acc[r] : 0.f;
        delta[m][r] += g[m][r];
      }
    }
    f32x4 dy = zero;
#pragma unroll
    for (int s = 0; s < KN; ++s) dy = K::mfma(NET == 0 ? WT.w1sT[s < K::KS ? s : 0] : WT.w1pT[s < K::KP ? s : 0], g[K::step_m(s)][K::step_r(s)], dy);
    return dy;
  }

  struct SA {  // wave A's share of a trajectory: state q, latent state 4+q (q < L)
    float a, b;
  };
  __device__ __forceinline__ static float in1(const SA& y, float t, int q) { return q < K::L ? y.b : (q == K::L ? t : 0.f); }
  __device__ __forceinline__ static SA axpyA(const SA& y, float h, const SA& k) { return {y.a + h * k.a, y.b + h * k.b}; }

  // ======================================================================================================================
  // forward
  // ======================================================================================================================
  struct FwdA {
    const Weights& W;
    const f32x4 (*hc)[MT];
    float* pub;  // [2][64][2]
    int lane, q, e;
    // publish the inputs of the next evaluation, hand over, evaluate the state network there
    __device__ __forceinline__ SA eval(float t, const SA& y) {
      const float b0 = y.a, b1 = in1(y, t, q);
      *reinterpret_cast<float2*>(pub + ((e & 1) * 64 + lane) * 2) = make_float2(b0, b1);
      ++e;
      sync();
      f32x4 h[K::MS];
      const f32x4 z = net_eval<0>(b0, b1, W, hc, h);
      SA d;
      d.a = bb_sigmoid(z[0]) - bb_sigmoid(z[1]) * y.a;
      d.b = q < K::L ? bb_sigmoid(z[2]) - bb_sigmoid(z[3]) * y.b : 0.f;
      return d;
    }
  };
  struct FwdB {
    const Weights& W;
    const f32x4 (*hc)[MT];
    const float* pub;
    int lane, e;
    float b0, b1;  // the inputs of the last hand-over
    __device__ __forceinline__ void take() {
      sync();
      const float2 in = *reinterpret_cast<const float2*>(pub + ((e & 1) * 64 + lane) * 2);
      ++e;
      b0 = in.x; b1 = in.y;
    }
    __device__ __forceinline__ float rate(float v) {  // dv/dt at the inputs last taken
      f32x4 g[K::MP];
      const f32x4 zp = net_eval<1>(b0, b1, W, hc, g);
      return bb_sigmoid(zp[0]) - bb_sigmoid(zp[1]) * v;
    }
  };

  template <int SOLVER>
  __device__ __forceinline__ static SA stepA(FwdA& F, float t0, float t1, float h0, const SA& y) {
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
      const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
      const SA k1 = F.eval(t0, y);
      const SA k2 = F.eval(t1, axpyA(y, h, k1));
      const float hh = 0.5f * h;
      return {y.a + hh * (k1.a + k2.a), y.b + hh * (k1.b + k2.b)};
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      return axpyA(y, t1 - t0, F.eval(t0, y));
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      const float dt = t1 - t0;
      const SA k1 = F.eval(t0, y);
      return axpyA(y, dt, F.eval(t0 + dt * 0.5f, axpyA(y, dt * 0.5f, k1)));
    } else {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
      const SA k1 = F.eval(t0, y);
      const SA k2 = F.eval(t0 + d3, axpyA(y, d3, k1));
      const SA y3 = {y.a + (dt * k2.a - d3 * k1.a), y.b + (dt * k2.b - d3 * k1.b)};
      const SA k3 = F.eval(t0 + 2.f * d3, y3);
      const SA y4 = {y.a + dt * (k1.a - k2.a + k3.a), y.b + dt * (k1.b - k2.b + k3.b)};
      const SA k4 = F.eval(t0 + dt, y4);
      return {y.a + (k1.a + 3.f * k2.a + 3.f * k3.a + k4.a) * d8, y.b + (k1.b + 3.f * k2.b + 3.f * k3.b + k4.b) * d8};
    }
  }
  // the precision state over one step; on entry the inputs of the step's first evaluation (the grid point) are taken
  template <int SOLVER>
  __device__ __forceinline__ static float stepB(FwdB& F, float t0, float t1, float h0, float v) {
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
      const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
      const float k1 = F.rate(v);
      F.take();
      const float k2 = F.rate(v + h * k1);
      return v + 0.5f * h * (k1 + k2);
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      return v + (t1 - t0) * F.rate(v);
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      const float dt = t1 - t0;
      const float k1 = F.rate(v);
      F.take();
      return v + dt * F.rate(v + dt * 0.5f * k1);
    } else {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
      const float k1 = F.rate(v);
      F.take();
      const float k2 = F.rate(v + d3 * k1);
      F.take();
      const float k3 = F.rate(v + (dt * k2 - d3 * k1));
      F.take();
      const float k4 = F.rate(v + dt * (k1 - k2 + k3));
      return v + (k1 + 3.f * k2 + 3.f * k3 + k4) * d8;
    }
  }
};

// Where the wavefronts land (tests/micro/wave_placement.hip, profiles/r05_wave_placement.log): the dispatcher deals a block's
// wavefronts to the CU's SIMDs in a fixed cyclic order, and the next block of the same CU (block j + 256 at 450 blocks: 194
// of the 256 CUs hold two) starts ONE position behind the first one's start.  With two wavefronts per block that is
// [p, p+1] and [p+1, p+2]: the second group's state wavefront shares a SIMD with the first group's precision wavefront while
// a fourth SIMD idles; with the adjoint's four in the order A, B, H1, H2 it pairs B with the other group's A.  So:
//   * the forward launches FOUR wavefronts per group and uses wavefronts 0 and 2 (A, B); 1 and 3 help with the sampling
//     stage and leave: [p, p+2] and [p+1, p+3], every chain on a SIMD of its own;
//   * the adjoint's wavefronts take their roles in the order A, H1, B, H2, which pairs every chain wavefront of one group
//     with a Gram helper of the other.
#ifndef VIHDS_BB_FWD_WAVES
#define VIHDS_BB_FWD_WAVES 4
#endif
#ifndef VIHDS_BB_BWD_ORDER
#define VIHDS_BB_BWD_ORDER 0x3120  // role of wavefront w = nibble w: A(0), H1(2), B(1), H2(3)
#endif
constexpr int kBbFwdThreads = 64 * VIHDS_BB_FWD_WAVES;

// grid: one block per 16 trajectories, two working wavefronts (see above)
template <class K, int SOLVER, bool THETA>
__device__ __forceinline__ void bb_split_fwd_body(const OdeArgs& a, const ThetaStageArgs* ts, int nb_max) {
  // THETA: the sampling stage and condition_theta first, for the block's sixteen trajectories (vihds_theta_ode_fwd)
  if constexpr (THETA) {
    extern __shared__ float bb_theta_scratch[];
    theta_stage_block<kBbFwdThreads>(a, *ts, blockIdx.x * K::TPW, K::TPW, nb_max, bb_theta_scratch);
  }
  using S = BbSplitT<K>;
  __shared__ float pub[S::LDS_FWD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (VIHDS_BB_FWD_WAVES == 4 && (wave & 1)) return;  // (a finished wavefront no longer counts at the workgroup barrier)
  const int role = VIHDS_BB_FWD_WAVES == 4 ? wave >> 1 : wave;
  const int jj = lane & 15, q = lane >> 4;
  const int i0 = blockIdx.x * K::TPW + jj;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const size_t n = a.n;
  typename K::Weights W;
  K::gather(a, lane, W);
  f32x4 hc[2][K::MT];
  K::hoist(a, lane, i, b, hc);
  const float h0 = a.times[1] - a.times[0];
  if (role == 0) {
    typename S::FwdA F = {W, hc, pub, lane, q, 0};
    typename S::SA y;
    y.a = a.theta[(size_t)a.slot_row[K::NLAT + q] * n + i];  // init_x, init_rfp, init_yfp, init_cfp
    y.b = q < K::L ? a.init_latent : 0.f;
    float tA = a.times[0], tB = a.times[1];
    for (int k = 0; k < a.T; ++k) {
      const float tC = (k + 1 < a.T) ? a.times[k + 1] : tB;
      if (k > 0) {
        y = S::template stepA<SOLVER>(F, tA, tB, h0, y);
        tA = tB;
      }
      tB = tC;
      if (a.traj && live) {
        a.traj[((size_t)k * K::NST + q) * n + i] = y.a;
        if (q < K::L) a.traj[((size_t)k * K::NST + 4 + q) * n + i] = y.b;
      }
      const float x0 = __shfl(y.a, jj, 64);  // OD lives in quarter 0 of the column
      if (a.xpred && live) a.xpred[((size_t)k * 4 + q) * n + i] = q == 0 ? x0 : x0 * y.a;
    }
    // the last grid point for the precision wavefront's log-likelihood (the other grid points travel as the first
    // evaluation of the step that starts there)
    *reinterpret_cast<float2*>(pub + ((F.e & 1) * 64 + lane) * 2) = make_float2(y.a, S::in1(y, tA, q));
    S::sync();
  } else {
    typename S::FwdB F = {W, hc, pub, lane, 0, 0.f, 0.f};
    float v = a.init_prec, lp = 0.f;
    const float* ob = a.obs + ((size_t)b * 4 + q) * a.T;
    float ob_cur = a.logp ? ob[0] : 0.f;
    float tA = a.times[0], tB = a.T > 1 ? a.times[1] : tA;
    for (int k = 0; k < a.T; ++k) {
      const float tC = (k + 2 < a.T) ? a.times[k + 2] : tB;
      const float ob_next = (a.logp && k + 1 < a.T) ? ob[k + 1] : 0.f;
      F.take();  // grid point k: y_a of this lane's state in b0
      if (a.traj && live) a.traj[((size_t)k * K::NST + 4 + K::L + q) * n + i] = v;
      const float x0 = __shfl(F.b0, jj, 64);
      const float xp = q == 0 ? x0 : x0 * F.b0;
      const float e = xp - ob_cur;
      lp += -0.5f * (LOG2PI_F - logf(v) + v * e * e);
      ob_cur = ob_next;
      if (k + 1 < a.T) v = S::template stepB<SOLVER>(F, tA, tB, h0, v);
      tA = tB;
      tB = tC;
    }
    if (a.logp && live) a.logp[(size_t)q * n + i] = lp;
  }
}
template <class K, int SOLVER>
__global__ void __launch_bounds__(kBbFwdThreads) bb_split_fwd_kernel(OdeArgs a) {
  bb_split_fwd_body<K, SOLVER, false>(a, nullptr, 0);
}
template <class K, int SOLVER>
__global__ void __launch_bounds__(kBbFwdThreads) bb_split_theta_fwd_kernel(OdeArgs a, int nb_max, ThetaStageArgs t) {
  bb_split_fwd_body<K, SOLVER, true>(a, &t, nb_max);
}

// ======================================================================================================================
// adjoint with the weight gradients on chip
// ======================================================================================================================
// grid: one block of four wavefronts (A, B, H1, H2) per 16 trajectories.  aux: the groups' partial Gram tiles
// [group][8][256] (H1 tiles 0-3, H2 tiles 4-7), then the tail (Delta, bias sums) exactly as bb_mfma_bwd_kernel leaves it.
template <class K, int SOLVER>
__global__ void __launch_bounds__(256) bb_split_bwd_kernel(OdeArgs a) {
  using S = BbSplitT<K>;
  using BB = typename K::BB;
  __shared__ __attribute__((aligned(16))) float lds[S::LDS_BWD];
  const int lane = threadIdx.x & 63, role = (VIHDS_BB_BWD_ORDER >> (4 * (threadIdx.x >> 6))) & 3;
  const int jj = lane & 15, q = lane >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  constexpr int NIN = S::n_in(SOLVER), NVJP = S::n_vjp(SOLVER);
  const int n_steps = a.T - 1;
#ifdef VIHDS_BB_STAMPS
  bool stamp_on = false;
  int stamp_i = 0;
#endif

  if (role >= 2) {
    // ---- Gram helpers: H1 (role 2) the state network's tiles 0-3, H2 (role 3) the precision network's 4-7 --------------
    const bool st = role == 2;
    constexpr int MH = K::MS > K::MP ? K::MS : K::MP;
    const int M = st ? K::MS : K::MP;  // tiles of this helper's network: G[m] = dz x h[m], G[MH + m] = g[m] x inputs
    f32x4 G[2 * MH];
#pragma unroll
    for (int tq = 0; tq < 2 * MH; ++tq) G[tq] = zero;
    int e = 0;
    for (int k = 0; k < n_steps; ++k) {
#ifdef VIHDS_BB_STAMPS
      stamp_on = k == n_steps / 2;
#endif
      VIHDS_BB_STOP
#pragma unroll
      for (int p = 0; p < NIN; ++p) S::sync();
#pragma unroll
      for (int p = 0; p < NVJP; ++p) {
        VIHDS_BB_STOP
        S::sync();
        VIHDS_BB_STOP
        const float* buf = lds + (e & 1) * K::GT_WAVE;
        ++e;
        const f32x4 X2 = K::get_rows(buf + (st ? K::T_DZ : K::T_DZP) * K::GT_TILE, lane);
        const f32x4 Yin = K::get_rows(buf + K::T_IN * K::GT_TILE, lane);
#pragma unroll
        for (int m = 0; m < MH; ++m) {
          if (m < M) {
            const f32x4 Yh = K::get_rows(buf + ((st ? K::T_H : K::T_G) + m) * K::GT_TILE, lane);
            K::gram_acc(G[m], X2, Yh);
            const f32x4 Xg = K::get_rows(buf + ((st ? K::T_GS : K::T_GP) + m) * K::GT_TILE, lane);
            K::gram_acc(G[MH + m], Xg, Yin);
          }
        }
        VIHDS_BB_STOP
      }
    }
    S::sync();  // (the epilogue's hand-over between A and B)
    float* gp = a.aux + ((size_t)blockIdx.x * K::NG + (st ? 0 : 2 * K::MS)) * 256;  // tile order: see K::gram_dest
#pragma unroll
    for (int m = 0; m < MH; ++m)
      if (m < M) {
        *reinterpret_cast<f32x4*>(gp + m * 256 + lane * 4) = G[m];
        *reinterpret_cast<f32x4*>(gp + (M + m) * 256 + lane * 4) = G[MH + m];
      }
    return;
  }

  const int i0 = blockIdx.x * K::TPW + jj;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const size_t n = a.n;
  const float lm = live ? 1.f : 0.f;  // (tail lanes shadow the last trajectory: their adjoint rows count as zero)
  typename K::Weights W;
  typename K::WeightsT WT;
  K::gather(a, lane, W);
  K::gather_t(a, lane, WT);
  f32x4 hc[2][K::MT];
  K::hoist(a, lane, i, b, hc);
  f32x4 delta[K::MT];
#pragma unroll
  for (int m = 0; m < K::MT; ++m) delta[m] = zero;
  const float w_iw = a.iw_logp ? iw_wave_weight(a, i, b) : 0.f;
  const float glp = ode_logp_grad(a, w_iw, i, q);
  const float* ob = a.obs + ((size_t)b * 4 + q) * a.T;
  const float h0 = a.times[1] - a.times[0];
  float* pub_in = lds + S::O_IN;
  float* pub_dy = lds + S::O_DY;
  float* pub_gc = lds + S::O_GC;
  int e_in = 0, e_vjp = 0;
  struct Y3 { float a, b, v; };
  auto load_state = [&](int k) {
    Y3 s;
    s.a = a.traj_in[((size_t)k * K::NST + q) * n + i];
    s.b = q < K::L ? a.traj_in[((size_t)k * K::NST + 4 + q) * n + i] : 0.f;
    s.v = a.traj_in[((size_t)k * K::NST + 4 + K::L + q) * n + i];
    return s;
  };
  Y3 ynext = load_state(a.T - 1);
  float ob_next = ob[a.T - 1];
  float tHi = a.times[a.T - 1], tLo = tHi;
  float* dd = a.aux + (size_t)gridDim.x * K::NG * 256;  // Delta [HS+HP][n] behind the Gram partial sums
  float* bbp = dd + (size_t)BB::NP * n;              // output-bias adjoint sums

  if (role == 0) {
    // ================================ wave A: NeuralStates ============================================================
    using SA = typename S::SA;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    SA lam = {0.f, 0.f};
    auto publish = [&](const SA& y, float t) {
      *reinterpret_cast<float2*>(pub_in + ((e_in & 1) * 64 + lane) * 2) = make_float2(y.a, S::in1(y, t, q));
      ++e_in;
      S::sync();
    };
    // One evaluation of the state network: the hidden tiles and the four sigmoids.  An evaluation point that is visited
    // twice in a step (the grid point by every scheme but Euler; rk4's stage points) is evaluated ONCE: the adjoint sweep
    // reuses what the forward pass of the step left (12 registers per point) instead of running the network again.
    struct ActA { f32x4 h[K::MS]; float sa, sd, sa2, sd2; };
    auto act = [&](float t, const SA& y) {
      ActA A;
      const f32x4 z = S::template net_eval<0>(y.a, S::in1(y, t, q), W, hc, A.h);
      A.sa = bb_sigmoid(z[0]); A.sd = bb_sigmoid(z[1]); A.sa2 = bb_sigmoid(z[2]); A.sd2 = bb_sigmoid(z[3]);
      return A;
    };
    auto rate = [&](const ActA& A, const SA& y) {
      SA d;
      d.a = A.sa - A.sd * y.a;
      d.b = q < K::L ? A.sa2 - A.sd2 * y.b : 0.f;
      return d;
    };
    auto eval_vjp = [&](float t, const SA& y, const SA& v, const ActA& A) {
      f32x4 gs[K::MS];
      SA yb;
      yb.a = -v.a * A.sd;
      yb.b = q < K::L ? -v.b * A.sd2 : 0.f;
      f32x4 dz;
      dz[0] = v.a * A.sa * (1.f - A.sa);
      dz[1] = -v.a * y.a * A.sd * (1.f - A.sd);
      dz[2] = q < K::L ? v.b * A.sa2 * (1.f - A.sa2) : 0.f;
      dz[3] = q < K::L ? -v.b * y.b * A.sd2 * (1.f - A.sd2) : 0.f;
      const f32x4 dy = S::template net_vjp<0>(dz, A.h, WT, gs, delta);
      bs[0] += dz[0]; bs[1] += dz[1]; bs[2] += dz[2]; bs[3] += dz[3];
      float* buf = lds + (e_vjp & 1) * K::GT_WAVE;
      const f32x4 xin = {y.a, q < K::L ? y.b : 0.f, q == 0 ? t : 0.f, 0.f};
      K::put_cols(buf + K::T_DZ * K::GT_TILE, dz * lm, lane);
      K::put_cols(buf + K::T_IN * K::GT_TILE, xin, lane);
#pragma unroll
      for (int m = 0; m < K::MS; ++m) {
        K::put_cols(buf + (K::T_H + m) * K::GT_TILE, A.h[m], lane);
        K::put_cols(buf + (K::T_GS + m) * K::GT_TILE, gs[m] * lm, lane);
      }
      VIHDS_BB_STOP
      S::sync();
      VIHDS_BB_STOP
      const float2 dyp = *reinterpret_cast<const float2*>(pub_dy + ((e_vjp & 1) * 64 + lane) * 2);
      ++e_vjp;
      yb.a += dy[0] + dyp.x;
      if (q < K::L) yb.b += dy[1] + dyp.y;
      return yb;
    };
    auto add = [](SA& x, const SA& w, float s) { x.a += s * w.a; x.b += s * w.b; };
    auto scaled = [](const SA& x, float s) { return SA{s * x.a, s * x.b}; };
    auto step_vjp = [&](float t0, float t1, const SA& y) {
      if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
        const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
        const ActA A0 = act(t0, y);
        const SA ya = S::axpyA(y, h, rate(A0, y));
        publish(ya, t1);
        SA vv = scaled(lam, 0.5f * h);
        const SA w = eval_vjp(t1, ya, vv, act(t1, ya));
        add(lam, w, 1.f);
        add(vv, w, h);
        add(lam, eval_vjp(t0, y, vv, A0), 1.f);
      } else if (SOLVER == VIHDS_SOLVER_EULER) {
        add(lam, eval_vjp(t0, y, scaled(lam, t1 - t0), act(t0, y)), 1.f);
      } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
        const float dt = t1 - t0;
        const ActA A0 = act(t0, y);
        const SA ym = S::axpyA(y, dt * 0.5f, rate(A0, y));
        publish(ym, t0 + dt * 0.5f);
        const SA w = eval_vjp(t0 + dt * 0.5f, ym, scaled(lam, dt), act(t0 + dt * 0.5f, ym));
        add(lam, w, 1.f);
        add(lam, eval_vjp(t0, y, scaled(w, 0.5f * dt), A0), 1.f);
      } else {
        const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
        const ActA A1 = act(t0, y);
        const SA k1 = rate(A1, y);
        const SA y2 = S::axpyA(y, d3, k1);
        publish(y2, t0 + d3);
        const ActA A2 = act(t0 + d3, y2);
        const SA k2 = rate(A2, y2);
        const SA y3 = {y.a + (dt * k2.a - d3 * k1.a), y.b + (dt * k2.b - d3 * k1.b)};
        publish(y3, t0 + 2.f * d3);
        const ActA A3 = act(t0 + 2.f * d3, y3);
        const SA k3 = rate(A3, y3);
        const SA y4 = {y.a + dt * (k1.a - k2.a + k3.a), y.b + dt * (k1.b - k2.b + k3.b)};
        publish(y4, t0 + dt);
        const SA k4b = scaled(lam, d8);
        SA k1b = k4b, k2b = scaled(k4b, 3.f), k3b = scaled(k4b, 3.f);
        SA w = eval_vjp(t0 + dt, y4, k4b, act(t0 + dt, y4));
        add(lam, w, 1.f); add(k1b, w, dt); add(k2b, w, -dt); add(k3b, w, dt);
        w = eval_vjp(t0 + 2.f * d3, y3, k3b, A3);
        add(lam, w, 1.f); add(k1b, w, -d3); add(k2b, w, dt);
        w = eval_vjp(t0 + d3, y2, k2b, A2);
        add(lam, w, 1.f); add(k1b, w, d3);
        add(lam, eval_vjp(t0, y, k1b, A1), 1.f);
      }
    };
    for (int k = a.T - 1; k >= 0; --k) {
      const Y3 y = ynext;
      const float obk = ob_next, tK = tLo;
      if (k > 0) {
        ynext = load_state(k - 1);
        ob_next = ob[k - 1];
        tLo = a.times[k - 1];
      }
#ifdef VIHDS_BB_STAMPS
      stamp_on = k == a.T / 2;
#endif
      VIHDS_BB_STOP
      if (k < a.T - 1) step_vjp(tK, tHi, SA{y.a, y.b});
      VIHDS_BB_STOP
      tHi = tK;
      // injection at time k: signal q = OD (q = 0) or OD * state q
      const float x0 = __shfl(y.a, jj, 64);
      const float xp = q == 0 ? x0 : x0 * y.a;
      const float e = xp - obk;
      float xpb = -glp * y.v * e;
      if (a.g_xpred) xpb += a.g_xpred[((size_t)k * 4 + q) * n + i];
      float to_od = q == 0 ? xpb : xpb * y.a;
      to_od += __shfl_xor(to_od, 16, 64);
      to_od += __shfl_xor(to_od, 32, 64);
      if (q == 0) lam.a += to_od;
      else lam.a += xpb * x0;
      if (a.g_traj) {
        lam.a += a.g_traj[((size_t)k * K::NST + q) * n + i];
        if (q < K::L) lam.b += a.g_traj[((size_t)k * K::NST + 4 + q) * n + i];
      }
    }
    // d loss / d latent theta through the hoisted inputs: (Wc^T)[const x slot] . Delta[slot x traj]; the precision
    // network's share comes from wave B
    const typename BB::Off o = BB::offsets(a.n_const);
    const float* w = a.weights;
    f32x4 gc = zero;  // rows = constants 0..15 (only the 12 latents are theta)
    _Pragma("unroll") for (int s = 0; s < K::KS; ++s) {
      const int u = K::unit_of(16 * K::step_m(s) + 4 * q + K::step_r(s), K::HS);
      const float av = (u >= 0 && jj < K::NLAT) ? w[o.wh + u * o.nin_s + K::NX + jj] : 0.f;
      gc = K::mfma(av, delta[K::step_m(s)][K::step_r(s)], gc);
    }
    S::sync();
    const f32x4 gcp = *reinterpret_cast<const f32x4*>(pub_gc + lane * 4);
    if (live) {
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {
        const int c = 4 * q + r;
        if (c < K::NLAT) a.g_theta[(size_t)a.slot_row[c] * n + i] = gc[r] + gcp[r];
      }
      a.g_theta[(size_t)a.slot_row[K::NLAT + q] * n + i] = lam.a;  // init_x .. init_cfp
      _Pragma("unroll") for (int m = 0; m < K::MS; ++m)
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {
          const int us = K::unit_of(16 * m + 4 * q + r, K::HS);
          if (us >= 0) dd[(size_t)us * n + i] = delta[m][r];
        }
      // output-bias adjoint sums, VALU-kernel order: prod states (6), degr states (6), prod prec (4), degr prec (4)
      bbp[(size_t)q * n + i] = bs[0];
      bbp[(size_t)(K::NX + q) * n + i] = bs[1];
      if (q < K::L) { bbp[(size_t)(4 + q) * n + i] = bs[2]; bbp[(size_t)(K::NX + 4 + q) * n + i] = bs[3]; }
    }
  } else {
    // ================================ wave B: NeuralPrecisions ========================================================
    float bs[2] = {0.f, 0.f};
    float lam = 0.f;
    struct In { float b0, b1; };
    auto take = [&]() {
      S::sync();
      const float2 in = *reinterpret_cast<const float2*>(pub_in + ((e_in & 1) * 64 + lane) * 2);
      ++e_in;
      return In{in.x, in.y};
    };
    struct ActB { f32x4 g[K::MP]; float pa, pd; };  // (see ActA)
    auto act = [&](const In& in) {
      ActB A;
      const f32x4 zp = S::template net_eval<1>(in.b0, in.b1, W, hc, A.g);
      A.pa = bb_sigmoid(zp[0]); A.pd = bb_sigmoid(zp[1]);
      return A;
    };
    auto rate = [&](const ActB& A, float v) { return A.pa - A.pd * v; };
    auto eval_vjp = [&](float yv, float vv, const ActB& A) {
      f32x4 gp[K::MP];
      const float ybv = -vv * A.pd;
      f32x4 dzp = zero;
      dzp[0] = vv * A.pa * (1.f - A.pa);
      dzp[1] = -vv * yv * A.pd * (1.f - A.pd);
      const f32x4 dy = S::template net_vjp<1>(dzp, A.g, WT, gp, delta);
      bs[0] += dzp[0]; bs[1] += dzp[1];
      float* buf = lds + (e_vjp & 1) * K::GT_WAVE;
      K::put_cols(buf + K::T_DZP * K::GT_TILE, dzp * lm, lane);
#pragma unroll
      for (int m = 0; m < K::MP; ++m) {
        K::put_cols(buf + (K::T_G + m) * K::GT_TILE, A.g[m], lane);
        K::put_cols(buf + (K::T_GP + m) * K::GT_TILE, gp[m] * lm, lane);
      }
      *reinterpret_cast<float2*>(pub_dy + ((e_vjp & 1) * 64 + lane) * 2) = make_float2(dy[0], dy[1]);
      ++e_vjp;
      VIHDS_BB_STOP
      S::sync();
      VIHDS_BB_STOP
      return ybv;
    };
    auto step_vjp = [&](float t0, float t1, const Y3& y) {
      const In in0 = {y.a, q < K::L ? y.b : (q == K::L ? t0 : 0.f)};
      if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
        const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
        const ActB A0 = act(in0);
        const float yav = y.v + h * rate(A0, y.v);
        const ActB Aa = act(take());
        float vv = lam * (0.5f * h);
        const float w = eval_vjp(yav, vv, Aa);
        lam += w;
        vv += h * w;
        lam += eval_vjp(y.v, vv, A0);
      } else if (SOLVER == VIHDS_SOLVER_EULER) {
        lam += eval_vjp(y.v, lam * (t1 - t0), act(in0));
      } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
        const float dt = t1 - t0;
        const ActB A0 = act(in0);
        const float ymv = y.v + dt * 0.5f * rate(A0, y.v);
        const ActB Am = act(take());
        const float w = eval_vjp(ymv, lam * dt, Am);
        lam += w;
        lam += eval_vjp(y.v, w * (0.5f * dt), A0);
      } else {
        const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
        const ActB A1 = act(in0);
        const float k1 = rate(A1, y.v);
        const float y2 = y.v + d3 * k1;
        const ActB A2 = act(take());
        const float k2 = rate(A2, y2);
        const float y3 = y.v + (dt * k2 - d3 * k1);
        const ActB A3 = act(take());
        const float k3 = rate(A3, y3);
        const float y4 = y.v + dt * (k1 - k2 + k3);
        const ActB A4 = act(take());
        const float k4b = lam * d8;
        float k1b = k4b, k2b = 3.f * k4b, k3b = 3.f * k4b;
        float w = eval_vjp(y4, k4b, A4);
        lam += w; k1b += dt * w; k2b -= dt * w; k3b += dt * w;
        w = eval_vjp(y3, k3b, A3);
        lam += w; k1b -= d3 * w; k2b += dt * w;
        w = eval_vjp(y2, k2b, A2);
        lam += w; k1b += d3 * w;
        lam += eval_vjp(y.v, k1b, A1);
      }
    };
    for (int k = a.T - 1; k >= 0; --k) {
      const Y3 y = ynext;
      const float obk = ob_next, tK = tLo;
      if (k > 0) {
        ynext = load_state(k - 1);
        ob_next = ob[k - 1];
        tLo = a.times[k - 1];
      }
#ifdef VIHDS_BB_STAMPS
      stamp_on = k == a.T / 2;
#endif
      VIHDS_BB_STOP
      if (k < a.T - 1) step_vjp(tK, tHi, y);
      VIHDS_BB_STOP
      tHi = tK;
      // injection at time k: precision q is an ODE state
      const float x0 = __shfl(y.a, jj, 64);
      const float xp = q == 0 ? x0 : x0 * y.a;
      const float e = xp - obk;
      lam += glp * (0.5f / y.v - 0.5f * e * e);
      if (a.g_traj) lam += a.g_traj[((size_t)k * K::NST + 4 + K::L + q) * n + i];
    }
    const typename BB::Off o = BB::offsets(a.n_const);
    const float* w = a.weights;
    f32x4 gc = zero;
    _Pragma("unroll") for (int s = 0; s < K::KP; ++s) {
      const int u = K::unit_of(16 * K::step_m(s) + 4 * q + K::step_r(s), K::HP);
      const float av = (u >= 0 && jj < K::NLAT) ? w[o.vh + u * o.nin_p + 1 + K::NX + jj] : 0.f;
      gc = K::mfma(av, delta[K::step_m(s)][K::step_r(s)], gc);
    }
    *reinterpret_cast<f32x4*>(pub_gc + lane * 4) = gc;
    S::sync();
    if (live) {
      _Pragma("unroll") for (int m = 0; m < K::MP; ++m)
        _Pragma("unroll") for (int r = 0; r < 4; ++r) {
          const int up = K::unit_of(16 * m + 4 * q + r, K::HP);
          if (up >= 0) dd[(size_t)(K::HS + up) * n + i] = delta[m][r];
        }
      bbp[(size_t)(2 * K::NX + q) * n + i] = bs[0];
      bbp[(size_t)(2 * K::NX + 4 + q) * n + i] = bs[1];
    }
  }
}

// one fixed-grid scheme (a side library compiles one scheme per object: csrc/sized/)
template <class K, int SV>
inline int launch_bb_split_solver(bool backward, const OdeArgs& a, hipStream_t st) {
  const dim3 grid(K::gram_groups(a.n));
  if (backward) hipLaunchKernelGGL((bb_split_bwd_kernel<K, SV>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((bb_split_fwd_kernel<K, SV>), grid, dim3(kBbFwdThreads), 0, st, a);
  return VIHDS_OK;
}
// ... with the direction a compile-time choice: a translation unit that only launches forwards holds no adjoint kernel (the
// built-in ICML sizes compile the two directions with different flags: csrc/Makefile, -fno-slp-vectorize)
template <class K, bool BACKWARD>
inline int launch_bb_split_dir(int solver, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts = nullptr) {
  const dim3 grid(K::gram_groups(a.n));
  if (ts) {  // forward with the sampling stage and condition_theta in front (vihds_theta_ode_fwd)
    if constexpr (BACKWARD) return VIHDS_E_UNSUPPORTED;
    else {
      const int nb_max = min(a.B, (K::TPW - 1) / a.S + 2);
      const size_t lds = sizeof(float) * theta_stage_lds_floats(nb_max, ts->P, K::TPW);
      if (lds > 48 * 1024) return VIHDS_E_UNSUPPORTED;
#define VIHDS_BB_TH(SV)                                                                                                  \
  case SV: hipLaunchKernelGGL((bb_split_theta_fwd_kernel<K, SV>), grid, dim3(kBbFwdThreads), lds, st, a, nb_max, *ts); return VIHDS_OK;
      switch (solver) {
        VIHDS_BB_TH(VIHDS_SOLVER_MODEULER)
        VIHDS_BB_TH(VIHDS_SOLVER_MODEULERWHILE)
        VIHDS_BB_TH(VIHDS_SOLVER_EULER)
        VIHDS_BB_TH(VIHDS_SOLVER_MIDPOINT)
        VIHDS_BB_TH(VIHDS_SOLVER_RK4)
      }
#undef VIHDS_BB_TH
      return VIHDS_E_BADARG;
    }
  }
#define VIHDS_BB_DIR(SV)                                                                                   \
  case SV:                                                                                                 \
    if constexpr (BACKWARD) hipLaunchKernelGGL((bb_split_bwd_kernel<K, SV>), grid, dim3(256), 0, st, a);   \
    else hipLaunchKernelGGL((bb_split_fwd_kernel<K, SV>), grid, dim3(kBbFwdThreads), 0, st, a);                     \
    return VIHDS_OK;
  switch (solver) {
    VIHDS_BB_DIR(VIHDS_SOLVER_MODEULER)
    VIHDS_BB_DIR(VIHDS_SOLVER_MODEULERWHILE)
    VIHDS_BB_DIR(VIHDS_SOLVER_EULER)
    VIHDS_BB_DIR(VIHDS_SOLVER_MIDPOINT)
    VIHDS_BB_DIR(VIHDS_SOLVER_RK4)
  }
#undef VIHDS_BB_DIR
  return VIHDS_E_BADARG;
}
template <class K>
inline int launch_bb_split(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  switch (solver) {
    case VIHDS_SOLVER_MODEULER: return launch_bb_split_solver<K, VIHDS_SOLVER_MODEULER>(backward, a, st);
    case VIHDS_SOLVER_MODEULERWHILE: return launch_bb_split_solver<K, VIHDS_SOLVER_MODEULERWHILE>(backward, a, st);
    case VIHDS_SOLVER_EULER: return launch_bb_split_solver<K, VIHDS_SOLVER_EULER>(backward, a, st);
    case VIHDS_SOLVER_MIDPOINT: return launch_bb_split_solver<K, VIHDS_SOLVER_MIDPOINT>(backward, a, st);
    case VIHDS_SOLVER_RK4: return launch_bb_split_solver<K, VIHDS_SOLVER_RK4>(backward, a, st);
  }
  return VIHDS_E_BADARG;
}

}  // namespace vihds
