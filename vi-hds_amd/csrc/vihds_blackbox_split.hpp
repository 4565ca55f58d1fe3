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
//   forward, "in" barrier: A has published the inputs (y_a, y_b | t) of an evaluation (or of a grid point's log-likelihood)
//         in a double-buffered entry; B reads them (and, from quarter 0's words of the same entry, the column's OD).
//   adjoint, "out" barrier: A and B have left their tiles and B its input adjoint; A adds it, H1 / H2 consume the tiles.
//         There is no "in" hand-over in the adjoint: B evaluates the state network at the earlier stage points itself
//         (round 5; B then depends on nobody, and A no longer waits for a B that could only start behind it).
// Everything that is published is multi-buffered by the index of the hand-over, so that a buffer is rewritten only
// behind a later barrier, which its readers reach after they have read it.
// MFMA work per group and step (midpoint): A 52, B 47, H1 32, H2 32 instead of 172 on one pipe.
//
// Round 5, what the launches were made of besides MFMAs (profiles/r05_blackbox.md; config 4: forward 103.5 -> 65 us,
// adjoint 214 -> 181 us):
//   * where the dispatcher puts a CU's second block (VIHDS_BB_FWD_WAVES, VIHDS_BB_BWD_ORDER below);
//   * chains of MFMAs on one accumulator with a VALU instruction scheduled between every two links (chain_fence);
//   * the forward's stores: every wait for a load inside the time loop also waited for the step's trajectory stores
//     (staged inputs, bb_split_fwd_body);
//   * the forward's address arithmetic, ds_bpermute round trips and serialised sigmoid pairs.
// Measured and NOT kept (round 5, commit 0b5146b has the code; profiles/r05_blackbox.md the numbers): hand-overs through
// flag-counted LDS rings that a consumer polls, so that no wavefront waits for a slower phase of another -- forward 70.5
// against 64.8 us with barriers, adjoint 194.9 against 182.4: a polling wavefront takes issue slots from the chain wavefront
// of the CU's other block that shares its SIMD.
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
#ifdef VIHDS_BB_CYCLES_ONLY
#define VIHDS_BB_STOP
#else
#define VIHDS_BB_STOP                                                                                      \
  if (vihds_bb_stamp_buf && stamp_on && lane == 0 && blockIdx.x < 64 && stamp_i < 32)                      \
    vihds_bb_stamp_buf[((size_t)blockIdx.x * 4 + role) * 32 + stamp_i++] = wall_clock64();
#endif
// ... and the forward's state wavefront the shader-cycle counter at points of one step (tests/probe/bb_fwd_cycles.py)
#ifdef VIHDS_BB_CYC_COARSE
#define VIHDS_BB_CYCF(on, slot)
#else
#define VIHDS_BB_CYCF(on, slot) VIHDS_BB_CYC(on, slot)
#endif
#define VIHDS_BB_CYC(on, slot)                                                                 \
  if (vihds_bb_stamp_buf && (on) && (threadIdx.x & 63) == 0 && blockIdx.x < 8)                 \
    vihds_bb_stamp_buf[8192 + blockIdx.x * 64 + (slot)] = __builtin_readcyclecounter();
#else
#define VIHDS_BB_STOP
#define VIHDS_BB_CYC(on, slot)
#define VIHDS_BB_CYCF(on, slot)
#endif

template <class KT>
struct BbSplitT {
  using K = KT;
  using BB = typename K::BB;
  using Weights = typename K::Weights;
  using WeightsT = typename K::WeightsT;
  static constexpr int MT = K::MT;
  // LDS of the adjoint (floats): wave A's tiles [2][NTA] | wave B's tiles [2][NTB] | input adjoints [2][64][2] | gc [64][4].
  // A's entry: dz, inputs, h[m], gs[m]; B's: dzp, inputs (its own copy), g[m], gp[m]
  static constexpr int TA_DZ = 0, TA_IN = 1, TA_H = 2, TA_GS = 2 + K::MS, NTA = 2 + 2 * K::MS;
  static constexpr int TB_DZ = 0, TB_IN = 1, TB_G = 2, TB_GP = 2 + K::MP, NTB = 2 + 2 * K::MP;
  static constexpr int TA_WAVE = NTA * K::GT_TILE, TB_WAVE = NTB * K::GT_TILE;
  static constexpr int O_TB = 2 * TA_WAVE, O_DY = O_TB + 2 * TB_WAVE, O_GC = O_DY + 2 * 128, LDS_BWD = O_GC + 256;
  // LDS of the forward (floats): the published inputs [2][64][2], double-buffered by the hand-over's index
  static constexpr int LDS_FWD = 2 * 128;

  __device__ __forceinline__ static void sync() { K::pair_sync(); }
  // sum over lanes l, l ^ 16, l ^ 32, l ^ 48 as (x + x^16) + the same of l ^ 32, on gfx950's row swaps (two VALU instructions
  // instead of two ds_bpermute round trips; through asm: the builtin's two results came back as one register, ROCm 7.2)
  __device__ __forceinline__ static float quarter_sum(float x) {
    float a = x, b = x;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    const float s = a + b;
    float c = s, d = s;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(d));
    return c + d;
  }
#ifndef VIHDS_BB_CHAIN_FENCE
#define VIHDS_BB_CHAIN_FENCE 1
#endif
  __device__ __forceinline__ static void chain_fence() {
#if VIHDS_BB_CHAIN_FENCE
    __builtin_amdgcn_sched_barrier(0);
#endif
  }

  // hand-overs per step of the adjoint: evaluations with an adjoint
  __host__ __device__ static constexpr int n_vjp(int solver) {
    return solver == VIHDS_SOLVER_EULER ? 1 : (solver == VIHDS_SOLVER_RK4 ? 4 : 2);
  }

  // ---- one network: first layer (two tiles, two K-steps) -> ReLU -> second layer -> pre-sigmoid outputs ---------------
  template <int NET>
  __device__ __forceinline__ static f32x4 net_eval(float b0, float b1, const Weights& W, const f32x4 hc[2][MT],
                                                   f32x4 h[NET == 0 ? K::MS : K::MP]) {
    constexpr int M = NET == 0 ? K::MS : K::MP, KN = NET == 0 ? K::KS : K::KP;
    // VIHDS_BB_CHAIN_FENCE: nothing is scheduled BETWEEN the MFMAs of a chain on one accumulator.  Left alone, LLVM spreads
    // the ReLUs (and the adjoint's masks) between the second layer's K-steps to "hide" them; on gfx950 one extra issue
    // slot between two MFMAs on the same accumulator costs ~43 cycles on top of the 40 of the dependency (a cliff:
    // MI355X_MICROARCH.md), which made every K-step of these chains ~100 cycles instead of 40.
#pragma unroll
    for (int m = 0; m < M; ++m) {
      if (NET == 0) h[m] = K::mfma(W.w1s[m < K::MS ? m : 0][1], b1, K::mfma(W.w1s[m < K::MS ? m : 0][0], b0, hc[0][m]));
      else h[m] = K::mfma(W.w1p[m < K::MP ? m : 0][1], b1, K::mfma(W.w1p[m < K::MP ? m : 0][0], b0, hc[1][m]));
    }
    chain_fence();
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) h[m][r] = __builtin_amdgcn_fmed3f(h[m][r], 0.f, __builtin_inff());  // ReLU as ONE v_med3 (fmaxf adds a canonicalising v_max)
    f32x4 z = NET == 0 ? W.b2s : W.b2p;
    chain_fence();
#pragma unroll
    for (int s = 0; s < KN; ++s) z = K::mfma(NET == 0 ? W.w2s[s < K::KS ? s : 0] : W.w2p[s < K::KP ? s : 0], h[K::step_m(s)][K::step_r(s)], z);
    chain_fence();
    return z;
  }
  // transposed layers: second-layer adjoint dz -> hidden pre-activation adjoints g (masked by the ReLU, added to delta)
  // -> input adjoint (rows 4q'+0 = state q', 4q'+1 = latent state 4+q')
  template <int NET>
  __device__ __forceinline__ static f32x4 net_vjp(const f32x4& dz, const f32x4* h, const WeightsT& WT, f32x4* g, f32x4* delta) {
    constexpr int M = NET == 0 ? K::MS : K::MP, KN = NET == 0 ? K::KS : K::KP, NR = NET == 0 ? 4 : 2;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[M];
    chain_fence();
    // (r outermost: the M chains walk side by side, each MFMA issued in the shadow of the other chains' dependencies)
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
      for (int m = 0; m < M; ++m)
        acc[m] = K::mfma(NET == 0 ? WT.w2sT[m < K::MS ? m : 0][r] : WT.w2pT[m < K::MP ? m : 0][r < 2 ? r : 0], dz[r], r == 0 ? zero : acc[m]);
    chain_fence();
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        g[m][r] = h[m][r] > 0.f ? acc[m][r] : 0.f;
        delta[m][r] += g[m][r];
      }
    f32x4 dy = zero;
    chain_fence();
#pragma unroll
    for (int s = 0; s < KN; ++s) dy = K::mfma(NET == 0 ? WT.w1sT[s < K::KS ? s : 0] : WT.w1pT[s < K::KP ? s : 0], g[K::step_m(s)][K::step_r(s)], dy);
    chain_fence();
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
    bool cyc = false;  // (profiling build: this step is stamped)
    int cyc_i = 0;
    // hand-over e: the inputs of an evaluation (or of a grid point's log-likelihood), then the barrier
    __device__ __forceinline__ void publish(float b0, float b1) {
      *reinterpret_cast<float2*>(pub + ((e & 1) * 64 + lane) * 2) = make_float2(b0, b1);
      ++e;
      sync();
    }
    // publish the inputs of the next evaluation, evaluate the state network there
    __device__ __forceinline__ SA eval(float t, const SA& y) {
      const float b0 = y.a, b1 = in1(y, t, q);
      VIHDS_BB_CYC(cyc, cyc_i++)
      publish(b0, b1);
      VIHDS_BB_CYC(cyc, cyc_i++)
      f32x4 h[K::MS];
      const f32x4 z = net_eval<0>(b0, b1, W, hc, h);
      VIHDS_BB_CYC(cyc && z[0] != 12345.f, cyc_i++)
      // (all four sigmoids in every lane, side by side: a branch on the quarter would run the latent pair's chain BEHIND the
      // species pair's, 200 cycles instead of 110 with one wavefront on the SIMD)
      const float s0 = bb_sigmoid(z[0]), s1 = bb_sigmoid(z[1]), s2 = bb_sigmoid(z[2]), s3 = bb_sigmoid(z[3]);
      SA d;
      d.a = s0 - s1 * y.a;
      d.b = q < K::L ? s2 - s3 * y.b : 0.f;
      VIHDS_BB_CYC(cyc && d.a != 12345.f && d.b != 12345.f, cyc_i++)
      return d;
    }
  };
  struct FwdB {
    const Weights& W;
    const f32x4 (*hc)[MT];
    const float* pub;
    int lane, e;
    float b0, b1;  // the inputs of the last hand-over
    float od = 0.f;  // ... and the first state (OD) of this lane's trajectory at that point (quarter 0's b0)
    __device__ __forceinline__ void take() {
      sync();
      const float* slot = pub + ((e & 1) * 64 + lane) * 2;
      const float2 in = *reinterpret_cast<const float2*>(slot);
      od = pub[((e & 1) * 64 + (lane & 15)) * 2];  // (read beside the entry: no ds_bpermute round trip in the step)
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
// The time grid and the observations of the block's data rows (at most stage_rows of them) are staged in dynamic LDS
// [T] | [stage_rows][4][T] first, so that the time loops hold NO vector-memory load: the counter that a load's s_waitcnt
// waits on counts the trajectory's stores as well (in order), and behind exec-masked stores the compiler can only wait for
// all of them -- every step of both wavefronts used to sit through the round trip of its own stores.  (Unconditionally: a
// run-time choice between staged and global reads keeps the wait at the place where the two paths meet.)
template <class K, int SOLVER, bool THETA>
__device__ __forceinline__ void bb_split_fwd_body(const OdeArgs& a, const ThetaStageArgs* ts, int nb_max, int stage_rows) {
  extern __shared__ float bb_dyn[];  // the sampling stage's scratch, then the staged inputs
  // THETA: the sampling stage and condition_theta first, for the block's sixteen trajectories (vihds_theta_ode_fwd)
  if constexpr (THETA) {
    theta_stage_block<kBbFwdThreads>(a, *ts, blockIdx.x * K::TPW, K::TPW, nb_max, bb_dyn);
    __syncthreads();
  }
  using S = BbSplitT<K>;
  __shared__ __attribute__((aligned(16))) float pub[S::LDS_FWD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int row0 = (blockIdx.x * K::TPW) / a.S;  // first data row of the block
  {
    const int nrow = min(stage_rows, a.B - row0);
    for (int e = threadIdx.x; e < a.T; e += kBbFwdThreads) bb_dyn[e] = a.times[e];
    const float* src = a.obs + (size_t)row0 * 4 * a.T;
    for (int e = threadIdx.x; e < nrow * 4 * a.T; e += kBbFwdThreads) bb_dyn[a.T + e] = src[e];
  }
  __syncthreads();
  if (VIHDS_BB_FWD_WAVES == 4 && (wave & 1)) return;
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
  auto time_at = [&](int k) { return bb_dyn[k]; };
  if (role == 0) {
    typename S::FwdA F = {W, hc, pub, lane, q, 0};
    typename S::SA y;
    y.a = a.theta[(size_t)a.slot_row[K::NLAT + q] * n + i];  // init_x, init_rfp, init_yfp, init_cfp
    y.b = q < K::L ? a.init_latent : 0.f;
    float tA = a.times[0], tB = a.times[1];
    // the stores' addresses walk with the loop (one 64-bit add per array and step instead of the index arithmetic)
    const size_t step_traj = (size_t)K::NST * n, step_xp = (size_t)4 * n;
    float* p_ya = a.traj ? a.traj + (size_t)q * n + i : nullptr;
    float* p_yb = a.traj ? a.traj + (size_t)(4 + q) * n + i : nullptr;
    float* p_xp = a.xpred ? a.xpred + (size_t)q * n + i : nullptr;
    // (everything loaded so far has arrived: a wait for one of these values INSIDE the loop would wait for the loop's stores too)
    __builtin_amdgcn_s_waitcnt(0);
    for (int k = 0; k < a.T; ++k) {
      const float tC = (k + 1 < a.T) ? time_at(k + 1) : tB;
#ifdef VIHDS_BB_STAMPS
      F.cyc = k == a.T / 2 || k == a.T / 2 + 1;
      if (k == a.T / 2) F.cyc_i = 0;
#endif
      VIHDS_BB_CYC(F.cyc, F.cyc_i++)
      if (k > 0) {
        y = S::template stepA<SOLVER>(F, tA, tB, h0, y);
        tA = tB;
      }
      VIHDS_BB_CYC(F.cyc && y.a != 12345.f, F.cyc_i++)
      tB = tC;
      if (a.traj && live) {
        *p_ya = y.a;
        if (q < K::L) *p_yb = y.b;
      }
      p_ya += step_traj;
      p_yb += step_traj;
      if (a.xpred) {  // (not in a training step: nobody reads x_predict there)
        const float x0 = __shfl(y.a, jj, 64);  // OD lives in quarter 0 of the column
        if (live) *p_xp = q == 0 ? x0 : x0 * y.a;
        p_xp += step_xp;
      }
    }
    // the last grid point for the precision wavefront's log-likelihood (the other grid points travel as the first
    // evaluation of the step that starts there)
    F.publish(y.a, S::in1(y, tA, q));
  } else {
    typename S::FwdB F = {W, hc, pub, lane, 0, 0.f, 0.f};
    float v = a.init_prec, lp = 0.f;
    const float* ob = bb_dyn + a.T + ((size_t)(b - row0) * 4 + q) * a.T;
    float ob_cur = a.logp ? ob[0] : 0.f;
    float tA = a.times[0], tB = a.T > 1 ? a.times[1] : tA;
    float* p_v = a.traj ? a.traj + (size_t)(4 + K::L + q) * n + i : nullptr;
    const size_t step_traj = (size_t)K::NST * n;
    __builtin_amdgcn_s_waitcnt(0);  // (as in wave A)
    for (int k = 0; k < a.T; ++k) {
      const float tC = (k + 2 < a.T) ? time_at(k + 2) : tB;
      const float ob_next = (a.logp && k + 1 < a.T) ? ob[k + 1] : 0.f;
      F.take();  // grid point k: y_a of this lane's state in b0
      if (a.traj && live) *p_v = v;
      p_v += step_traj;
      const float x0 = F.od;  // (read beside the entry: no ds_bpermute round trip in the step)
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
__global__ void __launch_bounds__(kBbFwdThreads) bb_split_fwd_kernel(OdeArgs a, int stage_rows) {
  bb_split_fwd_body<K, SOLVER, false>(a, nullptr, 0, stage_rows);
}
template <class K, int SOLVER>
__global__ void __launch_bounds__(kBbFwdThreads) bb_split_theta_fwd_kernel(OdeArgs a, int nb_max, ThetaStageArgs t, int stage_rows) {
  bb_split_fwd_body<K, SOLVER, true>(a, &t, nb_max, stage_rows);
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
  // Wave B evaluates the state network at the step's earlier stage points ITSELF (the same instructions on the same operands
  // as wave A) instead of waiting for A to publish the stage inputs (round 5): no "in" hand-over is left in the adjoint, B --
  // whose evaluation at a stage point could only start when A's previous one was done -- no longer arrives last at the
  // step's first hand-over (stamps: A waited 0.6 of a step's 2.8 us there), and B depends on nobody.
  // What is left is one-directional -- B -> A (the precision network's input adjoint), A -> H1 and B -> H2 (tiles) -- and goes
  // through LDS at ONE workgroup barrier per evaluation with an adjoint.
  constexpr int NVJP = S::n_vjp(SOLVER);
  const int n_steps = a.T - 1;
#ifdef VIHDS_BB_STAMPS
  bool stamp_on = false;
  int stamp_i = 0, stamp_c = 0;
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
      for (int p = 0; p < NVJP; ++p) {
        VIHDS_BB_STOP
        VIHDS_BB_CYC(stamp_on, 48 + 8 * (role - 2) + stamp_c++)
        S::sync();
        VIHDS_BB_STOP
        const float* buf = st ? lds + (e & 1) * S::TA_WAVE : lds + S::O_TB + (e & 1) * S::TB_WAVE;
        ++e;
        // (tile numbers of the two entries coincide: dz, inputs, second-layer inputs [m], first-layer adjoints [m])
        static_assert(S::TA_DZ == S::TB_DZ && S::TA_IN == S::TB_IN && S::TA_H == S::TB_G, "entry layouts");
        const int t_x = st ? S::TA_GS : S::TB_GP;
        const f32x4 X2 = K::get_rows(buf + S::TA_DZ * K::GT_TILE, lane);
        const f32x4 Yin = K::get_rows(buf + S::TA_IN * K::GT_TILE, lane);
        f32x4 Yh[MH], Xg[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) {
          if (m < M) {
            Yh[m] = K::get_rows(buf + (S::TA_H + m) * K::GT_TILE, lane);
            Xg[m] = K::get_rows(buf + (t_x + m) * K::GT_TILE, lane);
          }
        }
#pragma unroll
        for (int m = 0; m < MH; ++m) {
          if (m < M) {
            K::gram_acc(G[m], X2, Yh[m]);
            K::gram_acc(G[MH + m], Xg[m], Yin);
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
  float* pub_dy = lds + S::O_DY;
  float* pub_gc = lds + S::O_GC;
  int e_vjp = 0;
  // the stored trajectory at a grid point: this lane's state, latent state and precision, and the column's first state (OD,
  // which the observation map multiplies the others with: loaded with the rest one step ahead instead of a ds_bpermute
  // round trip from quarter 0 in every step); the addresses walk with the loop
  struct Y3 { float a, b, v, od; };
  const ptrdiff_t step_traj = (ptrdiff_t)K::NST * (ptrdiff_t)n;
  const float* p_state = a.traj_in + ((size_t)(a.T - 1) * K::NST) * n + i;
  auto load_state = [&]() {  // ... at the grid point p_state stands at; then one step down
    Y3 s;
    s.a = p_state[(size_t)q * n];
    s.b = q < K::L ? p_state[(size_t)(4 + q) * n] : 0.f;
    s.v = p_state[(size_t)(4 + K::L + q) * n];
    s.od = p_state[0];
    p_state -= step_traj;
    return s;
  };
  Y3 ynext = load_state();
  float ob_next = ob[a.T - 1];
  float tHi = a.times[a.T - 1], tLo = tHi;
  float* dd = a.aux + (size_t)gridDim.x * K::NG * 256;  // Delta [HS+HP][n] behind the Gram partial sums
  float* bbp = dd + (size_t)BB::NP * n;              // output-bias adjoint sums

  if (role == 0) {
    // ================================ wave A: NeuralStates ============================================================
    using SA = typename S::SA;
    float bs[4] = {0.f, 0.f, 0.f, 0.f};
    SA lam = {0.f, 0.f};
    // One evaluation of the state network: the hidden tiles and the four sigmoids.  An evaluation point that is visited
    // twice in a step (the grid point by every scheme but Euler; rk4's stage points) is evaluated ONCE: the adjoint sweep
    // reuses what the forward pass of the step left (12 registers per point) instead of running the network again.
    struct ActA { f32x4 h[K::MS]; float sa, sd, sa2, sd2; };
    auto act = [&](float t, const SA& y) {
      ActA A;
      const f32x4 z = S::template net_eval<0>(y.a, S::in1(y, t, q), W, hc, A.h);
      A.sa = bb_sigmoid(z[0]); A.sd = bb_sigmoid(z[1]); A.sa2 = bb_sigmoid(z[2]); A.sd2 = bb_sigmoid(z[3]);
      VIHDS_BB_CYCF(stamp_on && A.sa != 12345.f && A.sd != 12345.f && A.sa2 != 12345.f && A.sd2 != 12345.f, 16 + stamp_c++)
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
      VIHDS_BB_CYCF(stamp_on, 16 + stamp_c++)
      const f32x4 dy = S::template net_vjp<0>(dz, A.h, WT, gs, delta);
      VIHDS_BB_CYCF(stamp_on && dy[0] != 12345.f, 16 + stamp_c++)
      bs[0] += dz[0]; bs[1] += dz[1]; bs[2] += dz[2]; bs[3] += dz[3];
      float* buf = lds + (e_vjp & 1) * S::TA_WAVE;
      const f32x4 xin = {y.a, q < K::L ? y.b : 0.f, q == 0 ? t : 0.f, 0.f};
      K::put_cols(buf + S::TA_DZ * K::GT_TILE, dz * lm, lane);
      K::put_cols(buf + S::TA_IN * K::GT_TILE, xin, lane);
#pragma unroll
      for (int m = 0; m < K::MS; ++m) {
        K::put_cols(buf + (S::TA_H + m) * K::GT_TILE, A.h[m], lane);
        K::put_cols(buf + (S::TA_GS + m) * K::GT_TILE, gs[m] * lm, lane);
      }
      VIHDS_BB_STOP
      VIHDS_BB_CYC(stamp_on, 16 + stamp_c++)
      S::sync();
      VIHDS_BB_CYCF(stamp_on, 16 + stamp_c++)
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
        const SA w = eval_vjp(t0 + dt * 0.5f, ym, scaled(lam, dt), act(t0 + dt * 0.5f, ym));
        add(lam, w, 1.f);
        add(lam, eval_vjp(t0, y, scaled(w, 0.5f * dt), A0), 1.f);
      } else {
        const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
        const ActA A1 = act(t0, y);
        const SA k1 = rate(A1, y);
        const SA y2 = S::axpyA(y, d3, k1);
        const ActA A2 = act(t0 + d3, y2);
        const SA k2 = rate(A2, y2);
        const SA y3 = {y.a + (dt * k2.a - d3 * k1.a), y.b + (dt * k2.b - d3 * k1.b)};
        const ActA A3 = act(t0 + 2.f * d3, y3);
        const SA k3 = rate(A3, y3);
        const SA y4 = {y.a + dt * (k1.a - k2.a + k3.a), y.b + dt * (k1.b - k2.b + k3.b)};
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
        ynext = load_state();
        ob_next = ob[k - 1];
        tLo = a.times[k - 1];
      }
#ifdef VIHDS_BB_STAMPS
      stamp_on = k == a.T / 2;
      if (k == a.T / 2 - 1) { VIHDS_BB_CYC(true, 16 + stamp_c++) }
#endif
      VIHDS_BB_STOP
      VIHDS_BB_CYC(stamp_on, 16 + stamp_c++)
      if (k < a.T - 1) step_vjp(tK, tHi, SA{y.a, y.b});
      VIHDS_BB_CYC(stamp_on && lam.a != 12345.f, 16 + stamp_c++)
      VIHDS_BB_STOP
      tHi = tK;
      // injection at time k: signal q = OD (q = 0) or OD * state q
      const float x0 = y.od;
      const float xp = q == 0 ? x0 : x0 * y.a;
      const float e = xp - obk;
      float xpb = -glp * y.v * e;
      if (a.g_xpred) xpb += a.g_xpred[((size_t)k * 4 + q) * n + i];
      float to_od = q == 0 ? xpb : xpb * y.a;
      to_od = S::quarter_sum(to_od);  // over the column's four quarters (x + x^16, then + ^32: the order the shuffles had)
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
    // the state network's rate at (t, y), as wave A forms it
    using SA = typename S::SA;
    auto state_rate = [&](float t, const SA& y) {
      f32x4 h[K::MS];
      const f32x4 z = S::template net_eval<0>(y.a, S::in1(y, t, q), W, hc, h);
      SA d;
      d.a = bb_sigmoid(z[0]) - bb_sigmoid(z[1]) * y.a;
      d.b = q < K::L ? bb_sigmoid(z[2]) - bb_sigmoid(z[3]) * y.b : 0.f;
      return d;
    };
    auto in_at = [&](const SA& y, float t) { return In{y.a, S::in1(y, t, q)}; };
    struct ActB { f32x4 g[K::MP]; float pa, pd; };  // (see ActA)
    auto act = [&](const In& in) {
      ActB A;
      const f32x4 zp = S::template net_eval<1>(in.b0, in.b1, W, hc, A.g);
      A.pa = bb_sigmoid(zp[0]); A.pd = bb_sigmoid(zp[1]);
      return A;
    };
    auto rate = [&](const ActB& A, float v) { return A.pa - A.pd * v; };
    auto eval_vjp = [&](const In& in, float t, float yv, float vv, const ActB& A) {
      f32x4 gp[K::MP];
      const float ybv = -vv * A.pd;
      f32x4 dzp = zero;
      dzp[0] = vv * A.pa * (1.f - A.pa);
      dzp[1] = -vv * yv * A.pd * (1.f - A.pd);
      const f32x4 dy = S::template net_vjp<1>(dzp, A.g, WT, gp, delta);
      bs[0] += dzp[0]; bs[1] += dzp[1];
      float* buf = lds + S::O_TB + (e_vjp & 1) * S::TB_WAVE;
      // (the evaluation's inputs as wave A lays them out for the Gram tiles: state, latent state, time)
      const f32x4 xin = {in.b0, q < K::L ? in.b1 : 0.f, q == 0 ? t : 0.f, 0.f};
      K::put_cols(buf + S::TB_DZ * K::GT_TILE, dzp * lm, lane);
      K::put_cols(buf + S::TB_IN * K::GT_TILE, xin, lane);
#pragma unroll
      for (int m = 0; m < K::MP; ++m) {
        K::put_cols(buf + (S::TB_G + m) * K::GT_TILE, A.g[m], lane);
        K::put_cols(buf + (S::TB_GP + m) * K::GT_TILE, gp[m] * lm, lane);
      }
      *reinterpret_cast<float2*>(pub_dy + ((e_vjp & 1) * 64 + lane) * 2) = make_float2(dy[0], dy[1]);
      ++e_vjp;
      VIHDS_BB_STOP
      VIHDS_BB_CYC(stamp_on, 32 + stamp_c++)
      S::sync();
      VIHDS_BB_STOP
      return ybv;
    };
    auto step_vjp = [&](float t0, float t1, const Y3& y) {
      const In in0 = {y.a, q < K::L ? y.b : (q == K::L ? t0 : 0.f)};
      const SA ys = {y.a, y.b};
      if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
        const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
        const ActB A0 = act(in0);
        const float yav = y.v + h * rate(A0, y.v);
        const In ina = in_at(S::axpyA(ys, h, state_rate(t0, ys)), t1);
        const ActB Aa = act(ina);
        float vv = lam * (0.5f * h);
        const float w = eval_vjp(ina, t1, yav, vv, Aa);
        lam += w;
        vv += h * w;
        lam += eval_vjp(in0, t0, y.v, vv, A0);
      } else if (SOLVER == VIHDS_SOLVER_EULER) {
        lam += eval_vjp(in0, t0, y.v, lam * (t1 - t0), act(in0));
      } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
        const float dt = t1 - t0, tm = t0 + dt * 0.5f;
        const ActB A0 = act(in0);
        const float ymv = y.v + dt * 0.5f * rate(A0, y.v);
        const In inm = in_at(S::axpyA(ys, dt * 0.5f, state_rate(t0, ys)), tm);
        const ActB Am = act(inm);
        const float w = eval_vjp(inm, tm, ymv, lam * dt, Am);
        lam += w;
        lam += eval_vjp(in0, t0, y.v, w * (0.5f * dt), A0);
      } else {
        const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
        const ActB A1 = act(in0);
        const float k1 = rate(A1, y.v);
        const float y2 = y.v + d3 * k1;
        // (the state's stage points as wave A's step_vjp forms them)
        const SA s1 = state_rate(t0, ys);
        const SA ys2 = S::axpyA(ys, d3, s1);
        const In in2 = in_at(ys2, t0 + d3);
        const ActB A2 = act(in2);
        const float k2 = rate(A2, y2);
        const float y3 = y.v + (dt * k2 - d3 * k1);
        const SA s2 = state_rate(t0 + d3, ys2);
        const SA ys3 = {ys.a + (dt * s2.a - d3 * s1.a), ys.b + (dt * s2.b - d3 * s1.b)};
        const In in3 = in_at(ys3, t0 + 2.f * d3);
        const ActB A3 = act(in3);
        const float k3 = rate(A3, y3);
        const float y4 = y.v + dt * (k1 - k2 + k3);
        const SA s3 = state_rate(t0 + 2.f * d3, ys3);
        const SA ys4 = {ys.a + dt * (s1.a - s2.a + s3.a), ys.b + dt * (s1.b - s2.b + s3.b)};
        const In in4 = in_at(ys4, t0 + dt);
        const ActB A4 = act(in4);
        const float k4b = lam * d8;
        float k1b = k4b, k2b = 3.f * k4b, k3b = 3.f * k4b;
        float w = eval_vjp(in4, t0 + dt, y4, k4b, A4);
        lam += w; k1b += dt * w; k2b -= dt * w; k3b += dt * w;
        w = eval_vjp(in3, t0 + 2.f * d3, y3, k3b, A3);
        lam += w; k1b -= d3 * w; k2b += dt * w;
        w = eval_vjp(in2, t0 + d3, y2, k2b, A2);
        lam += w; k1b += d3 * w;
        lam += eval_vjp(in0, t0, y.v, k1b, A1);
      }
    };
    for (int k = a.T - 1; k >= 0; --k) {
      const Y3 y = ynext;
      const float obk = ob_next, tK = tLo;
      if (k > 0) {
        ynext = load_state();
        ob_next = ob[k - 1];
        tLo = a.times[k - 1];
      }
#ifdef VIHDS_BB_STAMPS
      stamp_on = k == a.T / 2;
#endif
      VIHDS_BB_STOP
      VIHDS_BB_CYC(stamp_on, 32 + stamp_c++)
      if (k < a.T - 1) step_vjp(tK, tHi, y);
      VIHDS_BB_CYC(stamp_on && lam != 12345.f, 32 + stamp_c++)
      VIHDS_BB_STOP
      tHi = tK;
      // injection at time k: precision q is an ODE state
      const float x0 = y.od;
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

// the forward's staged inputs (bb_split_fwd_body): data rows a block of TPW trajectories can touch, or 0 when they would not
// fit (very long time grids; tiny S: many rows per block)
template <class K>
inline int bb_fwd_stage_rows(const OdeArgs& a, size_t* bytes) {
  const int rows = min(a.B, (K::TPW - 1) / a.S + 2);
  *bytes = sizeof(float) * ((size_t)a.T + (size_t)rows * 4 * a.T);
  if (*bytes > 48 * 1024) { *bytes = 0; return 0; }  // (the caller takes another forward kernel)
  return rows;
}
// one fixed-grid scheme (a side library compiles one scheme per object: csrc/sized/)
template <class K, int SV>
inline int launch_bb_split_solver(bool backward, const OdeArgs& a, hipStream_t st) {
  const dim3 grid(K::gram_groups(a.n));
  if (backward) hipLaunchKernelGGL((bb_split_bwd_kernel<K, SV>), grid, dim3(256), 0, st, a);
  else {
    size_t sb;
    const int rows = bb_fwd_stage_rows<K>(a, &sb);
    if (rows == 0) return VIHDS_E_UNSUPPORTED;
    hipLaunchKernelGGL((bb_split_fwd_kernel<K, SV>), grid, dim3(kBbFwdThreads), sb, st, a, rows);
  }
  return VIHDS_OK;
}
// ... with the direction a compile-time choice: a translation unit that only launches forwards holds no adjoint kernel (the
// built-in ICML sizes compile the two directions with different flags: csrc/Makefile, -fno-slp-vectorize)
template <class K, bool BACKWARD>
inline int launch_bb_split_dir(int solver, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts = nullptr) {
  const dim3 grid(K::gram_groups(a.n));
  size_t stage_bytes = 0;
  const int stage_rows = BACKWARD ? 0 : bb_fwd_stage_rows<K>(a, &stage_bytes);
  if (!BACKWARD && stage_rows == 0) return VIHDS_E_UNSUPPORTED;
  if (ts) {  // forward with the sampling stage and condition_theta in front (vihds_theta_ode_fwd)
    if constexpr (BACKWARD) return VIHDS_E_UNSUPPORTED;
    else {
      const int nb_max = min(a.B, (K::TPW - 1) / a.S + 2);
      size_t lds = sizeof(float) * theta_stage_lds_floats(nb_max, ts->P, K::TPW), sb;
      if (lds > 48 * 1024) return VIHDS_E_UNSUPPORTED;
      const int rows = bb_fwd_stage_rows<K>(a, &sb);
      if (sb > lds) lds = sb;
#define VIHDS_BB_TH(SV)                                                                                                  \
  case SV: hipLaunchKernelGGL((bb_split_theta_fwd_kernel<K, SV>), grid, dim3(kBbFwdThreads), lds, st, a, nb_max, *ts, rows); return VIHDS_OK;
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
    else hipLaunchKernelGGL((bb_split_fwd_kernel<K, SV>), grid, dim3(kBbFwdThreads), stage_bytes, st, a, stage_rows); \
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
