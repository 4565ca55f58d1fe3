// dr_blackbox on the matrix cores: v_mfma_f32_16x16x4_f32 (exact fp32), 16 trajectories per wavefront.
//
// The MLPs are evaluated TRANSPOSED -- rows = units of a layer, columns = trajectories -- so that a layer's output
// tile is already a valid B operand of the next layer's MFMA and activations never leave the register file:
//   C/D layout of 16x16x4:  lane l = (j = l & 15, q = l >> 4) holds rows 4q + r (r = register 0..3) of column j
//   B layout             :  lane l supplies B[k = q][j]
// Feeding register r of an output tile as B means k-slot q carries unit 4q + r; the weight (A) operand of that step
// is gathered once per kernel in the matching order: A[i][k = q] = W[i][unit 4q + r].  The K order of a dot product
// is arbitrary, so no shuffle, no LDS, no transposition is ever needed.
//
// Ownership falls out of the same layout: quarter q of the lanes of column j owns state q (OD, RFP, YFP, CFP),
// precision state q and observed signal q of trajectory j; quarters 0,1 also own the two latent states.  The output
// rows of the last layers are numbered so that the production / degradation pre-activations of a state land in the
// lane that owns it (row 4q+0 = prod_q, 4q+1 = degr_q, 4q+2 = prod_{4+q}, 4q+3 = degr_{4+q}).
//
// Hidden units are renumbered into "slots" so that the 25 (states) / 20 (precisions) units need only 7 / 5 K-steps:
//   slots 0..15 = units 0..15;  tile 1: register 0 -> units 16..19 (q = 0..3), register 1 -> 20..23, register 2, q=0 -> 24
// (in general: full tiles keep their order, the last tile's units fill register 0 of the four quarters first).
//
// Per RHS evaluation: 20 MFMAs forward (8 first-layer, 12 second-layer); the adjoint adds 24 transposed ones.  The
// 21 time-invariant inputs are folded into the first layers' accumulator initial values by 24 MFMAs in the prologue.
// Weight gradients: accumulated on chip as Gram tiles by the cooperating-wavefront kernels (vihds_blackbox_split.hpp,
// the default); the VALU kernels (vihds_blackbox.hpp: kernel_variant 1, the adaptive solvers) write a per-evaluation dump for
// vihds_gram_blocks instead.
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_args.hpp"
#include "vihds_blackbox.hpp"
#include "vihds_models.hpp"
#include "vihds_iwae_inline.hpp"
#include "vihds_theta_stage.hpp"

namespace vihds {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// The size-generic part (round 3): everything the cooperating-wavefront kernels (vihds_blackbox_split.hpp) use, for
// n_latent_species = 2 and up to 64 / 32 hidden units (MS / MP tiles of 16) -- the reference's default
// n_hidden_decoder = 50 (vihds/config.py:71) included -- and up to 16 latent theta inputs.
template <class BBT, int LV, int HSV, int HPV, int NLATV>
struct BbMfmaT {
  using BB = BBT;
  static constexpr int L = LV, NX = 4 + LV, NST = 8 + LV, HS = HSV, HP = HPV, NLAT = NLATV;  // NST: ODE states (species, latents, precisions)
  static constexpr int TPW = 16, TPB = 64;  // trajectories per wave / per 256-thread block
  // 16-trajectory groups = partial Gram tile sets the adjoint with on-chip weight gradients leaves (vihds_blackbox_split.hpp)
  __host__ __device__ static int gram_groups(int n) { return (n + TPW - 1) / TPW; }
  // hidden tiles of 16 slots; second-layer K-steps: four per full tile, ceil(units / 4) for the last one (its units are
  // numbered down the registers first: unit j of the tile sits in register j / 4 of quarter j % 4)
  __host__ __device__ static constexpr int tiles(int n) { return (n + 15) / 16; }
  __host__ __device__ static constexpr int ksteps(int n) { return 4 * (tiles(n) - 1) + (n - 16 * (tiles(n) - 1) + 3) / 4; }
  static constexpr int MS = tiles(HS), MP = tiles(HP), MT = MS > MP ? MS : MP;
  static constexpr int KS = ksteps(HS), KP = ksteps(HP);  // (25 / 20 units: 7 / 5)
  static constexpr int NG = 2 * MS + 2 * MP;               // Gram tiles per 16-trajectory group
  static_assert(L >= 1 && L <= 3 && MS <= 4 && MP <= 2 && NLAT <= 16,
                "matrix-core dr_blackbox: at most 3 latent species (4 + L states and the time fill two K-steps), 64 / 32 hidden units, 16 latent inputs");

  __device__ static f32x4 mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

  // hidden slot (16 m + 4 q + r) -> unit id or -1
  __host__ __device__ static int unit_of(int slot, int n_units) {
    const int M = (n_units + 15) / 16, m = slot >> 4;
    if (m >= M) return -1;
    if (m < M - 1) return slot;
    const int base = 16 * (M - 1), local = slot - base, r = local & 3, qq = local >> 2, j = 4 * r + qq;
    return j < n_units - base ? base + j : -1;
  }
  // K-step s of a second layer -> (tile m, register r)
  __device__ __host__ static constexpr int step_m(int s) { return s >> 2; }
  __device__ __host__ static constexpr int step_r(int s) { return s & 3; }

  struct Weights {  // A operands (one VGPR each) and second-layer biases, gathered once per kernel
    float w1s[MS][2], w2s[KS], w1p[MP][2], w2p[KP];
    f32x4 b2s, b2p;
  };
  struct WeightsT {  // transposed operands for the adjoint
    float w2sT[MS][4], w1sT[KS], w2pT[MP][2], w1pT[KP];
  };

  // input feature of first-layer K-step s for k-slot kq: 0..5 state, 6 = time, -1 = none
  __device__ static int l1_input(int s, int kq) { return s == 0 ? kq : (kq < L ? 4 + kq : (kq == L ? NX : -1)); }
  // output row i (0..15) of the states' second layer -> (is_degradation, state) or state = -1
  __device__ static int l2s_state(int i) { const int qq = i >> 2, r = i & 3; return r < 2 ? qq : (qq < L ? 4 + qq : -1); }
  __device__ static int l2s_degr(int i) { return i & 1; }
  __device__ static int l2p_out(int i) { return (i & 3) < 2 ? (i >> 2) : -1; }

  __device__ static void gather(const OdeArgs& a, int lane, Weights& W) {
    const typename BB::Off o = BB::offsets(a.n_const);
    const float* w = a.weights;
    const int i = lane & 15, kq = lane >> 4;
    _Pragma("unroll") for (int m = 0; m < MT; ++m)
      _Pragma("unroll") for (int s = 0; s < 2; ++s) {
        const int in = l1_input(s, kq);
        const int us = unit_of(16 * m + i, HS), up = unit_of(16 * m + i, HP);
        if (m < MS) W.w1s[m < MS ? m : 0][s] = (us >= 0 && in >= 0 && in < NX) ? w[o.wh + us * o.nin_s + in] : 0.f;
        if (m < MP) W.w1p[m < MP ? m : 0][s] = (up >= 0 && in >= 0) ? w[o.vh + up * o.nin_p + (in == NX ? 0 : 1 + in)] : 0.f;
      }
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {
      const int u = unit_of(16 * step_m(s) + 4 * kq + step_r(s), HS);
      const int st = l2s_state(i);
      W.w2s[s] = (u >= 0 && st >= 0) ? w[(l2s_degr(i) ? o.wd : o.wp) + st * HS + u] : 0.f;
    }
    _Pragma("unroll") for (int s = 0; s < KP; ++s) {
      const int u = unit_of(16 * step_m(s) + 4 * kq + step_r(s), HP);
      const int ou = l2p_out(i);
      W.w2p[s] = (u >= 0 && ou >= 0) ? w[((i & 1) ? o.vd : o.vp) + ou * HP + u] : 0.f;
    }
    // second-layer biases in C layout: lane (j, q) register r = bias of row 4q + r
    const int q = lane >> 4;
    _Pragma("unroll") for (int r = 0; r < 4; ++r) {
      const int row = 4 * q + r;
      const int st = l2s_state(row);
      W.b2s[r] = st >= 0 ? w[(l2s_degr(row) ? o.bd : o.bp) + st] : 0.f;
      const int ou = l2p_out(row);
      W.b2p[r] = ou >= 0 ? w[((row & 1) ? o.cd : o.cp) + ou] : 0.f;
    }
  }
  __device__ static void gather_t(const OdeArgs& a, int lane, WeightsT& W) {
    const typename BB::Off o = BB::offsets(a.n_const);
    const float* w = a.weights;
    const int i = lane & 15, kq = lane >> 4;
    // d h[slot 16m+i] = sum_rows W2[row][unit] d z[row]; K-step r carries rows 4kq + r
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {
      const int us = unit_of(16 * m + i, HS), up = unit_of(16 * m + i, HP);
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r, st = l2s_state(row);
        if (m < MS) W.w2sT[m < MS ? m : 0][r] = (us >= 0 && st >= 0) ? w[(l2s_degr(row) ? o.wd : o.wp) + st * HS + us] : 0.f;
      }
      _Pragma("unroll") for (int r = 0; r < 2; ++r) {
        const int row = 4 * kq + r, ou = l2p_out(row);
        if (m < MP) W.w2pT[m < MP ? m : 0][r] = (up >= 0 && ou >= 0) ? w[((row & 1) ? o.vd : o.vp) + ou * HP + up] : 0.f;
      }
    }
    // d y[row i]: row 4q'+0 = state q', row 4q'+1 = state 4+q' (q' < 2); K-step (m, r) carries hidden slot 16m + 4kq + r
    const int st = (i & 3) == 0 ? (i >> 2) : (((i & 3) == 1 && (i >> 2) < L) ? 4 + (i >> 2) : -1);
    _Pragma("unroll") for (int s = 0; s < KS; ++s) {
      const int u = unit_of(16 * step_m(s) + 4 * kq + step_r(s), HS);
      W.w1sT[s] = (u >= 0 && st >= 0) ? w[o.wh + u * o.nin_s + st] : 0.f;
    }
    _Pragma("unroll") for (int s = 0; s < KP; ++s) {
      const int u = unit_of(16 * step_m(s) + 4 * kq + step_r(s), HP);
      W.w1pT[s] = (u >= 0 && st >= 0) ? w[o.vh + u * o.nin_p + 1 + st] : 0.f;
    }
  }

  // time-invariant input c (0..n_const-1) of trajectory (i, b); 0 beyond
  __device__ static float const_input(const OdeArgs& a, int c, int i, int b) {
    if (c < NLAT) return a.theta[(size_t)a.slot_row[c] * a.n + i];
    if (c < NLAT + a.C) return a.cond[b * a.C + (c - NLAT)];
    if (c < a.n_const) return a.dev1hot[b * a.D + (c - NLAT - a.C)];
    return 0.f;
  }
  // hoisted first-layer accumulator initial values hc[net][tile] (C layout) via MFMA over the constant inputs
  __device__ static void hoist(const OdeArgs& a, int lane, int i, int b, f32x4 hc[2][MT]) {
    const typename BB::Off o = BB::offsets(a.n_const);
    const float* w = a.weights;
    const int ii = lane & 15, kq = lane >> 4, q = lane >> 4;
    _Pragma("unroll") for (int m = 0; m < MT; ++m)
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {
        const int us = unit_of(16 * m + 4 * q + r, HS), up = unit_of(16 * m + 4 * q + r, HP);
        hc[0][m][r] = us >= 0 ? w[o.bh + us] : 0.f;
        hc[1][m][r] = up >= 0 ? w[o.ch + up] : 0.f;
      }
    const int nsteps = (a.n_const + 3) / 4;
    for (int s = 0; s < nsteps; ++s) {
      const int c = 4 * s + kq;
      const float bval = const_input(a, c, i, b);
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {
        const int us = unit_of(16 * m + ii, HS), up = unit_of(16 * m + ii, HP);
        const float as_ = (us >= 0 && c < a.n_const) ? w[o.wh + us * o.nin_s + NX + c] : 0.f;
        const float ap_ = (up >= 0 && c < a.n_const) ? w[o.vh + up * o.nin_p + 1 + NX + c] : 0.f;
        if (m < MS) hc[0][m] = mfma(as_, bval, hc[0][m]);
        if (m < MP) hc[1][m] = mfma(ap_, bval, hc[1][m]);
      }
    }
  }

  // ---- weight gradients on chip (round 2) -------------------------------------------------------------------------
  // Every Gram-type weight gradient is  G[i][j] = sum over (evaluation, trajectory) of X[i][traj] Y[j][traj]  with X a
  // tile of pre-activation adjoints and Y a tile of layer inputs, both already in registers in the C/D layout (lane =
  // (trajectory, quarter)).  The contraction index of an MFMA is K, so the tiles are turned into "row" layout -- lane
  // (row i, k-slot kq), register s = T[i][trajectory 4s + kq], which is the A layout and the B layout at once -- through
  // an LDS buffer (one 16-byte store and four loads per tile; since round 3 the stores are the main wavefronts', the loads and
  // the MFMAs the helper wavefronts': vihds_blackbox_split.hpp), and 32 MFMAs per evaluation accumulate the eight 16x16
  // output tiles (at the ICML sizes) in registers:
  //   tiles 0,1: d z (states' 2nd layer, 16 rows) x h[m]     -> Wp / Wd        tiles 2,3: gs[m] x inputs -> Wh
  //   tiles 4,5: d zp (precisions' 2nd layer)     x g[m]     -> Vp / Vd        tiles 6,7: gp[m] x inputs -> Vh
  // The 573 MB per-evaluation dump and its contraction pass (vihds_gram_blocks) disappear; a wavefront leaves 8 KB of
  // partial sums, added up in a fixed order by bb_gram_reduce_kernel.
  static constexpr int GT_LD = 20, GT_TILE = 16 * GT_LD, GT_NT = 3 + 2 * MS + 2 * MP, GT_WAVE = GT_NT * GT_TILE;  // floats
  // LDS tile slots of one hand-over: dz, dzp, inputs, then h[m], gs[m] (states), g[m], gp[m] (precisions)
  static constexpr int T_DZ = 0, T_DZP = 1, T_IN = 2, T_H = 3, T_GS = T_H + MS, T_G = T_GS + MS, T_GP = T_G + MP;
  __device__ __forceinline__ static void put_cols(float* buf, const f32x4& t, int lane) {
    *reinterpret_cast<f32x4*>(buf + (lane & 15) * GT_LD + 4 * (lane >> 4)) = t;
  }
  __device__ __forceinline__ static f32x4 get_rows(const float* buf, int lane) {
    f32x4 o;
#pragma unroll
    for (int s = 0; s < 4; ++s) o[s] = buf[(4 * s + (lane >> 4)) * GT_LD + (lane & 15)];
    return o;
  }
  __device__ __forceinline__ static void gram_acc(f32x4& G, const f32x4& X, const f32x4& Y) {
#pragma unroll
    for (int s = 0; s < 4; ++s) G = mfma(X[s], Y[s], G);
  }
  // workgroup barrier for data handed over through LDS (waits for this wavefront's LDS operations only: the main
  // wavefronts keep their global prefetches in flight across it)
  __device__ __forceinline__ static void pair_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  // weight index of element (tile, lane, register) of a group's partial sums, or -1.  Tiles: [0, MS) dz x h[m] -> Wp / Wd;
  // [MS, 2 MS) gs[m] x inputs -> Wh; [2 MS, 2 MS + MP) dzp x g[m] -> Vp / Vd; [2 MS + MP, NG) gp[m] x inputs -> Vh
  __host__ __device__ static int gram_dest(int tile, int lane, int reg, int n_const) {
    const typename BB::Off o = BB::offsets(n_const);
    const int row = 4 * (lane >> 4) + reg, col = lane & 15;
    // the input tile's rows: 4q = state q, 4q+1 = latent state 4+q (q < 2), row 2 = time
    auto input_of = [](int j) { return (j & 3) == 0 ? (j >> 2) : (((j & 3) == 1 && (j >> 2) < L) ? 4 + (j >> 2) : (j == 2 ? NX : -1)); };
    if (tile < MS) {
      const int u = unit_of(16 * tile + col, HS), st = l2s_state_h(row);
      return (u >= 0 && st >= 0) ? ((row & 1) ? o.wd : o.wp) + st * HS + u : -1;
    }
    if (tile < 2 * MS) {
      const int u = unit_of(16 * (tile - MS) + row, HS), in = input_of(col);
      return (u >= 0 && in >= 0 && in < NX) ? o.wh + u * o.nin_s + in : -1;
    }
    if (tile < 2 * MS + MP) {
      const int u = unit_of(16 * (tile - 2 * MS) + col, HP), ou = (row & 3) < 2 ? (row >> 2) : -1;
      return (u >= 0 && ou >= 0) ? ((row & 1) ? o.vd : o.vp) + ou * HP + u : -1;
    }
    const int u = unit_of(16 * (tile - 2 * MS - MP) + row, HP), in = input_of(col);
    return (u >= 0 && in >= 0) ? o.vh + u * o.nin_p + (in == NX ? 0 : 1 + in) : -1;
  }
  __host__ __device__ static int l2s_state_h(int i) { const int qq = i >> 2, r = i & 3; return r < 2 ? qq : (qq < L ? 4 + qq : -1); }
  // floats of aux ahead of the tail when the Gram tiles are accumulated on chip
  __host__ __device__ static size_t gram_floats(int n) { return (size_t)gram_groups(n) * NG * 256; }
};

// ---- the ICML sizes (specs/dr_blackbox_icml.yaml:17-31) with the one-wavefront-per-group formulation on top -------------
struct BbMfma : BbMfmaT<Blackbox<2, 25, 20, 5, 5, 2>, 2, 25, 20, 12> {};
// (Round 1-5 also kept the ONE-wavefront-per-group kernels here -- forward, and an adjoint dumping every evaluation for
// vihds_gram_blocks: `kernel_variant 4`.  The cooperating-wavefront kernels of vihds_blackbox_split.hpp superseded them at every
// size; removed in round 6.)

// floats of aux ahead of the tail (Delta, bias sums): the per-evaluation dump, or the wavefronts' Gram partial sums
__host__ __device__ inline size_t bb_mfma_head_floats(int n, int T, int solver, bool gram) {
  using BB = BbMfma::BB;
  if (gram) return BbMfma::gram_floats(n);
  return (size_t)(T - 1) * BB::stages(solver) * BB::NF * n;
}

// sums the wavefronts' partial tiles in a fixed order (deterministic) and scatters them into the flat weight gradient.
// One block = 64 elements of a tile x 16 interleaved slices of the wavefronts (as one thread per element walking all
// 450 wavefronts -- 113 dependent rounds of loads -- this took 39 us).
template <class K>
__global__ void __launch_bounds__(1024) bb_gram_reduce_kernel(int n_waves, int n_const, const float* __restrict__ partial,
                                                              float* __restrict__ g_weights) {
  __shared__ float part[16][64];
  constexpr size_t SET = (size_t)K::NG * 256;  // floats of one group's partial tiles
  const int tile = blockIdx.x >> 2, e = (blockIdx.x & 3) * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
  const float* src = partial + (size_t)tile * 256 + e;
  float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
  int w = slice;
  for (; w + 48 < n_waves; w += 64) {
    acc0 += src[(size_t)(w + 0) * SET];
    acc1 += src[(size_t)(w + 16) * SET];
    acc2 += src[(size_t)(w + 32) * SET];
    acc3 += src[(size_t)(w + 48) * SET];
  }
  for (; w < n_waves; w += 16) acc0 += src[(size_t)w * SET];
  part[slice][threadIdx.x & 63] = (acc0 + acc1) + (acc2 + acc3);
  __syncthreads();
  if (slice == 0) {
    const int dest = K::gram_dest(tile, e >> 2, e & 3, n_const);
    if (dest >= 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x & 63];
      g_weights[dest] = t;
    }
  }
}
template <class K>
inline void launch_bb_gram_reduce(const OdeArgs& a, const float* aux, float* g_weights, hipStream_t st) {
  const int n_waves = K::gram_groups(a.n);
  hipLaunchKernelGGL((bb_gram_reduce_kernel<K>), dim3(K::NG * 4), dim3(1024), 0, st, n_waves, a.n_const, aux, g_weights);
}

}  // namespace vihds
