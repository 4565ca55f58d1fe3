// Rectangular blocks of a Gram matrix over a field-major dump:  out[dest0 + i*dsa + j*dsb] = sum_c X[a0+i][c] * X[b0+j][c]
// for a handful of rectangles (a0, na, b0, nb), X = [F][C] with C ~ 10^6 columns.
//
// Used for the dr_blackbox weight gradients (reference: autograd through NeuralStates / NeuralPrecisions,
// vihds/ode.py:134-138, vihds/precisions.py:76-87): the adjoint kernel dumps, per RHS evaluation and trajectory, the
// pre-activation adjoints and the layer inputs ([F = 117][E evaluations][n trajectories], 573 MB at B=36, S=200, T=86,
// midpoint); every weight gradient is the dot product of two of those rows over all C = E*n columns, and the pairs
// form seven dense rectangles (hidden-adjoint x inputs, output-adjoint x hidden, for both networks).  The first
// version ran this as five strided batched library GEMMs plus ~25 small reductions / concatenations (320 + 150 us).
//
// Here the dump is read ONCE.  A block stages a [F][128]-column tile in LDS (60 KB, so two blocks per CU overlap each
// other's loads); a thread owns a 4x4 register tile of one rectangle and one of four interleaved column groups (8 LDS
// reads per 16 FMAs -- one product per thread would be LDS-bandwidth bound at 2 reads per FMA) and accumulates over
// the block's tiles; per-block partials are then summed in a fixed order (deterministic) by a second kernel, straight
// into the flat weight-gradient buffer.
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"

namespace vihds {

constexpr int GRAM_TILE = 128;          // columns per tile (60 KB of LDS at 117 fields: two blocks per CU)
constexpr int GRAM_LD = GRAM_TILE + 1;  // padded row stride in LDS
constexpr int GRAM_CG = 4;              // interleaved column groups per register tile
constexpr int GRAM_MAX_THREADS = 512;

struct GramPlan {
  int n_rect;
  vihds_gram_rect r[VIHDS_GRAM_MAX_RECTS];
  int first[VIHDS_GRAM_MAX_RECTS + 1];  // prefix sums of 4x4 register tiles per rectangle
};

__device__ __forceinline__ int gram_find(const GramPlan& pl, int tt, int& ta, int& tb) {
  int q = 0;
  while (q + 1 < pl.n_rect && tt >= pl.first[q + 1]) ++q;
  const int local = tt - pl.first[q];
  const int nbg = (pl.r[q].nb + 3) >> 2;
  ta = local / nbg;
  tb = local - ta * nbg;
  return q;
}

__global__ void __launch_bounds__(GRAM_MAX_THREADS)
gram_partial_kernel(int F, long long C, GramPlan pl, const float* __restrict__ X, float* __restrict__ partial) {
  extern __shared__ float tile[];  // [F][GRAM_LD]
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int n_tt = pl.first[pl.n_rect];
  const int tt = tid >> 2, cg = tid & 3;
  const bool active = tt < n_tt;
  int ta = 0, tb = 0;
  const int q = active ? gram_find(pl, tt, ta, tb) : 0;
  int ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {  // rows beyond the rectangle re-read its last row; their results are never stored
    ra[i] = (pl.r[q].a0 + min(4 * ta + i, pl.r[q].na - 1)) * GRAM_LD;
    rb[i] = (pl.r[q].b0 + min(4 * tb + i, pl.r[q].nb - 1)) * GRAM_LD;
  }
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const long long n_tiles = (C + GRAM_TILE - 1) / GRAM_TILE;
  const bool vec_ok = (C % 4 == 0) && ((reinterpret_cast<size_t>(X) & 15) == 0);
  for (long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const long long c0 = t * GRAM_TILE;
    const int w = (int)min((long long)GRAM_TILE, C - c0);
    if (w == GRAM_TILE && vec_ok) {
      // 16-byte loads, a batch of them requested before the first is consumed: the stage is a pure streaming read.
      // (Requesting the NEXT tile into registers before computing on this one -- a software pipeline -- measured
      // slower: 327 vs 217 us; the second resident block already fills the load gaps.)
      constexpr int LB = 6, R4 = GRAM_TILE / 4;  // float4 per row
      const int n4 = F * R4;
      for (int base = 0; base < n4; base += nthreads * LB) {
        float4 buf[LB];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int e = base + u * nthreads + tid;
          if (e < n4) buf[u] = *reinterpret_cast<const float4*>(X + (size_t)(e / R4) * C + c0 + 4 * (e % R4));
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int e = base + u * nthreads + tid;
          if (e < n4) {
            float* d = tile + (e / R4) * GRAM_LD + 4 * (e % R4);
            d[0] = buf[u].x; d[1] = buf[u].y; d[2] = buf[u].z; d[3] = buf[u].w;
          }
        }
      }
    } else {
      for (int e = tid; e < F * GRAM_TILE; e += nthreads) {
        const int f = e / GRAM_TILE, c = e - f * GRAM_TILE;
        tile[f * GRAM_LD + c] = c < w ? X[(size_t)f * C + c0 + c] : 0.f;
      }
    }
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int c = cg; c < GRAM_TILE; c += GRAM_CG) {
        float av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { av[i] = tile[ra[i] + c]; bv[i] = tile[rb[i] + c]; }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
      }
    }
    __syncthreads();
  }
  if (active) {
    float* dst = partial + ((size_t)blockIdx.x * n_tt * GRAM_CG + (size_t)tt * GRAM_CG + cg) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[i * 4 + j] = acc[i][j];
  }
}

// one wave per register tile entry group: out = sum over (block, column group) in a fixed order
__global__ void __launch_bounds__(256)
gram_reduce_kernel(int n_blocks, GramPlan pl, const float* __restrict__ partial, float* __restrict__ out) {
  const int n_tt = pl.first[pl.n_rect];
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n_tt * 16) return;
  const int tt = wave >> 4, ij = wave & 15, i = ij >> 2, j = ij & 3;
  int ta, tb;
  const int q = gram_find(pl, tt, ta, tb);
  const int ai = 4 * ta + i, bj = 4 * tb + j;
  if (ai >= pl.r[q].na || bj >= pl.r[q].nb) return;  // (uniform per wave)
  float s = 0.f;
  const int n_src = n_blocks * GRAM_CG;
  for (int k = lane; k < n_src; k += 64) {
    const int blk = k / GRAM_CG, cg = k - blk * GRAM_CG;
    s += partial[((size_t)blk * n_tt * GRAM_CG + (size_t)tt * GRAM_CG + cg) * 16 + ij];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) out[pl.r[q].dest0 + ai * pl.r[q].dest_stride_a + bj * pl.r[q].dest_stride_b] = s;
}

static int gram_make_plan(int n_rect, const vihds_gram_rect* rects, GramPlan& pl) {
  if (n_rect <= 0 || n_rect > VIHDS_GRAM_MAX_RECTS) return VIHDS_E_BADARG;
  pl.n_rect = n_rect;
  pl.first[0] = 0;
  for (int q = 0; q < n_rect; ++q) {
    if (rects[q].na <= 0 || rects[q].nb <= 0 || rects[q].a0 < 0 || rects[q].b0 < 0) return VIHDS_E_BADARG;
    pl.r[q] = rects[q];
    pl.first[q + 1] = pl.first[q] + ((rects[q].na + 3) / 4) * ((rects[q].nb + 3) / 4);
  }
  return pl.first[n_rect] * GRAM_CG <= GRAM_MAX_THREADS ? VIHDS_OK : VIHDS_E_UNSUPPORTED;
}
int gram_blocks(long long C) {
  const long long n_tiles = (C + GRAM_TILE - 1) / GRAM_TILE;
  return (int)min(n_tiles, (long long)512);  // two resident blocks per CU
}
long long gram_scratch_floats(long long C, int n_rect, const vihds_gram_rect* rects) {
  GramPlan pl;
  if (gram_make_plan(n_rect, rects, pl) != VIHDS_OK) return -1;
  return (long long)gram_blocks(C) * pl.first[n_rect] * GRAM_CG * 16;
}
int launch_gram(int F, long long C, int n_rect, const vihds_gram_rect* rects, const float* X, float* partial, float* out,
                hipStream_t st) {
  GramPlan pl;
  if (int rc = gram_make_plan(n_rect, rects, pl)) return rc;
  for (int q = 0; q < n_rect; ++q)
    if (rects[q].a0 + rects[q].na > F || rects[q].b0 + rects[q].nb > F) return VIHDS_E_BADARG;
  const size_t lds = (size_t)F * GRAM_LD * sizeof(float);
  if (lds > 64 * 1024) return VIHDS_E_UNSUPPORTED;
  const int nb = gram_blocks(C);
  const int threads = ((pl.first[n_rect] * GRAM_CG + 63) / 64) * 64;
  hipLaunchKernelGGL(gram_partial_kernel, dim3(nb), dim3(threads), lds, st, F, C, pl, X, partial);
  const int n_waves = pl.first[n_rect] * 16;
  hipLaunchKernelGGL(gram_reduce_kernel, dim3((n_waves + 3) / 4), dim3(256), 0, st, nb, pl, partial, out);
  return VIHDS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// dr_blackbox weight gradients that are NOT Gram rectangles over the dump (include/vihds_hip.h:
// vihds_blackbox_tail_grads).  The adjoint kernel leaves, behind the dump, Delta[h][n] = sum over RHS evaluations of
// the hidden pre-activation adjoint of unit h for trajectory n, and the per-trajectory output-bias adjoint sums.  The
// hidden layers' columns for the 21 time-invariant inputs (latent theta rows, treatments, device one-hot) are
//   g_W[h][k] = sum_n Delta[h][n] * const_k(n),
// the hidden biases sum_n Delta[h][n] and the output biases sum_n tail[r][n].  One block per tail row; the constants
// are read where they already live (theta rows are contiguous in n; treatments / one-hot are per plate row b = n / S).
constexpr int TAIL_LAT = 16, TAIL_C = 4, TAIL_D = 12, TAIL_ACC = TAIL_LAT + TAIL_C + TAIL_D + 1, TAIL_THREADS = 1024;
struct BbTailArgs {
  int n_rows, n_dot_rows, n, S, n_lat, C, D;
  int lat_row[TAIL_LAT];
};

__global__ void __launch_bounds__(TAIL_THREADS)
bb_tail_kernel(BbTailArgs a, const float* __restrict__ theta, const float* __restrict__ cond,
               const float* __restrict__ dev1hot, const float* __restrict__ tail, const int* __restrict__ dest,
               float* __restrict__ out) {
  __shared__ float sm[TAIL_THREADS / 64][TAIL_ACC];
  const int r = blockIdx.x, nc = a.n_lat + a.C + a.D;
  const bool dot = r < a.n_dot_rows;
  const float* row = tail + (size_t)r * a.n;
  // accumulators: [0,16) latent inputs, [16,20) treatments, [20,32) device one-hot, [32] the plain row sum.  Slots past
  // the actual counts read a clamped (valid) address and are never written out: no branch sits between the loads of
  // one column, so they are all in flight together
  float acc[TAIL_ACC];
#pragma unroll
  for (int k = 0; k < TAIL_ACC; ++k) acc[k] = 0.f;
  for (int col = threadIdx.x; col < a.n; col += TAIL_THREADS) {
    const float v = row[col];
    acc[TAIL_ACC - 1] += v;
    if (dot) {
      const int b = col / a.S;
      float c[TAIL_ACC - 1];
#pragma unroll
      for (int k = 0; k < TAIL_LAT; ++k) c[k] = theta[(size_t)a.lat_row[k] * a.n + col];
#pragma unroll
      for (int k = 0; k < TAIL_C; ++k) c[TAIL_LAT + k] = cond[b * a.C + min(k, a.C - 1)];
#pragma unroll
      for (int k = 0; k < TAIL_D; ++k) c[TAIL_LAT + TAIL_C + k] = dev1hot[b * a.D + min(k, a.D - 1)];
#pragma unroll
      for (int k = 0; k < TAIL_ACC - 1; ++k) acc[k] = fmaf(v, c[k], acc[k]);
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < TAIL_ACC; ++k) {
    float t = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) sm[wid][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < TAIL_ACC) {
    const int j = threadIdx.x;
    float t = 0.f;
    for (int w = 0; w < TAIL_THREADS / 64; ++w) t += sm[w][j];
    int k = -1;  // which time-invariant input this accumulator belongs to
    if (j < TAIL_LAT) k = j < a.n_lat ? j : -1;
    else if (j < TAIL_LAT + TAIL_C) k = j - TAIL_LAT < a.C ? a.n_lat + (j - TAIL_LAT) : -1;
    else if (j < TAIL_ACC - 1) k = j - TAIL_LAT - TAIL_C < a.D ? a.n_lat + a.C + (j - TAIL_LAT - TAIL_C) : -1;
    if (j == TAIL_ACC - 1) out[dest[a.n_dot_rows * nc + r]] = t;
    else if (dot && k >= 0) out[dest[r * nc + k]] = t;
  }
}

int launch_bb_tail(int n_rows, int n_dot_rows, int n, int S, int n_lat, int C, int D, const int* lat_row,
                   const float* theta, const float* cond, const float* dev1hot, const float* tail, const int* dest,
                   float* out, hipStream_t st) {
  if (n_lat < 1 || n_lat > TAIL_LAT || C < 1 || C > TAIL_C || D < 1 || D > TAIL_D) return VIHDS_E_UNSUPPORTED;
  BbTailArgs a;
  a.n_rows = n_rows; a.n_dot_rows = n_dot_rows; a.n = n; a.S = S; a.n_lat = n_lat; a.C = C; a.D = D;
  for (int k = 0; k < TAIL_LAT; ++k) a.lat_row[k] = lat_row[k < n_lat ? k : n_lat - 1];
  hipLaunchKernelGGL(bb_tail_kernel, dim3(n_rows), dim3(TAIL_THREADS), 0, st, a, theta, cond, dev1hot, tail, dest, out);
  return VIHDS_OK;
}

}  // namespace vihds
