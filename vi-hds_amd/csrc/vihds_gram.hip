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
#include "vihds_wave.hpp"

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

// ---------------------------------------------------------------------------------------------------------------
// The same contraction on the matrix cores.  The sum over columns IS the K dimension of a GEMM, and the dump's
// field-major layout is already what v_mfma_f32_16x16x4_f32 (exact fp32) wants from both operands: lane l supplies
// row (l & 15) of a 16-row block at four consecutive columns 4*(l >> 4) .. +3 -- one 16-byte load, no LDS, no
// transposes.  Every rectangle is cut into <= 16 x 16 products on the host (14 for the dr_blackbox plan); a wave walks
// its share of the columns 64 at a time (sixteen MFMAs per product and step), the four waves of a block are added in a
// fixed order through LDS, and a second kernel sums the per-block partials, again in a fixed order.  Rows past a
// block's end re-read its last row (their products are never stored).
typedef float gm_f32x4 __attribute__((ext_vector_type(4)));
constexpr int GM_MAX_PROD = 16, GM_WAVES = 4, GM_BLOCKS = 512;
struct GramMfmaPlan {
  int n_prod;
  int a_row0[GM_MAX_PROD], a_rows[GM_MAX_PROD], b_row0[GM_MAX_PROD], b_rows[GM_MAX_PROD];
  int dest0[GM_MAX_PROD], dsa[GM_MAX_PROD], dsb[GM_MAX_PROD];
  int out_first[GM_MAX_PROD + 1];  // prefix sums of a_rows * b_rows
  // the distinct 16-row blocks the products are made of, in first-use order, and each product's two blocks
  int n_tiles, t_row0[GM_MAX_PROD], t_rows[GM_MAX_PROD], pa[GM_MAX_PROD], pb[GM_MAX_PROD];
};

// Which products share which row blocks, fixed at compile time so that every block is loaded ONCE per step into
// registers the compiler can name (a run-time block index would mean indexed registers or one load per use -- the
// latter was measured: 28 block loads per step instead of 14 make the kernel L2-bound at 262 us).  This is the
// dr_blackbox plan of vihds/ops.py (_blackbox_grad_plan): blocks in first-use order
//   0 gs[0:16] 1 y 2 gs[16:] 3 za 4 hs[0:16] 5 hs[16:] 6 zd 7 gp[0:16] 8 t 9 gp[16:] 10 zap 11 hp[0:16] 12 hp[16:] 13 zdp
// Row numbers and counts stay run-time; only the sharing pattern is matched (gram_mfma_schema_matches).
struct BbGramSchema {
  static constexpr int NT = 14, NPR = 14;
  static constexpr int TA[NPR] = {0, 2, 3, 3, 6, 6, 7, 9, 7, 9, 10, 10, 13, 13};
  static constexpr int TB[NPR] = {1, 1, 4, 5, 4, 5, 8, 8, 1, 1, 11, 12, 11, 12};
};

template <class SC>
__global__ void __launch_bounds__(64 * GM_WAVES)
gram_mfma_schema_kernel(long long C, GramMfmaPlan pl, const float* __restrict__ X, float* __restrict__ partial) {
  __shared__ float red[GM_MAX_PROD * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
  gm_f32x4 acc[SC::NPR];
  int off[SC::NT];
#pragma unroll
  for (int p = 0; p < SC::NPR; ++p) acc[p] = gm_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < SC::NT; ++t) off[t] = (int)((long long)(pl.t_row0[t] + min(m, pl.t_rows[t] - 1)) * C);
  // a wave takes 32 columns per step: lane (m, kq) reads columns 8*kq .. 8*kq+7 of row m of every block (two 16-byte
  // loads), the four waves of a block take neighbouring groups: 512 contiguous bytes per dump row and block step
  const long long n_groups = C >> 5, stride = (long long)gridDim.x * GM_WAVES;
  for (long long g = (long long)blockIdx.x * GM_WAVES + wid; g < n_groups; g += stride) {
    const float* base = X + (g << 5) + 8 * kq;
    float4 v[SC::NT][2];
#pragma unroll
    for (int t = 0; t < SC::NT; ++t) {
      v[t][0] = *reinterpret_cast<const float4*>(base + off[t]);
      v[t][1] = *reinterpret_cast<const float4*>(base + off[t] + 4);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
      for (int p = 0; p < SC::NPR; ++p) {
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[SC::TA[p]][c].x, v[SC::TB[p]][c].x, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[SC::TA[p]][c].y, v[SC::TB[p]][c].y, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[SC::TA[p]][c].z, v[SC::TB[p]][c].z, acc[p], 0, 0, 0);
        acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[SC::TA[p]][c].w, v[SC::TB[p]][c].w, acc[p], 0, 0, 0);
      }
    }
  }
  for (int wv = 0; wv < GM_WAVES; ++wv) {
    if (wid == wv) {
#pragma unroll
      for (int p = 0; p < SC::NPR; ++p) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int idx = p * 256 + (4 * kq + r) * 16 + m;
          red[idx] = (wv ? red[idx] : 0.f) + acc[p][r];
        }
      }
    }
    __syncthreads();
  }
  const int n_out = SC::NPR * 256;
  for (int e = threadIdx.x; e < n_out; e += 64 * GM_WAVES) partial[(size_t)blockIdx.x * n_out + e] = red[e];
}

template <class SC>
static bool gram_mfma_schema_matches(const GramMfmaPlan& pl) {
  if (pl.n_prod != SC::NPR || pl.n_tiles != SC::NT) return false;
  for (int p = 0; p < SC::NPR; ++p)
    if (pl.pa[p] != SC::TA[p] || pl.pb[p] != SC::TB[p]) return false;
  return true;
}

__global__ void __launch_bounds__(64 * GM_WAVES)
gram_mfma_kernel(long long C, GramMfmaPlan pl, const float* __restrict__ X, float* __restrict__ partial) {
  __shared__ float red[GM_MAX_PROD * 256];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, m = lane & 15, kq = lane >> 4;
  gm_f32x4 acc[GM_MAX_PROD];
  int off_a[GM_MAX_PROD], off_b[GM_MAX_PROD];
#pragma unroll
  for (int p = 0; p < GM_MAX_PROD; ++p) {
    acc[p] = gm_f32x4{0.f, 0.f, 0.f, 0.f};
    const bool on = p < pl.n_prod;
    off_a[p] = on ? (int)((long long)(pl.a_row0[p] + min(m, pl.a_rows[p] - 1)) * C) : 0;
    off_b[p] = on ? (int)((long long)(pl.b_row0[p] + min(m, pl.b_rows[p] - 1)) * C) : 0;
  }
  // a wave takes 64 columns per step: lane (m, kq) reads the 16 consecutive columns 16*kq.. of row m (64 contiguous
  // bytes, the four kq lanes together 256), and the four waves of a block take neighbouring 64-column groups -- 1 KB
  // contiguous per dump row and block step, which is what keeps the HBM pages open.  Which column a k index of the MFMA
  // stands for does not matter as long as both operands agree.
  const long long n_groups = C >> 6, stride = (long long)gridDim.x * GM_WAVES;
  for (long long g = (long long)blockIdx.x * GM_WAVES + wid; g < n_groups; g += stride) {
    const float* base = X + (g << 6) + 16 * kq;
#pragma unroll
    for (int p = 0; p < GM_MAX_PROD; ++p) {
      if (p < pl.n_prod) {
        float4 a[4], b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a[c] = *reinterpret_cast<const float4*>(base + off_a[p] + 4 * c);
          b[c] = *reinterpret_cast<const float4*>(base + off_b[p] + 4 * c);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].x, b[c].x, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].y, b[c].y, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].z, b[c].z, acc[p], 0, 0, 0);
          acc[p] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c].w, b[c].w, acc[p], 0, 0, 0);
        }
      }
    }
  }
  // D layout: lane holds rows 4*kq + r (operand A's rows), column m (operand B's rows)
  for (int wv = 0; wv < GM_WAVES; ++wv) {
    if (wid == wv) {
#pragma unroll
      for (int p = 0; p < GM_MAX_PROD; ++p) {
        if (p < pl.n_prod) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int idx = p * 256 + (4 * kq + r) * 16 + m;
            red[idx] = (wv ? red[idx] : 0.f) + acc[p][r];
          }
        }
      }
    }
    __syncthreads();
  }
  const int n_out = pl.n_prod * 256;
  for (int e = threadIdx.x; e < n_out; e += 64 * GM_WAVES) partial[(size_t)blockIdx.x * n_out + e] = red[e];
}

__global__ void __launch_bounds__(256)
gram_mfma_reduce_kernel(int n_blocks, GramMfmaPlan pl, const float* __restrict__ partial, float* __restrict__ out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= pl.out_first[pl.n_prod]) return;
  int p = 0;
  while (p + 1 < pl.n_prod && wave >= pl.out_first[p + 1]) ++p;
  const int local = wave - pl.out_first[p];
  const int i = local / pl.b_rows[p], j = local - i * pl.b_rows[p];
  float s = 0.f;
  for (int blk = lane; blk < n_blocks; blk += 64) s += partial[((size_t)blk * pl.n_prod + p) * 256 + i * 16 + j];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  if (lane == 0) out[pl.dest0[p] + i * pl.dsa[p] + j * pl.dsb[p]] = s;
}

// VIHDS_OK, or VIHDS_E_UNSUPPORTED when the matrix-core path does not apply (then the LDS-tiled kernels above run)
static int gram_mfma_make_plan(int F, long long C, int n_rect, const vihds_gram_rect* rects, GramMfmaPlan& pl) {
  if ((C & 31) != 0 || (long long)F * C >= (1ll << 31)) return VIHDS_E_UNSUPPORTED;
  pl.n_prod = 0;
  pl.n_tiles = 0;
  pl.out_first[0] = 0;
  auto tile_of = [&pl](int row0, int rows) {
    for (int t = 0; t < pl.n_tiles; ++t)
      if (pl.t_row0[t] == row0 && pl.t_rows[t] == rows) return t;
    if (pl.n_tiles == GM_MAX_PROD) return -1;
    pl.t_row0[pl.n_tiles] = row0;
    pl.t_rows[pl.n_tiles] = rows;
    return pl.n_tiles++;
  };
  for (int q = 0; q < n_rect; ++q) {
    for (int ti = 0; 16 * ti < rects[q].na; ++ti)
      for (int tj = 0; 16 * tj < rects[q].nb; ++tj) {
        if (pl.n_prod == GM_MAX_PROD) return VIHDS_E_UNSUPPORTED;
        const int p = pl.n_prod++;
        pl.a_row0[p] = rects[q].a0 + 16 * ti;
        pl.a_rows[p] = min(16, rects[q].na - 16 * ti);
        pl.b_row0[p] = rects[q].b0 + 16 * tj;
        pl.b_rows[p] = min(16, rects[q].nb - 16 * tj);
        pl.dsa[p] = rects[q].dest_stride_a;
        pl.dsb[p] = rects[q].dest_stride_b;
        pl.dest0[p] = rects[q].dest0 + 16 * ti * pl.dsa[p] + 16 * tj * pl.dsb[p];
        pl.out_first[p + 1] = pl.out_first[p] + pl.a_rows[p] * pl.b_rows[p];
        pl.pa[p] = tile_of(pl.a_row0[p], pl.a_rows[p]);
        pl.pb[p] = tile_of(pl.b_row0[p], pl.b_rows[p]);
      }
  }
  for (int t = pl.n_tiles; t < GM_MAX_PROD; ++t) { pl.t_row0[t] = 0; pl.t_rows[t] = 1; }
  for (int p = pl.n_prod; p < GM_MAX_PROD; ++p) {
    pl.a_row0[p] = pl.b_row0[p] = pl.dest0[p] = pl.dsa[p] = pl.dsb[p] = 0;
    pl.a_rows[p] = pl.b_rows[p] = 1;
    pl.pa[p] = pl.pb[p] = -1;
    pl.out_first[p + 1] = pl.out_first[p];
  }
  return VIHDS_OK;
}
static int gram_mfma_blocks(long long C) { return (int)min((long long)GM_BLOCKS, ((C >> 6) + GM_WAVES - 1) / GM_WAVES); }

long long gram_scratch_floats(long long C, int n_rect, const vihds_gram_rect* rects) {
  GramPlan pl;
  if (gram_make_plan(n_rect, rects, pl) != VIHDS_OK) return -1;
  const long long lds_tiled = (long long)gram_blocks(C) * pl.first[n_rect] * GRAM_CG * 16;
  const long long mfma = (long long)GM_BLOCKS * GM_MAX_PROD * 256;
  return max(lds_tiled, mfma);
}
int launch_gram(int F, long long C, int n_rect, const vihds_gram_rect* rects, const float* X, float* partial, float* out,
                hipStream_t st) {
  GramPlan pl;
  if (int rc = gram_make_plan(n_rect, rects, pl)) return rc;
  for (int q = 0; q < n_rect; ++q)
    if (rects[q].a0 + rects[q].na > F || rects[q].b0 + rects[q].nb > F) return VIHDS_E_BADARG;
  GramMfmaPlan mp;
  const bool mfma_ok = gram_mfma_make_plan(F, C, n_rect, rects, mp) == VIHDS_OK;
  const bool schema = mfma_ok && gram_mfma_schema_matches<BbGramSchema>(mp);
  if (schema || (mfma_ok && (C & 63) == 0)) {  // (the generic kernel steps 64 columns at a time, the schema one 32)
    const int nb = gram_mfma_blocks(C);
    if (schema)
      hipLaunchKernelGGL(gram_mfma_schema_kernel<BbGramSchema>, dim3(nb), dim3(64 * GM_WAVES), 0, st, C, mp, X, partial);
    else
      hipLaunchKernelGGL(gram_mfma_kernel, dim3(nb), dim3(64 * GM_WAVES), 0, st, C, mp, X, partial);
    const int n_out = mp.out_first[mp.n_prod];
    hipLaunchKernelGGL(gram_mfma_reduce_kernel, dim3((n_out + 3) / 4), dim3(256), 0, st, nb, mp, partial, out);
    return VIHDS_OK;
  }
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
  // (a thread's columns are walked two at a time with every load of the pair requested before the first FMA: at 7 200
  // trajectories the seven rounds of a thread were seven memory round trips in a row; columns past the end read a valid
  // address at weight 0)
  for (int col0 = threadIdx.x; col0 < a.n; col0 += 2 * TAIL_THREADS) {
    float v[2], c[2][TAIL_ACC - 1];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int cu = col0 + u * TAIL_THREADS;
      const int col = cu < a.n ? cu : col0;
      v[u] = row[col];
      if (dot) {
        const int b = col / a.S;
#pragma unroll
        for (int k = 0; k < TAIL_LAT; ++k) c[u][k] = theta[(size_t)a.lat_row[k] * a.n + col];
#pragma unroll
        for (int k = 0; k < TAIL_C; ++k) c[u][TAIL_LAT + k] = cond[b * a.C + min(k, a.C - 1)];
#pragma unroll
        for (int k = 0; k < TAIL_D; ++k) c[u][TAIL_LAT + TAIL_C + k] = dev1hot[b * a.D + min(k, a.D - 1)];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float vv = col0 + u * TAIL_THREADS < a.n ? v[u] : 0.f;
      acc[TAIL_ACC - 1] += vv;
      if (dot) {
#pragma unroll
        for (int k = 0; k < TAIL_ACC - 1; ++k) acc[k] = fmaf(vv, c[u][k], acc[k]);
      }
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < TAIL_ACC; ++k) {
    // (DPP scan: 33 shuffle trees per wavefront, 16 wavefronts, were ~3 000 ds_bpermute through the CU's one LDS pipe)
    const float t = wave_total(acc[k]);
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
