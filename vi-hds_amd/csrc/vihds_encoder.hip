// q(theta | data) encoder of the VAE (reference vihds/encoders.py): Conv1d -> AvgPool1d(stride 1) -> flatten ->
// Linear -> tanh (ConditionalEncoder, :16-55), then one (mu, log_prec) pair of Linear(n,1) heads per local parameter
// on [hidden, treatments?, device one-hot?] (Q_Local :126-169), bias-free heads per global-conditioned parameter on
// [treatments?, device one-hot?] (Q_Global_Cond :172-213), free scalars per global parameter (Q_Global :216-239) and
// fixed constants (Q_Constant :242-253).
//
// The arithmetic is tiny (2.6 MFLOP forward at B=36) but as framework ops it is 8 forward + 15 backward launches per
// training step -- about half of the step once the ODE is fused, because a launch costs 4-5 us inside the step's
// hipGraph whatever it computes.  Here: ONE forward launch (block per data row, everything staged through LDS,
// results written straight into the theta kernel's level-blocked [2P,B] table) and TWO backward launches (per-row
// chain, then all parameter-gradient reductions over the rows in a fixed order: deterministic).
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"

namespace vihds {

namespace {
struct EncDims {
  int Lc, Lp, NPOOL, NX, NG;  // conv length, pooled length, F*Lp, local head inputs, gcond head inputs
};
__host__ __device__ inline EncDims enc_dims(const vihds_encoder_shape& s) {
  EncDims d;
  d.Lc = s.L - s.K + 1;
  d.Lp = d.Lc - s.pool + 1;
  d.NPOOL = s.F * d.Lp;
  d.NX = s.H + (s.l_tr ? s.n_tr : 0) + (s.l_dv ? s.D : 0);
  d.NG = (s.g_tr ? s.n_tr : 0) + (s.g_dv ? s.D : 0);
  return d;
}
__device__ __forceinline__ float wave_sum_e(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
// the same sum as a DPP scan (row_shr 1, 2, 4, 8, row_bcast 15, 31; total in lane 63, returned uniformly): six dependent
// VALU steps instead of six ds_bpermute round trips through the LDS crossbar -- the forward kernel is a chain of such sums
#ifdef VIHDS_TAIL_STAMPS
// profiling build (tests/probe/enc_stamps.py): every wavefront writes the 100 MHz wall clock at each phase boundary
static __device__ unsigned long long* vihds_enc_stamp_buf = nullptr;  // [256 blocks][16 waves][8]
#define VIHDS_ENC_STOP(PH)                                                                          \
  if (vihds_enc_stamp_buf && (threadIdx.x & 63) == 0 && blockIdx.x < 256)                              \
    vihds_enc_stamp_buf[(((size_t)blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (PH)] = wall_clock64();
#else
#define VIHDS_ENC_STOP(PH)
#endif
// barrier for phases that hand data over through LDS only (`__syncthreads()` also drains every outstanding global load)
__device__ __forceinline__ void enc_sync_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float enc_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_total_e(float v) {
  v += enc_dpp<0x111, 0xf>(v);
  v += enc_dpp<0x112, 0xf>(v);
  v += enc_dpp<0x114, 0xf>(v);
  v += enc_dpp<0x118, 0xf>(v);
  v += enc_dpp<0x142, 0xa>(v);
  v += enc_dpp<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// head inputs of data row b: local = [hidden, treatments?, dev_1hot?]; gcond = [treatments?, dev_1hot?]
__device__ __forceinline__ float local_input(const vihds_encoder_shape& s, const float* hid, const float* inputs,
                                             const float* dev1hot, int b, int i) {
  if (i < s.H) return hid[i];
  i -= s.H;
  if (s.l_tr) {
    if (i < s.n_tr) return inputs[b * s.n_tr + i];
    i -= s.n_tr;
  }
  return dev1hot[b * s.D + i];
}
__device__ __forceinline__ float gcond_input(const vihds_encoder_shape& s, const float* inputs, const float* dev1hot,
                                             int b, int i) {
  if (s.g_tr) {
    if (i < s.n_tr) return inputs[b * s.n_tr + i];
    i -= s.n_tr;
  }
  return dev1hot[b * s.D + i];
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// forward: one block per data row
// LDS: x [C_in*L] | conv weights [F*C_in*K] | conv bias [F] | conv out [F*Lc] | pooled [F*Lp] | hidden [H]
constexpr int ENC_T = 1024;  // threads per block: these kernels are pure latency, so every phase is spread as wide as
                             // its output count allows (760 conv outputs, 720 pooled values, 16 waves for the Linear)
// LIN_C: 64-wide column chunks of a Linear row a lane holds in registers (pooled vector <= 64 LIN_C: 12 covers the
// reference's 86-point grids at 720, 16 the relay plate's 99-point grid at 850).
template <int LIN_C>
__global__ void __launch_bounds__(ENC_T)
encoder_fwd_kernel(vihds_encoder_shape s, const float* __restrict__ delta_obs, const float* __restrict__ inputs,
                   const float* __restrict__ dev1hot, const float* __restrict__ conv_w,
                   const float* __restrict__ conv_b, const float* __restrict__ lin_w, const float* __restrict__ lin_b,
                   const float* __restrict__ local_w, const float* __restrict__ local_b,
                   const float* __restrict__ gcond_w, const float* __restrict__ global_free,
                   const float* __restrict__ const_values, float* __restrict__ q_all, float* __restrict__ pooled_out,
                   float* __restrict__ hidden_out) {
  extern __shared__ float lds[];
  const EncDims d = enc_dims(s);
  const int b = blockIdx.x, tid = threadIdx.x, B = s.B;
  float* x = lds;
  float* cw = x + s.C_in * s.L;
  float* cb = cw + s.F * s.C_in * s.K;  // conv bias (read per conv output: from LDS, not a global load behind the barrier)
  float* cv = cb + s.F;
  float* pl = cv + s.F * d.Lc;
  float* hid = pl + d.NPOOL;
  const int lane = tid & 63, wid = tid >> 6;
  constexpr int NW = ENC_T / 64;
  VIHDS_ENC_STOP(0)
  // Every global read the later phases need is issued NOW, so that the kernel pays one memory round trip instead of
  // one per phase: this wave's Linear rows (4 output units x up to 12 x 64 inputs = 48 registers per lane) and its
  // head rows.  Shapes beyond those bounds take the plain loops further down.
  constexpr int LIN_U = 4, HEAD_R = 4;
  const bool fast_lin = s.H <= LIN_U * NW && d.NPOOL <= LIN_C * 64;
  const int n_dot = 2 * (s.nl + s.ng);
  const bool fast_heads = d.NX <= 64 && d.NG <= 64 && n_dot <= HEAD_R * NW;
  // inputs -> LDS.  First pass of every staging loop through registers: all its loads are requested before the first LDS
  // store (a plain `lds[q] = global[q]` loop is a load -> wait -> store round trip of its own; three of them in a row
  // were 1.5 us of this kernel), then the rare further passes.
  {
    const int nx = s.C_in * s.L, ncw = s.F * s.C_in * s.K;
    const float rx = delta_obs[(size_t)b * nx + min(tid, nx - 1)];
    const float rcw = conv_w[min(tid, ncw - 1)];
    const float rcb = conv_b[min(tid, s.F - 1)];
    if (tid < nx) x[tid] = rx;
    if (tid < ncw) cw[tid] = rcw;
    if (tid < s.F) cb[tid] = rcb;
    for (int q = tid + ENC_T; q < nx; q += ENC_T) x[q] = delta_obs[(size_t)b * nx + q];
    for (int q = tid + ENC_T; q < ncw; q += ENC_T) cw[q] = conv_w[q];
    for (int q = tid + ENC_T; q < s.F; q += ENC_T) cb[q] = conv_b[q];
  }
  // rows of the table that are plain copies (global free scalars, constants, zeros): nothing to wait for, written now
  {
    const int n_rows_all = 2 * (s.nl + s.ng + s.ngl + s.nc), n_dot0 = 2 * (s.nl + s.ng);
    for (int r = n_dot0 + tid; r < n_rows_all; r += ENC_T) {
      float v;
      if (r < 2 * (s.nl + s.ng + s.ngl)) {
        v = global_free[r - n_dot0];
      } else {
        const int rr = r - 2 * (s.nl + s.ng + s.ngl);
        v = rr < s.nc ? const_values[rr] : 0.f;
      }
      q_all[(size_t)r * B + b] = v;
    }
  }
  // what the heads multiply besides the hidden units (treatments, device one-hot) and their biases: requested here, used
  // after the last barrier (as loads in the head phase they were one more memory round trip at the kernel's tail)
  float xl_pre = 0.f, xg_pre = 0.f, hb[HEAD_R] = {0.f, 0.f, 0.f, 0.f};
  if (fast_heads) {
    // (pointers selected, loads unconditional, values selected afterwards)
    const float* pl_ = lin_b;
    if (lane >= s.H && lane < d.NX) {
      int i = lane - s.H;
      if (s.l_tr && i < s.n_tr) pl_ = inputs + b * s.n_tr + i;
      else pl_ = dev1hot + b * s.D + (i - (s.l_tr ? s.n_tr : 0));
    }
    const float* pg_ = lin_b;
    if (lane < d.NG) {
      if (s.g_tr && lane < s.n_tr) pg_ = inputs + b * s.n_tr + lane;
      else pg_ = dev1hot + b * s.D + (lane - (s.g_tr ? s.n_tr : 0));
    }
    const float vl = *pl_, vg = *pg_;
    float hbv[HEAD_R];
#pragma unroll
    for (int rr = 0; rr < HEAD_R; ++rr) {
      const int r = wid + rr * NW;
      hbv[rr] = *((r < 2 * s.nl && local_b) ? local_b + r : lin_b);
    }
    xl_pre = (lane >= s.H && lane < d.NX) ? vl : 0.f;
    xg_pre = lane < d.NG ? vg : 0.f;
#pragma unroll
    for (int rr = 0; rr < HEAD_R; ++rr) hb[rr] = (wid + rr * NW < 2 * s.nl && local_b) ? hbv[rr] : 0.f;
  }
  // The block's big load -- this wavefront's Linear rows, 48 values per lane, 144 KB per block -- is requested only NOW,
  // behind the waits of the small loads above and ahead of barriers that do not wait for it (enc_sync_lds): it lands
  // while the convolution and the pooling run.  (Requested first, every barrier's vmcnt(0) made the conv wait for it.)
  float hw[HEAD_R];
  if (fast_heads) {
#pragma unroll
    for (int rr = 0; rr < HEAD_R; ++rr) {
      // (one unconditional load through a selected pointer each, all four requested before any is looked at: under nested
      // branches every one of them was a memory round trip of its own; rows / lanes that do not exist read lin_b[0])
      const int r = wid + rr * NW;
      const bool is_l = r < 2 * s.nl, is_g = !is_l && r < n_dot;
      const float* wp = lin_b;
      if (is_l && lane < d.NX) wp = local_w + (size_t)r * d.NX + lane;
      if (is_g && lane < d.NG) wp = gcond_w + (size_t)(r - 2 * s.nl) * d.NG + lane;
      hw[rr] = *wp;
    }
#pragma unroll
    for (int rr = 0; rr < HEAD_R; ++rr) {
      const int r = wid + rr * NW;
      const bool is_l = r < 2 * s.nl, is_g = !is_l && r < n_dot;
      if (!((is_l && lane < d.NX) || (is_g && lane < d.NG))) hw[rr] = 0.f;
    }
  }
  VIHDS_ENC_STOP(1)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  VIHDS_ENC_STOP(2)
  float lw[LIN_U][LIN_C], lb[LIN_U];
  auto load_unit = [&](int u) {  // this wavefront's Linear row u: 12 loads per lane + its bias
    // (unconditional, clamped addresses and NO select behind them: a column past the pooled vector meets pv = 0 below and
    // a row past H is never written out -- a load under a branch gets a wait at the join, a select behind every load
    // serialises the batch)
    const int j = min(wid + u * NW, s.H - 1);
    lb[u] = lin_b[j];
#pragma unroll
    for (int c = 0; c < LIN_C; ++c) lw[u][c] = lin_w[(size_t)j * d.NPOOL + min(lane + 64 * c, d.NPOOL - 1)];
  };
  enc_sync_lds();
  VIHDS_ENC_STOP(3)
  // Conv1d (cross-correlation, no padding): out[o][t] = bias[o] + sum_c sum_k w[o][c][k] x[c][t+k]
  // The block's big load (the Linear rows: 144 KB) is issued IN BETWEEN the convolution's input channels: issuing 52 loads
  // per lane back to back keeps a wavefront waiting on the load queue for ~2.5 us (stamps: tests/probe/enc_stamps.py)
  // while the convolution needs LDS and VALU only -- one row's 13 loads, one channel's taps, and so on.
  const bool interleave = fast_lin && s.F * d.Lc <= ENC_T && s.C_in <= LIN_U;
  if (interleave) {
    const int q = tid;
    const bool has = q < s.F * d.Lc;
    const int o = has ? q / d.Lc : 0, t = has ? q - o * d.Lc : 0;
    float acc0 = cb[o], acc1 = 0.f;
#pragma unroll
    for (int c = 0; c < LIN_U; ++c) {
      load_unit(c);
      asm volatile("" ::: "memory");
      if (c < s.C_in) {
        const float* wr = cw + (o * s.C_in + c) * s.K;
        const float* xr = x + c * s.L + t;
        int k = 0;
#pragma unroll 5
        for (; k + 1 < s.K; k += 2) {
          acc0 += wr[k] * xr[k];
          acc1 += wr[k + 1] * xr[k + 1];
        }
        if (k < s.K) acc0 += wr[k] * xr[k];
      }
      asm volatile("" ::: "memory");
    }
    if (has) cv[q] = acc0 + acc1;
  } else {
    if (fast_lin) {
#pragma unroll
      for (int u = 0; u < LIN_U; ++u) load_unit(u);
    }
    for (int q = tid; q < s.F * d.Lc; q += ENC_T) {
      const int o = q / d.Lc, t = q - o * d.Lc;
      // two accumulators and an unrolled tap loop: the LDS reads of several taps are in flight together (as one
      // dependent chain of C_in*K = 40 read-read-FMA steps this phase took 1.75 us)
      float acc0 = cb[o], acc1 = 0.f;
      for (int c = 0; c < s.C_in; ++c) {
        const float* wr = cw + (o * s.C_in + c) * s.K;
        const float* xr = x + c * s.L + t;
        int k = 0;
#pragma unroll 5
        for (; k + 1 < s.K; k += 2) {
          acc0 += wr[k] * xr[k];
          acc1 += wr[k + 1] * xr[k + 1];
        }
        if (k < s.K) acc0 += wr[k] * xr[k];
      }
      cv[q] = acc0 + acc1;
    }
  }
  enc_sync_lds();
  VIHDS_ENC_STOP(4)
  // AvgPool1d(pool, stride 1)
  const float inv_pool = 1.f / (float)s.pool;
  for (int q = tid; q < d.NPOOL; q += ENC_T) {
    const int o = q / d.Lp, t = q - o * d.Lp;
    float acc = 0.f;
    for (int k = 0; k < s.pool; ++k) acc += cv[o * d.Lc + t + k];
    const float v = acc * inv_pool;
    pl[q] = v;
    pooled_out[(size_t)b * d.NPOOL + q] = v;
  }
  enc_sync_lds();
  VIHDS_ENC_STOP(5)
  // Linear + tanh: a wave owns output units j = wid, wid+16, ...; lanes stride over the inputs
  if (fast_lin) {
    float acc[LIN_U] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < LIN_C; ++c) {
      const int k = lane + 64 * c;
      const float pv = k < d.NPOOL ? pl[k] : 0.f;
#pragma unroll
      for (int u = 0; u < LIN_U; ++u) acc[u] += lw[u][c] * pv;
    }
#pragma unroll
    for (int u = 0; u < LIN_U; ++u) {
      const int j = wid + u * NW;
      const float t = wave_total_e(acc[u]);
      if (lane == 0 && j < s.H) {
        const float h = tanhf(t + lb[u]);
        hid[j] = h;
        hidden_out[(size_t)b * s.H + j] = h;
      }
    }
  } else {
    for (int j0 = wid; j0 < s.H; j0 += 4 * NW) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int k = lane; k < d.NPOOL; k += 64) {
        const float pv = pl[k];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u * NW;
          if (j < s.H) acc[u] += lin_w[(size_t)j * d.NPOOL + k] * pv;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = j0 + u * NW;
        const float t = wave_total_e(acc[u]);
        if (lane == 0 && j < s.H) {
          const float h = tanhf(t + lin_b[j]);
          hid[j] = h;
          hidden_out[(size_t)b * s.H + j] = h;
        }
      }
    }
  }
  enc_sync_lds();
  VIHDS_ENC_STOP(6)
  // heads -> rows of the level-blocked table [local mu; local lp; gcond mu; gcond lp; global mu; global lp; const; 0]
  // rows with a dot product: one wave per row (lanes over the inputs); the rest are copies
  if (fast_heads) {
#pragma unroll
    for (int rr = 0; rr < HEAD_R; ++rr) {
      const int r = wid + rr * NW;
      if (r >= n_dot) break;  // (uniform per wave)
      float xin = 0.f;
      if (r < 2 * s.nl) { if (lane < d.NX) xin = lane < s.H ? hid[lane] : xl_pre; }
      else if (lane < d.NG) xin = xg_pre;
      const float acc = wave_total_e(hw[rr] * xin);
      if (lane == 0) q_all[(size_t)r * B + b] = acc + hb[rr];
    }
  } else {
    for (int r = wid; r < n_dot; r += NW) {
      float acc = 0.f;
      if (r < 2 * s.nl) {
        const float* wr = local_w + (size_t)r * d.NX;
        for (int i = lane; i < d.NX; i += 64) acc += wr[i] * local_input(s, hid, inputs, dev1hot, b, i);
      } else {
        const float* wr = gcond_w + (size_t)(r - 2 * s.nl) * d.NG;
        for (int i = lane; i < d.NG; i += 64) acc += wr[i] * gcond_input(s, inputs, dev1hot, b, i);
      }
      acc = wave_total_e(acc);
      if (lane == 0) q_all[(size_t)r * B + b] = acc + ((r < 2 * s.nl && local_b) ? local_b[r] : 0.f);
    }
  }
  VIHDS_ENC_STOP(7)
}

// ---------------------------------------------------------------------------------------------------------------
// backward, per-row chain: g_all[:, b] -> g_pre[b] (adjoint of the Linear's pre-activation) and g_conv[b] (adjoint of
// the conv output).  LDS: g_x [NX] | g_pre [H] | g_pooled [F*Lp]
__global__ void __launch_bounds__(ENC_T)
encoder_bwd_row_kernel(vihds_encoder_shape s, const float* __restrict__ g_all, const float* __restrict__ hidden,
                       const float* __restrict__ lin_w, const float* __restrict__ local_w,
                       float* __restrict__ g_pre_out, float* __restrict__ g_conv_out) {
  extern __shared__ float lds[];
  const EncDims d = enc_dims(s);
  const int b = blockIdx.x, tid = threadIdx.x, B = s.B;
  float* gpre = lds;
  float* gpl = gpre + s.H;
  // this thread's column of lin_w is requested first: it arrives while g_pre is being formed
  constexpr int HMAX = 64;
  const bool fast = s.H <= HMAX && d.NPOOL <= ENC_T;
  float wcol[HMAX];
  if (fast && tid < d.NPOOL) {
#pragma unroll
    for (int j = 0; j < HMAX; ++j) wcol[j] = j < s.H ? lin_w[(size_t)j * d.NPOOL + tid] : 0.f;
  }
  // hidden adjoint through the local heads, then tanh'
  for (int j = tid; j < s.H; j += ENC_T) {
    float acc = 0.f;
    // (sixteen head rows at a time, every load of the chunk requested before the first product: as a plain loop the 2 nl
    // rows were 2 nl dependent round trips, most of this launch; rows past the end re-read the last one at weight 0;
    // the products are added in row order as before)
    constexpr int RC = 16;
    const int nr = 2 * s.nl;
    for (int r0 = 0; r0 < nr; r0 += RC) {
      float w[RC], g[RC];
#pragma unroll
      for (int u = 0; u < RC; ++u) {
        const int r = min(r0 + u, nr - 1);
        w[u] = local_w[(size_t)r * d.NX + j];
        g[u] = g_all[(size_t)r * B + b];
      }
#pragma unroll
      for (int u = 0; u < RC; ++u)
        if (r0 + u < nr) acc += w[u] * g[u];
    }
    const float h = hidden[(size_t)b * s.H + j];
    const float g = acc * (1.f - h * h);
    gpre[j] = g;
    g_pre_out[(size_t)b * s.H + j] = g;
  }
  __syncthreads();
  // pooled adjoint: g_pooled[k] = sum_j lin_w[j][k] g_pre[j]   (threads stride over k: coalesced rows; the H loads
  // of a thread are independent)
  if (fast) {
    if (tid < d.NPOOL) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int j = 0; j < HMAX; j += 2) {
        a0 += wcol[j] * (j < s.H ? gpre[j] : 0.f);
        a1 += wcol[j + 1] * (j + 1 < s.H ? gpre[j + 1] : 0.f);
      }
      gpl[tid] = a0 + a1;
    }
  } else {
    for (int k = tid; k < d.NPOOL; k += ENC_T) {
      float a0 = 0.f, a1 = 0.f;
      int j = 0;
      for (; j + 1 < s.H; j += 2) {
        a0 += lin_w[(size_t)j * d.NPOOL + k] * gpre[j];
        a1 += lin_w[(size_t)(j + 1) * d.NPOOL + k] * gpre[j + 1];
      }
      if (j < s.H) a0 += lin_w[(size_t)j * d.NPOOL + k] * gpre[j];
      gpl[k] = a0 + a1;
    }
  }
  __syncthreads();
  // conv-output adjoint: every pooled window that contains t contributes 1/pool
  const float inv_pool = 1.f / (float)s.pool;
  for (int q = tid; q < s.F * d.Lc; q += ENC_T) {
    const int o = q / d.Lc, t = q - o * d.Lc;
    const int lo = max(0, t - s.pool + 1), hi = min(t, d.Lp - 1);
    float acc = 0.f;
    for (int tp = lo; tp <= hi; ++tp) acc += gpl[o * d.Lp + tp];
    g_conv_out[(size_t)b * s.F * d.Lc + q] = acc * inv_pool;
  }
}

// backward, parameter gradients: every output is a fixed-order sum over the data rows.  Tasks by block range (the
// kernel lasts as long as its slowest block, so each small task has blocks of its own):
//   lin_w  [H][F*Lp]    one thread per element, B-term dot over rows
//   conv_w [F][C_in][K] one block per element, threads over (row, t): 11 terms per thread, one batch of loads
//                       (one wave per element = 43 dependent rounds: 9.0 us for the kernel; two waves: 8.0; a block: 6.5)
//   conv_b [F]          one block per element
//   local_w, gcond_w    one thread per element
//   lin_b, local_b, global_free: one block together (B-term sums)
struct EncReduceTasks {
  int nb_lin, nb_conv, nb_convb, nb_localw, nb_gcondw;
};
__global__ void __launch_bounds__(256)
encoder_bwd_reduce_kernel(vihds_encoder_shape s, EncReduceTasks tk, const float* __restrict__ g_all,
                          const float* __restrict__ delta_obs, const float* __restrict__ inputs,
                          const float* __restrict__ dev1hot, const float* __restrict__ pooled,
                          const float* __restrict__ hidden, const float* __restrict__ g_pre,
                          const float* __restrict__ g_conv, float* __restrict__ g_conv_w,
                          float* __restrict__ g_conv_b, float* __restrict__ g_lin_w, float* __restrict__ g_lin_b,
                          float* __restrict__ g_local_w, float* __restrict__ g_local_b,
                          float* __restrict__ g_gcond_w, float* __restrict__ g_global_free) {
  const EncDims d = enc_dims(s);
  const int tid = threadIdx.x, B = s.B;
  const int lane = tid & 63, wid = tid >> 6;
  int blk = blockIdx.x;
  if (blk < tk.nb_lin) {
    const int e = blk * 256 + tid;
    if (e >= s.H * d.NPOOL) return;
    const int j = e / d.NPOOL, k = e - j * d.NPOOL;
    float acc = 0.f;  // (one accumulator, fixed order; unrolled so the 2 x 12 loads of an iteration are in flight together)
#pragma unroll 12
    for (int b = 0; b < B; ++b) acc += g_pre[(size_t)b * s.H + j] * pooled[(size_t)b * d.NPOOL + k];
    g_lin_w[e] = acc;
    return;
  }
  blk -= tk.nb_lin;
  if (blk < tk.nb_conv) {
    // one block per (o, c, k): 11 rounds of B*Lc = 2 772 terms per thread at the headline shape, one batch of loads
    __shared__ float partc[4];
    const int e = blk;
    const int o = e / (s.C_in * s.K), c = (e / s.K) % s.C_in, k = e % s.K;
    float acc = 0.f;
    const int n = B * d.Lc;
#pragma unroll 11
    for (int q = tid; q < n; q += 256) {
      const int b = q / d.Lc, t = q - b * d.Lc;
      acc += g_conv[((size_t)b * s.F + o) * d.Lc + t] * delta_obs[((size_t)b * s.C_in + c) * s.L + t + k];
    }
    acc = wave_sum_e(acc);
    if (lane == 0) partc[wid] = acc;
    __syncthreads();
    if (tid == 0) g_conv_w[e] = (partc[0] + partc[1]) + (partc[2] + partc[3]);
    return;
  }
  blk -= tk.nb_conv;
  if (blk < tk.nb_convb) {
    // one block per filter: B*Lc = 2 772 terms -> 11 per thread, all loads in flight at once.  (As one wave per
    // filter with a plain loop -- 43 dependent rounds -- these ten sums were the long pole of the whole kernel.)
    __shared__ float partb[4];
    const int o = blk;
    float acc = 0.f;
#pragma unroll 11
    for (int q = tid; q < B * d.Lc; q += 256) {
      const int b = q / d.Lc, t = q - b * d.Lc;
      acc += g_conv[((size_t)b * s.F + o) * d.Lc + t];
    }
    acc = wave_sum_e(acc);
    if (lane == 0) partb[wid] = acc;
    __syncthreads();
    if (tid == 0) g_conv_b[o] = (partb[0] + partb[1]) + (partb[2] + partb[3]);
    return;
  }
  blk -= tk.nb_convb;
  if (blk < tk.nb_localw) {
    const int e = blk * 256 + tid;
    if (e >= 2 * s.nl * d.NX) return;
    const int r = e / d.NX;
    int i = e - r * d.NX;
    // which per-row input this column multiplies: chosen once, so the B-term loop is branch-free and its loads batch
    const float* src;
    int stride;
    if (i < s.H) { src = hidden + i; stride = s.H; }
    else {
      i -= s.H;
      if (s.l_tr && i < s.n_tr) { src = inputs + i; stride = s.n_tr; }
      else { src = dev1hot + (i - (s.l_tr ? s.n_tr : 0)); stride = s.D; }
    }
    float acc = 0.f;
#pragma unroll 12
    for (int b = 0; b < B; ++b) acc += g_all[(size_t)r * B + b] * src[(size_t)b * stride];
    g_local_w[e] = acc;
    return;
  }
  blk -= tk.nb_localw;
  if (blk < tk.nb_gcondw) {
    const int e = blk * 256 + tid;
    if (e >= 2 * s.ng * d.NG) return;
    const int r = e / d.NG, i = e - r * d.NG;
    const float* src;
    int stride;
    if (s.g_tr && i < s.n_tr) { src = inputs + i; stride = s.n_tr; }
    else { src = dev1hot + (i - (s.g_tr ? s.n_tr : 0)); stride = s.D; }
    float acc = 0.f;
#pragma unroll 12
    for (int b = 0; b < B; ++b) acc += g_all[(size_t)(2 * s.nl + r) * B + b] * src[(size_t)b * stride];
    g_gcond_w[e] = acc;
    return;
  }
  // ---- bias / free-scalar sums, one block: every item is a B-term sum of a strided column
  const int n_lb = g_local_b ? 2 * s.nl : 0;
  for (int item = tid; item < s.H + n_lb + 2 * s.ngl; item += 256) {
    const float* src;
    int stride;
    float* dst;
    if (item < s.H) { src = g_pre + item; stride = s.H; dst = g_lin_b + item; }
    else if (item < s.H + n_lb) { const int r = item - s.H; src = g_all + (size_t)r * B; stride = 1; dst = g_local_b + r; }
    else {
      const int r = item - s.H - n_lb;
      src = g_all + (size_t)(2 * (s.nl + s.ng) + r) * B; stride = 1; dst = g_global_free + r;
    }
    float acc = 0.f;
#pragma unroll 12
    for (int b = 0; b < B; ++b) acc += src[(size_t)b * stride];
    *dst = acc;
  }
}

// ---- launchers -------------------------------------------------------------------------------------------------
size_t encoder_fwd_lds_bytes(const vihds_encoder_shape& s) {
  const EncDims d = enc_dims(s);
  return sizeof(float) * ((size_t)s.C_in * s.L + (size_t)s.F * s.C_in * s.K + s.F + (size_t)s.F * d.Lc + d.NPOOL + s.H);
}
size_t encoder_bwd_lds_bytes(const vihds_encoder_shape& s) {
  const EncDims d = enc_dims(s);
  return sizeof(float) * ((size_t)s.H + d.NPOOL);
}
void launch_encoder_fwd(const vihds_encoder_shape& s, const float* delta_obs, const float* inputs, const float* dev1hot,
                        const float* conv_w, const float* conv_b, const float* lin_w, const float* lin_b,
                        const float* local_w, const float* local_b, const float* gcond_w, const float* global_free,
                        const float* const_values, float* q_all, float* pooled, float* hidden, hipStream_t st) {
  if (enc_dims(s).NPOOL <= 12 * 64)
    hipLaunchKernelGGL(encoder_fwd_kernel<12>, dim3(s.B), dim3(ENC_T), encoder_fwd_lds_bytes(s), st, s, delta_obs, inputs,
                     dev1hot, conv_w, conv_b, lin_w, lin_b, local_w, local_b, gcond_w, global_free, const_values, q_all,
                     pooled, hidden);
  else
    hipLaunchKernelGGL(encoder_fwd_kernel<16>, dim3(s.B), dim3(ENC_T), encoder_fwd_lds_bytes(s), st, s, delta_obs, inputs,
                     dev1hot, conv_w, conv_b, lin_w, lin_b, local_w, local_b, gcond_w, global_free, const_values, q_all,
                     pooled, hidden);
}
void launch_encoder_bwd(const vihds_encoder_shape& s, const float* g_all, const float* delta_obs, const float* inputs,
                        const float* dev1hot, const float* lin_w, const float* local_w, const float* pooled,
                        const float* hidden, float* g_pre, float* g_conv, float* g_conv_w, float* g_conv_b,
                        float* g_lin_w, float* g_lin_b, float* g_local_w, float* g_local_b, float* g_gcond_w,
                        float* g_global_free, hipStream_t st) {
  const EncDims d = enc_dims(s);
  hipLaunchKernelGGL(encoder_bwd_row_kernel, dim3(s.B), dim3(ENC_T), encoder_bwd_lds_bytes(s), st, s, g_all, hidden,
                     lin_w, local_w, g_pre, g_conv);
  EncReduceTasks tk;
  tk.nb_lin = (s.H * d.NPOOL + 255) / 256;
  tk.nb_conv = s.F * s.C_in * s.K;
  tk.nb_convb = s.F;
  tk.nb_localw = (2 * s.nl * d.NX + 255) / 256;
  tk.nb_gcondw = (2 * s.ng * d.NG + 255) / 256;
  const int nblocks = tk.nb_lin + tk.nb_conv + tk.nb_convb + tk.nb_localw + tk.nb_gcondw + 1;
  hipLaunchKernelGGL(encoder_bwd_reduce_kernel, dim3(nblocks), dim3(256), 0, st, s, tk, g_all,
                     delta_obs, inputs, dev1hot, pooled, hidden, g_pre, g_conv, g_conv_w, g_conv_b, g_lin_w, g_lin_b,
                     g_local_w, g_local_b, g_gcond_w, g_global_free);
}

}  // namespace vihds
#ifdef VIHDS_TAIL_STAMPS
extern "C" int vihds_debug_enc_stamps(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(vihds::vihds_enc_stamp_buf), &buf, sizeof(buf));
}
#endif
