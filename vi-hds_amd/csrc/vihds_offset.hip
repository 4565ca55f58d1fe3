// dr_blackbox's condition_theta (reference models/dr_blackbox.py:86-96): y_i += offset_layer(dev_1hot)_i, a Linear(D, n_y)
// on the device one-hot added to n_y theta rows for every sample.  Forward: one launch writes the conditioned rows
// (the simulator reads those; theta's own y rows keep the sampled values log q / log p are taken of).  Backward: one
// launch routes the conditioned rows' gradient back to the y rows and forms the layer's weight and bias gradients
// (what autograd's Linear backward, a row sum and an add did in six launches).
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"

namespace vihds {

// grid (B, n): theta[dst+i][b][:] = theta[src+i][b][:] + W[i,:] . dev1hot[b,:] + bias[i]
__global__ void __launch_bounds__(256) offset_rows_fwd_kernel(int B, int S, int D, int src, int dst,
                                                              const float* __restrict__ W, const float* __restrict__ bias,
                                                              const float* __restrict__ dev1hot, float* __restrict__ theta) {
  const int b = blockIdx.x, i = blockIdx.y;
  float off = bias[i];
  for (int d = 0; d < D; ++d) off = fmaf(W[i * D + d], dev1hot[b * D + d], off);
  const float* in = theta + ((size_t)(src + i) * B + b) * S;
  float* out = theta + ((size_t)(dst + i) * B + b) * S;
  for (int s = threadIdx.x; s < S; s += blockDim.x) out[s] = in[s] + off;
}

// grid (n), 1024 threads: wavefront w sums the rows b = w, w + 16, ... of g_theta[dst+i] over the samples while adding
// them into g_theta[src+i] (accumulate = 0: assigning them -- the caller left those rows unwritten); then g_W[i][d] = sum_b rs[b] dev1hot[b][d], g_bias[i] = sum_b rs[b] (fixed order)
constexpr int OFFSET_BWD_THREADS = 1024;
__global__ void __launch_bounds__(OFFSET_BWD_THREADS)
offset_rows_bwd_kernel(int B, int S, int D, int n, int src, int dst, int accumulate,
                       const float* __restrict__ dev1hot, float* __restrict__ g_theta, float* __restrict__ g_wb) {
  extern __shared__ float rs[];  // [B]
  const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  // Up to 3 rows per wavefront x 4 chunks of 64 samples (B <= 48, S <= 256: the ICML plate) -- every load of the wavefront
  // is requested before the first one is used (as nested loops these were up to twelve memory round trips in a row);
  // the sums are taken in the loops' order: per row, a lane over its samples in increasing s, then the shuffle tree.
  constexpr int RB = 3, SC = 4;
  if (B <= RB * (OFFSET_BWD_THREADS / 64) && S <= SC * 64 && n_waves == OFFSET_BWD_THREADS / 64) {
    float g[RB][SC], old[RB][SC];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int b = min(wave + rb * n_waves, B - 1);
#pragma unroll
      for (int sc = 0; sc < SC; ++sc) {
        const int sidx = min(lane + 64 * sc, S - 1);
        g[rb][sc] = g_theta[((size_t)(dst + i) * B + b) * S + sidx];
        old[rb][sc] = accumulate ? g_theta[((size_t)(src + i) * B + b) * S + sidx] : 0.f;
      }
    }
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
      const int b = wave + rb * n_waves;
      float acc = 0.f;
#pragma unroll
      for (int sc = 0; sc < SC; ++sc) {
        const int sidx = lane + 64 * sc;
        if (b < B && sidx < S) {
          acc += g[rb][sc];
          g_theta[((size_t)(src + i) * B + b) * S + sidx] = old[rb][sc] + g[rb][sc];
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
      if (lane == 0 && b < B) rs[b] = acc;
    }
  } else
  for (int b = wave; b < B; b += n_waves) {
    const float* gd = g_theta + ((size_t)(dst + i) * B + b) * S;
    float* gs = g_theta + ((size_t)(src + i) * B + b) * S;
    float acc = 0.f;
    for (int s = lane; s < S; s += 64) {
      const float g = gd[s];
      acc += g;
      gs[s] = accumulate ? gs[s] + g : g;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) rs[b] = acc;
  }
  __syncthreads();
  if (g_wb == nullptr) return;
  if ((int)threadIdx.x < D) {
    float w = 0.f;
    for (int b = 0; b < B; ++b) w = fmaf(rs[b], dev1hot[b * D + threadIdx.x], w);
    g_wb[i * D + threadIdx.x] = w;
  } else if ((int)threadIdx.x == D) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += rs[b];
    g_wb[n * D + i] = s;
  }
}

void launch_offset_rows_fwd(int B, int S, int D, int n, int src, int dst, const float* W, const float* bias,
                            const float* dev1hot, float* theta, hipStream_t st) {
  hipLaunchKernelGGL(offset_rows_fwd_kernel, dim3(B, n), dim3(256), 0, st, B, S, D, src, dst, W, bias, dev1hot, theta);
}
void launch_offset_rows_bwd(int B, int S, int D, int n, int src, int dst, int accumulate, const float* dev1hot,
                            float* g_theta, float* g_wb, hipStream_t st) {
  hipLaunchKernelGGL(offset_rows_bwd_kernel, dim3(n), dim3(OFFSET_BWD_THREADS), (size_t)B * sizeof(float), st, B, S, D, n,
                     src, dst, accumulate, dev1hot, g_theta, g_wb);
}
// Device-resident batching (reference training.py:108-113: DataLoader(shuffle) + collate_merged :55-68 stack the rows of a
// batch on the host every step): rows idx[0..B) of the training set, which is resident in HBM as a whole, -> the step's
// batch buffers, and the encoder's input-only preprocessing delta_obs[b][c][t] = obs[b][c][t+1] - obs[b][c][t]
// (encoders.py:385) while the row is at hand.  One block per batch row.
__global__ void __launch_bounds__(256) gather_batch_kernel(int n_src, int CT, int T, int n_tr, int D,
                                                           const long long* __restrict__ idx,
                                                           const float* __restrict__ obs_src,
                                                           const float* __restrict__ inputs_src,
                                                           const float* __restrict__ dev1hot_src, float* __restrict__ obs,
                                                           float* __restrict__ inputs, float* __restrict__ dev1hot,
                                                           float* __restrict__ delta_obs) {
  const int b = blockIdx.x;
  long long r = idx[b];
  r = r < 0 ? 0 : (r >= n_src ? n_src - 1 : r);  // (an index outside the set is clamped, never dereferenced)
  const float* o = obs_src + (size_t)r * CT;
  for (int q = threadIdx.x; q < CT; q += 256) {
    const float v = o[q];
    obs[(size_t)b * CT + q] = v;
    const int c = q / T, t = q - c * T;
    if (delta_obs && t + 1 < T) delta_obs[((size_t)b * (CT / T) + c) * (T - 1) + t] = o[q + 1] - v;
  }
  for (int q = threadIdx.x; q < n_tr; q += 256) inputs[(size_t)b * n_tr + q] = inputs_src[(size_t)r * n_tr + q];
  for (int q = threadIdx.x; q < D; q += 256) dev1hot[(size_t)b * D + q] = dev1hot_src[(size_t)r * D + q];
}
void launch_gather_batch(int B, int n_src, int C4, int T, int n_tr, int D, const long long* idx, const float* obs_src,
                         const float* inputs_src, const float* dev1hot_src, float* obs, float* inputs, float* dev1hot,
                         float* delta_obs, hipStream_t st) {
  hipLaunchKernelGGL(gather_batch_kernel, dim3(B), dim3(256), 0, st, n_src, C4 * T, T, n_tr, D, idx, obs_src, inputs_src,
                     dev1hot_src, obs, inputs, dev1hot, delta_obs);
}

}  // namespace vihds
