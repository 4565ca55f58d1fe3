// The importance weights of the ELBO formed INSIDE an ODE adjoint kernel (vihds_ode_bwd_elbo, ABI 13).
//
// In a training step the adjoint's only upstream gradient is d loss / d logp[j][b][s] = -(1/B) softmax_s(log_w[b][.]) for all
// four signals j (reference training.py:135-149: log_w = sum_j log p(x_j | theta) + log p(theta) - log q(theta), loss =
// -mean_b(logsumexp_s log_w - log S)), so instead of a launch of its own between the forward and the adjoint (vihds_iwae_loss_fwd:
// 5.7 us + a kernel boundary at B=36, S=200) every WAVEFRONT of the adjoint forms the row-wise logsumexp of the data rows its
// trajectories belong to -- S x 6 loads and two DPP reductions per row, no block barrier, so it fits kernels whose wavefronts
// part ways early (dr_blackbox's cooperating wavefronts) -- and each lane its own trajectory's weight.  Same summation order per
// sample as the tail's rows kernel (vihds_step_tail.hip), which recomputes the weights for the theta adjoint and writes
// log_w / lse / -ELBO.
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_args.hpp"
#include "vihds_wave.hpp"

namespace vihds {

__device__ __forceinline__ float iw_log_w(const OdeArgs& a, size_t i) {
  const size_t n = a.n;
  float v = ((a.iw_logp[i] + a.iw_logp[n + i]) + a.iw_logp[2 * n + i]) + a.iw_logp[3 * n + i];
  if (a.iw_log_p) v += a.iw_log_p[i];
  if (a.iw_log_q) v -= a.iw_log_q[i];
  return v;
}

// logsumexp over the S samples of data row `row`, by one wavefront (all 64 lanes call it; the result is uniform)
__device__ __forceinline__ float iw_wave_row_lse(const OdeArgs& a, int row) {
  constexpr int RC = 4;  // rounds of 64 samples held in registers (S <= 256: one pass over memory)
  const int lane = threadIdx.x & 63, S = a.S;
  const size_t base = (size_t)row * S;
  float v[RC], m = -INFINITY;
#pragma unroll
  for (int c = 0; c < RC; ++c) {
    const int s = lane + 64 * c;
    const float x = iw_log_w(a, base + min(s, S - 1));  // (unconditional loads, value selected)
    v[c] = s < S ? x : -INFINITY;
    m = fmaxf(m, v[c]);
  }
  for (int s = lane + 64 * RC; s < S; s += 64) m = fmaxf(m, iw_log_w(a, base + s));
  m = wave_max_total(m);
  float se = 0.f;
  if (m > -INFINITY) {
#pragma unroll
    for (int c = 0; c < RC; ++c) se += __expf(v[c] - m);  // (exp(-inf) = 0 for the padding)
    for (int s = lane + 64 * RC; s < S; s += 64) se += __expf(iw_log_w(a, base + s) - m);
  }
  se = wave_total(se);
  return m + __logf(se);
}

// d loss / d log_w of this lane's trajectory i (data row b): every lane of the wavefront calls it (tail lanes shadow a live
// trajectory, as everywhere in the adjoint kernels)
__device__ __forceinline__ float iw_wave_weight(const OdeArgs& a, int i, int b) {
  const int b_lo = -(int)wave_max_total((float)-b), b_hi = (int)wave_max_total((float)b);
  float lse = 0.f;
  for (int r = b_lo; r <= b_hi; ++r) {
    const float l = iw_wave_row_lse(a, r);
    lse = b == r ? l : lse;
  }
  return -(1.f / (float)a.B) * __expf(iw_log_w(a, (size_t)i) - lse);
}

// the log-likelihood gradient of (trajectory i, signal j): the importance weight formed here, or the caller's array
__device__ __forceinline__ float ode_logp_grad(const OdeArgs& a, float w_iw, int i, int j) {
  if (a.iw_logp) return w_iw;
  return a.g_logp ? a.g_logp[(a.logp_grad_broadcast ? 0 : (size_t)j * a.n) + i] : 0.f;
}

}  // namespace vihds
