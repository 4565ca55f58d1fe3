// theta-side and ELBO-side kernels:
//   theta_fwd/bwd : ChainedDistribution.sample + p.clip + q.log_prob + p.log_prob for all P parameters at once
//                   (reference vihds/distributions.py:64-85,119-142,327-381 and vihds/vae.py:31-34)
//   iwae_fwd/bwd  : importance-weight row reductions of Training.cost (vihds/training.py:135-149)
//   iw_summaries  : Results.init importance-weighted summaries (vihds/utils.py:79-99) without host copies
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"
#include "vihds_rng.hpp"
#include "vihds_wave.hpp"

namespace vihds {

constexpr float LOG2PI_F = 1.8378770664093453f;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  return v;
}
// result valid in every thread
template <int BLOCK>
__device__ __forceinline__ float block_sum(float v, float* sm) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) r += sm[w];
  __syncthreads();
  return r;
}
template <int BLOCK>
__device__ __forceinline__ float block_max(float v, float* sm) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm[wid] = v;
  __syncthreads();
  float r = sm[0];
#pragma unroll
  for (int w = 1; w < BLOCK / 64; ++w) r = fmaxf(r, sm[w]);
  __syncthreads();
  return r;
}

// Normal log-density as the reference writes it (distributions.py:338-345): note -log(2*pi), not -0.5*log(2*pi)
__device__ __forceinline__ float normal_lp(float mu, float prec, float x) {
  const float d = mu - x;
  return -LOG2PI_F + 0.5f * logf(prec + 1e-12f) - 0.5f * prec * d * d;
}

// four lanes per (b, s): lane q of the quad handles parameters p = q, q+4, ...; the quad reads 16 contiguous bytes of
// u[b][s][:] per iteration, every theta row store covers runs of 16 consecutive trajectories, and log q / log p are
// quad-reduced with two DPP adds (no LDS, no atomics).
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
  return v;
}
// rng: {seed lo, seed hi, step, ticket} or NULL.  With rng the kernel draws u itself and writes it to `u` (the adjoint
// and the host read it from there); the last block to finish advances the step, so replays of a captured graph get
// fresh draws without any host-side bookkeeping.
__global__ void theta_fwd_kernel(int P, int B, int S, const int* __restrict__ kind, const float* __restrict__ q_mu,
                                 const float* __restrict__ q_prec, const int* __restrict__ q_rows, int prec_is_log,
                                 const float* __restrict__ p_mu, const float* __restrict__ p_prec,
                                 const float* __restrict__ clip_lo, const float* __restrict__ clip_hi,
                                 float* __restrict__ u, unsigned int* rng, int S_total, int s_off,
                                 float* __restrict__ theta, float* __restrict__ log_q, float* __restrict__ log_p) {
  const int n = B * S;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = t >> 2, q = t & 3;
  const bool live = i0 < n;
  const int i = live ? i0 : n - 1;
  const int b = i / S;
  unsigned int k0 = 0, k1 = 0, step = 0, gidx = 0;
  if (rng) {
    k0 = rng[0]; k1 = rng[1]; step = rng[2];
    gidx = (unsigned int)(b * S_total + s_off + (i - b * S));
  }
  float lq = 0.f, lp = 0.f;
  // lane q of the quad owns parameter blocks kb = q, q+4, ... (4 consecutive parameters each): one Philox call
  // yields exactly the block's four normals, and a lane reads / writes 16 contiguous bytes of u[b][s][:]
  for (int kb = q; 4 * kb < P; kb += 4) {
    float z4[4];
    if (rng) philox_normal4(gidx, (unsigned int)kb, step, 0u, k0, k1, z4);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = 4 * kb + jj;
      if (p >= P) break;
      const int kd = kind[p];
      float uu;
      if (rng) {
        uu = z4[jj];
        if (live) u[(size_t)i * P + p] = uu;
      } else {
        uu = u[(size_t)i * P + p];
      }
      const int rm = q_rows ? q_rows[p] : p, rp = q_rows ? q_rows[P + p] : p;
      const float mu = q_mu[rm * B + b];
      float x;
      if (kd == KIND_CONSTANT) {
        x = 0.f * uu + mu;  // zeros_like(u) + value (distributions.py:241-242)
      } else {
        const float pr = q_prec[rp * B + b];
        const float prec = prec_is_log ? expf(pr) : pr;
        const float sigma = 1.f / sqrtf(prec);
        float z = mu + sigma * uu;
        x = (kd == KIND_LOGNORMAL) ? expf(z) : z;
        const float lo = clip_lo[p], hi = clip_hi[p];
        x = x < lo ? lo : (x > hi ? hi : x);
        const float v = (kd == KIND_LOGNORMAL) ? logf(x + 1e-12f) : x;
        const float jac = (kd == KIND_LOGNORMAL) ? v : 0.f;
        lq += normal_lp(mu, prec, v) - jac;
        lp += normal_lp(p_mu[p], p_prec[p], v) - jac;
      }
      if (live) theta[(size_t)p * n + i] = x;
    }
  }
  lq = quad_sum(lq);
  lp = quad_sum(lp);
  if (live && q == 0) {
    if (log_q) log_q[i] = lq;
    if (log_p) log_p[i] = lp;
  }
  if (rng) {  // every block has read rng[2] by the time the last one gets here
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int ticket = atomicAdd(&rng[3], 1u);
      if (ticket == gridDim.x - 1) {
        rng[2] = step + 1u;
        rng[3] = 0u;
      }
    }
  }
}

// The same forward with everything that depends only on (data row, parameter) computed once per block and kept in
// LDS: kind, mu, sigma = 1/sqrt(prec), prec, the two log-density constants -log(2 pi) + 0.5 log(prec + 1e-12), the clip
// bounds and the prior.  A lane's per-parameter work is then u -> z -> exp -> clip -> log -> two FMAs, with no global
// loads besides u; before, each of a lane's ~9-12 parameters cost a round of dependent table loads plus a log, a
// sqrt and an exp of the precision.  256 threads = 64 trajectories per block, spanning nb <= 63/S + 2 data rows.
constexpr int THETA_LDS_FIELDS = 10;
__global__ void __launch_bounds__(256)
theta_fwd_lds_kernel(int P, int B, int S, int nb_max, const int* __restrict__ kind, const float* __restrict__ q_mu,
                     const float* __restrict__ q_prec, const int* __restrict__ q_rows, int prec_is_log,
                     const float* __restrict__ p_mu, const float* __restrict__ p_prec,
                     const float* __restrict__ clip_lo, const float* __restrict__ clip_hi, float* __restrict__ u,
                     unsigned int* rng, int advance, int S_total, int s_off, float* __restrict__ theta,
                     float* __restrict__ log_q, float* __restrict__ log_p) {
  extern __shared__ float tab[];  // [field][nb_max * P]
  const int n = B * S;
  const int first = blockIdx.x * 64, last = min(first + 64, n) - 1;
  const int b0 = first / S, nb = last / S - b0 + 1;
  const int stride = nb_max * P;
  float* t_kind = tab;                 // (as float: 0, 1, 2)
  float* t_mu = tab + stride;
  float* t_sigma = tab + 2 * stride;
  float* t_prec = tab + 3 * stride;
  float* t_cq = tab + 4 * stride;      // -log(2 pi) + 0.5 log(prec + 1e-12)
  float* t_lo = tab + 5 * stride;
  float* t_hi = tab + 6 * stride;
  float* t_pmu = tab + 7 * stride;
  float* t_cp = tab + 8 * stride;      // prior: the same constant
  float* t_pprec = tab + 9 * stride;
  for (int e = threadIdx.x; e < nb * P; e += 256) {
    const int bb = e / P, p = e - bb * P, b = b0 + bb;
    const int kd = kind[p];
    const int rm = q_rows ? q_rows[p] : p, rp = q_rows ? q_rows[P + p] : p;
    const float pr = q_prec[rp * B + b];
    const float prec = (kd == KIND_CONSTANT) ? 1.f : (prec_is_log ? expf(pr) : pr);
    t_kind[e] = (float)kd;
    t_mu[e] = q_mu[rm * B + b];
    t_sigma[e] = 1.f / sqrtf(prec);
    t_prec[e] = prec;
    t_cq[e] = -LOG2PI_F + 0.5f * logf(prec + 1e-12f);
    t_lo[e] = clip_lo[p];
    t_hi[e] = clip_hi[p];
    t_pmu[e] = p_mu[p];
    t_cp[e] = -LOG2PI_F + 0.5f * logf(p_prec[p] + 1e-12f);
    t_pprec[e] = p_prec[p];
  }
  // generator state: read before the barrier, so that every thread of the block holds the step counter when the block
  // takes its ticket right after it; the ticket's round trip then hides behind the sampling work and the holder of the
  // last ticket advances the counter at its end (a barrier + atomic on every block's tail cost microseconds)
  unsigned int k0 = 0, k1 = 0, step = 0;
  if (rng) { k0 = rng[0]; k1 = rng[1]; step = rng[2]; }
  __syncthreads();
  unsigned int ticket = 0u;
  // (advance == 0: a launch of rng_advance_kernel behind this one moves the step -- thousands of blocks taking tickets from
  // ONE address queue up behind each other: at 3 656 blocks the returning atomics were most of this kernel's 72 us)
  if (rng && advance && threadIdx.x == 0) ticket = atomicAdd(&rng[3], 1u);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i0 = t >> 2, q = t & 3;
  const bool live = i0 < n;
  const int i = live ? i0 : n - 1;
  const int b = i / S;
  const int row = (b - b0) * P;
  unsigned int gidx = 0;
  if (rng) gidx = (unsigned int)(b * S_total + s_off + (i - b * S));
  float lq = 0.f, lp = 0.f;
  for (int kb = q; 4 * kb < P; kb += 4) {
    float z4[4];
    if (rng) philox_normal4(gidx, (unsigned int)kb, step, 0u, k0, k1, z4);
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = 4 * kb + jj;
      if (p >= P) break;
      float uu;
      if (rng) {
        uu = z4[jj];
        if (live) u[(size_t)i * P + p] = uu;
      } else {
        uu = u[(size_t)i * P + p];
      }
      // straight-line: the lanes of a trajectory hold parameters of different kinds, a branch per kind runs every side
      const int e = row + p;
      const float kdf = t_kind[e], mu = t_mu[e];
      const bool cst = kdf == (float)KIND_CONSTANT, ln = kdf == (float)KIND_LOGNORMAL;
      const float zz = mu + t_sigma[e] * uu;
      float x = ln ? expf(zz) : zz;
      const float lo = t_lo[e], hi = t_hi[e];
      x = x < lo ? lo : (x > hi ? hi : x);
      const float v = ln ? logf(x + 1e-12f) : x;
      const float jac = ln ? v : 0.f;
      const float dq = mu - v, dp = t_pmu[e] - v;
      const float tq = t_cq[e] - 0.5f * t_prec[e] * dq * dq - jac;
      const float tp = t_cp[e] - 0.5f * t_pprec[e] * dp * dp - jac;
      lq += cst ? 0.f : tq;
      lp += cst ? 0.f : tp;
      x = cst ? 0.f * uu + mu : x;
      if (live) theta[(size_t)p * n + i] = x;
    }
  }
  lq = quad_sum(lq);
  lp = quad_sum(lp);
  if (live && q == 0) {
    if (log_q) log_q[i] = lq;
    if (log_p) log_p[i] = lp;
  }
  if (rng && advance && threadIdx.x == 0 && ticket == gridDim.x - 1) {
    rng[2] = step + 1u;
    rng[3] = 0u;
  }
}
// the step of a generator state {seed lo, seed hi, step, ticket}, moved by a launch of its own behind a grid too large
// to take tickets
__global__ void rng_advance_kernel(unsigned int* rng);
void launch_rng_advance(unsigned int* rng, hipStream_t st) { hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, st, rng); }
__global__ void rng_advance_kernel(unsigned int* rng) {
  rng[2] = rng[2] + 1u;
  rng[3] = 0u;
}
// grids up to this many blocks take tickets (one returning atomic per block on one address costs ~18 ns each once they
// queue: 915 blocks of device_condition_kernel = 16 us, 3 656 of theta_fwd_lds_kernel = 72 us, both the whole launch);
// beyond, the one-thread launch above (~2 us) moves the step
constexpr int RNG_TICKET_BLOCKS = 256;

// one block per (data row b, chunk of THETA_BWD_PCHUNK parameters), ONE WAVE PER PARAMETER: the chunk's parameters run
// side by side instead of one after the other (each used to cost a round of loads plus two block reductions with
// barriers); the S per-sample contributions are reduced inside the wave in a fixed order: deterministic gradients.
// prec_is_log: the q precision table holds log-precisions and g_q_prec receives d/d log_prec = prec * d/d prec.
constexpr int THETA_BWD_PCHUNK = 4;
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
theta_bwd_kernel(int P, int B, int S, const int* __restrict__ kind, const float* __restrict__ q_mu,
                 const float* __restrict__ q_prec, const int* __restrict__ q_rows, int prec_is_log,
                 const float* __restrict__ p_mu, const float* __restrict__ p_prec,
                 const float* __restrict__ clip_lo, const float* __restrict__ clip_hi, const float* __restrict__ u,
                 const float* __restrict__ g_theta, const float* __restrict__ g_theta_scale,
                 const float* __restrict__ g_log_q, const float* __restrict__ g_log_p, float* __restrict__ g_q_mu,
                 float* __restrict__ g_q_prec, vihds_iwae_job iw) {
  static_assert(BLOCK == 64 * THETA_BWD_PCHUNK, "one wave per parameter of the chunk");
  extern __shared__ float wsm[];  // [S] with an IWAE job: d loss / d log_w of this row
  __shared__ float sm[BLOCK / 64];
  __shared__ int is_last;
  const int n = B * S;
  const int b = blockIdx.x;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int p = blockIdx.y * THETA_BWD_PCHUNK + wid;
  const bool iwae = iw.logp != nullptr;
  // this wavefront's parameter: its constants and the first four rounds of its per-sample inputs are requested before
  // the row's importance weights are formed (with an IWAE job that prologue is three block reductions long)
  const bool has_p = p < P;
  const int kd = has_p ? kind[p] : KIND_CONSTANT;
  const int rm = has_p ? (q_rows ? q_rows[p] : p) : 0, rp = has_p ? (q_rows ? q_rows[P + p] : p) : 0;
  const bool live_p = has_p && kd != KIND_CONSTANT;
  float uu_pre[4] = {0.f, 0.f, 0.f, 0.f}, gx_pre[4] = {0.f, 0.f, 0.f, 0.f};
  float mu = 0.f, prec_raw = 0.f, pm = 0.f, pp = 1.f, lo = 0.f, hi = 0.f;
  if (live_p) {
    mu = q_mu[rm * B + b];
    prec_raw = q_prec[rp * B + b];
    pm = p_mu[p]; pp = p_prec[p]; lo = clip_lo[p]; hi = clip_hi[p];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int sidx = lane + 64 * c;
      if (sidx < S) {
        const int i = b * S + sidx;
        uu_pre[c] = u[(size_t)i * P + p];
        gx_pre[c] = g_theta ? g_theta[(size_t)p * n + i] : 0.f;
      }
    }
  }
  if (iwae) {
    // the arithmetic of iwae_loss_rows_kernel, once per block (every chunk of the row repeats it: S terms)
    float m = -INFINITY;
    for (int s = threadIdx.x; s < S; s += BLOCK) {
      const int i = b * S + s;
      float v = ((iw.logp[i] + iw.logp[n + i]) + iw.logp[2 * n + i]) + iw.logp[3 * n + i];
      v = v + (iw.log_p ? iw.log_p[i] : 0.f) - (iw.log_q ? iw.log_q[i] : 0.f);
      wsm[s] = v;
      if (blockIdx.y == 0) iw.log_w[i] = v;
      m = fmaxf(m, v);
    }
    m = block_max<BLOCK>(m, sm);
    float se = 0.f;
    for (int s = threadIdx.x; s < S; s += BLOCK) se += expf(wsm[s] - m);
    se = block_sum<BLOCK>(se, sm);
    const float l = m + logf(se);
    for (int s = threadIdx.x; s < S; s += BLOCK) wsm[s] = -(1.f / (float)B) * expf(wsm[s] - l);
    if (blockIdx.y == 0) {
      if (threadIdx.x == 0) {
        iw.lse[b] = l;
        __threadfence();
        is_last = atomicAdd(iw.ticket, 1u) == gridDim.x - 1;
      }
      __syncthreads();
      if (is_last) {  // every row's lse is visible: -ELBO, rows added in a fixed order
        __threadfence();
        const float log_n = logf((float)iw.n_iwae_total);
        float acc = 0.f;
        for (int r = threadIdx.x; r < B; r += BLOCK)
          acc += __hip_atomic_load(&iw.lse[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - log_n;
        acc = block_sum<BLOCK>(acc, sm);
        if (threadIdx.x == 0) {
          iw.loss[0] = -acc / (float)B;
          *iw.ticket = 0u;
        }
      }
    }
    __syncthreads();
  }
  if (!has_p) return;
  if (kd == KIND_CONSTANT) {
    // constants carry no trainable distribution parameters in the reference (encoders.py:242-253)
    if (lane == 0) { g_q_mu[rm * B + b] = 0.f; g_q_prec[rp * B + b] = 0.f; }
    return;
  }
  const float prec = prec_is_log ? expf(prec_raw) : prec_raw;
  const float sigma = 1.f / sqrtf(prec);
  float am = 0.f, ap = 0.f;
  auto body = [&](int sidx, float uu, float gx) {
    const int i = b * S + sidx;
    const float z = mu + sigma * uu;
    const float xr = (kd == KIND_LOGNORMAL) ? expf(z) : z;
    const float x = xr < lo ? lo : (xr > hi ? hi : xr);
    const float pass = (xr >= lo && xr <= hi) ? 1.f : 0.f;
    const float gw = iwae ? wsm[sidx] : 0.f;  // d loss / d log_w
    const float glq = iwae ? (iw.log_q ? -gw : 0.f) : (g_log_q ? g_log_q[i] : 0.f);
    const float glp = iwae ? (iw.log_p ? gw : 0.f) : (g_log_p ? g_log_p[i] : 0.f);
    if (iwae) gx *= gw;
    else if (g_theta_scale) gx *= g_theta_scale[i];
    float v, dv_dx;
    if (kd == KIND_LOGNORMAL) { v = logf(x + 1e-12f); dv_dx = 1.f / (x + 1e-12f); }
    else { v = x; dv_dx = 1.f; }
    const float jac = (kd == KIND_LOGNORMAL) ? 1.f : 0.f;
    // d lq/dv = prec*(mu - v) - jac ; d lp/dv = pp*(pm - v) - jac
    const float gv = glq * (prec * (mu - v) - jac) + glp * (pp * (pm - v) - jac);
    gx += gv * dv_dx;
    // back through clip and (for LogNormal) exp
    float gz = gx * pass;
    if (kd == KIND_LOGNORMAL) gz *= xr;
    // z = mu + u / sqrt(prec)
    am += gz;
    ap += gz * uu * (-0.5f) * sigma / prec;
    // explicit dependence of log q on (mu, prec)
    const float d = mu - v;
    am += glq * (-prec * d);
    ap += glq * (0.5f / (prec + 1e-12f) - 0.5f * d * d);
  };
#pragma unroll
  for (int c = 0; c < 4; ++c)
    if (lane + 64 * c < S) body(lane + 64 * c, uu_pre[c], gx_pre[c]);
#pragma unroll 4
  for (int sidx = lane + 256; sidx < S; sidx += 64) {
    const int i = b * S + sidx;
    body(sidx, u[(size_t)i * P + p], g_theta ? g_theta[(size_t)p * n + i] : 0.f);
  }
  am = wave_sum(am);
  ap = wave_sum(ap);
  if (lane == 0) { g_q_mu[rm * B + b] = am; g_q_prec[rp * B + b] = prec_is_log ? ap * prec : ap; }
}

// log_w = sum_j logp[j] + log_p - log_q ; per-row max and sum-exp.  One block per row.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
iwae_fwd_kernel(int B, int S, const float* __restrict__ logp, const float* __restrict__ log_p,
                const float* __restrict__ log_q, float* __restrict__ log_w, float* __restrict__ row_max,
                float* __restrict__ row_sumexp) {
  __shared__ float sm[BLOCK / 64];
  const int n = B * S, b = blockIdx.x;
  float m = -INFINITY;
  for (int s = threadIdx.x; s < S; s += BLOCK) {
    const int i = b * S + s;
    float lw = ((logp[i] + logp[n + i]) + logp[2 * n + i]) + logp[3 * n + i];
    lw = lw + (log_p ? log_p[i] : 0.f) - (log_q ? log_q[i] : 0.f);
    log_w[i] = lw;
    m = fmaxf(m, lw);
  }
  m = block_max<BLOCK>(m, sm);
  float se = 0.f;
  for (int s = threadIdx.x; s < S; s += BLOCK) se += expf(log_w[b * S + s] - m);
  se = block_sum<BLOCK>(se, sm);
  if (threadIdx.x == 0) { row_max[b] = m; row_sumexp[b] = se; }
}

__global__ void iwae_bwd_kernel(int B, int S, const float* __restrict__ log_w, const float* __restrict__ lse,
                                const float* __restrict__ g_lse, float* __restrict__ g_logw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  const int b = i / S;
  g_logw[i] = g_lse[b] * expf(log_w[i] - lse[b]);
}

// lse[b] = max + log(sumexp); loss = -mean_b(lse - log n_iwae)  (vihds/training.py:144-149); one block, fixed order
__global__ void __launch_bounds__(256)
iwae_finish_kernel(int B, float log_n, const float* __restrict__ row_max, const float* __restrict__ row_sumexp,
                   float* __restrict__ lse, float* __restrict__ loss) {
  __shared__ float sm[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float l = row_max[b] + logf(row_sumexp[b]);
    lse[b] = l;
    acc += l - log_n;
  }
  acc = block_sum<256>(acc, sm);
  if (threadIdx.x == 0) loss[0] = -acc / (float)B;
}


// The whole single-process IWAE loss in ONE launch for small batches (headline shape: 36 rows x 200 samples): one
// 1024-thread block, one wave per row (rows w, w+16, ...), wave-level max / sum-exp, then the mean over rows in a
// fixed order.  Replaces iwae_fwd_kernel + iwae_finish_kernel (each launch costs ~3-4 us in the step's graph).
__global__ void __launch_bounds__(1024)
iwae_loss_small_kernel(int B, int S, float log_n, const float* __restrict__ logp, const float* __restrict__ log_p,
                       const float* __restrict__ log_q, float* __restrict__ log_w, float* __restrict__ row_max,
                       float* __restrict__ row_sumexp, float* __restrict__ lse, float* __restrict__ loss,
                       float* __restrict__ unit_g_logw, float* __restrict__ unit_g_neg_logw) {
  // unit_g_logw / unit_g_neg_logw (optional): d loss / d log_w for an upstream gradient of exactly 1 -- what the
  // training step's backward asks for -- so that step needs no launch of iwae_loss_bwd_kernel.
  // B <= 64 rows (wave w owns rows w, w+16, w+32, w+48), S <= 256 samples (4 per lane): everything a wave needs
  // is loaded up front and kept in registers, so the only serial part is two wave reductions
  __shared__ float sm[16];
  const int n = B * S, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float lw[4][4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int b = wid + 16 * rr;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int sidx = lane + 64 * c;
      float v = -INFINITY;
      if (b < B && sidx < S) {
        const int i = b * S + sidx;
        v = ((logp[i] + logp[n + i]) + logp[2 * n + i]) + logp[3 * n + i];
        v = v + (log_p ? log_p[i] : 0.f) - (log_q ? log_q[i] : 0.f);
        log_w[i] = v;
      }
      lw[rr][c] = v;
    }
  }
  float acc = 0.f;
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    const int b = wid + 16 * rr;
    if (b >= B) break;  // (uniform per wave)
    float m = fmaxf(fmaxf(lw[rr][0], lw[rr][1]), fmaxf(lw[rr][2], lw[rr][3]));
    m = __shfl(wave_max(m), 0, 64);
    float se = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) se += (lane + 64 * c < S) ? expf(lw[rr][c] - m) : 0.f;
    se = wave_sum(se);
    const float l = m + logf(__shfl(se, 0, 64));
    if (lane == 0) {
      row_max[b] = m;
      row_sumexp[b] = se;
      lse[b] = l;
      acc += l - log_n;
    }
    if (unit_g_logw) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int sidx = lane + 64 * c;
        if (sidx < S) {
          const float g = -(1.f / (float)B) * expf(lw[rr][c] - l);  // same expression as iwae_loss_bwd_kernel, g_loss = 1
          unit_g_logw[b * S + sidx] = g;
          if (unit_g_neg_logw) unit_g_neg_logw[b * S + sidx] = -g;
        }
      }
    }
  }
  if (lane == 0) sm[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sm[w];
    loss[0] = -t / (float)B;
  }
}

// The same reduction with one block per row (any B, S <= 1024) and the mean over rows taken by the last block to
// finish: `ticket` is a zero-initialised device counter the caller keeps; the last block resets it, so the launch can
// be replayed from a hipGraph.  At B=36, S=200 this spreads the 6 x 7200 loads and the two reductions per row over 36
// CUs instead of one (9.6 -> ~5 us, the launch floor); the single-block kernel above stays for callers without a ticket.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
iwae_loss_rows_kernel(int B, int S, float log_n, const float* __restrict__ logp, const float* __restrict__ log_p,
                      const float* __restrict__ log_q, float* __restrict__ log_w, float* __restrict__ row_max,
                      float* __restrict__ row_sumexp, float* __restrict__ lse, float* __restrict__ loss,
                      float* __restrict__ unit_g_logw, float* __restrict__ unit_g_neg_logw, unsigned int* ticket) {
  __shared__ float sm[BLOCK / 64];
  __shared__ int is_last;
  const int n = B * S, b = blockIdx.x;
  float lw[4];
  float m = -INFINITY;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int sidx = threadIdx.x + BLOCK * c;
    float v = -INFINITY;
    if (sidx < S) {
      const int i = b * S + sidx;
      v = ((logp[i] + logp[n + i]) + logp[2 * n + i]) + logp[3 * n + i];
      v = v + (log_p ? log_p[i] : 0.f) - (log_q ? log_q[i] : 0.f);
      log_w[i] = v;
    }
    lw[c] = v;
    m = fmaxf(m, v);
  }
  m = block_max<BLOCK>(m, sm);
  float se = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) se += (threadIdx.x + BLOCK * c < S) ? expf(lw[c] - m) : 0.f;
  se = block_sum<BLOCK>(se, sm);
  const float l = m + logf(se);
  if (threadIdx.x == 0) {
    row_max[b] = m;
    row_sumexp[b] = se;
    lse[b] = l;
  }
  if (unit_g_logw) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int sidx = threadIdx.x + BLOCK * c;
      if (sidx < S) {
        const float g = -(1.f / (float)B) * expf(lw[c] - l);  // same expression as iwae_loss_bwd_kernel, g_loss = 1
        unit_g_logw[b * S + sidx] = g;
        if (unit_g_neg_logw) unit_g_neg_logw[b * S + sidx] = -g;
      }
    }
  }
  // the last block to get here sees every row's lse (release: fence before the ticket; acquire: fence after it)
  if (threadIdx.x == 0) {
    __threadfence();
    is_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
  }
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  float acc = 0.f;
  for (int r = threadIdx.x; r < B; r += BLOCK)
    acc += __hip_atomic_load(&lse[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - log_n;
  acc = block_sum<BLOCK>(acc, sm);  // fixed order: thread r takes rows r, r+BLOCK, ...; lanes, then waves
  if (threadIdx.x == 0) {
    loss[0] = -acc / (float)B;
    *ticket = 0u;
  }
}

// S sharded over ranks: after the per-rank (row max, row sum-exp) pairs have been all-gathered ([N][2][B]), everything
// that remains of the loss is ONE launch: lse[b] = M + log sum_r se_r exp(m_r - M), loss = -mean_b(lse - log n_total),
// and (optionally) d loss / d log_w of this rank's samples for a unit upstream gradient.  One block, wave per row.
__global__ void __launch_bounds__(1024)
iwae_combine_kernel(int N, int B, int S, float log_n, const float* __restrict__ gathered,
                    const float* __restrict__ log_w, float* __restrict__ lse, float* __restrict__ loss,
                    float* __restrict__ unit_g_logw, float* __restrict__ unit_g_neg_logw) {
  __shared__ float sm[16];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float acc = 0.f;
  for (int b = wid; b < B; b += 16) {
    float m = -INFINITY;
    for (int r = lane; r < N; r += 64) m = fmaxf(m, gathered[(size_t)(2 * r) * B + b]);
    m = __shfl(wave_max(m), 0, 64);
    float se = 0.f;
    for (int r = lane; r < N; r += 64)
      se += gathered[(size_t)(2 * r + 1) * B + b] * expf(gathered[(size_t)(2 * r) * B + b] - m);
    se = __shfl(wave_sum(se), 0, 64);
    const float l = m + logf(se);
    if (lane == 0) {
      lse[b] = l;
      acc += l - log_n;
    }
    if (unit_g_logw) {
      for (int sidx = lane; sidx < S; sidx += 64) {
        const float g = -(1.f / (float)B) * expf(log_w[b * S + sidx] - l);
        unit_g_logw[b * S + sidx] = g;
        if (unit_g_neg_logw) unit_g_neg_logw[b * S + sidx] = -g;
      }
    }
  }
  if (lane == 0) sm[wid] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += sm[w];
    loss[0] = -t / (float)B;
  }
}

// d loss / d log_w = -(g_loss / B) * softmax_s(log_w)
__global__ void iwae_loss_bwd_kernel(int B, int S, const float* __restrict__ log_w, const float* __restrict__ lse,
                                     const float* __restrict__ g_loss, float* __restrict__ g_logw,
                                     float* __restrict__ g_neg_logw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * S) return;
  const int b = i / S;
  const float g = -(g_loss[0] / (float)B) * expf(log_w[i] - lse[b]);
  g_logw[i] = g;
  if (g_neg_logw) g_neg_logw[i] = -g;  // d loss / d log_q
}

// OdeModel.device_conditioner applied to ones (reference vihds/ode.py:43-58, models/dr_constant.py:124-131):
//   out[e][b][s] = (default_e ? 1 : 0) + relu( sum_d (w_mean + w_std*z[e][d]) * dev1hot[r][d] * rel[e][d] ),
//   r = (b*S + s) mod B   -- the reference tiles the [B,1] conditioner output with .repeat([S,1]) against a
//   row-major flattening of [B,S] (ode.py:46,52-57); kept as is.
// rng (optional, {seed lo, seed hi, step, ticket} as in vihds_theta_opts): z is drawn here instead of read
// (counter = (e*D + d, 0xC04D, step, 0): a stream disjoint from theta's, whose second word is a parameter block < 2^16).
__global__ void device_condition_kernel(int E, int B, int S, int S_total, int s_off, int D, float w_mean, float w_std,
                                        const float* __restrict__ z, unsigned int* rng, int advance,
                                        const float* __restrict__ dev1hot, const float* __restrict__ rel,
                                        const int* __restrict__ is_default, float* __restrict__ out) {
  const int n = B * S;
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i0 < n;
  const int i = live ? i0 : n - 1;
  // the tiling quirk is defined on the global [B, S_total] sample grid: a rank holding samples s_off.. of every row
  // conditions them exactly as the unsharded run would
  const int bb = i / S;
  const int r = (int)(((long long)bb * S_total + s_off + (i - bb * S)) % B);
  unsigned int k0 = 0, k1 = 0, step = 0;
  if (rng) { k0 = rng[0]; k1 = rng[1]; step = rng[2]; }
  for (int e = 0; e < E; ++e) {
    float c = 0.f;
    for (int d = 0; d < D; ++d) {
      const float hot = dev1hot[r * D + d] * rel[e * D + d];
      float zz = 0.f;
      if (rng) { if (hot != 0.f) zz = philox_normal((unsigned int)(e * D + d), 0xC04Du, step, 0u, k0, k1, 0); }
      else zz = z[e * D + d];
      c += (w_mean + w_std * zz) * hot;
    }
    c = fmaxf(c, 0.f);
    if (live) out[(size_t)e * n + i] = (is_default[e] ? 1.f : 0.f) + c;
  }
  if (rng && advance) {  // (advance == 0: rng_advance_kernel follows, see theta_fwd_lds_kernel)
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int ticket = atomicAdd(&rng[3], 1u);
      if (ticket == gridDim.x - 1) { rng[2] = step + 1u; rng[3] = 0u; }
    }
  }
}

// Results.init (vihds/utils.py:79-99): one block per (b, t); rows of the [T][*][B][S] buffers are contiguous in s.
// ONE pass over the block's samples: a thread forms the importance weight of a sample once and feeds the 12 + n_species
// accumulators from 4 + 4 + n_species independent row loads (all in flight together); one barrier for all the block
// sums (thread-serial over s, a DPP scan over the wavefront, wavefronts in order; iw_summaries_rows_kernel, the fallback
// for more species, sums over the wavefront with a shuffle tree: the two agree to rounding).
constexpr int IWS_MAXSP = 16;
// VEC = 4 (S a multiple of 4, so every row of the [.][B][S] buffers starts on 16 bytes): a thread takes four CONSECUTIVE
// samples per round with one global_load_dwordx4 per row -- 16 B per lane is what a streaming read needs to approach the
// HBM rate on this chip (/opt/skills/guides/MI355X_MICROARCH.md: dwordx4 ~10 B/cycle/CU); the thread-serial part of the
// sum then runs over s = 4 tid .. 4 tid + 3 (+ 4 BLOCK per round) instead of tid (+ BLOCK per round).  VEC = 1: any S.
template <int VEC>
struct IwsVec {
  float v[VEC];
};
template <int VEC>
__device__ __forceinline__ IwsVec<VEC> iws_load(const float* __restrict__ p) {
  IwsVec<VEC> r;
  if constexpr (VEC == 4) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    r.v[0] = q.x; r.v[1] = q.y; r.v[2] = q.z; r.v[3] = q.w;
  } else {
    r.v[0] = *p;
  }
  return r;
}
// NSPV: the species rows the loop is unrolled for (n_species <= NSPV; the rows past n_species re-read row n_species - 1, a
// cache hit, so that no load sits behind a branch).
template <int BLOCK, int VEC, int NSPV>
__global__ void __launch_bounds__(BLOCK)
iw_summaries_kernel(int B, int S, int T, int N_total, int n_species, const float* __restrict__ log_w,
                    const float* __restrict__ lse, const float* __restrict__ traj, const float* __restrict__ xpred,
                    int obs_kind, const float* __restrict__ theta, int pr0, int pr1, int pr2, int pr3,
                    float* __restrict__ mu_out, float* __restrict__ std_out, float* __restrict__ states_out,
                    float* __restrict__ var_out) {
  constexpr int NW = BLOCK / 64, NV = 12 + NSPV;
  __shared__ float sm[NW][NV];
  const int b = blockIdx.x, t = blockIdx.y;
  const size_t n = (size_t)B * S;
  const float l = lse[b];
  const int prow[4] = {pr0, pr1, pr2, pr3};
  float acc[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) acc[k] = 0.f;
  for (int s = threadIdx.x * VEC; s < S; s += BLOCK * VEC) {
    const size_t i = (size_t)b * S + s;
    IwsVec<VEC> xpv[4], pcv[4], stv[NSPV];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      pcv[j] = iws_load<VEC>(theta ? theta + (size_t)prow[j] * n + i : traj + ((size_t)t * N_total + n_species + j) * n + i);
#pragma unroll
    for (int j = 0; j < NSPV; ++j)
      stv[j] = iws_load<VEC>(traj + ((size_t)t * N_total + (j < n_species ? j : n_species - 1)) * n + i);
    if (xpred) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xpv[j] = iws_load<VEC>(xpred + ((size_t)t * 4 + j) * n + i);
    }
    const IwsVec<VEC> lw = iws_load<VEC>(log_w + i);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float xp[4], st[NSPV > 6 ? NSPV : 6];
#pragma unroll
      for (int j = 0; j < (NSPV > 6 ? NSPV : 6); ++j) st[j] = j < NSPV && j < n_species ? stv[j < NSPV ? j : 0].v[e] : 0.f;
      if (xpred) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = xpv[j].v[e];
      } else {
        // the observed signals from the states this block reads anyway (OdeModel.observe, reference ode.py:84-93 and the
        // models' overrides): x_predict then never has to be written by the forward kernel nor read back here
        xp[0] = st[0];
        xp[1] = st[0] * st[1];
        if (obs_kind == VIHDS_OBS_DEFAULT) { xp[2] = st[0] * (st[2] + st[4]); xp[3] = st[0] * (st[3] + st[5]); }
        else if (obs_kind == VIHDS_OBS_INDUCER) { xp[2] = st[0] * (st[2] + st[3]); xp[3] = st[0] * st[4]; }
        else { xp[2] = st[0] * st[2]; xp[3] = st[0] * st[3]; }
      }
      const float w = expf(lw.v[e] - l);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float iv = 1.f / pcv[j].v[e];
        acc[3 * j] += w * xp[j];
        acc[3 * j + 1] += w * (xp[j] * xp[j] + iv);
        acc[3 * j + 2] += w * iv;
      }
#pragma unroll
      for (int j = 0; j < NSPV; ++j) acc[12 + j] += w * st[j];
    }
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    if (k < 12 + n_species) {
      const float v = wave_total(acc[k]);  // (DPP: twenty-odd sums per wavefront would otherwise queue on the LDS pipe)
      if (lane == 0) sm[wid][k] = v;
    }
  }
  __syncthreads();
  const int k = threadIdx.x;
  if (k < 4) {
    float r[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      r[q] = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) r[q] += sm[w][3 * k + q];
    }
    const size_t o = ((size_t)b * 4 + k) * T + t;
    mu_out[o] = r[0];
    std_out[o] = sqrtf(r[1] - r[0] * r[0]);
    var_out[o] = r[2];
  } else if (k >= 64 && k < 64 + n_species) {
    const int j = k - 64;
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) r += sm[w][12 + j];
    states_out[((size_t)b * n_species + j) * T + t] = r;
  }
}

// The kernel above for S <= 4 BLOCK (one round of float4 loads covers a row), as a loop over `tpb` consecutive time points
// of ONE data row per block: the importance weights (and, with constant precisions, the four 1/precision) are formed once
// per block instead of once per time point -- 5 of the 13 row loads of a time point disappear -- and the 8 (+4) row
// loads of time point t+1 are in flight while the sums of time point t are taken, so a CU's blocks no longer alternate
// between a load phase and a reduction phase with nothing in flight.  The barrier between the wavefront sums and the
// block sums waits for LDS only (a plain __syncthreads() would drain the prefetch); the LDS slots alternate by parity of
// t, which spares the second barrier.  Same per-sample arithmetic and the same summation order as the VEC = 4 kernel above
// (bit-identical results: test_iw_summaries_pipelined_kernel_matches_one_block_per_time_point).
template <int BLOCK, int NSPV, bool CONST_PREC>
__global__ void __launch_bounds__(BLOCK)
iw_summaries_pipe_kernel(int B, int S, int T, int tpb, int N_total, int n_species, const float* __restrict__ log_w,
                         const float* __restrict__ lse, const float* __restrict__ traj, const float* __restrict__ xpred,
                         int obs_kind, const float* __restrict__ theta, int pr0, int pr1, int pr2, int pr3,
                         float* __restrict__ mu_out, float* __restrict__ std_out, float* __restrict__ states_out,
                         float* __restrict__ var_out) {
  constexpr int NW = BLOCK / 64, NV = 12 + NSPV, NST = NSPV > 6 ? NSPV : 6;
  __shared__ float sm[2][NW][NV];
  const int b = blockIdx.x, t0 = blockIdx.y * tpb, t1 = min(T, t0 + tpb);
  const size_t n = (size_t)B * S;
  const bool live = (int)threadIdx.x * 4 < S;
  const size_t i = (size_t)b * S + (live ? threadIdx.x * 4 : 0);  // (idle threads re-read the row's first samples at weight 0)
  const int prow[4] = {pr0, pr1, pr2, pr3};
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const bool has_xp = xpred != nullptr;

  auto rows = [&](int t, IwsVec<4> (&stv)[NSPV], IwsVec<4> (&xpv)[4], IwsVec<4> (&pcv)[4]) {
#pragma unroll
    for (int j = 0; j < NSPV; ++j)
      stv[j] = iws_load<4>(traj + ((size_t)t * N_total + (j < n_species ? j : n_species - 1)) * n + i);
    if (!CONST_PREC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pcv[j] = iws_load<4>(traj + ((size_t)t * N_total + n_species + j) * n + i);
    }
    if (has_xp) {
#pragma unroll
      for (int j = 0; j < 4; ++j) xpv[j] = iws_load<4>(xpred + ((size_t)t * 4 + j) * n + i);
    }
  };

  IwsVec<4> cur_st[NSPV], cur_xp[4], cur_pc[4], nxt_st[NSPV], nxt_xp[4], nxt_pc[4];
  rows(t0, cur_st, cur_xp, cur_pc);
  float w[4], ivc[4][4];
  {
    const IwsVec<4> lw = iws_load<4>(log_w + i);
    const float l = lse[b];
    IwsVec<4> pc[4];
    if (CONST_PREC) {
#pragma unroll
      for (int j = 0; j < 4; ++j) pc[j] = iws_load<4>(theta + (size_t)prow[j] * n + i);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      w[e] = live ? expf(lw.v[e] - l) : 0.f;
      if (CONST_PREC) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ivc[j][e] = 1.f / pc[j].v[e];
      }
    }
  }
  for (int t = t0; t < t1; ++t) {
    rows(t + 1 < t1 ? t + 1 : t, nxt_st, nxt_xp, nxt_pc);  // (the last round re-reads its own rows: cache hits, no branch)
    float acc[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) acc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float xp[4], st[NST];
#pragma unroll
      for (int j = 0; j < NST; ++j) st[j] = j < NSPV && j < n_species ? cur_st[j < NSPV ? j : 0].v[e] : 0.f;
      if (has_xp) {
#pragma unroll
        for (int j = 0; j < 4; ++j) xp[j] = cur_xp[j].v[e];
      } else {
        xp[0] = st[0];
        xp[1] = st[0] * st[1];
        if (obs_kind == VIHDS_OBS_DEFAULT) { xp[2] = st[0] * (st[2] + st[4]); xp[3] = st[0] * (st[3] + st[5]); }
        else if (obs_kind == VIHDS_OBS_INDUCER) { xp[2] = st[0] * (st[2] + st[3]); xp[3] = st[0] * st[4]; }
        else { xp[2] = st[0] * st[2]; xp[3] = st[0] * st[3]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float iv = CONST_PREC ? ivc[j][e] : 1.f / cur_pc[j].v[e];
        acc[3 * j] += w[e] * xp[j];
        acc[3 * j + 1] += w[e] * (xp[j] * xp[j] + iv);
        acc[3 * j + 2] += w[e] * iv;
      }
#pragma unroll
      for (int j = 0; j < NSPV; ++j) acc[12 + j] += w[e] * st[j];
    }
    float(&slot)[NW][NV] = sm[(t - t0) & 1];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      if (k < 12 + n_species) {
        const float v = wave_total(acc[k]);
        if (lane == 0) slot[wid][k] = v;
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the LDS writes above; the prefetched rows stay in flight
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int k = threadIdx.x;
    if (k < 4) {
      float r[3];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        r[q] = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) r[q] += slot[wv][3 * k + q];
      }
      const size_t o = ((size_t)b * 4 + k) * T + t;
      mu_out[o] = r[0];
      std_out[o] = sqrtf(r[1] - r[0] * r[0]);
      var_out[o] = r[2];
    } else if (k >= 64 && k < 64 + n_species) {
      const int j = k - 64;
      float r = 0.f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) r += slot[wv][12 + j];
      states_out[((size_t)b * n_species + j) * T + t] = r;
    }
#pragma unroll
    for (int j = 0; j < NSPV; ++j) cur_st[j] = nxt_st[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) { cur_xp[j] = nxt_xp[j]; cur_pc[j] = nxt_pc[j]; }
  }
}

// The same, one pass over the samples per output row (any number of species): the fallback of iw_summaries_kernel.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK)
iw_summaries_rows_kernel(int B, int S, int T, int N_total, int n_species, const float* __restrict__ log_w,
                    const float* __restrict__ lse, const float* __restrict__ traj, const float* __restrict__ xpred,
                    int obs_kind, const float* __restrict__ theta, int pr0, int pr1, int pr2, int pr3,
                    float* __restrict__ mu_out, float* __restrict__ std_out, float* __restrict__ states_out,
                    float* __restrict__ var_out) {
  __shared__ float sm[BLOCK / 64];
  const int b = blockIdx.x, t = blockIdx.y;
  const size_t n = (size_t)B * S;
  const float l = lse[b];
  const int prow[4] = {pr0, pr1, pr2, pr3};
  for (int j = 0; j < 4; ++j) {
    float a_mu = 0.f, a_sq = 0.f, a_var = 0.f;
    for (int s = threadIdx.x; s < S; s += BLOCK) {
      const size_t i = (size_t)b * S + s;
      const float w = expf(log_w[i] - l);
      float xp;
      if (xpred) {
        xp = xpred[((size_t)t * 4 + j) * n + i];
      } else {
        auto y = [&](int q) { return traj[((size_t)t * N_total + q) * n + i]; };
        const float x0 = y(0);
        if (j == 0) xp = x0;
        else if (j == 1) xp = x0 * y(1);
        else if (obs_kind == VIHDS_OBS_DEFAULT) xp = j == 2 ? x0 * (y(2) + y(4)) : x0 * (y(3) + y(5));
        else if (obs_kind == VIHDS_OBS_INDUCER) xp = j == 2 ? x0 * (y(2) + y(3)) : x0 * y(4);
        else xp = x0 * y(j);
      }
      const float prec = theta ? theta[(size_t)prow[j] * n + i] : traj[((size_t)t * N_total + n_species + j) * n + i];
      const float iv = 1.f / prec;
      a_mu += w * xp;
      a_sq += w * (xp * xp + iv);
      a_var += w * iv;
    }
    a_mu = block_sum<BLOCK>(a_mu, sm);
    a_sq = block_sum<BLOCK>(a_sq, sm);
    a_var = block_sum<BLOCK>(a_var, sm);
    if (threadIdx.x == 0) {
      const size_t o = ((size_t)b * 4 + j) * T + t;
      mu_out[o] = a_mu;
      std_out[o] = sqrtf(a_sq - a_mu * a_mu);
      var_out[o] = a_var;
    }
  }
  for (int j = 0; j < n_species; ++j) {
    float a = 0.f;
    for (int s = threadIdx.x; s < S; s += BLOCK) {
      const size_t i = (size_t)b * S + s;
      a += expf(log_w[i] - l) * traj[((size_t)t * N_total + j) * n + i];
    }
    a = block_sum<BLOCK>(a, sm);
    if (threadIdx.x == 0) states_out[((size_t)b * n_species + j) * T + t] = a;
  }
}

// ---- launchers (called from vihds_api.hip) ---------------------------------------------------------
void launch_theta_fwd(int P, int B, int S, const int* kind, const float* q_mu, const float* q_prec, const float* p_mu,
                      const float* p_prec, const float* lo, const float* hi, float* u, float* theta, float* log_q,
                      float* log_p, const vihds_theta_opts& o, hipStream_t st) {
  const int n = B * S, blk = 64;
  const int nb_max = min(B, 63 / S + 2);
  const size_t lds = (size_t)THETA_LDS_FIELDS * nb_max * P * sizeof(float);
  if (lds <= 48 * 1024) {  // per-(row, parameter) constants staged in LDS
    const int blocks = (n + 63) / 64;
    const int advance = blocks <= RNG_TICKET_BLOCKS;
    hipLaunchKernelGGL(theta_fwd_lds_kernel, dim3(blocks), dim3(256), lds, st, P, B, S, nb_max, kind, q_mu, q_prec,
                       o.q_rows, o.q_prec_is_log, p_mu, p_prec, lo, hi, u, o.rng, advance, o.rng ? o.S_total : S,
                       o.rng ? o.s_offset : 0, theta, log_q, log_p);
    if (o.rng && !advance) hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, st, o.rng);
    return;
  }
  hipLaunchKernelGGL(theta_fwd_kernel, dim3((4 * n + blk - 1) / blk), dim3(blk), 0, st, P, B, S, kind, q_mu, q_prec,
                     o.q_rows, o.q_prec_is_log, p_mu, p_prec, lo, hi, u, o.rng, o.rng ? o.S_total : S,
                     o.rng ? o.s_offset : 0, theta, log_q, log_p);
}
void launch_theta_bwd(int P, int B, int S, const int* kind, const float* q_mu, const float* q_prec, const float* p_mu,
                      const float* p_prec, const float* lo, const float* hi, const float* u, const float* g_theta,
                      const float* g_log_q, const float* g_log_p, float* g_q_mu, float* g_q_prec,
                      const vihds_theta_opts& o, hipStream_t st) {
  vihds_iwae_job iw = {};
  if (o.iwae) iw = *o.iwae;
  const size_t lds = o.iwae ? (size_t)S * sizeof(float) : 0;
  hipLaunchKernelGGL((theta_bwd_kernel<256>), dim3(B, (P + THETA_BWD_PCHUNK - 1) / THETA_BWD_PCHUNK), dim3(256), lds, st,
                     P, B, S, kind, q_mu, q_prec, o.q_rows, o.q_prec_is_log, p_mu, p_prec, lo, hi, u, g_theta,
                     o.g_theta_scale, g_log_q, g_log_p, g_q_mu, g_q_prec, iw);
}
void launch_iwae_fwd(int B, int S, const float* logp, const float* log_p, const float* log_q, float* log_w,
                     float* row_max, float* row_sumexp, hipStream_t st) {
  hipLaunchKernelGGL((iwae_fwd_kernel<256>), dim3(B), dim3(256), 0, st, B, S, logp, log_p, log_q, log_w, row_max,
                     row_sumexp);
}
void launch_iwae_bwd(int B, int S, const float* log_w, const float* lse, const float* g_lse, float* g_logw,
                     hipStream_t st) {
  const int n = B * S, blk = 256;
  hipLaunchKernelGGL(iwae_bwd_kernel, dim3((n + blk - 1) / blk), dim3(blk), 0, st, B, S, log_w, lse, g_lse, g_logw);
}
void launch_iwae_finish(int B, float log_n, const float* row_max, const float* row_sumexp, float* lse, float* loss,
                        hipStream_t st) {
  hipLaunchKernelGGL(iwae_finish_kernel, dim3(1), dim3(256), 0, st, B, log_n, row_max, row_sumexp, lse, loss);
}
void launch_iwae_loss_small(int B, int S, float log_n, const float* logp, const float* log_p, const float* log_q,
                            float* log_w, float* row_max, float* row_sumexp, float* lse, float* loss,
                            float* unit_g_logw, float* unit_g_neg_logw, hipStream_t st) {
  hipLaunchKernelGGL(iwae_loss_small_kernel, dim3(1), dim3(1024), 0, st, B, S, log_n, logp, log_p, log_q, log_w, row_max,
                     row_sumexp, lse, loss, unit_g_logw, unit_g_neg_logw);
}
void launch_iwae_loss_rows(int B, int S, float log_n, const float* logp, const float* log_p, const float* log_q,
                           float* log_w, float* row_max, float* row_sumexp, float* lse, float* loss,
                           float* unit_g_logw, float* unit_g_neg_logw, unsigned int* ticket, hipStream_t st) {
  hipLaunchKernelGGL((iwae_loss_rows_kernel<256>), dim3(B), dim3(256), 0, st, B, S, log_n, logp, log_p, log_q, log_w,
                     row_max, row_sumexp, lse, loss, unit_g_logw, unit_g_neg_logw, ticket);
}
void launch_iwae_combine(int N, int B, int S, float log_n, const float* gathered, const float* log_w, float* lse,
                         float* loss, float* unit_g_logw, float* unit_g_neg_logw, hipStream_t st) {
  hipLaunchKernelGGL(iwae_combine_kernel, dim3(1), dim3(1024), 0, st, N, B, S, log_n, gathered, log_w, lse, loss,
                     unit_g_logw, unit_g_neg_logw);
}
void launch_iwae_loss_bwd(int B, int S, const float* log_w, const float* lse, const float* g_loss, float* g_logw,
                          float* g_neg_logw, hipStream_t st) {
  const int n = B * S, blk = 256;
  hipLaunchKernelGGL(iwae_loss_bwd_kernel, dim3((n + blk - 1) / blk), dim3(blk), 0, st, B, S, log_w, lse, g_loss,
                     g_logw, g_neg_logw);
}
void launch_device_condition(int E, int B, int S, int S_total, int s_off, int D, float w_mean, float w_std,
                             const float* z, unsigned int* rng,
                             const float* dev1hot, const float* rel, const int* is_default, float* out,
                             hipStream_t st) {
  const int n = B * S, blk = 256;
  const int blocks = (n + blk - 1) / blk;
  const int advance = blocks <= RNG_TICKET_BLOCKS;
  hipLaunchKernelGGL(device_condition_kernel, dim3(blocks), dim3(blk), 0, st, E, B, S, S_total, s_off, D, w_mean, w_std, z,
                     rng, advance, dev1hot, rel, is_default, out);
  if (rng && !advance) hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, st, rng);
}

// ---------------------------------------------------------------------------------------------------------------
// Adam over a table of parameter tensors (include/vihds_hip.h: vihds_adam_step).  One block = 1024 consecutive
// elements of one tensor; torch's fused multi-tensor Adam gives a whole 64k chunk to a single block, which leaves
// the 36 000-element encoder matrix on one CU for ~23 us.
constexpr int ADAM_CHUNK = 1024;
__global__ void __launch_bounds__(256) adam_kernel(vihds_adam_tensors t, float* __restrict__ m, float* __restrict__ v,
                                                   float* state, const float* lr_dev, float lr, float beta1,
                                                   float beta2, float eps, float grad_scale,
                                                   const float* __restrict__ gate) {
  // which tensor does this block belong to
  int blk = blockIdx.x, k = 0, off = 0;
  for (; k < t.n; ++k) {
    const int nb = (t.size[k] + ADAM_CHUNK - 1) / ADAM_CHUNK;
    if (blk < nb) break;
    blk -= nb;
    off += t.size[k];
  }
  const float step = state[0] + 1.f;
  // gate = the step's loss: not finite -> nothing is updated and the step is not counted (all or nothing per step: the
  // reference stops before optimizer.step on a NaN ELBO, training.py:331-334)
  const bool open = gate == nullptr || fabsf(gate[0]) <= 3.402823466e38f;
  if (open && k < t.n && t.grad[k] != nullptr) {
    const float bc1 = 1.f - powf(beta1, step);
    const float bc2_sqrt = sqrtf(1.f - powf(beta2, step));
    const float step_size = (lr_dev ? lr_dev[0] : lr) / bc1;
    float* p = t.param[k];
    const float* g = t.grad[k];
    const int end = min(t.size[k], (blk + 1) * ADAM_CHUNK);
    for (int e = blk * ADAM_CHUNK + threadIdx.x; e < end; e += 256) {
      const float ge = g[e] * grad_scale;
      // second line of defence (a finite loss with an overflowed gradient element): that element's parameter and
      // moments stay as they were
      if (!(fabsf(ge) <= 3.402823466e38f)) continue;
      float me = m[off + e], ve = v[off + e];
      me += (ge - me) * (1.f - beta1);
      ve = ve * beta2 + (1.f - beta2) * ge * ge;
      m[off + e] = me;
      v[off + e] = ve;
      p[e] -= step_size * (me / (sqrtf(ve) / bc2_sqrt + eps));
    }
  }
  // the last block to get here has seen every other block read state[0]
  __syncthreads();
  if (threadIdx.x == 0) {
    const float ticket = atomicAdd(&state[1], 1.f);
    if (ticket == (float)(gridDim.x - 1)) {
      if (open) state[0] = step;
      state[1] = 0.f;
    }
  }
}

void launch_adam(const vihds_adam_tensors& t, float* m, float* v, float* state, const float* lr_dev, float lr,
                 float beta1, float beta2, float eps, float grad_scale, const float* gate, hipStream_t st) {
  int blocks = 0;
  for (int k = 0; k < t.n; ++k) blocks += (t.size[k] + ADAM_CHUNK - 1) / ADAM_CHUNK;
  if (blocks == 0) return;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, st, t, m, v, state, lr_dev, lr, beta1, beta2, eps,
                     grad_scale, gate);
}
// tests: > 0 fixes the time points per block of the pipelined kernel, < 0 takes the one-block-per-time-point kernel
int iw_summaries_tpb_override = 0;
void launch_iw_summaries(int B, int S, int T, int N_total, int n_species, const float* log_w, const float* lse,
                         const float* traj, const float* xpred, int obs_kind, const float* theta, const int* prec_rows,
                         float* mu, float* sd, float* states, float* var, hipStream_t st) {
  const int tpb_override = iw_summaries_tpb_override;
  int r[4] = {0, 0, 0, 0};
  if (theta && prec_rows) for (int j = 0; j < 4; ++j) r[j] = prec_rows[j];
#define VIHDS_IWS(VEC, NSPV)                                                                                          \
  hipLaunchKernelGGL((iw_summaries_kernel<256, VEC, NSPV>), dim3(B, T), dim3(256), 0, st, B, S, T, N_total, n_species,  \
                     log_w, lse, traj, xpred, obs_kind, theta, r[0], r[1], r[2], r[3], mu, sd, states, var)
#define VIHDS_IWS_SPECIES(VEC)                                                                                        \
  do {                                                                                                                \
    if (n_species <= 4) VIHDS_IWS(VEC, 4);                                                                            \
    else if (n_species <= 8) VIHDS_IWS(VEC, 8);                                                                       \
    else if (n_species <= 12) VIHDS_IWS(VEC, 12);                                                                     \
    else VIHDS_IWS(VEC, 16);                                                                                          \
  } while (0)
#define VIHDS_IWS_PIPE(NSPV)                                                                                        \
  do {                                                                                                              \
    if (theta)                                                                                                      \
      hipLaunchKernelGGL((iw_summaries_pipe_kernel<256, NSPV, true>), dim3(B, (T + tpb - 1) / tpb), dim3(256), 0, st, B, \
                         S, T, tpb, N_total, n_species, log_w, lse, traj, xpred, obs_kind, theta, r[0], r[1], r[2], r[3], \
                         mu, sd, states, var);                                                                      \
    else                                                                                                            \
      hipLaunchKernelGGL((iw_summaries_pipe_kernel<256, NSPV, false>), dim3(B, (T + tpb - 1) / tpb), dim3(256), 0, st, B, \
                         S, T, tpb, N_total, n_species, log_w, lse, traj, xpred, obs_kind, theta, r[0], r[1], r[2], r[3], \
                         mu, sd, states, var);                                                                      \
  } while (0)
  // time points per block of the pipelined kernel: as many as leave >= 2048 blocks (8 per CU), between 2 and 8
  // (tests/probe/summaries_time.py at B=234, S=1000, T=86, N=8: 1 -> 217 us, 2 -> 159, 4 -> 130, 8 -> 129; one block per time
  // point: 162)
  const long long cells = (long long)B * T;
  const int tpb = tpb_override > 0 ? tpb_override : (int)(cells / 2048 < 2 ? 2 : (cells / 2048 > 8 ? 8 : cells / 2048));
  if (n_species >= 1 && n_species <= IWS_MAXSP && S % 4 == 0 && S >= 512 && S <= 1024 && tpb_override >= 0) {
    if (n_species <= 4) VIHDS_IWS_PIPE(4);
    else if (n_species <= 8) VIHDS_IWS_PIPE(8);
    else if (n_species <= 12) VIHDS_IWS_PIPE(12);
    else VIHDS_IWS_PIPE(16);
  } else if (n_species >= 1 && n_species <= IWS_MAXSP && S % 4 == 0 && S >= 512) VIHDS_IWS_SPECIES(4);
  else if (n_species >= 1 && n_species <= IWS_MAXSP) VIHDS_IWS_SPECIES(1);
  else
    hipLaunchKernelGGL((iw_summaries_rows_kernel<256>), dim3(B, T), dim3(256), 0, st, B, S, T, N_total, n_species,
                       log_w, lse, traj, xpred, obs_kind, theta, r[0], r[1], r[2], r[3], mu, sd, states, var);
}

}  // namespace vihds
