// The sampling stage of a training step as a PROLOGUE of an ODE forward launch (vihds_theta_ode_fwd, ABI 13), for kernels
// whose blocks own a contiguous run of trajectories (the lane-split relay / degrader / prpr / auto kernels, dr_blackbox's
// cooperating wavefronts): theta = p.clip(q.sample(u), 4 sigma) with log q(theta), log p(theta) (reference vae.py:31-34,
// distributions.py:64-85,119-142,327-381 -- the arithmetic of theta_fwd_lds_kernel, csrc/vihds_elbo.hip) for the block's own
// trajectories, and dr_blackbox's condition_theta (y + offset_layer(dev_1hot), models/dr_blackbox.py:86-96) behind it.
// theta / u / log q / log p are written to global memory as the separate launch writes them (they are API outputs: the tail's
// theta adjoint reads u and theta's tables again); the block's integrator reads its theta rows back after one barrier.
//
// Work decomposition: one item = (trajectory of the block, block of four consecutive parameters) -- the unit of the in-kernel
// generator, one Philox call -- handed out thread by thread, whatever the integrator's own lane layout is; per-(row, parameter)
// constants (sigma, density constants, clip bounds) are tabulated once per block in LDS.  The generator's step is READ here
// and advanced by the launch that follows in the step (vihds_step_tail, `rng_advance`): no ticket, no returning atomic.
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_args.hpp"
#include "vihds_rng.hpp"

namespace vihds {

// LDS floats the stage needs: ten tables of nb_max * P entries, then two partial-sum arrays [ntraj][KB]
__host__ __device__ inline size_t theta_stage_lds_floats(int nb_max, int P, int ntraj) {
  return (size_t)10 * nb_max * P + (size_t)2 * ntraj * ((P + 3) / 4);
}

template <int NT>
__device__ __forceinline__ void theta_stage_block(const OdeArgs& a, const ThetaStageArgs& t, int first, int ntraj, int nb_max,
                                                  float* sc) {
  constexpr float LOG2PI = 1.8378770664093453f;
  const int n = a.n, P = t.P, B = a.B, S = a.S, tid = threadIdx.x;
  const int last = min(first + ntraj, n) - 1;
  const int b0 = first / S, nb = last / S - b0 + 1;
  const int stride = nb_max * P;
  float* t_kind = sc;
  float* t_mu = sc + stride;
  float* t_sigma = sc + 2 * stride;
  float* t_prec = sc + 3 * stride;
  float* t_cq = sc + 4 * stride;
  float* t_lo = sc + 5 * stride;
  float* t_hi = sc + 6 * stride;
  float* t_pmu = sc + 7 * stride;
  float* t_cp = sc + 8 * stride;
  float* t_pprec = sc + 9 * stride;
  const int KB = (P + 3) / 4;
  float* lq_part = sc + 10 * stride;
  float* lp_part = lq_part + ntraj * KB;
  // ---- tables: entry e = (row bb of the block, parameter p)
  for (int e = tid; e < nb * P; e += NT) {
    const int bb = e / P, p = e - bb * P;
    const int rm = t.q_rows ? t.q_rows[p] : p, rp = t.q_rows ? t.q_rows[P + p] : p;
    const int kd = t.kind[p];
    const float pr = t.q_prec[(size_t)rp * B + b0 + bb], mu = t.q_mu[(size_t)rm * B + b0 + bb];
    const float lo = t.clip_lo[p], hi = t.clip_hi[p], pmu = t.p_mu[p], pp = t.p_prec[p];
    const float prec = (kd == KIND_CONSTANT) ? 1.f : (t.prec_is_log ? expf(pr) : pr);
    t_kind[e] = (float)kd;
    t_mu[e] = mu;
    t_sigma[e] = 1.f / sqrtf(prec);
    t_prec[e] = prec;
    t_cq[e] = -LOG2PI + 0.5f * logf(prec + 1e-12f);
    t_lo[e] = lo;
    t_hi[e] = hi;
    t_pmu[e] = pmu;
    t_cp[e] = -LOG2PI + 0.5f * logf(pp + 1e-12f);
    t_pprec[e] = pp;
  }
  unsigned int k0 = 0, k1 = 0, step = 0;
  if (t.rng) { k0 = t.rng[0]; k1 = t.rng[1]; step = t.rng[2]; }
  __syncthreads();
  // ---- items: (trajectory tl of the block, parameter block kb)
  for (int it = tid; it < ntraj * KB; it += NT) {
    const int tl = it / KB, kb = it - tl * KB;
    const int i0 = first + tl;
    const bool live = i0 < n;
    const int i = live ? i0 : n - 1;
    const int b = i / S, row = (b - b0) * P;
    float z4[4] = {0.f, 0.f, 0.f, 0.f};
    if (t.rng) {
      const unsigned int gidx = (unsigned int)(b * t.S_total + t.s_off + (i - b * S));
      philox_normal4(gidx, (unsigned int)kb, step, 0u, k0, k1, z4);
    }
    float lq = 0.f, lp = 0.f;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = 4 * kb + jj;
      if (p >= P) break;
      float uu;
      if (t.rng) {
        uu = z4[jj];
        if (live) t.u[(size_t)i * P + p] = uu;
      } else {
        uu = t.u[(size_t)i * P + p];
      }
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
      if (live) t.theta[(size_t)p * n + i] = x;
    }
    lq_part[it] = lq;
    lp_part[it] = lp;
  }
  __syncthreads();
  // ---- per trajectory: the chained sums (parameter blocks in order), then dr_blackbox's conditioned rows
  for (int tl = tid; tl < ntraj; tl += NT) {
    const int i = first + tl;
    if (i < n) {
      float lq = 0.f, lp = 0.f;
      for (int kb = 0; kb < KB; ++kb) { lq += lq_part[tl * KB + kb]; lp += lp_part[tl * KB + kb]; }
      if (t.log_q) t.log_q[i] = lq;
      if (t.log_p) t.log_p[i] = lp;
    }
  }
  if (t.off_n > 0) {
    for (int it = tid; it < ntraj * t.off_n; it += NT) {
      const int tl = it / t.off_n, k = it - tl * t.off_n;
      const int i = first + tl;
      if (i < n) {
        const int b = i / S;
        float o = t.off_b[k];  // (offset_rows_fwd_kernel's arithmetic, csrc/vihds_offset.hip)
        for (int d = 0; d < a.D; ++d) o = fmaf(t.off_w[k * a.D + d], a.dev1hot[(size_t)b * a.D + d], o);
        t.theta[(size_t)(t.off_dst + k) * n + i] = t.theta[(size_t)(t.off_src + k) * n + i] + o;
      }
    }
  }
  __syncthreads();  // theta of this block's trajectories is in memory; the scratch region is free again
}

}  // namespace vihds
