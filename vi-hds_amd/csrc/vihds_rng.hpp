// Shared by the theta kernels (vihds_elbo.hip) and the fused decoder-step kernel (vihds_dr_lanes.hpp): distribution
// kinds and the counter-based normal generator.
#pragma once
#include <hip/hip_runtime.h>

namespace vihds {

enum { KIND_NORMAL = 0, KIND_LOGNORMAL = 1, KIND_CONSTANT = 2 };

// Counter-based standard normals for u (the reference draws u ~ N(0,1)[B,S,P] on the host, vae.py:22-24; this is the
// graph-capturable device alternative): Philox4x32-10 (Salmon et al., SC'11) keyed by the 64-bit seed, counter =
// (global sample index b*S_total + s, parameter block p/4, step lo, step hi); the four 32-bit outputs give the four
// normals of parameters 4k..4k+3 through two Box-Muller pairs, u1 = (x + 0.5) 2^-32, u2 = (y + 0.5) 2^-32,
// z0 = sqrt(-2 ln u1) cos(2 pi u2), z1 = sqrt(-2 ln u1) sin(2 pi u2).  Independent of the launch geometry and of how
// S is sharded over ranks.  tests/test_hip_parity.py re-implements it in numpy (with the Random123 known answer).
__device__ __forceinline__ void philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                              unsigned int k0, unsigned int k1, unsigned int* out) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned int hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned int hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ void philox_normal4(unsigned int, unsigned int, unsigned int, unsigned int, unsigned int,
                                               unsigned int, float*);
__device__ __forceinline__ float philox_normal(unsigned int idx, unsigned int pblock, unsigned int step_lo,
                                               unsigned int step_hi, unsigned int k0, unsigned int k1, int q) {
  float z[4];
  philox_normal4(idx, pblock, step_lo, step_hi, k0, k1, z);
  return z[q];
}

// all four normals of one counter: (r0, r1) -> (z0, z1), (r2, r3) -> (z2, z3).  v_log / v_sqrt / v_sin / v_cos
// (v_sin_f32 and v_cos_f32 take their argument in revolutions, which is exactly u2): the draws only have to be good
// normals, and whatever is drawn is written out as `u`, so everything downstream sees the same values.
__device__ __forceinline__ void philox_normal4(unsigned int idx, unsigned int pblock, unsigned int step_lo,
                                               unsigned int step_hi, unsigned int k0, unsigned int k1, float* z) {
  unsigned int r[4];
  philox4x32_10(idx, pblock, step_lo, step_hi, k0, k1, r);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = fminf(((float)r[2 * h] + 0.5f) * 2.3283064365386963e-10f, 0.99999994f);
    const float u2 = ((float)r[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
    const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln u1 = -2 ln2 log2 u1
    z[2 * h] = rad * __builtin_amdgcn_cosf(u2);
    z[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
  }
}


}  // namespace vihds
