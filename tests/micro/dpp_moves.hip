// Check: the cross-row lane moves of vihds_dr_scan.hpp without ds_bpermute (gfx950 = GFX9 DPP controls + the row swaps).
// Build/run: hipcc --offload-arch=gfx950 -O3 tests/micro/dpp_moves.hip -o /tmp/dpp_moves && /tmp/dpp_moves
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float bperm(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, ROWMASK, 0xf, false));
}
__global__ void k(float* o) {
  const int lane = threadIdx.x;
  const float v = o[lane];
  // 0: lane - 1 (wave_shr:1)   1: lane + 1 (wave_shl:1)   2: lane 15 of the row below, rows 1 and 3 (row_bcast:15)
  // 3: lane 16 / 48 for the lower row (readlane)   4: lane ^ 16 sum (permlane16_swap)
  // (every move with all lanes active, the selection afterwards: a move under a partial EXEC mask reads `old` / zero from
  // the lanes that are switched off)
  const float m1 = bperm(v, lane - 1), d1 = dpp<0x138, 0xf>(-1.f, v);
  const float m2 = bperm(v, lane + 1), d2 = dpp<0x130, 0xf>(-1.f, v);
  const float m3 = bperm(v, (lane & 32) + 15), d3 = dpp<0x142, 0xa>(-1.f, v);
  const float m4 = bperm(v, (lane & 32) + 16);
  const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  o[64 * 1 + lane] = lane > 0 ? m1 : -1.f;
  o[64 * 2 + lane] = lane > 0 ? d1 : -1.f;
  o[64 * 3 + lane] = lane < 63 ? m2 : -1.f;
  o[64 * 4 + lane] = lane < 63 ? d2 : -1.f;
  o[64 * 5 + lane] = (lane & 31) >= 16 ? m3 : -1.f;
  o[64 * 6 + lane] = (lane & 31) >= 16 ? d3 : -1.f;
  o[64 * 7 + lane] = (lane & 31) < 16 ? m4 : -1.f;
  o[64 * 8 + lane] = (lane & 31) < 16 ? (lane < 32 ? s0 : s1) : -1.f;
  o[64 * 9 + lane] = v + bperm(v, lane ^ 16);
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  o[64 * 10 + lane] = a + b;
}
int main() {
  float h[64 * 11];
  for (int i = 0; i < 64; ++i) h[i] = 1.0f / (i + 3);
  float* d;
  (void)hipMalloc(&d, sizeof(h));
  (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"lane - 1  wave_shr:1", "lane + 1  wave_shl:1", "row below's lane 15  row_bcast:15", "row above's lane 0  readlane", "v + v[lane ^ 16]  permlane16_swap"};
  for (int t = 0; t < 5; ++t) {
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad += h[64 * (1 + 2 * t) + i] != h[64 * (2 + 2 * t) + i];
    printf("%-44s mismatches %d\n", names[t], bad);
  }
  return 0;
}
