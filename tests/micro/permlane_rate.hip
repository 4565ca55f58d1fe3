// Micro-benchmark: issue cost of gfx950's v_permlane16_swap / v_permlane32_swap against a DPP addition, per wavefront.
// Build/run: hipcc --offload-arch=gfx950 -O3 tests/micro/permlane_rate.hip -o /tmp/permlane_rate && /tmp/permlane_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int MODE>
__global__ void k(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N_IT; ++i) {
    if (MODE == 0) {  // 4 independent permlane32 swaps (8 registers)
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x2), "+v"(x3));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x4), "+v"(x5));
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x6), "+v"(x7));
    } else if (MODE == 1) {  // 4 permlane16 swaps
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x2), "+v"(x3));
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x4), "+v"(x5));
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(x6), "+v"(x7));
    } else if (MODE == 2) {  // 4 swaps without the s_nop
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x0), "+v"(x1));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x2), "+v"(x3));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x4), "+v"(x5));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x6), "+v"(x7));
    } else if (MODE == 3) {  // 8 independent DPP additions (row_ror:1)
      x0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x121, 0xf, 0xf, false));
      x1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x1), 0x121, 0xf, 0xf, false));
      x2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x2), 0x121, 0xf, 0xf, false));
      x3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x3), 0x121, 0xf, 0xf, false));
      x4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x4), 0x121, 0xf, 0xf, false));
      x5 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x5), 0x121, 0xf, 0xf, false));
      x6 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x6), 0x121, 0xf, 0xf, false));
      x7 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x7), 0x121, 0xf, 0xf, false));
    } else if (MODE == 4) {  // 8 independent fma (reference)
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (MODE == 5) {  // 8 row_bcast:31 DPP additions (cross-row broadcast)
      x0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0x143, 0xc, 0xf, false));
      x1 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x1), 0x143, 0xc, 0xf, false));
      x2 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x2), 0x142, 0xa, 0xf, false));
      x3 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x3), 0x142, 0xa, 0xf, false));
      x4 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x4), 0x143, 0xc, 0xf, false));
      x5 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x5), 0x143, 0xc, 0xf, false));
      x6 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x6), 0x142, 0xa, 0xf, false));
      x7 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x7), 0x142, 0xa, 0xf, false));
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[4096] = t1 - t0;
}
template <int MODE>
static void run(const char* name, int per_iter, int waves_per_simd) {
  float* d;
  (void)hipMalloc(&d, 1 << 20);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256 * waves_per_simd), 0, 0, d, 1.0001f, 0.5f);
  (void)hipDeviceSynchronize();
  long long c;
  (void)hipMemcpy(&c, (char*)d + 4096 * 8, 8, hipMemcpyDeviceToHost);
  printf("%-44s %d wave(s)/SIMD: %6.1f cycles per instruction\n", name, waves_per_simd, (double)c / N_IT / per_iter);
  (void)hipFree(d);
}
int main() {
  for (int w = 1; w <= 4; w *= 2) {
    run<0>("v_permlane32_swap (+ s_nop 1)", 4, w);
    run<1>("v_permlane16_swap (+ s_nop 1)", 4, w);
    run<2>("v_permlane32_swap", 4, w);
    run<3>("v_add_f32 dpp row_ror:1", 8, w);
    run<5>("v_add_f32 dpp row_bcast:15 / 31", 8, w);
    run<4>("v_fma_f32", 8, w);
  }
  return 0;
}
