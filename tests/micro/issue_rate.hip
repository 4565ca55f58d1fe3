// Micro-benchmark: per-wavefront VALU issue interval and dependent latency on gfx950, one wave per SIMD.
// Build/run: hipcc --offload-arch=gfx950 -O3 tests/micro/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_IT 4096
template <int MODE>
__global__ void k(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int i = 0; i < N_IT; ++i) {
    if (MODE == 0) {  // 8 dependent fma
      x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
      x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
    } else if (MODE == 1) {  // 8 independent fma
      x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
    } else if (MODE == 2) {  // 8 dependent rcp
      x0 = __builtin_amdgcn_rcpf(x0); x0 = __builtin_amdgcn_rcpf(x0); x0 = __builtin_amdgcn_rcpf(x0); x0 = __builtin_amdgcn_rcpf(x0);
      x0 = __builtin_amdgcn_rcpf(x0); x0 = __builtin_amdgcn_rcpf(x0); x0 = __builtin_amdgcn_rcpf(x0); x0 = __builtin_amdgcn_rcpf(x0);
    } else if (MODE == 3) {  // 8 dependent (dpp mov + fma)
      for (int q = 0; q < 8; ++q) {
        float t = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x0), 0xB1, 0xf, 0xf, false));
        x0 = fmaf(t, a, b);
      }
    } else if (MODE == 4) {  // 8 independent rcp
      x0 = __builtin_amdgcn_rcpf(x0); x1 = __builtin_amdgcn_rcpf(x1); x2 = __builtin_amdgcn_rcpf(x2); x3 = __builtin_amdgcn_rcpf(x3);
      x4 = __builtin_amdgcn_rcpf(x4); x5 = __builtin_amdgcn_rcpf(x5); x6 = __builtin_amdgcn_rcpf(x6); x7 = __builtin_amdgcn_rcpf(x7);
    } else if (MODE == 5) {  // 8 dependent pk_fma
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 v = {x0, x1}, aa = {a, a}, bb = {b, b};
      for (int q = 0; q < 8; ++q) v = __builtin_elementwise_fma(v, aa, bb);
      x0 = v.x; x1 = v.y;
    }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) ((long long*)out)[1024] = t1 - t0;
}
template <int MODE>
void run(const char* name, float* d, int blocks, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 0.999f, 0.001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 0.999f, 0.001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long cyc; hipMemcpy(&cyc, ((long long*)d) + 1024, 8, hipMemcpyDeviceToHost);
  printf("%-28s blocks=%4d threads=%3d: %7.1f us  -> %.2f ns per instr per wave ; s_memtime/cyclecounter ticks per instr %.2f\n", name, blocks, threads, ms * 1e3,
         ms * 1e6 / (N_IT * 8.0), (double)cyc / (N_IT * 8.0));
}
int main() {
  float* d; hipMalloc(&d, 1 << 22);
  for (int rep = 0; rep < 1; ++rep) {
    run<0>("dependent fma", d, 256, 64);   run<1>("independent fma", d, 256, 64);
    run<2>("dependent rcp", d, 256, 64);   run<4>("independent rcp", d, 256, 64);
    run<3>("dependent dpp+fma", d, 256, 64); run<5>("dependent pk_fma", d, 256, 64);
    run<0>("dependent fma 4w/CU", d, 256, 256); run<1>("independent fma 4w/CU", d, 256, 256);
    run<0>("dependent fma 8w/CU", d, 256, 512); run<1>("independent fma 8w/CU", d, 256, 512);
    run<1>("independent fma 16w/CU", d, 256, 1024);
  }
  return 0;
}
