// Micro-probe (not a test): duration of back-to-back launches of an (almost) empty kernel as a function of its dynamic
// LDS size, block count and kernel-argument size -- what part of a launch is ramp.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[160]; };
__global__ void __launch_bounds__(256) k_small(float* out) { extern __shared__ float lds[]; if (out && threadIdx.x == 9999) out[0] = lds[0]; }
__global__ void __launch_bounds__(256) k_big(Big b, float* out) { extern __shared__ float lds[]; if (out && threadIdx.x == 9999) out[0] = lds[b.v[3]]; }
template <class F> static float timeit(F f, int n = 200) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) f();
  hipDeviceSynchronize(); hipEventRecord(e0);
  for (int i = 0; i < n; ++i) f();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / n;
}
int main() {
  hipFuncSetAttribute((const void*)k_small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute((const void*)k_big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  Big b{}; 
  for (int blocks : {36, 225, 900}) for (size_t lds : {0, 16 * 1024, 91 * 1024, 160 * 1024}) {
    float a = timeit([&] { hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), lds, 0, nullptr); });
    float c = timeit([&] { hipLaunchKernelGGL(k_big, dim3(blocks), dim3(256), lds, 0, b, nullptr); });
    printf("blocks %4d  lds %6zu B : %.2f us per launch (8 B of arguments), %.2f us (648 B of arguments)\n", blocks, lds, a, c);
  }
  return 0;
}
