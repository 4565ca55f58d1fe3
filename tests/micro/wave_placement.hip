// Probe: where the dispatcher puts the wavefronts of a launch shaped like the dr_blackbox kernels (450 blocks of 2 or 4
// wavefronts, the adjoint with 9 KB of LDS): CU and SIMD of every wavefront (HW_ID / XCC_ID), and which blocks share a CU.
// Build/run: hipcc --offload-arch=gfx950 -O3 tests/micro/wave_placement.hip -o /tmp/wave_placement && /tmp/wave_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <string>
#include <vector>
#include <algorithm>
struct Rec { unsigned hw, xcc; unsigned long long t0, t1; };
__global__ void k(Rec* out, int spin, int lds_floats) {
  extern __shared__ float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned long long t0 = (unsigned long long)wall_clock64();
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
  float x = threadIdx.x;
  if (lds_floats) lds[threadIdx.x] = x;
  while ((unsigned long long)wall_clock64() - t0 < (unsigned long long)spin) x = fmaf(x, 1.0001f, 0.5f);
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = {hw, xcc, t0, (unsigned long long)wall_clock64()};
  if (x == 12345.678f) out[0].hw = 0;
}
static void run(int blocks, int waves, int lds_bytes) {
  Rec* d;
  hipMalloc(&d, sizeof(Rec) * blocks * waves);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), lds_bytes, 0, d, 2000 /* 20 us at 100 MHz */, lds_bytes / 4);
  hipDeviceSynchronize();
  std::vector<Rec> r(blocks * waves);
  hipMemcpy(r.data(), d, sizeof(Rec) * r.size(), hipMemcpyDeviceToHost);
  unsigned long long tmin = ~0ull;
  for (auto& x : r) tmin = std::min(tmin, x.t0);
  // gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  std::map<unsigned, std::vector<int>> cu_blocks;
  auto cu_key = [&](const Rec& x) { return ((x.xcc & 0xf) << 16) | (((x.hw >> 13) & 7) << 8) | (((x.hw >> 12) & 1) << 4) | ((x.hw >> 8) & 0xf); };
  int split_blocks = 0;
  for (int b = 0; b < blocks; ++b) {
    unsigned key = cu_key(r[b * waves]);
    for (int w = 1; w < waves; ++w) if (cu_key(r[b * waves + w]) != key) { ++split_blocks; break; }
    cu_blocks[key].push_back(b);
  }
  std::map<int, int> hist;
  for (auto& kv : cu_blocks) hist[(int)kv.second.size()]++;
  printf("== %d blocks x %d waves, %d B LDS: %zu CUs used, blocks with waves on different CUs: %d\n", blocks, waves, lds_bytes, cu_blocks.size(), split_blocks);
  for (auto& kv : hist) printf("   CUs holding %d block(s): %d\n", kv.first, kv.second);
  int late = 0;
  for (int b = 0; b < blocks; ++b) if (r[b * waves].t0 - tmin > 1000) ++late;
  printf("   blocks that started > 10 us after the first (waited for a slot): %d\n", late);
  // SIMD of wave w, and for CUs with two blocks: how the second block's waves sit relative to the first's
  std::map<std::string, int> pat1, pat2;
  int shown = 0, partner256 = 0, two = 0;
  for (auto& kv : cu_blocks) {
    char buf[128];
    for (size_t j = 0; j < kv.second.size(); ++j) {
      int b = kv.second[j], n = 0;
      for (int w = 0; w < waves; ++w) n += snprintf(buf + n, sizeof(buf) - n, "%u", (r[b * waves + w].hw >> 4) & 3);
      (j == 0 ? pat1 : pat2)[buf]++;
    }
    if (kv.second.size() == 2) {
      ++two;
      if (std::abs(kv.second[0] - kv.second[1]) == 256) ++partner256;
      if (shown < 6) {
        printf("   CU %05x: blocks %d and %d, SIMDs of waves:", kv.first, kv.second[0], kv.second[1]);
        for (int j = 0; j < 2; ++j) { printf("  ["); for (int w = 0; w < waves; ++w) printf("%u", (r[kv.second[j] * waves + w].hw >> 4) & 3); printf("]"); }
        printf("\n");
        ++shown;
      }
    }
  }
  printf("   CUs with two blocks: %d, of which blocks (j, j+256): %d\n", two, partner256);
  printf("   SIMD pattern of a CU's first block:"); for (auto& kv : pat1) printf(" %s x%d", kv.first.c_str(), kv.second); printf("\n");
  printf("   SIMD pattern of a CU's later blocks:"); for (auto& kv : pat2) printf(" %s x%d", kv.first.c_str(), kv.second); printf("\n");
  hipFree(d);
}
int main() {
  run(450, 2, 1024);       // bb_split_fwd_kernel at config 4
  run(450, 2, 16 * 1024);  // ... with the sampling stage's scratch
  run(450, 4, 9 * 1024);   // bb_split_bwd_kernel
  run(225, 4, 9 * 1024);
  run(450, 4, 84 * 1024);  // (one block per CU by LDS)
  run(900, 4, 9 * 1024);
  run(2250, 4, 9 * 1024);
  return 0;
}
