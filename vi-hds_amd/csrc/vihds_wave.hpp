// Wavefront-wide sums / maxima on the DPP path (no LDS traffic, unlike __shfl_*'s ds_bpermute): an inclusive scan in six
// VALU steps whose total is read from lane 63 and returned uniformly.  The summation tree differs from a shuffle-down
// tree, so a kernel uses one or the other consistently wherever results are compared bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace vihds {

// DPP steps (gfx9 encodings): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {  // lanes without a source (or masked off) read 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
// sum over the wavefront, returned uniformly: an inclusive scan in six DPP steps, total read from lane 63
__device__ __forceinline__ float wave_total(float v) {
  v += dpp_f<0x111, 0xf>(v);
  v += dpp_f<0x112, 0xf>(v);
  v += dpp_f<0x114, 0xf>(v);
  v += dpp_f<0x118, 0xf>(v);
  v += dpp_f<0x142, 0xa>(v);
  v += dpp_f<0x143, 0xc>(v);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_keep(float v) {  // lanes without a source (or masked off) read their own value
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL,
                                                               ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_max_total(float v) {
  v = fmaxf(v, dpp_keep<0x111, 0xf>(v));
  v = fmaxf(v, dpp_keep<0x112, 0xf>(v));
  v = fmaxf(v, dpp_keep<0x114, 0xf>(v));
  v = fmaxf(v, dpp_keep<0x118, 0xf>(v));
  v = fmaxf(v, dpp_keep<0x142, 0xa>(v));
  v = fmaxf(v, dpp_keep<0x143, 0xc>(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Which run of trajectories a workgroup takes, XCD-aware.  The dispatcher hands consecutive workgroup ids to the chip's eight XCDs
// in turn (id % 8), and every XCD has its own L2.  A kernel whose neighbouring workgroups touch the two halves of the same
// 128-byte lines (sixteen trajectories x 4 bytes per row: the lane kernels of vihds_relay_lanes.hpp) then fetches every line
// into two L2s -- measured as twice the algorithmic bytes from HBM (round 5, PMC).  With this map the workgroups of ONE XCD
// take a contiguous range of logical blocks, so the two halves of a line meet in one L2.  A bijection on [0, nblk).
__device__ __forceinline__ int xcd_block(int wg, int nblk) {
  constexpr int NXCD = 8;
  const int x = wg % NXCD, local = wg / NXCD;
  const int q = nblk / NXCD, r = nblk % NXCD;
  return x * q + (x < r ? x : r) + local;
}

}  // namespace vihds
