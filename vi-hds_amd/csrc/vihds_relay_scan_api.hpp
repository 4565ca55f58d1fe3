// What the model translation units and the C ABI need of the time-parallel kernels (vihds_relay_scan.hpp) without
// compiling them: applicability and buffer sizes.
#pragma once
#include "vihds_relay_lanes.hpp"

namespace vihds {
constexpr int RS_TPB = 4;         // trajectories per block: 2 per wavefront, 32 lanes each (12 KB of adjoint records each)
constexpr int RS_T = 32 * RS_TPB;
constexpr int RS_MAX_ITEMS = 8;   // steps per lane: T <= 257

__host__ __device__ inline bool relay_scan_applicable(int T, int solver, int kernel_variant, int n_hidden_prec) {
  return kernel_variant == 5 && T >= 2 && T - 1 <= 32 * RS_MAX_ITEMS && solver >= VIHDS_SOLVER_MODEULER &&
         solver <= VIHDS_SOLVER_RK4 && n_hidden_prec < 1;
}
__host__ __device__ inline int relay_scan_items(int T) { return (T - 1 + 31) / 32; }
// aux of the adjoint: one weight-gradient partial row per block, then Lambda of the four precision states at the end of
// every step [4][n][ITEMS][32]
__host__ __device__ inline long long relay_scan_aux_floats(int n, int T, int n_species) {
  return (long long)((n + RS_TPB - 1) / RS_TPB) * rl_nwg(1 + n_species) + 4LL * n * relay_scan_items(T) * 32;
}
}  // namespace vihds
