// dr_constant (version 2): thread-per-trajectory kernels for large batches, lane-split kernels (8 lanes per
// trajectory, vihds_dr_lanes.hpp) below VIHDS_LANE_SPLIT_MAX_N trajectories.
#include <cstdlib>
#include <string>

#include "vihds_ode_kernels.hpp"
#include "vihds_dr_lanes.hpp"
#include "vihds_dr_scan.hpp"

namespace vihds {
static int lane_split_max_n_v2() {
  static const int v = [] {
    const char* e = std::getenv("VIHDS_LANE_SPLIT_MAX_N");
    return e ? std::atoi(e) : 16384;
  }();
  return v;
}
int launch_dr_constant_v2(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  const bool lanes = a.kernel_variant == 2 || (a.kernel_variant == 0 && a.n <= lane_split_max_n_v2());
  // (the adaptive pairs and their step-size controller exist in the thread-per-trajectory kernels only)
  if (lanes && !solver_is_adaptive(solver) && !g_adaptive_ctl) return launch_dr_lanes<2>(backward, solver, a, st);
  return launch_ode<DrConstant<2>>(backward, solver, a, st);
}
// fused log-likelihood + unit-weight adjoint: the time-parallel kernel (vihds_dr_scan.hpp; any batch size, time grids up to
// 129 points).  VIHDS_E_UNSUPPORTED beyond that: the caller takes vihds_ode_fwd + vihds_ode_bwd.
int launch_dr_constant_train_v2(int solver, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts) {
  if ((a.kernel_variant & 0xff) == 1) return VIHDS_E_UNSUPPORTED;  // (one thread per trajectory asked for: no fused form)
  return launch_dr_scan_train<2>(solver, a, st, ts);
}
int n_slots_dr_constant_v2() { return DrConstant<2>::NSLOT; }
int n_states_dr_constant_v2() { return DrConstant<2>::N; }
int n_cond_dr_constant_v2() { return DrConstant<2>::NC; }
const char* slot_name_dr_constant_v2(int s) { return DrConstant<2>::slot_name(s); }
}  // namespace vihds
