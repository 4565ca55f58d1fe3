// Time-parallel kernels (vihds_relay_scan.hpp, kernel_variant 5) of degrader_constant_precisions: a translation unit of its own so that the
// library keeps building in parallel.
#include "vihds_relay_scan.hpp"

namespace vihds {
int launch_scan_degrader_constant_prec(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return relay_scan_launch<RlDegrader, true>(backward, solver, a, st);
}
}  // namespace vihds
