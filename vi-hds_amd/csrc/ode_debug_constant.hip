// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_debug_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<DebugConstant>(backward, solver, a, st);
}
int n_slots_debug_constant() { return DebugConstant::NSLOT; }
int n_states_debug_constant() { return DebugConstant::N; }
int n_cond_debug_constant() { return DebugConstant::NC; }
const char* slot_name_debug_constant(int s) { return DebugConstant::slot_name(s); }
}  // namespace vihds
