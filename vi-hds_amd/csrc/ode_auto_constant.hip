// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_auto_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<AutoConstant>(backward, solver, a, st);
}
int n_slots_auto_constant() { return AutoConstant::NSLOT; }
int n_states_auto_constant() { return AutoConstant::N; }
int n_cond_auto_constant() { return AutoConstant::NC; }
const char* slot_name_auto_constant(int s) { return AutoConstant::slot_name(s); }
}  // namespace vihds
