// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_degrader_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<DegraderConstant>(backward, solver, a, st);
}
int n_slots_degrader_constant() { return DegraderConstant::NSLOT; }
int n_states_degrader_constant() { return DegraderConstant::N; }
int n_cond_degrader_constant() { return DegraderConstant::NC; }
const char* slot_name_degrader_constant(int s) { return DegraderConstant::slot_name(s); }
}  // namespace vihds
