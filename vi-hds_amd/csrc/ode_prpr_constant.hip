// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_prpr_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<PrprConstant>(backward, solver, a, st);
}
int n_slots_prpr_constant() { return PrprConstant::NSLOT; }
int n_states_prpr_constant() { return PrprConstant::N; }
int n_cond_prpr_constant() { return PrprConstant::NC; }
const char* slot_name_prpr_constant(int s) { return PrprConstant::slot_name(s); }
}  // namespace vihds
