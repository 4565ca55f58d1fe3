// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_relay_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<RelayConstant>(backward, solver, a, st);
}
int n_slots_relay_constant() { return RelayConstant::NSLOT; }
int n_states_relay_constant() { return RelayConstant::N; }
int n_cond_relay_constant() { return RelayConstant::NC; }
const char* slot_name_relay_constant(int s) { return RelayConstant::slot_name(s); }
}  // namespace vihds
