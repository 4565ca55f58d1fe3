// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_relay_constant_prec(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<WithPrec<RelayConstant>>(backward, solver, a, st);
}
int n_slots_relay_constant_prec() { return WithPrec<RelayConstant>::NSLOT; }
int n_states_relay_constant_prec() { return WithPrec<RelayConstant>::N; }
int n_cond_relay_constant_prec() { return WithPrec<RelayConstant>::NC; }
const char* slot_name_relay_constant_prec(int s) { return WithPrec<RelayConstant>::slot_name(s); }
}  // namespace vihds
