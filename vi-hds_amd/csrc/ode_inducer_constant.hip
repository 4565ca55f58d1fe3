// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_inducer_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<InducerConstant>(backward, solver, a, st);
}
int n_slots_inducer_constant() { return InducerConstant::NSLOT; }
int n_states_inducer_constant() { return InducerConstant::N; }
int n_cond_inducer_constant() { return InducerConstant::NC; }
const char* slot_name_inducer_constant(int s) { return InducerConstant::slot_name(s); }
}  // namespace vihds
