// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"

namespace vihds {
int launch_dr_constant_prec_v2(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  return launch_ode<WithPrec<DrConstant<2>>>(backward, solver, a, st);
}
int n_slots_dr_constant_prec_v2() { return WithPrec<DrConstant<2>>::NSLOT; }
int n_states_dr_constant_prec_v2() { return WithPrec<DrConstant<2>>::N; }
int n_cond_dr_constant_prec_v2() { return WithPrec<DrConstant<2>>::NC; }
const char* slot_name_dr_constant_prec_v2(int s) { return WithPrec<DrConstant<2>>::slot_name(s); }
}  // namespace vihds
