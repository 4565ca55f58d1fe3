// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"
#include "vihds_relay_lanes.hpp"

namespace vihds {
int launch_degrader_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  if (g_theta_stage && g_adaptive_ctl)
    return VIHDS_E_UNSUPPORTED;  // (vihds_theta_ode_fwd: the sampling stage exists in the lane-split kernels only)
  // below 16 384 trajectories: one lane per state, sixteen lanes per trajectory (vihds_relay_lanes.hpp, RlDegrader); the adaptive
  // controller, a hidden layer in the precision network and kernel_variant 1 keep one thread per trajectory
  if (!g_adaptive_ctl && relay_lanes_applicable(a.n, solver, a.kernel_variant, a.n_hidden_prec) &&
      !(backward && false && a.g_weights && !a.aux))
    return relay_lanes_launch<RlDegrader, false>(backward, solver, a, st, g_theta_stage);
  if (g_theta_stage) return VIHDS_E_UNSUPPORTED;
  return launch_ode<DegraderConstant>(backward, solver, a, st);
}
int n_slots_degrader_constant() { return DegraderConstant::NSLOT; }
int n_states_degrader_constant() { return DegraderConstant::N; }
int n_cond_degrader_constant() { return DegraderConstant::NC; }
const char* slot_name_degrader_constant(int s) { return DegraderConstant::slot_name(s); }
}  // namespace vihds
