// Instantiates the fused forward / adjoint ODE kernels for one model (one translation unit per model so the
// library builds in parallel).  Model definition: vihds_models.hpp.
#include "vihds_ode_kernels.hpp"
#include "vihds_relay_lanes.hpp"

namespace vihds {
int launch_relay_constant(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  if (g_theta_stage && g_adaptive_ctl)
    return VIHDS_E_UNSUPPORTED;  // (vihds_theta_ode_fwd: the sampling stage exists in the lane-split kernels only)
  // below 16 384 trajectories: sixteen lanes per trajectory (vihds_relay_lanes.hpp); the adaptive controller, a hidden
  // layer in the precision network and kernel_variant 1 keep one thread per trajectory.  (A backward that wants the
  // network's weight gradients brings the small per-block buffer of vihds_ode_bwd_aux_floats in `aux`.)
  if (!g_adaptive_ctl && relay_lanes_applicable(a.n, solver, a.kernel_variant, a.n_hidden_prec) &&
      !(backward && false && a.g_weights && !a.aux))
    return relay_lanes_launch<RlRelay, false>(backward, solver, a, st, g_theta_stage);
  if (g_theta_stage) return VIHDS_E_UNSUPPORTED;
  return launch_ode<RelayConstant>(backward, solver, a, st);
}
int n_slots_relay_constant() { return RelayConstant::NSLOT; }
int n_states_relay_constant() { return RelayConstant::N; }
int n_cond_relay_constant() { return RelayConstant::NC; }
const char* slot_name_relay_constant(int s) { return RelayConstant::slot_name(s); }
}  // namespace vihds
