// One dr_blackbox size set as a side library (see ../vihds_bb_variant.hpp); built with
//   -DVIHDS_BB_L=<n_latent_species> -DVIHDS_BB_HS=<n_hidden_decoder> -DVIHDS_BB_HP=<n_hidden_decoder_precisions>
//   -DVIHDS_BB_NLAT=<n_z + n_x + n_y>
#include "../vihds_ode_kernels.hpp"
#include "../vihds_bb_variant.hpp"

#if !defined(VIHDS_BB_L) || !defined(VIHDS_BB_HS) || !defined(VIHDS_BB_HP) || !defined(VIHDS_BB_NLAT)
#error "define VIHDS_BB_L, VIHDS_BB_HS, VIHDS_BB_HP and VIHDS_BB_NLAT"
#endif

namespace vihds {
thread_local AdaptiveCtl* g_adaptive_ctl = nullptr;  // this library's own copy (it does not link against libvihds_hip.so)
using BBV = Blackbox<VIHDS_BB_L, VIHDS_BB_HS, VIHDS_BB_HP, VIHDS_BB_NLAT, 0, 0>;
static int n_weights_sized(int n_const) { return BBV::n_weights(n_const); }
static int launch_sized(bool backward, int solver, const OdeArgs& a, hipStream_t st, AdaptiveCtl* ctl) {
  g_adaptive_ctl = ctl;
  const int rc = launch_ode<BBV>(backward, solver, a, st);
  g_adaptive_ctl = nullptr;
  return rc;
}
}  // namespace vihds

extern "C" const vihds::BbVariant* vihds_bb_variant(void) {
  using namespace vihds;
  static const BbVariant v = {VIHDS_BB_L, VIHDS_BB_HS, VIHDS_BB_HP, VIHDS_BB_NLAT, BBV::N, BBV::NSLOT,
                              BBV::NF,    BBV::NTAIL,  n_weights_sized, launch_sized};
  return &v;
}
