// dr_blackbox at network sizes other than specs/dr_blackbox_icml.yaml (reference models/dr_blackbox.py:61-84 takes
// n_latent_species, n_hidden_decoder, n_hidden_decoder_precisions and n_z / n_x / n_y from the YAML).  The kernels are
// templates over the sizes (fully unrolled MLPs, state in registers), so every size set is its own build: the ICML
// sizes live in libvihds_hip.so (with the matrix-core formulation), any other set is a side library
//   libvihds_bb_<L>_<HS>_<HP>_<NLAT>.so      (make -C vi-hds_amd/csrc blackbox L=.. HS=.. HP=.. NLAT=..)
// next to it, holding the thread-per-trajectory kernels (vihds_blackbox.hpp) of every solver, loaded on first use
// (vihds_api.hip: bb_lookup).  NLAT = n_z + n_x + n_y (the kernels only see the total: z, x, y are consecutive slots).
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_args.hpp"

namespace vihds {
struct AdaptiveCtl;
struct BbVariant {
  int L, HS, HP, NLAT;
  int n_states;     // 4 + L + 4
  int n_slots;      // NLAT + 4 (latents, then init_x / init_rfp / init_yfp / init_cfp)
  int dump_fields;  // fields per RHS evaluation in the adjoint's dump
  int n_tail;       // rows behind the dump: Delta (HS + HP), then the output-bias adjoint sums (2 NX + 8)
  int (*n_weights)(int n_const);
  // ctl != nullptr: run the step-size controller of an adaptive solver instead of the integration
  int (*launch)(bool backward, int solver, const OdeArgs& a, hipStream_t st, AdaptiveCtl* ctl);
};
}  // namespace vihds
extern "C" const vihds::BbVariant* vihds_bb_variant(void);  // the one symbol a side library exports
