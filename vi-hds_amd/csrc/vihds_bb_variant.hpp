// dr_blackbox at network sizes other than specs/dr_blackbox_icml.yaml (reference models/dr_blackbox.py:61-84 takes
// n_latent_species, n_hidden_decoder, n_hidden_decoder_precisions and n_z / n_x / n_y from the YAML).  The kernels are
// templates over the sizes (fully unrolled MLPs, state in registers), so every size set is its own build: the ICML
// sizes live in libvihds_hip.so (with the matrix-core formulation), any other set is a side library
//   libvihds_bb_<L>_<HS>_<HP>_<NLAT>.so      (make -C vi-hds_amd/csrc blackbox L=.. HS=.. HP=.. NLAT=..)
// next to it, holding the thread-per-trajectory kernels (vihds_blackbox.hpp) of every solver and -- for up to three latent
// species and 64 / 32 hidden units -- the matrix-core kernels (vihds_blackbox_split.hpp), loaded on first use
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
  // Matrix-core kernels on cooperating wavefronts with the weight gradients on chip (vihds_blackbox_split.hpp): built into
  // the side library for at most 3 latent species, 64 / 32 hidden units and 16 latent inputs -- the reference's
  // default n_hidden_decoder = 50 included.  `launch` then takes them for kernel_variant != 1 on a fixed-grid solver.
  int mfma;
  long long (*gram_floats)(int n);  // floats of aux ahead of the tail in that mode
  void (*gram_reduce)(const OdeArgs& a, const float* aux, float* g_weights, hipStream_t st);
};
}  // namespace vihds
extern "C" const vihds::BbVariant* vihds_bb_variant_v2(void);  // the one symbol a side library exports (_v2: the record grew)
