// dr_blackbox kernels for the configuration of the reference's specs/dr_blackbox_icml.yaml:17-31
// (n_latent_species 2, n_hidden_decoder 25, n_hidden_decoder_precisions 20, n_z 5, n_x 5, n_y 2).
#include "vihds_ode_kernels.hpp"
#include "vihds_blackbox_split.hpp"

// Compiled TWICE (csrc/Makefile): VIHDS_BB_PART 1 = the forward launch of the cooperating-wavefront kernels alone, built
// with -fno-slp-vectorize (its VALU stream is faster unpacked: 99.0 -> 94.4 us at config 4); everything else -- the adjoint,
// which is faster with LLVM's packed fp32 (213.8 vs 226.6 us), the other variants, the tables -- as part 2 without the flag.
namespace vihds {
using BB = Blackbox<2, 25, 20, 5, 5, 2>;
#if defined(VIHDS_BB_PART) && VIHDS_BB_PART == 1
int launch_dr_blackbox_split_fwd(int solver, const OdeArgs& a, hipStream_t st) {
  return launch_bb_split_dir<BbMfma, false>(solver, a, st, g_theta_stage);
}
}  // namespace vihds
#ifdef VIHDS_BB_STAMPS
extern "C" int vihds_debug_bb_fwd_stamps(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(vihds::vihds_bb_stamp_buf), &buf, sizeof(buf));
}
#endif
#else
int launch_dr_blackbox_split_fwd(int solver, const OdeArgs& a, hipStream_t st);
int launch_dr_blackbox(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  // kernel_variant 1 = VALU, one thread per trajectory (vihds_blackbox.hpp); otherwise the MFMA formulation
  // (vihds_theta_ode_fwd: the sampling stage exists in the cooperating-wavefront forward only)
  if (g_theta_stage && (backward || a.kernel_variant == 1 || solver_is_adaptive(solver) || g_adaptive_ctl))
    return VIHDS_E_UNSUPPORTED;
  if (a.kernel_variant == 1 || solver_is_adaptive(solver) || g_adaptive_ctl) return launch_ode<BB>(backward, solver, a, st);
  // otherwise the two networks on two wavefronts and the Gram tiles on two more (vihds_blackbox_split.hpp)
  if (!backward) {
    const int rc = launch_dr_blackbox_split_fwd(solver, a, st);
    // (a time grid too long for the cooperating-wavefront forward's staged inputs: the thread-per-trajectory forward)
    return (rc == VIHDS_E_UNSUPPORTED && !g_theta_stage) ? launch_ode<BB>(false, solver, a, st) : rc;
  }
  return launch_bb_split_dir<BbMfma, true>(solver, a, st);
}
int n_slots_dr_blackbox() { return BB::NSLOT; }
int n_states_dr_blackbox() { return BB::N; }
int n_cond_dr_blackbox() { return BB::NC; }
const char* slot_name_dr_blackbox(int s) { return BB::slot_name(s); }
int bb_n_weights(int n_const) { return BB::n_weights(n_const); }
// kernel_variant 0 (anything but 1) with a fixed-grid solver: the matrix-core adjoint with the Gram tiles on chip
static bool bb_gram_mode(int solver, int kernel_variant) {
  return kernel_variant != 1 && !solver_is_adaptive(solver);
}
long long bb_aux_floats(int n, int T, int solver, int kernel_variant) {
  return (long long)bb_mfma_head_floats(n, T, solver, bb_gram_mode(solver, kernel_variant)) + (long long)BB::NTAIL * n;
}
long long bb_tail_offset_floats(int n, int T, int solver, int kernel_variant) {
  return (long long)bb_mfma_head_floats(n, T, solver, bb_gram_mode(solver, kernel_variant));
}
int bb_gram_on_chip(int solver, int kernel_variant) { return bb_gram_mode(solver, kernel_variant) ? 1 : 0; }
void bb_gram_reduce(const OdeArgs& a, const float* aux, float* g_weights, hipStream_t st) {
  launch_bb_gram_reduce<BbMfma>(a, aux, g_weights, st);
}
int bb_check(int L, int HS, int HP, int n_const, int C, int D) {
  return L == 2 && HS == 25 && HP == 20 && n_const == BB::NLAT + C + D;
}
int bb_dump_fields() { return BB::NF; }
}  // namespace vihds
#endif  // VIHDS_BB_PART
#ifdef VIHDS_BB_STAMPS
extern "C" int vihds_debug_bb_stamps(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(vihds::vihds_bb_stamp_buf), &buf, sizeof(buf));
}
#endif
