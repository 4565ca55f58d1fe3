// One dr_blackbox size set as a side library (see ../vihds_bb_variant.hpp); built with
//   -DVIHDS_BB_L=<n_latent_species> -DVIHDS_BB_HS=<n_hidden_decoder> -DVIHDS_BB_HP=<n_hidden_decoder_precisions>
//   -DVIHDS_BB_NLAT=<n_z + n_x + n_y>
// as ten objects compiled in parallel (Makefile, target `blackbox`): one per solver (-DVIHDS_ONLY_SOLVER=<id>: the
// kernels of that solver behind vihds_bb_launch_<id>) and the table object (no VIHDS_ONLY_SOLVER: the BbVariant
// record that dispatches to them).
#include "../vihds_ode_kernels.hpp"
#include "../vihds_bb_variant.hpp"
#include "../vihds_blackbox_split.hpp"

#if !defined(VIHDS_BB_L) || !defined(VIHDS_BB_HS) || !defined(VIHDS_BB_HP) || !defined(VIHDS_BB_NLAT)
#error "define VIHDS_BB_L, VIHDS_BB_HS, VIHDS_BB_HP and VIHDS_BB_NLAT"
#endif

namespace vihds {
using BBV = Blackbox<VIHDS_BB_L, VIHDS_BB_HS, VIHDS_BB_HP, VIHDS_BB_NLAT, 0, 0>;
typedef int (*bb_launch_fn)(bool, int, const OdeArgs&, hipStream_t, AdaptiveCtl*);
// the matrix-core formulation exists for this size set
constexpr bool BBV_MFMA = VIHDS_BB_L >= 1 && VIHDS_BB_L <= 3 && VIHDS_BB_HS <= 64 && VIHDS_BB_HP <= 32 && VIHDS_BB_NLAT <= 16;
using KV = BbMfmaT<BBV, BBV_MFMA ? VIHDS_BB_L : 2, BBV_MFMA ? VIHDS_BB_HS : 16, BBV_MFMA ? VIHDS_BB_HP : 16,
                   BBV_MFMA ? VIHDS_BB_NLAT : 16>;
__host__ inline bool bbv_takes_mfma(int solver, const OdeArgs& a, AdaptiveCtl* ctl) {
  return BBV_MFMA && !ctl && a.kernel_variant != 1 && solver >= VIHDS_SOLVER_MODEULER && solver <= VIHDS_SOLVER_RK4;
}
}  // namespace vihds

#define VIHDS_BB_CAT2(a, b) a##b
#define VIHDS_BB_CAT(a, b) VIHDS_BB_CAT2(a, b)

#ifdef VIHDS_ONLY_SOLVER
extern "C" int VIHDS_BB_CAT(vihds_bb_launch_, VIHDS_ONLY_SOLVER)(bool backward, int solver, const vihds::OdeArgs& a,
                                                                hipStream_t st, vihds::AdaptiveCtl* ctl) {
  using namespace vihds;
#if VIHDS_ONLY_SOLVER <= 4
  static_assert(VIHDS_SOLVER_RK4 == 4 && VIHDS_SOLVER_MODEULER == 0, "fixed-grid schemes are solvers 0..4");
  if constexpr (BBV_MFMA)
    if (bbv_takes_mfma(solver, a, ctl)) {
      const int rc = launch_bb_split_solver<KV, VIHDS_ONLY_SOLVER>(backward, a, st);
      if (rc != VIHDS_E_UNSUPPORTED) return rc;  // (forward with a time grid too long for its staged inputs: the VALU kernel)
    }
#endif
  g_adaptive_ctl = ctl;
  const int rc = launch_ode<BBV>(backward, solver, a, st);
  g_adaptive_ctl = nullptr;
  return rc;
}
#else
#define VIHDS_BB_DECL(k) \
  extern "C" int vihds_bb_launch_##k(bool, int, const vihds::OdeArgs&, hipStream_t, vihds::AdaptiveCtl*);
VIHDS_BB_DECL(0) VIHDS_BB_DECL(1) VIHDS_BB_DECL(2) VIHDS_BB_DECL(3) VIHDS_BB_DECL(4) VIHDS_BB_DECL(5) VIHDS_BB_DECL(6)
VIHDS_BB_DECL(7) VIHDS_BB_DECL(8)
static_assert(VIHDS_SOLVER_COUNT == 9, "one object per solver: extend the table and the Makefile");
namespace vihds {
thread_local AdaptiveCtl* g_adaptive_ctl = nullptr;  // this library's own (it does not link against libvihds_hip.so)
// (never set here: the device-resident adaptive solver does not serve models with shared neural weights, but launch_ode
// looks at it -- without this definition the library did not load: an undefined symbol only libvihds_hip.so has)
thread_local AdaptiveDevCtl* g_adaptive_dev = nullptr;
thread_local const SummArgs* g_summ = nullptr;  // (the same: vihds_ode_fwd_summaries does not serve dr_blackbox)
static int n_weights_sized(int n_const) { return BBV::n_weights(n_const); }
static long long gram_floats_sized(int n) { return BBV_MFMA ? (long long)KV::gram_floats(n) : -1; }
static void gram_reduce_sized(const OdeArgs& a, const float* aux, float* g_weights, hipStream_t st) {
  if constexpr (BBV_MFMA) launch_bb_gram_reduce<KV>(a, aux, g_weights, st);
}
static int launch_sized(bool backward, int solver, const OdeArgs& a, hipStream_t st, AdaptiveCtl* ctl) {
  static const bb_launch_fn table[VIHDS_SOLVER_COUNT] = {vihds_bb_launch_0, vihds_bb_launch_1, vihds_bb_launch_2,
                                                         vihds_bb_launch_3, vihds_bb_launch_4, vihds_bb_launch_5,
                                                         vihds_bb_launch_6, vihds_bb_launch_7, vihds_bb_launch_8};
  if (solver < 0 || solver >= VIHDS_SOLVER_COUNT) return VIHDS_E_BADARG;
  return table[solver](backward, solver, a, st, ctl);
}
}  // namespace vihds

extern "C" const vihds::BbVariant* vihds_bb_variant_v2(void) {
  using namespace vihds;
  static const BbVariant v = {VIHDS_BB_L, VIHDS_BB_HS, VIHDS_BB_HP, VIHDS_BB_NLAT, BBV::N, BBV::NSLOT,
                              BBV::NF,    BBV::NTAIL,  n_weights_sized, launch_sized,
                              BBV_MFMA ? 1 : 0, gram_floats_sized, gram_reduce_sized};
  return &v;
}
#endif
