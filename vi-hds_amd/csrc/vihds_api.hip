// C ABI (include/vihds_hip.h): argument checking, model registry and dispatch to the kernel launchers.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>

#include "../../include/vihds_hip.h"
#include "vihds_ode_kernels.hpp"
#include "vihds_bb_variant.hpp"

#include "vihds_relay_lanes.hpp"
#include "vihds_relay_lanes.hpp"

namespace vihds {
thread_local AdaptiveCtl* g_adaptive_ctl = nullptr;
thread_local AdaptiveDevCtl* g_adaptive_dev = nullptr;
thread_local const ThetaStageArgs* g_theta_stage = nullptr;
thread_local const SummArgs* g_summ = nullptr;
// per-model translation units (ode_<model>.hip)
#define VIHDS_DECL(name)                                                        \
  int launch_##name(bool backward, int solver, const OdeArgs& a, hipStream_t st); \
  int n_slots_##name();                                                         \
  int n_states_##name();                                                        \
  int n_cond_##name();                                                          \
  const char* slot_name_##name(int s);
VIHDS_DECL(dr_constant_v1)
VIHDS_DECL(dr_constant_v2)
VIHDS_DECL(auto_constant)
VIHDS_DECL(prpr_constant)
VIHDS_DECL(relay_constant)
VIHDS_DECL(degrader_constant)
VIHDS_DECL(dr_constant_prec_v1)
VIHDS_DECL(dr_constant_prec_v2)
VIHDS_DECL(auto_constant_prec)
VIHDS_DECL(inducer_constant)
VIHDS_DECL(inducer_constant_prec)
VIHDS_DECL(debug_constant)
VIHDS_DECL(prpr_constant_prec)
VIHDS_DECL(relay_constant_prec)
VIHDS_DECL(degrader_constant_prec)
VIHDS_DECL(dr_blackbox)
#undef VIHDS_DECL
int bb_n_weights(int n_const);
long long bb_aux_floats(int n, int T, int solver, int kernel_variant);
long long bb_tail_offset_floats(int n, int T, int solver, int kernel_variant);
int bb_gram_on_chip(int solver, int kernel_variant);
void bb_gram_reduce(const OdeArgs& a, const float* aux, float* g_weights, hipStream_t st);
int bb_check(int L, int HS, int HP, int n_const, int C, int D);
int bb_dump_fields();

int launch_dr_constant_train_v1(int, const OdeArgs&, hipStream_t, const ThetaStageArgs*);
int launch_dr_constant_train_v2(int, const OdeArgs&, hipStream_t, const ThetaStageArgs*);
// vihds_elbo.hip
void launch_theta_fwd(int, int, int, const int*, const float*, const float*, const float*, const float*, const float*,
                      const float*, float*, float*, float*, float*, const vihds_theta_opts&, hipStream_t);
void launch_theta_bwd(int, int, int, const int*, const float*, const float*, const float*, const float*, const float*,
                      const float*, const float*, const float*, const float*, const float*, float*, float*,
                      const vihds_theta_opts&, hipStream_t);
void launch_iwae_fwd(int, int, const float*, const float*, const float*, float*, float*, float*, hipStream_t);
void launch_iwae_bwd(int, int, const float*, const float*, const float*, float*, hipStream_t);
void launch_iwae_finish(int, float, const float*, const float*, float*, float*, hipStream_t);
void launch_iwae_loss_rows(int, int, float, const float*, const float*, const float*, float*, float*, float*, float*,
                           float*, float*, float*, unsigned int*, hipStream_t);
void launch_iwae_loss_small(int, int, float, const float*, const float*, const float*, float*, float*, float*, float*,
                            float*, float*, float*, hipStream_t);
void launch_iwae_combine(int, int, int, float, const float*, const float*, float*, float*, float*, float*, hipStream_t);
void launch_iwae_loss_bwd(int, int, const float*, const float*, const float*, float*, float*, hipStream_t);
void launch_device_condition(int, int, int, int, int, int, float, float, const float*, unsigned int*, const float*, const float*, const int*,
                             float*, hipStream_t);
void launch_rng_advance(unsigned int* rng, hipStream_t st);
// vihds_offset.hip
void launch_gather_batch(int, int, int, int, int, int, const long long*, const float*, const float*, const float*, float*, float*,
                         float*, float*, hipStream_t);
void launch_offset_rows_fwd(int, int, int, int, int, int, const float*, const float*, const float*, float*, hipStream_t);
void launch_offset_rows_bwd(int, int, int, int, int, int, int, const float*, float*, float*, hipStream_t);
// vihds_gram.hip
long long gram_scratch_floats(long long, int, const vihds_gram_rect*);
int launch_gram(int, long long, int, const vihds_gram_rect*, const float*, float*, float*, hipStream_t);
int launch_bb_tail(int, int, int, int, int, int, int, const int*, const float*, const float*, const float*, const float*,
                   const int*, float*, hipStream_t);
// vihds_encoder.hip
size_t encoder_fwd_lds_bytes(const vihds_encoder_shape&);
size_t encoder_bwd_lds_bytes(const vihds_encoder_shape&);
void launch_encoder_fwd(const vihds_encoder_shape&, const float*, const float*, const float*, const float*, const float*,
                        const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                        float*, float*, float*, hipStream_t);
void launch_encoder_bwd(const vihds_encoder_shape&, const float*, const float*, const float*, const float*, const float*,
                        const float*, const float*, const float*, float*, float*, float*, float*, float*, float*, float*,
                        float*, float*, float*, hipStream_t);
bool step_tail_supported(const vihds_encoder_shape&, int, int);
void launch_step_tail(const vihds_encoder_shape&, const vihds_step_tail_args&, hipStream_t);
void launch_adam(const vihds_adam_tensors&, float*, float*, float*, const float*, float, float, float, float, float, const float*,
                 hipStream_t);
extern int iw_summaries_tpb_override;
void launch_iw_summaries(int, int, int, int, int, const float*, const float*, const float*, const float*, int,
                         const float*, const int*, float*, float*, float*, float*, hipStream_t);

struct ModelEntry {
  int (*launch)(bool, int, const OdeArgs&, hipStream_t);
  int (*n_slots)();
  int (*n_states)();
  int (*n_cond)();  // treatments the model reads per data row (cond[b*C + q], q < n_cond)
  const char* (*slot_name)(int);
  bool neural_prec;
};
#define VIHDS_ENTRY(name, np) \
  { launch_##name, n_slots_##name, n_states_##name, n_cond_##name, slot_name_##name, np }
static const ModelEntry kModels[VIHDS_MODEL_COUNT] = {
    VIHDS_ENTRY(dr_constant_v1, false),     // VIHDS_MODEL_DR_CONSTANT
    VIHDS_ENTRY(dr_constant_v2, false),     // VIHDS_MODEL_DR_CONSTANT_V2
    VIHDS_ENTRY(auto_constant, false),      // VIHDS_MODEL_AUTO_CONSTANT
    VIHDS_ENTRY(prpr_constant, false),      // VIHDS_MODEL_PRPR_CONSTANT
    VIHDS_ENTRY(relay_constant, false),     // VIHDS_MODEL_RELAY_CONSTANT
    VIHDS_ENTRY(degrader_constant, false),  // VIHDS_MODEL_DEGRADER_CONSTANT
    VIHDS_ENTRY(dr_constant_prec_v1, true),     // VIHDS_MODEL_DR_CONSTANT_PRECISIONS
    VIHDS_ENTRY(dr_constant_prec_v2, true),     // VIHDS_MODEL_DR_CONSTANT_PRECISIONS_V2
    VIHDS_ENTRY(auto_constant_prec, true),      // VIHDS_MODEL_AUTO_CONSTANT_PRECISIONS
    VIHDS_ENTRY(prpr_constant_prec, true),      // VIHDS_MODEL_PRPR_CONSTANT_PRECISIONS
    VIHDS_ENTRY(relay_constant_prec, true),     // VIHDS_MODEL_RELAY_CONSTANT_PRECISIONS
    VIHDS_ENTRY(degrader_constant_prec, true),  // VIHDS_MODEL_DEGRADER_CONSTANT_PRECISIONS
    VIHDS_ENTRY(dr_blackbox, true),             // VIHDS_MODEL_DR_BLACKBOX
    VIHDS_ENTRY(inducer_constant, false),       // VIHDS_MODEL_INDUCER_CONSTANT
    VIHDS_ENTRY(inducer_constant_prec, true),   // VIHDS_MODEL_INDUCER_CONSTANT_PRECISIONS
    VIHDS_ENTRY(debug_constant, false),         // VIHDS_MODEL_DEBUG_CONSTANT
};

static thread_local char g_err[256] = "";
static unsigned int* g_newton_hist = nullptr;  // vihds_debug_newton_hist: telemetry buffer of the time-parallel decoder kernel
static int fail(int code, const char* msg) {
  std::snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int check_hip(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    std::snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return VIHDS_E_HIP;
  }
  return VIHDS_OK;
}
static const char* kPrecNames[4] = {"prec_x", "prec_rfp", "prec_yfp", "prec_cfp"};

static const ModelEntry* entry(int model) {
  if (model < 0 || model >= VIHDS_MODEL_COUNT) return nullptr;
  return kModels[model].launch ? &kModels[model] : nullptr;
}

// ---- dr_blackbox size sets (vihds_bb_variant.hpp) --------------------------------------------------------------------
// The ICML sizes are built in (bb_check); any other set is looked for as libvihds_bb_<L>_<HS>_<HP>_<NLAT>.so in the
// directory this library was loaded from, once per process.
static bool bb_sizes_ok(const vihds_ode_problem* p) {
  return p->n_latent_states >= 0 && p->n_hidden_states > 0 && p->n_hidden_prec > 0 && p->n_const >= p->C + p->D &&
         p->C >= 0 && p->D >= 0;
}
static bool bb_builtin(const vihds_ode_problem* p) {
  return bb_check(p->n_latent_states, p->n_hidden_states, p->n_hidden_prec, p->n_const, p->C, p->D) != 0;
}
static const BbVariant* bb_sized(const vihds_ode_problem* p) {
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, int>, const BbVariant*> loaded;
  if (!bb_sizes_ok(p)) {
    fail(VIHDS_E_BADARG, "dr_blackbox: bad network sizes (n_const must cover the C treatments and the D-wide one-hot)");
    return nullptr;
  }
  const int nlat = p->n_const - p->C - p->D;
  const auto key = std::make_tuple(p->n_latent_states, p->n_hidden_states, p->n_hidden_prec, nlat);
  std::lock_guard<std::mutex> lock(mu);
  auto it = loaded.find(key);
  if (it != loaded.end()) return it->second;
  char name[96];
  std::snprintf(name, sizeof(name), "libvihds_bb_%d_%d_%d_%d.so", p->n_latent_states, p->n_hidden_states,
                p->n_hidden_prec, nlat);
  std::string path = name;
  Dl_info info;
  if (dladdr((const void*)&vihds_abi_version, &info) && info.dli_fname) {
    const std::string self = info.dli_fname;
    const size_t cut = self.rfind('/');
    if (cut != std::string::npos) path = self.substr(0, cut + 1) + name;
  }
  const BbVariant* v = nullptr;
  if (void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) {
    typedef const BbVariant* (*entry_fn)(void);
    if (entry_fn f = (entry_fn)dlsym(h, "vihds_bb_variant_v2")) v = f();
    if (v && (v->L != p->n_latent_states || v->HS != p->n_hidden_states || v->HP != p->n_hidden_prec || v->NLAT != nlat))
      v = nullptr;
  }
  if (!v) {
    std::snprintf(g_err, sizeof(g_err),
                  "dr_blackbox at n_latent_species=%d n_hidden_decoder=%d n_hidden_decoder_precisions=%d n_z+n_x+n_y=%d: "
                  "%s not found (make -C vi-hds_amd/csrc blackbox L=%d HS=%d HP=%d NLAT=%d)",
                  p->n_latent_states, p->n_hidden_states, p->n_hidden_prec, nlat, name, p->n_latent_states,
                  p->n_hidden_states, p->n_hidden_prec, nlat);
    return nullptr;  // (not cached: the file may be built later in this process)
  }
  loaded[key] = v;
  return v;
}
// a side library's matrix-core kernels with the weight gradients on chip apply (same rule as its launch function)
static bool bb_sized_gram(const BbVariant* v, const vihds_ode_problem* p) {
  return v->mfma && p->kernel_variant != 1 && p->solver >= VIHDS_SOLVER_MODEULER && p->solver <= VIHDS_SOLVER_RK4;
}
static long long bb_sized_dump_floats(const BbVariant* v, const vihds_ode_problem* p) {
  return (long long)(p->T - 1) * ode_stages(p->solver) * v->dump_fields * p->B * p->S;
}

static int build_args(const vihds_ode_problem* p, const ModelEntry* e, OdeArgs& a, const BbVariant* v = nullptr) {
  if (p->B <= 0 || p->S <= 0 || p->T < 2) return fail(VIHDS_E_BADARG, "B, S must be > 0 and T >= 2");
  if ((long long)p->B * p->S > 0x7fffffffLL) return fail(VIHDS_E_BADARG, "B*S exceeds int range");
  if (p->C < e->n_cond()) return fail(VIHDS_E_BADARG, "the model reads more treatments per row than C provides");
  if (p->model == VIHDS_MODEL_DR_BLACKBOX && p->n_const < p->C + p->D)
    return fail(VIHDS_E_BADARG, "dr_blackbox: n_const must cover the C treatments and the D-wide device one-hot");
  const int ns = v ? v->n_slots : e->n_slots() + (e->neural_prec ? 0 : 4);
  std::memset(&a, 0, sizeof(a));
  a.B = p->B; a.S = p->S; a.T = p->T; a.C = p->C; a.n = p->B * p->S;
  a.solver = p->solver; a.kernel_variant = p->kernel_variant; a.logp_grad_broadcast = p->logp_grad_broadcast; a.D = p->D; a.n_const = p->n_const; a.init_latent = p->init_latent; a.init_prec = p->init_prec;
  a.n_hidden_prec = (e->neural_prec && p->model != VIHDS_MODEL_DR_BLACKBOX && p->n_hidden_prec > 0) ? p->n_hidden_prec : 0;
  for (int q = 0; q < ns; ++q) {
    if (p->slot_row[q] < 0 || p->slot_row[q] >= p->n_rows) return fail(VIHDS_E_BADARG, "slot_row out of range");
    a.slot_row[q] = p->slot_row[q];
  }
  return VIHDS_OK;
}
}  // namespace vihds

using namespace vihds;

extern "C" {

int vihds_abi_version(void) { return VIHDS_ABI_VERSION; }
int vihds_debug_newton_hist(unsigned int* hist) {
  g_newton_hist = hist;
  return VIHDS_OK;
}
const char* vihds_last_error(void) { return g_err; }

int vihds_model_n_states(int model) {
  const ModelEntry* e = entry(model);
  return e ? e->n_states() : VIHDS_E_UNSUPPORTED;
}
int vihds_model_n_species(int model) {
  const ModelEntry* e = entry(model);
  return e ? e->n_states() - (e->neural_prec ? 4 : 0) : VIHDS_E_UNSUPPORTED;
}
int vihds_model_n_slots(int model) {
  const ModelEntry* e = entry(model);
  return e ? e->n_slots() + (e->neural_prec ? 0 : 4) : VIHDS_E_UNSUPPORTED;
}
const char* vihds_model_slot_name(int model, int slot) {
  const ModelEntry* e = entry(model);
  if (!e || slot < 0) return nullptr;
  const int ns = e->n_slots();
  if (slot < ns) return e->slot_name(slot);
  if (!e->neural_prec && slot < ns + 4) return kPrecNames[slot - ns];
  return nullptr;
}
int vihds_model_n_weights(const vihds_ode_problem* p) {
  if (!p) return VIHDS_E_BADARG;
  const ModelEntry* e = entry(p->model);
  if (!e) return VIHDS_E_UNSUPPORTED;
  if (!e->neural_prec) return 0;
  if (p->model == VIHDS_MODEL_DR_BLACKBOX) {
    if (bb_builtin(p)) return bb_n_weights(p->n_const);
    const BbVariant* v = bb_sized(p);
    return v ? v->n_weights(p->n_const) : VIHDS_E_UNSUPPORTED;
  }
  const int n_in = e->n_states() - 4 + 1;
  const int H = p->n_hidden_prec;
  if (H > 0) return H * n_in + H + 2 * (4 * H + 4);  // hidden layer (reference precisions.py:63-74): Wh, bh, Wp, bp, Wd, bd
  return 2 * (4 * n_in + 4);
}

static vihds_theta_opts theta_opts(const vihds_theta_opts* o) {
  vihds_theta_opts d = {nullptr, 0, nullptr, 0, 0, nullptr, nullptr};
  return o ? *o : d;
}

int vihds_ode_logp_grad(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                        const float* times, const float* obs, float* logp, float* g_theta_unit, void* stream) {
  if (!p || !theta || !cond || !times || !obs || !logp || !g_theta_unit) return fail(VIHDS_E_BADARG, "null argument");
  if (p->model != VIHDS_MODEL_DR_CONSTANT && p->model != VIHDS_MODEL_DR_CONSTANT_V2)
    return fail(VIHDS_E_UNSUPPORTED, "fused log-likelihood + adjoint exists for dr_constant / dr_constant_v2 only");
  if (p->solver < 0 || p->solver > VIHDS_SOLVER_RK4) return fail(VIHDS_E_BADARG, "unknown solver (the fused training kernels take the fixed-grid schemes)");
  const ModelEntry* e = entry(p->model);
  OdeArgs a;
  if (int rc = build_args(p, e, a)) return rc;
  if (p->C < 2) return fail(VIHDS_E_BADARG, "dr_constant needs two treatments");
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = obs;
  a.logp = logp; a.g_theta = g_theta_unit;
  a.newton_hist = g_newton_hist;
  const int rc = p->model == VIHDS_MODEL_DR_CONSTANT
                     ? launch_dr_constant_train_v1(p->solver, a, (hipStream_t)stream, nullptr)
                     : launch_dr_constant_train_v2(p->solver, a, (hipStream_t)stream, nullptr);
  if (rc == VIHDS_E_UNSUPPORTED)
    return fail(rc, "shape outside the fused kernel's regime (lane-split size limit, or time grid too long for LDS)");
  if (rc != VIHDS_OK) return fail(rc, "fused kernel launch failed");
  return check_hip("vihds_ode_logp_grad launch");
}

int vihds_theta_ode_logp_grad(const vihds_ode_problem* p, int P, const int* kind, const float* q_mu,
                              const float* q_prec, const float* p_mu, const float* p_prec, const float* clip_lo,
                              const float* clip_hi, float* u, const vihds_theta_opts* opts,
                              const vihds_conditioner* co, const float* cond, const float* dev1hot, const float* times,
                              const float* obs, float* theta, float* log_q, float* log_p, float* logp,
                              float* g_theta_unit, void* stream) {
  if (!p || !kind || !q_mu || !q_prec || !p_mu || !p_prec || !clip_lo || !clip_hi || !u || !cond || !times || !obs ||
      !theta || !logp || !g_theta_unit)
    return fail(VIHDS_E_BADARG, "null argument");
  if (p->model != VIHDS_MODEL_DR_CONSTANT && p->model != VIHDS_MODEL_DR_CONSTANT_V2)
    return fail(VIHDS_E_UNSUPPORTED, "the fused decoder step exists for dr_constant / dr_constant_v2 only");
  if (p->solver < 0 || p->solver > VIHDS_SOLVER_RK4) return fail(VIHDS_E_BADARG, "unknown solver (the fused training kernels take the fixed-grid schemes)");
  if (P <= 0 || P > p->n_rows) return fail(VIHDS_E_BADARG, "P out of range");
  if (opts && co && opts->rng && opts->rng == co->rng)
    return fail(VIHDS_E_BADARG, "the sampling stage and the conditioner need separate generator states");
  const ModelEntry* e = entry(p->model);
  OdeArgs a;
  if (int rc = build_args(p, e, a)) return rc;
  if (p->C < 2) return fail(VIHDS_E_BADARG, "dr_constant needs two treatments");
  const vihds_theta_opts o = theta_opts(opts);
  ThetaStageArgs t;
  std::memset(&t, 0, sizeof(t));
  t.P = P; t.kind = kind; t.q_mu = q_mu; t.q_prec = q_prec; t.q_rows = o.q_rows; t.prec_is_log = o.q_prec_is_log;
  t.p_mu = p_mu; t.p_prec = p_prec; t.clip_lo = clip_lo; t.clip_hi = clip_hi; t.u = u; t.rng = o.rng;
  t.S_total = p->S; t.s_off = 0;
  if (o.S_total > 0) {
    if (o.s_offset < 0 || o.s_offset + p->S > o.S_total) return fail(VIHDS_E_BADARG, "bad sample window");
    t.S_total = o.S_total; t.s_off = o.s_offset;
  }
  t.theta = theta; t.n_rows = p->n_rows; t.log_q = log_q; t.log_p = log_p;
  if (co && co->E > 0) {
    if (!dev1hot || !co->relevance || !co->is_default || (!co->z && !co->rng) || p->D <= 0)
      return fail(VIHDS_E_BADARG, "conditioner: missing input");
    if (co->first_row < P || co->first_row + co->E > p->n_rows) return fail(VIHDS_E_BADARG, "conditioner rows out of range");
    t.E = co->E; t.cond_row0 = co->first_row; t.w_mean = co->w_mean; t.w_std = co->w_std; t.z = co->z;
    t.crng = co->rng; t.rel = co->relevance; t.is_default = co->is_default;
  }
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = obs;
  a.logp = logp; a.g_theta = g_theta_unit;
  a.newton_hist = g_newton_hist;
  const int rc = p->model == VIHDS_MODEL_DR_CONSTANT
                     ? launch_dr_constant_train_v1(p->solver, a, (hipStream_t)stream, &t)
                     : launch_dr_constant_train_v2(p->solver, a, (hipStream_t)stream, &t);
  if (rc == VIHDS_E_UNSUPPORTED)
    return fail(rc, "shape outside the fused kernel's regime (lane-split size limit, or time grid too long for LDS)");
  if (rc != VIHDS_OK) return fail(rc, "fused kernel launch failed");
  return check_hip("vihds_theta_ode_logp_grad launch");
}

// *_precisions models whose adjoint can run one lane per state (vihds_relay_lanes.hpp) and then leaves the precision
// network's weight gradients as one partial row per block: the number of species, 0 for any other model
static int lane_model_species(int model) {
  switch (model) {
    case VIHDS_MODEL_RELAY_CONSTANT_PRECISIONS: return RlRelay::NSP;
    case VIHDS_MODEL_DEGRADER_CONSTANT_PRECISIONS: return RlDegrader::NSP;
    case VIHDS_MODEL_PRPR_CONSTANT_PRECISIONS: return RlPrpr::NSP;
    case VIHDS_MODEL_AUTO_CONSTANT_PRECISIONS: return RlAuto::NSP;
  }
  return 0;
}
// (ABI 14 keeps the entry point: the time-fastest [B][S][N][T] layout belonged to kernel_variant 5 -- the time-parallel
// kernels for relay / degrader / prpr / auto_constant, built in round 4, never faster than the lane kernels and removed in
// round 6 -- so every kernel family now writes [T][N][B][S])
int vihds_ode_traj_layout(const vihds_ode_problem* p) {
  (void)p;
  return 0;
}
int vihds_ode_bwd_reduces_weights(const vihds_ode_problem* p) {
  if (!p) return 0;
  return lane_model_species(p->model) > 0 &&
         relay_lanes_applicable(p->B * p->S, p->solver, p->kernel_variant, p->n_hidden_prec) ? 1 : 0;
}

long long vihds_ode_bwd_aux_floats(const vihds_ode_problem* p) {
  if (!p) return VIHDS_E_BADARG;
  if (p->model == VIHDS_MODEL_DR_BLACKBOX) {
    if (bb_builtin(p)) return bb_aux_floats(p->B * p->S, p->T, p->solver, p->kernel_variant);
    const BbVariant* v = bb_sized(p);
    if (v && bb_sized_gram(v, p)) return v->gram_floats(p->B * p->S) + (long long)v->n_tail * p->B * p->S;
    return v ? bb_sized_dump_floats(v, p) + (long long)v->n_tail * p->B * p->S : VIHDS_E_UNSUPPORTED;
  }
  const ModelEntry* e = entry(p->model);
  if (!e) return VIHDS_E_UNSUPPORTED;
  if (!e->neural_prec) return 0;
  if (vihds_ode_bwd_reduces_weights(p)) return relay_lanes_aux_floats(p->B * p->S, lane_model_species(p->model));  // one partial row per block
  // white-box + neural precisions: [8 + NIN][E][n], NIN = 1 + core states (optional: see vihds_ode_bwd)
  const long long stages = ode_stages(p->solver);
  const long long fields = 8 + e->n_states() - 4 + 1 + (p->n_hidden_prec > 0 ? 2 * p->n_hidden_prec : 0);
  return fields * (p->T - 1) * stages * p->B * p->S;
}
int vihds_blackbox_dump_fields(void) { return bb_dump_fields(); }
int vihds_problem_dump_fields(const vihds_ode_problem* p) {
  if (!p || p->model != VIHDS_MODEL_DR_BLACKBOX) return VIHDS_E_BADARG;
  if (bb_builtin(p)) return bb_dump_fields();
  const BbVariant* v = bb_sized(p);
  return v ? v->dump_fields : VIHDS_E_UNSUPPORTED;
}
int vihds_problem_n_states(const vihds_ode_problem* p) {
  if (!p) return VIHDS_E_BADARG;
  if (p->model == VIHDS_MODEL_DR_BLACKBOX && !bb_builtin(p)) {
    const BbVariant* v = bb_sized(p);
    return v ? v->n_states : VIHDS_E_UNSUPPORTED;
  }
  return vihds_model_n_states(p->model);
}
int vihds_problem_n_slots(const vihds_ode_problem* p) {
  if (!p) return VIHDS_E_BADARG;
  if (p->model == VIHDS_MODEL_DR_BLACKBOX && !bb_builtin(p)) {
    const BbVariant* v = bb_sized(p);
    return v ? v->n_slots : VIHDS_E_UNSUPPORTED;
  }
  return vihds_model_n_slots(p->model);
}
int vihds_blackbox_gram_on_chip(const vihds_ode_problem* p) {
  if (!p || p->model != VIHDS_MODEL_DR_BLACKBOX) return 0;
  if (!bb_builtin(p)) {
    const BbVariant* v = bb_sized(p);
    return v && bb_sized_gram(v, p) ? 1 : 0;
  }
  return bb_gram_on_chip(p->solver, p->kernel_variant);
}
long long vihds_blackbox_tail_offset_floats(const vihds_ode_problem* p) {
  if (!p || p->model != VIHDS_MODEL_DR_BLACKBOX) return VIHDS_E_BADARG;
  if (!bb_builtin(p)) {
    const BbVariant* v = bb_sized(p);
    if (v && bb_sized_gram(v, p)) return v->gram_floats(p->B * p->S);
    return v ? bb_sized_dump_floats(v, p) : VIHDS_E_UNSUPPORTED;
  }
  return bb_tail_offset_floats(p->B * p->S, p->T, p->solver, p->kernel_variant);
}
int vihds_blackbox_gram_reduce(const vihds_ode_problem* p, const float* aux, float* g_weights, void* stream) {
  if (!p || !aux || !g_weights) return fail(VIHDS_E_BADARG, "null argument");
  if (p->model != VIHDS_MODEL_DR_BLACKBOX || !vihds_blackbox_gram_on_chip(p))
    return fail(VIHDS_E_BADARG, "vihds_blackbox_gram_reduce: not an on-chip Gram problem (see vihds_blackbox_gram_on_chip)");
  const ModelEntry* e = entry(p->model);
  const BbVariant* v = bb_builtin(p) ? nullptr : bb_sized(p);
  OdeArgs a;
  if (int rc = build_args(p, e, a, v)) return rc;
  if (v) v->gram_reduce(a, aux, g_weights, (hipStream_t)stream);
  else bb_gram_reduce(a, aux, g_weights, (hipStream_t)stream);
  return check_hip("vihds_blackbox_gram_reduce launch");
}

int vihds_ode_fwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                  const float* times, const float* obs, const float* weights, float* traj, float* xpred, float* logp,
                  void* stream) {
  if (!p || !theta || !times) return fail(VIHDS_E_BADARG, "null problem/theta/times");
  const BbVariant* sized = nullptr;
  const ModelEntry* e = entry(p->model);
  if (!e) return fail(VIHDS_E_UNSUPPORTED, "model not supported by this build");
  if (logp && !obs) return fail(VIHDS_E_BADARG, "logp requested without obs");
  if (e->neural_prec) {
    if (!weights) return fail(VIHDS_E_BADARG, "model has neural blocks: weights must not be NULL");
    if (p->model == VIHDS_MODEL_DR_BLACKBOX) {
      if (!bb_builtin(p) && !(sized = bb_sized(p))) return VIHDS_E_UNSUPPORTED;
      if (!dev1hot || (p->C > 0 && !cond)) return fail(VIHDS_E_BADARG, "dr_blackbox needs cond and dev1hot");
    } else if (p->n_hidden_prec > 256) {
      return fail(VIHDS_E_UNSUPPORTED, "neural precisions: at most 256 hidden units");
    }
  }
  OdeArgs a;
  int rc = build_args(p, e, a, sized);
  if (rc) return rc;
  if (p->C > 0 && !cond) return fail(VIHDS_E_BADARG, "null cond");
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = obs; a.weights = weights;
  a.traj = traj; a.xpred = xpred; a.logp = logp;
  rc = sized ? sized->launch(false, p->solver, a, (hipStream_t)stream, nullptr) : e->launch(false, p->solver, a, (hipStream_t)stream);
  if (rc) return fail(rc, "unknown solver");
  return check_hip("vihds_ode_fwd launch");
}

// ---- evaluation summaries by a second forward pass (csrc/vihds_ode_kernels.hpp: ode_fwd_summ_kernel) ---------------------
namespace vihds {
// partial [B][nch][T][nvp] -> the four summaries, chunks added in a fixed order; one thread per (data row, time point, value):
// consecutive threads read consecutive floats of a partial row
// (var_at_first: constant precisions -- the sums of w / precision were taken at the first time point only)
__global__ void __launch_bounds__(256) summ_finish_kernel(int B, int T, int nch, int nvp, int ns, int var_at_first,
                                                          const float* partial, float* mu, float* sd, float* states,
                                                          float* var) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)B * T * nvp) return;
  const int v = (int)(idx % nvp);
  const int t = (int)((idx / nvp) % T), b = (int)(idx / ((long long)nvp * T));
  if (v >= ns + 12) return;
  const size_t stride = (size_t)T * nvp;
  auto total = [&](int vv, int tt) {
    const float* src = partial + ((size_t)b * nch * T + tt) * nvp + vv;
    float r = 0.f;
    int ch = 0;
    for (; ch + 8 <= nch; ch += 8) {  // (eight loads in flight, added in chunk order)
      float x[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = src[(size_t)(ch + e) * stride];
#pragma unroll
      for (int e = 0; e < 8; ++e) r += x[e];
    }
    for (; ch < nch; ++ch) r += src[(size_t)ch * stride];
    return r;
  };
  if (v < ns) {
    states[((size_t)b * ns + v) * T + t] = total(v, t);
  } else if (v < ns + 4) {
    mu[((size_t)b * 4 + (v - ns)) * T + t] = total(v, t);
  } else if (v < ns + 8) {
    const float m = total(v - 4, t), m2 = total(v, t);  // (the mean again: its own thread's sum, in the same order)
    sd[((size_t)b * 4 + (v - ns - 4)) * T + t] = sqrtf(m2 - m * m);
  } else {
    var[((size_t)b * 4 + (v - ns - 8)) * T + t] = total(v, var_at_first ? 0 : t);
  }
}
}  // namespace vihds
static int summ_species(const vihds_ode_problem* p, const ModelEntry* e) {
  const int n_states = vihds_problem_n_states(p);
  return n_states < 0 ? n_states : (e->neural_prec ? n_states - 4 : n_states);
}
int vihds_ode_fwd_summaries_supported(const vihds_ode_problem* p) {
  if (!p) return 0;
  const ModelEntry* e = entry(p->model);
  if (!e || p->model == VIHDS_MODEL_DR_BLACKBOX || solver_is_adaptive(p->solver)) return 0;
  if (e->neural_prec && p->n_hidden_prec > 256) return 0;
  // ... and the forward launch of this problem is the thread-per-trajectory kernel (whose integration the second pass
  // repeats bit for bit): asked for, or an evaluation-sized launch
  static const long long lane_max = [] {
    const char* ev = std::getenv("VIHDS_LANE_SPLIT_MAX_N");
    return ev ? std::atoll(ev) : 16384LL;
  }();
  const long long n = (long long)p->B * p->S;
  return p->kernel_variant == 1 || (p->kernel_variant == 0 && n > lane_max);
}
long long vihds_ode_fwd_summaries_workspace_floats(const vihds_ode_problem* p) {
  if (!p) return VIHDS_E_BADARG;
  const ModelEntry* e = entry(p->model);
  if (!e) return VIHDS_E_UNSUPPORTED;
  const int ns = summ_species(p, e);
  if (ns < 0) return ns;
  const long long nvp = (ns + 12 + 3) & ~3, nch = ((p->S + 255) / 256) * 4;
  return (long long)p->B * nch * p->T * nvp;
}
int vihds_ode_fwd_summaries(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                            const float* times, const float* weights, const float* log_w, const float* lse,
                            float* workspace, float* iw_predict_mu, float* iw_predict_std, float* iw_states,
                            float* iw_variance, void* stream) {
  if (!p || !theta || !times || !log_w || !lse || !workspace || !iw_predict_mu || !iw_predict_std || !iw_states || !iw_variance)
    return fail(VIHDS_E_BADARG, "null argument");
  const ModelEntry* e = entry(p->model);
  if (!e) return fail(VIHDS_E_UNSUPPORTED, "model not supported by this build");
  if (p->model == VIHDS_MODEL_DR_BLACKBOX || solver_is_adaptive(p->solver))
    return VIHDS_E_UNSUPPORTED;  // (dr_blackbox and the adaptive pairs keep vihds_ode_fwd + vihds_iw_summaries_states)
  if (e->neural_prec) {
    if (!weights) return fail(VIHDS_E_BADARG, "model has neural blocks: weights must not be NULL");
    if (p->n_hidden_prec > 256) return fail(VIHDS_E_UNSUPPORTED, "neural precisions: at most 256 hidden units");
  }
  OdeArgs a;
  int rc = build_args(p, e, a, nullptr);
  if (rc) return rc;
  if (p->C > 0 && !cond) return fail(VIHDS_E_BADARG, "null cond");
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = nullptr; a.weights = weights;
  a.traj = nullptr; a.xpred = nullptr; a.logp = nullptr;
  a.kernel_variant = 1;  // (every model's launcher sends this to launch_ode: the thread-per-trajectory kernels)
  const int ns = summ_species(p, e);
  if (ns < 0) return ns;
  SummArgs sa;
  sa.log_w = log_w; sa.lse = lse; sa.partial = workspace;
  sa.nch = ((p->S + 255) / 256) * 4;
  sa.nvp = (ns + 12 + 3) & ~3;
  g_summ = &sa;
  rc = e->launch(false, p->solver, a, (hipStream_t)stream);
  g_summ = nullptr;
  if (rc) return rc == VIHDS_E_UNSUPPORTED ? rc : fail(rc, "vihds_ode_fwd_summaries: launch refused");
  const long long items = (long long)p->B * p->T * sa.nvp;
  hipLaunchKernelGGL(vihds::summ_finish_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p->B, p->T,
                     sa.nch, sa.nvp, ns, e->neural_prec ? 0 : 1, workspace, iw_predict_mu, iw_predict_std, iw_states,
                     iw_variance);
  return check_hip("vihds_ode_fwd_summaries launch");
}

long long vihds_ode_adaptive_workspace_floats(const vihds_ode_problem* p) {
  if (!p) return VIHDS_E_BADARG;
  const ModelEntry* e = entry(p->model);
  if (!e) return VIHDS_E_UNSUPPORTED;
  const long long n = (long long)p->B * p->S;
  const int n_states = vihds_problem_n_states(p);
  if (n_states < 0) return n_states;
  return 2 * (long long)n_states * n + 2 * ((n + 255) / 256);
}

int vihds_ode_adaptive_grid(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                            const float* weights, const float* times_host, float rtol, float atol, float* workspace,
                            float* grid_host, int max_grid, int* index_host, void* stream) {
  if (!p || !theta || !times_host || !workspace || !grid_host || !index_host || max_grid < 2)
    return fail(VIHDS_E_BADARG, "null argument");
  if (!solver_is_adaptive(p->solver)) return fail(VIHDS_E_BADARG, "vihds_ode_adaptive_grid needs an adaptive solver id");
  if (!(rtol > 0.f) || !(atol > 0.f)) return fail(VIHDS_E_BADARG, "rtol and atol must be positive");
  const BbVariant* sized = nullptr;
  const ModelEntry* e = entry(p->model);
  if (!e) return fail(VIHDS_E_UNSUPPORTED, "model not supported by this build");
  if (e->neural_prec) {
    if (!weights) return fail(VIHDS_E_BADARG, "model has neural blocks: weights must not be NULL");
    if (p->model == VIHDS_MODEL_DR_BLACKBOX) {
      if (!bb_builtin(p) && !(sized = bb_sized(p))) return VIHDS_E_UNSUPPORTED;
      if (!dev1hot || (p->C > 0 && !cond)) return fail(VIHDS_E_BADARG, "dr_blackbox needs cond and dev1hot");
    }
  }
  for (int k = 1; k < p->T; ++k)
    if (!(times_host[k] > times_host[k - 1])) return fail(VIHDS_E_BADARG, "output times must increase");
  OdeArgs a;
  int rc = build_args(p, e, a, sized);
  if (rc) return rc;
  if (p->C > 0 && !cond) return fail(VIHDS_E_BADARG, "null cond");
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.weights = weights;
  AdaptiveCtl ctl = {times_host, rtol, atol, workspace, grid_host, max_grid, index_host, 0};
  if (sized) {
    rc = sized->launch(false, p->solver, a, (hipStream_t)stream, &ctl);
  } else {
    g_adaptive_ctl = &ctl;
    rc = e->launch(false, p->solver, a, (hipStream_t)stream);
    g_adaptive_ctl = nullptr;
  }
  if (rc == VIHDS_E_UNSUPPORTED) return fail(rc, "the accepted grid does not fit max_grid points");
  if (rc == VIHDS_E_BADARG) return fail(rc, "step size underflow or non-finite error estimate");
  if (rc) return fail(rc, "adaptive step controller failed");
  if (int h = check_hip("vihds_ode_adaptive_grid")) return h;
  return ctl.result;
}

// ---- torchdiffeq's adaptive algorithm on the device (csrc/vihds_rk_adaptive_device.hpp) -------------------------------------
static bool adaptive_device_model(const vihds_ode_problem* p, const ModelEntry* e) {
  // (round 5: the white-box models with neural precisions too, when their network has no hidden layer; not dr_blackbox)
  const bool net_ok = e && (!e->neural_prec || (p->model != VIHDS_MODEL_DR_BLACKBOX && p->n_hidden_prec < 1));
  return net_ok && p->solver >= VIHDS_SOLVER_DOPRI5 && p->solver <= VIHDS_SOLVER_ADAPTIVE_HEUN &&
         (long long)p->B * p->S <= (long long)ADP_MAX_BLOCKS * ADP_BLOCK;
}
long long vihds_ode_adaptive_tape_floats(const vihds_ode_problem* p, int max_steps) {
  if (!p || max_steps < 1) return VIHDS_E_BADARG;
  const ModelEntry* e = entry(p->model);
  if (!adaptive_device_model(p, e)) return VIHDS_E_UNSUPPORTED;
  const int n = p->B * p->S, nblk = (n + ADP_BLOCK - 1) / ADP_BLOCK;
  return (long long)AdaptiveLayout(nblk, max_steps, p->T, e->n_states(), (size_t)n).total;
}
static int adaptive_device_call(int mode, const vihds_ode_problem* p, const float* theta, const float* cond,
                                const float* dev1hot, const float* times, float rtol, float atol, int max_steps,
                                float* workspace, float* traj, const float* g_traj, float* g_theta, void* stream,
                                const float* weights = nullptr, float* g_weights = nullptr) {
  if (!p || !theta || !times || !workspace || max_steps < 1) return fail(VIHDS_E_BADARG, "null problem/theta/times/workspace");
  const ModelEntry* e = entry(p->model);
  if (!adaptive_device_model(p, e))
    return fail(VIHDS_E_UNSUPPORTED, "device-resident adaptive solver: dopri5 / bosh3 / adaptive_heun on a white-box model (neural "
                                     "precisions without a hidden layer included), at most 65 536 trajectories (use "
                                     "vihds_ode_adaptive_grid otherwise)");
  if (e->neural_prec && !weights) return fail(VIHDS_E_BADARG, "model has neural precisions: weights must not be NULL");
  OdeArgs a;
  int rc = build_args(p, e, a, nullptr);
  if (rc) return rc;
  if (p->C > 0 && !cond) return fail(VIHDS_E_BADARG, "null cond");
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times;
  a.traj = traj; a.g_traj = g_traj; a.g_theta = g_theta; a.weights = weights; a.g_weights = g_weights;
  AdaptiveDevCtl ctl = {mode, {workspace, rtol, atol, max_steps}, 0};
  g_adaptive_dev = &ctl;
  rc = e->launch(mode == 2, p->solver, a, (hipStream_t)stream);
  g_adaptive_dev = nullptr;
  if (rc) return fail(rc, "device-resident adaptive solver: not available for this model / solver");
  return check_hip(mode == 1 ? "vihds_ode_adaptive_fwd launch" : "vihds_ode_adaptive_bwd launch");
}
int vihds_ode_adaptive_fwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                           const float* times, float rtol, float atol, int max_steps, float* workspace, float* traj,
                           void* stream) {
  if (!traj) return fail(VIHDS_E_BADARG, "null traj");
  return adaptive_device_call(1, p, theta, cond, dev1hot, times, rtol, atol, max_steps, workspace, traj, nullptr, nullptr, stream);
}
int vihds_ode_adaptive_fwd_w(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                             const float* weights, const float* times, float rtol, float atol, int max_steps, float* workspace,
                             float* traj, void* stream) {
  if (!traj) return fail(VIHDS_E_BADARG, "null traj");
  return adaptive_device_call(1, p, theta, cond, dev1hot, times, rtol, atol, max_steps, workspace, traj, nullptr, nullptr, stream,
                              weights, nullptr);
}
int vihds_ode_adaptive_bwd_w(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                             const float* weights, const float* times, int max_steps, const float* workspace,
                             const float* g_traj, float* g_theta, float* g_weights, void* stream) {
  if (!g_theta) return fail(VIHDS_E_BADARG, "null g_theta");
  return adaptive_device_call(2, p, theta, cond, dev1hot, times, 0.f, 0.f, max_steps, const_cast<float*>(workspace), nullptr,
                              g_traj, g_theta, stream, weights, g_weights);
}
int vihds_ode_adaptive_bwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                           const float* times, int max_steps, const float* workspace, const float* g_traj, float* g_theta,
                           void* stream) {
  if (!g_theta) return fail(VIHDS_E_BADARG, "null g_theta");
  return adaptive_device_call(2, p, theta, cond, dev1hot, times, 0.f, 0.f, max_steps, const_cast<float*>(workspace), nullptr,
                              g_traj, g_theta, stream);
}

int vihds_ode_bwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                  const float* times, const float* obs, const float* weights, const float* traj, const float* g_traj,
                  const float* g_xpred, const float* g_logp, float* g_theta, float* g_weights, float* aux,
                  void* stream) {
  if (!p || !theta || !times || !traj || !g_theta) return fail(VIHDS_E_BADARG, "null problem/theta/times/traj/g_theta");
  const BbVariant* sized = nullptr;
  const ModelEntry* e = entry(p->model);
  if (!e) return fail(VIHDS_E_UNSUPPORTED, "model not supported by this build");
  if (!obs) return fail(VIHDS_E_BADARG, "null obs");
  if (e->neural_prec) {
    if (!weights) return fail(VIHDS_E_BADARG, "model has neural blocks: weights must not be NULL");
    if (p->model == VIHDS_MODEL_DR_BLACKBOX) {
      if (!bb_builtin(p) && !(sized = bb_sized(p))) return VIHDS_E_UNSUPPORTED;
      if (!dev1hot || (p->C > 0 && !cond)) return fail(VIHDS_E_BADARG, "dr_blackbox needs cond and dev1hot");
    } else if (p->n_hidden_prec > 256) {
      return fail(VIHDS_E_UNSUPPORTED, "neural precisions: at most 256 hidden units");
    }
  }
  OdeArgs a;
  int rc = build_args(p, e, a, sized);
  if (rc) return rc;
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = obs; a.weights = weights;
  a.g_weights = g_weights; a.aux = aux;
  // (white-box + hidden-layer precisions without aux: the state / theta adjoints only -- the three weight matrices have
  // no per-thread accumulators, they come from the dump; g_weights then receives nothing)
  if (p->model == VIHDS_MODEL_DR_BLACKBOX && !aux) return fail(VIHDS_E_BADARG, "dr_blackbox backward needs the aux buffer");
  a.traj_in = traj; a.g_traj = g_traj; a.g_xpred = g_xpred; a.g_logp = g_logp; a.g_theta = g_theta;
  rc = sized ? sized->launch(true, p->solver, a, (hipStream_t)stream, nullptr) : e->launch(true, p->solver, a, (hipStream_t)stream);
  if (rc) return fail(rc, "unknown solver");
  return check_hip("vihds_ode_bwd launch");
}

int vihds_rng_advance(unsigned int* rng, void* stream) {
  if (!rng) return fail(VIHDS_E_BADARG, "null rng");
  launch_rng_advance(rng, (hipStream_t)stream);
  return check_hip("vihds_rng_advance launch");
}

int vihds_theta_ode_fwd(const vihds_ode_problem* p, int P, const int* kind, const float* q_mu, const float* q_prec,
                        const float* p_mu, const float* p_prec, const float* clip_lo, const float* clip_hi, float* u,
                        const vihds_theta_opts* opts, const vihds_offset_layer* offset, const float* cond, const float* dev1hot,
                        const float* times, const float* obs, const float* weights, float* theta, float* log_q, float* log_p,
                        float* traj, float* xpred, float* logp, void* stream) {
  if (!p || !kind || !q_mu || !q_prec || !p_mu || !p_prec || !clip_lo || !clip_hi || !u || !times || !theta)
    return fail(VIHDS_E_BADARG, "null argument");
  const ModelEntry* e = entry(p->model);
  if (!e) return fail(VIHDS_E_UNSUPPORTED, "model not supported by this build");
  const bool lane_family = p->model == VIHDS_MODEL_AUTO_CONSTANT || p->model == VIHDS_MODEL_PRPR_CONSTANT ||
                           p->model == VIHDS_MODEL_RELAY_CONSTANT || p->model == VIHDS_MODEL_DEGRADER_CONSTANT ||
                           lane_model_species(p->model) > 0;
  if (!lane_family && !(p->model == VIHDS_MODEL_DR_BLACKBOX && bb_builtin(p)))
    return fail(VIHDS_E_UNSUPPORTED, "vihds_theta_ode_fwd: no sampling stage in this model's forward kernels");
  if (logp && !obs) return fail(VIHDS_E_BADARG, "logp requested without obs");
  if (P <= 0 || P > p->n_rows) return fail(VIHDS_E_BADARG, "P out of range");
  if (e->neural_prec && !weights) return fail(VIHDS_E_BADARG, "model has neural blocks: weights must not be NULL");
  if (p->model == VIHDS_MODEL_DR_BLACKBOX && (!dev1hot || (p->C > 0 && !cond))) return fail(VIHDS_E_BADARG, "dr_blackbox needs cond and dev1hot");
  OdeArgs a;
  int rc = build_args(p, e, a, nullptr);
  if (rc) return rc;
  if (p->C > 0 && !cond) return fail(VIHDS_E_BADARG, "null cond");
  const vihds_theta_opts o = theta_opts(opts);
  ThetaStageArgs t;
  std::memset(&t, 0, sizeof(t));
  t.P = P; t.kind = kind; t.q_mu = q_mu; t.q_prec = q_prec; t.q_rows = o.q_rows; t.prec_is_log = o.q_prec_is_log;
  t.p_mu = p_mu; t.p_prec = p_prec; t.clip_lo = clip_lo; t.clip_hi = clip_hi; t.u = u; t.rng = o.rng;
  t.S_total = p->S; t.s_off = 0;
  if (o.S_total > 0) {
    if (o.s_offset < 0 || o.s_offset + p->S > o.S_total) return fail(VIHDS_E_BADARG, "bad sample window");
    t.S_total = o.S_total; t.s_off = o.s_offset;
  }
  t.theta = theta; t.n_rows = p->n_rows; t.log_q = log_q; t.log_p = log_p;
  if (offset && offset->n > 0) {
    if (!offset->W || !offset->bias || !dev1hot || p->D <= 0 || offset->src_row < 0 || offset->src_row + offset->n > P ||
        offset->dst_row < P || offset->dst_row + offset->n > p->n_rows)
      return fail(VIHDS_E_BADARG, "bad offset layer");
    t.off_n = offset->n; t.off_src = offset->src_row; t.off_dst = offset->dst_row; t.off_w = offset->W; t.off_b = offset->bias;
  }
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = obs; a.weights = weights;
  a.traj = traj; a.xpred = xpred; a.logp = logp;
  g_theta_stage = &t;
  rc = e->launch(false, p->solver, a, (hipStream_t)stream);
  g_theta_stage = nullptr;
  if (rc == VIHDS_E_UNSUPPORTED)
    return fail(rc, "vihds_theta_ode_fwd: kernel variant / solver / shape outside the kernels that carry the sampling stage");
  if (rc) return fail(rc, "unknown solver");
  return check_hip("vihds_theta_ode_fwd launch");
}

/* vihds_ode_bwd with the log-likelihood gradient formed in the kernel: g_logp[j][b][s] = -(1/B) softmax_s(log_w[b][.]) for the
 * four signals j, log_w = sum_j logp[j] + log_p - log_q (reference training.py:135-149) -- the training step's adjoint without
 * an IWAE launch in front of it (csrc/vihds_iwae_inline.hpp). */
int vihds_ode_bwd_elbo(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                       const float* times, const float* obs, const float* weights, const float* traj, const float* logp,
                       const float* log_p, const float* log_q, float* g_theta, float* g_weights, float* aux,
                       void* stream) {
  if (!p || !theta || !times || !traj || !g_theta || !logp) return fail(VIHDS_E_BADARG, "null problem/theta/times/traj/g_theta/logp");
  const BbVariant* sized = nullptr;
  const ModelEntry* e = entry(p->model);
  if (!e) return fail(VIHDS_E_UNSUPPORTED, "model not supported by this build");
  if (!obs) return fail(VIHDS_E_BADARG, "null obs");
  if (solver_is_adaptive(p->solver)) return fail(VIHDS_E_UNSUPPORTED, "vihds_ode_bwd_elbo: fixed-grid solvers");
  if (e->neural_prec) {
    if (!weights) return fail(VIHDS_E_BADARG, "model has neural blocks: weights must not be NULL");
    if (p->model == VIHDS_MODEL_DR_BLACKBOX) {
      // (side libraries were built against the same headers: their kernels form the weights too)
      if (!bb_builtin(p) && !(sized = bb_sized(p))) return VIHDS_E_UNSUPPORTED;
      if (!dev1hot || (p->C > 0 && !cond)) return fail(VIHDS_E_BADARG, "dr_blackbox needs cond and dev1hot");
    } else if (p->n_hidden_prec > 256) {
      return fail(VIHDS_E_UNSUPPORTED, "neural precisions: at most 256 hidden units");
    }
  }
  OdeArgs a;
  int rc = build_args(p, e, a, sized);
  if (rc) return rc;
  a.theta = theta; a.cond = cond; a.dev1hot = dev1hot; a.times = times; a.obs = obs; a.weights = weights;
  a.g_weights = g_weights; a.aux = aux;
  if (p->model == VIHDS_MODEL_DR_BLACKBOX && !aux) return fail(VIHDS_E_BADARG, "dr_blackbox backward needs the aux buffer");
  a.traj_in = traj; a.g_theta = g_theta;
  a.iw_logp = logp; a.iw_log_p = log_p; a.iw_log_q = log_q;
  a.logp_grad_broadcast = 1;
  rc = sized ? sized->launch(true, p->solver, a, (hipStream_t)stream, nullptr) : e->launch(true, p->solver, a, (hipStream_t)stream);
  if (rc) return fail(rc, "unknown solver");
  return check_hip("vihds_ode_bwd_elbo launch");
}

int vihds_theta_fwd(int P, int B, int S, const int* kind, const float* q_mu, const float* q_prec, const float* p_mu,
                    const float* p_prec, const float* clip_lo, const float* clip_hi, float* u, float* theta,
                    float* log_q, float* log_p, const vihds_theta_opts* opts, void* stream) {
  if (P <= 0 || B <= 0 || S <= 0) return fail(VIHDS_E_BADARG, "P, B, S must be > 0");
  if (!kind || !q_mu || !q_prec || !p_mu || !p_prec || !clip_lo || !clip_hi || !u || !theta)
    return fail(VIHDS_E_BADARG, "null argument");
  const vihds_theta_opts o = theta_opts(opts);
  if (o.rng && (o.S_total < S || o.s_offset < 0 || o.s_offset + S > o.S_total))
    return fail(VIHDS_E_BADARG, "rng: need 0 <= s_offset and s_offset + S <= S_total");
  launch_theta_fwd(P, B, S, kind, q_mu, q_prec, p_mu, p_prec, clip_lo, clip_hi, u, theta, log_q, log_p, o,
                   (hipStream_t)stream);
  return check_hip("vihds_theta_fwd launch");
}

int vihds_theta_bwd(int P, int B, int S, const int* kind, const float* q_mu, const float* q_prec, const float* p_mu,
                    const float* p_prec, const float* clip_lo, const float* clip_hi, const float* u,
                    const float* g_theta, const float* g_log_q, const float* g_log_p, float* g_q_mu, float* g_q_prec,
                    const vihds_theta_opts* opts, void* stream) {
  if (P <= 0 || B <= 0 || S <= 0) return fail(VIHDS_E_BADARG, "P, B, S must be > 0");
  if (!kind || !q_mu || !q_prec || !p_mu || !p_prec || !clip_lo || !clip_hi || !u || !g_q_mu || !g_q_prec)
    return fail(VIHDS_E_BADARG, "null argument");
  if (opts && opts->iwae) {
    const vihds_iwae_job* j = opts->iwae;
    if (!j->logp || !j->log_w || !j->lse || !j->loss || !j->ticket || j->n_iwae_total <= 0)
      return fail(VIHDS_E_BADARG, "vihds_iwae_job: null buffer or n_iwae_total <= 0");
    if ((size_t)S * sizeof(float) > 60 * 1024) return fail(VIHDS_E_UNSUPPORTED, "vihds_iwae_job: S too large for the row buffer");
  }
  launch_theta_bwd(P, B, S, kind, q_mu, q_prec, p_mu, p_prec, clip_lo, clip_hi, u, g_theta, g_log_q, g_log_p, g_q_mu,
                   g_q_prec, theta_opts(opts), (hipStream_t)stream);
  return check_hip("vihds_theta_bwd launch");
}

int vihds_iwae_fwd(int B, int S, const float* logp, const float* log_p, const float* log_q, float* log_w,
                   float* row_max, float* row_sumexp, void* stream) {
  if (B <= 0 || S <= 0 || !logp || !log_w || !row_max || !row_sumexp) return fail(VIHDS_E_BADARG, "bad argument");
  launch_iwae_fwd(B, S, logp, log_p, log_q, log_w, row_max, row_sumexp, (hipStream_t)stream);
  return check_hip("vihds_iwae_fwd launch");
}

int vihds_iwae_bwd(int B, int S, const float* log_w, const float* lse, const float* g_lse, float* g_logw,
                   void* stream) {
  if (B <= 0 || S <= 0 || !log_w || !lse || !g_lse || !g_logw) return fail(VIHDS_E_BADARG, "bad argument");
  launch_iwae_bwd(B, S, log_w, lse, g_lse, g_logw, (hipStream_t)stream);
  return check_hip("vihds_iwae_bwd launch");
}

int vihds_iwae_loss_fwd(int B, int S, int n_iwae_total, const float* logp, const float* log_p, const float* log_q,
                        float* log_w, float* row_max, float* row_sumexp, float* lse, float* loss, float* unit_g_logw,
                        float* unit_g_neg_logw, unsigned int* ticket, void* stream) {
  if (B <= 0 || S <= 0 || n_iwae_total <= 0 || !logp || !log_w || !row_max || !row_sumexp || !lse || !loss)
    return fail(VIHDS_E_BADARG, "bad argument");
  const float log_n = logf((float)n_iwae_total);
  if (ticket && S <= 1024) {  // one launch, block per row, the last block takes the mean over rows
    launch_iwae_loss_rows(B, S, log_n, logp, log_p, log_q, log_w, row_max, row_sumexp, lse, loss, unit_g_logw,
                          unit_g_neg_logw, ticket, (hipStream_t)stream);
    return check_hip("vihds_iwae_loss_fwd launch");
  }
  if (B <= 64 && S <= 256) {  // one launch, one block
    launch_iwae_loss_small(B, S, log_n, logp, log_p, log_q, log_w, row_max, row_sumexp, lse, loss, unit_g_logw,
                           unit_g_neg_logw, (hipStream_t)stream);
    return check_hip("vihds_iwae_loss_fwd launch");
  }
  if (unit_g_logw)
    return fail(VIHDS_E_UNSUPPORTED, "unit-gradient outputs need a ticket and S <= 1024, or B <= 64 and S <= 256 "
                                     "(vihds_iwae_loss_unit_grad)");
  launch_iwae_fwd(B, S, logp, log_p, log_q, log_w, row_max, row_sumexp, (hipStream_t)stream);
  launch_iwae_finish(B, log_n, row_max, row_sumexp, lse, loss, (hipStream_t)stream);
  return check_hip("vihds_iwae_loss_fwd launch");
}

int vihds_iwae_combine(int n_ranks, int B, int S, int n_iwae_total, const float* gathered, const float* log_w,
                       float* lse, float* loss, float* unit_g_logw, float* unit_g_neg_logw, void* stream) {
  if (n_ranks <= 0 || B <= 0 || S <= 0 || n_iwae_total <= 0 || !gathered || !lse || !loss)
    return fail(VIHDS_E_BADARG, "bad argument");
  if (unit_g_logw && !log_w) return fail(VIHDS_E_BADARG, "unit-gradient outputs need log_w");
  launch_iwae_combine(n_ranks, B, S, logf((float)n_iwae_total), gathered, log_w, lse, loss, unit_g_logw,
                      unit_g_neg_logw, (hipStream_t)stream);
  return check_hip("vihds_iwae_combine launch");
}

/* 1 if vihds_iwae_loss_fwd can also emit the unit-upstream-gradient outputs for this shape */
int vihds_iwae_loss_unit_grad(int B, int S, int with_ticket) {
  if (B <= 0 || S <= 0) return 0;
  return ((with_ticket && S <= 1024) || (B <= 64 && S <= 256)) ? 1 : 0;
}

int vihds_iwae_loss_bwd(int B, int S, const float* log_w, const float* lse, const float* g_loss, float* g_logw,
                        float* g_neg_logw, void* stream) {
  if (B <= 0 || S <= 0 || !log_w || !lse || !g_loss || !g_logw) return fail(VIHDS_E_BADARG, "bad argument");
  launch_iwae_loss_bwd(B, S, log_w, lse, g_loss, g_logw, g_neg_logw, (hipStream_t)stream);
  return check_hip("vihds_iwae_loss_bwd launch");
}

int vihds_device_condition(int E, int B, int S, int S_total, int s_offset, int D, float w_mean, float w_std,
                           const float* z, unsigned int* rng, const float* dev1hot, const float* relevance,
                           const int* is_default, float* out, void* stream) {
  if (E <= 0 || B <= 0 || S <= 0 || D <= 0 || (!z && !rng) || !dev1hot || !relevance || !is_default || !out)
    return fail(VIHDS_E_BADARG, "bad argument");
  if (S_total <= 0) { S_total = S; s_offset = 0; }
  if (s_offset < 0 || s_offset + S > S_total) return fail(VIHDS_E_BADARG, "need 0 <= s_offset, s_offset + S <= S_total");
  launch_device_condition(E, B, S, S_total, s_offset, D, w_mean, w_std, z, rng, dev1hot, relevance, is_default, out,
                          (hipStream_t)stream);
  return check_hip("vihds_device_condition launch");
}

int vihds_iw_summaries(int B, int S, int T, int N_total, int n_species, const float* log_w, const float* lse,
                       const float* traj, const float* xpred, const float* theta, const int* prec_rows,
                       float* iw_predict_mu, float* iw_predict_std, float* iw_states, float* iw_variance,
                       void* stream) {
  if (B <= 0 || S <= 0 || T <= 0 || !log_w || !lse || !traj || !xpred || !iw_predict_mu || !iw_predict_std ||
      !iw_states || !iw_variance)
    return fail(VIHDS_E_BADARG, "bad argument");
  if (theta && !prec_rows) return fail(VIHDS_E_BADARG, "theta given without prec_rows");
  if (!theta && N_total < n_species + 4) return fail(VIHDS_E_BADARG, "neural precisions need N_total >= n_species+4");
  launch_iw_summaries(B, S, T, N_total, n_species, log_w, lse, traj, xpred, VIHDS_OBS_DEFAULT, theta, prec_rows,
                      iw_predict_mu, iw_predict_std, iw_states, iw_variance, (hipStream_t)stream);
  return check_hip("vihds_iw_summaries launch");
}
int vihds_iw_summaries_plan(int time_points_per_block) {
  const int before = iw_summaries_tpb_override;
  iw_summaries_tpb_override = time_points_per_block;
  return before;
}
int vihds_iw_summaries_states(int B, int S, int T, int N_total, int n_species, int observe_kind, const float* log_w,
                              const float* lse, const float* traj, const float* theta, const int* prec_rows,
                              float* iw_predict_mu, float* iw_predict_std, float* iw_states, float* iw_variance,
                              void* stream) {
  if (B <= 0 || S <= 0 || T <= 0 || !log_w || !lse || !traj || !iw_predict_mu || !iw_predict_std || !iw_states ||
      !iw_variance)
    return fail(VIHDS_E_BADARG, "bad argument");
  if (theta && !prec_rows) return fail(VIHDS_E_BADARG, "theta given without prec_rows");
  if (!theta && N_total < n_species + 4) return fail(VIHDS_E_BADARG, "neural precisions need N_total >= n_species+4");
  const int need = observe_kind == VIHDS_OBS_DEFAULT ? 6 : (observe_kind == VIHDS_OBS_INDUCER ? 5 : 4);
  if (observe_kind < VIHDS_OBS_DEFAULT || observe_kind > VIHDS_OBS_INDUCER || n_species < need)
    return fail(VIHDS_E_BADARG, "observe_kind / n_species: the observation map reads more species than the model has");
  launch_iw_summaries(B, S, T, N_total, n_species, log_w, lse, traj, nullptr, observe_kind, theta, prec_rows,
                      iw_predict_mu, iw_predict_std, iw_states, iw_variance, (hipStream_t)stream);
  return check_hip("vihds_iw_summaries_states launch");
}

static int check_encoder_shape(const vihds_encoder_shape* s) {
  if (!s) return fail(VIHDS_E_BADARG, "null shape");
  if (s->B <= 0 || s->C_in <= 0 || s->L <= 0 || s->F <= 0 || s->K <= 0 || s->pool <= 0 || s->H <= 0)
    return fail(VIHDS_E_BADARG, "encoder dimensions must be > 0");
  if (s->L - s->K + 1 - s->pool + 1 <= 0) return fail(VIHDS_E_BADARG, "time axis shorter than filter + pool");
  if (s->nl < 0 || s->ng < 0 || s->ngl < 0 || s->nc < 0 || s->n_tr < 0 || s->D < 0)
    return fail(VIHDS_E_BADARG, "negative count");
  if (encoder_fwd_lds_bytes(*s) > 60 * 1024 || encoder_bwd_lds_bytes(*s) > 60 * 1024)
    return fail(VIHDS_E_UNSUPPORTED, "encoder working set exceeds the 60 KB LDS budget of the fused kernels");
  return VIHDS_OK;
}

int vihds_encoder_fwd(const vihds_encoder_shape* s, const float* delta_obs, const float* inputs, const float* dev1hot,
                      const float* conv_w, const float* conv_b, const float* lin_w, const float* lin_b,
                      const float* local_w, const float* local_b, const float* gcond_w, const float* global_free,
                      const float* const_values, float* q_all, float* pooled, float* hidden, void* stream) {
  if (int rc = check_encoder_shape(s)) return rc;
  if (!delta_obs || !conv_w || !conv_b || !lin_w || !lin_b || !q_all || !pooled || !hidden)
    return fail(VIHDS_E_BADARG, "null argument");
  if ((s->nl > 0 && !local_w) || (s->ng > 0 && !gcond_w) || (s->ngl > 0 && !global_free) || (s->nc > 0 && !const_values))
    return fail(VIHDS_E_BADARG, "missing parameter tensor");
  if (((s->l_tr || s->g_tr) && s->n_tr > 0 && !inputs) || ((s->l_dv || s->g_dv) && s->D > 0 && !dev1hot))
    return fail(VIHDS_E_BADARG, "missing conditioning input");
  launch_encoder_fwd(*s, delta_obs, inputs, dev1hot, conv_w, conv_b, lin_w, lin_b, local_w, local_b, gcond_w,
                     global_free, const_values, q_all, pooled, hidden, (hipStream_t)stream);
  return check_hip("vihds_encoder_fwd launch");
}

int vihds_encoder_bwd(const vihds_encoder_shape* s, const float* g_all, const float* delta_obs, const float* inputs,
                      const float* dev1hot, const float* lin_w, const float* local_w, const float* pooled,
                      const float* hidden, float* g_pre, float* g_conv, float* g_conv_w, float* g_conv_b,
                      float* g_lin_w, float* g_lin_b, float* g_local_w, float* g_local_b, float* g_gcond_w,
                      float* g_global_free, void* stream) {
  if (int rc = check_encoder_shape(s)) return rc;
  if (!g_all || !delta_obs || !lin_w || !pooled || !hidden || !g_pre || !g_conv || !g_conv_w || !g_conv_b || !g_lin_w ||
      !g_lin_b)
    return fail(VIHDS_E_BADARG, "null argument");
  if ((s->nl > 0 && (!local_w || !g_local_w)) || (s->ng > 0 && !g_gcond_w) || (s->ngl > 0 && !g_global_free))
    return fail(VIHDS_E_BADARG, "missing gradient tensor");
  launch_encoder_bwd(*s, g_all, delta_obs, inputs, dev1hot, lin_w, local_w, pooled, hidden, g_pre, g_conv, g_conv_w,
                     g_conv_b, g_lin_w, g_lin_b, g_local_w, g_local_b, g_gcond_w, g_global_free, (hipStream_t)stream);
  return check_hip("vihds_encoder_bwd launch");
}

long long vihds_gram_scratch_floats(long long n_columns, int n_rects, const vihds_gram_rect* rects) {
  if (n_columns <= 0 || !rects) return VIHDS_E_BADARG;
  return gram_scratch_floats(n_columns, n_rects, rects);
}

int vihds_gram_blocks(int n_fields, long long n_columns, int n_rects, const vihds_gram_rect* rects, const float* X,
                      float* scratch, float* out, void* stream) {
  if (n_fields <= 0 || n_columns <= 0 || !rects || !X || !scratch || !out) return fail(VIHDS_E_BADARG, "bad argument");
  const int rc = launch_gram(n_fields, n_columns, n_rects, rects, X, scratch, out, (hipStream_t)stream);
  if (rc == VIHDS_E_UNSUPPORTED) return fail(rc, "too many 4x4 register tiles (128 max) or fields (126 max) per call");
  if (rc != VIHDS_OK) return fail(rc, "bad rectangle");
  return check_hip("vihds_gram_blocks launch");
}

int vihds_blackbox_tail_grads(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                              const float* tail, const int* dest, float* g_weights, void* stream) {
  if (!p || !theta || !dev1hot || !tail || !dest || !g_weights || (p->C > 0 && !cond))
    return fail(VIHDS_E_BADARG, "null argument");
  if (p->model != VIHDS_MODEL_DR_BLACKBOX) return fail(VIHDS_E_BADARG, "dr_blackbox only");
  if (p->B <= 0 || p->S <= 0) return fail(VIHDS_E_BADARG, "B and S must be positive");
  const int n_lat = p->n_const - p->C - p->D, NX = 4 + p->n_latent_states, NP = p->n_hidden_states + p->n_hidden_prec;
  if (n_lat < 0 || NP <= 0) return fail(VIHDS_E_BADARG, "bad network shape");
  for (int k = 0; k < n_lat && k < VIHDS_MAX_SLOTS; ++k)
    if (p->slot_row[k] < 0 || p->slot_row[k] >= p->n_rows) return fail(VIHDS_E_BADARG, "slot_row out of range");
  const int rc = launch_bb_tail(NP + 2 * NX + 8, NP, p->B * p->S, p->S, n_lat, p->C, p->D, p->slot_row, theta, cond,
                                dev1hot, tail, dest, g_weights, (hipStream_t)stream);
  if (rc) return fail(rc, "needs 1..16 latent inputs, 1..4 treatments and a 1..12 wide device one-hot");
  return check_hip("vihds_blackbox_tail_grads launch");
}

static int offset_rows_check(int B, int S, int D, int n, int n_rows, int src_row, int dst_row) {
  if (B <= 0 || S <= 0 || D <= 0 || n <= 0) return fail(VIHDS_E_BADARG, "B, S, D, n must be positive");
  if (B > 8192) return fail(VIHDS_E_UNSUPPORTED, "at most 8192 data rows");
  if (src_row < 0 || dst_row < 0 || src_row + n > n_rows || dst_row + n > n_rows)
    return fail(VIHDS_E_BADARG, "theta rows out of range");
  if (src_row < dst_row + n && dst_row < src_row + n) return fail(VIHDS_E_BADARG, "source and destination rows overlap");
  return VIHDS_OK;
}
int vihds_offset_rows_fwd(int B, int S, int D, int n, int n_rows, int src_row, int dst_row, const float* W,
                          const float* bias, const float* dev1hot, float* theta, void* stream) {
  if (!W || !bias || !dev1hot || !theta) return fail(VIHDS_E_BADARG, "null argument");
  if (int rc = offset_rows_check(B, S, D, n, n_rows, src_row, dst_row)) return rc;
  launch_offset_rows_fwd(B, S, D, n, src_row, dst_row, W, bias, dev1hot, theta, (hipStream_t)stream);
  return check_hip("vihds_offset_rows_fwd launch");
}
int vihds_offset_rows_bwd(int B, int S, int D, int n, int n_rows, int src_row, int dst_row, int accumulate,
                          const float* dev1hot, float* g_theta, float* g_wb, void* stream) {
  if (!dev1hot || !g_theta) return fail(VIHDS_E_BADARG, "null argument");
  if (int rc = offset_rows_check(B, S, D, n, n_rows, src_row, dst_row)) return rc;
  if (D + 1 > 1024) return fail(VIHDS_E_UNSUPPORTED, "device one-hot wider than 1023");
  launch_offset_rows_bwd(B, S, D, n, src_row, dst_row, accumulate, dev1hot, g_theta, g_wb, (hipStream_t)stream);
  return check_hip("vihds_offset_rows_bwd launch");
}

int vihds_gather_batch(int B, int n_src, int C4, int T, int n_tr, int D, const long long* idx, const float* obs_src,
                       const float* inputs_src, const float* dev1hot_src, float* obs, float* inputs, float* dev1hot,
                       float* delta_obs, void* stream) {
  if (B <= 0 || n_src <= 0 || C4 <= 0 || T <= 1 || n_tr < 0 || D < 0) return fail(VIHDS_E_BADARG, "bad sizes");
  if (!idx || !obs_src || !obs || (n_tr > 0 && (!inputs_src || !inputs)) || (D > 0 && (!dev1hot_src || !dev1hot)))
    return fail(VIHDS_E_BADARG, "null argument");
  launch_gather_batch(B, n_src, C4, T, n_tr, D, idx, obs_src, inputs_src, dev1hot_src, obs, inputs, dev1hot, delta_obs,
                      (hipStream_t)stream);
  return check_hip("vihds_gather_batch launch");
}

int vihds_adam_step(const vihds_adam_tensors* t, float* m, float* v, float* state, const float* lr_dev, float lr,
                    float beta1, float beta2, float eps, float grad_scale, const float* gate, void* stream) {
  if (!t || !m || !v || !state) return fail(VIHDS_E_BADARG, "null argument");
  if (t->n < 0 || t->n > VIHDS_ADAM_MAX_TENSORS) return fail(VIHDS_E_BADARG, "tensor count out of range");
  for (int k = 0; k < t->n; ++k)
    if (t->size[k] < 0 || !t->param[k]) return fail(VIHDS_E_BADARG, "bad tensor table entry");
  launch_adam(*t, m, v, state, lr_dev, lr, beta1, beta2, eps, grad_scale, gate, (hipStream_t)stream);
  return check_hip("vihds_adam_step launch");
}

int vihds_step_tail_supported(const vihds_encoder_shape* s, int P, int S) {
  if (!s || P <= 0 || S <= 0 || check_encoder_shape(s) != VIHDS_OK) return 0;
  return step_tail_supported(*s, P, S) ? 1 : 0;
}

int vihds_step_tail(const vihds_encoder_shape* s, const vihds_step_tail_args* a, void* stream) {
  if (int rc = check_encoder_shape(s)) return rc;
  if (!a) return fail(VIHDS_E_BADARG, "null argument");
  if (a->P <= 0 || a->S <= 0) return fail(VIHDS_E_BADARG, "P, S must be > 0");
  if (!a->kind || !a->q_all || !a->q_rows || !a->p_mu || !a->p_prec || !a->clip_lo || !a->clip_hi || !a->u ||
      !a->g_theta_unit || !a->g_all)
    return fail(VIHDS_E_BADARG, "null theta-side argument");
  const vihds_iwae_job& j = a->iwae;
  if (!j.logp || !j.log_w || !j.lse || !j.loss || j.n_iwae_total <= 0)
    return fail(VIHDS_E_BADARG, "vihds_iwae_job: null buffer or n_iwae_total <= 0");
  if (!a->delta_obs || !a->lin_w || !a->pooled || !a->hidden || !a->g_pre || !a->g_conv)
    return fail(VIHDS_E_BADARG, "null encoder-side argument");
  if (s->nl > 0 && !a->local_w) return fail(VIHDS_E_BADARG, "missing local head weights");
  if (((s->l_tr || s->g_tr) && s->n_tr > 0 && !a->inputs) || ((s->l_dv || s->g_dv) && s->D > 0 && !a->dev1hot))
    return fail(VIHDS_E_BADARG, "missing conditioning input");
  const bool need[8] = {s->ngl > 0, true, true, true, true, s->nl > 0, false, s->ng > 0};
  for (int k = 0; k < 8; ++k) {
    if (need[k] && (!a->param[k] || !a->grad[k])) return fail(VIHDS_E_BADARG, "missing parameter / gradient tensor");
    if (a->param[k] && (!a->grad[k] || a->mv_offset[k] < 0)) return fail(VIHDS_E_BADARG, "bad tensor table entry");
  }
  if (a->state && (!a->m || !a->v)) return fail(VIHDS_E_BADARG, "state without moment buffers");
  // ABI 13: decoder-side tensors, the offset layer, shifted gradient rows
  if (a->n_extra < 0 || a->n_extra > VIHDS_TAIL_MAX_EXTRA) return fail(VIHDS_E_BADARG, "n_extra out of range");
  for (int k = 0; k < a->n_extra; ++k) {
    const vihds_tail_tensor& t = a->extra[k];
    if (!t.param || !t.grad || !t.grad_src || t.size <= 0 || t.nparts <= 0 || t.mv_offset < 0 ||
        (t.nparts > 1 && t.part_stride <= 0))
      return fail(VIHDS_E_BADARG, "bad decoder-side tensor entry");
  }
  if (a->off_n < 0 || (a->off_n > 0 && (!a->off_w || !a->off_b || !a->off_gw || !a->off_gb || !a->off_rowsum ||
                                        !a->dev1hot || s->D <= 0 || a->off_row0 < 0 || a->off_mv_w < 0 || a->off_mv_b < 0)))
    return fail(VIHDS_E_BADARG, "bad offset-layer entry");
  if (a->off_n * (s->D + 1) > 256) return fail(VIHDS_E_UNSUPPORTED, "offset layer larger than one block of the update launch");
  if (a->g_shift_n < 0 || (a->g_shift_n > 0 && (a->g_shift_lo < 0 || a->g_shift_lo + a->g_shift_n > a->P ||
                                                a->g_shift_lo + a->g_shift < 0)))
    return fail(VIHDS_E_BADARG, "bad gradient-row shift");
  if (!step_tail_supported(*s, a->P, a->S))
    return fail(VIHDS_E_UNSUPPORTED, "vihds_step_tail: working set exceeds the 60 KB LDS budget (or more than 10 filter taps)");
  launch_step_tail(*s, *a, (hipStream_t)stream);
  return check_hip("vihds_step_tail launch");
}

}  // extern "C"
