// Kernel argument block shared by every ODE kernel (passed by value).
#pragma once
#include "../../include/vihds_hip.h"

namespace vihds {

struct OdeArgs {
  int B, S, T, C, n;  // n = B*S
  int solver;
  int kernel_variant;  // 0 auto, 1 thread-per-trajectory, 2 lane-split
  int logp_grad_broadcast;  // backward: g_logp is one [B][S] array applied to all four species
  int slot_row[VIHDS_MAX_SLOTS];
  const float* theta;
  const float* cond;
  const float* dev1hot;  // [B][D] (dr_blackbox only)
  int D, n_const;        // device_depth; blackbox: #time-invariant MLP inputs
  const float* times;
  const float* obs;
  const float* weights;  // shared neural weights (NULL for white-box models)
  float* g_weights;      // backward: += gradient of the shared weights
  float* traj;
  float* xpred;
  float* logp;
  const float* traj_in;  // backward: stored trajectory
  const float* g_traj;
  const float* g_xpred;
  const float* g_logp;
  float* g_theta;
  float* aux;            // backward scratch (dr_blackbox: per-evaluation dump for the weight-gradient GEMMs)
  float init_latent, init_prec;
  int n_hidden_prec;     // white-box models with neural precisions: hidden units of the precision network (0: none)
  // backward, vihds_ode_bwd_elbo: the log-likelihood gradient is the importance weight, formed in the kernel from these
  // (vihds_iwae_inline.hpp); iw_logp NULL: g_logp as given
  const float* iw_logp;   // [4][B][S]
  const float* iw_log_p;  // [B][S] or NULL
  const float* iw_log_q;  // [B][S] or NULL
  // telemetry of the time-parallel decoder kernel (vihds_debug_newton_hist; NULL: none): [0..32] wavefronts by the number of
  // walks their Newton iteration over the OD chain took, [33] wavefronts that left through the first-order correction
  unsigned int* newton_hist;
};

// Everything the sampling stage needs when it runs inside the decoder-step kernel (vihds_theta_ode_logp_grad):
// ChainedDistribution.sample + p.clip + log q + log p, then the device conditioner rows.
struct ThetaStageArgs {
  int P;
  const int* kind;
  const float *q_mu, *q_prec;
  const int* q_rows;
  int prec_is_log;
  const float *p_mu, *p_prec, *clip_lo, *clip_hi;
  float* u;
  unsigned int* rng;
  int S_total, s_off;
  float* theta;  // same buffer as OdeArgs::theta, written here
  int n_rows;    // rows of theta
  float *log_q, *log_p;
  // device conditioner (E = 0: none)
  int E, cond_row0;
  float w_mean, w_std;
  const float* z;
  unsigned int* crng;
  const float* rel;
  const int* is_default;
  // dr_blackbox's condition_theta inside the stage (vihds_theta_stage.hpp; off_n = 0: none):
  // theta[off_dst + k] = theta[off_src + k] + off_w[k][.] . dev1hot[b][.] + off_b[k]
  int off_n, off_src, off_dst;
  const float *off_w, *off_b;
};

// vihds_ode_fwd_summaries: the evaluation's second forward pass (csrc/vihds_ode_kernels.hpp, ode_fwd_summ_kernel)
struct SummArgs {
  const float* log_w;  // [B][S] unnormalised log importance weights
  const float* lse;    // [B] their row-wise logsumexp
  float* partial;      // [B][nch][T][nvp] per-wavefront sums
  int nch, nvp;
};

}  // namespace vihds
