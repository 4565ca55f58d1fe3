/*
 * vihds_hip.h -- C ABI of the MI355X (gfx950) implementation of vi-hds's batched ODE-integration + ELBO path.
 *
 * This is the drop-in boundary.  The reference is pure Python/PyTorch, so the binding
 * a maintainer adds is a ctypes stub; every entry point takes plain pointers and sizes, is
 * asynchronous on the given HIP stream, allocates nothing, keeps no global state besides a thread-local
 * error string, and returns 0 on success or a negative code (VIHDS_E_*).
 *
 * Memory layout (all fp32, all buffers caller-owned device memory):
 *   n_traj = B*S trajectories, flat index i = b*S + s (IWAE sample fastest).
 *   theta   [n_rows][B][S]   structure-of-arrays: one row per named parameter; `slot_row` maps the kernel's
 *                            fixed per-model slot order (vihds_model_slot_name) to rows of this buffer.
 *   cond    [B][C]           log(1+c) treatments (vihds/datasets.py:87)
 *   dev1hot [B][D]           device one-hot blocks (only dr_blackbox reads it)
 *   times   [T]              observation time grid
 *   obs     [B][4][T]        observations, the reference's own layout (vihds/training.py:47-52)
 *   traj    [T][N][B][S]     ODE solution.  The reference returns sol.permute(1,2,3,0) -- a non-contiguous
 *                            [B,S,N,T] view of a [T,B,S,N] stack (vihds/ode.py:82); the host wraps this buffer
 *                            in the equivalent strided view.
 *   xpred   [T][4][B][S]     observed signals (vihds/ode.py:84-93)
 *   logp    [4][B][S]        sum over time of the Gaussian observation log-density (vihds/training.py:24-44)
 */
#ifndef VIHDS_HIP_H
#define VIHDS_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define VIHDS_ABI_VERSION 14

/* error codes */
#define VIHDS_OK 0
#define VIHDS_E_BADARG (-1)   /* null pointer / bad size / unknown enum */
#define VIHDS_E_UNSUPPORTED (-2)
#define VIHDS_E_HIP (-3)      /* a HIP runtime call failed; see vihds_last_error() */

/* models: keys of models.LOOKUP (reference models/__init__.py:19-35) */
enum vihds_model {
  VIHDS_MODEL_DR_CONSTANT = 0,                /* models/dr_constant.py:115  (version 1) */
  VIHDS_MODEL_DR_CONSTANT_V2 = 1,             /* models/dr_constant.py:163  */
  VIHDS_MODEL_AUTO_CONSTANT = 2,              /* models/auto_constant.py:66 */
  VIHDS_MODEL_PRPR_CONSTANT = 3,              /* models/prpr_constant.py:72 */
  VIHDS_MODEL_RELAY_CONSTANT = 4,             /* models/relay_constant.py:137 */
  VIHDS_MODEL_DEGRADER_CONSTANT = 5,          /* models/degrader_constant.py:146 */
  VIHDS_MODEL_DR_CONSTANT_PRECISIONS = 6,     /* models/dr_constant.py:169 */
  VIHDS_MODEL_DR_CONSTANT_PRECISIONS_V2 = 7,  /* models/dr_constant.py:212 */
  VIHDS_MODEL_AUTO_CONSTANT_PRECISIONS = 8,   /* models/auto_constant.py:100 */
  VIHDS_MODEL_PRPR_CONSTANT_PRECISIONS = 9,   /* models/prpr_constant.py:101 */
  VIHDS_MODEL_RELAY_CONSTANT_PRECISIONS = 10, /* models/relay_constant.py:199 */
  VIHDS_MODEL_DEGRADER_CONSTANT_PRECISIONS = 11, /* models/degrader_constant.py:195 */
  VIHDS_MODEL_DR_BLACKBOX = 12,               /* models/dr_blackbox.py:61 */
  VIHDS_MODEL_INDUCER_CONSTANT = 13,          /* models/inducer_constant.py:83 */
  VIHDS_MODEL_INDUCER_CONSTANT_PRECISIONS = 14, /* models/inducer_constant.py:117 */
  VIHDS_MODEL_DEBUG_CONSTANT = 15,            /* models/debug.py:11 */
  VIHDS_MODEL_COUNT = 16
};

/* solvers: values of params.solver (reference vihds/ode.py:75-81, vihds/config.py:59) */
enum vihds_solver {
  VIHDS_SOLVER_MODEULER = 0,      /* vihds/solvers.py:9-17   h = times[1]-times[0] for every step */
  VIHDS_SOLVER_MODEULERWHILE = 1, /* vihds/solvers.py:20-41  h = t2-t1 */
  VIHDS_SOLVER_EULER = 2,         /* torchdiffeq==0.1 fixed-grid 'euler'    (restated; parity unpinned) */
  VIHDS_SOLVER_MIDPOINT = 3,      /* torchdiffeq==0.1 fixed-grid 'midpoint' (restated; parity unpinned) */
  VIHDS_SOLVER_RK4 = 4,           /* torchdiffeq==0.1 fixed-grid 'rk4' = 3/8 rule (restated; parity unpinned) */
  /* Adaptive pairs of torchdiffeq==0.1 (vihds/ode.py:79-81, tests/test_ode_solvers.py:66-80; restated; parity unpinned).
     With one of these, `times` passed to vihds_ode_fwd / vihds_ode_bwd is the ACCEPTED grid that
     vihds_ode_adaptive_grid returned (which contains the output times), integrated with the pair's higher-order
     tableau; the adjoint is the discrete adjoint of those steps. */
  VIHDS_SOLVER_DOPRI5 = 5,        /* Dormand-Prince 5(4) */
  VIHDS_SOLVER_BOSH3 = 6,         /* Bogacki-Shampine 3(2) */
  VIHDS_SOLVER_ADAPTIVE_HEUN = 7, /* Heun-Euler 2(1) */
  VIHDS_SOLVER_DOPRI8 = 8,        /* `solver: dopri8`: an 8th-order Dormand-Prince pair -- Hairer's DOP853 (12 stages), NOT
                                     torchdiffeq's 8(7) 13-stage tableau (unavailable offline); same controller */
  VIHDS_SOLVER_COUNT = 9
};

#define VIHDS_MAX_SLOTS 64

/* One decoder problem: replaces the argument bundle of OdeModel.simulate + expand_precisions + observe +
 * log_prob_observations (vihds/decoders.py:28-45, vihds/training.py:24-33). */
typedef struct vihds_ode_problem {
  int model;    /* enum vihds_model */
  int solver;   /* enum vihds_solver */
  int B, S, T;  /* data rows, IWAE samples per row, time points */
  int C, D;     /* #conditions (cond row length), device_depth (dev1hot row length) */
  int n_rows;   /* rows of the theta / g_theta buffers */
  int slot_row[VIHDS_MAX_SLOTS]; /* kernel slot -> theta row; must be set for every slot of the model */
  /* sizes of the neural blocks (0 when the model has none); weights buffer layout: hidden W, b; production W, b; degradation W, b (row-major), states network first */
  int n_hidden_prec;   /* params.n_hidden_decoder_precisions (vihds/precisions.py:55) */
  int n_hidden_states; /* params.n_hidden_decoder (dr_blackbox only) */
  int n_latent_states; /* params.n_latent_species (dr_blackbox only) */
  int n_const;         /* dr_blackbox: length of the time-invariant feature vector (n_z+n_x+n_y+C+D) */
  float init_latent;   /* dr_blackbox.py:101 */
  float init_prec;     /* dr_blackbox.py:102 */
  int logp_grad_broadcast; /* backward only: 1 = g_logp is ONE [B][S] array applied to all four species
                              (what the IWAE reduction hands back); 0 = [4][B][S] */
  int kernel_variant;      /* 0 = auto; 1 = one thread per trajectory; 2 = lane-split (8 lanes per trajectory,
                              dr_constant family only; auto picks it below 16 384 trajectories); 3 = dr_constant's
                              time-parallel training kernel (vihds_ode_logp_grad's default where it applies).  (4,
                              dr_blackbox's single-wavefront MFMA kernels, and 5, the time axis in parallel for relay /
                              degrader / prpr / auto_constant, were removed in round 6: superseded / never faster;
                              the values now select what 0 does) */
} vihds_ode_problem;

int vihds_abi_version(void);
const char* vihds_last_error(void);

/* Introspection so the host maps YAML parameter names to kernel slots without duplicating tables. */
int vihds_model_n_states(int model);      /* ODE state size N (incl. 4 precision states for *_precisions) */
int vihds_model_n_species(int model);     /* states handed to observe(): N minus neural precision states */
int vihds_model_n_slots(int model);       /* number of theta slots the kernel reads */
const char* vihds_model_slot_name(int model, int slot); /* reference parameter name of a slot */
int vihds_model_n_weights(const vihds_ode_problem* p);  /* floats in the `weights` buffer (0 if none) */

/* Forward: integrate, observe, and reduce the observation log-likelihood over time.
 * traj / xpred / logp may each be NULL to skip that output.  `weights` is NULL for white-box models. */
int vihds_ode_fwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                  const float* times, const float* obs, const float* weights, float* traj, float* xpred,
                  float* logp, void* stream);

/* Backward (discrete adjoint of the chosen scheme = what autograd computes in the reference).
 * Upstream gradients g_traj [T][N][B][S], g_xpred [T][4][B][S], g_logp [4][B][S] may each be NULL (= zero).
 * Writes g_theta [n_rows][B][S] for every slot row (other rows untouched: pre-zero the buffer).
 * Shared-weight gradients:
 *   - white-box models with neural precisions: ADDED into g_weights (pre-zero it);
 *   - dr_blackbox: the contraction over (trajectory x RHS evaluation) is left to the caller as batched GEMMs:
 *     the kernel fills `aux` (vihds_ode_bwd_aux_floats floats) with the field-major dump
 *     [F = vihds_blackbox_dump_fields()][E evaluations][B*S] (layer inputs and pre-activation gradients; field
 *     order in profiles/LOG.md, appendix section 4.4) followed by Delta [HS+HP][B*S] and the output-bias adjoint sums [2*NX+8][B*S];
 *     g_weights is not touched.
 *   - white-box models with neural precisions, aux != NULL (optional, vihds_ode_bwd_aux_floats floats): the same
 *     scheme -- aux receives [8 + NIN][E][B*S] (fields 0..3 production and 4..7 degradation pre-activation adjoints,
 *     8.. the NIN = 1 + core-states layer inputs tanh([t, species])), only the 8 bias gradients are ADDED into
 *     g_weights, and the caller contracts the two weight matrices with vihds_gram_blocks (rectangles fields 0..3 x
 *     8.. and 4..7 x 8..).  This keeps 2*4*NIN accumulators out of every thread's registers (relay: -36 % time).
 *   - the same models with a hidden layer in the precision network (p->n_hidden_prec = H >= 1, reference
 *     precisions.py:63-74; weights = Wh [H][NIN], bh [H], Wp [4][H], bp [4], Wd [4][H], bd [4]): aux receives
 *     [8 + NIN + 2H][E][B*S] (0..3 / 4..7 the output pre-activation adjoints, 8.. the NIN layer inputs [t, species],
 *     then the H hidden pre-activation adjoints, then the H hidden activations); the output biases are ADDED into
 *     g_weights, Wh / Wp / Wd are three rectangles for vihds_gram_blocks and bh the row sums of the hidden adjoints.
 *     Without aux only g_theta is produced.
 *   aux may be NULL otherwise. */
int vihds_ode_bwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                  const float* times, const float* obs, const float* weights, const float* traj,
                  const float* g_traj, const float* g_xpred, const float* g_logp, float* g_theta,
                  float* g_weights, float* aux, void* stream);

/* vihds_ode_bwd for the ELBO's own upstream gradient (ABI 13): the log-likelihood gradient of every (signal, row, sample) is
 * the importance weight d loss / d log_w[b][s] = -(1/B) softmax_s(log_w[b][.]), log_w = sum_j logp[j] + log_p - log_q
 * (reference training.py:135-149), and the kernel forms it itself from logp [4][B][S], log_p, log_q [B][S] (either may be
 * NULL) -- every wavefront the row-wise logsumexp of its trajectories' data rows -- so a training step needs no IWAE launch
 * between the forward and the adjoint.  Everything else as vihds_ode_bwd (no g_traj / g_xpred); fixed-grid solvers. */
int vihds_ode_bwd_elbo(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                       const float* times, const float* obs, const float* weights, const float* traj, const float* logp,
                       const float* log_p, const float* log_q, float* g_theta, float* g_weights, float* aux,
                       void* stream);
long long vihds_ode_bwd_aux_floats(const vihds_ode_problem* p);
/* 1: vihds_ode_bwd itself completes g_weights for this problem (relay_constant_precisions below 16 384 trajectories on a
 * fixed-grid solver: sixteen lanes per trajectory, the precision network's weight gradients accumulated per lane, one
 * partial row per block in aux -- vihds_ode_bwd_aux_floats is then that small size -- and added up in block order by a
 * second launch of the same call); 0: aux receives the dump described above and the caller contracts it. */
int vihds_ode_bwd_reduces_weights(const vihds_ode_problem* p);
/* Layout of the traj / xpred buffers of vihds_ode_fwd (and of traj / g_traj / g_xpred of vihds_ode_bwd) for this problem:
 * always 0 = [T][N][B][S] and [T][4][B][S]; logp is [4][B][S].  (1 = time fastest was the layout of the removed
 * kernel_variant 5; the entry point stays so that ABI 14 is unchanged.) */
int vihds_ode_traj_layout(const vihds_ode_problem* p);

/* torchdiffeq==0.1's adaptive algorithm ITSELF, resident on the device (round 4; reference call site vihds/ode.py:79-81,
 * `odeint(func, y0, t, method="dopri5" | "bosh3" | "adaptive_heun")`; csrc/vihds_rk_adaptive_device.hpp): one persistent launch
 * holds the step size and the accept / reject decisions in device memory (one grid barrier per trial step, no host round trip:
 * asynchronous on `stream` and hipGraph-capturable), steps are NOT shortened to hit output times -- every output time is
 * evaluated from the quartic interpolant (interp._interp_fit) of the accepted step that contains it -- and the accepted steps
 * (start time, size, state) are logged in `workspace` for vihds_ode_adaptive_bwd, the discrete adjoint of exactly that
 * computation with the step sizes held constant.  Models without shared neural weights, at most 65 536 trajectories;
 * VIHDS_E_UNSUPPORTED otherwise (then: vihds_ode_adaptive_grid below).
 *   times: [T] on the DEVICE.  traj: [T][N][B][S], the solution at the output times.  g_traj: its upstream gradient.
 *   workspace: vihds_ode_adaptive_tape_floats(p, max_steps) floats; its first 32-bit words after the launch:
 *   [1] error (0 ok, 1 barrier time-out, 2 more than max_steps accepted steps, 3 step-size underflow, 4 non-finite error
 *   estimate), [2] accepted steps, [3] rejected steps. */
long long vihds_ode_adaptive_tape_floats(const vihds_ode_problem* p, int max_steps);
int vihds_ode_adaptive_fwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                           const float* times, float rtol, float atol, int max_steps, float* workspace, float* traj,
                           void* stream);
int vihds_ode_adaptive_bwd(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                           const float* times, int max_steps, const float* workspace, const float* g_traj, float* g_theta,
                           void* stream);
/* The same pair for the white-box models WITH neural precisions (*_precisions, no hidden layer: every such spec of the
 * reference; ABI 13): weights as for vihds_ode_fwd; g_weights [n_weights] receives += the network's weight gradient
 * (per-thread accumulators, one atomic per wavefront and entry: zero it first).  Models without a network: weights /
 * g_weights NULL, identical to the calls above.  dr_blackbox and a hidden precision layer: VIHDS_E_UNSUPPORTED. */
int vihds_ode_adaptive_fwd_w(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                             const float* weights, const float* times, float rtol, float atol, int max_steps, float* workspace,
                             float* traj, void* stream);
int vihds_ode_adaptive_bwd_w(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                             const float* weights, const float* times, int max_steps, const float* workspace,
                             const float* g_traj, float* g_theta, float* g_weights, void* stream);

/* Adaptive solvers (VIHDS_SOLVER_DOPRI5 / BOSH3 / ADAPTIVE_HEUN / DOPRI8; reference vihds/ode.py:79-81 -> torchdiffeq==0.1
 * odeint / odeint_adjoint, absent from the tree: restated, parity unpinned).  Step-size controller, SYNCHRONOUS on
 * `stream` (one device round trip per trial step): walks the whole batch from times_host[0] to times_host[T-1] with ONE
 * step size for all trajectories (error ratio = mean over all state elements of (err / (atol + rtol max(|y0|,|y1|)))^2,
 * accepted when <= 1; next step = step / max(0.1, min(ratio^(1/(2 order)) / 0.9, 1 / dfactor)), dfactor 0.2 after a
 * rejection), every accepted step clipped to the next output time.  Returns the number G of grid points written to
 * grid_host (host memory, capacity max_grid), index_host[k] = position of output time k in the grid; negative: error.
 * The caller then runs vihds_ode_fwd / vihds_ode_bwd with p->T = G and `times` = the grid (device copy): they integrate
 * with the pair's higher-order tableau on it, and the adjoint is the discrete adjoint of those steps.
 * workspace: vihds_ode_adaptive_workspace_floats(p) floats of device memory. */
long long vihds_ode_adaptive_workspace_floats(const vihds_ode_problem* p);
int vihds_ode_adaptive_grid(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                            const float* weights, const float* times_host, float rtol, float atol, float* workspace,
                            float* grid_host, int max_grid, int* index_host, void* stream);

/* Training fast path: the per-species log-likelihood logp [4][B][S] of vihds_ode_fwd AND, in the same launch, the
 * gradient g_theta_unit [n_rows][B][S] that vihds_ode_bwd would return for g_logp == 1 (g_traj = g_xpred = NULL).
 * In the ELBO d loss / d logp[j][b][s] is one number w[b][s] for all four signals (training.py:135-149) and the
 * adjoint is linear in it, so d loss / d theta = w[b][s] * g_theta_unit: the adjoint can run right behind the forward
 * sweep, the trajectory never leaves LDS, and neither trajectory nor x_predict is written.  dr_constant /
 * dr_constant_v2 in the lane-split regime only; VIHDS_E_UNSUPPORTED otherwise (callers then use fwd + bwd). */
int vihds_ode_logp_grad(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                        const float* times, const float* obs, float* logp, float* g_theta_unit, void* stream);
int vihds_blackbox_dump_fields(void);  /* at the built-in (specs/dr_blackbox_icml.yaml) sizes */

/* dr_blackbox at other network sizes (reference models/dr_blackbox.py:61-84 reads n_latent_species, n_hidden_decoder,
 * n_hidden_decoder_precisions, n_z, n_x, n_y from the YAML).  The kernels are compiled per size set: the ICML set is
 * part of this library; any other set is looked for, on first use, as
 *     libvihds_bb_<n_latent_states>_<n_hidden_states>_<n_hidden_prec>_<n_const - C - D>.so
 * in the directory this library was loaded from (build: make -C vi-hds_amd/csrc blackbox L=.. HS=.. HP=.. NLAT=..;
 * thread-per-trajectory kernels, every solver, weight gradients through the dump).  A missing file makes every entry
 * point return VIHDS_E_UNSUPPORTED with the file name and the build command in vihds_last_error().  The theta slots of
 * a sized problem are its n_const - C - D latent inputs in YAML order (z.., x.., y..) followed by init_x, init_rfp,
 * init_yfp, init_cfp.  The problem-level queries below answer for any model (they fall back to the vihds_model_*
 * values); vihds_model_n_states / n_slots / slot_name describe dr_blackbox at the ICML sizes. */
int vihds_problem_n_states(const vihds_ode_problem* p);
int vihds_problem_n_slots(const vihds_ode_problem* p);
int vihds_problem_dump_fields(const vihds_ode_problem* p); /* dr_blackbox only */

/* theta side: ChainedDistribution.sample + p.clip + q.log_prob + p.log_prob
 * (vihds/distributions.py:64-85,119-142,327-381; vihds/vae.py:31-34) over all P parameters at once.
 *   kind   [P]      0 Normal, 1 LogNormal, 2 Constant
 *   q_mu   [P][B], q_prec [P][B]   (global / constant parameters broadcast over B by the caller)
 *   p_mu   [P],    p_prec [P]
 *   clip_lo[P],    clip_hi[P]      p.clip bounds: mu -/+ stddevs*sigma, exp'd for LogNormal
 *                                  (distributions.py:332-336,377-381); ignored for Constant
 *   u      [B][S][P]               reference layout (vihds/vae.py:22-24)
 *   theta  [n_rows>=P][B][S] rows 0..P-1 written;  log_q, log_p [B][S] */
/* IWAE loss folded into vihds_theta_bwd (training.py:135-149 + the backward of everything up to q): with opts->iwae set,
 * every block forms its data row's importance weights itself -- log_w = sum_j logp[j] + log_p - log_q, row max and
 * sum-exp, lse -- and uses d loss / d log_w = -softmax / B as the factor of g_theta and as the log q / log p upstream
 * gradients (g_log_q, g_log_p and opts->g_theta_scale are then ignored); the blocks of parameter chunk 0 write log_w
 * and lse, and the last of them -ELBO = -mean_b(lse - log n_iwae_total) to loss.  One launch fewer per training step:
 * the separate vihds_iwae_loss_fwd is not needed when nothing else consumes its outputs before the backward. */
typedef struct vihds_iwae_job {
  const float* logp;   /* [4][B][S] */
  const float* log_p;  /* [B][S] or NULL */
  const float* log_q;  /* [B][S] or NULL */
  int n_iwae_total;
  float* log_w;        /* [B][S] */
  float* lse;          /* [B] */
  float* loss;         /* [1] */
  unsigned int* ticket; /* one device word, zero before the first call, left at zero */
} vihds_iwae_job;

typedef struct vihds_theta_opts {
  /* Where q's parameters live.  NULL: parameter p is row p of q_mu and row p of q_prec.  Else [2P]: entries 0..P-1
   * are the rows of q_mu holding mu_p, entries P..2P-1 the rows of q_prec holding prec_p (the encoder's level-blocked
   * table [local mu; local log-prec; global-conditioned mu; ...] is then consumed, and its gradient produced, in
   * place; g_q_mu / g_q_prec are indexed the same way). */
  const int* q_rows;
  /* q_prec holds log-precisions (what the encoder heads emit, encoders.py:150-165): the kernel exponentiates and the
   * adjoint returns d/d log_prec. */
  int q_prec_is_log;
  /* NULL: u is an input (the reference's host draw, vae.py:22-24).  Else 4 device words {seed lo, seed hi, step,
   * ticket}: the forward draws u ~ N(0,1) itself (Philox4x32-10 + Box-Muller, see csrc/vihds_elbo.hip), WRITES it
   * to u, and the last block advances `step` - fresh draws on every replay of a captured graph. */
  unsigned int* rng;
  /* rng: this call covers samples s_offset .. s_offset+S-1 of S_total per data row (S sharded over ranks: every rank
   * gets the slice of the same global draw). */
  int S_total, s_offset;
  /* vihds_theta_bwd only: [B][S] factor applied to g_theta (g_theta[p][b][s] * g_theta_scale[b][s]) -- lets the
   * unit-weight gradient of vihds_ode_logp_grad be consumed without a separate scaling pass. */
  const float* g_theta_scale;
  /* vihds_theta_bwd only: NULL, or the IWAE loss to evaluate inside the launch (see vihds_iwae_job). */
  const vihds_iwae_job* iwae;
} vihds_theta_opts;
int vihds_theta_fwd(int P, int B, int S, const int* kind, const float* q_mu, const float* q_prec,
                    const float* p_mu, const float* p_prec, const float* clip_lo, const float* clip_hi,
                    float* u, float* theta, float* log_q, float* log_p, const vihds_theta_opts* opts /* or NULL */,
                    void* stream);
/* g_theta [P..][B][S], g_log_q, g_log_p [B][S] (each may be NULL) -> g_q_mu, g_q_prec [P][B] (overwritten) */
int vihds_theta_bwd(int P, int B, int S, const int* kind, const float* q_mu, const float* q_prec,
                    const float* p_mu, const float* p_prec, const float* clip_lo, const float* clip_hi,
                    const float* u, const float* g_theta, const float* g_log_q, const float* g_log_p,
                    float* g_q_mu, float* g_q_prec, const vihds_theta_opts* opts /* or NULL */, void* stream);

/* The decoder side of a training step in ONE launch (dr_constant / dr_constant_v2, lane-split regime): what
 * vihds_theta_fwd, vihds_device_condition and vihds_ode_logp_grad do one after the other -- sample and clip theta with
 * log q / log p, the device-conditioner rows, log-likelihood, unit-weight adjoint -- with theta handed from the
 * sampling stage to the integrator inside the block.  theta [n_rows][B][S], u, log_q, log_p, logp, g_theta_unit as in
 * those calls.  opts: q_rows / q_prec_is_log / rng / (S_total, s_offset) as for vihds_theta_fwd.  conditioner: NULL or
 * E rows starting at first_row (>= P). */
typedef struct vihds_conditioner {
  int E, first_row;
  float w_mean, w_std;
  const float* z;        /* [E][D] standard normals, or NULL with rng */
  unsigned int* rng;     /* 4 device words, a state of its own (see vihds_device_condition) */
  const float* relevance; /* [E][D] */
  const int* is_default;  /* [E] */
} vihds_conditioner;
int vihds_theta_ode_logp_grad(const vihds_ode_problem* p, int P, const int* kind, const float* q_mu,
                              const float* q_prec, const float* p_mu, const float* p_prec, const float* clip_lo,
                              const float* clip_hi, float* u, const vihds_theta_opts* opts,
                              const vihds_conditioner* conditioner, const float* cond, const float* dev1hot,
                              const float* times, const float* obs, float* theta, float* log_q, float* log_p,
                              float* logp, float* g_theta_unit, void* stream);

/* The sampling stage as a prologue of the ODE FORWARD launch (ABI 13), for every model whose forward kernels own a
 * contiguous run of trajectories per block -- the lane-split kernels of relay / degrader / prpr / auto_constant and their
 * _precisions forms, dr_blackbox's cooperating wavefronts: vihds_theta_fwd (+ dr_blackbox's condition_theta, the offset
 * layer of models/dr_blackbox.py:86-96, as vihds_offset_rows_fwd) + vihds_ode_fwd in ONE launch, each block sampling its own
 * trajectories' parameters.  Same outputs as the separate calls: theta [n_rows][B][S] (rows dst_row.. of the offset layer
 * included), u (written when opts->rng draws it), log_q, log_p, traj, xpred (may be NULL), logp.
 * The in-kernel generator's step counter (opts->rng[2]) is READ, not advanced: the launch that follows in a training step
 * advances it (vihds_step_tail_args.rng_advance), or vihds_rng_advance.  VIHDS_E_UNSUPPORTED for any other model / kernel
 * variant / shape: use the separate calls. */
typedef struct vihds_offset_layer {
  int n, src_row, dst_row; /* theta[dst_row + k] = theta[src_row + k] + W[k][.] . dev1hot[b][.] + bias[k], k < n */
  const float* W;          /* [n][D] */
  const float* bias;       /* [n] */
} vihds_offset_layer;
int vihds_theta_ode_fwd(const vihds_ode_problem* p, int P, const int* kind, const float* q_mu, const float* q_prec,
                        const float* p_mu, const float* p_prec, const float* clip_lo, const float* clip_hi, float* u,
                        const vihds_theta_opts* opts, const vihds_offset_layer* offset /* or NULL */, const float* cond,
                        const float* dev1hot, const float* times, const float* obs, const float* weights, float* theta,
                        float* log_q, float* log_p, float* traj, float* xpred, float* logp, void* stream);
/* rng[2] += 1 (one thread): the step of an in-kernel generator state {seed lo, seed hi, step, ticket} */
int vihds_rng_advance(unsigned int* rng, void* stream);

/* IWAE reduction (vihds/training.py:135-149):  log_w = sum_j logp[j] + log_p - log_q;  per row b:
 * row_max[b] = max_s log_w, row_sumexp[b] = sum_s exp(log_w - row_max[b]).  The host finishes
 * lse = row_max + log(row_sumexp) (after the cross-rank combine when S is sharded) and the mean over B. */
int vihds_iwae_fwd(int B, int S, const float* logp, const float* log_p, const float* log_q, float* log_w,
                   float* row_max, float* row_sumexp, void* stream);
/* g_lse [B], lse [B] -> g_logw [B][S] = g_lse[b] * exp(log_w - lse[b]) */
int vihds_iwae_bwd(int B, int S, const float* log_w, const float* lse, const float* g_lse, float* g_logw,
                   void* stream);

/* Single-process convenience: the two calls above plus the finish, i.e. all of vihds/training.py:135-149:
 * lse[b] = row_max + log(row_sumexp), loss[0] = -mean_b(lse[b] - log(n_iwae_total)).
 * Backward: g_logw[b][s] = -(g_loss[0]/B) * exp(log_w - lse[b]); g_neg_logw (optional) receives its negation, the
 * gradient w.r.t. log_q.
 * unit_g_logw / unit_g_neg_logw (optional, only where vihds_iwae_loss_unit_grad(B,S) == 1): the forward also writes
 * the backward's outputs for g_loss = 1 -- what loss.backward() asks for in the training step -- so that step needs no
 * backward launch for the loss.
 * ticket (optional): a device counter the caller owns, zero before the first call and not shared between launches
 * that may run concurrently.  With it (and S <= 1024) the reduction runs one block per row and the last block to
 * finish takes the mean over rows and resets the counter (graph-replayable); without it B <= 64, S <= 256 run in one
 * block and larger shapes in two launches. */
int vihds_iwae_loss_fwd(int B, int S, int n_iwae_total, const float* logp, const float* log_p, const float* log_q,
                        float* log_w, float* row_max, float* row_sumexp, float* lse, float* loss, float* unit_g_logw,
                        float* unit_g_neg_logw, unsigned int* ticket, void* stream);
int vihds_iwae_loss_unit_grad(int B, int S, int with_ticket);
/* S sharded over n_ranks processes: `gathered` [n_ranks][2][B] holds every rank's (row_max, row_sumexp) from
 * vihds_iwae_fwd (one all-gather).  Computes the global lse[b] = M + log sum_r se_r exp(m_r - M), loss[0] =
 * -mean_b(lse[b] - log n_iwae_total) and, optionally, this rank's d loss / d log_w [B][S] for a unit upstream gradient
 * (and its negation) -- the rest of training.py:141-149 in one launch. */
int vihds_iwae_combine(int n_ranks, int B, int S, int n_iwae_total, const float* gathered, const float* log_w,
                       float* lse, float* loss, float* unit_g_logw, float* unit_g_neg_logw, void* stream);
int vihds_iwae_loss_bwd(int B, int S, const float* log_w, const float* lse, const float* g_loss, float* g_logw,
                        float* g_neg_logw, void* stream);

/* dr_blackbox's condition_theta (reference models/dr_blackbox.py:86-96): y_i += offset_layer(dev_1hot)_i with
 * offset_layer = Linear(D, n).  theta [n_rows][B][S]; W [n][D], bias [n], dev1hot [B][D].
 * fwd: theta[dst_row + i][b][s] = theta[src_row + i][b][s] + W[i,:] . dev1hot[b,:] + bias[i]   (the simulator's slots
 *      for y point at the dst rows; the src rows keep the sampled y that log q / log p are taken of).
 * bwd: g_theta[src_row + i] += g_theta[dst_row + i] (accumulate = 0: '=', for callers that left the src rows of g_theta
 *      unwritten and so need no zero fill);  g_wb (n*D + n floats, or NULL) = the layer's weight gradient
 *      [n][D] (sum over b, s of g_theta[dst_row + i][b][s] dev1hot[b][d]) followed by its bias gradient [n]. */
int vihds_offset_rows_fwd(int B, int S, int D, int n, int n_rows, int src_row, int dst_row, const float* W,
                          const float* bias, const float* dev1hot, float* theta, void* stream);
int vihds_offset_rows_bwd(int B, int S, int D, int n, int n_rows, int src_row, int dst_row, int accumulate,
                          const float* dev1hot, float* g_theta, float* g_wb, void* stream);

/* Device-resident batching (reference training.py:108-113,55-68: DataLoader(shuffle=True) + collate_merged build every
 * batch on the host): the whole training set stays in HBM (obs_src [n_src][C4][T], inputs_src [n_src][n_tr], dev1hot_src
 * [n_src][D]) and a step's batch is gathered by row index -- idx [B] int64 on the device, e.g. one batch of the epoch's
 * permutation -- into obs [B][C4][T], inputs [B][n_tr], dev1hot [B][D], with delta_obs [B][C4][T-1] (the encoder's first
 * differences, encoders.py:385; may be NULL) formed on the way.  One launch; capturable (the index buffer is refreshed
 * between replays). */
int vihds_gather_batch(int B, int n_src, int C4, int T, int n_tr, int D, const long long* idx, const float* obs_src,
                       const float* inputs_src, const float* dev1hot_src, float* obs, float* inputs, float* dev1hot,
                       float* delta_obs, void* stream);

/* OdeModel.device_conditioner applied to a tensor of ones (vihds/ode.py:43-58; models/dr_constant.py:124-131), for E
 * parameters at once: out[e][b][s] = (is_default[e] ? 1 : 0) + relu(sum_d (w_mean + w_std*z[e][d]) * dev1hot[r][d] *
 * relevance[e][d]) with r = (b*S+s) mod B (the reference's .repeat tiling, kept).  z [E][D] are standard normals
 * (w_mean=2, w_std=1.5 reproduce DeviceConditioner's init) or final weights (w_mean=0, w_std=1).
 * rng (optional; 4 device words as vihds_theta_opts.rng, a state of its own): the kernel draws z itself and advances
 * the step - the reference re-randomises the conditioner on every call (ode.py:48), this keeps that inside a
 * captured graph without a launch for the draw.  z may then be NULL.
 * (S_total, s_offset): this call covers samples s_offset.. of S_total per row (S sharded over ranks); the tiling
 * index r is taken on the global grid, r = (b*S_total + s_offset + s) mod B.  S_total <= 0 means S_total = S. */
int vihds_device_condition(int E, int B, int S, int S_total, int s_offset, int D, float w_mean, float w_std,
                           const float* z, unsigned int* rng, const float* dev1hot, const float* relevance,
                           const int* is_default, float* out, void* stream);

/* Evaluation summaries (vihds/utils.py:79-99, Results.init) on device: importance-weighted mean / std of the
 * predictions, mean of the states, mean of 1/precision.  w = exp(log_w - lse).
 *   prec_theta_rows: for constant precisions the 4 theta rows holding prec_*; prec_traj: neural precisions
 *   taken from traj states [n_species..n_species+3]. */
int vihds_iw_summaries(int B, int S, int T, int N_total, int n_species, const float* log_w, const float* lse,
                       const float* traj, const float* xpred, const float* theta, const int* prec_rows,
                       float* iw_predict_mu /*[B][4][T]*/, float* iw_predict_std /*[B][4][T]*/,
                       float* iw_states /*[B][n_species][T]*/, float* iw_variance /*[B][4][T]*/, void* stream);

/* The same summaries without a stored x_predict: the four observed signals are formed from the states the kernel reads
 * anyway, by the model's observation map (OdeModel.observe, vihds/ode.py:84-93; overridden by the inducer and the
 * direct-read models) -- the forward launch can then be called with xpred == NULL and the evaluation pass neither
 * writes nor re-reads the [T][4][B][S] array.  observe_kind: */
#define VIHDS_OBS_DEFAULT 0 /* x, x*y1, x*(y2+y4), x*(y3+y5)   (needs n_species >= 6) */
#define VIHDS_OBS_DIRECT 1  /* x, x*y1, x*y2, x*y3             (needs n_species >= 4) */
#define VIHDS_OBS_INDUCER 2 /* x, x*y1, x*(y2+y3), x*y4        (needs n_species >= 5) */
/* Tuning knob of both summaries entry points (process-wide, read at launch): how many consecutive time points one block
 * of the pipelined kernel walks (S a multiple of 4 in 512..1024, <= 16 species).  0 = automatic (as many as leave >= 2048
 * blocks, 2 to 8), > 0 = exactly that many, < 0 = the one-block-per-time-point kernel.  Returns the previous value. */
int vihds_iw_summaries_plan(int time_points_per_block);
int vihds_iw_summaries_states(int B, int S, int T, int N_total, int n_species, int observe_kind, const float* log_w,
                              const float* lse, const float* traj, const float* theta, const int* prec_rows,
                              float* iw_predict_mu, float* iw_predict_std, float* iw_states, float* iw_variance,
                              void* stream);

/* The same summaries WITHOUT the trajectory's round trip through HBM (round 5): the evaluation pass calls vihds_ode_fwd with
 * traj == NULL and xpred == NULL (log-likelihoods only), forms log_w / lse (vihds_iwae_loss_fwd), and this entry point
 * integrates a second time -- the same kernel arithmetic, so the trajectory it sums is bit for bit the one the weights came
 * from -- adding up  w y,  w x_predict,  w (x_predict^2 + 1 / precision)  and  w / precision  per time point on the way
 * (w = exp(log_w - lse)): 26 MB of per-wavefront partial rows instead of 644 MB written and read back at B = 234, S = 1 000.
 * Models: all but dr_blackbox; fixed-grid solvers (VIHDS_E_UNSUPPORTED otherwise: vihds_ode_fwd + vihds_iw_summaries_states).
 *   workspace  vihds_ode_fwd_summaries_workspace_floats(p) floats
 *   outputs    as vihds_iw_summaries (n_species = the model's states without the four precision states of *_precisions)
 * vihds_ode_fwd_summaries_supported(p): 1 when the problem's vihds_ode_fwd launch is the thread-per-trajectory kernel this
 * pass repeats (kernel_variant 1, or an evaluation-sized launch) and the model / solver are served. */
int vihds_ode_fwd_summaries_supported(const vihds_ode_problem* p);
long long vihds_ode_fwd_summaries_workspace_floats(const vihds_ode_problem* p);
int vihds_ode_fwd_summaries(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                            const float* times, const float* weights, const float* log_w, const float* lse,
                            float* workspace, float* iw_predict_mu, float* iw_predict_std, float* iw_states,
                            float* iw_variance, void* stream);

/* q(theta | data) encoder (vihds/encoders.py): ConditionalEncoder.forward :49-55 (Conv1d -> AvgPool1d(stride 1) ->
 * Linear -> tanh), the per-parameter Linear(n,1) heads of Q_Local :143-169 and Q_Global_Cond :187-213, Q_Global's free
 * scalars :216-239 and Q_Constant :242-253 -- everything Encoder.evaluate_q :383-404 computes, as ONE forward launch
 * writing the [2P][B] table of means and log-precisions the theta kernel consumes (row order = vihds_theta_opts.q_rows
 * of the level-blocked layout: [local mu; local log_prec; gcond mu; gcond log_prec; global mu; global log_prec;
 * constants; zeros]), and two backward launches producing every parameter gradient (fixed summation order).
 *   delta_obs [B][C_in][L]   first differences of the observations (encoders.py:385)
 *   inputs [B][n_tr], dev1hot [B][D]
 *   conv_w [F][C_in][K], conv_b [F]; lin_w [H][F*Lp], lin_b [H]           (torch layouts)
 *   local_w [2*nl][NX] (all mu heads, then all log_prec heads; NX = H + n_tr*l_tr + D*l_dv), local_b [2*nl]
 *   gcond_w [2*ng][NG] (NG = n_tr*g_tr + D*g_dv; no bias); global_free [2][ngl]; const_values [nc]
 *   pooled [B][F*Lp], hidden [B][H]: forward by-products the backward needs. */
typedef struct vihds_encoder_shape {
  int B;
  int C_in, L;        /* conv input channels (observed signals) and length (n_times - 1) */
  int F, K, pool;     /* filters, filter size, average-pool width (stride 1) */
  int H;              /* hidden units */
  int n_tr, D;        /* treatments, device depth */
  int nl, l_tr, l_dv; /* local parameters; their heads see [hidden, treatments if l_tr, dev1hot if l_dv] */
  int ng, g_tr, g_dv; /* global-conditioned parameters; heads see [treatments if g_tr, dev1hot if g_dv] */
  int ngl, nc;        /* global parameters, constants */
} vihds_encoder_shape;
int vihds_encoder_fwd(const vihds_encoder_shape* s, const float* delta_obs, const float* inputs, const float* dev1hot,
                      const float* conv_w, const float* conv_b, const float* lin_w, const float* lin_b,
                      const float* local_w, const float* local_b, const float* gcond_w, const float* global_free,
                      const float* const_values, float* q_all, float* pooled, float* hidden, void* stream);
/* g_all [2P][B] -> gradients of every parameter tensor (each overwritten; g_local_b may be NULL with local_b).
 * scratch: g_pre [B][H] and g_conv [B][F][Lc] (Lc = L-K+1) are work buffers of the call. */
int vihds_encoder_bwd(const vihds_encoder_shape* s, const float* g_all, const float* delta_obs, const float* inputs,
                      const float* dev1hot, const float* lin_w, const float* local_w, const float* pooled,
                      const float* hidden, float* g_pre, float* g_conv, float* g_conv_w, float* g_conv_b,
                      float* g_lin_w, float* g_lin_b, float* g_local_w, float* g_local_b, float* g_gcond_w,
                      float* g_global_free, void* stream);

/* Rectangular blocks of the Gram matrix of a field-major buffer X [n_fields][n_columns] (row stride n_columns):
 *   out[dest0 + i*dest_stride_a + j*dest_stride_b] = sum_c X[a0+i][c] * X[b0+j][c],  i < na, j < nb, per rectangle.
 * This is the dr_blackbox weight-gradient contraction over the adjoint kernel's dump (aux of vihds_ode_bwd: fields x
 * (RHS evaluations x trajectories)): every weight gradient of NeuralStates / NeuralPrecisions (reference vihds/ode.py:
 * 134-138, precisions.py:76-87, obtained there by autograd) is the dot product of a pre-activation-adjoint row and a
 * layer-input row, and the pairs form dense rectangles.  The buffer is read once; fixed summation order.
 * scratch: vihds_gram_scratch_floats floats. */
#define VIHDS_GRAM_MAX_RECTS 8
typedef struct vihds_gram_rect {
  int a0, na, b0, nb;
  int dest0, dest_stride_a, dest_stride_b;
} vihds_gram_rect;
long long vihds_gram_scratch_floats(long long n_columns, int n_rects, const vihds_gram_rect* rects);
int vihds_gram_blocks(int n_fields, long long n_columns, int n_rects, const vihds_gram_rect* rects, const float* X,
                      float* scratch, float* out, void* stream);

/* dr_blackbox, matrix-core adjoint (kernel_variant 0, fixed-grid solvers): the Gram-type weight gradients are accumulated
 * ON CHIP -- every wavefront keeps the eight 16x16 output tiles of the four rectangles in registers (32 MFMAs per RHS
 * evaluation over tiles transposed through LDS) and leaves 8 KB of partial sums at the head of aux instead of the
 * [117][E][B*S] dump; vihds_blackbox_gram_reduce adds them in wavefront order and scatters them into g_weights (flat
 * weight layout).  The tail (Delta, bias sums) follows at vihds_blackbox_tail_offset_floats(p) floats into aux and is
 * consumed by vihds_blackbox_tail_grads as before.  (kernel_variant 1, the adaptive solvers and the *_precisions models keep the dump + vihds_gram_blocks pair; the
 * one-wavefront MFMA kernels of kernel_variant 4 were removed in round 6: the value now selects what 0 does.) */
int vihds_blackbox_gram_on_chip(const vihds_ode_problem* p);           /* 1 / 0 */
long long vihds_blackbox_tail_offset_floats(const vihds_ode_problem* p);
int vihds_blackbox_gram_reduce(const vihds_ode_problem* p, const float* aux, float* g_weights, void* stream);

/* dr_blackbox: the weight-gradient entries that are not Gram rectangles, from the tail vihds_ode_bwd leaves behind the
 * dump (tail = aux + F*E*B*S: Delta [HS+HP][B*S], then the output-bias adjoint sums [2*NX+8][B*S]):
 *   g_weights[dest[h*n_const + k]]          = sum_n Delta[h][n] * const_k(n)   h < HS+HP, k < n_const
 *   g_weights[dest[(HS+HP)*n_const + r]]    = sum_n tail[r][n]                 r < HS+HP+2*NX+8
 * const_k(n) is what the integrator fed the hidden layers: theta row slot_row[k] for the n_const-C-D latent inputs,
 * then cond[b][.], then dev1hot[b][.], b = n / S.  In the reference these are the autograd gradients of the
 * time-invariant input columns and of the biases of NeuralStates / NeuralPrecisions (vihds/ode.py:119-146,
 * precisions.py:44-87).  Fixed summation order. */
int vihds_blackbox_tail_grads(const vihds_ode_problem* p, const float* theta, const float* cond, const float* dev1hot,
                              const float* tail, const int* dest, float* g_weights, void* stream);

/* Adam update of the encoder / decoder-network parameters (reference training.py:82,338: torch.optim.Adam, default
 * betas/eps, no weight decay, no amsgrad) as one launch over up to VIHDS_ADAM_MAX_TENSORS parameter tensors:
 *   m += (g - m)(1-beta1);  v = beta2 v + (1-beta2) g^2;  t = step+1
 *   p -= lr / (1-beta1^t) * m / (sqrt(v) / sqrt(1-beta2^t) + eps)
 * m / v are flat buffers holding the tensors back to back in table order.  `state` is two device floats
 * {step count, block ticket}: the kernel reads the count, and the last block to finish increments it, so the whole
 * update is graph-capturable with no host-side step.  lr_dev (optional) overrides lr with a device scalar (learning
 * rate schedules under a captured graph). */
#define VIHDS_ADAM_MAX_TENSORS 32
typedef struct vihds_adam_tensors {
  int n;
  int size[VIHDS_ADAM_MAX_TENSORS];
  float* param[VIHDS_ADAM_MAX_TENSORS];
  const float* grad[VIHDS_ADAM_MAX_TENSORS]; /* NULL: tensor received no gradient this step, skipped */
} vihds_adam_tensors;
int vihds_adam_step(const vihds_adam_tensors* t, float* m, float* v, float* state, const float* lr_dev, float lr,
                    float beta1, float beta2, float eps, float grad_scale, const float* gate, void* stream);
/* grad_scale multiplies every gradient as it is read (1/world after a SUM all-reduce: the average of the replicas'
 * gradients at no extra pass).
 * gate (optional): one device float, the step's loss (-ELBO).  When it is not finite the launch updates NOTHING and
 * does not count as a step -- the reference stops before optimizer.step on a NaN ELBO (training.py:331-334); the host
 * reads the same value to report it.  (Without a gate, and always as a second line of defence, a non-finite gradient
 * ELEMENT leaves its parameter and moments untouched.) */

/* Telemetry of the time-parallel decoder kernel (vihds_theta_ode_logp_grad / vihds_ode_logp_grad on the headline path): its OD
 * chain is solved by Newton's method over the lanes, so its run time depends on the values.  hist: 34 zeroed device words (or
 * NULL to switch it off, the default): hist[w], w = 1..32, counts the wavefronts (two trajectories each) whose iteration took w
 * walks of the chain, hist[33] those that finished through the first-order correction instead of a last walk.  A
 * process-wide setting read at launch time (the one piece of state besides the error string); captured launches keep the
 * pointer they were captured with. */
int vihds_debug_newton_hist(unsigned int* hist);

/* The rest of a training step behind the decoder launch, for a single process whose trainable parameters are the
 * encoder's: IWAE loss (training.py:135-149), the backward through theta / log q / log p to q's tables, the backward
 * through the encoder, and Adam (training.py:334-337) -- what vihds_theta_bwd (with a vihds_iwae_job) +
 * vihds_encoder_bwd + vihds_adam_step do in five launches -- in TWO: one block per data row for everything that
 * needs no other row (importance weights, theta adjoint, the encoder's per-row chain), then every parameter gradient
 * as a fixed-order sum over the rows with the Adam update applied by the thread that formed it, and -ELBO.  Neither
 * launch contains a fence, a ticket or a returning atomic.  A non-finite loss (some row's lse not finite) skips the
 * whole update and the step count.
 *   theta side: as vihds_theta_bwd with opts->q_rows / q_prec_is_log = 1 / opts->iwae (iwae.ticket is not used);
 *     q_all [2P][B] is the encoder's table (means and LOG-precisions), g_all [2P][B] receives its gradient;
 *     g_theta_unit [>= P][B][S] is vihds_theta_ode_logp_grad's unit-weight gradient.
 *   encoder side: as vihds_encoder_bwd (g_pre [B][H], g_conv [B][F][Lc]: work buffers).
 *   tensors, in this order: 0 global_free [2][ngl], 1 conv_w, 2 conv_b, 3 lin_w, 4 lin_b, 5 local_w, 6 local_b,
 *     7 gcond_w: param[k] (updated in place), grad[k] (written: the gradients stay observable), mv_offset[k] = offset
 *     of tensor k in the flat m / v buffers.  An absent tensor (no parameters at that level) has param[k] = NULL.
 *   state: FOUR device floats {step count, (unused), 1/(1-beta1^t), sqrt(1-beta2^t)}: the first launch counts the
 *     step and writes the two bias corrections for the second.  state = NULL: gradients and loss only, no update.
 * VIHDS_E_UNSUPPORTED when the row working set (S importance weights + the encoder's row buffers) exceeds 60 KB of LDS. */
#define VIHDS_TAIL_MAX_EXTRA 4
typedef struct vihds_tail_tensor { /* one decoder-side parameter tensor of vihds_step_tail (see extra[] below) */
  float* param;
  float* grad;
  const float* grad_src;
  const int* map;
  int size, nparts;
  long long part_stride;
  int mv_offset;
} vihds_tail_tensor;
typedef struct vihds_step_tail_args {
  int P, S;
  const int* kind;
  const float* q_all;
  const int* q_rows;
  const float *p_mu, *p_prec, *clip_lo, *clip_hi;
  const float* u;
  const float* g_theta_unit;
  vihds_iwae_job iwae;
  float* g_all;
  const float *delta_obs, *inputs, *dev1hot, *lin_w, *local_w, *pooled, *hidden;
  float *g_pre, *g_conv;
  float* param[8];
  float* grad[8];
  int mv_offset[8];
  float *m, *v, *state;
  const float* lr_dev;
  float lr, beta1, beta2, eps;
  /* ---- ABI 13: any model, not only the ones whose trainable parameters are all in the encoder (all zero / NULL: the
   * form above, fed by vihds_theta_ode_logp_grad's unit-weight gradient) -------------------------------------------
   * g_theta_weighted = 1: g_theta_unit already carries the importance weight d loss / d log_w (it is vihds_ode_bwd's
   *   output for the g_logp that vihds_iwae_loss_fwd / vihds_ode_bwd_elbo formed), so the theta adjoint does not
   *   multiply by it again.
   * g_shift_lo / g_shift_n / g_shift: parameters p in [g_shift_lo, g_shift_lo + g_shift_n) take their gradient from row
   *   p + g_shift of g_theta_unit instead of row p (dr_blackbox: the sampled y receive the gradient of the
   *   device-conditioned rows the integrator read, models/dr_blackbox.py:86-96).
   * extra[k], k < n_extra: decoder-side parameter tensors (NeuralPrecisions / NeuralStates weights, reference
   *   precisions.py:44-87, ode.py:119-146), updated by the same Adam launch: the gradient of element e is the sum over
   *   `nparts` partial rows, grad_src[part * part_stride + (map ? map[e] : e)], parts in ascending order (nparts = 1: an
   *   already reduced flat gradient; nparts > 1: one partial row per block of the lane-split adjoint, vihds_ode_bwd's
   *   aux).  grad[e] receives the sum, param / m / v are updated in place (mv_offset as above).
   * offset layer (dr_blackbox's Linear(D, n_y) of condition_theta): off_n rows starting at g_theta row off_row0;
   *   g_W[i][d] = sum_b dev1hot[b][d] rs[i][b], g_b[i] = sum_b rs[i][b], rs[i][b] = sum_s g_theta[off_row0+i][b][s]
   *   (x the importance weight unless g_theta_weighted); off_rowsum [off_n][B] is a work buffer. */
  int g_theta_weighted;
  int g_shift_lo, g_shift_n, g_shift;
  int n_extra;
  vihds_tail_tensor extra[VIHDS_TAIL_MAX_EXTRA];
  int off_n, off_row0;
  float *off_w, *off_b, *off_gw, *off_gb;
  int off_mv_w, off_mv_b;
  float* off_rowsum;
  /* phase: 0 = both launches (rows, then update); 1 = the rows launch only; 2 = the update launch only -- so that work the
   * update depends on but the rows launch does not (dr_blackbox's weight-gradient reductions) can run beside the rows launch on
   * another stream, and join before the update */
  int phase;
  /* NULL, or the in-kernel generator state whose step the rows launch advances (vihds_theta_ode_fwd left it to its successor) */
  unsigned int* rng_advance;
} vihds_step_tail_args;
int vihds_step_tail(const vihds_encoder_shape* s, const vihds_step_tail_args* a, void* stream);
int vihds_step_tail_supported(const vihds_encoder_shape* s, int P, int S); /* 1 / 0: shapes vihds_step_tail takes (LDS budget,
                                                                             at most 10 filter taps) */

#ifdef __cplusplus
}
#endif
#endif /* VIHDS_HIP_H */
