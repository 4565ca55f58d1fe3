// The rest of a training step behind the decoder launch (reference vihds/training.py:135-149 IWAE loss, autograd's
// backward through theta = clip(sample(q, u)) / log q / log p (distributions.py:64-85,119-142,327-381) and through the
// encoder (encoders.py:16-55,126-253), then torch.optim.Adam (training.py:82,334-337)) as TWO launches:
//
//   step_tail_rows_kernel    one block per data row: the row's importance weights (the arithmetic of
//                            iwae_loss_rows_kernel), the theta adjoint of all P parameters (theta_bwd_kernel's
//                            arithmetic, one wavefront per parameter, u staged in LDS with coalesced loads), and the
//                            encoder's per-row adjoint chain (encoder_bwd_row_kernel's arithmetic) -- everything that
//                            needs no other row.  Replaces theta_bwd_kernel + encoder_bwd_row_kernel.
//   step_tail_update_kernel  every encoder parameter gradient as a fixed-order sum over the rows (the tasks of
//                            encoder_bwd_reduce_kernel) with the Adam update applied to the element by the thread that
//                            summed it (adam_kernel's arithmetic), and -ELBO from the rows' lse.  Replaces
//                            encoder_bwd_reduce_kernel + adam_kernel.
//
// Why two and not one: the second half needs every row's g_pre / g_conv / g_all, i.e. a grid-wide dependency.  On
// gfx950 a kernel boundary is the cheapest grid-wide hand-off there is (1.6 us measured here between the two; an
// in-launch barrier across XCDs costs an agent-scope release + acquire pair, 3-4 us:
// /opt/skills/guides/MI355X_MICROARCH.md, rows `boundary` and `barrier-xcd`).  Neither kernel contains a fence, a
// ticket or a returning atomic, and both are written as ONE round of global loads (everything requested at entry, into
// registers) followed by the dependent arithmetic: these launches are latency, not bandwidth (tests/probe/tail_stamps.py).
//
// Arithmetic notes (all within the parity tolerances, checked against the five-launch path and the reference fixtures):
// the theta adjoint's per-sample exp / log / reciprocal use v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp) -- at one block per
// row its 35 x 200 (parameter, sample) terms are VALU issue on ONE CU, and the IEEE sequences were 5 us of the launch;
// wavefront sums run as DPP scans instead of ds_bpermute trees (different, still fixed, summation order).
//
// A non-finite loss (any row's lse not finite) skips the WHOLE update, step counter included -- the reference stops
// before optimizer.step on a NaN ELBO (training.py:331-334).
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"
#include "vihds_rng.hpp"
#include "vihds_wave.hpp"

namespace vihds {

namespace {
constexpr int ROWS_T = 1024;  // threads of a row block (16 wavefronts)
constexpr int ROWS_NW = ROWS_T / 64;
constexpr int UPD_T = 256;
constexpr int UPD_NW = UPD_T / 64;
constexpr int UPD_KMAX = 10;  // conv taps a block accumulates side by side (the reference's filter_size: config.py:63)
constexpr int UPD_RUN = 11;   // consecutive time points of one row a thread multiplies out of registers
constexpr int UPD_CH = 36;    // data rows whose operands a thread holds in registers at a time

struct TailDims {
  int Lc, Lp, NPOOL, NX, NG;
};
__host__ __device__ inline TailDims tail_dims(const vihds_encoder_shape& s) {
  TailDims d;
  d.Lc = s.L - s.K + 1;
  d.Lp = d.Lc - s.pool + 1;
  d.NPOOL = s.F * d.Lp;
  d.NX = s.H + (s.l_tr ? s.n_tr : 0) + (s.l_dv ? s.D : 0);
  d.NG = (s.g_tr ? s.n_tr : 0) + (s.g_dv ? s.D : 0);
  return d;
}
__host__ __device__ inline int pad4(int n) { return (n + 3) & ~3; }

__device__ __forceinline__ bool finite_f(float x) { return fabsf(x) <= 3.402823466e38f; }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
}  // namespace

#ifdef VIHDS_TAIL_STAMPS
// profiling build (tests/probe/tail_stamps.py): every wavefront writes the 100 MHz wall clock at each phase boundary
static __device__ unsigned long long* vihds_tail_stamp_buf = nullptr;  // [2 kernels][1024 blocks][16 waves][8]
#define VIHDS_TAIL_STOP(KRN, PH)                                                                                  \
  if (vihds_tail_stamp_buf && (threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                           \
    vihds_tail_stamp_buf[((((size_t)(KRN)) * 1024 + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (PH)] = wall_clock64();
#else
#define VIHDS_TAIL_STOP(KRN, PH)
#endif

// A barrier for blocks that exchange data through LDS only: `__syncthreads()` also waits for every outstanding global load
// of the wavefront (vmcnt(0)) -- here that would be the 50-deep lin_w column, which nothing needs before stage 3.
__device__ __forceinline__ void sync_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// Grid (B, NPART).  Part 0 of a row: the nl local parameters (the only ones the encoder's hidden layer sees, through the
// local heads) and the encoder's row chain; part y >= 1: parameters nl + 16 (y - 1) .. -- one wavefront per parameter --
// whose q gradients only have to be written out.  Every part forms the row's importance weights itself (S terms).
// LDS (floats, every region padded to a multiple of 4): u [S*P] (when staged) | w [S] | gall [2P] | gpre [64] | gpl [NPOOL]
// | lw [2nl*H] | red [2*NW] | ctab [12 P]
constexpr int CT = 12;  // ctab row: kind, mu row, log-prec row, mu, prec, sigma, c_ap, c_lq, prior mu, prior prec, lo, hi
template <bool U_LDS, int HMAX>
__global__ void __launch_bounds__(ROWS_T)
step_tail_rows_kernel(vihds_encoder_shape s, vihds_step_tail_args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const TailDims d = tail_dims(s);
  const int b = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int B = s.B, P = a.P, S = a.S, n = B * S;
  const bool chain = part == 0;
  float* u_l = lds;
  float* wsm = u_l + (U_LDS ? pad4(S * P) : 0);
  float* gall = wsm + pad4(S);
  float* gpre = gall + pad4(2 * P);
  float* gpl = gpre + 64;
  float* lw_l = gpl + pad4(d.NPOOL);
  float* red = lw_l + pad4(2 * s.nl * s.H);
  float* ctab = red + 2 * ROWS_NW;
  const vihds_iwae_job& iw = a.iwae;
  // this block's tasks: [p_lo, p_hi), wavefront w takes p_lo + w, p_lo + w + 16, ...  Tasks 0..P-1 are the parameters;
  // tasks P..P+off_n-1 (ABI 13: dr_blackbox's offset layer) are plain row sums over the samples of g_theta rows off_row0..
  const int P_tasks = P + a.off_n;
  const int p_lo = chain ? 0 : s.nl + ROWS_NW * (part - 1);
  const int p_hi = chain ? s.nl : min(P_tasks, p_lo + ROWS_NW);
  // the row of g_theta_unit a task reads (ABI 13: a shifted range of parameters; the offset layer's rows)
  auto g_row = [&](int p) {
    if (p >= P) return a.off_row0 + (p - P);
    return p + ((p >= a.g_shift_lo && p < a.g_shift_lo + a.g_shift_n) ? a.g_shift : 0);
  };
  const float unit_g = a.g_theta_weighted ? 0.f : 1.f;  // 1: g_theta_unit is the unit-weight gradient (x the weight here)
  VIHDS_TAIL_STOP(0, 0)

  // ---- every global read of the block is requested here, into registers: one memory round trip for the whole kernel --
  // the Adam step of this launch pair: counted by one thread (which has little else to request), its bias corrections
  // handed to the update kernel
  float adam_t = 0.f;
  const bool adam_thread = b == 0 && part == 0 && tid == ROWS_T - 1 && a.state;
  if (adam_thread) adam_t = a.state[0];
  // (d) the row's draws (to LDS below: the adjoint reads u[s][p] with stride P, which LDS serves conflict-free for odd P
  //     and which costs a cache line per lane from global memory); two 16-byte loads per thread cover S*P <= 8192
  const float* ur = a.u + (size_t)b * S * P;
  const int ucnt = S * P;
  const bool u_vec = U_LDS && (ucnt & 3) == 0 && ucnt <= 8 * ROWS_T;
  float4 ureg0 = make_float4(0.f, 0.f, 0.f, 0.f), ureg1 = ureg0;
  if (u_vec) {  // (uniform; the two loads themselves unconditional, past-the-end threads re-read the last 16 bytes)
    const int last4 = ucnt / 4 - 1;
    ureg0 = reinterpret_cast<const float4*>(ur)[min(tid, last4)];
    ureg1 = reinterpret_cast<const float4*>(ur)[min(tid + ROWS_T, last4)];
  }
  // (e) local-head weights of the hidden units (to LDS below) and the row's hidden activations (stage 3)
  const int lw_cnt = chain ? 2 * s.nl * s.H : 0;
  float lwreg = 0.f, hid = 0.f;
  if (chain) {  // (uniform per block)
    const int tq = min(tid, max(lw_cnt - 1, 0));
    const int r = tq / s.H, j = tq - r * s.H;
    if (s.nl > 0) lwreg = a.local_w[(size_t)r * d.NX + j];
    hid = a.hidden[(size_t)b * s.H + min(tid, s.H - 1)];
  }
  // (b) the importance-weight inputs of sample s = tid (further samples in the loop below)
  //     (all unconditional, optional arrays replaced by a valid one and the value selected afterwards: a load under a branch
  //     is followed by a wait for everything at the join)
  float lw0;
  {
    const int i = b * S + min(tid, S - 1);
    const float* lpp = iw.log_p ? iw.log_p : iw.logp;
    const float* lqp = iw.log_q ? iw.log_q : iw.logp;
    const float v0 = iw.logp[i], v1 = iw.logp[n + i], v2 = iw.logp[2 * n + i], v3 = iw.logp[3 * n + i];
    const float vp = lpp[i], vq = lqp[i];
    const float v = ((v0 + v1) + v2) + v3;
    lw0 = v + (iw.log_p ? vp : 0.f) - (iw.log_q ? vq : 0.f);
    lw0 = tid < S ? lw0 : -INFINITY;
  }
  // (c) the block's parameters' constants (thread t fetches parameter p_lo + t; derived values to LDS below, so that a
  //     wavefront's per-parameter prologue is a dozen LDS reads), and the first rounds of the wavefront's unit-weight gradient
  constexpr int RC = 4;  // sample rounds held in registers
  const int cp = p_lo + tid;
  const int cpc = min(cp, P - 1);
  const int c_kd = a.kind[cpc], c_rm = a.q_rows[cpc], c_rp = a.q_rows[P + cpc];
  const float c_pm = a.p_mu[cpc], c_pp = a.p_prec[cpc], c_lo = a.clip_lo[cpc], c_hi = a.clip_hi[cpc];
  const float c_mu = a.q_all[(size_t)c_rm * B + b], c_pr = a.q_all[(size_t)c_rp * B + b];
  const int p_w = p_lo + wid;
  float gx[RC];
#pragma unroll
  for (int c = 0; c < RC; ++c)
    gx[c] = a.g_theta_unit[(size_t)g_row(min(p_w, P_tasks - 1)) * n + b * S + min(lane + 64 * c, S - 1)];
  VIHDS_TAIL_STOP(0, 1)
  // ---- registers -> LDS for what other threads read
  if (U_LDS) {
    if (u_vec) {
      if (4 * tid < ucnt) reinterpret_cast<float4*>(u_l)[tid] = ureg0;
      if (4 * (tid + ROWS_T) < ucnt) reinterpret_cast<float4*>(u_l)[tid + ROWS_T] = ureg1;
    } else {
      for (int q = tid; q < ucnt; q += ROWS_T) u_l[q] = ur[q];
    }
  }
  if (tid < lw_cnt) lw_l[tid] = lwreg;
  for (int q = tid + ROWS_T; q < lw_cnt; q += ROWS_T) {
    const int r = q / s.H, j = q - r * s.H;
    lw_l[q] = a.local_w[(size_t)r * d.NX + j];
  }
  auto fill_ctab = [&](int p, int kdp, int rmp, int rpp, float mup, float prp, float pmp, float ppp, float lop, float hip_) {
    float* c = ctab + CT * (p - p_lo);
    const float prec = expf(prp);  // the table holds log-precisions (encoders.py:150-165)
    const float sigma = 1.f / sqrtf(prec);
    c[0] = __int_as_float(kdp); c[1] = __int_as_float(rmp); c[2] = __int_as_float(rpp);
    c[3] = mup; c[4] = prec; c[5] = sigma;
    c[6] = -0.5f * sigma / prec;       // d z / d prec = -0.5 u prec^-3/2
    c[7] = 0.5f / (prec + 1e-12f);     // d/d prec of 0.5 log(prec + 1e-12)
    c[8] = pmp; c[9] = ppp; c[10] = lop; c[11] = hip_;
    if (kdp == KIND_CONSTANT) {
      // constants carry no trainable distribution parameters in the reference (encoders.py:242-253)
      a.g_all[(size_t)rmp * B + b] = 0.f; a.g_all[(size_t)rpp * B + b] = 0.f;
      gall[rmp] = 0.f; gall[rpp] = 0.f;
    }
  };
  if (cp < min(p_hi, P)) fill_ctab(cp, c_kd, c_rm, c_rp, c_mu, c_pr, c_pm, c_pp, c_lo, c_hi);
  for (int p = cp + ROWS_T; p < min(p_hi, P); p += ROWS_T) {  // (more than 1024 local parameters: not a case that exists; kept correct)
    const int rmp = a.q_rows[p], rpp = a.q_rows[P + p];
    fill_ctab(p, a.kind[p], rmp, rpp, a.q_all[(size_t)rmp * B + b], a.q_all[(size_t)rpp * B + b], a.p_mu[p], a.p_prec[p],
              a.clip_lo[p], a.clip_hi[p]);
  }
  if (a.rng_advance && b == 0 && part == 0 && tid == ROWS_T - 2) a.rng_advance[2] = a.rng_advance[2] + 1u;
  if (adam_thread) {
    const float t = adam_t + 1.f;
    a.state[0] = t;
    a.state[2] = 1.f / (1.f - powf(a.beta1, t));
    a.state[3] = sqrtf(1.f - powf(a.beta2, t));
  }

  // (a) the lin_w column of this thread (pooled adjoint, stage 3), requested HERE -- behind the waits of the small loads
  //     above, ahead of two barriers that do not wait for it: it lands during stages 1 and 2.  HMAX: the column lives in
  //     registers: 52 covers the reference's 50 hidden units
  const bool fast = s.H <= HMAX && d.NPOOL <= ROWS_T;
  float wcol[HMAX];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (nothing older may be left for a later wait to trip over the column)
  if (chain && fast && tid < d.NPOOL) {
#pragma unroll
    for (int j = 0; j < HMAX; ++j)  // (rows past H re-read row 0: they meet the zero fill of gpre; no branch per load)
      wcol[j] = a.lin_w[(size_t)(j < s.H ? j : 0) * d.NPOOL + tid];
  }

  // ---- stage 1: importance weights of the row (training.py:135-149): per-wavefront (max, sum-exp) pairs, ONE barrier, every
  //      thread combines the pairs in wavefront order
  float mw = lw0;
  if (tid < S) wsm[tid] = lw0;
  for (int sidx = tid + ROWS_T; sidx < S; sidx += ROWS_T) {
    const int i = b * S + sidx;
    float v = ((iw.logp[i] + iw.logp[n + i]) + iw.logp[2 * n + i]) + iw.logp[3 * n + i];
    v = v + (iw.log_p ? iw.log_p[i] : 0.f) - (iw.log_q ? iw.log_q[i] : 0.f);
    wsm[sidx] = v;
    mw = fmaxf(mw, v);
  }
  mw = wave_max_total(mw);
  float sew = 0.f;
  if (mw > -INFINITY)
    for (int sidx = tid; sidx < S; sidx += ROWS_T) sew += expf(wsm[sidx] - mw);
  sew = wave_total(sew);
  if (lane == 0) { red[wid] = mw; red[ROWS_NW + wid] = sew; }
  sync_lds();
  VIHDS_TAIL_STOP(0, 2)
  float m = red[0];
#pragma unroll
  for (int w = 1; w < ROWS_NW; ++w) m = fmaxf(m, red[w]);
  float se = 0.f;
#pragma unroll
  for (int w = 0; w < ROWS_NW; ++w) {
    const float mwv = red[w];
    se += (mwv > -INFINITY) ? red[ROWS_NW + w] * expf(mwv - m) : 0.f;
  }
  const float l = m + logf(se);
  for (int sidx = tid; sidx < S; sidx += ROWS_T) {
    const float v = wsm[sidx];
    if (chain) iw.log_w[b * S + sidx] = v;
    wsm[sidx] = -(1.f / (float)B) * expf(v - l);  // d loss / d log_w
  }
  if (chain && tid == 0) iw.lse[b] = l;
  sync_lds();
  VIHDS_TAIL_STOP(0, 3)

  // ---- stage 2: theta adjoint, one wavefront per parameter (theta_bwd_kernel's arithmetic) ---------------------------
  const float has_q = iw.log_q ? 1.f : 0.f, has_p = iw.log_p ? 1.f : 0.f;
  // (the first parameter of a wavefront reads its unit-weight gradient from the registers filled at entry; only further
  // parameters / samples beyond 64 * RC load inside the loop -- kept apart so that no wait for a load sits on the main
  // path, where it would also wait for the lin_w column)
  auto run_param = [&](int p, auto first_rounds_g) {
    const float* c = ctab + CT * (p - p_lo);
    const int kdk = __float_as_int(c[0]), rmk = __float_as_int(c[1]), rpk = __float_as_int(c[2]);
    if (kdk == KIND_CONSTANT) return;  // (zeroed above)
    const float muk = c[3], prec = c[4], sigma = c[5], c_ap = c[6], c_lq = c[7], pmk = c[8], ppk = c[9], lok = c[10],
                hik = c[11];
    const bool ln = kdk == KIND_LOGNORMAL;
    const float jac = ln ? 1.f : 0.f;
    float am = 0.f, ap = 0.f;
    auto body = [&](int sidx, float uu, float g) {
      const float z = muk + sigma * uu;
      const float xr = ln ? __expf(z) : z;
      const float x = xr < lok ? lok : (xr > hik ? hik : xr);
      const float pass = (xr >= lok && xr <= hik) ? 1.f : 0.f;
      const float gw = wsm[sidx];
      const float glq = -gw * has_q, glp = gw * has_p;
      const float xe = x + 1e-12f;  // (>= 1e-12: v_log_f32 needs no denormal handling)
      const float v = ln ? __builtin_amdgcn_logf(xe) * 0.6931471805599453f : x;
      const float dv_dx = ln ? fast_rcp(xe) : 1.f;
      // d lq/dv = prec*(mu - v) - jac ; d lp/dv = pp*(pm - v) - jac
      const float dd = muk - v;
      const float gv = glq * (prec * dd - jac) + glp * (ppk * (pmk - v) - jac);
      float gz = (g * (unit_g != 0.f ? gw : 1.f) + gv * dv_dx) * pass;  // back through clip ...
      if (ln) gz *= xr;                         // ... and exp
      am += gz;                                 // z = mu + u / sqrt(prec)
      ap += gz * uu * c_ap;
      am += glq * (-prec * dd);                 // explicit dependence of log q on (mu, prec)
      ap += glq * (c_lq - 0.5f * dd * dd);
    };
    const float* ug = a.u + (size_t)b * S * P;
#pragma unroll
    for (int cc = 0; cc < RC; ++cc) {
      const int sidx = lane + 64 * cc;
      if (sidx < S) body(sidx, U_LDS ? u_l[sidx * P + p] : ug[(size_t)sidx * P + p], first_rounds_g(cc, sidx));
    }
    for (int sidx = lane + 64 * RC; sidx < S; sidx += 64)
      body(sidx, U_LDS ? u_l[sidx * P + p] : ug[(size_t)sidx * P + p], a.g_theta_unit[(size_t)g_row(p) * n + b * S + sidx]);
    am = wave_total(am);
    ap = wave_total(ap);
    if (lane == 0) {
      const float gp = ap * prec;  // d / d log_prec
      a.g_all[(size_t)rmk * B + b] = am; a.g_all[(size_t)rpk * B + b] = gp;
      gall[rmk] = am; gall[rpk] = gp;
    }
  };
  // (ABI 13) a row-sum task: rs[b] = sum_s g_theta[row][b][s] (x the importance weight when the gradient is unit-weight)
  auto run_rowsum = [&](int p, auto first_rounds_g) {
    float acc = 0.f;
#pragma unroll
    for (int cc = 0; cc < RC; ++cc) {
      const int sidx = lane + 64 * cc;
      if (sidx < S) acc += first_rounds_g(cc, sidx) * (unit_g != 0.f ? wsm[sidx] : 1.f);
    }
    for (int sidx = lane + 64 * RC; sidx < S; sidx += 64)
      acc += a.g_theta_unit[(size_t)g_row(p) * n + b * S + sidx] * (unit_g != 0.f ? wsm[sidx] : 1.f);
    acc = wave_total(acc);
    if (lane == 0) a.off_rowsum[(size_t)(p - P) * B + b] = acc;
  };
  if (p_w < p_hi) {
    if (p_w < P) run_param(p_w, [&](int cc, int) { return gx[cc]; });
    else run_rowsum(p_w, [&](int cc, int) { return gx[cc]; });
  }
  for (int p = p_w + ROWS_NW; p < p_hi; p += ROWS_NW) {
    auto ld = [&](int, int sidx) { return a.g_theta_unit[(size_t)g_row(p) * n + b * S + sidx]; };
    if (p < P) run_param(p, ld);
    else run_rowsum(p, ld);
  }
  VIHDS_TAIL_STOP(0, 4)
  if (!chain) return;
  sync_lds();
  VIHDS_TAIL_STOP(0, 5)

  // ---- stage 3: the encoder's per-row chain (encoder_bwd_row_kernel's arithmetic) --------------------------------------
  // hidden adjoint through the local heads, then tanh' (zero-filled up to 64 entries for the 16-byte reads below)
  if (tid < 64) {
    float g = 0.f;
    if (tid < s.H) {
      float acc = 0.f;
      for (int r = 0; r < 2 * s.nl; ++r) acc += lw_l[r * s.H + tid] * gall[r];
      g = acc * (1.f - hid * hid);
      a.g_pre[(size_t)b * s.H + tid] = g;
    }
    gpre[tid] = g;
  }
  for (int j = 64 + tid; j < s.H; j += ROWS_T) {  // (more than 64 hidden units: the general path)
    float acc = 0.f;
    for (int r = 0; r < 2 * s.nl; ++r) acc += lw_l[r * s.H + j] * gall[r];
    const float h = a.hidden[(size_t)b * s.H + j];
    a.g_pre[(size_t)b * s.H + j] = acc * (1.f - h * h);
  }
  if (fast) sync_lds(); else __syncthreads();  // (the general path re-reads g_pre from global memory)
  // pooled adjoint: g_pooled[k] = sum_j lin_w[j][k] g_pre[j]
  if (fast) {
    if (chain && tid < d.NPOOL) {
      float a0 = 0.f, a1 = 0.f;
      const float4* g4 = reinterpret_cast<const float4*>(gpre);
#pragma unroll
      for (int j = 0; j < HMAX; j += 4) {
        const float4 gv = g4[j >> 2];
        a0 += wcol[j] * gv.x;
        a1 += wcol[j + 1] * gv.y;
        a0 += wcol[j + 2] * gv.z;
        a1 += wcol[j + 3] * gv.w;
      }
      gpl[tid] = a0 + a1;
    }
  } else {
    for (int k = tid; k < d.NPOOL; k += ROWS_T) {
      float a0 = 0.f, a1 = 0.f;
      int j = 0;
      for (; j + 1 < s.H; j += 2) {
        a0 += a.lin_w[(size_t)j * d.NPOOL + k] * a.g_pre[(size_t)b * s.H + j];
        a1 += a.lin_w[(size_t)(j + 1) * d.NPOOL + k] * a.g_pre[(size_t)b * s.H + j + 1];
      }
      if (j < s.H) a0 += a.lin_w[(size_t)j * d.NPOOL + k] * a.g_pre[(size_t)b * s.H + j];
      gpl[k] = a0 + a1;
    }
  }
  sync_lds();
  // conv-output adjoint: every pooled window that contains t contributes 1/pool
  const float inv_pool = 1.f / (float)s.pool;
  for (int q = tid; q < s.F * d.Lc; q += ROWS_T) {
    const int o = q / d.Lc, t = q - o * d.Lc;
    const int w0 = max(0, t - s.pool + 1), w1 = min(t, d.Lp - 1);
    float acc = 0.f;
    for (int tp = w0; tp <= w1; ++tp) acc += gpl[o * d.Lp + tp];
    a.g_conv[(size_t)b * s.F * d.Lc + q] = acc * inv_pool;
  }
  VIHDS_TAIL_STOP(0, 6)
}

// ---------------------------------------------------------------------------------------------------------------
// Parameter gradients (fixed-order sums over the data rows) with Adam applied in place.  Tensor order of param / grad /
// mv_offset: 0 global_free, 1 conv_w, 2 conv_b, 3 lin_w, 4 lin_b, 5 local_w, 6 local_b, 7 gcond_w.  Block ranges:
//   lin_w  [H][F*Lp]     one thread per element; g_pre -> LDS, the thread's B pooled values in registers
//   conv   (o, c)        one block per (filter, input channel): g_conv[:, o, :] and delta_obs[:, c, :] -> LDS, the K taps
//                        accumulated side by side (round 2: one block per (o, c, k) re-reading both 10 times, its 22 loads
//                        per thread in 11 dependent rounds: 5 us, the launch's long pole); c == 0 also sums conv_b[o]
//   local_w, gcond_w     one thread per element, operands in registers
//   last block           lin_b, local_b, global_free sums; -ELBO
// Every staging loop requests ALL of its loads before the first LDS store (register arrays, compile-time trip counts): a
// plain `lds[q] = global[q]` loop is compiled into one load -> wait -> store round trip per iteration, and a dozen of those
// in a row is the whole launch.
struct TailTasks {
  int nb_lin, nb_conv, nb_localw, nb_gcondw;
  int nb_off;                          // ABI 13: the offset layer's block (0 / 1)
  int nb_extra[VIHDS_TAIL_MAX_EXTRA];  // ABI 13: blocks of each decoder-side tensor
};
struct AdamScalars {
  float step_size, bc2_sqrt, one_m_b1, beta2, one_m_b2, eps;
  bool apply;
};
__device__ __forceinline__ void adam_elem(const vihds_step_tail_args& a, const AdamScalars& k, int tensor, int e,
                                          float ge, float pe, float me, float ve) {
  a.grad[tensor][e] = ge;
  // (a non-finite element is left alone, as in adam_kernel; a non-finite LOSS has switched the whole update off)
  if (!k.apply || !finite_f(ge)) return;
  me += (ge - me) * k.one_m_b1;
  ve = ve * k.beta2 + k.one_m_b2 * ge * ge;
  a.m[a.mv_offset[tensor] + e] = me;
  a.v[a.mv_offset[tensor] + e] = ve;
  a.param[tensor][e] = pe - k.step_size * (me / (sqrtf(ve) / k.bc2_sqrt + k.eps));
}
// B-term fixed-order dot product  sum_b x[b * sx] * y[b * sy]  (y == nullptr: plain sum) with the operands of up to UPD_CH
// rows requested together
__device__ __forceinline__ float dot_rows(const float* x, int sx, const float* y, int sy, int B) {
  // (every load is unconditional -- rows past B re-read row B - 1 and are left out of the sum: a load under a branch gets a
  // conservative wait at the join and the batch turns into one round trip per row)
  float acc = 0.f;
  for (int b0 = 0; b0 < B; b0 += UPD_CH) {
    float xv[UPD_CH], yv[UPD_CH];
#pragma unroll
    for (int i = 0; i < UPD_CH; ++i) {
      const int bb = min(b0 + i, B - 1);
      xv[i] = x[(size_t)bb * sx];
      yv[i] = y ? y[(size_t)bb * sy] : 1.f;
    }
#pragma unroll
    for (int i = 0; i < UPD_CH; ++i)
      if (b0 + i < B) acc += xv[i] * yv[i];
  }
  return acc;
}
// dst[q] = *src(q) for q < cnt, the block's threads striding over q: NIT loads per thread in flight, then the stores
template <int NIT, class Src>
__device__ __forceinline__ void stage_lds(float* dst, int cnt, Src src) {
  for (int q0 = 0; q0 < cnt; q0 += NIT * UPD_T) {
    float r[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = q0 + threadIdx.x + i * UPD_T;
      r[i] = src(min(q, cnt - 1));  // (unconditional: see dot_rows)
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int q = q0 + threadIdx.x + i * UPD_T;
      if (q < cnt) dst[q] = r[i];
    }
  }
}

__global__ void __launch_bounds__(UPD_T)
step_tail_update_kernel(vihds_encoder_shape s, TailTasks tk, vihds_step_tail_args a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ float part[UPD_NW][UPD_KMAX + 1];
  __shared__ int bad_s;
  const TailDims d = tail_dims(s);
  const int tid = threadIdx.x, B = s.B;
  const int lane = tid & 63, wid = tid >> 6;
  VIHDS_TAIL_STOP(1, 0)
  // the loss gate and the Adam scalars: requested now, looked at after the block's own loads are out
  const float lse_v = a.iwae.lse[min(tid, B - 1)];
  // (optional pointers are replaced by a valid one and the VALUE selected afterwards: a load under a branch is followed by
  // a wait for everything at the join -- at kernel entry that is a whole memory round trip before the block's real loads)
  const float* sp = a.state ? a.state : a.iwae.lse;
  const float st2_raw = sp[a.state ? 2 : 0], st3_raw = sp[a.state ? 3 : 0];
  const float lr_raw = (a.lr_dev ? a.lr_dev : sp)[0];
  const float st2 = a.state ? st2_raw : 0.f, st3 = a.state ? st3_raw : 1.f;
  const float lr = a.lr_dev ? lr_raw : a.lr;
  auto scalars = [&]() {  // (two barriers; every thread ends up with the same gate)
    int bad = tid < B && !finite_f(lse_v);
    for (int r = tid + UPD_T; r < B; r += UPD_T) bad |= !finite_f(a.iwae.lse[r]);
    if (tid == 0) bad_s = 0;
    __syncthreads();
    if (bad) bad_s = 1;
    __syncthreads();
    AdamScalars k;
    k.bc2_sqrt = st3;
    k.step_size = lr * st2;
    k.one_m_b1 = 1.f - a.beta1;
    k.beta2 = a.beta2;
    k.one_m_b2 = 1.f - a.beta2;
    k.eps = a.eps;
    k.apply = a.state != nullptr && !bad_s;
    return k;
  };
  int blk = blockIdx.x;
  if (blk < tk.nb_lin) {
    // ---- lin_w: g_lin_w[j][k] = sum_b g_pre[b][j] pooled[b][k]
    const int e = blk * UPD_T + tid;
    const bool live = e < s.H * d.NPOOL;
    const int j = live ? e / d.NPOOL : 0, kk = live ? e - j * d.NPOOL : 0;
    const int ec = live ? e : 0;
    const float pe = a.param[3][ec], me = a.m[a.mv_offset[3] + ec], ve = a.v[a.mv_offset[3] + ec];
    float* gp = lds;  // [B][H]
    float acc = 0.f;
    for (int b0 = 0; b0 < B; b0 += UPD_CH) {
      float pv[UPD_CH];
#pragma unroll
      for (int i = 0; i < UPD_CH; ++i) pv[i] = a.pooled[(size_t)min(b0 + i, B - 1) * d.NPOOL + kk];
      if (b0 == 0) {
        stage_lds<8>(gp, B * s.H, [&](int q) { return a.g_pre[q]; });
        __syncthreads();
      }
#pragma unroll
      for (int i = 0; i < UPD_CH; ++i)
        if (b0 + i < B) acc += gp[(b0 + i) * s.H + j] * pv[i];
    }
    const AdamScalars k = scalars();
    VIHDS_TAIL_STOP(1, 2)
    if (live) adam_elem(a, k, 3, e, acc, pe, me, ve);
    VIHDS_TAIL_STOP(1, 3)
    return;
  }
  blk -= tk.nb_lin;
  if (blk < tk.nb_conv) {
    // ---- conv_w[o][c][:] (and conv_b[o] when c == 0): sums over (row, t) of g_conv[b][o][t] delta_obs[b][c][t + k]
    const int o = blk / s.C_in, c = blk - o * s.C_in;
    float* Gl = lds;                    // [B][Lc]
    float* Xl = lds + pad4(B * d.Lc);   // [B][L]
    const bool tap = tid < s.K, bias = tid == UPD_KMAX && c == 0;
    const int e_w = (o * s.C_in + c) * s.K + (tap ? tid : 0);
    // (one unconditional load each: the bias thread points at the bias tensor, everybody else at a tap)
    const float* pp_ = bias ? a.param[2] + o : a.param[1] + e_w;
    const float* pm_ = bias ? a.m + a.mv_offset[2] + o : a.m + a.mv_offset[1] + e_w;
    const float* pv_ = bias ? a.v + a.mv_offset[2] + o : a.v + a.mv_offset[1] + e_w;
    const float pe = *pp_, me = *pm_, ve = *pv_;
    // (row, offset) of a flat index without an integer division: exact for the index ranges here (< 2^22)
    const float inv_lc = 1.f / (float)d.Lc, inv_l = 1.f / (float)s.L;
    {  // both operands in ONE round of loads (two stage_lds calls would be two round trips)
      constexpr int NIT = 12;
      const int cg = B * d.Lc, cx = B * s.L;
      for (int q0 = 0; q0 < max(cg, cx); q0 += NIT * UPD_T) {
        float rg[NIT], rx[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = min(q0 + tid + i * UPD_T, cg - 1);
          const int bb = (int)(((float)q + 0.5f) * inv_lc);
          rg[i] = a.g_conv[((size_t)bb * s.F + o) * d.Lc + (q - bb * d.Lc)];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = min(q0 + tid + i * UPD_T, cx - 1);
          const int bb = (int)(((float)q + 0.5f) * inv_l);
          rx[i] = a.delta_obs[((size_t)bb * s.C_in + c) * s.L + (q - bb * s.L)];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
          const int q = q0 + tid + i * UPD_T;
          if (q < cg) Gl[q] = rg[i];
          if (q < cx) Xl[q] = rx[i];
        }
      }
    }
    __syncthreads();
    // A thread owns runs of consecutive t inside one row: its g values and the x window they meet sit in registers and the
    // K x RUN products are straight-line FMAs (one (row, t) element per thread and iteration cost an LDS read per product).
    // rpr runs per row, each `run` <= UPD_RUN long; items = (row, run) pairs, thread-strided.
    float acc[UPD_KMAX], accb = 0.f;
#pragma unroll
    for (int kq = 0; kq < UPD_KMAX; ++kq) acc[kq] = 0.f;
    const int rpr = max(max(1, UPD_T / B), (d.Lc + UPD_RUN - 1) / UPD_RUN);
    const int run = (d.Lc + rpr - 1) / rpr;
    const float inv_rpr = 1.f / (float)rpr;
    for (int item = tid; item < B * rpr; item += UPD_T) {
      const int bb = (int)(((float)item + 0.5f) * inv_rpr), t0 = (item - bb * rpr) * run;
      const float* gr = Gl + bb * d.Lc;
      const float* xr = Xl + bb * s.L;
      float gv[UPD_RUN], xv[UPD_RUN + UPD_KMAX - 1];
#pragma unroll
      for (int i = 0; i < UPD_RUN; ++i) {
        const int t = t0 + i;
        gv[i] = (i < run && t < d.Lc) ? gr[min(t, d.Lc - 1)] : 0.f;  // (zero past the run: no predicate on the products)
        accb += gv[i];
      }
#pragma unroll
      for (int i = 0; i < UPD_RUN + UPD_KMAX - 1; ++i) xv[i] = xr[min(t0 + i, s.L - 1)];
#pragma unroll
      for (int kq = 0; kq < UPD_KMAX; ++kq)
#pragma unroll
        for (int i = 0; i < UPD_RUN; ++i) acc[kq] += gv[i] * xv[i + kq];  // (taps past K: computed, never used)
    }
    // block sums: partials through LDS as [tap][thread]; wavefront w adds taps w, w + 4, ... (4 values per lane, lanes, done)
    __syncthreads();  // (Gl / Xl are dead: the partials reuse their space)
    float* pl = lds;
#pragma unroll
    for (int kq = 0; kq < UPD_KMAX; ++kq)
      if (kq < s.K) pl[kq * UPD_T + tid] = acc[kq];
    pl[UPD_KMAX * UPD_T + tid] = accb;
    __syncthreads();
    for (int kq = wid; kq <= UPD_KMAX; kq += UPD_NW) {
      if (kq < s.K || kq == UPD_KMAX) {
        const float* row = pl + kq * UPD_T;
        const float t = wave_total(((row[lane] + row[lane + 64]) + row[lane + 128]) + row[lane + 192]);
        if (lane == 0) part[0][kq] = t;
      }
    }
    const AdamScalars k = scalars();  // (its barriers also publish `part`)
    VIHDS_TAIL_STOP(1, 2)
    if (tap) adam_elem(a, k, 1, e_w, part[0][tid], pe, me, ve);
    if (bias) adam_elem(a, k, 2, o, part[0][UPD_KMAX], pe, me, ve);
    VIHDS_TAIL_STOP(1, 3)
    return;
  }
  blk -= tk.nb_conv;
  if (blk < tk.nb_localw) {
    const int e = blk * UPD_T + tid;
    const bool live = e < 2 * s.nl * d.NX;
    const int r = live ? e / d.NX : 0;
    int i = live ? e - r * d.NX : 0;
    const int ec = live ? e : 0;
    const float pe = a.param[5][ec], me = a.m[a.mv_offset[5] + ec], ve = a.v[a.mv_offset[5] + ec];
    // which per-row input this column multiplies: chosen once, so the B-term loop is branch-free and its loads batch
    const float* src;
    int stride;
    if (i < s.H) { src = a.hidden + i; stride = s.H; }
    else {
      i -= s.H;
      if (s.l_tr && i < s.n_tr) { src = a.inputs + i; stride = s.n_tr; }
      else { src = a.dev1hot + (i - (s.l_tr ? s.n_tr : 0)); stride = s.D; }
    }
    const float acc = dot_rows(a.g_all + (size_t)r * B, 1, src, stride, B);
    const AdamScalars k = scalars();
    if (live) adam_elem(a, k, 5, e, acc, pe, me, ve);
    return;
  }
  blk -= tk.nb_localw;
  if (blk < tk.nb_gcondw) {
    const int e = blk * UPD_T + tid;
    const bool live = e < 2 * s.ng * d.NG;
    const int r = live ? e / d.NG : 0, i = live ? e - r * d.NG : 0;
    const int ec = live ? e : 0;
    const float pe = a.param[7][ec], me = a.m[a.mv_offset[7] + ec], ve = a.v[a.mv_offset[7] + ec];
    const float* src;
    int stride;
    if (s.g_tr && i < s.n_tr) { src = a.inputs + i; stride = s.n_tr; }
    else { src = a.dev1hot + (i - (s.g_tr ? s.n_tr : 0)); stride = s.D; }
    const float acc = dot_rows(a.g_all + (size_t)(2 * s.nl + r) * B, 1, src, stride, B);
    const AdamScalars k = scalars();
    if (live) adam_elem(a, k, 7, e, acc, pe, me, ve);
    return;
  }
  blk -= tk.nb_gcondw;
  if (blk < tk.nb_off) {
    // ---- (ABI 13) dr_blackbox's offset layer: weight [off_n][D] = rowsum x dev1hot over the rows, bias [off_n] = row sums
    const int nW = a.off_n * s.D, e = tid;
    const bool isw = e < nW, live = e < nW + a.off_n;
    const int ec = live ? e : 0;
    const int i = isw ? ec / s.D : (live ? ec - nW : 0), dd = isw ? ec - i * s.D : 0;
    float* pp_ = isw ? a.off_w + ec : a.off_b + i;
    const int mvo = isw ? a.off_mv_w + ec : a.off_mv_b + i;
    const float pe = *pp_, me = a.m[mvo], ve = a.v[mvo];
    const float acc = dot_rows(a.off_rowsum + (size_t)i * B, 1, isw ? a.dev1hot + dd : nullptr, s.D, B);
    const AdamScalars k = scalars();
    if (live) {
      (isw ? a.off_gw + ec : a.off_gb + i)[0] = acc;
      if (k.apply && finite_f(acc)) {  // (adam_elem's arithmetic)
        const float m1 = me + (acc - me) * k.one_m_b1, v1 = ve * k.beta2 + k.one_m_b2 * acc * acc;
        a.m[mvo] = m1; a.v[mvo] = v1;
        *pp_ = pe - k.step_size * (m1 / (sqrtf(v1) / k.bc2_sqrt + k.eps));
      }
    }
    return;
  }
  blk -= tk.nb_off;
#pragma unroll
  for (int x = 0; x < VIHDS_TAIL_MAX_EXTRA; ++x) {
    if (blk < tk.nb_extra[x]) {
      // ---- (ABI 13) decoder-side tensor x: gradient = fixed-order sum over its partial rows, Adam on the element
      const vihds_tail_tensor& t = a.extra[x];
      if (t.nparts == 1) {  // one thread per element
        const int e = blk * UPD_T + tid;
        const bool live = e < t.size;
        const int ec = live ? e : 0;
        const int src = t.map ? t.map[ec] : ec;
        const float pe = t.param[ec], me = a.m[t.mv_offset + ec], ve = a.v[t.mv_offset + ec];
        const float ge = t.grad_src[src];
        const AdamScalars k = scalars();
        if (live) {
          t.grad[e] = ge;
          if (k.apply && finite_f(ge)) {
            const float m1 = me + (ge - me) * k.one_m_b1, v1 = ve * k.beta2 + k.one_m_b2 * ge * ge;
            a.m[t.mv_offset + e] = m1; a.v[t.mv_offset + e] = v1;
            t.param[e] = pe - k.step_size * (m1 / (sqrtf(v1) / k.bc2_sqrt + k.eps));
          }
        }
      } else {  // one wavefront per element: lane q adds parts q, q + 64, ... (8 requested together), then the lanes in order
        const int e = blk * UPD_NW + wid;
        const bool live = e < t.size;
        const int ec = live ? e : 0;
        const int src = t.map ? t.map[ec] : ec;
        const float pe = t.param[ec], me = a.m[t.mv_offset + ec], ve = a.v[t.mv_offset + ec];
        float acc = 0.f;
        for (int q0 = 0; q0 < t.nparts; q0 += 64 * 8) {
          float vq[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) vq[q] = t.grad_src[(size_t)min(q0 + lane + 64 * q, t.nparts - 1) * t.part_stride + src];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (q0 + lane + 64 * q < t.nparts) acc += vq[q];
        }
        const float ge = wave_total(acc);
        const AdamScalars k = scalars();
        if (live && lane == 0) {
          t.grad[e] = ge;
          if (k.apply && finite_f(ge)) {
            const float m1 = me + (ge - me) * k.one_m_b1, v1 = ve * k.beta2 + k.one_m_b2 * ge * ge;
            a.m[t.mv_offset + e] = m1; a.v[t.mv_offset + e] = v1;
            t.param[e] = pe - k.step_size * (m1 / (sqrtf(v1) / k.bc2_sqrt + k.eps));
          }
        }
      }
      return;
    }
    blk -= tk.nb_extra[x];
  }
  // ---- the last block: bias / free-scalar sums (B-term sums of strided columns), then -ELBO
  const int n_lb = a.param[6] ? 2 * s.nl : 0;
  const int n_items = s.H + n_lb + 2 * s.ngl;
  // (tensor tables indexed with constants only: a per-thread index into the kernel-argument arrays becomes a vector load of
  // the pointer followed by a dependent load through it)
  struct Item { const float* src; float *pp, *mp, *vp, *gp; int stride, e; };
  auto item_of = [&](int item) {
    Item it;
    if (item < s.H) {
      it.e = item; it.src = a.g_pre + item; it.stride = s.H;
      it.pp = a.param[4]; it.gp = a.grad[4]; it.mp = a.m + a.mv_offset[4]; it.vp = a.v + a.mv_offset[4];
    } else if (item < s.H + n_lb) {
      it.e = item - s.H; it.src = a.g_all + (size_t)it.e * B; it.stride = 1;
      it.pp = a.param[6]; it.gp = a.grad[6]; it.mp = a.m + a.mv_offset[6]; it.vp = a.v + a.mv_offset[6];
    } else {
      it.e = item - s.H - n_lb; it.src = a.g_all + (size_t)(2 * (s.nl + s.ng) + it.e) * B; it.stride = 1;
      it.pp = a.param[0]; it.gp = a.grad[0]; it.mp = a.m + a.mv_offset[0]; it.vp = a.v + a.mv_offset[0];
    }
    return it;
  };
  auto finish_item = [&](const Item& it, const AdamScalars& k, float ge, float pe, float me, float ve) {
    it.gp[it.e] = ge;
    if (k.apply && finite_f(ge)) {  // (adam_elem's arithmetic)
      me += (ge - me) * k.one_m_b1;
      ve = ve * k.beta2 + k.one_m_b2 * ge * ge;
      it.mp[it.e] = me;
      it.vp[it.e] = ve;
      it.pp[it.e] = pe - k.step_size * (me / (sqrtf(ve) / k.bc2_sqrt + k.eps));
    }
  };
  // first round: every thread (threads past the list redo item 0's loads and write nothing), then the gate's barriers
  const bool has_item = tid < n_items;
  const Item it0 = item_of(has_item ? tid : 0);
  const float i_pe = it0.pp[it0.e], i_me = it0.mp[it0.e], i_ve = it0.vp[it0.e];
  const float i_acc = dot_rows(it0.src, it0.stride, nullptr, 0, B);
  const AdamScalars k = scalars();
  if (has_item) finish_item(it0, k, i_acc, i_pe, i_me, i_ve);
  for (int item = tid + UPD_T; item < n_items; item += UPD_T) {
    const Item it = item_of(item);
    const float pe = it.pp[it.e], me = it.mp[it.e], ve = it.vp[it.e];
    finish_item(it, k, dot_rows(it.src, it.stride, nullptr, 0, B), pe, me, ve);
  }
  // -ELBO = -mean_b(lse - log n_iwae) (training.py:144-149): thread r takes rows r, r + 256, ...; wavefronts, then the
  // four wavefront sums in order
  const float log_n = logf((float)a.iwae.n_iwae_total);
  float acc = tid < B ? lse_v - log_n : 0.f;
  for (int r = tid + UPD_T; r < B; r += UPD_T) acc += a.iwae.lse[r] - log_n;
  acc = wave_total(acc);
  __syncthreads();
  if (lane == 0) part[wid][0] = acc;
  __syncthreads();
  if (tid == 0) {
    const float t = (part[0][0] + part[1][0]) + (part[2][0] + part[3][0]);
    a.iwae.loss[0] = -t / (float)B;
    // a skipped update does not count as a step (the row kernel counted it: only this block may take it back, and every
    // other block of this launch is gated off by the same rows)
    if (a.state && !k.apply) a.state[0] = a.state[0] - 1.f;
  }
}

// ---- launcher ---------------------------------------------------------------------------------------------------
size_t step_tail_rows_lds_bytes(const vihds_encoder_shape& s, int P, int S, bool u_lds) {
  const TailDims d = tail_dims(s);
  return sizeof(float) * ((u_lds ? (size_t)pad4(S * P) : 0) + pad4(S) + pad4(2 * P) + 64 + pad4(d.NPOOL) +
                          pad4(2 * s.nl * s.H) + 2 * ROWS_NW + 12 * (size_t)(s.nl > ROWS_NW ? s.nl : ROWS_NW));
}
size_t step_tail_update_lds_bytes(const vihds_encoder_shape& s) {
  const TailDims d = tail_dims(s);
  size_t conv = (size_t)pad4(s.B * d.Lc) + (size_t)s.B * s.L;
  const size_t lin = (size_t)s.B * s.H, partials = (size_t)(UPD_KMAX + 1) * UPD_T;
  if (partials > conv) conv = partials;
  return sizeof(float) * (conv > lin ? conv : lin);
}
bool step_tail_supported(const vihds_encoder_shape& s, int P, int S) {
  return step_tail_rows_lds_bytes(s, P, S, false) <= 60 * 1024 && step_tail_update_lds_bytes(s) <= 60 * 1024 &&
         s.K <= UPD_KMAX;
}
void launch_step_tail(const vihds_encoder_shape& s, const vihds_step_tail_args& a, hipStream_t st) {
  const TailDims d = tail_dims(s);
  const bool u_lds = step_tail_rows_lds_bytes(s, a.P, a.S, true) <= 60 * 1024;
  const size_t lds = step_tail_rows_lds_bytes(s, a.P, a.S, u_lds);
  const int rest = a.P + a.off_n - s.nl;  // tasks without a path into the encoder's hidden layer: one wavefront each
  const dim3 grid(s.B, 1 + (rest > 0 ? (rest + ROWS_NW - 1) / ROWS_NW : 0));
#define VIHDS_TAIL_ROWS(UL, HM) hipLaunchKernelGGL((step_tail_rows_kernel<UL, HM>), grid, dim3(ROWS_T), lds, st, s, a)
  if (a.phase == 2) {
  } else if (u_lds) {
    if (s.H <= 32) VIHDS_TAIL_ROWS(true, 32);
    else if (s.H <= 52) VIHDS_TAIL_ROWS(true, 52);
    else VIHDS_TAIL_ROWS(true, 64);
  } else {
    if (s.H <= 32) VIHDS_TAIL_ROWS(false, 32);
    else if (s.H <= 52) VIHDS_TAIL_ROWS(false, 52);
    else VIHDS_TAIL_ROWS(false, 64);
  }
#undef VIHDS_TAIL_ROWS
  if (a.phase == 1) return;
  TailTasks tk;
  tk.nb_lin = (s.H * d.NPOOL + UPD_T - 1) / UPD_T;
  tk.nb_conv = s.F * s.C_in;
  tk.nb_localw = (2 * s.nl * d.NX + UPD_T - 1) / UPD_T;
  tk.nb_gcondw = (2 * s.ng * d.NG + UPD_T - 1) / UPD_T;
  tk.nb_off = a.off_n > 0 ? 1 : 0;
  int nb_x = 0;
  for (int x = 0; x < VIHDS_TAIL_MAX_EXTRA; ++x) {
    const vihds_tail_tensor& t = a.extra[x];
    tk.nb_extra[x] = x < a.n_extra ? (t.nparts == 1 ? (t.size + UPD_T - 1) / UPD_T : (t.size + UPD_NW - 1) / UPD_NW) : 0;
    nb_x += tk.nb_extra[x];
  }
  const int nblocks = tk.nb_lin + tk.nb_conv + tk.nb_localw + tk.nb_gcondw + tk.nb_off + nb_x + 1;
  hipLaunchKernelGGL(step_tail_update_kernel, dim3(nblocks), dim3(UPD_T), step_tail_update_lds_bytes(s), st, s, tk, a);
}

}  // namespace vihds
#ifdef VIHDS_TAIL_STAMPS
extern "C" int vihds_debug_tail_stamps(unsigned long long* buf) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(vihds::vihds_tail_stamp_buf), &buf, sizeof(buf));
}
#endif
