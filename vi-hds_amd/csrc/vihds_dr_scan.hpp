// Time-parallel ("scan") training kernel for the double-receiver model (dr_constant v1/v2): log-likelihood and the
// unit-weight theta adjoint in one launch, with the time axis spread over lanes.
//
// The system (reference models/dr_constant.py:77-112) is linear in every state except OD:
//     dx   = gamma(x, t) x,                      gamma = r sigmoid(4 (t - tlag)) (1 - x / K)            (:81-85, :98)
//     dy_j = F_j - (gamma + delta_j) y_j          j = rfp, f530, f480, luxR, lasR    (F_j = rc a_j)      (:99, :102-105)
//     dy_j = c_j P_j(luxR, lasR) - (gamma + delta_j) y_j        j = yfp, cfp                               (:88-95, :100-101)
// so once the scalar x chain has been walked, one explicit Runge-Kutta step of every other species is an AFFINE map
// y_{k+1} = A_k y_k + B_k whose coefficients depend on the x stage values of step k only, and the discrete adjoint is a
// linear recurrence in every component, x included (Lambda_k = A_k Lambda_{k+1} + offset_k).  The per-step work is then
// independent across k and the recurrences are prefix scans (the same arithmetic regrouped; checked against the
// oracle's autograd in float64 by tests/probe/scan_proto.py for all five solvers).
//
// Mapping: 32 lanes per trajectory, lane l owns the steps k = l*ITEMS .. l*ITEMS+ITEMS-1 (ITEMS = ceil((T-1)/32) <= 4),
// two trajectories per wavefront, all 8 species of a step in one lane's registers.  A wavefront never waits for
// another one (no block barrier): its tables sit in its own slice of LDS.
//   1. sigmoid table  sig[k][s]          every lane its own steps                           (state independent)
//   2. x chain                           serial over k, all lanes of the half-wave in step  (u = x / K: 2 dependent
//                                        instructions per stage), stage values u[k][s] -> LDS
//   3. rfp, W (f530 = a530 W, f480 = a480 W), luxR, lasR: per-step affine maps, composed per lane, DPP scan over lanes
//   4. promoters at the stage values of luxR / lasR -> yfp, cfp the same way
//   5. log-likelihood at the grid points, segment sum
//   6. adjoint: reverse scans for yfp, cfp -> luxR, lasR (+ rfp, W) -> x, each followed by the per-step VJPs whose
//      sums over k are the parameter gradients (segment sums), then the epilogue shared with the lane kernels' algebra.
// Against the 8-lanes-per-trajectory kernel (vihds_dr_lanes.hpp): about half the instructions per trajectory, and
// 3 600 wavefronts instead of 900 at B=36, S=200.
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_dr_lanes.hpp"

namespace vihds {

// ---- explicit Runge-Kutta tableaux of the five fixed-grid schemes, steps in units of h ----------
template <int SOLVER>
struct Rk {
  static constexpr int NS = SOLVER == VIHDS_SOLVER_EULER ? 1 : (SOLVER == VIHDS_SOLVER_RK4 ? 4 : 2);
  static constexpr bool FIXED_H = SOLVER == VIHDS_SOLVER_MODEULER;  // solvers.py:12: h = times[1] - times[0]
  static constexpr float a(int s, int r) {
    if (SOLVER == VIHDS_SOLVER_RK4) {  // torchdiffeq 0.1 rk4_alt_step_func (3/8 rule)
      if (s == 1) return r == 0 ? (1.f / 3.f) : 0.f;
      if (s == 2) return r == 0 ? (-1.f / 3.f) : (r == 1 ? 1.f : 0.f);
      if (s == 3) return r == 0 ? 1.f : (r == 1 ? -1.f : (r == 2 ? 1.f : 0.f));
      return 0.f;
    }
    if (SOLVER == VIHDS_SOLVER_MIDPOINT) return (s == 1 && r == 0) ? 0.5f : 0.f;
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) return (s == 1 && r == 0) ? 1.f : 0.f;
    return 0.f;
  }
  static constexpr float b(int s) {
    if (SOLVER == VIHDS_SOLVER_RK4) return (s == 0 || s == 3) ? 0.125f : 0.375f;
    if (SOLVER == VIHDS_SOLVER_MIDPOINT) return s == 1 ? 1.f : 0.f;
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) return 0.5f;
    return 1.f;
  }
  static constexpr float c(int s) {  // stage time = t0 + c (t1 - t0)
    if (SOLVER == VIHDS_SOLVER_RK4) return s == 0 ? 0.f : (s == 1 ? (1.f / 3.f) : (s == 2 ? (2.f / 3.f) : 1.f));
    if (SOLVER == VIHDS_SOLVER_MIDPOINT) return s == 1 ? 0.5f : 0.f;
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) return s == 1 ? 1.f : 0.f;
    return 0.f;
  }

  // one step of dy = F_s - a_s y as the affine map y' = A y + B
  __device__ __forceinline__ static void affine(float h, const float* a_s, const float* F, float& A, float& B) {
    float kap[NS], rho[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      float al = 1.f, be = 0.f;
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (a(s, r) != 0.f) {
          al = fmaf(h * a(s, r), kap[r], al);
          be = fmaf(h * a(s, r), rho[r], be);
        }
      kap[s] = -a_s[s] * al;
      rho[s] = fmaf(-a_s[s], be, F[s]);
    }
    A = 1.f;
    B = 0.f;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (b(s) != 0.f) {
        A = fmaf(h * b(s), kap[s], A);
        B = fmaf(h * b(s), rho[s], B);
      }
  }
  // the step itself: stage values Y[s], returns y'
  __device__ __forceinline__ static float real(float h, const float* a_s, const float* F, float y, float* Y) {
    float k[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      float v = y;
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (a(s, r) != 0.f) v = fmaf(h * a(s, r), k[r], v);
      Y[s] = v;
      k[s] = fmaf(-a_s[s], v, F[s]);
    }
    float o = y;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (b(s) != 0.f) o = fmaf(h * b(s), k[s], o);
    return o;
  }
  // transposed step: lam1 = adjoint of y', J[s] = adjoints arriving at the stage values from elsewhere;
  // returns the adjoint of y, kbar[s] = adjoints of the stage derivatives.  Linear in (lam1, J).
  __device__ __forceinline__ static float reverse(float h, const float* a_s, float lam1, const float* J, float* kbar) {
    float Yb[NS];
    float lam = lam1;
    VIHDS_UNROLL for (int s = NS - 1; s >= 0; --s) {
      float kb = (h * b(s)) * lam1;
      VIHDS_UNROLL for (int r = s + 1; r < NS; ++r)
        if (a(r, s) != 0.f) kb = fmaf(h * a(r, s), Yb[r], kb);
      kbar[s] = kb;
      Yb[s] = fmaf(-a_s[s], kb, J[s]);
      lam += Yb[s];
    }
    return lam;
  }
  // x in units of K (u = x / K): du = gr u (1 - u), with C[s] = h gr_s given.  Stage values us[s], returns u'.
  // The dependent chain is two instructions per stage (u_s - u_s^2, then one fma with a pre-multiplied coefficient);
  // everything else is off the chain.
  __device__ __forceinline__ static float xstep(const float* C, float u, float* us) {
    float p[NS], w[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      float base = u;
      VIHDS_UNROLL for (int r = 0; r + 1 < s; ++r)
        if (a(s, r) != 0.f) base = fmaf(a(s, r), w[r], base);
      const float v = (s > 0 && a(s, s > 0 ? s - 1 : 0) != 0.f) ? fmaf(a(s, s > 0 ? s - 1 : 0) * C[s > 0 ? s - 1 : 0], p[s > 0 ? s - 1 : 0], base) : base;
      us[s] = v;
      p[s] = fmaf(-v, v, v);
      w[s] = C[s] * p[s];
    }
    float o = u;
    VIHDS_UNROLL for (int s = 0; s + 1 < NS; ++s)
      if (b(s) != 0.f) o = fmaf(b(s), w[s], o);
    return fmaf(b(NS - 1) * C[NS - 1], p[NS - 1], o);
  }
  // NS consecutive floats from / to LDS as one access
  __device__ __forceinline__ static void load(const float* p, float* o) {
    if (NS == 4) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[1] = v.y; o[NS > 2 ? 2 : 0] = v.z; o[NS > 3 ? 3 : 0] = v.w; }
    else if (NS == 2) { const float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[NS > 1 ? 1 : 0] = v.y; }
    else o[0] = p[0];
  }
  __device__ __forceinline__ static void store(float* p, const float* o) {
    if (NS == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[NS > 2 ? 2 : 0], o[NS > 3 ? 3 : 0]);
    else if (NS == 2) *reinterpret_cast<float2*>(p) = make_float2(o[0], o[NS > 1 ? 1 : 0]);
    else p[0] = o[0];
  }
};

// ---- affine maps and their scans over the 32 lanes of a trajectory -------------------------------------------------
struct Aff {
  float a, b;  // y -> a y + b
};
// g after f
__device__ __forceinline__ Aff after(const Aff& g, const Aff& f) { return {g.a * f.a, fmaf(g.a, f.b, g.b)}; }

__device__ __forceinline__ float lane_read(float v, int src_lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v)));
}
// inclusive scan in increasing lane order: lane l ends with  m_l o m_{l-1} o ... o m_0  (lower lanes act first)
__device__ __forceinline__ Aff scan_up32(Aff m, int lane) {
#define VIHDS_UP(CTRL)                          \
  {                                             \
    Aff p;                                      \
    p.a = dpp_mov<CTRL>(1.f, m.a);              \
    p.b = dpp_mov<CTRL>(0.f, m.b);              \
    m = after(m, p);                            \
  }
  VIHDS_UP(0x111) VIHDS_UP(0x112) VIHDS_UP(0x114) VIHDS_UP(0x118)  // row_shr:1,2,4,8 (lanes without a source keep identity)
#undef VIHDS_UP
  Aff p;
  p.a = lane_read(m.a, (lane & 32) + 15);
  p.b = lane_read(m.b, (lane & 32) + 15);
  if ((lane & 31) >= 16) m = after(m, p);
  return m;
}
// inclusive scan in decreasing lane order: lane l ends with  m_l o m_{l+1} o ... o m_31  (higher lanes act first)
__device__ __forceinline__ Aff scan_down32(Aff m, int lane) {
#define VIHDS_DN(CTRL)                          \
  {                                             \
    Aff p;                                      \
    p.a = dpp_mov<CTRL>(1.f, m.a);              \
    p.b = dpp_mov<CTRL>(0.f, m.b);              \
    m = after(m, p);                            \
  }
  VIHDS_DN(0x101) VIHDS_DN(0x102) VIHDS_DN(0x104) VIHDS_DN(0x108)  // row_shl:1,2,4,8
#undef VIHDS_DN
  Aff p;
  p.a = lane_read(m.a, (lane & 32) + 16);
  p.b = lane_read(m.b, (lane & 32) + 16);
  if ((lane & 31) < 16) m = after(m, p);
  return m;
}
// sum over the 32 lanes of a trajectory, result in all of them
__device__ __forceinline__ float sum32(float v, int lane) {
  v += dpp_all<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_all<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_all<0x141>(v);  // row_half_mirror
  v += dpp_all<0x140>(v);  // row_mirror
  v += lane_read(v, lane ^ 16);
  return v;
}
// no code motion of LDS accesses across this point (the lanes of a wavefront exchange data through LDS; the hardware
// executes a wavefront's LDS instructions in order, so no s_barrier is needed)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

constexpr int DR_SCAN_TPB = 2;       // trajectories per block: one wavefront, 32 lanes each
constexpr int DR_SCAN_THREADS = 64;

// LDS per block (floats): per trajectory the tables G (sigmoid) and U (x / K) [32 ITEMS + 1][NS]; per lane a record of
// 14 + NS fields per step (states at the step's grid point, the yfp / cfp maps, log-likelihood injections, gamma adjoints)
template <int SOLVER>
__host__ __device__ inline size_t dr_scan_lds_floats(int items) {
  return (size_t)DR_SCAN_TPB * 2 * (32 * items + 1) * Rk<SOLVER>::NS + (size_t)(14 + Rk<SOLVER>::NS) * items * DR_SCAN_THREADS +
         (size_t)(32 * items + 4);  // + the time grid
}
#define VIHDS_ROLLED _Pragma("clang loop unroll(disable)")
// profiling aid: kernel_variant = 3 | (phase << 8) makes the kernel return after that phase (tests/probe/scan_phases.py)
#define VIHDS_SCAN_STOP(PH) if ((a.kernel_variant >> 8) == (PH)) return;

template <int VERSION, int SOLVER, int ITEMS>
__device__ __forceinline__ void dr_scan_train_body(const OdeArgs& a, float* lds) {
  using M = DrConstant<VERSION>;
  using D = DrLanes<VERSION>;
  using R = Rk<SOLVER>;
  constexpr int NS = R::NS;
  constexpr int KP = 32 * ITEMS + 1;
  constexpr int NT = DR_SCAN_THREADS;
  const int lane = threadIdx.x & 63, l = lane & 31, half = lane >> 5;
  const int i0 = blockIdx.x * DR_SCAN_TPB + half;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int K = a.T - 1;
  const size_t n = a.n;
  float* tG = lds + (size_t)half * 2 * KP * NS;
  float* tU = tG + KP * NS;
  // per-lane record: field f of this lane's step m at rec[(f * ITEMS + m) * NT]  (consecutive lanes, consecutive words)
  float* rec = lds + (size_t)DR_SCAN_TPB * 2 * KP * NS + lane;
  float* tT = lds + (size_t)DR_SCAN_TPB * 2 * KP * NS + (size_t)(14 + NS) * ITEMS * NT;  // time grid [T]
  enum { RFP, WW, LUXR, LASR, YFP, CFP, NSP };
  enum { F_Y = 0, F_A2 = 6, F_B2 = 8, F_OFF = 8, F_Q = 10, F_GB = 14, F_TA = 6, F_TG = 7 };  // (B2 / OFF and A2 / TA,TG share)
  auto fld = [&](int f, int m) -> float& { return rec[(f * ITEMS + m) * NT]; };
  auto th = [&](int slot) { return a.theta[(size_t)a.slot_row[slot] * n + i]; };

  // the time grid -> LDS; this lane's observations -> the record fields that later hold the log-likelihood injections
  // (issued first: their latency hides behind the parameter stage)
  const int k0 = l * ITEMS;
  const float* ob = a.obs + (size_t)b * 4 * a.T;
  for (int k = lane; k < a.T; k += NT) tT[k] = a.times[k];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    const int kc = min(k0 + m, K - 1);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) fld(F_Q + j, m) = ob[j * a.T + kc];
  }
  float obK[4];
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) obK[j] = ob[j * a.T + K];
  wave_sync();

  // ---- parameters of this trajectory (every lane of its 32 holds them) -----------------------------------------
  float c[2];
  c[0] = clampf(expf(a.cond[b * a.C + 0]) - 1.f, 1e-12f, 1e6f);
  c[1] = clampf(expf(a.cond[b * a.C + 1]) - 1.f, 1e-12f, 1e6f);
  const float r = clampf(th(M::S_r), 0.f, 4.f), Kc = clampf(th(M::S_K), 0.f, 4.f), invK = frcp(Kc);
  const float tlag = th(M::S_tlag), rc = th(M::S_rc);
  typename D::HillTerm H;
  float fR, fS;
  D::hill(a, i, l & 7, c, H, fR, fS);  // (each 8-lane group evaluates the power terms side by side)
  float delta[NSP], F1[4];
  delta[RFP] = clampf(th(M::S_drfp), 1e-12f, 2.f);
  delta[WW] = 0.f;
  delta[LUXR] = clampf(th(M::S_dR), 1e-12f, 5.f);
  delta[LASR] = clampf(th(M::S_dS), 1e-12f, 5.f);
  delta[YFP] = clampf(th(M::S_dyfp), 1e-12f, 2.f);
  delta[CFP] = clampf(th(M::S_dcfp), 1e-12f, 2.f);
  const float aR = th(M::S_aR), aS = th(M::S_aS), aY = th(M::S_aYFP), aC = th(M::S_aCFP);
  const float a530 = th(M::S_a530), a480 = th(M::S_a480);
  F1[RFP] = rc; F1[WW] = rc; F1[LUXR] = rc * aR; F1[LASR] = rc * aS;
  // promoters: yfp <- P81, cfp <- P76;  c P = c e + c (1 - e) t,  t = kb / (1 + kb),  kb = KGR fR luxR^2 + KGS fS lasR^2
  float pe[2], pKR[2], pKS[2], pc[2], pcR[2], pcS[2];
  pe[0] = th(M::S_e81); pKR[0] = th(M::S_KGR81); pKS[0] = th(M::S_KGS81); pc[0] = rc * aY;
  pe[1] = th(M::S_e76); pKR[1] = th(M::S_KGR76); pKS[1] = th(M::S_KGS76); pc[1] = rc * aC;
  VIHDS_UNROLL for (int q = 0; q < 2; ++q) { pcR[q] = pKR[q] * fR; pcS[q] = pKS[q] * fS; }
  float prec[4];
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) prec[j] = th(M::NSLOT + j);
  float y0[NSP];
  const float x0 = th(M::SI + 0);
  y0[RFP] = th(M::SI + 1); y0[YFP] = th(M::SI + 2); y0[CFP] = th(M::SI + 3); y0[WW] = 0.f;
  y0[LUXR] = th(M::SI + 4); y0[LASR] = th(M::SI + 5);

  VIHDS_SCAN_STOP(1)
  const float h0 = tT[1] - tT[0];
  // step m of this lane: grid index (clamped for the padding steps beyond K), validity, step size
  struct Item {
    int kc;
    bool valid;
    float h, invh, t0, dt;
  };
  auto item = [&](int m) {
    Item it;
    it.valid = k0 + m < K;
    it.kc = it.valid ? k0 + m : K - 1;
    it.t0 = tT[it.kc];
    it.dt = tT[it.kc + 1] - it.t0;
    it.h = R::FIXED_H ? h0 : it.dt;
    it.invh = frcp(it.h);
    return it;
  };
  auto load_stage = [&](const Item& it, float* gam, float* us) {  // gamma_s = gr_s (1 - u_s) at the stages of step kc
    float C[NS];
    R::load(&tG[it.kc * NS], C);
    R::load(&tU[it.kc * NS], us);
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      const float g = C[s] * it.invh;
      gam[s] = fmaf(-g, us[s], g);
    }
  };
  auto load_q = [&](int m, float* q) { VIHDS_UNROLL for (int j = 0; j < 4; ++j) q[j] = fld(F_Q + j, m); };
  auto stage_sigmoid = [&](const Item& it, int s) { return sigmoid_f(4.f * (fmaf(R::c(s), it.dt, it.t0) - tlag)); };

  // ---- 1. table C[k][s] = h_k r sigmoid(4 (t_{k,s} - tlag)): what the x chain consumes (state independent) -----------
  VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
    const Item it = item(m);
    float C[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) C[s] = (it.h * r) * stage_sigmoid(it, s);
    R::store(&tG[(k0 + m) * NS], C);
  }
  wave_sync();

  VIHDS_SCAN_STOP(2)
  // ---- 2. the x chain (u = x / K), redundantly in all lanes of the half-wave; lane 0 records the stage values ------
  {
    float u = x0 * invK;
    float Cn[NS];
    R::load(&tG[0], Cn);
    for (int k = 0; k < K; ++k) {
      float C[NS], us[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) C[s] = Cn[s];
      R::load(&tG[(k + 1) * NS], Cn);  // (one step ahead; the table has a padding entry behind the last step)
      u = R::xstep(C, u, us);
      if (l == 0) R::store(&tU[k * NS], us);
    }
    if (l == 0) tU[K * NS] = u;
  }
  wave_sync();
  VIHDS_SCAN_STOP(3)
  const float xK = Kc * tU[K * NS];  // x at the last grid point

  // ---- 3. level 1 (rfp, W, luxR, lasR): per-step affine maps composed over this lane's steps, scan over lanes ----------
  float ys[NSP];  // state at this lane's first grid point
  {
    Aff lm[4];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) lm[j] = {1.f, 0.f};
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      float gam[NS], us[NS];
      load_stage(it, gam, us);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        float as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta[j]; Fs[s] = F1[j]; }
        Aff st;
        R::affine(it.h, as, Fs, st.a, st.b);
        if (!it.valid) st = {1.f, 0.f};
        lm[j] = after(st, lm[j]);
      }
    }
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      const Aff sc = scan_up32(lm[j], lane);
      const float end = fmaf(sc.a, y0[j], sc.b);
      const float prev = lane_read(end, lane - 1);
      ys[j] = l == 0 ? y0[j] : prev;
    }
  }
  VIHDS_SCAN_STOP(4)
  // ---- 4. the steps themselves for level 1 (states at this lane's grid points, stage values of luxR / lasR), promoters,
  //         affine maps of yfp / cfp ----------------------------------------------------------------------------------
  float yend[NSP];  // state after this lane's last valid step
  {
    float cur[4];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) cur[j] = ys[j];
    Aff lm[2] = {{1.f, 0.f}, {1.f, 0.f}};
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      float gam[NS], us[NS];
      load_stage(it, gam, us);
      float YR[NS], YS[NS], dummy[NS];
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        fld(F_Y + j, m) = cur[j];
        float as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta[j]; Fs[s] = F1[j]; }
        const float nx = R::real(it.h, as, Fs, cur[j], j == LUXR ? YR : (j == LASR ? YS : dummy));
        cur[j] = it.valid ? nx : cur[j];
      }
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
        float as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const float kb = fmaf(pcS[q], YS[s] * YS[s], pcR[q] * (YR[s] * YR[s]));
          const float t = kb * frcp(1.f + kb);
          Fs[s] = pc[q] * fmaf(1.f - pe[q], t, pe[q]);
          as[s] = gam[s] + delta[YFP + q];
        }
        Aff st;
        R::affine(it.h, as, Fs, st.a, st.b);
        if (!it.valid) st = {1.f, 0.f};
        fld(F_A2 + q, m) = st.a;
        fld(F_B2 + q, m) = st.b;
        lm[q] = after(st, lm[q]);
      }
    }
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) yend[j] = cur[j];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const Aff sc = scan_up32(lm[q], lane);
      const float end = fmaf(sc.a, y0[YFP + q], sc.b);
      const float prev = lane_read(end, lane - 1);
      ys[YFP + q] = l == 0 ? y0[YFP + q] : prev;
      yend[YFP + q] = end;
    }
  }
  VIHDS_SCAN_STOP(5)
  // ---- 5. log-likelihood at this lane's grid points (+ the last grid point in the lane that owns step K-1) ---------
  float precb[4] = {0.f, 0.f, 0.f, 0.f};
  float a530b = 0.f, a480b = 0.f;
  float qK[4];  // d logp_j / d xpred_j at the last grid point (unit weight on the four signals)
  const bool owner_last = (K - 1) / ITEMS == l;
  {
    float lc[4], lp[4] = {0.f, 0.f, 0.f, 0.f};
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) lc[j] = LOG2PI_F - logf(prec[j]);
    auto point = [&](const float* obs_k, bool on, float x, float rfp, float yf, float cf, float w, float* qo) {
      const float xp[4] = {x, x * rfp, x * fmaf(a530, w, yf), x * fmaf(a480, w, cf)};
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float e = xp[j] - obs_k[j];
        lp[j] += on ? -0.5f * fmaf(prec[j] * e, e, lc[j]) : 0.f;
        qo[j] = on ? -prec[j] * e : 0.f;
        precb[j] += on ? (0.5f / prec[j] - 0.5f * e * e) : 0.f;
      }
      a530b += qo[2] * x * w;  // f530 = a530 W, f480 = a480 W: their amplitudes only enter here
      a480b += qo[3] * x * w;
    };
    float cy = ys[YFP], cc = ys[CFP];
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      fld(F_Y + YFP, m) = cy;
      fld(F_Y + CFP, m) = cc;
      float qo[4], obk[4];
      load_q(m, obk);
      point(obk, it.valid, Kc * tU[it.kc * NS], fld(F_Y + RFP, m), cy, cc, fld(F_Y + WW, m), qo);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) fld(F_Q + j, m) = qo[j];
      cy = fmaf(fld(F_A2 + 0, m), cy, fld(F_B2 + 0, m));  // (identity map on the padding steps)
      cc = fmaf(fld(F_A2 + 1, m), cc, fld(F_B2 + 1, m));
    }
    point(obK, owner_last, xK, yend[RFP], yend[YFP], yend[CFP], yend[WW], qK);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      const float tot = sum32(lp[j], lane);
      if (a.logp && live && l == 0) a.logp[(size_t)j * n + i] = tot;
    }
  }

  VIHDS_SCAN_STOP(6)
  // ---- 6. adjoint ---------------------------------------------------------------------------------------------------
  // grid-point injection of species j:  rfp: q1 x, yfp: q2 x, cfp: q3 x, W: x (a530 q2 + a480 q3), luxR / lasR: none;
  // x: q0 + q1 rfp + q2 (yfp + f530) + q3 (cfp + f480).
  auto ginj = [&](int j, const float* q, float x) {
    return j == RFP ? q[1] * x : (j == YFP ? q[2] * x : (j == CFP ? q[3] * x : (j == WW ? x * fmaf(a530, q[2], a480 * q[3]) : 0.f)));
  };
  // Reverse recurrence of one species over this lane's steps:
  //     Lambda_k = A_k (Lambda_{k+1} + post_k) + g_k,    post_k = gK at k = K-1 (the terminal injection), else 0,
  // with (A_k, g_k) read through getA / getG.  lane_map: the composition over this lane's steps; lane_entry: Lambda
  // entering this lane from the steps above it (0 beyond the last lane).
  auto lane_map = [&](auto getA, auto getG, float gK) {
    Aff lm = {1.f, 0.f};
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      const bool valid = k0 + m < K, last = k0 + m == K - 1;
      const float A = getA(m);
      Aff st = {A, fmaf(A, last ? gK : 0.f, getG(m))};
      if (!valid) st = {1.f, 0.f};
      lm = after(st, lm);
    }
    return lm;
  };
  auto lane_entry = [&](const Aff& lm) {
    const Aff sc = scan_down32(lm, lane);  // applied to 0: Lambda at this lane's first grid point
    const float nxt = lane_read(sc.b, lane + 1);
    return l == 31 ? 0.f : nxt;
  };
  float sv[NSP], degb[NSP], lam0[NSP];
  VIHDS_UNROLL for (int j = 0; j < NSP; ++j) { sv[j] = 0.f; degb[j] = 0.f; }
  float svt[2] = {0.f, 0.f}, svr[2] = {0.f, 0.f}, c1b[2] = {0.f, 0.f}, c2b[2] = {0.f, 0.f};

  // level 2 (yfp, cfp) with the promoter adjoints -> stage injections for luxR / lasR
  {
    float lam[2];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const Aff lm = lane_map([&](int m) { return fld(F_A2 + q, m); },
                              [&](int m) { float qq[4]; load_q(m, qq); return ginj(YFP + q, qq, Kc * tU[min(k0 + m, K - 1) * NS]); },
                              ginj(YFP + q, qK, xK));
      lam[q] = lane_entry(lm);
    }
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      const bool last = k0 + m == K - 1;
      float gam[NS], us[NS], qq[4];
      load_stage(it, gam, us);
      load_q(m, qq);
      // stage values of luxR / lasR again, promoters, stage values of yfp / cfp
      float YR[NS], YS[NS], aR_[NS], aS_[NS], FR[NS], FS[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        aR_[s] = gam[s] + delta[LUXR]; aS_[s] = gam[s] + delta[LASR];
        FR[s] = F1[LUXR]; FS[s] = F1[LASR];
      }
      R::real(it.h, aR_, FR, fld(F_Y + LUXR, m), YR);
      R::real(it.h, aS_, FS, fld(F_Y + LASR, m), YS);
      float JR[NS], JS[NS], gb[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { JR[s] = 0.f; JS[s] = 0.f; gb[s] = 0.f; }
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
        float as[NS], Fs[NS], t[NS], rd[NS], Y[NS], kbar[NS], Jz[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const float kb = fmaf(pcS[q], YS[s] * YS[s], pcR[q] * (YR[s] * YR[s]));
          rd[s] = frcp(1.f + kb);
          t[s] = kb * rd[s];
          Fs[s] = pc[q] * fmaf(1.f - pe[q], t[s], pe[q]);
          as[s] = gam[s] + delta[YFP + q];
          Jz[s] = 0.f;
        }
        R::real(it.h, as, Fs, fld(F_Y + YFP + q, m), Y);
        const float lin = lam[q] + (last ? ginj(YFP + q, qK, xK) : 0.f);  // Lambda_{k+1} (+ terminal injection)
        R::reverse(it.h, as, it.valid ? lin : 0.f, Jz, kbar);
        if (it.valid) lam[q] = fmaf(fld(F_A2 + q, m), lin, ginj(YFP + q, qq, Kc * us[0]));
        const float cm = pc[q] * (1.f - pe[q]);
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const float ab = -Y[s] * kbar[s];
          gb[s] += ab;
          degb[YFP + q] += ab;
          sv[YFP + q] += kbar[s];
          svt[q] = fmaf(kbar[s], t[s], svt[q]);
          svr[q] = fmaf(kbar[s], rd[s], svr[q]);
          const float kbb = (kbar[s] * cm) * (rd[s] * rd[s]);  // d t / d kb = rd^2
          c1b[q] = fmaf(kbb, YR[s] * YR[s], c1b[q]);
          c2b[q] = fmaf(kbb, YS[s] * YS[s], c2b[q]);
          JR[s] = fmaf(kbb * (2.f * pcR[q]), YR[s], JR[s]);
          JS[s] = fmaf(kbb * (2.f * pcS[q]), YS[s], JS[s]);
        }
      }
      // the part of luxR / lasR's step adjoint that is driven by the stage injections (linear: taken here; the part
      // driven by Lambda_{k+1} follows after their scan)
      float kv[NS];
      const float oR = R::reverse(it.h, aR_, 0.f, JR, kv);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { const float ab = -YR[s] * kv[s]; gb[s] += ab; degb[LUXR] += ab; sv[LUXR] += kv[s]; }
      const float oS = R::reverse(it.h, aS_, 0.f, JS, kv);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { const float ab = -YS[s] * kv[s]; gb[s] += ab; degb[LASR] += ab; sv[LASR] += kv[s]; }
      fld(F_OFF + 0, m) = it.valid ? oR : 0.f;
      fld(F_OFF + 1, m) = it.valid ? oS : 0.f;
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) fld(F_GB + s, m) = gb[s];
    }
    lam0[YFP] = lam[0];
    lam0[CFP] = lam[1];
  }
  VIHDS_SCAN_STOP(7)
  // level 1 (rfp, W, luxR, lasR): multipliers and scan, then the Lambda-driven part of the step adjoints
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {  // multiplier and injection of every step -> TA, TG
      const Item it = item(m);
      float gam[NS], us[NS], as[NS], Fz[NS], qq[4], A, dB;
      load_stage(it, gam, us);
      load_q(m, qq);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta[j]; Fz[s] = 0.f; }
      R::affine(it.h, as, Fz, A, dB);
      fld(F_TA, m) = A;
      fld(F_TG, m) = ginj(j, qq, Kc * us[0]) + (j == LUXR ? fld(F_OFF + 0, m) : (j == LASR ? fld(F_OFF + 1, m) : 0.f));
    }
    const float gK = ginj(j, qK, xK);
    float lam = lane_entry(lane_map([&](int m) { return fld(F_TA, m); }, [&](int m) { return fld(F_TG, m); }, gK));
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      const bool last = k0 + m == K - 1;
      float gam[NS], us[NS], as[NS], Fs[NS], Y[NS], kbar[NS], Jz[NS];
      load_stage(it, gam, us);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta[j]; Fs[s] = F1[j]; Jz[s] = 0.f; }
      R::real(it.h, as, Fs, fld(F_Y + j, m), Y);
      const float lin = lam + (last ? gK : 0.f);
      R::reverse(it.h, as, it.valid ? lin : 0.f, Jz, kbar);
      if (it.valid) lam = fmaf(fld(F_TA, m), lin, fld(F_TG, m));
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const float ab = -Y[s] * kbar[s];
        degb[j] += ab;
        sv[j] += kbar[s];
        fld(F_GB + s, m) += ab;
      }
    }
    lam0[j] = lam;
  }
  VIHDS_SCAN_STOP(8)
  // x: tangent multipliers a_s = -gr_s (1 - 2 u_s), stage injections -gamma_bar gr / K, scan, then r, tlag, K
  float rb = 0.f, tlb = 0.f, gbx = 0.f, lamx;
  {
    auto x_stage = [&](int m, const Item& it, float* sg, float* us, float* ax, float* gbo, float* Jx) {
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        sg[s] = stage_sigmoid(it, s);
        us[s] = tU[it.kc * NS + s];
        const float g = r * sg[s];
        ax[s] = -g * fmaf(-2.f, us[s], 1.f);
        gbo[s] = it.valid ? fld(F_GB + s, m) : 0.f;
        Jx[s] = -gbo[s] * g * invK;
      }
    };
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      float sg[NS], us[NS], ax[NS], gbo[NS], Jx[NS], Fz[NS], kv[NS], qq[4], A, dB;
      x_stage(m, it, sg, us, ax, gbo, Jx);
      load_q(m, qq);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) Fz[s] = 0.f;
      R::affine(it.h, ax, Fz, A, dB);
      const float off = it.valid ? R::reverse(it.h, ax, 0.f, Jx, kv) : 0.f;
      const float w = fld(F_Y + WW, m);
      fld(F_TA, m) = A;
      fld(F_TG, m) = off + qq[0] + qq[1] * fld(F_Y + RFP, m) + qq[2] * fmaf(a530, w, fld(F_Y + YFP, m)) +
                     qq[3] * fmaf(a480, w, fld(F_Y + CFP, m));
    }
    const float gK = qK[0] + qK[1] * yend[RFP] + qK[2] * fmaf(a530, yend[WW], yend[YFP]) + qK[3] * fmaf(a480, yend[WW], yend[CFP]);
    float lam = lane_entry(lane_map([&](int m) { return fld(F_TA, m); }, [&](int m) { return fld(F_TG, m); }, gK));
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      const bool last = k0 + m == K - 1;
      float sg[NS], us[NS], ax[NS], gbo[NS], Jx[NS], kbar[NS];
      x_stage(m, it, sg, us, ax, gbo, Jx);
      const float lin = lam + (last ? gK : 0.f);
      R::reverse(it.h, ax, it.valid ? lin : 0.f, Jx, kbar);
      if (it.valid) lam = fmaf(fld(F_TA, m), lin, fld(F_TG, m));
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const float xs = Kc * us[s];
        const float gtot = fmaf(kbar[s], xs, gbo[s]);  // adjoint of gamma_s from every species
        const float grb = gtot * (1.f - us[s]);        // adjoint of gr_s
        const float g = r * sg[s];
        rb = fmaf(grb, sg[s], rb);
        tlb = fmaf(grb * g, 1.f - sg[s], tlb);
        gbx = fmaf(gtot * g, xs, gbx);
      }
    }
    lamx = lam;
  }
  VIHDS_SCAN_STOP(9)
  // ---- epilogue: sums over the time axis, raw accumulators -> gradients of the theta rows --------------------------------
  {
    auto put = [&](int slot, float v) { a.g_theta[(size_t)a.slot_row[slot] * n + i] = v; };
    auto raw = [&](int slot) { return a.theta[(size_t)a.slot_row[slot] * n + i]; };
    VIHDS_UNROLL for (int j = 0; j < NSP; ++j) { sv[j] = sum32(sv[j], lane); degb[j] = sum32(degb[j], lane); }
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      svt[q] = sum32(svt[q], lane); svr[q] = sum32(svr[q], lane); c1b[q] = sum32(c1b[q], lane); c2b[q] = sum32(c2b[q], lane);
    }
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) precb[j] = sum32(precb[j], lane);
    rb = sum32(rb, lane); tlb = sum32(tlb, lane); gbx = sum32(gbx, lane);
    a530b = sum32(a530b, lane); a480b = sum32(a480b, lane);
    // c P = c e + c (1 - e) t:  c_bar = sv e + svt (1 - e),  e_bar = c sum kbar (1 - t) = c svr
    float cbar[2], ebar[2];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      cbar[q] = fmaf(svt[q], 1.f - pe[q], sv[YFP + q] * pe[q]);
      ebar[q] = pc[q] * svr[q];
    }
    const float fRb = c1b[0] * pKR[0] + c1b[1] * pKR[1], fSb = c2b[0] * pKS[0] + c2b[1] * pKS[1];
    const float rcb = sv[RFP] + sv[WW] + sv[LUXR] * aR + sv[LASR] * aS + cbar[0] * aY + cbar[1] * aC;
    const typename D::HillAdj HA = D::hill_vjp(a, i, l & 7, c, H, fRb, fSb);
    if (live && l == 0) {  // (the initial-state adjoints sit in lane 0 of the trajectory)
      put(M::SI + 0, lamx);
      put(M::SI + 1, lam0[RFP]); put(M::SI + 2, lam0[YFP]); put(M::SI + 3, lam0[CFP]);
      put(M::SI + 4, lam0[LUXR]); put(M::SI + 5, lam0[LASR]);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) put(M::NSLOT + j, precb[j]);
      put(M::S_r, rb * clamp_pass(raw(M::S_r), 0.f, 4.f));
      put(M::S_K, gbx * invK * invK * clamp_pass(raw(M::S_K), 0.f, 4.f));
      put(M::S_tlag, -4.f * tlb);
      put(M::S_rc, rcb);
      put(M::S_drfp, degb[RFP] * clamp_pass(raw(M::S_drfp), 1e-12f, 2.f));
      put(M::S_dyfp, degb[YFP] * clamp_pass(raw(M::S_dyfp), 1e-12f, 2.f));
      put(M::S_dcfp, degb[CFP] * clamp_pass(raw(M::S_dcfp), 1e-12f, 2.f));
      put(M::S_dR, degb[LUXR] * clamp_pass(raw(M::S_dR), 1e-12f, 5.f));
      put(M::S_dS, degb[LASR] * clamp_pass(raw(M::S_dS), 1e-12f, 5.f));
      put(M::S_e81, ebar[0]); put(M::S_KGR81, c1b[0] * fR); put(M::S_KGS81, c2b[0] * fS);
      put(M::S_e76, ebar[1]); put(M::S_KGR76, c1b[1] * fR); put(M::S_KGS76, c2b[1] * fS);
      put(M::S_aYFP, cbar[0] * rc); put(M::S_aCFP, cbar[1] * rc);
      put(M::S_a530, a530b); put(M::S_a480, a480b);
      put(M::S_aR, sv[LUXR] * rc); put(M::S_aS, sv[LASR] * rc);
      put(M::S_nR, HA.nR); put(M::S_nS, HA.nS); put(M::S_H0, HA.H0); put(M::S_H1, HA.H1);
      if (VERSION == 1) { put(M::S_H2, HA.H2); put(M::S_H3, HA.H3); }
    }
  }
}

template <int VERSION, int SOLVER, int ITEMS>
__global__ void __launch_bounds__(DR_SCAN_THREADS) dr_scan_train_kernel(OdeArgs a) {
  extern __shared__ float lds[];
  dr_scan_train_body<VERSION, SOLVER, ITEMS>(a, lds);
}

// returns VIHDS_E_UNSUPPORTED when the time grid is longer than 32 lanes x 4 steps
template <int VERSION>
inline int launch_dr_scan_train(int solver, const OdeArgs& a, hipStream_t st) {
  const int K = a.T - 1;
  const int items = (K + 31) / 32;
  if (items > 4) return VIHDS_E_UNSUPPORTED;
  const dim3 grid((a.n + DR_SCAN_TPB - 1) / DR_SCAN_TPB), block(DR_SCAN_THREADS);
#define VIHDS_SCASE2(SV, IT)                                                                                \
  case IT: {                                                                                                \
    const size_t lds = dr_scan_lds_floats<SV>(IT) * sizeof(float);                                          \
    hipLaunchKernelGGL((dr_scan_train_kernel<VERSION, SV, IT>), grid, block, lds, st, a);                   \
    return VIHDS_OK;                                                                                        \
  }
#define VIHDS_SCASE(SV)                                   \
  case SV:                                                \
    switch (items) {                                      \
      VIHDS_SCASE2(SV, 1) VIHDS_SCASE2(SV, 2) VIHDS_SCASE2(SV, 3) VIHDS_SCASE2(SV, 4) \
    }                                                     \
    break;
  switch (solver) {
    VIHDS_SCASE(VIHDS_SOLVER_MODEULER)
    VIHDS_SCASE(VIHDS_SOLVER_MODEULERWHILE)
    VIHDS_SCASE(VIHDS_SOLVER_EULER)
    VIHDS_SCASE(VIHDS_SOLVER_MIDPOINT)
    VIHDS_SCASE(VIHDS_SOLVER_RK4)
  }
#undef VIHDS_SCASE
#undef VIHDS_SCASE2
  return VIHDS_E_BADARG;
}

}  // namespace vihds
