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
  // x in units of K (u = x / K): du = gr u (1 - u).  Stage values us[s], returns u'.
  __device__ __forceinline__ static float xstep(float h, const float* gr, float u, float* us) {
    float w[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      float v = u;
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (a(s, r) != 0.f) v = fmaf(h * a(s, r), w[r], v);
      us[s] = v;
      w[s] = gr[s] * fmaf(-v, v, v);
    }
    float o = u;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (b(s) != 0.f) o = fmaf(h * b(s), w[s], o);
    return o;
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

constexpr int DR_SCAN_TPB = 8;  // trajectories per 256-thread block (2 per wavefront)

template <int SOLVER>
__host__ __device__ inline size_t dr_scan_lds_floats_per_traj(int items) {
  return (size_t)3 * (32 * items + 1) * Rk<SOLVER>::NS;  // tables G (sigmoid), U (x / K), GB (gamma adjoints)
}

template <int VERSION, int SOLVER, int ITEMS>
__device__ __forceinline__ void dr_scan_train_body(const OdeArgs& a, float* lds) {
  using M = DrConstant<VERSION>;
  using D = DrLanes<VERSION>;
  using R = Rk<SOLVER>;
  constexpr int NS = R::NS;
  constexpr int KP = 32 * ITEMS + 1;
  const int lane = threadIdx.x & 63, l = lane & 31;
  const int tib = threadIdx.x >> 5;  // trajectory within the block
  const int i0 = blockIdx.x * DR_SCAN_TPB + tib;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int K = a.T - 1;
  const size_t n = a.n;
  float* tG = lds + (size_t)tib * 3 * KP * NS;
  float* tU = tG + KP * NS;
  float* tB = tU + KP * NS;
  auto th = [&](int slot) { return a.theta[(size_t)a.slot_row[slot] * n + i]; };

  // ---- parameters of this trajectory (every lane of its 32 holds them) -----------------------------------------
  float c[2];
  c[0] = clampf(expf(a.cond[b * a.C + 0]) - 1.f, 1e-12f, 1e6f);
  c[1] = clampf(expf(a.cond[b * a.C + 1]) - 1.f, 1e-12f, 1e6f);
  const float r = clampf(th(M::S_r), 0.f, 4.f), Kc = clampf(th(M::S_K), 0.f, 4.f), invK = frcp(Kc);
  const float tlag = th(M::S_tlag), rc = th(M::S_rc);
  typename D::HillTerm H;
  float fR, fS;
  D::hill(a, i, l & 7, c, H, fR, fS);  // (each 8-lane group evaluates the power terms side by side)
  enum { RFP, WW, LUXR, LASR, YFP, CFP, NSP };
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
  // promoters: yfp <- P81, cfp <- P76;  c P = ce + cm t,  t = kb / (1 + kb),  kb = KGR fR luxR^2 + KGS fS lasR^2
  float pe[2], pKR[2], pKS[2], pc[2], pcR[2], pcS[2];
  pe[0] = th(M::S_e81); pKR[0] = th(M::S_KGR81); pKS[0] = th(M::S_KGS81); pc[0] = rc * aY;
  pe[1] = th(M::S_e76); pKR[1] = th(M::S_KGR76); pKS[1] = th(M::S_KGS76); pc[1] = rc * aC;
  VIHDS_UNROLL for (int q = 0; q < 2; ++q) { pcR[q] = pKR[q] * fR; pcS[q] = pKS[q] * fS; }
  float prec[4], lc[4];
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) { prec[j] = th(M::NSLOT + j); lc[j] = LOG2PI_F - logf(prec[j]); }
  float y0[NSP];
  const float x0 = th(M::SI + 0);
  y0[RFP] = th(M::SI + 1); y0[YFP] = th(M::SI + 2); y0[CFP] = th(M::SI + 3); y0[WW] = 0.f;
  y0[LUXR] = th(M::SI + 4); y0[LASR] = th(M::SI + 5);

  // ---- this lane's steps ---------------------------------------------------------------------------------------
  const int k0 = l * ITEMS;
  const float h0 = a.times[1] - a.times[0];
  float hh[ITEMS], tt0[ITEMS], dtt[ITEMS];
  bool valid[ITEMS];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    const int k = k0 + m;
    valid[m] = k < K;
    const int kc = valid[m] ? k : K - 1;
    tt0[m] = a.times[kc];
    dtt[m] = a.times[kc + 1] - tt0[m];
    hh[m] = R::FIXED_H ? h0 : dtt[m];
  }
  const float* ob = a.obs + (size_t)b * 4 * a.T;

  // ---- 1. sigmoid table -------------------------------------------------------------------------------------------
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      tG[(k0 + m) * NS + s] = sigmoid_f(4.f * (fmaf(R::c(s), dtt[m], tt0[m]) - tlag));
  }
  wave_sync();

  // ---- 2. the x chain (u = x / K), redundantly in all lanes of the half-wave; lane 0 records the stage values ------
  {
    float u = x0 * invK;
    float tk = a.times[0];
    for (int k = 0; k < K; ++k) {
      const float tn = a.times[k + 1];
      const float h = R::FIXED_H ? h0 : tn - tk;
      tk = tn;
      float gr[NS], us[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) gr[s] = r * tG[k * NS + s];
      u = R::xstep(h, gr, u, us);
      if (l == 0) {
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) tU[k * NS + s] = us[s];
      }
    }
    if (l == 0) tU[K * NS] = u;
  }
  wave_sync();

  // gamma at this lane's stages
  float gam[ITEMS][NS], us_[ITEMS][NS];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      const int e = (valid[m] ? k0 + m : K - 1) * NS + s;
      us_[m][s] = tU[e];
      const float g = r * tG[e];
      gam[m][s] = fmaf(-g, us_[m][s], g);
    }
  }
  const float xK = Kc * tU[K * NS];  // state of x at the last grid point

  // ---- 3. level 1: rfp, W, luxR, lasR --------------------------------------------------------------------------------
  float ys[NSP];  // state at the start of this lane's first step
  {
    Aff lm[4];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) lm[j] = {1.f, 0.f};
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        float as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[m][s] + delta[j]; Fs[s] = F1[j]; }
        Aff st;
        R::affine(hh[m], as, Fs, st.a, st.b);
        if (!valid[m]) st = {1.f, 0.f};
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
  // ---- 4. real steps of level 1 (states at this lane's grid points, stage values of luxR / lasR), promoters,
  //         affine maps of yfp / cfp ----------------------------------------------------------------------------------
  float yk[NSP][ITEMS];   // state at the START of step m (grid point k0 + m)
  float yend[NSP];        // state after this lane's last valid step
  Aff st2[2][ITEMS];
  {
    float cur[4];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) cur[j] = ys[j];
    Aff lm[2] = {{1.f, 0.f}, {1.f, 0.f}};
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      float YR[NS], YS[NS], dummy[NS];
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        yk[j][m] = cur[j];
        float as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[m][s] + delta[j]; Fs[s] = F1[j]; }
        const float nx = R::real(hh[m], as, Fs, cur[j], j == LUXR ? YR : (j == LASR ? YS : dummy));
        cur[j] = valid[m] ? nx : cur[j];
      }
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
        float as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const float kb = fmaf(pcS[q], YS[s] * YS[s], pcR[q] * (YR[s] * YR[s]));
          const float t = kb * frcp(1.f + kb);
          Fs[s] = pc[q] * fmaf(1.f - pe[q], t, pe[q]);
          as[s] = gam[m][s] + delta[YFP + q];
        }
        R::affine(hh[m], as, Fs, st2[q][m].a, st2[q][m].b);
        if (!valid[m]) st2[q][m] = {1.f, 0.f};
        lm[q] = after(st2[q][m], lm[q]);
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
  // ---- 5. log-likelihood at this lane's grid points (+ the last grid point in the lane that owns step K-1) ---------
  float qinj[4][ITEMS], qK[4];  // d logp_j / d xpred_j at the grid points (unit weight on the four signals)
  float precb[4] = {0.f, 0.f, 0.f, 0.f};
  float a530b = 0.f, a480b = 0.f;
  const bool owner_last = (K - 1) / ITEMS == l;
  {
    float lp[4] = {0.f, 0.f, 0.f, 0.f};
    auto point = [&](int k, bool on, float x, float rfp, float yf, float cf, float w, float* qo) {
      const float f530 = a530 * w, f480 = a480 * w;
      const float xp[4] = {x, x * rfp, x * (yf + f530), x * (cf + f480)};
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float e = xp[j] - ob[j * a.T + k];
        const float t = -0.5f * fmaf(prec[j] * e, e, lc[j]);
        lp[j] += on ? t : 0.f;
        qo[j] = on ? -prec[j] * e : 0.f;
        precb[j] += on ? (0.5f / prec[j] - 0.5f * e * e) : 0.f;
      }
      // f530 = a530 W, f480 = a480 W: their amplitudes only enter here
      a530b += qo[2] * x * w;
      a480b += qo[3] * x * w;
    };
    float cy = ys[YFP], cc = ys[CFP];
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      yk[YFP][m] = cy;
      yk[CFP][m] = cc;
      const int kc = valid[m] ? k0 + m : K - 1;
      float qo[4];
      point(kc, valid[m], Kc * us_[m][0], yk[RFP][m], cy, cc, yk[WW][m], qo);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) qinj[j][m] = qo[j];
      cy = valid[m] ? fmaf(st2[0][m].a, cy, st2[0][m].b) : cy;
      cc = valid[m] ? fmaf(st2[1][m].a, cc, st2[1][m].b) : cc;
    }
    point(K, owner_last, xK, yend[RFP], yend[YFP], yend[CFP], yend[WW], qK);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      const float tot = sum32(lp[j], lane);
      if (a.logp && live && l == 0) a.logp[(size_t)j * n + i] = tot;
    }
  }

  // ---- 6. adjoint ---------------------------------------------------------------------------------------------------
  // grid-point injections of species j at point k:  rfp: q1 x, yfp: q2 x, cfp: q3 x, W: x (a530 q2 + a480 q3),
  // x: q0 + q1 rfp + q2 (yfp + f530) + q3 (cfp + f480); luxR, lasR: none.
  auto ginj = [&](int j, const float* q, float x) {
    return j == RFP ? q[1] * x : (j == YFP ? q[2] * x : (j == CFP ? q[3] * x : (j == WW ? x * fmaf(a530, q[2], a480 * q[3]) : 0.f)));
  };
  float sv[NSP], degb[NSP];
  VIHDS_UNROLL for (int j = 0; j < NSP; ++j) { sv[j] = 0.f; degb[j] = 0.f; }
  float svt[2] = {0.f, 0.f}, svr[2] = {0.f, 0.f}, c1b[2] = {0.f, 0.f}, c2b[2] = {0.f, 0.f};
  float lam0[NSP];  // adjoint of the initial state
  float offR[ITEMS], offS[ITEMS];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    offR[m] = 0.f;
    offS[m] = 0.f;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) tB[(k0 + m) * NS + s] = 0.f;
  }
  // reverse recurrence of one species over this lane's steps: Lambda_k = A_k (Lambda_{k+1} + post_k) + g_k + off_k,
  // post_k = g_K for k = K-1 (the terminal injection), 0 otherwise.  Returns Lambda_{k+1} + post_k for every step.
  auto reverse_lane = [&](const float* Ak, const float* gk, const float* offk, float gK, float* lam_next, float& lam_first) {
    Aff lm = {1.f, 0.f};
    VIHDS_UNROLL for (int m = ITEMS - 1; m >= 0; --m) {
      const bool last = valid[m] && (k0 + m == K - 1);
      Aff st = {Ak[m], fmaf(Ak[m], last ? gK : 0.f, gk[m] + offk[m])};
      if (!valid[m]) st = {1.f, 0.f};
      lm = after(st, lm);
    }
    const Aff sc = scan_down32(lm, lane);  // applied to 0 (everything beyond the last step): Lambda at this lane's first step
    const float nxt = lane_read(sc.b, lane + 1);
    float lam = l == 31 ? 0.f : nxt;
    VIHDS_UNROLL for (int m = ITEMS - 1; m >= 0; --m) {
      const bool last = valid[m] && (k0 + m == K - 1);
      lam += last ? gK : 0.f;
      lam_next[m] = lam;
      lam = valid[m] ? fmaf(Ak[m], lam, gk[m] + offk[m]) : lam;
    }
    lam_first = lam;
  };
  const float zero_items[ITEMS] = {};
  // level 2 (yfp, cfp) with the promoter adjoints -> stage injections for luxR / lasR
  {
    float lamn[2][ITEMS];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      float Ak[ITEMS], gk[ITEMS];
      VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
        Ak[m] = st2[q][m].a;
        float qq[4];
        VIHDS_UNROLL for (int j = 0; j < 4; ++j) qq[j] = qinj[j][m];
        gk[m] = ginj(YFP + q, qq, Kc * us_[m][0]);
      }
      reverse_lane(Ak, gk, zero_items, ginj(YFP + q, qK, xK), lamn[q], lam0[YFP + q]);
    }
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      // stage values of luxR / lasR again, promoters, stage values of yfp / cfp
      float YR[NS], YS[NS], aR_[NS], aS_[NS], FR[NS], FS[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        aR_[s] = gam[m][s] + delta[LUXR]; aS_[s] = gam[m][s] + delta[LASR];
        FR[s] = F1[LUXR]; FS[s] = F1[LASR];
      }
      R::real(hh[m], aR_, FR, yk[LUXR][m], YR);
      R::real(hh[m], aS_, FS, yk[LASR][m], YS);
      float JR[NS], JS[NS], gb[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { JR[s] = 0.f; JS[s] = 0.f; gb[s] = 0.f; }
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
        float as[NS], Fs[NS], t[NS], rd[NS], Y[NS], kbar[NS], Jz[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const float kb = fmaf(pcS[q], YS[s] * YS[s], pcR[q] * (YR[s] * YR[s]));
          rd[s] = frcp(1.f + kb);
          t[s] = kb * rd[s];
          Fs[s] = pc[q] * fmaf(1.f - pe[q], t[s], pe[q]);
          as[s] = gam[m][s] + delta[YFP + q];
          Jz[s] = 0.f;
        }
        R::real(hh[m], as, Fs, yk[YFP + q][m], Y);
        R::reverse(hh[m], as, valid[m] ? lamn[q][m] : 0.f, Jz, kbar);
        const float cm = pc[q] * (1.f - pe[q]);
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const float ab = -Y[s] * kbar[s];
          gb[s] += ab;
          degb[YFP + q] += ab;
          sv[YFP + q] += kbar[s];
          svt[q] = fmaf(kbar[s], t[s], svt[q]);
          svr[q] = fmaf(kbar[s], rd[s], svr[q]);
          const float kbb = (kbar[s] * cm) * (rd[s] * rd[s]);  // d t / d kb = rd (1 - t) = rd^2
          c1b[q] = fmaf(kbb, YR[s] * YR[s], c1b[q]);
          c2b[q] = fmaf(kbb, YS[s] * YS[s], c2b[q]);
          JR[s] = fmaf(kbb * (2.f * pcR[q]), YR[s], JR[s]);
          JS[s] = fmaf(kbb * (2.f * pcS[q]), YS[s], JS[s]);
        }
      }
      // the part of luxR / lasR's step adjoint that is driven by the stage injections (linear: added here, the part
      // driven by Lambda_{k+1} follows after their scan)
      float kv[NS];
      offR[m] = R::reverse(hh[m], aR_, 0.f, JR, kv);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { const float ab = -YR[s] * kv[s]; gb[s] += ab; degb[LUXR] += ab; sv[LUXR] += kv[s]; }
      offS[m] = R::reverse(hh[m], aS_, 0.f, JS, kv);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { const float ab = -YS[s] * kv[s]; gb[s] += ab; degb[LASR] += ab; sv[LASR] += kv[s]; }
      if (valid[m]) {
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) tB[(k0 + m) * NS + s] = gb[s];
      } else {
        offR[m] = 0.f;
        offS[m] = 0.f;
      }
    }
  }
  // level 1 (rfp, W, luxR, lasR): multipliers, scan, the Lambda-driven part of the step adjoints
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
    float Ak[ITEMS], gk[ITEMS], lamn[ITEMS];
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      float as[NS], Fz[NS], dB;
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[m][s] + delta[j]; Fz[s] = 0.f; }
      R::affine(hh[m], as, Fz, Ak[m], dB);
      float qq[4];
      VIHDS_UNROLL for (int jj = 0; jj < 4; ++jj) qq[jj] = qinj[jj][m];
      gk[m] = ginj(j, qq, Kc * us_[m][0]);
    }
    reverse_lane(Ak, gk, j == LUXR ? offR : (j == LASR ? offS : zero_items), ginj(j, qK, xK), lamn, lam0[j]);
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      float as[NS], Fs[NS], Y[NS], kbar[NS], Jz[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[m][s] + delta[j]; Fs[s] = F1[j]; Jz[s] = 0.f; }
      R::real(hh[m], as, Fs, yk[j][m], Y);
      R::reverse(hh[m], as, valid[m] ? lamn[m] : 0.f, Jz, kbar);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const float ab = -Y[s] * kbar[s];
        degb[j] += ab;
        sv[j] += kbar[s];
        if (valid[m]) tB[(k0 + m) * NS + s] += ab;
      }
    }
  }
  // x: tangent multipliers a_s = -gr_s (1 - 2 u_s), injections -gamma_bar gr / K, scan, then r, tlag, K
  float rb = 0.f, tlb = 0.f, gbx = 0.f;
  {
    float Ak[ITEMS], gk[ITEMS], offk[ITEMS], lamn[ITEMS];
    float sg[ITEMS][NS], ax[ITEMS][NS], Jx[ITEMS][NS], gbo[ITEMS][NS];
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      float Fz[NS], kv[NS], dB;
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        sg[m][s] = tG[(valid[m] ? k0 + m : K - 1) * NS + s];
        const float g = r * sg[m][s];
        ax[m][s] = -g * fmaf(-2.f, us_[m][s], 1.f);
        gbo[m][s] = valid[m] ? tB[(k0 + m) * NS + s] : 0.f;
        Jx[m][s] = -gbo[m][s] * g * invK;
        Fz[s] = 0.f;
      }
      R::affine(hh[m], ax[m], Fz, Ak[m], dB);
      offk[m] = valid[m] ? R::reverse(hh[m], ax[m], 0.f, Jx[m], kv) : 0.f;
      const float inner[4] = {1.f, yk[RFP][m], yk[YFP][m] + a530 * yk[WW][m], yk[CFP][m] + a480 * yk[WW][m]};
      gk[m] = qinj[0][m] * inner[0] + qinj[1][m] * inner[1] + qinj[2][m] * inner[2] + qinj[3][m] * inner[3];
    }
    const float gK = qK[0] + qK[1] * yend[RFP] + qK[2] * (yend[YFP] + a530 * yend[WW]) + qK[3] * (yend[CFP] + a480 * yend[WW]);
    float lamx0;
    reverse_lane(Ak, gk, offk, gK, lamn, lamx0);
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      float kbar[NS];
      R::reverse(hh[m], ax[m], valid[m] ? lamn[m] : 0.f, Jx[m], kbar);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const float xs = Kc * us_[m][s];
        const float gtot = fmaf(kbar[s], xs, gbo[m][s]);       // adjoint of gamma_s from every species
        const float grb = gtot * (1.f - us_[m][s]);            // adjoint of gr_s
        const float g = r * sg[m][s];
        rb = fmaf(grb, sg[m][s], rb);
        tlb = fmaf(grb * g, 1.f - sg[m][s], tlb);
        gbx = fmaf(gtot * g, xs, gbx);
      }
    }
    // ---- epilogue: sums over the time axis, raw accumulators -> gradients of the theta rows ----------------------------
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
    // the initial-state adjoints sit in lane 0 of the trajectory
    if (live && l == 0) {
      put(M::SI + 0, lamx0);
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
__global__ void __launch_bounds__(256, 4) dr_scan_train_kernel(OdeArgs a) {
  extern __shared__ float lds[];
  dr_scan_train_body<VERSION, SOLVER, ITEMS>(a, lds);
}

// returns VIHDS_E_UNSUPPORTED when the time grid is longer than 32 lanes x 4 steps
template <int VERSION>
inline int launch_dr_scan_train(int solver, const OdeArgs& a, hipStream_t st) {
  const int K = a.T - 1;
  const int items = (K + 31) / 32;
  if (items > 4) return VIHDS_E_UNSUPPORTED;
  const dim3 grid((a.n + DR_SCAN_TPB - 1) / DR_SCAN_TPB), block(256);
#define VIHDS_SCASE2(SV, IT)                                                                                \
  case IT: {                                                                                                \
    const size_t lds = dr_scan_lds_floats_per_traj<SV>(IT) * DR_SCAN_TPB * sizeof(float);                   \
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
