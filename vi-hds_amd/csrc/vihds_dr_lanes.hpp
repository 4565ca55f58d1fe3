// Lane-split kernels for the double-receiver model (dr_constant v1/v2): 8 lanes per trajectory, one species per
// lane, 8 trajectories per wavefront.
//
// Why: at the headline shape (B=36, S=200 -> 7 200 trajectories) one thread per trajectory gives 113 wavefronts
// for 1 024 SIMDs and every one of them walks a ~340-deep dependent chain; the chip is idle and the kernel is
// latency-bound (SURVEY.md 7 "hard parts").  All eight species obey the same form
//     dy_j = c_j * P_j + (s_j * gamma - deg_j) * y_j,      P_j = (e_j + KGR_j bR + KGS_j bS) / (1 + KGR_j bR + KGS_j bS)
// (reference models/dr_constant.py:98-105; for the six species without a promoter e_j = 1, KGR_j = KGS_j = 0 so P_j = 1),
// so one lane can own one species with per-lane constants, 8x more wavefronts are in flight and the chain per lane
// is ~5x shorter.  Cross-lane traffic is DPP only (no LDS): x, LuxR, LasR are broadcast inside each 8-lane group
// (quad_perm + row_shr/row_shl:4 with a bank mask), the adjoint's three sums are butterfly reductions
// (quad_perm xor 1, xor 2, row_half_mirror).
//
// Used below VIHDS_LANE_SPLIT_MAX_N trajectories (default 16 384); above that the one-thread-per-trajectory kernels
// fill the machine and do less redundant work.  Numerics: same formulas as vihds_models.hpp (DrConstant), same tests.
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_args.hpp"
#include "vihds_models.hpp"
#include "vihds_rng.hpp"
#include "vihds_iwae_inline.hpp"

namespace vihds {

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                               CTRL, ROW_MASK, BANK_MASK, false));
}
// value of lane J (0..7) of this lane's 8-lane group, in all 8 lanes
// full-mask DPP move: every lane is written, so there is no 'old' value to preserve (saves a v_mov per use)
template <int CTRL>
__device__ __forceinline__ float dpp_all(float src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, src), CTRL, 0xf, 0xf, true));
}
template <int J>
__device__ __forceinline__ float bcast8(float v) {
  constexpr int q = J & 3;
  float t = dpp_all<q | (q << 2) | (q << 4) | (q << 6)>(v);  // quad_perm: lane q of each quad
  if (J < 4) t = dpp_mov<0x114, 0xf, 0xA>(t, t);                  // row_shr:4 into banks 1,3 (lanes 4-7, 12-15)
  else t = dpp_mov<0x104, 0xf, 0x5>(t, t);                        // row_shl:4 into banks 0,2
  return t;
}
// sum over the 8-lane group, result in all 8 lanes
__device__ __forceinline__ float sum8(float v) {
  v += dpp_all<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_all<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_all<0x141>(v);  // row_half_mirror: lane l <- lane 7-l of the same half row
  return v;
}
// sum over the quad, result in all 4 lanes
__device__ __forceinline__ float sum4(float v) {
  v += dpp_all<0xB1>(v);
  v += dpp_all<0x4E>(v);
  return v;
}

struct DrLane {
  // shared by the 8 lanes of a trajectory
  float r, invK, K, tlag, rc, fR, fS;
  // this lane's species
  float sgn, deg, a, c, e, KGR, KGS, prec;
  // promoter in the lane's own operand order: lanes 2 / 3 see (v1, v2) = (LuxR, LasR) / (LasR, LuxR) after one
  // row_shl:4 and one swap, so  KGR*bR + KGS*bS = c1*v1^2 + c2*v2^2  with c1, c2 folded per lane (0 elsewhere)
  float c1, c2, c1x2, c2x2, ce, cm;  // ce = c*e, cm = c*(1-e):  c*P = ce + cm * kb/(1+kb)
  float m0, m6, m7, mo1, mo2;  // lane masks (1.0 / 0.0): species 0, 6, 7; observe: own state, state j+2
  float m0invK;
  float tm1, tm2;              // this lane's stage-time multipliers (lane j&3 evaluates the sigmoid of stage j&3)
  int js;                      // j & 3
};

constexpr int QP_SWAP23 = 0 | (1 << 2) | (3 << 4) | (2 << 6);  // quad_perm [0,1,3,2]

template <int VERSION>
struct DrLanes {
  using M = DrConstant<VERSION>;
  static constexpr int TPB = 32;  // trajectories per 256-thread block

  __device__ static float th(const OdeArgs& a, int slot, int i) { return a.theta[(size_t)a.slot_row[slot] * a.n + i]; }

  // slot tables per species j
  __device__ static int deg_slot(int j) {
    const int t[8] = {-1, M::S_drfp, M::S_dyfp, M::S_dcfp, -1, -1, M::S_dR, M::S_dS};
    return t[j];
  }
  __device__ static int a_slot(int j) {
    const int t[8] = {-1, -1, M::S_aYFP, M::S_aCFP, M::S_a530, M::S_a480, M::S_aR, M::S_aS};
    return t[j];
  }
  __device__ static int init_slot(int j) {
    const int t[8] = {M::SI + 0, M::SI + 1, M::SI + 2, M::SI + 3, -1, -1, M::SI + 4, M::SI + 5};
    return t[j];
  }

  // Hill fractions (dr_constant.py:58-73), one power term per lane.  v1: f = (a^n + b^n) / (1 + a + b)^n with
  // a = K6 c6, b = K12 c12 -> lanes 0-2 hold the LuxR terms (a, b, 1+a+b; exponent nR), lanes 3-5 the LasR terms.
  // v2: fR = c6^nR + (eR12 c12)^nR, fS = (eS6 c6)^nS + c12^nS -> lanes 0-3.  The accurate powf is ~150 instructions;
  // evaluated serially in every lane it was most of the kernel prologue (and, with its adjoint, of the epilogue).
  struct HillTerm {
    float base, n, pw;
  };
  __device__ static void hill(const OdeArgs& a, int i, int j, const float* c, HillTerm& H, float& fR, float& fS) {
    const float nR = clampf(th(a, M::S_nR, i), 0.5f, 3.f), nS = clampf(th(a, M::S_nS, i), 0.5f, 3.f);
    if (VERSION == 1) {
      const bool isR = j < 3;
      // (both rows read, then chosen: a lane-dependent slot index turns a.slot_row[] into a per-lane global load from the
      // kernel-argument segment, a memory round trip ahead of the row's own load)
      const float h0 = th(a, M::S_H0, i), h1 = th(a, M::S_H1, i), h2 = th(a, M::S_H2, i), h3 = th(a, M::S_H3, i);
      const float K6 = clampf(isR ? h0 : h2, 1e-12f, 1.f);
      const float K12 = clampf(isR ? h1 : h3, 1e-12f, 1.f);
      const float ta = K6 * c[0], tb = K12 * c[1];
      const int k = isR ? j : j - 3;
      H.base = j >= 6 ? 1.f : (k == 0 ? ta : (k == 1 ? tb : 1.f + ta + tb));
      H.n = j >= 6 ? 1.f : (isR ? nR : nS);
      H.pw = powf(H.base, H.n);
      fR = (bcast8<0>(H.pw) + bcast8<1>(H.pw)) / bcast8<2>(H.pw);
      fS = (bcast8<3>(H.pw) + bcast8<4>(H.pw)) / bcast8<5>(H.pw);
    } else {
      const float eS6 = clampf(th(a, M::S_H0, i), 1e-12f, 1.f), eR12 = clampf(th(a, M::S_H1, i), 1e-12f, 1.f);
      H.base = j == 0 ? c[0] : (j == 1 ? eR12 * c[1] : (j == 2 ? eS6 * c[0] : (j == 3 ? c[1] : 1.f)));
      H.n = j < 2 ? nR : (j < 4 ? nS : 1.f);
      H.pw = powf(H.base, H.n);
      fR = bcast8<0>(H.pw) + bcast8<1>(H.pw);
      fS = bcast8<2>(H.pw) + bcast8<3>(H.pw);
    }
  }
  // adjoint: every lane differentiates its own term (same formulas as pow_vjp), then the per-parameter sums are
  // gathered inside the 8-lane group.  Outputs are the raw-parameter adjoints (clamp pass-through applied).
  struct HillAdj {
    float nR, nS, H0, H1, H2, H3;
  };
  __device__ static HillAdj hill_vjp(const OdeArgs& a, int i, int j, const float* c, const HillTerm& H, float fRb,
                                     float fSb) {
    const float pass[6] = {clamp_pass(th(a, M::S_nR, i), 0.5f, 3.f), clamp_pass(th(a, M::S_nS, i), 0.5f, 3.f),
                           clamp_pass(th(a, M::S_H0, i), 1e-12f, 1.f), clamp_pass(th(a, M::S_H1, i), 1e-12f, 1.f),
                           VERSION == 1 ? clamp_pass(th(a, M::S_H0 + 2, i), 1e-12f, 1.f) : 0.f,
                           VERSION == 1 ? clamp_pass(th(a, M::S_H0 + 3, i), 1e-12f, 1.f) : 0.f};
    return hill_vjp_pass(j, c, H, fRb, fSb, pass);
  }
  // pass[] = where torch.clamp lets the gradient through (1 / 0) for nR, nS, H0..H3
  __device__ static HillAdj hill_vjp_pass(int j, const float* c, const HillTerm& H, float fRb, float fSb,
                                          const float* pass) {
    float g;
    if (VERSION == 1) {
      const float p0 = bcast8<0>(H.pw), p1 = bcast8<1>(H.pw), p2 = bcast8<2>(H.pw);
      const float p3 = bcast8<3>(H.pw), p4 = bcast8<4>(H.pw), p5 = bcast8<5>(H.pw);
      const float numbR = fRb / p2, dnbR = -fRb * (p0 + p1) / (p2 * p2);
      const float numbS = fSb / p5, dnbS = -fSb * (p3 + p4) / (p5 * p5);
      g = j < 2 ? numbR : (j == 2 ? dnbR : (j < 5 ? numbS : (j == 5 ? dnbS : 0.f)));
    } else {
      g = j < 2 ? fRb : (j < 4 ? fSb : 0.f);
    }
    // d/da a^n = n a^n / a and d/dn a^n = a^n ln a from the power already at hand (a > 0 by its clamps, dr_constant.py:58-73):
    // one reciprocal and one v_log_f32 where pow_vjp's powf + logf were ~250 instructions in every wavefront's epilogue
    const float ab = g * H.n * (H.pw * frcp(H.base));
    const float nb = g * (H.pw * (0.6931471805599453f * __builtin_amdgcn_logf(H.base)));
    HillAdj o;
    if (VERSION == 1) {
      o.nR = sum8(j < 3 ? nb : 0.f) * pass[0];
      o.nS = sum8((j >= 3 && j < 6) ? nb : 0.f) * pass[1];
      const float a0 = bcast8<0>(ab), a1 = bcast8<1>(ab), a2 = bcast8<2>(ab);
      const float a3 = bcast8<3>(ab), a4 = bcast8<4>(ab), a5 = bcast8<5>(ab);
      o.H0 = (a0 + a2) * c[0] * pass[2];
      o.H1 = (a1 + a2) * c[1] * pass[3];
      o.H2 = (a3 + a5) * c[0] * pass[4];
      o.H3 = (a4 + a5) * c[1] * pass[5];
    } else {
      o.nR = sum8(j < 2 ? nb : 0.f) * pass[0];
      o.nS = sum8((j >= 2 && j < 4) ? nb : 0.f) * pass[1];
      o.H0 = bcast8<2>(ab) * c[0] * pass[2];  // eS6
      o.H1 = bcast8<1>(ab) * c[1] * pass[3];  // eR12
      o.H2 = 0.f; o.H3 = 0.f;
    }
    return o;
  }

  template <int SOLVER>
  __device__ static void prepare(const OdeArgs& a, int i, int b, int j, DrLane& L, float* c, float& y0,
                                 HillTerm& H) {
    c[0] = clampf(expf(a.cond[b * a.C + 0]) - 1.f, 1e-12f, 1e6f);
    c[1] = clampf(expf(a.cond[b * a.C + 1]) - 1.f, 1e-12f, 1e6f);
    L.r = clampf(th(a, M::S_r, i), 0.f, 4.f);
    L.K = clampf(th(a, M::S_K, i), 0.f, 4.f);
    L.invK = frcp(L.K);
    L.tlag = th(a, M::S_tlag, i);
    L.rc = th(a, M::S_rc, i);
    hill(a, i, j, c, H, L.fR, L.fS);
    L.sgn = j == 0 ? 1.f : -1.f;
    const int ds = deg_slot(j);
    L.deg = ds < 0 ? 0.f : clampf(th(a, ds, i), 1e-12f, (j == 6 || j == 7) ? 5.f : 2.f);
    const int as = a_slot(j);
    L.a = j == 0 ? 0.f : (as < 0 ? 1.f : th(a, as, i));
    L.c = L.rc * L.a;
    if (j == 2) { L.e = th(a, M::S_e81, i); L.KGR = th(a, M::S_KGR81, i); L.KGS = th(a, M::S_KGS81, i); }
    else if (j == 3) { L.e = th(a, M::S_e76, i); L.KGR = th(a, M::S_KGR76, i); L.KGS = th(a, M::S_KGS76, i); }
    else { L.e = 1.f; L.KGR = 0.f; L.KGS = 0.f; }
    // lane 2: (v1, v2) = (LuxR, LasR); lane 3: (LasR, LuxR)
    L.c1 = j == 3 ? L.KGS * L.fS : L.KGR * L.fR;
    L.c2 = j == 3 ? L.KGR * L.fR : L.KGS * L.fS;
    if (j != 2 && j != 3) { L.c1 = 0.f; L.c2 = 0.f; }
    L.c1x2 = 2.f * L.c1; L.c2x2 = 2.f * L.c2;
    L.ce = L.c * L.e;
    L.cm = (j == 2 || j == 3) ? L.c * (1.f - L.e) : 0.f;
    L.prec = j < 4 ? th(a, M::NSLOT + j, i) : 1.f;
    L.m0 = j == 0 ? 1.f : 0.f; L.m6 = j == 6 ? 1.f : 0.f; L.m7 = j == 7 ? 1.f : 0.f;
    L.mo1 = (j >= 1 && j <= 3) ? 1.f : 0.f; L.mo2 = (j == 2 || j == 3) ? 1.f : 0.f;
    L.m0invK = L.m0 * L.invK;
    L.js = j & 3;
    // stage times (torchdiffeq fixed-grid tableaux; solvers.py:9-41 for the Heun variants)
    L.tm1 = 0.f; L.tm2 = 0.f;
    if (SOLVER == VIHDS_SOLVER_RK4) {         // t0, t0 + dt/3, t0 + 2 dt/3, t0 + dt
      L.tm1 = L.js == 1 ? 1.f : (L.js == 2 ? 2.f : 0.f);
      L.tm2 = L.js == 3 ? 1.f : 0.f;
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {  // t0, t0 + dt/2
      L.tm1 = L.js == 1 ? 0.5f : 0.f;
    }
    const int is = init_slot(j);
    y0 = is < 0 ? 0.f : th(a, is, i);
  }

  // ---- growth-rate sigmoid, one stage per lane ------------------------------------------------------------
  // gr(t) = r * sigmoid(4 (t - tlag)) does not depend on the state, so the (up to) four stage times of a step are
  // evaluated side by side in the four lanes of each quad and handed to the stages by a quad_perm broadcast:
  // one exp + one rcp per step instead of one per stage.
  struct Sig {
    float sig, gr;
  };
  template <int SOLVER>
  __device__ __forceinline__ static Sig stage_sigmoid(float t0, float t1, const DrLane& L) {
    float tj;
    if (SOLVER == VIHDS_SOLVER_RK4) {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f);
      tj = fmaf(L.tm2, dt, fmaf(L.tm1, d3, t0));
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      tj = fmaf(L.tm1, t1 - t0, t0);
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      tj = t0;
    } else {  // Heun: f(t0, .), f(t1, .)
      tj = L.js == 1 ? t1 : t0;
    }
    Sig s;
    s.sig = sigmoid_f(4.f * (tj - L.tlag));
    s.gr = L.r * s.sig;
    return s;
  }
  template <int STAGE>
  __device__ __forceinline__ static float stage_gr(const Sig& s) {
    return dpp_all<STAGE | (STAGE << 2) | (STAGE << 4) | (STAGE << 6)>(s.gr);
  }

  struct Eval {
    float x, v1, v2, b1, b2, rd, t, g, gr, coef;
  };
  // dy_j = c_j P_j + (s_j gamma - deg_j) y_j  with  c P = ce + cm * kb / (1 + kb),  kb = KGR bR + KGS bS
  // (the same promoter as dr_constant.py:88-95 with the numerator split off: (e + kb)/(1 + kb) = e + (1-e) kb/(1+kb))
  __device__ __forceinline__ static float rhs(float gr, float y, const DrLane& L, Eval& E) {
    E.x = bcast8<0>(y);
    E.v1 = dpp_mov<0x104, 0xf, 0x5>(0.f, y);  // row_shl:4 into lanes 0-3 of the group: lane 2 <- LuxR, lane 3 <- LasR
    E.v2 = dpp_all<QP_SWAP23>(E.v1);           // lane 2 <- LasR, lane 3 <- LuxR
    E.gr = gr;
    E.b1 = E.v1 * E.v1;
    E.b2 = E.v2 * E.v2;
    const float kb = fmaf(L.c2, E.b2, L.c1 * E.b1);
    E.rd = frcp(1.f + kb);
    E.t = kb * E.rd;
    E.g = fmaf(-L.invK, E.x, 1.f);
    E.coef = fmaf(L.sgn, gr * E.g, -L.deg);
    return fmaf(E.coef, y, fmaf(L.cm, E.t, L.ce));
  }
  __device__ __forceinline__ static float rhs(float gr, float y, const DrLane& L) {
    Eval E;
    return rhs(gr, y, L, E);
  }

  // Adjoint accumulators.  Everything that is linear in the stage adjoints is only summed here and mapped to the
  // parameters once, after the time loop (bwd kernel tail).
  struct Adj {
    float sv, svt;    // sum v, sum v*t          -> c
    float svr;        // sum v*(1-t) = sum v*rd  -> e  (accumulated directly: sv - svt cancels when the promoter saturates)
    float degb;       // -sum v*y                -> deg
    float c1b, c2b;   // sum kb_bar * v1^2, v2^2 -> KGR, KGS, fR, fS
    float gbx;        // sum gamma_bar*gr*x      -> K
    float rb, tl;     // per stage lane: sum gr_bar*sig, sum gr_bar*r*sig*(1-sig) -> r, tlag
  };
  // returns (d rhs/d y)^T v for this lane's state; accumulates parameter adjoints; grb = adjoint of this stage's gr
  __device__ __forceinline__ static float rhs_vjp(float y, const DrLane& L, float v, Adj& A, const Eval& E, float& grb) {
    float yb = v * E.coef;
    A.sv += v;
    A.svt = fmaf(v, E.t, A.svt);
    A.svr = fmaf(v, E.rd, A.svr);
    const float vy = v * y;
    A.degb -= vy;
    const float gamb = sum8(L.sgn * vy);
    // promoter (non-zero in lanes 2, 3 only): t = kb rd, d t / d kb = rd (1 - t)
    const float kbb = (v * L.cm) * fmaf(-E.t, E.rd, E.rd);
    A.c1b = fmaf(kbb, E.b1, A.c1b);
    A.c2b = fmaf(kbb, E.b2, A.c2b);
    const float v1b = (kbb * L.c1x2) * E.v1;
    const float v2b = (kbb * L.c2x2) * E.v2;
    const float tot = v1b + dpp_all<QP_SWAP23>(v2b);  // lane 2: -> LuxR, lane 3: -> LasR
    yb += dpp_mov<0x114, 0xf, 0xA>(0.f, tot);          // row_shr:4: lanes 6, 7 receive from lanes 2, 3
    const float gb = gamb * E.gr;
    yb = fmaf(-L.m0invK, gb, yb);
    A.gbx = fmaf(gb, E.x, A.gbx);
    grb = gamb * E.g;
    return yb;
  }
  // per-stage gr adjoints -> this lane's stage accumulators
  __device__ __forceinline__ static void stage_sigmoid_vjp(const Sig& s, const DrLane& L, float g1, float g2, float g3,
                                                           float g4, Adj& A) {
    const float gsel = L.js == 0 ? g1 : (L.js == 1 ? g2 : (L.js == 2 ? g3 : g4));
    A.rb = fmaf(gsel, s.sig, A.rb);
    A.tl = fmaf(gsel, s.gr * (1.f - s.sig), A.tl);
  }

  // number of RHS evaluations per step
  template <int SOLVER>
  __device__ __forceinline__ static float step(float t0, float t1, float h0, float y, const DrLane& L) {
    const Sig s = stage_sigmoid<SOLVER>(t0, t1, L);
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
      const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
      const float k1 = rhs(stage_gr<0>(s), y, L);
      const float k2 = rhs(stage_gr<1>(s), y + h * k1, L);
      return y + (0.5f * h) * (k1 + k2);
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      return y + (t1 - t0) * rhs(stage_gr<0>(s), y, L);
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      const float dt = t1 - t0;
      const float k1 = rhs(stage_gr<0>(s), y, L);
      return y + dt * rhs(stage_gr<1>(s), y + k1 * dt * 0.5f, L);
    } else {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f);
      const float k1 = rhs(stage_gr<0>(s), y, L);
      const float k2 = rhs(stage_gr<1>(s), y + d3 * k1, L);
      const float k3 = rhs(stage_gr<2>(s), y + (dt * k2 - d3 * k1), L);
      const float k4 = rhs(stage_gr<3>(s), y + dt * (k1 - k2 + k3), L);
      return y + (k1 + 3.f * k2 + 3.f * k3 + k4) * (dt * 0.125f);
    }
  }
  // lam: adjoint of y_{k+1} -> returns adjoint of y_k
  template <int SOLVER>
  __device__ __forceinline__ static float step_vjp(float t0, float t1, float h0, float y, const DrLane& L, float lam,
                                                   Adj& A) {
    const Sig s = stage_sigmoid<SOLVER>(t0, t1, L);
    float g1 = 0.f, g2 = 0.f, g3 = 0.f, g4 = 0.f;
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
      const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
      Eval E1, E2;
      const float k1 = rhs(stage_gr<0>(s), y, L, E1);
      const float ya = y + h * k1;
      rhs(stage_gr<1>(s), ya, L, E2);
      float v = 0.5f * h * lam;
      const float w = rhs_vjp(ya, L, v, A, E2, g2);
      lam += w;
      v += h * w;
      lam += rhs_vjp(y, L, v, A, E1, g1);
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      Eval E1;
      rhs(stage_gr<0>(s), y, L, E1);
      lam += rhs_vjp(y, L, (t1 - t0) * lam, A, E1, g1);
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      const float dt = t1 - t0;
      Eval E1, E2;
      const float k1 = rhs(stage_gr<0>(s), y, L, E1);
      const float ym = y + k1 * dt * 0.5f;
      rhs(stage_gr<1>(s), ym, L, E2);
      const float w = rhs_vjp(ym, L, dt * lam, A, E2, g2);
      lam += w;
      lam += rhs_vjp(y, L, 0.5f * dt * w, A, E1, g1);
    } else {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
      Eval E1, E2, E3, E4;  // stage intermediates of the forward recomputation, reused by the transposed stages
      const float k1 = rhs(stage_gr<0>(s), y, L, E1);
      const float y2 = y + d3 * k1;
      const float k2 = rhs(stage_gr<1>(s), y2, L, E2);
      const float y3 = y + (dt * k2 - d3 * k1);
      const float k3 = rhs(stage_gr<2>(s), y3, L, E3);
      const float y4 = y + dt * (k1 - k2 + k3);
      rhs(stage_gr<3>(s), y4, L, E4);
      const float k4b = d8 * lam;
      float k1b = k4b, k2b = 3.f * k4b, k3b = 3.f * k4b;
      float w = rhs_vjp(y4, L, k4b, A, E4, g4);
      lam += w; k1b += dt * w; k2b -= dt * w; k3b += dt * w;
      w = rhs_vjp(y3, L, k3b, A, E3, g3);
      lam += w; k1b -= d3 * w; k2b += dt * w;
      w = rhs_vjp(y2, L, k2b, A, E2, g2);
      lam += w; k1b += d3 * w;
      lam += rhs_vjp(y, L, k1b, A, E1, g1);
    }
    stage_sigmoid_vjp(s, L, g1, g2, g3, g4, A);
    return lam;
  }

  // observed signal of lanes 0..3 (reference vihds/ode.py:84-93): OD, OD*RFP, OD*(YFP+F530), OD*(CFP+F480)
  __device__ __forceinline__ static float observe(float y, float x, const DrLane& L, float& inner) {
    const float ysh = dpp_mov<0x102>(0.f, y);  // row_shl:2 -> state j+2
    inner = L.mo1 * y + L.mo2 * ysh + L.m0;    // species 0: inner = 1
    return x * inner;
  }
};

// LDS_IN: the block's time grid and observation rows are staged in LDS once ([T] + [nb][4][T] floats, nb = batch
// rows the block's 32 trajectories span).  The time loop then holds no vector-memory loads at all, so its stores
// are never waited for: with a global load in the loop every s_waitcnt vmcnt(0) for the load also waits for the
// step's trajectory / x_predict stores (measured ~80 ns of a ~410 ns rk4 step at B=36, S=200).
template <int VERSION, int SOLVER, bool LDS_IN>
__global__ void __launch_bounds__(256) dr_lane_fwd_kernel(OdeArgs a) {
  using D = DrLanes<VERSION>;
  extern __shared__ float lds[];
  const int tl = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int i0 = blockIdx.x * D::TPB + tl;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  int ob_off = 0;
  if (LDS_IN) {
    const int first = blockIdx.x * D::TPB, last = min(first + D::TPB, a.n) - 1;
    const int b0 = first / a.S, nb = last / a.S - b0 + 1;
    for (int q = threadIdx.x; q < a.T; q += 256) lds[q] = a.times[q];
    const float* src = a.obs + (size_t)b0 * 4 * a.T;
    for (int q = threadIdx.x; q < nb * 4 * a.T; q += 256) lds[a.T + q] = src[q];
    __syncthreads();
    ob_off = a.T + ((b - b0) * 4 + (j & 3)) * a.T;
  }
  const float* ob = a.obs + ((size_t)b * 4 + (j & 3)) * a.T;
  auto time_at = [&](int k) { return LDS_IN ? lds[k] : a.times[k]; };
  auto obs_at = [&](int k) { return LDS_IN ? lds[ob_off + k] : ob[k]; };
  DrLane L;
  float c[2], y;
  typename D::HillTerm H;
  D::template prepare<SOLVER>(a, i, b, j, L, c, y, H);
  const float lc = LOG2PI_F - logf(L.prec);
  float lp = 0.f;
  const float h0 = a.times[1] - a.times[0];
  const size_t n = a.n;
  // times and observations are fetched one step ahead so that their latency is off the dependent chain
  const bool want_lp = a.logp && j < 4;
  float tA = time_at(0), tB = time_at(1);
  float ob_cur = want_lp ? obs_at(0) : 0.f;
  for (int k = 0; k < a.T; ++k) {
    const float tC = (k + 1 < a.T) ? time_at(k + 1) : tB;
    const float ob_next = (want_lp && k + 1 < a.T) ? obs_at(k + 1) : 0.f;
    if (k > 0) {
      y = D::template step<SOLVER>(tA, tB, h0, y, L);
      tA = tB;
    }
    tB = tC;
    if (a.traj && live) a.traj[((size_t)k * 8 + j) * n + i] = y;
    float inner;
    const float xp = D::observe(y, bcast8<0>(y), L, inner);
    if (j < 4) {
      if (a.xpred && live) a.xpred[((size_t)k * 4 + j) * n + i] = xp;
      const float e = xp - ob_cur;
      lp += -0.5f * (lc + L.prec * e * e);
    }
    ob_cur = ob_next;
  }
  if (a.logp && live && j < 4) a.logp[(size_t)j * n + i] = lp;
}

// The adjoint kernels' epilogue: raw accumulators -> gradients of the theta rows (each row written by exactly one lane).
template <int VERSION>
__device__ __forceinline__ void dr_lane_write_adjoints(const OdeArgs& a, int i, int j, bool live, const DrLane& L,
                                                       const float* c,
                                                       const typename DrLanes<VERSION>::HillTerm& H,
                                                       const typename DrLanes<VERSION>::Adj& A, float lam, float precb) {
  using D = DrLanes<VERSION>;
  using M = DrConstant<VERSION>;
  const size_t n = a.n;
  // ---- parameter adjoints -> theta rows (each slot row written by exactly one lane)
  // c enters as ce = c e and cm = c (1 - e):  c_bar = sv e + svt (1 - e),  e_bar = c (sv - svt) = c sum v (1 - t)
  const bool prom = j == 2 || j == 3;
  const float cb = prom ? A.sv * L.e + A.svt * (1.f - L.e) : A.sv;
  const float eb = L.c * A.svr;
  // c1 = K1 f1, c2 = K2 f2 with (K1, f1, K2, f2) = (KGR, fR, KGS, fS) in lane 2 and (KGS, fS, KGR, fR) in lane 3
  const float KGRb = j == 3 ? A.c2b * L.fR : A.c1b * L.fR;
  const float KGSb = j == 3 ? A.c1b * L.fS : A.c2b * L.fS;
  const float fRb = sum8(prom ? (j == 3 ? A.c2b : A.c1b) * L.KGR : 0.f);
  const float fSb = sum8(prom ? (j == 3 ? A.c1b : A.c2b) * L.KGS : 0.f);
  const float rcb = sum8(cb * L.a);
  const float rb = sum4(A.rb), tlagb = -4.f * sum4(A.tl);
  const float Kb = A.gbx * L.invK * L.invK;
  const typename D::HillAdj HA = D::hill_vjp(a, i, j, c, H, fRb, fSb);
  if (!live) return;
  auto put = [&](int slot, float v) { a.g_theta[(size_t)a.slot_row[slot] * n + i] = v; };
  auto raw = [&](int slot) { return a.theta[(size_t)a.slot_row[slot] * n + i]; };
  const int is = D::init_slot(j);
  if (is >= 0) put(is, lam);
  if (j < 4) put(M::NSLOT + j, precb);
  const int ds = D::deg_slot(j);
  if (ds >= 0) put(ds, A.degb * clamp_pass(raw(ds), 1e-12f, (j == 6 || j == 7) ? 5.f : 2.f));
  const int as = D::a_slot(j);
  if (as >= 0) put(as, cb * L.rc);
  if (j == 2) { put(M::S_e81, eb); put(M::S_KGR81, KGRb); put(M::S_KGS81, KGSb); }
  if (j == 3) { put(M::S_e76, eb); put(M::S_KGR76, KGRb); put(M::S_KGS76, KGSb); }
  if (j == 0) {
    put(M::S_r, rb * clamp_pass(raw(M::S_r), 0.f, 4.f));
    put(M::S_K, Kb * clamp_pass(raw(M::S_K), 0.f, 4.f));
    put(M::S_tlag, tlagb);
    put(M::S_rc, rcb);
    put(M::S_nR, HA.nR); put(M::S_nS, HA.nS); put(M::S_H0, HA.H0); put(M::S_H1, HA.H1);
    if (VERSION == 1) { put(M::S_H2, HA.H2); put(M::S_H3, HA.H3); }
  }
}

template <int VERSION, int SOLVER, bool LDS_IN>
__global__ void __launch_bounds__(256) dr_lane_bwd_kernel(OdeArgs a) {
  using D = DrLanes<VERSION>;
  extern __shared__ float lds[];
  const int tl = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int i0 = blockIdx.x * D::TPB + tl;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  int ob_off = 0;
  if (LDS_IN) {  // time grid + observation rows of this block in LDS (as in the forward kernel)
    const int first = blockIdx.x * D::TPB, last = min(first + D::TPB, a.n) - 1;
    const int b0 = first / a.S, nb = last / a.S - b0 + 1;
    for (int q = threadIdx.x; q < a.T; q += 256) lds[q] = a.times[q];
    const float* src = a.obs + (size_t)b0 * 4 * a.T;
    for (int q = threadIdx.x; q < nb * 4 * a.T; q += 256) lds[a.T + q] = src[q];
    __syncthreads();
    ob_off = a.T + ((b - b0) * 4 + (j & 3)) * a.T;
  }
  DrLane L;
  float c[2], y0;
  typename D::HillTerm H;
  D::template prepare<SOLVER>(a, i, b, j, L, c, y0, H);
  typename D::Adj A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float lam = 0.f, precb = 0.f;
  const size_t n = a.n;
  const float w_iw = a.iw_logp ? iw_wave_weight(a, i, b) : 0.f;
  const float glp = j < 4 ? ode_logp_grad(a, w_iw, i, j) : 0.f;
  const float* ob = a.obs + ((size_t)b * 4 + (j & 3)) * a.T;
  auto time_at = [&](int k) { return LDS_IN ? lds[k] : a.times[k]; };
  auto obs_at = [&](int k) { return LDS_IN ? lds[ob_off + k] : ob[k]; };
  const float h0 = a.times[1] - a.times[0];

  // The stored state is requested TWO steps ahead (a step lasts ~0.65 us; a read that misses L2 takes longer than
  // that to come back from HBM / the infinity cache), observations, upstream gradients and times one step ahead.
  auto traj_at = [&](int k) { return a.traj_in[((size_t)max(k, 0) * 8 + j) * n + i]; };
  float y_cur = traj_at(a.T - 1), y_n1 = traj_at(a.T - 2);
  float ob_next = j < 4 ? obs_at(a.T - 1) : 0.f;
  float gx_next = (a.g_xpred && j < 4) ? a.g_xpred[((size_t)(a.T - 1) * 4 + j) * n + i] : 0.f;
  float gt_next = a.g_traj ? a.g_traj[((size_t)(a.T - 1) * 8 + j) * n + i] : 0.f;
  float tHi = time_at(a.T - 1), tLo = tHi;  // step k uses (times[k], times[k+1]) = (tLo, tHi)
  for (int k = a.T - 1; k >= 0; --k) {
    const float y = y_cur, obk = ob_next, gxk = gx_next, gtk = gt_next;
    const float tK = tLo;  // times[k]
    y_cur = y_n1;
    if (k > 0) {
      ob_next = j < 4 ? obs_at(k - 1) : 0.f;
      if (a.g_xpred && j < 4) gx_next = a.g_xpred[((size_t)(k - 1) * 4 + j) * n + i];
      if (a.g_traj) gt_next = a.g_traj[((size_t)(k - 1) * 8 + j) * n + i];
      tLo = time_at(k - 1);
    }
    if (k > 1) y_n1 = traj_at(k - 2);  // (issued last: the youngest vector load, the only one left in flight)
    if (k < a.T - 1) lam = D::template step_vjp<SOLVER>(tK, tHi, h0, y, L, lam, A);
    tHi = tK;
    // injection at time k
    const float x = bcast8<0>(y);
    float inner;
    const float xp = D::observe(y, x, L, inner);
    float xpb = 0.f;
    if (j < 4) {
      const float e = xp - obk;
      xpb = -glp * L.prec * e + gxk;
      precb += glp * (0.5f / L.prec - 0.5f * e * e);
    }
    // xp_j = x * inner_j : d/dx -> lane 0 (sum over the quad), d/dy_j and d/dy_{j+2}
    const float q = xpb * x;
    lam += L.mo1 * q + dpp_mov<0x112>(0.f, L.mo2 * q);  // row_shr:2: lanes 4,5 receive from lanes 2,3
    const float xsum = sum4(xpb * inner);
    lam += L.m0 * xsum + gtk;
  }
  dr_lane_write_adjoints<VERSION>(a, i, j, live, L, c, H, A, lam, precb);
}

// ---------------------------------------------------------------------------------------------------------------
// (The lane-split TRAINING kernel -- forward sweep with the states in LDS, unit-weight adjoint behind it, the sampling stage in
// front: rounds 1-2's decoder launch -- was removed in round 6: the time-parallel kernel of vihds_dr_scan.hpp serves every
// shape it served except time grids of 130-150 points, which take vihds_ode_fwd + vihds_ode_bwd.)
// What the decoder launch's generator bookkeeping needs:
struct RngTickets {
  unsigned int u, c;  // this block's tickets of the two generators
};
// the holder of a launch's last ticket advances the generator's step (at the very end of the kernel)
__device__ __forceinline__ void rng_advance(unsigned int* rng, unsigned int ticket, int thread) {
  if (rng && threadIdx.x == thread && ticket == gridDim.x - 1) {
    rng[2] = rng[2] + 1u;
    rng[3] = 0u;
  }
}

// LDS floats the forward kernel stages per block: the time grid + the observation rows of the batch rows one block spans
inline size_t dr_lane_fwd_lds_bytes(const OdeArgs& a, int tpb) {
  const int nb = min(a.B, (tpb - 1) / a.S + 2);
  return ((size_t)a.T + (size_t)nb * 4 * a.T) * sizeof(float);
}

template <int VERSION>
inline int launch_dr_lanes(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  const dim3 grid((a.n + DrLanes<VERSION>::TPB - 1) / DrLanes<VERSION>::TPB), block(256);
  const size_t lds = dr_lane_fwd_lds_bytes(a, DrLanes<VERSION>::TPB);
  const bool lds_in = lds <= 48 * 1024;  // long grids / many rows per block fall back to global loads in the loop
#define VIHDS_LCASE(SV)                                                                                   \
  case SV:                                                                                                \
    if (backward && lds_in) hipLaunchKernelGGL((dr_lane_bwd_kernel<VERSION, SV, true>), grid, block, lds, st, a); \
    else if (backward) hipLaunchKernelGGL((dr_lane_bwd_kernel<VERSION, SV, false>), grid, block, 0, st, a); \
    else if (lds_in) hipLaunchKernelGGL((dr_lane_fwd_kernel<VERSION, SV, true>), grid, block, lds, st, a); \
    else hipLaunchKernelGGL((dr_lane_fwd_kernel<VERSION, SV, false>), grid, block, 0, st, a);             \
    return VIHDS_OK;
  switch (solver) {
    VIHDS_LCASE(VIHDS_SOLVER_MODEULER)
    VIHDS_LCASE(VIHDS_SOLVER_MODEULERWHILE)
    VIHDS_LCASE(VIHDS_SOLVER_EULER)
    VIHDS_LCASE(VIHDS_SOLVER_MIDPOINT)
    VIHDS_LCASE(VIHDS_SOLVER_RK4)
  }
#undef VIHDS_LCASE
  return VIHDS_E_BADARG;
}

}  // namespace vihds
