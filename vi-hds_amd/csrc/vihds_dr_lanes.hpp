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
// Training step in ONE launch: log-likelihood AND the adjoint for a unit upstream gradient.
//
// In the ELBO, d loss / d logp[j][b][s] is the same number w[b][s] for the four observed signals (the IWAE softmax
// weight), and the adjoint is linear in it.  So the adjoint for w = 1 can be computed right behind the forward sweep,
// before the weights exist; the caller scales the resulting theta gradient by w[b][s] afterwards (one elementwise
// pass).  The forward sweep then has nobody to write the trajectory for: the states of the block's 32 trajectories
// stay in LDS ([T][256] floats = 86 KB at T = 86) and the reverse sweep reads them from there.  Compared with
// dr_lane_fwd_kernel + dr_lane_bwd_kernel: one launch and one prologue instead of two, no trajectory / x_predict
// stores (29.7 MB) and no trajectory reads (19.8 MB).  Outputs: logp [4][n] and the unit-weight gradient g_theta.
// time grid and the observation rows of the batch rows this block spans -> LDS (no barrier: the caller's next one covers it)
__device__ __forceinline__ void dr_lane_stage_inputs(const OdeArgs& a, int tpb, float* lds) {
  const int first = blockIdx.x * tpb, last = min(first + tpb, a.n) - 1;
  const int b0 = first / a.S, nb = last / a.S - b0 + 1;
  for (int q = threadIdx.x; q < a.T; q += 256) lds[q] = a.times[q];
  const float* src = a.obs + (size_t)b0 * 4 * a.T;
  for (int q = threadIdx.x; q < nb * 4 * a.T; q += 256) lds[a.T + q] = src[q];
}
template <int VERSION, int SOLVER, bool INPUTS_STAGED = false>
__device__ __forceinline__ void dr_lane_train_body(const OdeArgs& a, int nb_max, float* lds) {
  using D = DrLanes<VERSION>;
  // lds: [T] times | [nb_max][4][T] observations | [T][256] states
  const int tl = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int i0 = blockIdx.x * D::TPB + tl;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int b0 = (blockIdx.x * D::TPB) / a.S;
  if (!INPUTS_STAGED) {
    dr_lane_stage_inputs(a, D::TPB, lds);
    __syncthreads();
  }
  const float* tm = lds;
  const float* ob = lds + a.T + ((b - b0) * 4 + (j & 3)) * a.T;
  float* ys = lds + a.T + (size_t)nb_max * 4 * a.T + threadIdx.x;  // this lane's column, stride 256
  DrLane L;
  float c[2], y;
  typename D::HillTerm H;
  D::template prepare<SOLVER>(a, i, b, j, L, c, y, H);
  const size_t n = a.n;
  const float h0 = tm[1] - tm[0];
  // ---- forward sweep: states to LDS, log-likelihood accumulated
  {
    const float lc = LOG2PI_F - logf(L.prec);
    float lp = 0.f;
    float tA = tm[0], tB = tm[1];
    float ob_cur = j < 4 ? ob[0] : 0.f;
    for (int k = 0; k < a.T; ++k) {
      const float tC = (k + 1 < a.T) ? tm[k + 1] : tB;
      const float ob_next = (j < 4 && k + 1 < a.T) ? ob[k + 1] : 0.f;
      if (k > 0) {
        y = D::template step<SOLVER>(tA, tB, h0, y, L);
        tA = tB;
      }
      tB = tC;
      ys[(size_t)k * 256] = y;
      float inner;
      const float xp = D::observe(y, bcast8<0>(y), L, inner);
      if (j < 4) {
        const float e = xp - ob_cur;
        lp += -0.5f * (lc + L.prec * e * e);
      }
      ob_cur = ob_next;
    }
    if (a.logp && live && j < 4) a.logp[(size_t)j * n + i] = lp;
  }
  // ---- reverse sweep with unit weight on the four log-likelihoods
  typename D::Adj A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float lam = 0.f, precb = 0.f;
  const float glp = j < 4 ? 1.f : 0.f;
  float y_next = y;  // = state at T-1
  float ob_next = j < 4 ? ob[a.T - 1] : 0.f;
  float tHi = tm[a.T - 1], tLo = tHi;
  for (int k = a.T - 1; k >= 0; --k) {
    const float yk = y_next, obk = ob_next;
    const float tK = tLo;
    if (k > 0) {
      y_next = ys[(size_t)(k - 1) * 256];
      ob_next = j < 4 ? ob[k - 1] : 0.f;
      tLo = tm[k - 1];
    }
    if (k < a.T - 1) lam = D::template step_vjp<SOLVER>(tK, tHi, h0, yk, L, lam, A);
    tHi = tK;
    const float x = bcast8<0>(yk);
    float inner;
    const float xp = D::observe(yk, x, L, inner);
    float xpb = 0.f;
    if (j < 4) {
      const float e = xp - obk;
      xpb = -glp * L.prec * e;
      precb += glp * (0.5f / L.prec - 0.5f * e * e);
    }
    const float q = xpb * x;
    lam += L.mo1 * q + dpp_mov<0x112>(0.f, L.mo2 * q);
    const float xsum = sum4(xpb * inner);
    lam += L.m0 * xsum;
  }
  dr_lane_write_adjoints<VERSION>(a, i, j, live, L, c, H, A, lam, precb);
}


// The sampling stage of the decoder step, run by the same block before its sweeps (vihds_theta_ode_logp_grad):
// theta = clip(sample(q, u)) with log q / log p for the block's 32 trajectories (the arithmetic of theta_fwd_lds_kernel;
// here the 8 lanes of a trajectory own parameter blocks j, j+8, ...), then the device-conditioner rows (the
// arithmetic of device_condition_kernel).  Writes theta / u / log_q / log_p to global memory; the sweeps read theta
// back after the barrier.  `scratch` (dr_lane_theta_stage_floats floats of LDS) holds the per-(row, parameter)
// constants and the conditioner's tables.  (Routing theta / u through LDS tiles -- coalesced stores, sweeps reading
// theta from LDS -- was measured slower: 100.7 vs 96.0 us.)
// LDS floats the stage needs behind the time grid / observation rows
__host__ __device__ inline size_t dr_lane_theta_stage_floats(int nb_max, int P, int E, int D, int B) {
  return (size_t)10 * nb_max * P + (size_t)2 * E * D + (size_t)B * D;
}
struct RngTickets {
  unsigned int u, c;  // this block's tickets of the two generators (valid in threads 0 and 64)
};
__device__ __forceinline__ RngTickets dr_lane_theta_stage(const OdeArgs& a, const ThetaStageArgs& t, int nb_max,
                                                          float* scratch) {
  constexpr int TPB = 32;
  constexpr float LOG2PI = 1.8378770664093453f;
  const int n = a.n, P = t.P, B = a.B, S = a.S;
  const int first = blockIdx.x * TPB, last = min(first + TPB, n) - 1;
  const int b0 = first / S, nb = last / S - b0 + 1;
  const int stride = nb_max * P;
  float* t_kind = scratch;
  float* t_mu = scratch + stride;
  float* t_sigma = scratch + 2 * stride;
  float* t_prec = scratch + 3 * stride;
  float* t_cq = scratch + 4 * stride;
  float* t_lo = scratch + 5 * stride;
  float* t_hi = scratch + 6 * stride;
  float* t_pmu = scratch + 7 * stride;
  float* t_cp = scratch + 8 * stride;
  float* t_pprec = scratch + 9 * stride;
  float* t_cw = scratch + 10 * stride;      // [E*D] conditioner weights of this call
  float* t_rel = t_cw + t.E * a.D;          // [E*D] relevance masks
  float* t_dev = t_rel + t.E * a.D;         // [B*D] device one-hot rows (the conditioner's tiling reads any row)
  const int tl = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int i0 = first + tl;
  const bool live = i0 < n;
  const int i = live ? i0 : n - 1;
  const int b = i / S;
  const int row = (b - b0) * P;
  // -- table loads are issued first; the draws below need none of them and run while they are in flight
  const int e1 = threadIdx.x;
  const bool filler = e1 < nb * P;
  int f_kd = 0, f_p = 0;
  float f_pr = 1.f, f_mu = 0.f, f_lo = 0.f, f_hi = 0.f, f_pmu = 0.f, f_pp = 1.f;
  if (filler) {
    const int bb = e1 / P;
    f_p = e1 - bb * P;
    const int rm = t.q_rows ? t.q_rows[f_p] : f_p, rp = t.q_rows ? t.q_rows[P + f_p] : f_p;
    f_kd = t.kind[f_p];
    f_pr = t.q_prec[rp * B + b0 + bb];
    f_mu = t.q_mu[rm * B + b0 + bb];
    f_lo = t.clip_lo[f_p];
    f_hi = t.clip_hi[f_p];
    f_pmu = t.p_mu[f_p];
    f_pp = t.p_prec[f_p];
  }
  if (t.E > 0) {  // conditioner tables: waves 2, 3 (the fillers sit in waves 0, 1)
    const int ed = t.E * a.D;
    for (int e = threadIdx.x - 128; e >= 0 && e < ed; e += 128) {
      float zz;
      if (t.crng) zz = philox_normal((unsigned int)e, 0xC04Du, t.crng[2], 0u, t.crng[0], t.crng[1], 0);
      else zz = t.z[e];
      t_cw[e] = t.w_mean + t.w_std * zz;
      t_rel[e] = t.rel[e];
    }
    for (int e = threadIdx.x; e < B * a.D; e += 256) t_dev[e] = a.dev1hot[e];
  }

  unsigned int k0 = 0, k1 = 0, step = 0, gidx = 0;
  if (t.rng) {
    k0 = t.rng[0]; k1 = t.rng[1]; step = t.rng[2];
    gidx = (unsigned int)(b * t.S_total + t.s_off + (i - b * S));
  }
  // this lane's parameter blocks are j, j + 8, (j + 16, ...): the first two are drawn ahead of the barrier
  float za[4] = {0.f, 0.f, 0.f, 0.f}, zb[4] = {0.f, 0.f, 0.f, 0.f};
  if (t.rng) {
    if (4 * j < P) philox_normal4(gidx, (unsigned int)j, step, 0u, k0, k1, za);
    if (4 * (j + 8) < P) philox_normal4(gidx, (unsigned int)(j + 8), step, 0u, k0, k1, zb);
  }
  for (int e = e1; e < nb * P; e += 256) {
    if (e != e1) {  // (more than 256 table entries: the remaining ones the plain way)
      const int bb = e / P;
      f_p = e - bb * P;
      const int rm = t.q_rows ? t.q_rows[f_p] : f_p, rp = t.q_rows ? t.q_rows[P + f_p] : f_p;
      f_kd = t.kind[f_p];
      f_pr = t.q_prec[rp * B + b0 + bb];
      f_mu = t.q_mu[rm * B + b0 + bb];
      f_lo = t.clip_lo[f_p]; f_hi = t.clip_hi[f_p]; f_pmu = t.p_mu[f_p]; f_pp = t.p_prec[f_p];
    }
    const float prec = (f_kd == KIND_CONSTANT) ? 1.f : (t.prec_is_log ? expf(f_pr) : f_pr);
    t_kind[e] = (float)f_kd;
    t_mu[e] = f_mu;
    t_sigma[e] = 1.f / sqrtf(prec);
    t_prec[e] = prec;
    t_cq[e] = -LOG2PI + 0.5f * logf(prec + 1e-12f);
    t_lo[e] = f_lo;
    t_hi[e] = f_hi;
    t_pmu[e] = f_pmu;
    t_cp[e] = -LOG2PI + 0.5f * logf(f_pp + 1e-12f);
    t_pprec[e] = f_pp;
  }
  __syncthreads();
  // Every thread of this block has read the generators' step counters by now (the loads were complete at the barrier),
  // so the block takes its tickets here and the atomics' round trips hide behind the sweeps; the last ticket holder
  // advances the step at the very end of the kernel (rng_advance), when every other block is past this point too.
  RngTickets tk = {0u, 0u};
  if (t.rng && threadIdx.x == 0) tk.u = atomicAdd(&t.rng[3], 1u);
  if (t.crng && threadIdx.x == 64) tk.c = atomicAdd(&t.crng[3], 1u);
  float lq = 0.f, lp = 0.f;
  int it = 0;
  for (int kb = j; 4 * kb < P; kb += 8, ++it) {
    float z4[4];
    if (t.rng) {
      if (it == 0) { z4[0] = za[0]; z4[1] = za[1]; z4[2] = za[2]; z4[3] = za[3]; }
      else if (it == 1) { z4[0] = zb[0]; z4[1] = zb[1]; z4[2] = zb[2]; z4[3] = zb[3]; }
      else philox_normal4(gidx, (unsigned int)kb, step, 0u, k0, k1, z4);
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int p = 4 * kb + jj;
      if (p >= P) break;
      float uu;
      if (t.rng) {
        uu = z4[jj];
        if (live) t.u[(size_t)i * P + p] = uu;
      } else {
        uu = t.u[(size_t)i * P + p];
      }
      // straight-line: the 8 lanes of a trajectory hold parameters of different kinds, so a branch per kind would
      // run every side anyway (measured: this loop was 3.5 us of the launch with branches)
      const int e = row + p;
      const float kdf = t_kind[e], mu = t_mu[e];
      const bool cst = kdf == (float)KIND_CONSTANT, ln = kdf == (float)KIND_LOGNORMAL;
      const float zz = mu + t_sigma[e] * uu;
      float x = ln ? expf(zz) : zz;
      const float lo = t_lo[e], hi = t_hi[e];
      x = x < lo ? lo : (x > hi ? hi : x);
      const float v = ln ? logf(x + 1e-12f) : x;
      const float jac = ln ? v : 0.f;
      const float dq = mu - v, dp = t_pmu[e] - v;
      const float tq = t_cq[e] - 0.5f * t_prec[e] * dq * dq - jac;
      const float tp = t_cp[e] - 0.5f * t_pprec[e] * dp * dp - jac;
      lq += cst ? 0.f : tq;
      lp += cst ? 0.f : tp;
      x = cst ? 0.f * uu + mu : x;
      if (live) t.theta[(size_t)p * n + i] = x;
    }
  }
  lq = sum8(lq);
  lp = sum8(lp);
  if (live && j == 0) {
    if (t.log_q) t.log_q[i] = lq;
    if (t.log_p) t.log_p[i] = lp;
  }
  // device conditioner: lane e of the trajectory's group produces row cond_row0 + e.  Weights, masks and one-hot rows
  // come from LDS (as global loads inside this data-dependent loop they were serialised: ~6 us of the launch)
  if (t.E > 0) {
    const int r = (int)(((long long)b * t.S_total + t.s_off + (i - b * S)) % B);
    for (int e = j; e < t.E; e += 8) {
      float c = 0.f;
      for (int d = 0; d < a.D; ++d) {
        const float hot = t_dev[r * a.D + d] * t_rel[e * a.D + d];
        c += t_cw[e * a.D + d] * hot;  // (hot == 0 contributes exactly 0, as when the weight is not drawn for it)
      }
      c = fmaxf(c, 0.f);
      if (live) t.theta[(size_t)(t.cond_row0 + e) * n + i] = (t.is_default[e] ? 1.f : 0.f) + c;
    }
  }
  __syncthreads();  // theta of this block's trajectories is in memory; the scratch region is free again
  return tk;
}
// the holder of the last ticket advances a generator's step: every block has read it before taking its ticket
__device__ __forceinline__ void rng_advance(unsigned int* rng, unsigned int ticket, int thread) {
  if (rng && threadIdx.x == thread && ticket == gridDim.x - 1) {
    rng[2] = rng[2] + 1u;
    rng[3] = 0u;
  }
}

template <int VERSION, int SOLVER>
__global__ void __launch_bounds__(256) dr_lane_train_kernel(OdeArgs a, int nb_max) {
  extern __shared__ float lds[];
  dr_lane_train_body<VERSION, SOLVER>(a, nb_max, lds);
}
// sampling stage + conditioner + sweeps: the whole decoder side of a training step
template <int VERSION, int SOLVER>
__global__ void __launch_bounds__(256) dr_lane_train_theta_kernel(OdeArgs a, int nb_max, ThetaStageArgs t) {
  extern __shared__ float lds[];
  dr_lane_stage_inputs(a, DrLanes<VERSION>::TPB, lds);  // in flight during the sampling stage, whose barriers cover it
  // scratch = the (still unused) states region
  const RngTickets tk = dr_lane_theta_stage(a, t, nb_max, lds + a.T + (size_t)nb_max * 4 * a.T);
  dr_lane_train_body<VERSION, SOLVER, true>(a, nb_max, lds);
  rng_advance(t.rng, tk.u, 0);
  rng_advance(t.crng, tk.c, 64);
}

inline size_t dr_lane_train_lds_bytes(const OdeArgs& a, int tpb, int* nb_max_out) {
  const int nb = min(a.B, (tpb - 1) / a.S + 2);
  if (nb_max_out) *nb_max_out = nb;
  return ((size_t)a.T + (size_t)nb * 4 * a.T + (size_t)a.T * 256) * sizeof(float);
}
constexpr size_t DR_LANE_TRAIN_MAX_LDS = 160 * 1024;

// returns VIHDS_E_UNSUPPORTED when the states of a block do not fit in LDS (long time grids)
template <int VERSION>
inline int launch_dr_lane_train(int solver, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts = nullptr) {
  int nb_max = 0;
  const size_t lds = dr_lane_train_lds_bytes(a, DrLanes<VERSION>::TPB, &nb_max);
  if (lds > DR_LANE_TRAIN_MAX_LDS) return VIHDS_E_UNSUPPORTED;
  // The states in LDS allow one block per CU: beyond one block per CU the launch runs in rounds and the forward +
  // adjoint pair (several waves per SIMD, trajectory through HBM) is faster (measured: 163 vs 145 us at 14 400
  // trajectories, 84 vs 99 us at 7 200).
  static const int n_cu = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      v = 256;
    return v;
  }();
  if ((a.n + DrLanes<VERSION>::TPB - 1) / DrLanes<VERSION>::TPB > n_cu) return VIHDS_E_UNSUPPORTED;
  if (ts && dr_lane_theta_stage_floats(nb_max, ts->P, ts->E, a.D, a.B) > (size_t)a.T * 256)
    return VIHDS_E_UNSUPPORTED;  // stage scratch must fit
  const dim3 grid((a.n + DrLanes<VERSION>::TPB - 1) / DrLanes<VERSION>::TPB), block(256);
#define VIHDS_TCASE(SV)                                                                                         \
  case SV: {                                                                                                    \
    auto kern = dr_lane_train_kernel<VERSION, SV>;                                                              \
    auto kern_t = dr_lane_train_theta_kernel<VERSION, SV>;                                                      \
    static size_t allowed = 64 * 1024, allowed_t = 64 * 1024; /* dynamic LDS opted in to so far, per kernel */  \
    size_t& have = ts ? allowed_t : allowed;                                                                    \
    if (lds > have) {                                                                                           \
      const void* f = ts ? reinterpret_cast<const void*>(kern_t) : reinterpret_cast<const void*>(kern);         \
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)DR_LANE_TRAIN_MAX_LDS) !=     \
          hipSuccess)                                                                                           \
        return VIHDS_E_HIP;                                                                                     \
      have = DR_LANE_TRAIN_MAX_LDS;                                                                             \
    }                                                                                                           \
    if (ts) hipLaunchKernelGGL(kern_t, grid, block, lds, st, a, nb_max, *ts);                                   \
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a, nb_max);                                             \
    return VIHDS_OK;                                                                                            \
  }
  switch (solver) {
    VIHDS_TCASE(VIHDS_SOLVER_MODEULER)
    VIHDS_TCASE(VIHDS_SOLVER_MODEULERWHILE)
    VIHDS_TCASE(VIHDS_SOLVER_EULER)
    VIHDS_TCASE(VIHDS_SOLVER_MIDPOINT)
    VIHDS_TCASE(VIHDS_SOLVER_RK4)
  }
#undef VIHDS_TCASE
  return VIHDS_E_BADARG;
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
