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
  float m0, m6, m7, mo1, mo2;  // lane masks (1.0 / 0.0): species 0, 6, 7; observe: own state, state j+2
};

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

  __device__ static void hill(const OdeArgs& a, int i, const float* c, float& fR, float& fS) {
    float thh[M::NSLOT];
    thh[M::S_nR] = th(a, M::S_nR, i); thh[M::S_nS] = th(a, M::S_nS, i);
    thh[M::S_H0] = th(a, M::S_H0, i); thh[M::S_H1] = th(a, M::S_H1, i);
    if (VERSION == 1) {
      thh[M::S_H2] = th(a, M::S_H2, i); thh[M::S_H3] = th(a, M::S_H3, i);
      fR = hill_frac(thh[M::S_nR], thh[M::S_H0], thh[M::S_H1], c[0], c[1]);
      fS = hill_frac(thh[M::S_nS], thh[M::S_H2], thh[M::S_H3], c[0], c[1]);
    } else {
      const float nR = clampf(thh[M::S_nR], 0.5f, 3.f), nS = clampf(thh[M::S_nS], 0.5f, 3.f);
      const float eS6 = clampf(thh[M::S_H0], 1e-12f, 1.f), eR12 = clampf(thh[M::S_H1], 1e-12f, 1.f);
      fR = powf(c[0], nR) + powf(eR12 * c[1], nR);
      fS = powf(eS6 * c[0], nS) + powf(c[1], nS);
    }
  }

  __device__ static void prepare(const OdeArgs& a, int i, int b, int j, DrLane& L, float* c, float& y0) {
    c[0] = clampf(expf(a.cond[b * a.C + 0]) - 1.f, 1e-12f, 1e6f);
    c[1] = clampf(expf(a.cond[b * a.C + 1]) - 1.f, 1e-12f, 1e6f);
    L.r = clampf(th(a, M::S_r, i), 0.f, 4.f);
    L.K = clampf(th(a, M::S_K, i), 0.f, 4.f);
    L.invK = frcp(L.K);
    L.tlag = th(a, M::S_tlag, i);
    L.rc = th(a, M::S_rc, i);
    hill(a, i, c, L.fR, L.fS);
    L.sgn = j == 0 ? 1.f : -1.f;
    const int ds = deg_slot(j);
    L.deg = ds < 0 ? 0.f : clampf(th(a, ds, i), 1e-12f, (j == 6 || j == 7) ? 5.f : 2.f);
    const int as = a_slot(j);
    L.a = j == 0 ? 0.f : (as < 0 ? 1.f : th(a, as, i));
    L.c = L.rc * L.a;
    if (j == 2) { L.e = th(a, M::S_e81, i); L.KGR = th(a, M::S_KGR81, i); L.KGS = th(a, M::S_KGS81, i); }
    else if (j == 3) { L.e = th(a, M::S_e76, i); L.KGR = th(a, M::S_KGR76, i); L.KGS = th(a, M::S_KGS76, i); }
    else { L.e = 1.f; L.KGR = 0.f; L.KGS = 0.f; }
    L.prec = j < 4 ? th(a, M::NSLOT + j, i) : 1.f;
    L.m0 = j == 0 ? 1.f : 0.f; L.m6 = j == 6 ? 1.f : 0.f; L.m7 = j == 7 ? 1.f : 0.f;
    L.mo1 = (j >= 1 && j <= 3) ? 1.f : 0.f; L.mo2 = (j == 2 || j == 3) ? 1.f : 0.f;
    const int is = init_slot(j);
    y0 = is < 0 ? 0.f : th(a, is, i);
  }

  struct Eval {
    float x, lR, lS, sig, gr, g, gam, bR, bS, den, P;
  };
  __device__ __forceinline__ static float rhs(float t, float y, const DrLane& L, Eval& E) {
    E.x = bcast8<0>(y);
    E.lR = bcast8<6>(y);
    E.lS = bcast8<7>(y);
    E.sig = sigmoid_f(4.f * (t - L.tlag));
    E.gr = L.r * E.sig;
    E.g = 1.f - E.x * L.invK;
    E.gam = E.gr * E.g;
    E.bR = E.lR * E.lR * L.fR;
    E.bS = E.lS * E.lS * L.fS;
    const float num = L.e + L.KGR * E.bR + L.KGS * E.bS;
    E.den = 1.f + L.KGR * E.bR + L.KGS * E.bS;
    E.P = fdiv(num, E.den);
    return L.c * E.P + (L.sgn * E.gam - L.deg) * y;
  }
  __device__ __forceinline__ static float rhs(float t, float y, const DrLane& L) {
    Eval E;
    return rhs(t, y, L, E);
  }

  struct Adj {  // parameter adjoints: per-lane (cb, degb, eb, KGRb, KGSb) and shared (identical in all 8 lanes)
    float cb, degb, eb, KGRb, KGSb, rb, Kb, tlagb, fRb, fSb;
  };
  // returns (d rhs/d y)^T v for this lane's state; accumulates parameter adjoints
  __device__ __forceinline__ static float rhs_vjp(float t, float y, const DrLane& L, float v, Adj& A) {
    Eval E;
    rhs(t, y, L, E);
    return rhs_vjp(y, L, v, A, E);
  }
  // the same with the stage's intermediates already at hand (from the forward recomputation)
  __device__ __forceinline__ static float rhs_vjp(float y, const DrLane& L, float v, Adj& A, const Eval& E) {
    float yb = v * (L.sgn * E.gam - L.deg);
    A.cb += v * E.P;
    A.degb -= v * y;
    const float nb = fdiv(v * L.c, E.den);
    const float s = nb - nb * E.P;
    A.eb += nb;
    A.KGRb += s * E.bR;
    A.KGSb += s * E.bS;
    const float bRb = sum8(s * L.KGR);
    const float bSb = sum8(s * L.KGS);
    const float gamb = sum8(v * L.sgn * y);
    yb += L.m6 * (bRb * 2.f * E.lR * L.fR) + L.m7 * (bSb * 2.f * E.lS * L.fS);
    A.fRb += bRb * E.lR * E.lR;
    A.fSb += bSb * E.lS * E.lS;
    const float grb = gamb * E.g, gb = gamb * E.gr;
    yb -= L.m0 * (gb * L.invK);
    A.Kb += gb * E.x * L.invK * L.invK;
    A.rb += grb * E.sig;
    A.tlagb -= 4.f * grb * L.r * E.sig * (1.f - E.sig);
    return yb;
  }

  template <int SOLVER>
  __device__ __forceinline__ static float step(float t0, float t1, float h0, float y, const DrLane& L) {
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
      const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
      const float k1 = rhs(t0, y, L);
      const float k2 = rhs(t1, y + h * k1, L);
      return y + (0.5f * h) * (k1 + k2);
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      return y + (t1 - t0) * rhs(t0, y, L);
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      const float dt = t1 - t0;
      const float k1 = rhs(t0, y, L);
      return y + dt * rhs(t0 + dt * 0.5f, y + k1 * dt * 0.5f, L);
    } else {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f);
      const float k1 = rhs(t0, y, L);
      const float k2 = rhs(t0 + d3, y + d3 * k1, L);
      const float k3 = rhs(t0 + 2.f * d3, y + (dt * k2 - d3 * k1), L);
      const float k4 = rhs(t0 + dt, y + dt * (k1 - k2 + k3), L);
      return y + (k1 + 3.f * k2 + 3.f * k3 + k4) * (dt * 0.125f);
    }
  }
  // lam: adjoint of y_{k+1} -> returns adjoint of y_k
  template <int SOLVER>
  __device__ __forceinline__ static float step_vjp(float t0, float t1, float h0, float y, const DrLane& L, float lam,
                                                   Adj& A) {
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) {
      const float h = (SOLVER == VIHDS_SOLVER_MODEULER) ? h0 : (t1 - t0);
      const float hh = 0.5f * h;
      Eval E1;
      const float k1 = rhs(t0, y, L, E1);
      const float ya = y + h * k1;
      float v = hh * lam;
      const float w = rhs_vjp(t1, ya, L, v, A);
      lam += w;
      v += h * w;
      return lam + rhs_vjp(y, L, v, A, E1);
    } else if (SOLVER == VIHDS_SOLVER_EULER) {
      return lam + rhs_vjp(t0, y, L, (t1 - t0) * lam, A);
    } else if (SOLVER == VIHDS_SOLVER_MIDPOINT) {
      const float dt = t1 - t0;
      Eval E1;
      const float k1 = rhs(t0, y, L, E1);
      const float ym = y + k1 * dt * 0.5f;
      const float w = rhs_vjp(t0 + dt * 0.5f, ym, L, dt * lam, A);
      lam += w;
      return lam + rhs_vjp(y, L, 0.5f * dt * w, A, E1);
    } else {
      const float dt = t1 - t0, d3 = dt * (1.f / 3.f), d8 = dt * 0.125f;
      Eval E1, E2, E3;  // stage intermediates of the forward recomputation, reused by the transposed stages
      const float k1 = rhs(t0, y, L, E1);
      const float y2 = y + d3 * k1;
      const float k2 = rhs(t0 + d3, y2, L, E2);
      const float y3 = y + (dt * k2 - d3 * k1);
      const float k3 = rhs(t0 + 2.f * d3, y3, L, E3);
      const float y4 = y + dt * (k1 - k2 + k3);
      const float k4b = d8 * lam;
      float k1b = k4b, k2b = 3.f * k4b, k3b = 3.f * k4b;
      float w = rhs_vjp(t0 + dt, y4, L, k4b, A);
      lam += w; k1b += dt * w; k2b -= dt * w; k3b += dt * w;
      w = rhs_vjp(y3, L, k3b, A, E3);
      lam += w; k1b -= d3 * w; k2b += dt * w;
      w = rhs_vjp(y2, L, k2b, A, E2);
      lam += w; k1b += d3 * w;
      return lam + rhs_vjp(y, L, k1b, A, E1);
    }
  }

  // observed signal of lanes 0..3 (reference vihds/ode.py:84-93): OD, OD*RFP, OD*(YFP+F530), OD*(CFP+F480)
  __device__ __forceinline__ static float observe(float y, float x, const DrLane& L, float& inner) {
    const float ysh = dpp_mov<0x102>(0.f, y);  // row_shl:2 -> state j+2
    inner = L.mo1 * y + L.mo2 * ysh + L.m0;    // species 0: inner = 1
    return x * inner;
  }
};

template <int VERSION, int SOLVER>
__global__ void __launch_bounds__(256) dr_lane_fwd_kernel(OdeArgs a) {
  using D = DrLanes<VERSION>;
  const int tl = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int i0 = blockIdx.x * D::TPB + tl;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  DrLane L;
  float c[2], y;
  D::prepare(a, i, b, j, L, c, y);
  const float lc = LOG2PI_F - logf(L.prec);
  float lp = 0.f;
  const float* ob = a.obs + ((size_t)b * 4 + (j & 3)) * a.T;
  const float h0 = a.times[1] - a.times[0];
  const size_t n = a.n;
  // times and observations are prefetched one step ahead: a load issued inside the step it is needed in would sit
  // on the dependent chain behind an s_waitcnt vmcnt(0) (measured: ~half of the loop time)
  const bool want_lp = a.logp && j < 4;
  float tA = a.times[0], tB = a.times[1];
  float ob_cur = want_lp ? ob[0] : 0.f;
  for (int k = 0; k < a.T; ++k) {
    const float tC = (k + 1 < a.T) ? a.times[k + 1] : tB;
    const float ob_next = (want_lp && k + 1 < a.T) ? ob[k + 1] : 0.f;
    if (k > 0) {
      y = D::template step<SOLVER>(tA, tB, h0, y, L);
      tA = tB;
    }
    tB = tC;
    // land the prefetched values here, before this step's stores are issued: the vmcnt wait the compiler needs for
    // them then covers the previous step's stores (a full step old) instead of stalling on the ones just issued
    float ob_nx = ob_next;
    asm volatile("" : "+v"(tB), "+v"(ob_nx));
    if (a.traj && live) a.traj[((size_t)k * 8 + j) * n + i] = y;
    float inner;
    const float xp = D::observe(y, bcast8<0>(y), L, inner);
    if (j < 4) {
      if (a.xpred && live) a.xpred[((size_t)k * 4 + j) * n + i] = xp;
      const float e = xp - ob_cur;
      lp += -0.5f * (lc + L.prec * e * e);
    }
    ob_cur = ob_nx;
  }
  if (a.logp && live && j < 4) a.logp[(size_t)j * n + i] = lp;
}

template <int VERSION, int SOLVER>
__global__ void __launch_bounds__(256) dr_lane_bwd_kernel(OdeArgs a) {
  using D = DrLanes<VERSION>;
  using M = DrConstant<VERSION>;
  const int tl = threadIdx.x >> 3, j = threadIdx.x & 7;
  const int i0 = blockIdx.x * D::TPB + tl;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  DrLane L;
  float c[2], y0;
  D::prepare(a, i, b, j, L, c, y0);
  typename D::Adj A = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float lam = 0.f, precb = 0.f;
  const size_t n = a.n;
  const float glp = (a.g_logp && j < 4) ? a.g_logp[(a.logp_grad_broadcast ? 0 : (size_t)j * n) + i] : 0.f;
  const float* ob = a.obs + ((size_t)b * 4 + (j & 3)) * a.T;
  const float h0 = a.times[1] - a.times[0];

  float ynext = a.traj_in[((size_t)(a.T - 1) * 8 + j) * n + i];
  float ob_next = j < 4 ? ob[a.T - 1] : 0.f;
  float gx_next = (a.g_xpred && j < 4) ? a.g_xpred[((size_t)(a.T - 1) * 4 + j) * n + i] : 0.f;
  float gt_next = a.g_traj ? a.g_traj[((size_t)(a.T - 1) * 8 + j) * n + i] : 0.f;
  float tHi = a.times[a.T - 1], tLo = a.times[a.T - 1];  // step k uses (times[k], times[k+1]) = (tLo, tHi)
  for (int k = a.T - 1; k >= 0; --k) {
    const float y = ynext, obk = ob_next, gxk = gx_next, gtk = gt_next;
    const float tK = tLo;  // times[k]
    if (k > 0) {  // prefetch everything the next (earlier) step needs
      ynext = a.traj_in[((size_t)(k - 1) * 8 + j) * n + i];
      ob_next = j < 4 ? ob[k - 1] : 0.f;
      if (a.g_xpred && j < 4) gx_next = a.g_xpred[((size_t)(k - 1) * 4 + j) * n + i];
      if (a.g_traj) gt_next = a.g_traj[((size_t)(k - 1) * 8 + j) * n + i];
      tLo = a.times[k - 1];
    }
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
  // ---- parameter adjoints -> theta rows (each slot row written by exactly one lane)
  const float rcb = sum8(A.cb * L.a);
  if (!live) return;
  auto put = [&](int slot, float v) { a.g_theta[(size_t)a.slot_row[slot] * n + i] = v; };
  auto raw = [&](int slot) { return a.theta[(size_t)a.slot_row[slot] * n + i]; };
  const int is = D::init_slot(j);
  if (is >= 0) put(is, lam);
  if (j < 4) put(M::NSLOT + j, precb);
  const int ds = D::deg_slot(j);
  if (ds >= 0) put(ds, A.degb * clamp_pass(raw(ds), 1e-12f, (j == 6 || j == 7) ? 5.f : 2.f));
  const int as = D::a_slot(j);
  if (as >= 0) put(as, A.cb * L.rc);
  if (j == 2) { put(M::S_e81, A.eb); put(M::S_KGR81, A.KGRb); put(M::S_KGS81, A.KGSb); }
  if (j == 3) { put(M::S_e76, A.eb); put(M::S_KGR76, A.KGRb); put(M::S_KGS76, A.KGSb); }
  if (j == 0) {
    put(M::S_r, A.rb * clamp_pass(raw(M::S_r), 0.f, 4.f));
    put(M::S_K, A.Kb * clamp_pass(raw(M::S_K), 0.f, 4.f));
    put(M::S_tlag, A.tlagb);
    put(M::S_rc, rcb);
    float thh[M::NSLOT], thb[M::NSLOT];
    thh[M::S_nR] = raw(M::S_nR); thh[M::S_nS] = raw(M::S_nS); thh[M::S_H0] = raw(M::S_H0); thh[M::S_H1] = raw(M::S_H1);
    if (VERSION == 1) {
      thh[M::S_H2] = raw(M::S_H2); thh[M::S_H3] = raw(M::S_H3);
      hill_frac_vjp(thh[M::S_nR], thh[M::S_H0], thh[M::S_H1], c[0], c[1], A.fRb, thb[M::S_nR], thb[M::S_H0], thb[M::S_H1]);
      hill_frac_vjp(thh[M::S_nS], thh[M::S_H2], thh[M::S_H3], c[0], c[1], A.fSb, thb[M::S_nS], thb[M::S_H2], thb[M::S_H3]);
      put(M::S_H2, thb[M::S_H2]); put(M::S_H3, thb[M::S_H3]);
    } else {
      const float nR = clampf(thh[M::S_nR], 0.5f, 3.f), nS = clampf(thh[M::S_nS], 0.5f, 3.f);
      const float eS6 = clampf(thh[M::S_H0], 1e-12f, 1.f), eR12 = clampf(thh[M::S_H1], 1e-12f, 1.f);
      float nRb = 0.f, nSb = 0.f, dummy = 0.f, a12b = 0.f, a6b = 0.f;
      const float a12 = eR12 * c[1], a6 = eS6 * c[0];
      pow_vjp(c[0], nR, powf(c[0], nR), A.fRb, dummy, nRb);
      pow_vjp(a12, nR, powf(a12, nR), A.fRb, a12b, nRb);
      pow_vjp(a6, nS, powf(a6, nS), A.fSb, a6b, nSb);
      pow_vjp(c[1], nS, powf(c[1], nS), A.fSb, dummy, nSb);
      thb[M::S_nR] = nRb * clamp_pass(thh[M::S_nR], 0.5f, 3.f);
      thb[M::S_nS] = nSb * clamp_pass(thh[M::S_nS], 0.5f, 3.f);
      thb[M::S_H0] = a6b * c[0] * clamp_pass(thh[M::S_H0], 1e-12f, 1.f);
      thb[M::S_H1] = a12b * c[1] * clamp_pass(thh[M::S_H1], 1e-12f, 1.f);
    }
    put(M::S_nR, thb[M::S_nR]); put(M::S_nS, thb[M::S_nS]); put(M::S_H0, thb[M::S_H0]); put(M::S_H1, thb[M::S_H1]);
  }
}

template <int VERSION>
inline int launch_dr_lanes(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  const dim3 grid((a.n + DrLanes<VERSION>::TPB - 1) / DrLanes<VERSION>::TPB), block(256);
#define VIHDS_LCASE(SV)                                                                          \
  case SV:                                                                                       \
    if (backward) hipLaunchKernelGGL((dr_lane_bwd_kernel<VERSION, SV>), grid, block, 0, st, a);  \
    else hipLaunchKernelGGL((dr_lane_fwd_kernel<VERSION, SV>), grid, block, 0, st, a);           \
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
