// relay_constant, degrader_constant, prpr_constant and their _precisions forms with SIXTEEN LANES PER TRAJECTORY -- one lane per
// ODE state; the lane models RlRelay / RlDegrader / RlPrpr below say which (written for relay: reference models/relay_constant.py:13-134
// RHS, :199-251 states, vihds/precisions.py:55-61,76-87 neural precisions without a hidden layer; BASELINE config 5).
//
// The thread-per-trajectory kernels (vihds_ode_kernels.hpp) put 7 200 trajectories on 113 wavefronts: each walks its
// ~200 RHS evaluations of ~340 instructions alone on a SIMD, and the launch lasts as long as that serial instruction
// stream (forward 207 us, adjoint 464 us at B=36, S=200, T=99, midpoint: 3.5 % / 1.3 % of the HBM roofline).  Here one
// lane owns one ODE state -- 12 species + the 4 neural precision states = 16 lanes, 4 trajectories per wavefront, 1 800
// wavefronts -- and every state obeys the same form
//
//     dy_l = F0_l + cP_l P_l(luxR, lasR) + cQ_l Q_l(x, I_l) + [prec] sigma(zp_l) - (s_l gamma(x, t) + deg_l + [prec] sigma(zd_l)) y_l
//
// with per-lane constants (s = -1 for OD itself, +1 for the diluted species, 0 for the AHL quadratures and the precisions;
// P = the promoter of the lane: P81 for yfp / luxI, P76 for cfp / lasI; Q = x I / (1 + I / K) with I = luxI for c6, lasI
// for c12; zp, zd = the precision network's rows of the lane), so the RHS is one straight-line evaluation for all lanes.
// What a lane needs from the others (x, luxR, lasR, its I, the tanh'd network inputs) goes through a 48-float LDS patch of
// the trajectory: one wavefront's LDS operations are served in order, so no barrier is involved.  The adjoint sends back
// through the same patch the network's pre-activation adjoints (to every input lane) and the quadratures' x / I adjoints,
// and sums over the 16 lanes (gamma's adjoint, the promoters' luxR / lasR adjoints) are four DPP steps.
//
// Every parameter gradient is accumulated per lane as the adjoint of that lane's OWN constants (8 numbers) plus the
// lane-uniform growth adjoints; one lane maps them to the model's prepared parameters after the time loop and hands them
// to RelayConstant::prepare_vjp / init_vjp -- the same code the thread-per-trajectory adjoint ends with.  The precision
// network's weight gradients are per-lane accumulators too (lane 12+o owns row o of both matrices: 2 x 13 + 2 numbers),
// added up over the block's 16 trajectories through LDS, left as one partial row per block in `aux` and summed in block
// order by relay_lane_wreduce_kernel (fixed order: deterministic; no 120 MB dump, no contraction pass).
//
// Schemes: the five fixed-grid ones as explicit Runge-Kutta tableaux (same stages and weights as ode_step /
// ode_step_vjp; the sums are taken in tableau order, a rounding-level difference).  The adjoint is the discrete adjoint.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/vihds_hip.h"
#include "vihds_args.hpp"
#include "vihds_models.hpp"
#include "vihds_iwae_inline.hpp"
#include "vihds_theta_stage.hpp"
#include "vihds_wave.hpp"

namespace vihds {

constexpr int RL_G = 16;                 // lanes per trajectory
constexpr int RL_T = 256;                // threads per block
constexpr int RL_TR = RL_T / RL_G;       // trajectories per block
constexpr int RL_NIN = 13;               // network inputs at most: t, 12 species (arrays; a model uses 1 + its species)
__host__ __device__ constexpr int rl_nwrow(int nin) { return 2 * nin + 2; }  // weight-gradient numbers per precision lane
__host__ __device__ constexpr int rl_nwg(int nin) { return 4 * rl_nwrow(nin); }  // per block partial row (relay: 112)
constexpr int RL_PATCH = 48;             // LDS floats per trajectory: y [16] | h [16] | adjoint scratch [16]
constexpr float RL_LOG2PI = 1.8378770664093453f;
typedef float rl_v2 __attribute__((ext_vector_type(2)));  // (production, degradation) pairs: v_pk_fma_f32

__host__ __device__ inline bool relay_lanes_applicable(int n, int solver, int kernel_variant, int n_hidden_prec) {
  return kernel_variant != 1 && n <= 16384 && solver >= VIHDS_SOLVER_MODEULER && solver <= VIHDS_SOLVER_RK4 &&
         n_hidden_prec < 1;
}
__host__ __device__ inline long long relay_lanes_aux_floats(int n, int n_species = 12) {
  return (long long)((n + RL_TR - 1) / RL_TR) * rl_nwg(1 + n_species);
}

// ---- tableaux -----------------------------------------------------------------------------------------------------
template <int SOLVER>
struct RlTab {
  static constexpr int S = SOLVER == VIHDS_SOLVER_EULER ? 1 : (SOLVER == VIHDS_SOLVER_RK4 ? 4 : 2);
  // Y_i = y + h sum_j A(i, j) k_j ;  y' = y + h sum_i B(i) k_i
  __device__ static constexpr float A(int i, int j) {
    if (SOLVER == VIHDS_SOLVER_MIDPOINT) return (i == 1 && j == 0) ? 0.5f : 0.f;
    if (SOLVER == VIHDS_SOLVER_RK4) {  // torchdiffeq 0.1 rk4_alt_step_func (3/8 rule)
      if (i == 1) return j == 0 ? (1.f / 3.f) : 0.f;
      if (i == 2) return j == 0 ? -(1.f / 3.f) : (j == 1 ? 1.f : 0.f);
      if (i == 3) return j == 1 ? -1.f : 1.f;
      return 0.f;
    }
    return (i == 1 && j == 0) ? 1.f : 0.f;  // Heun (vihds/solvers.py:12-16)
  }
  __device__ static constexpr float B(int i) {
    if (SOLVER == VIHDS_SOLVER_EULER) return 1.f;
    if (SOLVER == VIHDS_SOLVER_MIDPOINT) return i == 1 ? 1.f : 0.f;
    if (SOLVER == VIHDS_SOLVER_RK4) return (i == 0 || i == 3) ? 0.125f : 0.375f;
    return 0.5f;
  }
  // step size and stage times: modeuler uses h = times[1] - times[0] for every step and evaluates stage 2 at t1
  __device__ static float h(float t0, float t1, float h0) { return SOLVER == VIHDS_SOLVER_MODEULER ? h0 : t1 - t0; }
  __device__ static float ts(int i, float t0, float t1) {
    if (SOLVER == VIHDS_SOLVER_MODEULER || SOLVER == VIHDS_SOLVER_MODEULERWHILE) return i == 0 ? t0 : t1;
    const float dt = t1 - t0;
    if (SOLVER == VIHDS_SOLVER_MIDPOINT) return i == 0 ? t0 : t0 + dt * 0.5f;
    if (SOLVER == VIHDS_SOLVER_RK4) {
      const float d3 = dt * (1.f / 3.f);
      return i == 0 ? t0 : (i == 1 ? t0 + d3 : (i == 2 ? t0 + 2.f * d3 : t0 + dt));
    }
    return t0;
  }
};

// ---- cross-lane helpers (16-lane groups = DPP rows) ------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float rl_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rl_sum16(float v) {  // every lane of the row gets the row's sum
  v += rl_dpp<0xB1>(v);   // quad_perm [1,0,3,2]
  v += rl_dpp<0x4E>(v);   // quad_perm [2,3,0,1]
  v += rl_dpp<0x141>(v);  // row_half_mirror
  v += rl_dpp<0x140>(v);  // row_mirror
  return v;
}
__device__ __forceinline__ void rl_wave_fence() {  // keep LDS writes ahead of the reads that follow (same wavefront)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- per-lane model ------------------------------------------------------------------------------------------------
struct RlLane {
  // shared by the 16 lanes of a trajectory
  float r, iKx, tlag, rKx;  // growth rate, 1 / K, lag, (unused slot kept for alignment of the struct in registers)
  // this lane
  float gsgn, deg, F0, cP, e, aR, aS, cQ, iK, isP;
  int isrc, l;
  // which-lane-am-I as 0 / 1 factors: `x += m * v` is one FMA where `if (l == ...) x += v` is an exec-mask branch that also
  // cuts the basic block (the adjoint's loop body had 30 of them)
  float m0, m6, m7, sz, sw;              // OD lane, luxR, lasR; weights of the two quadratures' source adjoints
  float i0, i1, i2, i3;                  // observation injections: OD; rfp; yfp and f530; cfp and f480
  // where this lane's hand-overs go in the patch; a lane that has none writes a spare slot (44..47) instead of branching
  int a_h, a_zx, a_zy, a_q1, a_q2, a_ix, a_ip;
  float n0, n1, n2, n3;                  // observed signal j = l & 3: x * (n0 + n1 rfp + n2 (yfp + f530) + n3 (cfp + f480))
  // The precision network's eight pre-activations (production rows 0..3, degradation rows 4..7) are formed by lane PAIRS:
  // lane l works on output o(l) = (l - NSP + 4) & 7 over half h(l) = ((l - NSP + 4) & 15) >> 3 of the sixteen input slots
  // (eight FMAs), the two halves meet through one row_ror:8 DPP add, one sigmoid per lane; the precision lane NSP + o'
  // then holds sigma(zd_o') itself and takes sigma(zp_o') from lane l - 4 (row_shr:4).  (Round 3-4: every lane ran both of
  // "its" rows over all thirteen inputs -- 13 packed FMAs + two sigmoids in sixteen lanes for the four that used them.)
  float wq[8], bq;                       // this lane's half row and (first half only) its bias
  int a_hr;                              // where the half's eight input slots start in the patch (16 or 24)
  float cw[8];                           // column of both matrices that multiplies this lane's tanh (species lanes)
};
struct RlEval {  // what one RHS evaluation leaves behind for its VJP
  float x, luxR, lasR, I, sig, gr, g, gamma, a, den, P, denq, Q, sp, sd, hl;
  float lux2, las2;  // luxR^2, lasR^2 (the promoter's forward forms them; its VJP reads them again)
};
// sigma(4 (t - tlag)) of every (step, stage) of the grid depends on the trajectory's lag only: tabulated once per kernel
// in LDS ([trajectory][(T - 1) stages]), sixteen entries at a time by the trajectory's lanes; an evaluation reads its entry
// (one broadcast LDS read) where it used to spend six VALU instructions in all sixteen lanes
template <class Tab>
__device__ __forceinline__ void rl_sigma_table(float* sg, const float* tl, int T, float tlag, int l) {
  const int n = (T - 1) * Tab::S;
  for (int q = l; q < n; q += RL_G) {
    const int k = q / Tab::S, s = q - k * Tab::S;
    sg[q] = sigmoid_f(4.f * (Tab::ts(s, tl[k], tl[k + 1]) - tlag));
  }
}

// publish (Y_l, tanh) and evaluate dy_l.  `pt` = this trajectory's LDS patch.
// (TAIL_FENCE = false: the caller still reads the published states and places the fence itself)
template <class LM, bool PREC, bool TAIL_FENCE = true>
__device__ __forceinline__ float rl_rhs(const RlLane& c, float t, float sig, float Y, float* pt, RlEval& E) {
  constexpr int NSP = LM::NSP;
  const int l = c.l;
  pt[l] = Y;
  if (PREC) {
    const float hl = ftanh(l == NSP ? t : Y);
    E.hl = hl;
    pt[c.a_h] = hl;
  }
  rl_wave_fence();
  E.x = pt[0]; E.luxR = pt[6]; E.lasR = pt[7]; E.I = pt[c.isrc];
  E.sig = sig;
  E.gr = c.r * E.sig;
  E.g = 1.f - E.x * c.iKx;
  E.gamma = E.gr * E.g;
  E.a = 0.f; E.den = 1.f; E.P = 0.f; E.denq = 1.f; E.Q = 0.f; E.lux2 = 0.f; E.las2 = 0.f;
  float dy = c.F0;
  if (LM::HAS_P) {
    E.lux2 = E.luxR * E.luxR; E.las2 = E.lasR * E.lasR;
    E.a = c.aR * E.lux2 + c.aS * E.las2;
    E.den = 1.f + E.a;
    E.P = fdiv(c.e + E.a, E.den);
    dy += c.cP * E.P;
  }
  if (LM::HAS_Q) {
    E.denq = 1.f + E.I * c.iK;
    E.Q = fdiv(E.x * E.I, E.denq);
    dy += c.cQ * E.Q;
  }
  float D = c.gsgn * E.gamma + c.deg;
  E.sp = 0.f; E.sd = 0.f;
  if (PREC) {
    // this lane's half of its output row (input slots past the model's 1 + NSP were zeroed at kernel entry and are never
    // written; their weights are zero as well)
    const float4 ha = *reinterpret_cast<const float4*>(pt + c.a_hr), hb = *reinterpret_cast<const float4*>(pt + c.a_hr + 4);
    float z = fmaf(c.wq[0], ha.x, c.bq), zb = c.wq[1] * ha.y;  // (two chains)
    z = fmaf(c.wq[2], ha.z, z); zb = fmaf(c.wq[3], ha.w, zb);
    z = fmaf(c.wq[4], hb.x, z); zb = fmaf(c.wq[5], hb.y, zb);
    z = fmaf(c.wq[6], hb.z, z); zb = fmaf(c.wq[7], hb.w, zb);
    z += zb;
    z += rl_dpp<0x128>(z);  // row_ror:8: the other half of the same output
    const float sg_ = sigmoid_f(z);
    E.sd = sg_;                   // precision lane NSP + o': sigma(zd_o') is its own output ...
    E.sp = rl_dpp<0x114>(sg_);    // ... and sigma(zp_o') sits four lanes below (row_shr:4)
    dy += c.isP * E.sp;
    D += c.isP * E.sd;
  }
  if (TAIL_FENCE) rl_wave_fence();  // (the patch is rewritten by the next evaluation: reads first)
  return dy - D * Y;
}

// per-lane adjoint accumulators
struct RlAcc {
  float F0b, cPb, eb, aRb, aSb, cQb, iKb, degb;  // adjoints of the lane's own constants
  // growth parameters (the same numbers in every lane of the trajectory), kept as RAW sums over the evaluations and scaled
  // once after the time loop (rl_acc_finish): rb = sum grb sig;  Kb = iKx^2 sum gb x;  tlagb = -4 r sum grb (sig - sig^2)
  float rb, Kb, tlagb;
  // weight gradients of the precision network: lane l <= NSP owns COLUMN j(l) (the input it publishes: t or its own species)
  // of both matrices -- its own tanh times the eight pre-activation adjoints every lane reads anyway: 4 packed FMAs per
  // evaluation where a row per precision lane cost 13 in all sixteen lanes; the precision lanes keep the bias sums
  // (scalars, not (production, degradation) pairs: packed fp32 instructions cost this kernel time -- the pairs' register
  // shuffles and the v_pk_* issue rate; measured with -fno-slp-vectorize, round 5)
  float wcp[4], wcd[4], b2p, b2d;
};

// VJP of one evaluation whose forward quantities (E, hv) are at hand: v = adjoint of dy_l; returns the adjoint of Y_l;
// accumulates parameter adjoints.  Uses only the scratch third of the patch.
template <class LM, bool PREC>
__device__ __forceinline__ float rl_vjp_core(const RlLane& c, float Y, float v, const RlEval& E, float* pt, RlAcc& A) {
  const float D = c.gsgn * E.gamma + c.deg + (PREC ? c.isP * E.sd : 0.f);
  float yb = -v * D;
  A.F0b += v;
  A.cPb += v * E.P;
  A.cQb += v * E.Q;
  A.degb -= v * Y;
  const float gammab = rl_sum16(-v * c.gsgn * Y);
  // promoter
  float bRt = 0.f, bSt = 0.f;
  if (LM::HAS_P) {
    const float Pb = v * c.cP;
    const float nb = fdiv(Pb, E.den);
    const float sP = nb * (1.f - E.P);
    A.eb += nb;
    A.aRb = fmaf(sP, E.lux2, A.aRb);
    A.aSb = fmaf(sP, E.las2, A.aSb);
    bRt = rl_sum16(sP * c.aR);
    bSt = rl_sum16(sP * c.aS);
  }
  // quadrature Q = x I / (1 + I iK)
  float xq = 0.f, Iq = 0.f;
  if (LM::HAS_Q) {
    const float iden = frcp(E.denq);
    const float t1 = (v * c.cQ) * iden;   // Qb / denq
    xq = t1 * E.I;
    Iq = (t1 * E.x) * iden;
    A.iKb = fmaf(-(Iq * E.I), E.I, A.iKb);
  }
  // precision network
  if (PREC) {
    const float vP = c.isP * v;
    const float zbx = vP * fmaf(-E.sp, E.sp, E.sp);            // isP v sp (1 - sp)
    const float zby = -(vP * Y) * fmaf(-E.sd, E.sd, E.sd);     // -isP v Y sd (1 - sd)
    A.b2p += zbx;
    A.b2d += zby;
    pt[c.a_zx] = zbx;
    pt[c.a_zy] = zby;
  }
  if (LM::HAS_Q) { pt[c.a_q1] = xq; pt[c.a_q2] = Iq; }
  rl_wave_fence();
  const float4 q4 = *reinterpret_cast<const float4*>(pt + 40);
  if (PREC) {
    const float4 za = *reinterpret_cast<const float4*>(pt + 32), zd4 = *reinterpret_cast<const float4*>(pt + 36);
    const float hb = c.cw[0] * za.x + c.cw[1] * za.y + c.cw[2] * za.z + c.cw[3] * za.w + c.cw[4] * zd4.x + c.cw[5] * zd4.y +
                     c.cw[6] * zd4.z + c.cw[7] * zd4.w;
    yb += hb * (1.f - E.hl * E.hl);  // (the columns cw are zero outside the species lanes)
    A.wcp[0] = fmaf(za.x, E.hl, A.wcp[0]); A.wcp[1] = fmaf(za.y, E.hl, A.wcp[1]);
    A.wcp[2] = fmaf(za.z, E.hl, A.wcp[2]); A.wcp[3] = fmaf(za.w, E.hl, A.wcp[3]);
    A.wcd[0] = fmaf(zd4.x, E.hl, A.wcd[0]); A.wcd[1] = fmaf(zd4.y, E.hl, A.wcd[1]);
    A.wcd[2] = fmaf(zd4.z, E.hl, A.wcd[2]); A.wcd[3] = fmaf(zd4.w, E.hl, A.wcd[3]);
  }
  // growth: gamma = gr (1 - x / K)
  const float grb = gammab * E.g, gb = gammab * E.gr;
  A.Kb = fmaf(gb, E.x, A.Kb);                                   // (raw sums: rl_acc_finish scales them)
  A.rb = fmaf(grb, E.sig, A.rb);
  A.tlagb = fmaf(grb, fmaf(-E.sig, E.sig, E.sig), A.tlagb);
  yb = fmaf(c.m0, -gb * c.iKx + (LM::HAS_Q ? q4.x + q4.y : 0.f), yb);
  if (LM::HAS_P) yb = fmaf(c.m6, 2.f * E.luxR * bRt, fmaf(c.m7, 2.f * E.lasR * bSt, yb));
  if (LM::HAS_Q) yb = fmaf(c.sz, q4.z, fmaf(c.sw, q4.w, yb));  // the quadratures' sources (relay: luxI, lasI; degrader: aiiA)
  rl_wave_fence();
  return yb;
}
// the growth adjoints from their raw sums (see RlAcc)
__device__ __forceinline__ void rl_acc_finish(const RlLane& c, RlAcc& A) {
  A.Kb *= c.iKx * c.iKx;
  A.tlagb *= -4.f * c.r;
}
// ... with the forward quantities recomputed first
template <class LM, bool PREC>
__device__ __forceinline__ float rl_rhs_vjp(const RlLane& c, float t, float sig, float Y, float v, float* pt, RlAcc& A) {
  RlEval E;
  (void)rl_rhs<LM, PREC>(c, t, sig, Y, pt, E);
  return rl_vjp_core<LM, PREC>(c, Y, v, E, pt, A);
}

// ---- the models: which lane owns which state, and how the adjoints of the lanes' constants map back --------------------
// lane(): the constants of lane l from the prepared parameters p.  map(): pb += the adjoints of p from the lanes' accumulators
// T(lane, field), fields: 0 F0, 1 cP, 2 e, 3 aR, 4 aS, 5 cQ, 6 iK, 7 deg.  Q0: first of the two quadrature lanes;
// source_adjoint(): what the quadratures hand back to their source lanes (zI = adjoint of I in lane Q0, wI in lane Q0 + 1).
struct RlRelay {  // reference models/relay_constant.py:91-134
  using M = RelayConstant;
  static constexpr int NSP = 12, Q0 = 10, NCOND = 2;
  static constexpr bool HAS_P = true, HAS_Q = true, OBS_SUM = true;
  __device__ static float source_adjoint(int l, float zI, float wI) { return l == 8 ? zI : (l == 9 ? wI : 0.f); }
  __device__ static void lane(int l, const float* p, RlLane& c) {
    const float rc = p[M::P_rc], fR = p[M::P_fR], fS = p[M::P_fS];
    switch (l) {
      case 1: c.deg = p[M::P_drfp]; c.F0 = rc; break;
      case 2: c.deg = p[M::P_dyfp]; c.cP = rc * p[M::P_aYFP]; c.e = p[M::P_e81]; c.aR = p[M::P_KGR81] * fR; c.aS = p[M::P_KGS81] * fS; break;
      case 3: c.deg = p[M::P_dcfp]; c.cP = rc * p[M::P_aCFP]; c.e = p[M::P_e76]; c.aR = p[M::P_KGR76] * fR; c.aS = p[M::P_KGS76] * fS; break;
      case 4: c.F0 = rc * p[M::P_a530]; break;
      case 5: c.F0 = rc * p[M::P_a480]; break;
      case 6: c.deg = p[M::P_dR]; c.F0 = rc * p[M::P_aR]; break;
      case 7: c.deg = p[M::P_dS]; c.F0 = rc * p[M::P_aS]; break;
      case 8: c.deg = p[M::P_dluxI]; c.cP = rc; c.e = p[M::P_e81]; c.aR = p[M::P_KGR81] * fR; c.aS = p[M::P_KGS81] * fS; break;
      case 9: c.deg = p[M::P_dlasI]; c.cP = rc; c.e = p[M::P_e76]; c.aR = p[M::P_KGR76] * fR; c.aS = p[M::P_KGS76] * fS; break;
      case 10: c.cQ = p[M::P_KC6] * rc; c.iK = frcp(p[M::P_Klux]); c.isrc = 8; break;
      case 11: c.cQ = p[M::P_KC12] * rc; c.iK = frcp(p[M::P_Klas]); c.isrc = 9; break;
      default: break;
    }
    c.gsgn = l == 0 ? -1.f : (l <= 9 ? 1.f : 0.f);
  }
  template <class TF>
  __device__ static void map(TF T_, const float* p, float* pb) {
    const float rc = p[M::P_rc], fR = p[M::P_fR], fS = p[M::P_fS];
    pb[M::P_rc] = T_(1, 0) + T_(4, 0) * p[M::P_a530] + T_(5, 0) * p[M::P_a480] + T_(6, 0) * p[M::P_aR] + T_(7, 0) * p[M::P_aS] +
                  T_(2, 1) * p[M::P_aYFP] + T_(3, 1) * p[M::P_aCFP] + T_(8, 1) + T_(9, 1) + T_(10, 5) * p[M::P_KC6] +
                  T_(11, 5) * p[M::P_KC12];
    pb[M::P_drfp] = T_(1, 7); pb[M::P_dyfp] = T_(2, 7); pb[M::P_dcfp] = T_(3, 7); pb[M::P_dR] = T_(6, 7); pb[M::P_dS] = T_(7, 7);
    pb[M::P_dluxI] = T_(8, 7); pb[M::P_dlasI] = T_(9, 7);
    pb[M::P_e81] = T_(2, 2) + T_(8, 2); pb[M::P_e76] = T_(3, 2) + T_(9, 2);
    pb[M::P_KGR81] = (T_(2, 3) + T_(8, 3)) * fR; pb[M::P_KGS81] = (T_(2, 4) + T_(8, 4)) * fS;
    pb[M::P_KGR76] = (T_(3, 3) + T_(9, 3)) * fR; pb[M::P_KGS76] = (T_(3, 4) + T_(9, 4)) * fS;
    pb[M::P_fR] = (T_(2, 3) + T_(8, 3)) * p[M::P_KGR81] + (T_(3, 3) + T_(9, 3)) * p[M::P_KGR76];
    pb[M::P_fS] = (T_(2, 4) + T_(8, 4)) * p[M::P_KGS81] + (T_(3, 4) + T_(9, 4)) * p[M::P_KGS76];
    pb[M::P_aYFP] = T_(2, 1) * rc; pb[M::P_aCFP] = T_(3, 1) * rc;
    pb[M::P_a530] = T_(4, 0) * rc; pb[M::P_a480] = T_(5, 0) * rc; pb[M::P_aR] = T_(6, 0) * rc; pb[M::P_aS] = T_(7, 0) * rc;
    pb[M::P_KC6] = T_(10, 5) * rc; pb[M::P_KC12] = T_(11, 5) * rc;
    const float iKl = frcp(p[M::P_Klux]), iKs = frcp(p[M::P_Klas]);
    pb[M::P_Klux] = -T_(10, 6) * iKl * iKl; pb[M::P_Klas] = -T_(11, 6) * iKs * iKs;
  }
};
struct RlDegrader {  // reference models/degrader_constant.py:103-143: aiiA in lane 8, c6 / c12 = x rC aiiA in lanes 9, 10
  using M = DegraderConstant;
  static constexpr int NSP = 11, Q0 = 9, NCOND = 3;
  static constexpr bool HAS_P = true, HAS_Q = true, OBS_SUM = true;
  __device__ static float source_adjoint(int l, float zI, float wI) { return l == 8 ? zI + wI : 0.f; }
  __device__ static void lane(int l, const float* p, RlLane& c) {
    const float rc = p[M::P_rc], fR = p[M::P_fR], fS = p[M::P_fS];
    switch (l) {
      case 1: c.deg = p[M::P_drfp]; c.F0 = rc; break;
      case 2: c.deg = p[M::P_dyfp]; c.cP = rc * p[M::P_aYFP]; c.e = p[M::P_e81]; c.aR = p[M::P_KGR81] * fR; c.aS = p[M::P_KGS81] * fS; break;
      case 3: c.deg = p[M::P_dcfp]; c.cP = rc * p[M::P_aCFP]; c.e = p[M::P_e76]; c.aR = p[M::P_KGR76] * fR; c.aS = p[M::P_KGS76] * fS; break;
      case 4: c.F0 = rc * p[M::P_a530]; break;
      case 5: c.F0 = rc * p[M::P_a480]; break;
      case 6: c.deg = p[M::P_dR]; c.F0 = rc * p[M::P_aR]; break;
      case 7: c.deg = p[M::P_dS]; c.F0 = rc * p[M::P_aS]; break;
      case 8: c.F0 = rc * p[M::P_aI] * p[M::P_PBAD] - p[M::P_daiiA]; break;  // d aiiA = rc aI PBAD - (daiiA + gamma aiiA)
      case 9: c.cQ = p[M::P_rC6]; c.isrc = 8; break;                          // (no saturation: iK = 0)
      case 10: c.cQ = p[M::P_rC12]; c.isrc = 8; break;
      default: break;
    }
    c.gsgn = l == 0 ? -1.f : (l <= 8 ? 1.f : 0.f);
  }
  template <class TF>
  __device__ static void map(TF T_, const float* p, float* pb) {
    const float rc = p[M::P_rc], fR = p[M::P_fR], fS = p[M::P_fS];
    pb[M::P_rc] = T_(1, 0) + T_(4, 0) * p[M::P_a530] + T_(5, 0) * p[M::P_a480] + T_(6, 0) * p[M::P_aR] + T_(7, 0) * p[M::P_aS] +
                  T_(2, 1) * p[M::P_aYFP] + T_(3, 1) * p[M::P_aCFP] + T_(8, 0) * p[M::P_aI] * p[M::P_PBAD];
    pb[M::P_drfp] = T_(1, 7); pb[M::P_dyfp] = T_(2, 7); pb[M::P_dcfp] = T_(3, 7); pb[M::P_dR] = T_(6, 7); pb[M::P_dS] = T_(7, 7);
    pb[M::P_e81] = T_(2, 2); pb[M::P_e76] = T_(3, 2);
    pb[M::P_KGR81] = T_(2, 3) * fR; pb[M::P_KGS81] = T_(2, 4) * fS;
    pb[M::P_KGR76] = T_(3, 3) * fR; pb[M::P_KGS76] = T_(3, 4) * fS;
    pb[M::P_fR] = T_(2, 3) * p[M::P_KGR81] + T_(3, 3) * p[M::P_KGR76];
    pb[M::P_fS] = T_(2, 4) * p[M::P_KGS81] + T_(3, 4) * p[M::P_KGS76];
    pb[M::P_aYFP] = T_(2, 1) * rc; pb[M::P_aCFP] = T_(3, 1) * rc;
    pb[M::P_a530] = T_(4, 0) * rc; pb[M::P_a480] = T_(5, 0) * rc; pb[M::P_aR] = T_(6, 0) * rc; pb[M::P_aS] = T_(7, 0) * rc;
    pb[M::P_aI] = T_(8, 0) * rc * p[M::P_PBAD]; pb[M::P_PBAD] = T_(8, 0) * rc * p[M::P_aI]; pb[M::P_daiiA] = -T_(8, 0);
    pb[M::P_rC6] = T_(9, 5); pb[M::P_rC12] = T_(10, 5);
  }
};
struct RlPrpr {  // reference models/prpr_constant.py:46-69: every species with a constant production
  using M = PrprConstant;
  static constexpr int NSP = 6, Q0 = 14, NCOND = 0;  // (prpr_constant reads no treatment: a.cond may be empty)
  static constexpr bool HAS_P = false, HAS_Q = false, OBS_SUM = true;
  __device__ static float source_adjoint(int, float, float) { return 0.f; }
  __device__ static void lane(int l, const float* p, RlLane& c) {
    const float rc = p[M::P_rc];
    switch (l) {
      case 1: c.deg = p[M::P_drfp]; c.F0 = rc; break;
      case 2: c.deg = p[M::P_dyfp]; c.F0 = rc * p[M::P_aYFP]; break;
      case 3: c.deg = p[M::P_dcfp]; c.F0 = rc * p[M::P_aCFP]; break;
      case 4: c.F0 = rc * p[M::P_a530]; break;
      case 5: c.F0 = rc * p[M::P_a480]; break;
      default: break;
    }
    c.gsgn = l == 0 ? -1.f : (l <= 5 ? 1.f : 0.f);
  }
  template <class TF>
  __device__ static void map(TF T_, const float* p, float* pb) {
    const float rc = p[M::P_rc];
    pb[M::P_rc] = T_(1, 0) + T_(2, 0) * p[M::P_aYFP] + T_(3, 0) * p[M::P_aCFP] + T_(4, 0) * p[M::P_a530] + T_(5, 0) * p[M::P_a480];
    pb[M::P_drfp] = T_(1, 7); pb[M::P_dyfp] = T_(2, 7); pb[M::P_dcfp] = T_(3, 7);
    pb[M::P_aYFP] = T_(2, 0) * rc; pb[M::P_aCFP] = T_(3, 0) * rc; pb[M::P_a530] = T_(4, 0) * rc; pb[M::P_a480] = T_(5, 0) * rc;
  }
};

struct RlAuto {  // reference models/auto_constant.py:42-63 (observation: x, x rfp, x f530, x f480 -- :89-97)
  using M = AutoConstant;
  static constexpr int NSP = 4, Q0 = 14, NCOND = 0;
  static constexpr bool HAS_P = false, HAS_Q = false, OBS_SUM = false;
  __device__ static float source_adjoint(int, float, float) { return 0.f; }
  __device__ static void lane(int l, const float* p, RlLane& c) {
    const float rc = p[M::P_rc];
    switch (l) {
      case 1: c.deg = p[M::P_drfp]; c.F0 = rc; break;
      case 2: c.F0 = rc * p[M::P_a530]; break;
      case 3: c.F0 = rc * p[M::P_a480]; break;
      default: break;
    }
    c.gsgn = l == 0 ? -1.f : (l <= 3 ? 1.f : 0.f);
  }
  template <class TF>
  __device__ static void map(TF T_, const float* p, float* pb) {
    const float rc = p[M::P_rc];
    pb[M::P_rc] = T_(1, 0) + T_(2, 0) * p[M::P_a530] + T_(3, 0) * p[M::P_a480];
    pb[M::P_drfp] = T_(1, 7);
    pb[M::P_a530] = T_(2, 0) * rc; pb[M::P_a480] = T_(3, 0) * rc;
  }
};

// per-lane constants from theta (every lane loads the slots: cached, once per kernel)
template <class LM, bool PREC>
__device__ __forceinline__ void rl_setup(const OdeArgs& a, int i, int b, int l, RlLane& c, float* th, float* cc, float* p,
                                         float& y0, float* prec_const) {
  using M = typename LM::M;
  constexpr int NSP = LM::NSP, NIN = 1 + NSP;
#pragma unroll
  for (int q = 0; q < M::NSLOT; ++q) th[q] = a.theta[(size_t)a.slot_row[q] * a.n + i];
  float pinit[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) pinit[j] = a.theta[(size_t)a.slot_row[M::NSLOT + j] * a.n + i];  // init_prec_* or prec_*
#pragma unroll
  for (int q = 0; q < LM::NCOND; ++q) cc[q] = clampf(expf(a.cond[b * a.C + q]) - 1.f, 1e-12f, 1e6f);
  M::prepare(th, cc, p);
  float yi[NSP];
  M::init(th, cc, yi);
  c.l = l;
  c.r = p[M::P_r]; c.iKx = frcp(p[M::P_K]); c.tlag = p[M::P_tlag]; c.rKx = 0.f;
  c.gsgn = 0.f;
  c.deg = 0.f; c.F0 = 0.f; c.cP = 0.f; c.e = 0.f; c.aR = 0.f; c.aS = 0.f; c.cQ = 0.f; c.iK = 0.f; c.isrc = 0;
  c.isP = (PREC && l >= NSP && l < NSP + 4) ? 1.f : 0.f;
  y0 = 0.f;
#pragma unroll
  for (int j = 0; j < NSP; ++j) if (l == j) y0 = yi[j];
  LM::lane(l, p, c);
  c.m0 = l == 0 ? 1.f : 0.f; c.m6 = l == 6 ? 1.f : 0.f; c.m7 = l == 7 ? 1.f : 0.f;
  c.sz = LM::source_adjoint(l, 1.f, 0.f); c.sw = LM::source_adjoint(l, 0.f, 1.f);
  {
    const bool pl = l >= NSP && l < NSP + 4, ql = LM::HAS_Q && (l == LM::Q0 || l == LM::Q0 + 1);
    c.a_h = l <= NSP ? 16 + (l == NSP ? 0 : l + 1) : 44 + (l & 3);
    c.a_zx = pl ? 32 + (l - NSP) : 44 + (l & 1);
    c.a_zy = pl ? 36 + (l - NSP) : 46 + (l & 1);
    c.a_q1 = ql ? 40 + (l - LM::Q0) : 44 + (l & 1);
    c.a_q2 = ql ? 42 + (l - LM::Q0) : 46 + (l & 1);
    c.a_ix = l < 4 ? 32 + l : 44 + (l & 1);
    c.a_ip = l < 4 ? 36 + l : 46 + (l & 1);
  }
  c.n0 = (l & 3) == 0 ? 1.f : 0.f; c.n1 = (l & 3) == 1 ? 1.f : 0.f; c.n2 = (l & 3) == 2 ? 1.f : 0.f; c.n3 = (l & 3) == 3 ? 1.f : 0.f;
  c.i0 = c.m0; c.i1 = l == 1 ? 1.f : 0.f; c.i2 = (l == 2 || (LM::OBS_SUM && l == 4)) ? 1.f : 0.f;
  c.i3 = (l == 3 || (LM::OBS_SUM && l == 5)) ? 1.f : 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) prec_const[j] = pinit[j];
  c.bq = 0.f;
  c.a_hr = 16;
#pragma unroll
  for (int j = 0; j < 8; ++j) { c.wq[j] = 0.f; c.cw[j] = 0.f; }
  if (PREC) {
    // weights: Wp [4][NIN], bp [4], Wd [4][NIN], bd [4] (reference precisions.py:55-61)
    const float* w = a.weights;
    {
      const int o = (l - NSP + 4) & 7, half = ((l - NSP + 4) & 15) >> 3;  // see RlLane
      const int row0 = (o < 4 ? 0 : 4 * NIN + 4) + (o & 3) * NIN;        // first weight of output o's row
      c.a_hr = 16 + 8 * half;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = 8 * half + jj;
        const float wv = w[row0 + (j < NIN ? j : 0)];  // (unconditional load, value selected)
        c.wq[jj] = j < NIN ? wv : 0.f;
      }
      const float bv = w[(o < 4 ? 4 * NIN : 8 * NIN + 4) + (o & 3)];
      c.bq = half == 0 ? bv : 0.f;
    }
    if (l >= NSP && l < NSP + 4) {
      y0 = pinit[l - NSP];
    } else if (l < NSP) {
#pragma unroll
      for (int o = 0; o < 4; ++o) { c.cw[o] = w[o * NIN + l + 1]; c.cw[4 + o] = w[4 * NIN + 4 + o * NIN + l + 1]; }
    }
  }
}

// ---- forward -------------------------------------------------------------------------------------------------------
// dynamic LDS: times [T] | sigma table [RL_TR][(T - 1) stages] | obs rows [nb][4][T] | (THETA) the sampling stage's scratch
// THETA: the sampling stage (vihds_theta_stage.hpp) runs first, for the block's sixteen trajectories (vihds_theta_ode_fwd)
template <class LM, bool PREC, int SOLVER, bool THETA>
__device__ __forceinline__ void relay_lane_fwd_body(const OdeArgs& a, int sig_tab, const ThetaStageArgs* ts, int nb_max) {
  using M = typename LM::M;
  using Tab = RlTab<SOLVER>;
  constexpr int NSP = LM::NSP, N = PREC ? NSP + 4 : NSP;
  constexpr float OS = LM::OBS_SUM ? 1.f : 0.f;  // observed signals 2, 3: x (yfp + f530), x (cfp + f480) -- or x y2, x y3 (auto_constant)
  __shared__ __attribute__((aligned(16))) float patch[RL_TR][RL_PATCH];
  extern __shared__ float in_lds[];
  const int tid = threadIdx.x, l = tid & 15, g = tid >> 4;
  const int blk = xcd_block(blockIdx.x, gridDim.x);  // (neighbouring blocks share 128-byte lines: same XCD, same L2)
  const int i0 = blk * RL_TR + g;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  // time grid and the observation rows of the block's data rows -> LDS (no vector loads inside the time loop)
  const int first = blk * RL_TR, last = min(first + RL_TR, a.n) - 1;
  const int b0 = first / a.S, nb = last / a.S - b0 + 1;
  const int n_sg = sig_tab ? (a.T - 1) * Tab::S : 0, o_obs = a.T + RL_TR * n_sg;  // (sig_tab 0: a grid too long for the table)
  for (int q = tid; q < a.T; q += RL_T) in_lds[q] = a.times[q];
  if (a.obs) {
    const float* src = a.obs + (size_t)b0 * 4 * a.T;
    for (int q = tid; q < nb * 4 * a.T; q += RL_T) in_lds[o_obs + q] = src[q];
  }
  if constexpr (THETA) theta_stage_block<RL_T>(a, *ts, first, RL_TR, nb_max, in_lds + o_obs + nb_max * 4 * a.T);
  RlLane c;
  float th[M::NSLOT], cc[LM::NCOND > 0 ? LM::NCOND : 1], p[M::NP], y, pconst[4];
  rl_setup<LM, PREC>(a, i, b, l, c, th, cc, p, y, pconst);
  float* pt = patch[g];
  pt[16 + l] = 0.f;  // the network's input slots (those past the model's inputs stay zero: see rl_rhs)
  __syncthreads();
  const float* tl = in_lds;
  const float* ob = in_lds + o_obs + (b - b0) * 4 * a.T;
  float* sg = in_lds + a.T + g * n_sg;
  if (sig_tab) rl_sigma_table<Tab>(sg, tl, a.T, c.tlag, l);
  rl_wave_fence();
  auto sigma = [&](int q, float t) { return sig_tab ? sg[q] : sigmoid_f(4.f * (t - c.tlag)); };
  const float h0 = tl[1] - tl[0];
  const size_t n = a.n;
  float lp = 0.f;
  const int j = l & 3;  // observed signal of lanes 0..3
  const float lc = PREC ? 0.f : RL_LOG2PI - logf(pconst[j]);
  // observation of grid point k from the states as published in the patch (x_predict, log-likelihood: lanes 0..3)
  // (straight-line for all sixteen lanes -- signal j = l & 3 --; only lanes 0..3 store)
  auto observe = [&](int k) {
    const float x = pt[0];
    const float inner = c.n0 + c.n1 * pt[1] + c.n2 * (pt[2] + OS * pt[4]) + c.n3 * (pt[3] + OS * pt[5]);
    const float xp = x * inner;
    if (a.xpred && live && l < 4) a.xpred[((size_t)k * 4 + j) * n + i] = xp;
    if (a.logp) {
      const float e = xp - ob[j * a.T + k];
      if (PREC) {
        const float pr = pt[NSP + j];
        lp += -0.5f * (RL_LOG2PI - logf(pr) + pr * e * e);
      } else {
        lp += -0.5f * (lc + pconst[j] * e * e);
      }
    }
  };
  const bool obs_on = a.xpred || a.logp;
  if (a.traj && live && l < N) a.traj[(size_t)l * n + i] = y;
  for (int k = 1; k < a.T; ++k) {
    const float t0 = tl[k - 1], t1 = tl[k];
    const float h = Tab::h(t0, t1, h0);
    float kk[Tab::S];
#pragma unroll
    for (int s = 0; s < Tab::S; ++s) {
      float Y = y;
#pragma unroll
      for (int q = 0; q < s; ++q)
        if (Tab::A(s, q) != 0.f) Y += (Tab::A(s, q) * h) * kk[q];
      RlEval E;
      const float sig = sigma((k - 1) * Tab::S + s, Tab::ts(s, t0, t1));
      if (s == 0) {
        // the step's first evaluation publishes the grid point k - 1 itself: its observation rides on that exchange
        // instead of a publish / fence / read round of its own
        kk[s] = rl_rhs<LM, PREC, false>(c, Tab::ts(s, t0, t1), sig, Y, pt, E);
        if (obs_on) observe(k - 1);
        rl_wave_fence();
      } else {
        kk[s] = rl_rhs<LM, PREC>(c, Tab::ts(s, t0, t1), sig, Y, pt, E);
      }
    }
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < Tab::S; ++s)
      if (Tab::B(s) != 0.f) acc += Tab::B(s) * kk[s];
    y += h * acc;
    if (a.traj && live && l < N) a.traj[((size_t)k * N + l) * n + i] = y;
  }
  if (obs_on) {  // the last grid point
    pt[l] = y;
    rl_wave_fence();
    observe(a.T - 1);
  }
  if (a.logp && live && l < 4) a.logp[(size_t)j * n + i] = lp;
}
template <class LM, bool PREC, int SOLVER>
__global__ void __launch_bounds__(RL_T) relay_lane_fwd_kernel(OdeArgs a, int sig_tab) {
  relay_lane_fwd_body<LM, PREC, SOLVER, false>(a, sig_tab, nullptr, 0);
}
template <class LM, bool PREC, int SOLVER>
__global__ void __launch_bounds__(RL_T) relay_lane_theta_fwd_kernel(OdeArgs a, int sig_tab, int nb_max, ThetaStageArgs t) {
  relay_lane_fwd_body<LM, PREC, SOLVER, true>(a, sig_tab, &t, nb_max);
}

// ---- adjoint -------------------------------------------------------------------------------------------------------
template <class LM, bool PREC, int SOLVER>
__global__ void __launch_bounds__(RL_T) relay_lane_bwd_kernel(OdeArgs a, int sig_tab) {
  using M = typename LM::M;
  using Tab = RlTab<SOLVER>;
  constexpr int NSP = LM::NSP, N = PREC ? NSP + 4 : NSP, NIN = 1 + NSP, NWROW = rl_nwrow(NIN), NWG = rl_nwg(NIN);
  constexpr float OS = LM::OBS_SUM ? 1.f : 0.f;
  __shared__ __attribute__((aligned(16))) float patch[RL_TR][RL_PATCH];
  __shared__ float tab[RL_TR][RL_G][10];   // epilogue: per-lane accumulators of a trajectory
  __shared__ float wred[PREC ? RL_TR : 1][PREC ? NWG : 1];
  extern __shared__ float in_lds[];
  const int tid = threadIdx.x, l = tid & 15, g = tid >> 4;
  const int blk = xcd_block(blockIdx.x, gridDim.x);  // (neighbouring blocks share 128-byte lines: same XCD, same L2)
  const int i0 = blk * RL_TR + g;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;  // tail trajectories shadow the last one (they take part in the exchanges)
  const int b = i / a.S;
  const int first = blk * RL_TR, last = min(first + RL_TR, a.n) - 1;
  const int b0 = first / a.S, nb = last / a.S - b0 + 1;
  const int n_sg = sig_tab ? (a.T - 1) * Tab::S : 0, o_obs = a.T + RL_TR * n_sg;  // (sig_tab 0: a grid too long for the table)
  for (int q = tid; q < a.T; q += RL_T) in_lds[q] = a.times[q];
  {
    const float* src = a.obs + (size_t)b0 * 4 * a.T;
    for (int q = tid; q < nb * 4 * a.T; q += RL_T) in_lds[o_obs + q] = src[q];
  }
  RlLane c;
  float pconst[4];
  {  // (theta and the prepared parameters are not kept across the time loop: the epilogue's one lane fetches them again)
    float th[M::NSLOT], cc[LM::NCOND > 0 ? LM::NCOND : 1], p[M::NP], y_unused;
    rl_setup<LM, PREC>(a, i, b, l, c, th, cc, p, y_unused, pconst);
  }
  float* pt = patch[g];
  pt[16 + l] = 0.f;  // the network's input slots (those past the model's inputs stay zero: see rl_rhs)
  __syncthreads();
  const float* tl = in_lds;
  const float* ob = in_lds + o_obs + (b - b0) * 4 * a.T;
  float* sg = in_lds + a.T + g * n_sg;
  if (sig_tab) rl_sigma_table<Tab>(sg, tl, a.T, c.tlag, l);
  rl_wave_fence();
  auto sigma = [&](int q, float t) { return sig_tab ? sg[q] : sigmoid_f(4.f * (t - c.tlag)); };
  const float h0 = tl[1] - tl[0];
  const size_t n = a.n;
  const int j = l & 3;
  RlAcc A;
  A.F0b = A.cPb = A.eb = A.aRb = A.aSb = A.cQb = A.iKb = A.degb = A.rb = A.Kb = A.tlagb = 0.f;
  A.b2p = A.b2d = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) A.wcp[q] = A.wcd[q] = 0.f;
  float lam = 0.f, precb = 0.f;
  const float w_iw = a.iw_logp ? iw_wave_weight(a, i, b) : 0.f;
  const float glp = l < 4 ? ode_logp_grad(a, w_iw, i, j) : 0.f;
  const int lr = l < N ? l : 0;  // row this lane reads from the stored trajectory
  float yn = a.traj_in[((size_t)(a.T - 1) * N + lr) * n + i];
  float ox = 0.f, oy1 = 0.f, oy2 = 0.f, oy3 = 0.f, oy4 = 0.f, oy5 = 0.f, opr = 1.f;  // the grid point's observed states
  auto grab = [&]() {
    ox = pt[0]; oy1 = pt[1]; oy2 = pt[2]; oy3 = pt[3]; oy4 = pt[4]; oy5 = pt[5];
    if (PREC) opr = pt[NSP + j];
  };
  for (int k = a.T - 1; k >= 0; --k) {
    const float y = yn;
    if (k > 0) yn = a.traj_in[((size_t)(k - 1) * N + lr) * n + i];  // (one step ahead of its use)
    if (k < a.T - 1) {
      // reverse of step k -> k+1: recompute the stage states, then pull lam back through the stages
      const float t0 = tl[k], t1 = tl[k + 1];
      const float h = Tab::h(t0, t1, h0);
      if constexpr (Tab::S <= 2) {
        // one or two stages: every stage is evaluated ONCE (its forward quantities kept for its own VJP)
        RlEval E0, E1;
        // (the step's first evaluation publishes the grid point k itself: the injection below takes what it needs of it here)
        const float k0 = rl_rhs<LM, PREC, false>(c, Tab::ts(0, t0, t1), sigma(k * Tab::S, Tab::ts(0, t0, t1)), y, pt, E0);
        grab();
        rl_wave_fence();
        float Y1 = y;
        if constexpr (Tab::S == 2) {
          Y1 = y + (Tab::A(1, 0) * h) * k0;
          (void)rl_rhs<LM, PREC>(c, Tab::ts(1, t0, t1), sigma(k * Tab::S + 1, Tab::ts(1, t0, t1)), Y1, pt, E1);
          float kb0 = (Tab::B(0) * h) * lam;
          const float Yb1 = rl_vjp_core<LM, PREC>(c, Y1, (Tab::B(1) * h) * lam, E1, pt, A);
          lam += Yb1;
          kb0 += (Tab::A(1, 0) * h) * Yb1;
          lam += rl_vjp_core<LM, PREC>(c, y, kb0, E0, pt, A);
        } else {
          lam += rl_vjp_core<LM, PREC>(c, y, (Tab::B(0) * h) * lam, E0, pt, A);
        }
      } else {
        float Ys[Tab::S], kk[Tab::S], kb[Tab::S];
#pragma unroll
        for (int s = 0; s < Tab::S; ++s) {
          float Y = y;
#pragma unroll
          for (int q = 0; q < s; ++q)
            if (Tab::A(s, q) != 0.f) Y += (Tab::A(s, q) * h) * kk[q];
          Ys[s] = Y;
          if (s + 1 < Tab::S) {  // (the last stage's derivative is not needed to rebuild the states)
            RlEval E;
            if (s == 0) {
              kk[s] = rl_rhs<LM, PREC, false>(c, Tab::ts(s, t0, t1), sigma(k * Tab::S + s, Tab::ts(s, t0, t1)), Y, pt, E);
              grab();
              rl_wave_fence();
            } else {
              kk[s] = rl_rhs<LM, PREC>(c, Tab::ts(s, t0, t1), sigma(k * Tab::S + s, Tab::ts(s, t0, t1)), Y, pt, E);
            }
          }
        }
#pragma unroll
        for (int s = 0; s < Tab::S; ++s) kb[s] = (Tab::B(s) * h) * lam;
#pragma unroll
        for (int s = Tab::S - 1; s >= 0; --s) {
          const float Yb = rl_rhs_vjp<LM, PREC>(c, Tab::ts(s, t0, t1), sigma(k * Tab::S + s, Tab::ts(s, t0, t1)), Ys[s], kb[s], pt, A);
          lam += Yb;
#pragma unroll
          for (int q = 0; q < s; ++q)
            if (Tab::A(s, q) != 0.f) kb[q] += (Tab::A(s, q) * h) * Yb;
        }
      }
    }
    // gradient injected at time k: log-likelihood, x_predict and trajectory upstream gradients (the states of the grid
    // point were taken from the step's first exchange; the last grid point has no step: publish it here)
    if (k == a.T - 1) {
      pt[l] = y;
      rl_wave_fence();
      grab();
      rl_wave_fence();
    }
    const float x = ox;
    // (straight-line for all lanes -- glp is zero outside lanes 0..3 --, only the two stores are the four lanes')
    const float inner = c.n0 + c.n1 * oy1 + c.n2 * (oy2 + OS * oy4) + c.n3 * (oy3 + OS * oy5);
    const float e = x * inner - ob[j * a.T + k];
    const float pr = PREC ? opr : pconst[j];
    float xpb = -glp * pr * e;
    const float prb = glp * (0.5f * frcp(pr) - 0.5f * e * e);
    if (a.g_xpred) xpb += a.g_xpred[((size_t)k * 4 + j) * n + i];
    pt[c.a_ix] = xpb;
    pt[c.a_ip] = prb;
    if (!PREC) precb += prb;
    const float y1 = oy1, y2 = oy2, y3 = oy3, y4 = oy4, y5 = oy5;
    rl_wave_fence();
    const float4 xb4 = *reinterpret_cast<const float4*>(pt + 32);
    float inj = c.i0 * (xb4.x + xb4.y * y1 + xb4.z * (y2 + OS * y4) + xb4.w * (y3 + OS * y5)) + x * (c.i1 * xb4.y + c.i2 * xb4.z + c.i3 * xb4.w);
    if (PREC) inj = fmaf(c.isP, pt[36 + ((l - NSP) & 3)], inj);
    lam += inj;
    if (a.g_traj && l < N) lam += a.g_traj[((size_t)k * N + l) * n + i];
    rl_wave_fence();
  }
  // ---- epilogue: per-lane constants' adjoints -> adjoints of the model's prepared parameters -> theta ---------------
  {
    float* r = tab[g][l];
    r[0] = A.F0b; r[1] = A.cPb; r[2] = A.eb; r[3] = A.aRb; r[4] = A.aSb; r[5] = A.cQb; r[6] = A.iKb; r[7] = A.degb;
    r[8] = lam;
    r[9] = precb;
  }
  rl_wave_fence();
  if (l == 0) {
    float th[M::NSLOT], cc[LM::NCOND > 0 ? LM::NCOND : 1], p[M::NP], pb[M::NP], thb[M::NSLOT], lam0[NSP];
#pragma unroll
    for (int q = 0; q < M::NSLOT; ++q) th[q] = a.theta[(size_t)a.slot_row[q] * a.n + i];
#pragma unroll
    for (int q = 0; q < LM::NCOND; ++q) cc[q] = clampf(expf(a.cond[b * a.C + q]) - 1.f, 1e-12f, 1e6f);
    M::prepare(th, cc, p);
#pragma unroll
    for (int q = 0; q < M::NP; ++q) pb[q] = 0.f;
    auto T_ = [&](int lane, int f) { return tab[g][lane][f]; };
    rl_acc_finish(c, A);
    pb[M::P_r] = A.rb; pb[M::P_K] = A.Kb; pb[M::P_tlag] = A.tlagb;
    LM::map(T_, p, pb);
#pragma unroll
    for (int q = 0; q < M::NSLOT; ++q) thb[q] = 0.f;
    M::prepare_vjp(th, cc, p, pb, thb);
#pragma unroll
    for (int q = 0; q < NSP; ++q) lam0[q] = T_(q, 8);
    M::init_vjp(lam0, thb);
    if (live) {
#pragma unroll
      for (int q = 0; q < M::NSLOT; ++q) a.g_theta[(size_t)a.slot_row[q] * n + i] = thb[q];
#pragma unroll
      for (int q = 0; q < 4; ++q)  // init_prec_* (neural: the adjoint of the initial precision state) or prec_*
        a.g_theta[(size_t)a.slot_row[M::NSLOT + q] * n + i] = PREC ? T_(NSP + q, 8) : T_(q, 9);
    }
  }
  if (PREC) {
    // weight gradients: rows of the precision lanes, summed over the block's trajectories in trajectory order
    if (l <= NSP) {  // column j of both matrices, rows o = 0..3
      const int jc = l == NSP ? 0 : l + 1;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        wred[g][o * NWROW + jc] = live ? A.wcp[o] : 0.f;
        wred[g][o * NWROW + NIN + jc] = live ? A.wcd[o] : 0.f;
      }
    }
    if (l >= NSP && l < NSP + 4) {
      float* w = wred[g] + (l - NSP) * NWROW;
      w[2 * NIN] = live ? A.b2p : 0.f;
      w[2 * NIN + 1] = live ? A.b2d : 0.f;
    }
    __syncthreads();
    if (tid < NWG && a.aux) {
      float acc = 0.f;
#pragma unroll
      for (int q = 0; q < RL_TR; ++q) acc += wred[q][tid];
      a.aux[(size_t)blk * NWG + tid] = acc;
    }
  }
}

// partial rows [nblocks][4][2*13+2] (rows: Wp row, Wd row, bp, bd of output o) -> ADDED into g_weights
// (Wp [4][13], bp [4], Wd [4][13], bd [4]); blocks in order
// One wavefront per element: lane q adds rows q, q + 64, ... (all requested together), then the lanes' sums are added by a
// DPP scan -- a fixed order.
static __global__ void __launch_bounds__(256) relay_lane_wreduce_kernel(const float* __restrict__ partial, int nblocks,
                                                                        float* __restrict__ g_weights, int NIN) {
  const int NWROW = rl_nwrow(NIN), NWG = rl_nwg(NIN);
  const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (e >= NWG) return;
  float acc = 0.f;
  for (int b0 = 0; b0 < nblocks; b0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = partial[(size_t)min(b0 + lane + 64 * q, nblocks - 1) * NWG + e];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (b0 + lane + 64 * q < nblocks) acc += v[q];
  }
  // inclusive scan over the wavefront (row_shr 1, 2, 4, 8, row_bcast 15, 31): lane 63 holds the total
#define RL_SCAN(CTRL, RM) acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), CTRL, RM, 0xf, false))
  RL_SCAN(0x111, 0xf); RL_SCAN(0x112, 0xf); RL_SCAN(0x114, 0xf); RL_SCAN(0x118, 0xf); RL_SCAN(0x142, 0xa); RL_SCAN(0x143, 0xc);
#undef RL_SCAN
  if (lane != 63) return;
  const int o = e / NWROW, q = e - o * NWROW;
  int dst;
  if (q < NIN) dst = o * NIN + q;                                 // Wp[o][q]
  else if (q < 2 * NIN) dst = 4 * NIN + 4 + o * NIN + (q - NIN);  // Wd[o][.]
  else if (q == 2 * NIN) dst = 4 * NIN + o;                       // bp[o]
  else dst = 8 * NIN + 4 + o;                                     // bd[o]
  g_weights[dst] += acc;
}

// ---- launch ---------------------------------------------------------------------------------------------------------
template <class LM, bool PREC, int SOLVER>
inline int relay_lanes_launch_s(bool backward, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts = nullptr) {
  const int nblk = (a.n + RL_TR - 1) / RL_TR;
  const int nb_max = min(a.B, (RL_TR - 1) / a.S + 2);
  const size_t lds_in = sizeof(float) * ((size_t)a.T + (size_t)nb_max * 4 * a.T);
  const size_t lds_sg = sizeof(float) * (size_t)RL_TR * (a.T - 1) * RlTab<SOLVER>::S;
  const int sig_tab = lds_in + lds_sg <= 48 * 1024 ? 1 : 0;  // (beyond that the stage sigmoids are evaluated in place)
  const size_t lds = lds_in + (sig_tab ? lds_sg : 0);
  constexpr int NIN = 1 + LM::NSP;
  if (ts) {  // forward with the sampling stage in front (vihds_theta_ode_fwd)
    const size_t lds_t = lds + sizeof(float) * theta_stage_lds_floats(nb_max, ts->P, RL_TR);
    if (backward || lds_t > 60 * 1024) return VIHDS_E_UNSUPPORTED;
    hipLaunchKernelGGL((relay_lane_theta_fwd_kernel<LM, PREC, SOLVER>), dim3(nblk), dim3(RL_T), lds_t, st, a, sig_tab, nb_max, *ts);
    return VIHDS_OK;
  }
  if (!backward) {
    hipLaunchKernelGGL((relay_lane_fwd_kernel<LM, PREC, SOLVER>), dim3(nblk), dim3(RL_T), lds, st, a, sig_tab);
  } else {
    hipLaunchKernelGGL((relay_lane_bwd_kernel<LM, PREC, SOLVER>), dim3(nblk), dim3(RL_T), lds, st, a, sig_tab);
    if (PREC && a.g_weights && a.aux)
      hipLaunchKernelGGL(relay_lane_wreduce_kernel, dim3((rl_nwg(NIN) + 3) / 4), dim3(256), 0, st, a.aux, nblk, a.g_weights, NIN);
  }
  return VIHDS_OK;
}
template <class LM, bool PREC>
inline int relay_lanes_launch(bool backward, int solver, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts = nullptr) {
  switch (solver) {
    case VIHDS_SOLVER_MODEULER: return relay_lanes_launch_s<LM, PREC, VIHDS_SOLVER_MODEULER>(backward, a, st, ts);
    case VIHDS_SOLVER_MODEULERWHILE: return relay_lanes_launch_s<LM, PREC, VIHDS_SOLVER_MODEULERWHILE>(backward, a, st, ts);
    case VIHDS_SOLVER_EULER: return relay_lanes_launch_s<LM, PREC, VIHDS_SOLVER_EULER>(backward, a, st, ts);
    case VIHDS_SOLVER_MIDPOINT: return relay_lanes_launch_s<LM, PREC, VIHDS_SOLVER_MIDPOINT>(backward, a, st, ts);
    case VIHDS_SOLVER_RK4: return relay_lanes_launch_s<LM, PREC, VIHDS_SOLVER_RK4>(backward, a, st, ts);
  }
  return VIHDS_E_BADARG;
}

}  // namespace vihds
