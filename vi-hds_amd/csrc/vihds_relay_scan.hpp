// Time-parallel ("scan") forward and adjoint kernels for the white-box models in lane-model form -- relay_constant,
// degrader_constant, prpr_constant, auto_constant and their _precisions forms (reference models/relay_constant.py:91-134,
// degrader_constant.py:103-143, prpr_constant.py:46-69, auto_constant.py:42-63; vihds/precisions.py:55-61,76-87;
// BASELINE config 5 = relay_constant_precisions) -- kernel_variant 5.
//
// The lane kernels (vihds_relay_lanes.hpp) walk the T-1 steps of a trajectory serially in every wavefront: 1 800 wavefronts
// on 1 024 SIMDs, each a dependent stream of ~450 instructions per adjoint step (3.3 % / 8.6 % of the HBM
// roofline).  But every state of these models obeys
//
//     dy_j = F_j(t, lower states) - a_j(t, lower states) y_j ,
//
// with the states in dependency LEVELS (LM::level): 0 = OD, a scalar logistic chain; 1 = diluted species with a constant
// production (rfp, f530, f480, luxR, lasR; degrader's aiiA); 2 = promoter-driven species, forced through luxR / lasR (yfp,
// cfp, luxI, lasI); 3 = the AHL quadratures of x I / (1 + I / K); 4 = the four neural precision states, whose production /
// degradation rates sigma(W tanh[t, species]) read every species and none of the precisions.  Given the levels below it,
// one explicit Runge-Kutta step of a state is an AFFINE map y' = A_k y + B_k, the trajectory of a level is a prefix scan
// over k, and the discrete adjoint is a reverse scan level by level from the top (csrc/vihds_dr_scan.hpp does this for
// dr_constant with the per-step records in LDS).
//
// Mapping: 32 lanes per trajectory, lane l owns the steps k = l*ITEMS .. l*ITEMS+ITEMS-1 (ITEMS = ceil((T-1)/32), a run-time
// loop bound), two trajectories per wavefront, NO workgroup barrier: a wavefront never waits for another one.  The record
// that carries a level's result to the next level is the TRAJECTORY ITSELF, in the layout [B][S][N][T] (vihds_ode_problem
// kernel_variant 5: time fastest -- the reference's own logical layout, ode.py:82 -- so that a wavefront's stores and
// loads of one state are 32 consecutive floats per trajectory); a lane only ever reads back what itself wrote.  Per-step
// adjoint records (injection offsets of the twelve species, gamma adjoints, x injections, observation injections: 20
// floats per step with a two-stage scheme) live in LDS.
//
//   forward  relay_scan_fwd_kernel : x chain (serial, both trajectories of the wavefront side by side) -> per level: maps of
//            this lane's steps, DPP scan over the 32 lanes, the steps themselves -> trajectory; then x_predict and the
//            log-likelihood at the grid points.
//   adjoint  relay_scan_bwd_kernel : observation injections -> per level from the top: reverse scan of Lambda, then per step
//            the VJP of the level's states, whose stage injections into lower levels are folded AT ONCE into those levels'
//            per-step offsets (everything is linear in the adjoints) -> the per-state constants' adjoints -> the same epilogue
//            as the lane kernels (LM::map, prepare_vjp, init_vjp).
//   weights  relay_scan_wgrad_kernel : the precision network's 8 x (1 + species) + 8 weight gradients from the Lambda of
//            the precision states the adjoint left in `aux` (112 accumulators per lane for relay: kept out of the adjoint
//            kernel's register budget), one partial row per block, summed in block order by relay_lane_wreduce_kernel.
//
// Same arithmetic as the lane kernels regrouped (rounding-level differences); parity: tests/test_config5_parity.py (the
// MODIFIED reference's fixtures and the oracle, variant 5 beside 0 and 1), tests/test_hip_parity.py.
#pragma once
#include <hip/hip_runtime.h>

#include "vihds_ode_kernels.hpp"
#include "vihds_dr_scan.hpp"
#include "vihds_relay_lanes.hpp"
#include "vihds_relay_scan_api.hpp"

namespace vihds {

template <class LM, bool PREC, int SOLVER>
struct Rs {
  using M = typename LM::M;
  using R = Rk<SOLVER>;
  static constexpr int NSP = LM::NSP, N = PREC ? NSP + 4 : NSP, NS = R::NS, NIN = 1 + NSP;
  static constexpr float OS = LM::OBS_SUM ? 1.f : 0.f;
  static constexpr int TOP = PREC ? 4 : (LM::HAS_Q ? 3 : (LM::HAS_P ? 2 : 1));
  static constexpr int level(int j) { return j >= NSP ? 4 : LM::level(j); }
  static constexpr float gs(int j) { return (j >= 1 && j < NSP && LM::level(j) <= 2) ? 1.f : 0.f; }  // dilution by gamma
  enum { C_F0, C_cP, C_e, C_aR, C_aS, C_cQ, C_iK, C_deg, NCT };
  static constexpr int REC = 12 + 2 * NS + 4 + (PREC ? 4 : 0);  // adjoint: LDS floats per step and lane

  struct Tj {  // what the 32 lanes of a trajectory share
    float r, iK, tlag, h0;
    const float* ct;  // LDS [16][NCT]: the states' constants (RlLane's fields)
    const float* w;   // precision network: Wp [4][NIN], bp [4], Wd [4][NIN], bd [4]
    const float* tT;  // LDS: time grid
    int K;            // steps
  };
  struct Step {
    int kc;
    bool valid, last;
    float h, t0, dt;
    float sg[NS], gr[NS], x[NS], u[NS], gam[NS];
  };
  // stage values of x from the grid value (dx = gamma x, gamma = gr (1 - x / K)) and everything the other levels read of it
  __device__ __forceinline__ static void make_step(const Tj& tj, int k, float xk, Step& st) {
    st.valid = k < tj.K;
    st.kc = st.valid ? k : tj.K - 1;
    st.last = k == tj.K - 1;
    st.t0 = tj.tT[st.kc];
    st.dt = tj.tT[st.kc + 1] - st.t0;
    st.h = R::FIXED_H ? tj.h0 : st.dt;
    float kk[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      st.sg[s] = sigmoid_f(4.f * (fmaf(R::c(s), st.dt, st.t0) - tj.tlag));
      st.gr[s] = tj.r * st.sg[s];
      float v = xk;
      VIHDS_UNROLL for (int q = 0; q < s; ++q)
        if (R::a(s, q) != 0.f) v = fmaf(st.h * R::a(s, q), kk[q], v);
      st.x[s] = v;
      st.u[s] = v * tj.iK;
      st.gam[s] = st.gr[s] * (1.f - st.u[s]);
      kk[s] = st.gam[s] * v;
    }
  }
  __device__ __forceinline__ static float x_next(const Step& st, float xk) {
    float o = xk;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (R::b(s) != 0.f) o = fmaf(st.h * R::b(s), st.gam[s] * st.x[s], o);
    return o;
  }

  struct Ev {  // one step's forward quantities; whatever an instantiation does not read is never computed
    float Y[N][NS], a[N][NS], F[N][NS], nxt[N];
    float P[2][NS], rden[2][NS], R2[NS], S2[NS];
    float Q[2][NS], rdq[2][NS];
    float hin[NIN][NS], sp[4][NS], sd[4][NS];
  };
  // coefficients (a, F) of the states of levels 1..UPTO and the stage values of the levels below UPTO (of UPTO itself with
  // TOPSTAGES), from the grid values yk of the step's first point
  template <int UPTO, bool TOPSTAGES>
  __device__ __forceinline__ static void eval(const Tj& tj, const Step& st, const float* yk, Ev& e) {
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) e.Y[0][s] = st.x[s];
    static_for<1, UPTO + 1>([&](auto LL) {
      constexpr int L = decltype(LL)::value;
      if constexpr (L == 2 && LM::HAS_P) {
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          e.R2[s] = e.Y[6][s] * e.Y[6][s];
          e.S2[s] = e.Y[7][s] * e.Y[7][s];
          VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
            const float* c = tj.ct + (2 + q) * NCT;  // (the promoters' constants: lanes 2 and 3)
            const float aa = fmaf(c[C_aR], e.R2[s], c[C_aS] * e.S2[s]);
            e.rden[q][s] = frcp(1.f + aa);
            e.P[q][s] = (c[C_e] + aa) * e.rden[q][s];
          }
        }
      }
      if constexpr (L == 3 && LM::HAS_Q) {
        static_for<0, 2>([&](auto QQ) {
          constexpr int q = decltype(QQ)::value, j = LM::Q0 + q, src = LM::qsrc(j);
          VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
            const float I = e.Y[src][s];
            e.rdq[q][s] = frcp(fmaf(I, tj.ct[j * NCT + C_iK], 1.f));
            e.Q[q][s] = st.x[s] * I * e.rdq[q][s];
          }
        });
      }
      if constexpr (L == 4 && PREC) {
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          e.hin[0][s] = ftanh(fmaf(R::c(s), st.dt, st.t0));
          VIHDS_UNROLL for (int j = 0; j < NSP; ++j) e.hin[1 + j][s] = ftanh(e.Y[j][s]);
          VIHDS_UNROLL for (int o = 0; o < 4; ++o) {
            float zp = tj.w[4 * NIN + o], zd = tj.w[8 * NIN + 4 + o];
            VIHDS_UNROLL for (int q = 0; q < NIN; ++q) {
              zp = fmaf(tj.w[o * NIN + q], e.hin[q][s], zp);
              zd = fmaf(tj.w[4 * NIN + 4 + o * NIN + q], e.hin[q][s], zd);
            }
            e.sp[o][s] = sigmoid_f(zp);
            e.sd[o][s] = sigmoid_f(zd);
          }
        }
      }
      static_for<1, N>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (level(j) == L) {
          const float* c = tj.ct + (j < NSP ? j : 0) * NCT;
          VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
            if constexpr (L == 4) {
              e.a[j][s] = e.sd[j - NSP][s];
              e.F[j][s] = e.sp[j - NSP][s];
            } else {
              e.a[j][s] = fmaf(gs(j), st.gam[s], c[C_deg]);
              float f = c[C_F0];
              if constexpr (L == 2) f = fmaf(c[C_cP], e.P[LM::prom(j)][s], f);
              if constexpr (L == 3) f = fmaf(c[C_cQ], e.Q[j - LM::Q0][s], f);
              e.F[j][s] = f;
            }
          }
          if constexpr (L < UPTO || TOPSTAGES) e.nxt[j] = R::real(st.h, e.a[j], e.F[j], yk[j], e.Y[j]);
        }
      });
    });
  }
  // the map y' = A y + B of state j for this step (identity on the padding steps)
  template <int j>
  __device__ __forceinline__ static Aff map_of(const Step& st, const Ev& e) {
    float A, B;
    R::affine(st.h, e.a[j], e.F[j], A, B);
    Aff m = {A, B};
    if (!st.valid) m = {1.f, 0.f};
    return m;
  }
  // multiplier of the reverse recurrence (the same A, without the forcing)
  template <int j>
  __device__ __forceinline__ static float mult_of(const Step& st, const Ev& e) {
    float Fz[NS], A, B;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) Fz[s] = 0.f;
    R::affine(st.h, e.a[j], Fz, A, B);
    return A;
  }
};

// per-trajectory set-up shared by the three kernels: the lanes of a half-wave build the sixteen states' constants into LDS
// (RlLane's fields through rl_setup: the lane kernels' own code), returns the growth constants
template <class LM, bool PREC, int SOLVER>
__device__ __forceinline__ void rs_setup(const OdeArgs& a, int i, int b, int l, float* ct, float* y0tab, float* tT,
                                         typename Rs<LM, PREC, SOLVER>::Tj& tj, float* pconst) {
  using S_ = Rs<LM, PREC, SOLVER>;
  using M = typename LM::M;
  RlLane c;
  float th[M::NSLOT], cc[LM::NCOND > 0 ? LM::NCOND : 1], p[M::NP], y0;
  rl_setup<LM, PREC>(a, i, b, l & 15, c, th, cc, p, y0, pconst);
  if (l < 16) {
    float* r = ct + l * S_::NCT;
    r[S_::C_F0] = c.F0; r[S_::C_cP] = c.cP; r[S_::C_e] = c.e; r[S_::C_aR] = c.aR; r[S_::C_aS] = c.aS; r[S_::C_cQ] = c.cQ;
    r[S_::C_iK] = c.iK; r[S_::C_deg] = c.deg;
    y0tab[l] = y0;
  }
  tj.r = c.r; tj.iK = c.iKx; tj.tlag = c.tlag;
  tj.ct = ct; tj.w = a.weights; tj.tT = tT; tj.K = a.T - 1;
  tj.h0 = 0.f;
}

// ---- forward ---------------------------------------------------------------------------------------------------------
// dynamic LDS (floats): tT [T + 1 padded to 4] | per trajectory: ct [16][8] | y0 [16] | xk [32 ITEMS + 4] | vm [32 ITEMS][8] (PREC)
template <class LM, bool PREC, int SOLVER>
__host__ __device__ inline size_t relay_scan_fwd_lds_floats(int T) {
  const int items = relay_scan_items(T), slots = 32 * items;
  return (size_t)((T + 4) & ~3) + (size_t)RS_TPB * (16 * 8 + 16 + slots + 4 + (PREC ? slots * 8 : 0));
}
template <class LM, bool PREC, int SOLVER>
__global__ void __launch_bounds__(RS_T) relay_scan_fwd_kernel(OdeArgs a) {
  using S_ = Rs<LM, PREC, SOLVER>;
  using R = Rk<SOLVER>;
  constexpr int NSP = S_::NSP, N = S_::N, NS = S_::NS, TOP = S_::TOP;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, tib = tid >> 5;
  const int i0 = blockIdx.x * RS_TPB + tib;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int T = a.T, K = T - 1, ITEMS = relay_scan_items(T), slots = 32 * ITEMS;
  const int TP = (T + 4) & ~3;
  float* tT = lds;
  float* mine = lds + TP + (size_t)tib * (16 * 8 + 16 + slots + 4 + (PREC ? slots * 8 : 0));
  float* ct = mine;
  float* y0t = ct + 16 * 8;
  float* xk = y0t + 16;
  float* vm = xk + slots + 4;
  for (int q = lane; q < T; q += 64) tT[q] = a.times[q];  // (every wavefront writes the same values: no block barrier)
  typename S_::Tj tj;
  float pconst[4];
  rs_setup<LM, PREC, SOLVER>(a, i, b, l, ct, y0t, tT, tj, pconst);
  wave_sync();
  tj.h0 = tT[1] - tT[0];
  const int k0 = l * ITEMS;
  float* trj = a.traj + (size_t)i * N * T;   // [N][T] of this trajectory
  const bool owner_last = (K - 1) / ITEMS == l;

  // ---- level 0: the x chain, serially (every lane of the half-wave walks it; the grid values go to LDS) ----------------
  {
    float x = y0t[0];
    VIHDS_ROLLED for (int k = 0; k < K; ++k) {
      typename S_::Step st;
      S_::make_step(tj, k, x, st);
      xk[k] = x;
      x = S_::x_next(st, x);
    }
    xk[K] = x;
    wave_sync();
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m)
      if (live && k0 + m < K) trj[k0 + m] = xk[k0 + m];
    if (live && owner_last) trj[K] = xk[K];
  }
  // grid values of the lower states at the first point of step kc (what this lane itself stored)
  auto load_lower = [&](int kc, int below, float* yk) {
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if (S_::level(j) < below) yk[j] = trj[(size_t)j * T + kc];
    });
  };
  // ---- levels 1 .. TOP -----------------------------------------------------------------------------------------------
  static_for<1, TOP + 1>([&](auto LL) {
    constexpr int L = decltype(LL)::value;
    Aff lm[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) lm[j] = {1.f, 0.f};
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      typename S_::Step st;
      S_::make_step(tj, k0 + m, xk[min(k0 + m, K - 1)], st);
      float yk[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) yk[j] = 0.f;
      load_lower(st.kc, L, yk);
      typename S_::Ev e;
      S_::template eval<L, false>(tj, st, yk, e);
      static_for<1, N>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (S_::level(j) == L) {
          const Aff mp = S_::template map_of<j>(st, e);
          lm[j] = after(mp, lm[j]);
          if constexpr (L == 4) { vm[(m * 32 + l) * 8 + (j - NSP)] = mp.a; vm[(m * 32 + l) * 8 + 4 + (j - NSP)] = mp.b; }
        }
      });
    }
    float cur[N];
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if constexpr (S_::level(j) == L) {
        const float y0 = j < NSP ? y0t[j] : pconst[(j - NSP) & 3];
        const Aff sc = scan_up32(lm[j], lane);
        const float end = fmaf(sc.a, y0, sc.b);
        const float prev = lane_read(end, lane - 1);
        cur[j] = l == 0 ? y0 : prev;
      }
    });
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      const bool valid = k0 + m < K;
      if constexpr (L == 4) {
        static_for<NSP, N>([&](auto JJ) {
          constexpr int j = decltype(JJ)::value;
          if (live && valid) trj[(size_t)j * T + k0 + m] = cur[j];
          cur[j] = fmaf(vm[(m * 32 + l) * 8 + (j - NSP)], cur[j], vm[(m * 32 + l) * 8 + 4 + (j - NSP)]);
        });
      } else {
        typename S_::Step st;
        S_::make_step(tj, k0 + m, xk[min(k0 + m, K - 1)], st);
        float yk[N];
        VIHDS_UNROLL for (int j = 0; j < N; ++j) yk[j] = 0.f;
        load_lower(st.kc, L, yk);
        static_for<1, N>([&](auto JJ) {
          constexpr int j = decltype(JJ)::value;
          if constexpr (S_::level(j) == L) {
            yk[j] = cur[j];
            if (live && valid) trj[(size_t)j * T + k0 + m] = cur[j];
          }
        });
        typename S_::Ev e;
        S_::template eval<L, true>(tj, st, yk, e);
        static_for<1, N>([&](auto JJ) {
          constexpr int j = decltype(JJ)::value;
          if constexpr (S_::level(j) == L) cur[j] = valid ? e.nxt[j] : cur[j];
        });
      }
    }
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if constexpr (S_::level(j) == L)
        if (live && owner_last) trj[(size_t)j * T + K] = cur[j];
    });
  });
  // ---- x_predict and the log-likelihood at this lane's grid points (+ the last one in the lane that owns step K-1) ------
  if (a.xpred || a.logp) {
    const float* ob = a.obs ? a.obs + (size_t)b * 4 * T : nullptr;
    float lp[4] = {0.f, 0.f, 0.f, 0.f};
    auto point = [&](int k, bool on) {
      const int kc = min(k, K);
      const float x = trj[kc];
      float y[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      VIHDS_UNROLL for (int j = 1; j < 6; ++j)
        if (j < NSP) y[j] = trj[(size_t)j * T + kc];
      float xp[4];
      xp[0] = x;
      xp[1] = x * y[1];
      xp[2] = x * (y[2] + S_::OS * y[4]);
      xp[3] = x * (y[3] + S_::OS * y[5]);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        if (a.xpred && live && on) a.xpred[((size_t)i * 4 + j) * T + kc] = xp[j];
        if (a.logp) {
          const float pr = PREC ? trj[(size_t)(NSP + j) * T + kc] : pconst[j];
          const float e = xp[j] - ob[j * T + kc];
          lp[j] += on ? -0.5f * (RL_LOG2PI - logf(pr) + pr * e * e) : 0.f;
        }
      }
    };
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) point(k0 + m, k0 + m < K);
    point(K, owner_last);
    if (a.logp) {
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float tot = sum32(lp[j], lane);
        if (live && l == 0) a.logp[(size_t)j * a.n + i] = tot;
      }
    }
  }
}

// ---- adjoint ---------------------------------------------------------------------------------------------------------
// reverse recurrence helpers: Lambda_k = A_k (Lambda_{k+1} + post_k) + c_k, post = the terminal injection at k = K-1
__device__ __forceinline__ void rs_lane_step(Aff& lm, bool valid, bool last, float A, float c, float gK) {
  Aff st = {A, fmaf(A, last ? gK : 0.f, c)};
  if (!valid) st = {1.f, 0.f};
  lm = after(st, lm);
}
__device__ __forceinline__ float rs_lane_entry(const Aff& lm, int lane) {
  const Aff sc = scan_down32(lm, lane);
  const float nxt = lane_read(sc.b, lane + 1);
  return (lane & 31) == 31 ? 0.f : nxt;
}

// dynamic LDS (floats): tT | per trajectory: ct [16][8] | y0 [16] | rec [32 ITEMS][REC] (lane-major within a step slot)
template <class LM, bool PREC, int SOLVER>
__host__ __device__ inline size_t relay_scan_bwd_lds_floats(int T) {
  const int items = relay_scan_items(T), slots = 32 * items;
  return (size_t)((T + 4) & ~3) + (size_t)RS_TPB * (16 * 8 + 16 + (size_t)slots * Rs<LM, PREC, SOLVER>::REC);
}

template <class LM, bool PREC, int SOLVER>
__global__ void __launch_bounds__(RS_T, 2) relay_scan_bwd_kernel(OdeArgs a) {
  using S_ = Rs<LM, PREC, SOLVER>;
  using M = typename LM::M;
  using R = Rk<SOLVER>;
  constexpr int NSP = S_::NSP, N = S_::N, NS = S_::NS, NIN = S_::NIN, TOP = S_::TOP, REC = S_::REC;
  constexpr float OS = S_::OS;
  // record fields of a step: CS[12] injection-driven offsets of the species (slot 0: x's grid injection), GB[NS] gamma
  // adjoints from the other states, JX[NS] stage injections into x, Q[4] observation injections, PV[4] precision injections
  constexpr int O_CS = 0, O_GB = 12, O_JX = 12 + NS, O_Q = 12 + 2 * NS, O_PV = 12 + 2 * NS + 4;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, tib = tid >> 5;
  const int i0 = blockIdx.x * RS_TPB + tib;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int T = a.T, K = T - 1, ITEMS = relay_scan_items(T), slots = 32 * ITEMS;
  const int TP = (T + 4) & ~3;
  const size_t n = a.n;
  float* tT = lds;
  float* mine = lds + TP + (size_t)tib * (16 * 8 + 16 + (size_t)slots * REC);
  float* ct = mine;
  float* y0t = ct + 16 * 8;
  float* recs = y0t + 16;
  auto rec = [&](int m) { return recs + ((size_t)m * 32 + l) * REC; };
  for (int q = lane; q < T; q += 64) tT[q] = a.times[q];
  typename S_::Tj tj;
  float pconst[4];
  rs_setup<LM, PREC, SOLVER>(a, i, b, l, ct, y0t, tT, tj, pconst);
  wave_sync();
  tj.h0 = tT[1] - tT[0];
  const int k0 = l * ITEMS;
  const float* trj = a.traj_in + (size_t)i * N * T;
  const bool owner_last = (K - 1) / ITEMS == l;
  const float* ob = a.obs + (size_t)b * 4 * T;
  float glp[4];
  const float w_iw = a.iw_logp ? iw_wave_weight(a, i, b) : 0.f;
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) glp[j] = ode_logp_grad(a, w_iw, i, j);

  // parameter-adjoint accumulators: per state the adjoints of its own constants (RlAcc's fields; the entries an
  // instantiation never writes stay compile-time zeros), the growth parameters, the constant precisions
  float accF0[NSP], accCP[NSP], accE[NSP], accAR[NSP], accAS[NSP], accCQ[NSP], accIK[NSP], accDeg[NSP];
  VIHDS_UNROLL for (int j = 0; j < NSP; ++j) accF0[j] = accCP[j] = accE[j] = accAR[j] = accAS[j] = accCQ[j] = accIK[j] = accDeg[j] = 0.f;
  float rb = 0.f, Kb = 0.f, tlb = 0.f, precb[4] = {0.f, 0.f, 0.f, 0.f};
  float lam0[N];
  VIHDS_UNROLL for (int j = 0; j < N; ++j) lam0[j] = 0.f;

  // ---- pass 0: observation injections at this lane's grid points -> rec; the terminal point in the owner of step K-1 ----
  float qK[4] = {0.f, 0.f, 0.f, 0.f}, pvK[4] = {0.f, 0.f, 0.f, 0.f}, yK[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, xK = 0.f;
  {
    auto point = [&](int k, bool on, float* qo, float* pvo, float* y, float& x) {
      const int kc = min(k, K);
      x = trj[kc];
      VIHDS_UNROLL for (int j = 1; j < 6; ++j) y[j] = j < NSP ? trj[(size_t)j * T + kc] : 0.f;
      const float xp[4] = {x, x * y[1], x * (y[2] + OS * y[4]), x * (y[3] + OS * y[5])};
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float pr = PREC ? trj[(size_t)(NSP + j) * T + kc] : pconst[j];
        const float e = xp[j] - ob[j * T + kc];
        float q = -glp[j] * pr * e;
        if (a.g_xpred) q += a.g_xpred[((size_t)i * 4 + j) * T + kc];
        const float pb = glp[j] * (0.5f * frcp(pr) - 0.5f * e * e);
        qo[j] = on ? q : 0.f;
        pvo[j] = on ? pb : 0.f;
        if (!PREC) precb[j] += on ? pb : 0.f;
      }
    };
    VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
      float q[4], pv[4], y[6], x;
      point(k0 + m, k0 + m < K, q, pv, y, x);
      float* r = rec(m);
      VIHDS_UNROLL for (int j = 0; j < 12; ++j) r[O_CS + j] = 0.f;
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { r[O_GB + s] = 0.f; r[O_JX + s] = 0.f; }
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) r[O_Q + j] = q[j];
      if (PREC) { VIHDS_UNROLL for (int j = 0; j < 4; ++j) r[O_PV + j] = pv[j]; }
    }
    point(K, owner_last, qK, pvK, yK, xK);
  }
  // grid injection of species j (1 <= j < NSP) from the observation injections q at a point with OD x
  auto ginj = [&](int j, const float* q, float x) {
    return j == 1 ? q[1] * x : (j == 2 ? q[2] * x : (j == 3 ? q[3] * x : ((LM::OBS_SUM && j == 4) ? q[2] * x : ((LM::OBS_SUM && j == 5) ? q[3] * x : 0.f))));
  };
  auto gtraj = [&](int j, int kc) { return a.g_traj ? a.g_traj[((size_t)i * N + j) * T + kc] : 0.f; };
  auto load_all = [&](int kc, int upto, float* yk) {
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if (S_::level(j) <= upto) yk[j] = trj[(size_t)j * T + kc];
    });
  };

  // VJP of the coefficients of state j for stage adjoints kb[s] (adjoints of the stage derivatives k_s = F_s - a_s Y_s):
  // constants' adjoints, gamma adjoints gbs[s], stage injections into lower states Jl[.][s]
  auto push = [&](auto JJ, const typename S_::Step& st, const typename S_::Ev& e, const float* kb, float* gbs, float (*Jl)[NS],
                  float (*zb)[NS]) {
    constexpr int j = decltype(JJ)::value;
    constexpr int L = S_::level(j);
    const float* c = tj.ct + (j < NSP ? j : 0) * S_::NCT;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      const float ab = -e.Y[j][s] * kb[s];  // adjoint of a_s
      if constexpr (L <= 3) {
        accDeg[j] += ab;
        accF0[j] += kb[s];
        if (S_::gs(j) != 0.f) gbs[s] += ab;
      }
      if constexpr (L == 2) {
        constexpr int q = LM::prom(j);
        accCP[j] = fmaf(kb[s], e.P[q][s], accCP[j]);
        const float nb = kb[s] * c[S_::C_cP] * e.rden[q][s];  // adjoint of the numerator e + a
        const float sP = nb * (1.f - e.P[q][s]);               // adjoint of a = aR luxR^2 + aS lasR^2
        accE[j] += nb;
        accAR[j] = fmaf(sP, e.R2[s], accAR[j]);
        accAS[j] = fmaf(sP, e.S2[s], accAS[j]);
        Jl[6][s] = fmaf(2.f * sP * c[S_::C_aR], e.Y[6][s], Jl[6][s]);
        Jl[7][s] = fmaf(2.f * sP * c[S_::C_aS], e.Y[7][s], Jl[7][s]);
      }
      if constexpr (L == 3) {
        constexpr int q = j - LM::Q0, src = LM::qsrc(j);
        accCQ[j] = fmaf(kb[s], e.Q[q][s], accCQ[j]);
        const float Qb = kb[s] * c[S_::C_cQ];
        const float I = e.Y[src][s], id = e.rdq[q][s];
        Jl[0][s] = fmaf(Qb * I, id, Jl[0][s]);
        Jl[src][s] = fmaf(Qb * st.x[s], id * id, Jl[src][s]);
        accIK[j] -= Qb * st.x[s] * I * I * id * id;
      }
      if constexpr (L == 4) {
        constexpr int o = j - NSP;
        zb[o][s] = kb[s] * e.sp[o][s] * (1.f - e.sp[o][s]);
        zb[4 + o][s] = ab * e.sd[o][s] * (1.f - e.sd[o][s]);
      }
    }
  };
  // stage injections Jl into the species of the levels below `below` (and into x): fold them into those levels' per-step
  // offsets at once, level by level downwards -- their VJPs may inject further down
  auto cascade = [&](auto BELOW, const typename S_::Step& st, const typename S_::Ev& e, float* gbs, float (*Jl)[NS], float* r) {
    constexpr int below = decltype(BELOW)::value;
    static_for<1, below>([&](auto DD) {
      constexpr int L = below - decltype(DD)::value;  // below-1 .. 1
      static_for<1, NSP>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (S_::level(j) == L) {
          float kb[NS];
          const float off = R::reverse(st.h, e.a[j], 0.f, Jl[j], kb);
          r[O_CS + j] += st.valid ? off : 0.f;
          push(JJ, st, e, kb, gbs, Jl, (float (*)[NS]) nullptr);
        }
      });
    });
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      r[O_GB + s] += gbs[s];
      r[O_JX + s] += Jl[0][s];
    }
  };

  // ---- levels TOP .. 1 -------------------------------------------------------------------------------------------------
  static_for<0, TOP>([&](auto DD) {
    constexpr int L = TOP - decltype(DD)::value;
    Aff lm[N];
    float gKj[N];
    VIHDS_UNROLL for (int j = 0; j < N; ++j) { lm[j] = {1.f, 0.f}; gKj[j] = 0.f; }
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if constexpr (S_::level(j) == L) {
        float g = j < NSP ? ginj(j, qK, xK) : pvK[(j - NSP) & 3];
        if (owner_last) g += gtraj(j, K);
        gKj[j] = owner_last ? g : 0.f;
      }
    });
    // pass A: multipliers and offsets of this lane's steps, composed; scan
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      typename S_::Step st;
      S_::make_step(tj, k0 + m, trj[min(k0 + m, K - 1)], st);
      float yk[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) yk[j] = 0.f;
      load_all(st.kc, L == 4 ? 3 : L - 1, yk);
      typename S_::Ev e;
      S_::template eval<L, false>(tj, st, yk, e);
      const float* r = rec(m);
      float q[4];
      ldv<4>(r + O_Q, q);
      static_for<1, N>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (S_::level(j) == L) {
          const float A = S_::template mult_of<j>(st, e);
          float c = j < NSP ? r[O_CS + j] + ginj(j, q, st.x[0]) : r[O_PV + ((j - NSP) & 3)];
          c += gtraj(j, st.kc);
          rs_lane_step(lm[j], st.valid, st.last, A, c, gKj[j]);
        }
      });
    }
    float lam[N];
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if constexpr (S_::level(j) == L) lam[j] = rs_lane_entry(lm[j], lane);
    });
    // pass B: the Lambda-driven stage adjoints of the level's states, their VJPs, the cascade below
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      typename S_::Step st;
      S_::make_step(tj, k0 + m, trj[min(k0 + m, K - 1)], st);
      float yk[N];
      VIHDS_UNROLL for (int j = 0; j < N; ++j) yk[j] = 0.f;
      load_all(st.kc, L, yk);
      typename S_::Ev e;
      S_::template eval<L, true>(tj, st, yk, e);
      float* r = rec(m);
      float q[4], gbs[NS], Jl[NSP][NS], zb[8][NS];
      ldv<4>(r + O_Q, q);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        gbs[s] = 0.f;
        VIHDS_UNROLL for (int j = 0; j < NSP; ++j) Jl[j][s] = 0.f;
        VIHDS_UNROLL for (int j = 0; j < 8; ++j) zb[j][s] = 0.f;
      }
      static_for<1, N>([&](auto JJ) {
        constexpr int j = decltype(JJ)::value;
        if constexpr (S_::level(j) == L) {
          const float lin = lam[j] + (st.last ? gKj[j] : 0.f);
          if constexpr (L == 4) {
            if (a.aux && live) a.aux[(size_t)((n + RS_TPB - 1) / RS_TPB) * rl_nwg(NIN) + (((size_t)(j - NSP) * n + i) * ITEMS + m) * 32 + l] = st.valid ? lin : 0.f;
          }
          float kb[NS], Jz[NS];
          VIHDS_UNROLL for (int s = 0; s < NS; ++s) Jz[s] = 0.f;
          const float lk = R::reverse(st.h, e.a[j], st.valid ? lin : 0.f, Jz, kb);
          float c = j < NSP ? r[O_CS + j] + ginj(j, q, st.x[0]) : r[O_PV + ((j - NSP) & 3)];
          c += gtraj(j, st.kc);
          lam[j] = st.valid ? lk + c : lam[j];
          push(JJ, st, e, kb, gbs, Jl, zb);
        }
      });
      if constexpr (L == 4) {
        // input adjoints of the network: hb_i = sum_o Wp[o][i] zbp_o + Wd[o][i] zbd_o, through the tanh
        VIHDS_UNROLL for (int s = 0; s < NS; ++s)
          VIHDS_UNROLL for (int j = 0; j < NSP; ++j) {
            float hb = 0.f;
            VIHDS_UNROLL for (int o = 0; o < 4; ++o) {
              hb = fmaf(tj.w[o * NIN + 1 + j], zb[o][s], hb);
              hb = fmaf(tj.w[4 * NIN + 4 + o * NIN + 1 + j], zb[4 + o][s], hb);
            }
            Jl[j][s] = hb * (1.f - e.hin[1 + j][s] * e.hin[1 + j][s]);
          }
      }
      cascade(std::integral_constant<int, (L == 4 ? 4 : L)>{}, st, e, gbs, Jl, r);
    }
    static_for<1, N>([&](auto JJ) {
      constexpr int j = decltype(JJ)::value;
      if constexpr (S_::level(j) == L) lam0[j] = lam[j];
    });
  });
  // ---- level 0: x.  tangent multipliers a_s = -gr_s (1 - 2 u_s); stage injections JX - GB gr / K ---------------------------
  {
    const float gK = owner_last ? (qK[0] + qK[1] * yK[1] + qK[2] * (yK[2] + OS * yK[4]) + qK[3] * (yK[3] + OS * yK[5]) + gtraj(0, K)) : 0.f;
    auto x_stage = [&](const typename S_::Step& st, const float* r, float* ax, float* gbo, float* Jx) {
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        ax[s] = -st.gr[s] * fmaf(-2.f, st.u[s], 1.f);
        gbo[s] = st.valid ? r[O_GB + s] : 0.f;
        Jx[s] = st.valid ? fmaf(-gbo[s] * st.gr[s], tj.iK, r[O_JX + s]) : 0.f;
      }
    };
    auto x_inj = [&](const typename S_::Step& st, const float* r) {
      float q[4], y[6];
      ldv<4>(r + O_Q, q);
      VIHDS_UNROLL for (int j = 1; j < 6; ++j) y[j] = j < NSP ? trj[(size_t)j * T + st.kc] : 0.f;
      return q[0] + q[1] * y[1] + q[2] * (y[2] + OS * y[4]) + q[3] * (y[3] + OS * y[5]) + gtraj(0, st.kc);
    };
    Aff lm = {1.f, 0.f};
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      typename S_::Step st;
      S_::make_step(tj, k0 + m, trj[min(k0 + m, K - 1)], st);
      const float* r = rec(m);
      float ax[NS], gbo[NS], Jx[NS], Fz[NS], kv[NS], A, dB;
      x_stage(st, r, ax, gbo, Jx);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) Fz[s] = 0.f;
      R::affine(st.h, ax, Fz, A, dB);
      const float off = st.valid ? R::reverse(st.h, ax, 0.f, Jx, kv) : 0.f;
      rs_lane_step(lm, st.valid, st.last, A, off + x_inj(st, r), gK);
    }
    float lam = rs_lane_entry(lm, lane);
    VIHDS_ROLLED for (int m = ITEMS - 1; m >= 0; --m) {
      typename S_::Step st;
      S_::make_step(tj, k0 + m, trj[min(k0 + m, K - 1)], st);
      const float* r = rec(m);
      float ax[NS], gbo[NS], Jx[NS], kbar[NS];
      x_stage(st, r, ax, gbo, Jx);
      const float lin = lam + (st.last ? gK : 0.f);
      const float lk = R::reverse(st.h, ax, st.valid ? lin : 0.f, Jx, kbar);
      if (st.valid) lam = lk + x_inj(st, r);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const float gtot = fmaf(kbar[s], st.x[s], gbo[s]);  // adjoint of gamma_s from every state
        const float grb = gtot * (1.f - st.u[s]);            // adjoint of gr_s
        rb = fmaf(grb, st.sg[s], rb);
        tlb = fmaf(grb * st.gr[s], 1.f - st.sg[s], tlb);
        Kb = fmaf(gtot * st.gr[s], st.x[s], Kb);
      }
    }
    lam0[0] = lam;
  }
  // ---- epilogue: sums over the 32 lanes, per-state constants' adjoints -> prepared parameters -> theta (lane 0) ---------------
  VIHDS_UNROLL for (int j = 0; j < NSP; ++j) {
    accF0[j] = sum32(accF0[j], lane); accCP[j] = sum32(accCP[j], lane); accE[j] = sum32(accE[j], lane);
    accAR[j] = sum32(accAR[j], lane); accAS[j] = sum32(accAS[j], lane); accCQ[j] = sum32(accCQ[j], lane);
    accIK[j] = sum32(accIK[j], lane); accDeg[j] = sum32(accDeg[j], lane);
  }
  rb = sum32(rb, lane); tlb = sum32(tlb, lane); Kb = sum32(Kb, lane);
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) precb[j] = sum32(precb[j], lane);
  if (l == 0 && live) {
    float th[M::NSLOT], cc[LM::NCOND > 0 ? LM::NCOND : 1], p[M::NP], pb[M::NP], thb[M::NSLOT], li[NSP];
    VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) th[q] = a.theta[(size_t)a.slot_row[q] * n + i];
    VIHDS_UNROLL for (int q = 0; q < LM::NCOND; ++q) cc[q] = clampf(expf(a.cond[b * a.C + q]) - 1.f, 1e-12f, 1e6f);
    M::prepare(th, cc, p);
    VIHDS_UNROLL for (int q = 0; q < M::NP; ++q) pb[q] = 0.f;
    auto T_ = [&](int ln, int f) {
      return f == 0 ? accF0[ln] : (f == 1 ? accCP[ln] : (f == 2 ? accE[ln] : (f == 3 ? accAR[ln] : (f == 4 ? accAS[ln] : (f == 5 ? accCQ[ln] : (f == 6 ? accIK[ln] : accDeg[ln]))))));
    };
    pb[M::P_r] = rb; pb[M::P_K] = Kb * tj.iK * tj.iK; pb[M::P_tlag] = -4.f * tlb;
    LM::map(T_, p, pb);
    VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) thb[q] = 0.f;
    M::prepare_vjp(th, cc, p, pb, thb);
    VIHDS_UNROLL for (int q = 0; q < NSP; ++q) li[q] = lam0[q];
    M::init_vjp(li, thb);
    VIHDS_UNROLL for (int q = 0; q < M::NSLOT; ++q) a.g_theta[(size_t)a.slot_row[q] * n + i] = thb[q];
    VIHDS_UNROLL for (int q = 0; q < 4; ++q)
      a.g_theta[(size_t)a.slot_row[M::NSLOT + q] * n + i] = PREC ? lam0[NSP + q] : precb[q];
  }
}

// ---- weight gradients of the precision network -------------------------------------------------------------------------
template <class LM, int SOLVER>
__global__ void __launch_bounds__(RS_T, 2) relay_scan_wgrad_kernel(OdeArgs a) {
  using S_ = Rs<LM, true, SOLVER>;
  using R = Rk<SOLVER>;
  constexpr int NSP = S_::NSP, N = S_::N, NS = S_::NS, NIN = S_::NIN, NWROW = rl_nwrow(NIN), NWG = rl_nwg(NIN);
  extern __shared__ float lds[];
  __shared__ float wred[RS_T / 64][NWG];
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, tib = tid >> 5, wave = tid >> 6;
  const int i0 = blockIdx.x * RS_TPB + tib;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int T = a.T, K = T - 1, ITEMS = relay_scan_items(T);
  const int TP = (T + 4) & ~3;
  const size_t n = a.n;
  float* tT = lds;
  float* ct = lds + TP + (size_t)tib * (16 * 8 + 16);
  float* y0t = ct + 16 * 8;
  for (int q = lane; q < T; q += 64) tT[q] = a.times[q];
  typename S_::Tj tj;
  float pconst[4];
  rs_setup<LM, true, SOLVER>(a, i, b, l, ct, y0t, tT, tj, pconst);
  wave_sync();
  tj.h0 = tT[1] - tT[0];
  const int k0 = l * ITEMS;
  const float* trj = a.traj_in + (size_t)i * N * T;
  const float* lv = a.aux + (size_t)((n + RS_TPB - 1) / RS_TPB) * NWG;
  // rows of the partial: output o: Wp row [NIN], Wd row [NIN], bp, bd
  float wg[4][NWROW];
  VIHDS_UNROLL for (int o = 0; o < 4; ++o)
    VIHDS_UNROLL for (int q = 0; q < NWROW; ++q) wg[o][q] = 0.f;
  VIHDS_ROLLED for (int m = 0; m < ITEMS; ++m) {
    typename S_::Step st;
    S_::make_step(tj, k0 + m, trj[min(k0 + m, K - 1)], st);
    float yk[N];
    yk[0] = 0.f;
    static_for<1, N>([&](auto JJ) { constexpr int j = decltype(JJ)::value; yk[j] = trj[(size_t)j * T + st.kc]; });
    typename S_::Ev e;
    S_::template eval<4, true>(tj, st, yk, e);
    VIHDS_UNROLL for (int o = 0; o < 4; ++o) {
      const float lin = (live && st.valid) ? lv[(((size_t)o * n + i) * ITEMS + m) * 32 + l] : 0.f;
      float kb[NS], Jz[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) Jz[s] = 0.f;
      (void)R::reverse(st.h, e.a[NSP + o], lin, Jz, kb);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const float zp = kb[s] * e.sp[o][s] * (1.f - e.sp[o][s]);
        const float zd = -e.Y[NSP + o][s] * kb[s] * e.sd[o][s] * (1.f - e.sd[o][s]);
        VIHDS_UNROLL for (int q = 0; q < NIN; ++q) {
          wg[o][q] = fmaf(zp, e.hin[q][s], wg[o][q]);
          wg[o][NIN + q] = fmaf(zd, e.hin[q][s], wg[o][NIN + q]);
        }
        wg[o][2 * NIN] += zp;
        wg[o][2 * NIN + 1] += zd;
      }
    }
  }
  // sum over the wavefront (both trajectories), then over the block's wavefronts in order: one partial row per block
  VIHDS_UNROLL for (int o = 0; o < 4; ++o)
    VIHDS_UNROLL for (int q = 0; q < NWROW; ++q) {
      float v = sum32(wg[o][q], lane);
      v += lane_read(v, lane ^ 32);
      if (lane == 0) wred[wave][o * NWROW + q] = v;
    }
  __syncthreads();
  if (tid < NWG) {
    float acc = 0.f;
    VIHDS_UNROLL for (int q = 0; q < RS_T / 64; ++q) acc += wred[q][tid];
    a.aux[(size_t)blockIdx.x * NWG + tid] = acc;
  }
}

// ---- launch ------------------------------------------------------------------------------------------------------------
template <class LM, bool PREC, int SOLVER>
inline int relay_scan_launch_s(bool backward, const OdeArgs& a, hipStream_t st) {
  const int nblk = (a.n + RS_TPB - 1) / RS_TPB;
  if (!backward) {
    if (!a.traj) return VIHDS_E_UNSUPPORTED;  // (the trajectory is the kernel's own record between the levels)
    const size_t lds = relay_scan_fwd_lds_floats<LM, PREC, SOLVER>(a.T) * sizeof(float);
    auto kern = relay_scan_fwd_kernel<LM, PREC, SOLVER>;
    if (lds > 160 * 1024) return VIHDS_E_UNSUPPORTED;
    if (lds > 64 * 1024) {
      static bool opted = false;
      if (!opted && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        return VIHDS_E_HIP;
      opted = true;
    }
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(RS_T), lds, st, a);
    return VIHDS_OK;
  }
  const size_t lds = relay_scan_bwd_lds_floats<LM, PREC, SOLVER>(a.T) * sizeof(float);
  auto kern = relay_scan_bwd_kernel<LM, PREC, SOLVER>;
  if (lds > 160 * 1024) return VIHDS_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    static bool opted = false;
    if (!opted && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return VIHDS_E_HIP;
    opted = true;
  }
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(RS_T), lds, st, a);
  if constexpr (PREC) {
    if (a.g_weights && a.aux) {
      constexpr int NIN = 1 + LM::NSP;
      const size_t lw = (size_t)(((a.T + 4) & ~3) + RS_TPB * (16 * 8 + 16)) * sizeof(float);
      hipLaunchKernelGGL((relay_scan_wgrad_kernel<LM, SOLVER>), dim3(nblk), dim3(RS_T), lw, st, a);
      hipLaunchKernelGGL(relay_lane_wreduce_kernel, dim3((rl_nwg(NIN) + 3) / 4), dim3(256), 0, st, a.aux, nblk, a.g_weights, NIN);
    }
  }
  return VIHDS_OK;
}
template <class LM, bool PREC>
inline int relay_scan_launch(bool backward, int solver, const OdeArgs& a, hipStream_t st) {
  switch (solver) {
    case VIHDS_SOLVER_MODEULER: return relay_scan_launch_s<LM, PREC, VIHDS_SOLVER_MODEULER>(backward, a, st);
    case VIHDS_SOLVER_MODEULERWHILE: return relay_scan_launch_s<LM, PREC, VIHDS_SOLVER_MODEULERWHILE>(backward, a, st);
    case VIHDS_SOLVER_EULER: return relay_scan_launch_s<LM, PREC, VIHDS_SOLVER_EULER>(backward, a, st);
    case VIHDS_SOLVER_MIDPOINT: return relay_scan_launch_s<LM, PREC, VIHDS_SOLVER_MIDPOINT>(backward, a, st);
    case VIHDS_SOLVER_RK4: return relay_scan_launch_s<LM, PREC, VIHDS_SOLVER_RK4>(backward, a, st);
  }
  return VIHDS_E_BADARG;
}

}  // namespace vihds
