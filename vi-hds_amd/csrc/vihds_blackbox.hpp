// dr_blackbox: the right-hand side is two small MLPs shared by all trajectories
//   NeuralStates     (reference vihds/ode.py:119-146):  h = relu(Wh [x, const] + bh);
//                     dx = sigmoid(Wp h + bp) - sigmoid(Wd h + bd) * x                     x: NX = 4 + L states
//   NeuralPrecisions (reference vihds/precisions.py:63-87, hidden layer shared by prod and degr):
//                     g = relu(Vh [t, x, const] + ch);  dv = sigmoid(Vp g + cp) - sigmoid(Vd g + cd) * v
// with const = [z.., x.., y.. (theta), treatments, device one-hot]  (reference models/dr_blackbox.py:35-58).
//
// The NC time-invariant inputs are hoisted out of the time loop: once per trajectory
//   p[k]      = bh[k] + sum_i Wh[k][NX+i]   const_i      (k < HS)
//   p[HS + k] = ch[k] + sum_i Vh[k][1+NX+i] const_i      (k < HP)
// so one RHS evaluation costs HS*NX + HP*(1+NX) + 2*NX*HS + 8*HP MACs instead of the full 3 390 flop
// (SURVEY.md 8d keeps the declared numerator).  The loop's weights live in LDS in a compact compile-time layout.
//
// Weight gradients: the contraction over (trajectory x RHS evaluation) has K ~ 10^6 and tiny M,N; per-thread
// accumulators are impossible (1 760 weights).  The adjoint kernel therefore dumps, per evaluation, the
// pre-activation gradients and the layer inputs ([E][F][n], coalesced), plus the per-trajectory sums
// Delta = sum_evals gs (for the hoisted columns and biases); the host contracts them with batched GEMMs.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vihds_args.hpp"

namespace vihds {

template <class M>
struct is_blackbox : std::false_type {};

constexpr int BB_BSUM_MAX = 2 * (4 + 12) + 8;  // up to 12 latent species
struct BlackboxCtx {
  float* dump;    // &aux[i]
  size_t n;       // trajectories
  size_t fstride; // floats between consecutive fields = E * n  (dump layout [F][E][n])
  int e;          // evaluation counter
  float bsum[BB_BSUM_MAX]; // running sums of the second-layer pre-activation adjoints (-> output-bias gradients): 2 NX + 8 used
};

__device__ __forceinline__ float bb_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

template <int L, int HS, int HP, int NZ, int NXG, int NY>
struct Blackbox {
  static constexpr int NX = 4 + L;       // states of the NeuralStates net
  static_assert(2 * NX + 8 <= BB_BSUM_MAX, "dr_blackbox: at most 12 latent species");
  static constexpr int N = NX + 4;       // + 4 precision states
  static constexpr int NS = NX;          // x_states handed back by NeuralPrecisions.expand
  static constexpr int NC = 0;
  static constexpr int OBS = 1;          // OBS_DIRECT (dr_blackbox.py:112-121)
  static constexpr bool NEURAL_PREC = true;
  static constexpr int NLAT = NZ + NXG + NY;
  static constexpr int NSLOT = NLAT + 4;
  static constexpr int NP = HS + HP;
  static constexpr int NINP = 1 + NX;    // time + states
  // unroll factors of the loops over hidden units in the time loop: full for networks up to 32 units (the ICML sizes:
  // everything in registers); wider ones walk the units two at a time -- fully unrolled, the 50-unit network's adjoint
  // kept 4.8 KB of spills per lane (13.6 ms at B=36, S=200 where the 25-unit one takes 0.4)
  static constexpr int UH = HS <= 32 ? HS : 2, UP = HP <= 32 ? HP : 2;
  // LDS layout of the time-loop weights
  static constexpr int L_WH = 0, L_WP = L_WH + HS * NX, L_BP = L_WP + NX * HS, L_WD = L_BP + NX, L_BD = L_WD + NX * HS,
                       L_VH = L_BD + NX, L_VP = L_VH + HP * NINP, L_CP = L_VP + 4 * HP, L_VD = L_CP + 4,
                       L_CD = L_VD + 4 * HP, NW = L_CD + 4;
  // dump fields per evaluation
  static constexpr int F_ZA = 0, F_ZD = F_ZA + NX, F_HS = F_ZD + NX, F_GS = F_HS + HS, F_Y = F_GS + HS,
                       F_T = F_Y + NX, F_ZAP = F_T + 1, F_ZDP = F_ZAP + 4, F_HP = F_ZDP + 4, F_GP = F_HP + HP,
                       NF = F_GP + HP;

  __host__ static const char* slot_name(int s) {
    static char names[NSLOT][16];
    static bool init = false;
    if (!init) {
      int q = 0;
      for (int i = 0; i < NZ; ++i) snprintf(names[q++], 16, "z%d", i + 1);
      for (int i = 0; i < NXG; ++i) snprintf(names[q++], 16, "x%d", i + 1);
      for (int i = 0; i < NY; ++i) snprintf(names[q++], 16, "y%d", i + 1);
      const char* ini[4] = {"init_x", "init_rfp", "init_yfp", "init_cfp"};
      for (int i = 0; i < 4; ++i) snprintf(names[q++], 16, "%s", ini[i]);
      init = true;
    }
    return names[s];
  }

  // global weight buffer offsets (row strides depend on the runtime n_const)
  struct Off {
    int nin_s, nin_p, wh, bh, wp, bp, wd, bd, vh, ch, vp, cp, vd, cd;
  };
  __host__ __device__ static Off offsets(int n_const) {
    Off o;
    o.nin_s = NX + n_const;
    o.nin_p = 1 + NX + n_const;
    o.wh = 0;
    o.bh = o.wh + HS * o.nin_s;
    o.wp = o.bh + HS;
    o.bp = o.wp + NX * HS;
    o.wd = o.bp + NX;
    o.bd = o.wd + NX * HS;
    o.vh = o.bd + NX;
    o.ch = o.vh + HP * o.nin_p;
    o.vp = o.ch + HP;
    o.cp = o.vp + 4 * HP;
    o.vd = o.cp + 4;
    o.cd = o.vd + 4 * HP;
    return o;
  }
  __host__ static int n_weights(int n_const) { return offsets(n_const).cd + 4; }

  __device__ static void stage(const OdeArgs& a, float* lds) {
    const Off o = offsets(a.n_const);
    const float* w = a.weights;
    for (int q = threadIdx.x; q < NW; q += blockDim.x) {
      float v;
      if (q < L_WP) v = w[o.wh + (q / NX) * o.nin_s + (q % NX)];
      else if (q < L_BP) v = w[o.wp + (q - L_WP)];
      else if (q < L_WD) v = w[o.bp + (q - L_BP)];
      else if (q < L_BD) v = w[o.wd + (q - L_WD)];
      else if (q < L_VH) v = w[o.bd + (q - L_BD)];
      else if (q < L_VP) v = w[o.vh + ((q - L_VH) / NINP) * o.nin_p + ((q - L_VH) % NINP)];
      else if (q < L_CP) v = w[o.vp + (q - L_VP)];
      else if (q < L_VD) v = w[o.cp + (q - L_CP)];
      else if (q < L_CD) v = w[o.vd + (q - L_VD)];
      else v = w[o.cd + (q - L_CD)];
      lds[q] = v;
    }
  }

  // hoisted hidden pre-activations
  __device__ static void prepare_bb(const float* th, const OdeArgs& a, int b, float* p) {
    const Off o = offsets(a.n_const);
    const float* w = a.weights;
#pragma unroll
    for (int k = 0; k < HS; ++k) p[k] = w[o.bh + k];
#pragma unroll
    for (int k = 0; k < HP; ++k) p[HS + k] = w[o.ch + k];
#pragma unroll
    for (int i = 0; i < NLAT; ++i) {  // latents z, x, y (theta)
#pragma unroll
      for (int k = 0; k < HS; ++k) p[k] += w[o.wh + k * o.nin_s + NX + i] * th[i];
#pragma unroll
      for (int k = 0; k < HP; ++k) p[HS + k] += w[o.vh + k * o.nin_p + 1 + NX + i] * th[i];
    }
    for (int i = NLAT; i < a.n_const; ++i) {  // treatments (as given: log(1+c)), then the device one-hot
      const float ci = (i < NLAT + a.C) ? a.cond[b * a.C + (i - NLAT)] : a.dev1hot[b * a.D + (i - NLAT - a.C)];
#pragma unroll
      for (int k = 0; k < HS; ++k) p[k] += w[o.wh + k * o.nin_s + NX + i] * ci;
#pragma unroll
      for (int k = 0; k < HP; ++k) p[HS + k] += w[o.vh + k * o.nin_p + 1 + NX + i] * ci;
    }
  }
  // d loss / d theta through the hoisted pre-activations: const_i_bar = sum_k W[k][i] * Delta[k]
  __device__ static void prepare_vjp_bb(const float*, const OdeArgs& a, int, const float* pb, float* thb) {
    const Off o = offsets(a.n_const);
    const float* w = a.weights;
#pragma unroll
    for (int i = 0; i < NLAT; ++i) {
      float g = 0.f;
#pragma unroll
      for (int k = 0; k < HS; ++k) g += w[o.wh + k * o.nin_s + NX + i] * pb[k];
#pragma unroll
      for (int k = 0; k < HP; ++k) g += w[o.vh + k * o.nin_p + 1 + NX + i] * pb[HS + k];
      thb[i] = g;
    }
  }
  // Delta [NP][n] after the evaluation dump
  static constexpr int NTAIL = NP + 2 * NX + 8;  // Delta, then the output-bias adjoint sums
  __device__ static void store_delta(const OdeArgs& a, int i, const float* pb, const BlackboxCtx& ctx) {
    float* d = a.aux + (size_t)dump_evals(a) * NF * a.n;
#pragma unroll
    for (int k = 0; k < NP; ++k) d[(size_t)k * a.n + i] = pb[k];
#pragma unroll
    for (int k = 0; k < 2 * NX + 8; ++k) d[(size_t)(NP + k) * a.n + i] = ctx.bsum[k];
  }
  __host__ __device__ static int stages(int solver) {
    if (solver >= VIHDS_SOLVER_DOPRI5) return adaptive_stages(solver);
    return solver == VIHDS_SOLVER_EULER ? 1 : (solver == VIHDS_SOLVER_RK4 ? 4 : 2);
  }
  __device__ static int dump_evals(const OdeArgs& a) { return (a.T - 1) * stages(a.solver); }

  __device__ static void init_bb(const float* th, const OdeArgs& a, float* y) {
#pragma unroll
    for (int j = 0; j < 4; ++j) y[j] = th[NLAT + j];
#pragma unroll
    for (int j = 0; j < L; ++j) y[4 + j] = a.init_latent;
#pragma unroll
    for (int j = 0; j < 4; ++j) y[NX + j] = a.init_prec;
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
#pragma unroll
    for (int j = 0; j < 4; ++j) thb[NLAT + j] = yb[j];
  }

  __device__ static void rhs(float t, const float* y, const float* p, const float* w, float* dy) {
    // the LDS weights are loop-invariant for the time loop; without this fence the compiler hoists all NW loads
    // out of it and spills them (kilobytes of scratch per lane)
    __asm__ volatile("" ::: "memory");
    float za[NX], zd[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) { za[j] = w[L_BP + j]; zd[j] = w[L_BD + j]; }
#pragma unroll UH
    for (int k = 0; k < HS; ++k) {
      float h = p[k];
#pragma unroll
      for (int i = 0; i < NX; ++i) h += w[L_WH + k * NX + i] * y[i];
      h = fmaxf(h, 0.f);
#pragma unroll
      for (int j = 0; j < NX; ++j) { za[j] += w[L_WP + j * HS + k] * h; zd[j] += w[L_WD + j * HS + k] * h; }
    }
#pragma unroll
    for (int j = 0; j < NX; ++j) dy[j] = bb_sigmoid(za[j]) - bb_sigmoid(zd[j]) * y[j];
    float pa[4], pd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { pa[j] = w[L_CP + j]; pd[j] = w[L_CD + j]; }
#pragma unroll UP
    for (int k = 0; k < HP; ++k) {
      float g = p[HS + k] + w[L_VH + k * NINP] * t;
#pragma unroll
      for (int i = 0; i < NX; ++i) g += w[L_VH + k * NINP + 1 + i] * y[i];
      g = fmaxf(g, 0.f);
#pragma unroll
      for (int j = 0; j < 4; ++j) { pa[j] += w[L_VP + j * HP + k] * g; pd[j] += w[L_VD + j * HP + k] * g; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) dy[NX + j] = bb_sigmoid(pa[j]) - bb_sigmoid(pd[j]) * y[NX + j];
  }

  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float* w, const float* v, float* yb,
                                 float* pb, BlackboxCtx& ctx) {
    __asm__ volatile("" ::: "memory");
    float* D = ctx.dump + (size_t)ctx.e * ctx.n;
    const size_t n = ctx.fstride;  // distance between fields
    ctx.e += 1;
    // ---- states net: recompute, then transpose
    float hs[HS], za[NX], zd[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) { za[j] = w[L_BP + j]; zd[j] = w[L_BD + j]; }
#pragma unroll UH
    for (int k = 0; k < HS; ++k) {
      float h = p[k];
#pragma unroll
      for (int i = 0; i < NX; ++i) h += w[L_WH + k * NX + i] * y[i];
      h = fmaxf(h, 0.f);
      hs[k] = h;
      D[(size_t)(F_HS + k) * n] = h;
#pragma unroll
      for (int j = 0; j < NX; ++j) { za[j] += w[L_WP + j * HS + k] * h; zd[j] += w[L_WD + j * HS + k] * h; }
    }
    float zab[NX], zdb[NX];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const float a = bb_sigmoid(za[j]), d = bb_sigmoid(zd[j]);
      yb[j] -= v[j] * d;
      zab[j] = v[j] * a * (1.f - a);
      zdb[j] = -v[j] * y[j] * d * (1.f - d);
      D[(size_t)(F_ZA + j) * n] = zab[j];
      D[(size_t)(F_ZD + j) * n] = zdb[j];
      D[(size_t)(F_Y + j) * n] = y[j];
      ctx.bsum[j] += zab[j];
      ctx.bsum[NX + j] += zdb[j];
    }
#pragma unroll UH
    for (int k = 0; k < HS; ++k) {
      float hb = 0.f;
#pragma unroll
      for (int j = 0; j < NX; ++j) hb += w[L_WP + j * HS + k] * zab[j] + w[L_WD + j * HS + k] * zdb[j];
      const float gs = hs[k] > 0.f ? hb : 0.f;
      D[(size_t)(F_GS + k) * n] = gs;
      pb[k] += gs;
#pragma unroll
      for (int i = 0; i < NX; ++i) yb[i] += w[L_WH + k * NX + i] * gs;
    }
    // ---- precisions net
    float hp[HP], pa[4], pd[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { pa[j] = w[L_CP + j]; pd[j] = w[L_CD + j]; }
#pragma unroll UP
    for (int k = 0; k < HP; ++k) {
      float g = p[HS + k] + w[L_VH + k * NINP] * t;
#pragma unroll
      for (int i = 0; i < NX; ++i) g += w[L_VH + k * NINP + 1 + i] * y[i];
      g = fmaxf(g, 0.f);
      hp[k] = g;
      D[(size_t)(F_HP + k) * n] = g;
#pragma unroll
      for (int j = 0; j < 4; ++j) { pa[j] += w[L_VP + j * HP + k] * g; pd[j] += w[L_VD + j * HP + k] * g; }
    }
    float pab[4], pdb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = bb_sigmoid(pa[j]), d = bb_sigmoid(pd[j]);
      const float vj = v[NX + j];
      yb[NX + j] -= vj * d;
      pab[j] = vj * a * (1.f - a);
      pdb[j] = -vj * y[NX + j] * d * (1.f - d);
      D[(size_t)(F_ZAP + j) * n] = pab[j];
      D[(size_t)(F_ZDP + j) * n] = pdb[j];
      ctx.bsum[2 * NX + j] += pab[j];
      ctx.bsum[2 * NX + 4 + j] += pdb[j];
    }
#pragma unroll UP
    for (int k = 0; k < HP; ++k) {
      float gb = 0.f;
#pragma unroll
      for (int j = 0; j < 4; ++j) gb += w[L_VP + j * HP + k] * pab[j] + w[L_VD + j * HP + k] * pdb[j];
      const float gp = hp[k] > 0.f ? gb : 0.f;
      D[(size_t)(F_GP + k) * n] = gp;
      pb[HS + k] += gp;
#pragma unroll
      for (int i = 0; i < NX; ++i) yb[i] += w[L_VH + k * NINP + 1 + i] * gp;
    }
    D[(size_t)F_T * n] = t;
  }
};

template <int L, int HS, int HP, int NZ, int NXG, int NY>
struct is_blackbox<Blackbox<L, HS, HP, NZ, NXG, NY>> : std::true_type {};

}  // namespace vihds
