// Per-model right-hand sides, their vector-Jacobian products, and the theta -> effective-parameter maps.
//
// Each model is a struct of static __device__ functions operating on small register arrays with
// compile-time indices (everything is fully unrolled, nothing spills to scratch):
//
//   th[NSLOT]  raw (clipped) theta of one trajectory, in the model's slot order (slot_names())
//   p[NP]      effective parameters: clamps applied, Hill fractions pre-computed -- what the reference's
//              OdeFunc constructor stores on `self`
//   y[N]       ODE state
//
//   prepare(th, c, p)                 reference <Model>_RHS.__init__
//   prepare_vjp(th, c, p, pb, thb)    its transpose: thb[slot] = d loss / d th[slot]   (overwrites)
//   init(th, c, y), init_vjp(yb, thb) reference <Model>.initialize_state               (thb +=)
//   rhs(t, y, p, dy)                  reference <Model>_RHS.forward
//   rhs_vjp(t, y, p, v, yb, pb)       yb += (d rhs/d y)^T v ; pb += (d rhs/d p)^T v
//
// `c[]` are the treatments after  clamp(exp(cond) - 1, 1e-12, 1e6)  (e.g. dr_constant.py:26).
#pragma once
#include <hip/hip_runtime.h>

namespace vihds {

enum ObserveKind { OBS_DEFAULT = 0, OBS_DIRECT = 1, OBS_INDUCER = 2 };

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
  // torch.clamp semantics incl. NaN pass-through
  return x < lo ? lo : (x > hi ? hi : x);
}
__device__ __forceinline__ float clamp_pass(float x, float lo, float hi) {
  // torch.clamp backward: gradient flows where lo <= x <= hi (bounds included)
  return (x >= lo && x <= hi) ? 1.f : 0.f;
}
// Division and exp on the time-loop path.  v_rcp_f32 / v_exp_f32 are accurate to ~1 ulp, far inside the 1e-4
// parity budget, and replace the ~10-instruction IEEE division sequence and the range-reduced expf: the RHS is
// a long dependent chain on a handful of wavefronts, so instruction count is time.  (prepare/prepare_vjp run
// once per trajectory and keep the accurate powf/logf.)  Build with -DVIHDS_PRECISE_MATH to switch back.
#ifdef VIHDS_PRECISE_MATH
__device__ __forceinline__ float frcp(float x) { return 1.f / x; }
__device__ __forceinline__ float fdiv(float a, float b) { return a / b; }
__device__ __forceinline__ float fexp(float x) { return expf(x); }
#else
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fexp(float x) { return __expf(x); }
#endif
__device__ __forceinline__ float sigmoid_f(float x) { return frcp(1.f + fexp(-x)); }
// tanh on the time-loop path (inputs of the no-hidden-layer NeuralPrecisions, precisions.py:55-61): 1 - 2/(1 + e^{2x}),
// absolute error ~1e-7 (the libm tanhf is ~30 instructions, 13 of them per RHS evaluation of relay_constant_precisions)
#ifdef VIHDS_PRECISE_MATH
__device__ __forceinline__ float ftanh(float x) { return tanhf(x); }
#else
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * frcp(1.f + fexp(2.f * x)); }
#endif

// d/da a^n and d/dn a^n, matching autograd of torch.pow(tensor, tensor)
__device__ __forceinline__ void pow_vjp(float a, float n, float an, float g, float& ab, float& nb) {
  ab += g * n * powf(a, n - 1.f);
  nb += g * (an * logf(a));
}

// ---------------------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------------------
// growth: gr = r*sigmoid(4(t-tlag)); g = 1 - x/K; gamma = gr*g        (dr_constant.py:80-85)
struct Growth {
  float sig, gr, g, gamma;
};
__device__ __forceinline__ Growth growth(float t, float x, float r, float K, float tlag) {
  Growth o;
  o.sig = sigmoid_f(4.f * (t - tlag));
  o.gr = r * o.sig;
  o.g = 1.f - fdiv(x, K);
  o.gamma = o.gr * o.g;
  return o;
}
// given gamma_bar: accumulate into xb, rb, Kb, tlagb
__device__ __forceinline__ void growth_vjp(const Growth& G, float x, float r, float K, float gammab, float& xb,
                                           float& rb, float& Kb, float& tlagb) {
  float grb = gammab * G.g;
  float gb = gammab * G.gr;
  float invK = frcp(K);
  xb -= gb * invK;
  Kb += gb * x * invK * invK;
  rb += grb * G.sig;
  float sigb = grb * r;
  tlagb -= 4.f * sigb * G.sig * (1.f - G.sig);
}

// promoter: P = (e + KGR*bR + KGS*bS) / (1 + KGR*bR + KGS*bS)          (dr_constant.py:88-95)
__device__ __forceinline__ float promoter(float e, float KGR, float KGS, float bR, float bS, float& den) {
  float num = e + KGR * bR + KGS * bS;
  den = 1.f + KGR * bR + KGS * bS;
  return fdiv(num, den);
}
__device__ __forceinline__ void promoter_vjp(float P, float den, float KGR, float KGS, float bR, float bS, float Pb,
                                             float& eb, float& KGRb, float& KGSb, float& bRb, float& bSb) {
  float nb = fdiv(Pb, den);
  float db = -nb * P;
  float s = nb + db;
  eb += nb;
  KGRb += s * bR;
  KGSb += s * bS;
  bRb += s * KGR;
  bSb += s * KGS;
}

// Hill fractions (dr_constant.py:58-68): f = ((K6*c6)^n + (K12*c12)^n) / (1 + K6*c6 + K12*c12)^n
// with n = clamp(n_raw, .5, 3), K* = clamp(K*_raw, 1e-12, 1)
__device__ __forceinline__ float hill_frac(float n_raw, float K6_raw, float K12_raw, float c6, float c12) {
  float n = clampf(n_raw, 0.5f, 3.0f);
  float K6 = clampf(K6_raw, 1e-12f, 1.f), K12 = clampf(K12_raw, 1e-12f, 1.f);
  float a = K6 * c6, b = K12 * c12;
  return (powf(a, n) + powf(b, n)) / powf(1.f + a + b, n);
}
__device__ __forceinline__ void hill_frac_vjp(float n_raw, float K6_raw, float K12_raw, float c6, float c12, float fb,
                                              float& nb_raw, float& K6b_raw, float& K12b_raw) {
  float n = clampf(n_raw, 0.5f, 3.0f);
  float K6 = clampf(K6_raw, 1e-12f, 1.f), K12 = clampf(K12_raw, 1e-12f, 1.f);
  float a = K6 * c6, b = K12 * c12, d = 1.f + a + b;
  float an = powf(a, n), bn = powf(b, n), dn = powf(d, n);
  float num = an + bn;
  float numb = fb / dn;
  float dnb = -fb * num / (dn * dn);
  float ab = 0.f, bb = 0.f, db = 0.f, nb = 0.f;
  pow_vjp(a, n, an, numb, ab, nb);
  pow_vjp(b, n, bn, numb, bb, nb);
  pow_vjp(d, n, dn, dnb, db, nb);
  ab += db;
  bb += db;
  nb_raw = nb * clamp_pass(n_raw, 0.5f, 3.0f);
  K6b_raw = ab * c6 * clamp_pass(K6_raw, 1e-12f, 1.f);
  K12b_raw = bb * c12 * clamp_pass(K12_raw, 1e-12f, 1.f);
}

#define VIHDS_UNROLL _Pragma("unroll")

// ---------------------------------------------------------------------------------------------
// dr_constant (v1, v2)                     reference: models/dr_constant.py
// ---------------------------------------------------------------------------------------------
template <int VERSION>
struct DrConstant {
  static constexpr int N = 8;
  static constexpr int NS = 8;  // species handed to observe()
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;   // shared neural weights
  static constexpr int NC = 2;
  static constexpr int OBS = OBS_DEFAULT;
  enum Slot {
    S_r, S_K, S_tlag, S_rc, S_drfp, S_dyfp, S_dcfp, S_dR, S_dS, S_e76, S_e81, S_KGR76, S_KGS76, S_KGR81, S_KGS81,
    S_aYFP, S_aCFP, S_a530, S_a480, S_aR, S_aS, S_nR, S_nS, S_H0, S_H1, S_H2, S_H3,
    S_init_x, S_init_rfp, S_init_yfp, S_init_cfp, S_init_luxR, S_init_lasR, NSLOT_V1
  };
  // v2 has only two Hill slots (eS6, eR12): S_H2,S_H3 unused -> slots shift; keep one table per version
  static constexpr int NSLOT = (VERSION == 1) ? (int)NSLOT_V1 : (int)NSLOT_V1 - 2;
  static constexpr int SI = (VERSION == 1) ? (int)S_init_x : (int)S_init_x - 2;  // first init slot
  enum Par {
    P_r, P_K, P_tlag, P_rc, P_drfp, P_dyfp, P_dcfp, P_dR, P_dS, P_e76, P_e81, P_KGR76, P_KGS76, P_KGR81, P_KGS81,
    P_aYFP, P_aCFP, P_a530, P_a480, P_aR, P_aS, P_fR, P_fS, NP
  };
  __host__ static const char* slot_name(int s) {
    static const char* v1[] = {"r", "K", "tlag", "rc", "drfp", "dyfp", "dcfp", "dR", "dS", "e76", "e81", "KGR_76",
                               "KGS_76", "KGR_81", "KGS_81", "aYFP", "aCFP", "a530", "a480", "aR", "aS", "nR", "nS",
                               "KR6", "KR12", "KS6", "KS12", "init_x", "init_rfp", "init_yfp", "init_cfp",
                               "init_luxR", "init_lasR"};
    static const char* v2[] = {"r", "K", "tlag", "rc", "drfp", "dyfp", "dcfp", "dR", "dS", "e76", "e81", "KGR_76",
                               "KGS_76", "KGR_81", "KGS_81", "aYFP", "aCFP", "a530", "a480", "aR", "aS", "nR", "nS",
                               "eS6", "eR12", "init_x", "init_rfp", "init_yfp", "init_cfp", "init_luxR",
                               "init_lasR"};
    return VERSION == 1 ? v1[s] : v2[s];
  }

  __device__ static void prepare(const float* th, const float* c, float* p) {
    p[P_r] = clampf(th[S_r], 0.f, 4.f);
    p[P_K] = clampf(th[S_K], 0.f, 4.f);
    p[P_tlag] = th[S_tlag];
    p[P_rc] = th[S_rc];
    p[P_drfp] = clampf(th[S_drfp], 1e-12f, 2.f);
    p[P_dyfp] = clampf(th[S_dyfp], 1e-12f, 2.f);
    p[P_dcfp] = clampf(th[S_dcfp], 1e-12f, 2.f);
    p[P_dR] = clampf(th[S_dR], 1e-12f, 5.f);
    p[P_dS] = clampf(th[S_dS], 1e-12f, 5.f);
    VIHDS_UNROLL for (int k = P_e76; k <= P_aS; ++k) p[k] = th[S_e76 + (k - P_e76)];
    if (VERSION == 1) {
      p[P_fR] = hill_frac(th[S_nR], th[S_H0], th[S_H1], c[0], c[1]);
      p[P_fS] = hill_frac(th[S_nS], th[S_H2], th[S_H3], c[0], c[1]);
    } else {  // dr_constant.py:69-73: fR = c6^nR + (eR12*c12)^nR ; fS = (eS6*c6)^nS + c12^nS
      float nR = clampf(th[S_nR], 0.5f, 3.f), nS = clampf(th[S_nS], 0.5f, 3.f);
      float eS6 = clampf(th[S_H0], 1e-12f, 1.f), eR12 = clampf(th[S_H1], 1e-12f, 1.f);
      p[P_fR] = powf(c[0], nR) + powf(eR12 * c[1], nR);
      p[P_fS] = powf(eS6 * c[0], nS) + powf(c[1], nS);
    }
  }
  __device__ static void prepare_vjp(const float* th, const float* c, const float* p, const float* pb, float* thb) {
    thb[S_r] = pb[P_r] * clamp_pass(th[S_r], 0.f, 4.f);
    thb[S_K] = pb[P_K] * clamp_pass(th[S_K], 0.f, 4.f);
    thb[S_tlag] = pb[P_tlag];
    thb[S_rc] = pb[P_rc];
    thb[S_drfp] = pb[P_drfp] * clamp_pass(th[S_drfp], 1e-12f, 2.f);
    thb[S_dyfp] = pb[P_dyfp] * clamp_pass(th[S_dyfp], 1e-12f, 2.f);
    thb[S_dcfp] = pb[P_dcfp] * clamp_pass(th[S_dcfp], 1e-12f, 2.f);
    thb[S_dR] = pb[P_dR] * clamp_pass(th[S_dR], 1e-12f, 5.f);
    thb[S_dS] = pb[P_dS] * clamp_pass(th[S_dS], 1e-12f, 5.f);
    VIHDS_UNROLL for (int k = P_e76; k <= P_aS; ++k) thb[S_e76 + (k - P_e76)] = pb[k];
    if (VERSION == 1) {
      hill_frac_vjp(th[S_nR], th[S_H0], th[S_H1], c[0], c[1], pb[P_fR], thb[S_nR], thb[S_H0], thb[S_H1]);
      hill_frac_vjp(th[S_nS], th[S_H2], th[S_H3], c[0], c[1], pb[P_fS], thb[S_nS], thb[S_H2], thb[S_H3]);
    } else {
      float nR = clampf(th[S_nR], 0.5f, 3.f), nS = clampf(th[S_nS], 0.5f, 3.f);
      float eS6 = clampf(th[S_H0], 1e-12f, 1.f), eR12 = clampf(th[S_H1], 1e-12f, 1.f);
      float nRb = 0.f, nSb = 0.f, dummy = 0.f, a12b = 0.f, a6b = 0.f;
      float a12 = eR12 * c[1], a6 = eS6 * c[0];
      pow_vjp(c[0], nR, powf(c[0], nR), pb[P_fR], dummy, nRb);
      pow_vjp(a12, nR, powf(a12, nR), pb[P_fR], a12b, nRb);
      pow_vjp(a6, nS, powf(a6, nS), pb[P_fS], a6b, nSb);
      pow_vjp(c[1], nS, powf(c[1], nS), pb[P_fS], dummy, nSb);
      thb[S_nR] = nRb * clamp_pass(th[S_nR], 0.5f, 3.f);
      thb[S_nS] = nSb * clamp_pass(th[S_nS], 0.5f, 3.f);
      thb[S_H0] = a6b * c[0] * clamp_pass(th[S_H0], 1e-12f, 1.f);
      thb[S_H1] = a12b * c[1] * clamp_pass(th[S_H1], 1e-12f, 1.f);
    }
  }
  __device__ static void init(const float* th, const float*, float* y) {
    y[0] = th[SI + 0]; y[1] = th[SI + 1]; y[2] = th[SI + 2]; y[3] = th[SI + 3];
    y[4] = 0.f; y[5] = 0.f; y[6] = th[SI + 4]; y[7] = th[SI + 5];
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    thb[SI + 0] = yb[0]; thb[SI + 1] = yb[1]; thb[SI + 2] = yb[2]; thb[SI + 3] = yb[3];
    thb[SI + 4] = yb[6]; thb[SI + 5] = yb[7];
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float*, float* dy) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float bR = y[6] * y[6] * p[P_fR], bS = y[7] * y[7] * p[P_fS];
    float d76, d81;
    float P76 = promoter(p[P_e76], p[P_KGR76], p[P_KGS76], bR, bS, d76);
    float P81 = promoter(p[P_e81], p[P_KGR81], p[P_KGS81], bR, bS, d81);
    float rc = p[P_rc], gm = G.gamma;
    dy[0] = gm * y[0];
    dy[1] = rc - (gm + p[P_drfp]) * y[1];
    dy[2] = rc * p[P_aYFP] * P81 - (gm + p[P_dyfp]) * y[2];
    dy[3] = rc * p[P_aCFP] * P76 - (gm + p[P_dcfp]) * y[3];
    dy[4] = rc * p[P_a530] - gm * y[4];
    dy[5] = rc * p[P_a480] - gm * y[5];
    dy[6] = rc * p[P_aR] - (gm + p[P_dR]) * y[6];
    dy[7] = rc * p[P_aS] - (gm + p[P_dS]) * y[7];
  }
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float bR = y[6] * y[6] * p[P_fR], bS = y[7] * y[7] * p[P_fS];
    float d76, d81;
    float P76 = promoter(p[P_e76], p[P_KGR76], p[P_KGS76], bR, bS, d76);
    float P81 = promoter(p[P_e81], p[P_KGR81], p[P_KGS81], bR, bS, d81);
    float rc = p[P_rc], gm = G.gamma;
    float gammab = v[0] * y[0] - v[1] * y[1] - v[2] * y[2] - v[3] * y[3] - v[4] * y[4] - v[5] * y[5] - v[6] * y[6] -
                   v[7] * y[7];
    yb[0] += v[0] * gm;
    yb[1] -= v[1] * (gm + p[P_drfp]);
    yb[2] -= v[2] * (gm + p[P_dyfp]);
    yb[3] -= v[3] * (gm + p[P_dcfp]);
    yb[4] -= v[4] * gm;
    yb[5] -= v[5] * gm;
    yb[6] -= v[6] * (gm + p[P_dR]);
    yb[7] -= v[7] * (gm + p[P_dS]);
    pb[P_rc] += v[1] + v[2] * p[P_aYFP] * P81 + v[3] * p[P_aCFP] * P76 + v[4] * p[P_a530] + v[5] * p[P_a480] +
                v[6] * p[P_aR] + v[7] * p[P_aS];
    pb[P_drfp] -= v[1] * y[1];
    pb[P_dyfp] -= v[2] * y[2];
    pb[P_dcfp] -= v[3] * y[3];
    pb[P_dR] -= v[6] * y[6];
    pb[P_dS] -= v[7] * y[7];
    pb[P_aYFP] += v[2] * rc * P81;
    pb[P_aCFP] += v[3] * rc * P76;
    pb[P_a530] += v[4] * rc;
    pb[P_a480] += v[5] * rc;
    pb[P_aR] += v[6] * rc;
    pb[P_aS] += v[7] * rc;
    float P81b = v[2] * rc * p[P_aYFP], P76b = v[3] * rc * p[P_aCFP];
    float bRb = 0.f, bSb = 0.f;
    promoter_vjp(P76, d76, p[P_KGR76], p[P_KGS76], bR, bS, P76b, pb[P_e76], pb[P_KGR76], pb[P_KGS76], bRb, bSb);
    promoter_vjp(P81, d81, p[P_KGR81], p[P_KGS81], bR, bS, P81b, pb[P_e81], pb[P_KGR81], pb[P_KGS81], bRb, bSb);
    yb[6] += bRb * 2.f * y[6] * p[P_fR];
    yb[7] += bSb * 2.f * y[7] * p[P_fS];
    pb[P_fR] += bRb * y[6] * y[6];
    pb[P_fS] += bSb * y[7] * y[7];
    growth_vjp(G, y[0], p[P_r], p[P_K], gammab, yb[0], pb[P_r], pb[P_K], pb[P_tlag]);
  }
};

// ---------------------------------------------------------------------------------------------
// auto_constant                            reference: models/auto_constant.py:12-97
// ---------------------------------------------------------------------------------------------
struct AutoConstant {
  static constexpr int N = 4;
  static constexpr int NS = 4;  // species handed to observe()
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;   // shared neural weights
  static constexpr int NC = 0;
  static constexpr int OBS = OBS_DIRECT;
  enum Slot { S_r, S_K, S_tlag, S_rc, S_drfp, S_a530, S_a480, S_init_x, S_init_rfp, NSLOT };
  enum Par { P_r, P_K, P_tlag, P_rc, P_drfp, P_a530, P_a480, NP };
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"r", "K", "tlag", "rc", "drfp", "a530", "a480", "init_x", "init_rfp"};
    return n[s];
  }
  __device__ static void prepare(const float* th, const float*, float* p) {
    p[P_r] = clampf(th[S_r], 0.f, 4.f);
    p[P_K] = clampf(th[S_K], 0.f, 4.f);
    p[P_tlag] = th[S_tlag];
    p[P_rc] = th[S_rc];
    p[P_drfp] = clampf(th[S_drfp], 1e-12f, 2.f);
    p[P_a530] = th[S_a530];
    p[P_a480] = th[S_a480];
  }
  __device__ static void prepare_vjp(const float* th, const float*, const float*, const float* pb, float* thb) {
    thb[S_r] = pb[P_r] * clamp_pass(th[S_r], 0.f, 4.f);
    thb[S_K] = pb[P_K] * clamp_pass(th[S_K], 0.f, 4.f);
    thb[S_tlag] = pb[P_tlag];
    thb[S_rc] = pb[P_rc];
    thb[S_drfp] = pb[P_drfp] * clamp_pass(th[S_drfp], 1e-12f, 2.f);
    thb[S_a530] = pb[P_a530];
    thb[S_a480] = pb[P_a480];
  }
  __device__ static void init(const float* th, const float*, float* y) {
    y[0] = th[S_init_x]; y[1] = th[S_init_rfp]; y[2] = 0.f; y[3] = 0.f;
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    thb[S_init_x] = yb[0]; thb[S_init_rfp] = yb[1];
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float*, float* dy) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float rc = p[P_rc], gm = G.gamma;
    dy[0] = gm * y[0];
    dy[1] = rc - (gm + p[P_drfp]) * y[1];
    dy[2] = rc * p[P_a530] - gm * y[2];
    dy[3] = rc * p[P_a480] - gm * y[3];
  }
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float rc = p[P_rc], gm = G.gamma;
    float gammab = v[0] * y[0] - v[1] * y[1] - v[2] * y[2] - v[3] * y[3];
    yb[0] += v[0] * gm;
    yb[1] -= v[1] * (gm + p[P_drfp]);
    yb[2] -= v[2] * gm;
    yb[3] -= v[3] * gm;
    pb[P_rc] += v[1] + v[2] * p[P_a530] + v[3] * p[P_a480];
    pb[P_drfp] -= v[1] * y[1];
    pb[P_a530] += v[2] * rc;
    pb[P_a480] += v[3] * rc;
    growth_vjp(G, y[0], p[P_r], p[P_K], gammab, yb[0], pb[P_r], pb[P_K], pb[P_tlag]);
  }
};

// ---------------------------------------------------------------------------------------------
// inducer_constant                         reference: models/inducer_constant.py:11-80 (RHS), :92-97 (x0), :106-114
// (observe).  The reference class itself raises at construction (init_with_params does not exist, :85): parity
// unpinned, the equations are restated from the RHS class.  One treatment (arabinose): c[0].
// ---------------------------------------------------------------------------------------------
struct InducerConstant {
  static constexpr int N = 5;   // x, rfp, yfp, f530, f480
  static constexpr int NS = 5;
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;
  static constexpr int NC = 1;   // arabinose
  static constexpr int OBS = OBS_INDUCER;
  enum Slot { S_r, S_K, S_tlag, S_rc, S_a530, S_a480, S_drfp, S_dyfp, S_aYFP, S_nA, S_eA, S_KAra,
              S_init_x, S_init_rfp, S_init_yfp, NSLOT };
  enum Par { P_r, P_K, P_tlag, P_rc, P_a530, P_a480, P_drfp, P_dyfp, P_aYFP, P_PBAD, NP };
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"r", "K", "tlag", "rc", "a530", "a480", "drfp", "dyfp", "aYFP_Inducer", "nA", "eA",
                              "KAra", "init_x", "init_rfp", "init_yfp"};
    return n[s];
  }
  __device__ static void prepare(const float* th, const float* c, float* p) {
    p[P_r] = clampf(th[S_r], 0.f, 4.f);
    p[P_K] = clampf(th[S_K], 0.f, 4.f);
    p[P_tlag] = th[S_tlag];
    p[P_rc] = th[S_rc];
    p[P_a530] = th[S_a530];
    p[P_a480] = th[S_a480];
    p[P_drfp] = clampf(th[S_drfp], 1e-12f, 2.f);
    p[P_dyfp] = clampf(th[S_dyfp], 1e-12f, 2.f);
    p[P_aYFP] = th[S_aYFP];
    // PBAD = (ara^nA + eA*KAra^nA) / (ara^nA + KAra^nA)          inducer_constant.py:53-55
    const float nA = clampf(th[S_nA], 0.5f, 3.f);
    const float an = powf(c[0], nA), kn = powf(th[S_KAra], nA);
    p[P_PBAD] = (an + th[S_eA] * kn) / (an + kn);
  }
  __device__ static void prepare_vjp(const float* th, const float* c, const float*, const float* pb, float* thb) {
    thb[S_r] = pb[P_r] * clamp_pass(th[S_r], 0.f, 4.f);
    thb[S_K] = pb[P_K] * clamp_pass(th[S_K], 0.f, 4.f);
    thb[S_tlag] = pb[P_tlag];
    thb[S_rc] = pb[P_rc];
    thb[S_a530] = pb[P_a530];
    thb[S_a480] = pb[P_a480];
    thb[S_drfp] = pb[P_drfp] * clamp_pass(th[S_drfp], 1e-12f, 2.f);
    thb[S_dyfp] = pb[P_dyfp] * clamp_pass(th[S_dyfp], 1e-12f, 2.f);
    thb[S_aYFP] = pb[P_aYFP];
    const float nA = clampf(th[S_nA], 0.5f, 3.f);
    const float an = powf(c[0], nA), kn = powf(th[S_KAra], nA);
    const float num = an + th[S_eA] * kn, den = an + kn;
    const float numb = pb[P_PBAD] / den, denb = -pb[P_PBAD] * num / (den * den);
    const float anb = numb + denb, knb = numb * th[S_eA] + denb;
    float nAb = 0.f, dummy = 0.f, KAb = 0.f;
    pow_vjp(c[0], nA, an, anb, dummy, nAb);
    pow_vjp(th[S_KAra], nA, kn, knb, KAb, nAb);
    thb[S_nA] = nAb * clamp_pass(th[S_nA], 0.5f, 3.f);
    thb[S_eA] = numb * kn;
    thb[S_KAra] = KAb;
  }
  __device__ static void init(const float* th, const float*, float* y) {
    y[0] = th[S_init_x]; y[1] = th[S_init_rfp]; y[2] = th[S_init_yfp]; y[3] = 0.f; y[4] = 0.f;
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    thb[S_init_x] = yb[0]; thb[S_init_rfp] = yb[1]; thb[S_init_yfp] = yb[2];
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float*, float* dy) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    const float rc = p[P_rc], gm = G.gamma;
    dy[0] = gm * y[0];
    dy[1] = rc - (gm + p[P_drfp]) * y[1];
    dy[2] = rc * p[P_aYFP] * p[P_PBAD] - (gm + p[P_dyfp]) * y[2];
    dy[3] = rc * p[P_a530] - gm * y[3];
    dy[4] = rc * p[P_a480] - gm * y[4];
  }
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    const float rc = p[P_rc], gm = G.gamma;
    const float gammab = v[0] * y[0] - v[1] * y[1] - v[2] * y[2] - v[3] * y[3] - v[4] * y[4];
    yb[0] += v[0] * gm;
    yb[1] -= v[1] * (gm + p[P_drfp]);
    yb[2] -= v[2] * (gm + p[P_dyfp]);
    yb[3] -= v[3] * gm;
    yb[4] -= v[4] * gm;
    pb[P_rc] += v[1] + v[2] * p[P_aYFP] * p[P_PBAD] + v[3] * p[P_a530] + v[4] * p[P_a480];
    pb[P_drfp] -= v[1] * y[1];
    pb[P_dyfp] -= v[2] * y[2];
    pb[P_aYFP] += v[2] * rc * p[P_PBAD];
    pb[P_PBAD] += v[2] * rc * p[P_aYFP];
    pb[P_a530] += v[3] * rc;
    pb[P_a480] += v[4] * rc;
    growth_vjp(G, y[0], p[P_r], p[P_K], gammab, yb[0], pb[P_r], pb[P_K], pb[P_tlag]);
  }
};

// ---------------------------------------------------------------------------------------------
// debug_constant                           reference: models/debug.py:11-53.  gamma = r (1 - x); dx = x gamma;
// d s_k = 1 - (gamma + 1) s_k for the three fluorescent species; x0 = [init_x, 0, 0, 0].  The reference class is
// stale (gen_reaction_equations has the pre-refactoring signature, debug.py:35, and observe indexes the time axis,
// :25-31): parity unpinned; observe is taken as the evident [OD, OD*s1, OD*s2, OD*s3].
// ---------------------------------------------------------------------------------------------
struct DebugConstant {
  static constexpr int N = 4;
  static constexpr int NS = 4;
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;
  static constexpr int NC = 0;
  static constexpr int OBS = OBS_DIRECT;
  enum Slot { S_r, S_init_x, NSLOT };
  enum Par { P_r, NP };
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"r", "init_x"};
    return n[s];
  }
  __device__ static void prepare(const float* th, const float*, float* p) { p[P_r] = th[S_r]; }
  __device__ static void prepare_vjp(const float*, const float*, const float*, const float* pb, float* thb) {
    thb[S_r] = pb[P_r];
  }
  __device__ static void init(const float* th, const float*, float* y) {
    y[0] = th[S_init_x]; y[1] = 0.f; y[2] = 0.f; y[3] = 0.f;
  }
  __device__ static void init_vjp(const float* yb, float* thb) { thb[S_init_x] = yb[0]; }
  __device__ static void rhs(float, const float* y, const float* p, const float*, float* dy) {
    const float gm = p[P_r] * (1.f - y[0]);
    dy[0] = y[0] * gm;
    VIHDS_UNROLL for (int k = 1; k < 4; ++k) dy[k] = 1.f - (gm + 1.f) * y[k];
  }
  __device__ static void rhs_vjp(float, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    const float gm = p[P_r] * (1.f - y[0]);
    float gmb = v[0] * y[0];
    yb[0] += v[0] * gm;
    VIHDS_UNROLL for (int k = 1; k < 4; ++k) {
      gmb -= v[k] * y[k];
      yb[k] -= v[k] * (gm + 1.f);
    }
    pb[P_r] += gmb * (1.f - y[0]);
    yb[0] -= gmb * p[P_r];
  }
};

// ---------------------------------------------------------------------------------------------
// prpr_constant                            reference: models/prpr_constant.py:13-85
// ---------------------------------------------------------------------------------------------
struct PrprConstant {
  static constexpr int N = 6;
  static constexpr int NS = 6;  // species handed to observe()
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;   // shared neural weights
  static constexpr int NC = 0;
  static constexpr int OBS = OBS_DEFAULT;
  enum Slot { S_r, S_K, S_tlag, S_rc, S_drfp, S_dyfp, S_dcfp, S_aYFP, S_aCFP, S_a530, S_a480,
              S_init_x, S_init_rfp, S_init_yfp, S_init_cfp, NSLOT };
  enum Par { P_r, P_K, P_tlag, P_rc, P_drfp, P_dyfp, P_dcfp, P_aYFP, P_aCFP, P_a530, P_a480, NP };
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"r", "K", "tlag", "rc", "drfp", "dyfp", "dcfp", "aYFP_PR", "aCFP_PR", "a530", "a480",
                              "init_x", "init_rfp", "init_yfp", "init_cfp"};
    return n[s];
  }
  __device__ static void prepare(const float* th, const float*, float* p) {
    p[P_r] = clampf(th[S_r], 0.f, 4.f);
    p[P_K] = clampf(th[S_K], 0.f, 4.f);
    p[P_tlag] = th[S_tlag];
    p[P_rc] = th[S_rc];
    p[P_drfp] = clampf(th[S_drfp], 1e-12f, 2.f);
    p[P_dyfp] = clampf(th[S_dyfp], 1e-12f, 2.f);
    p[P_dcfp] = clampf(th[S_dcfp], 1e-12f, 2.f);
    p[P_aYFP] = th[S_aYFP]; p[P_aCFP] = th[S_aCFP]; p[P_a530] = th[S_a530]; p[P_a480] = th[S_a480];
  }
  __device__ static void prepare_vjp(const float* th, const float*, const float*, const float* pb, float* thb) {
    thb[S_r] = pb[P_r] * clamp_pass(th[S_r], 0.f, 4.f);
    thb[S_K] = pb[P_K] * clamp_pass(th[S_K], 0.f, 4.f);
    thb[S_tlag] = pb[P_tlag];
    thb[S_rc] = pb[P_rc];
    thb[S_drfp] = pb[P_drfp] * clamp_pass(th[S_drfp], 1e-12f, 2.f);
    thb[S_dyfp] = pb[P_dyfp] * clamp_pass(th[S_dyfp], 1e-12f, 2.f);
    thb[S_dcfp] = pb[P_dcfp] * clamp_pass(th[S_dcfp], 1e-12f, 2.f);
    thb[S_aYFP] = pb[P_aYFP]; thb[S_aCFP] = pb[P_aCFP]; thb[S_a530] = pb[P_a530]; thb[S_a480] = pb[P_a480];
  }
  __device__ static void init(const float* th, const float*, float* y) {
    y[0] = th[S_init_x]; y[1] = th[S_init_rfp]; y[2] = th[S_init_yfp]; y[3] = th[S_init_cfp]; y[4] = 0.f; y[5] = 0.f;
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    thb[S_init_x] = yb[0]; thb[S_init_rfp] = yb[1]; thb[S_init_yfp] = yb[2]; thb[S_init_cfp] = yb[3];
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float*, float* dy) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float rc = p[P_rc], gm = G.gamma;
    dy[0] = gm * y[0];
    dy[1] = rc - (gm + p[P_drfp]) * y[1];
    dy[2] = rc * p[P_aYFP] - (gm + p[P_dyfp]) * y[2];
    dy[3] = rc * p[P_aCFP] - (gm + p[P_dcfp]) * y[3];
    dy[4] = rc * p[P_a530] - gm * y[4];
    dy[5] = rc * p[P_a480] - gm * y[5];
  }
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float rc = p[P_rc], gm = G.gamma;
    float gammab = v[0] * y[0] - v[1] * y[1] - v[2] * y[2] - v[3] * y[3] - v[4] * y[4] - v[5] * y[5];
    yb[0] += v[0] * gm;
    yb[1] -= v[1] * (gm + p[P_drfp]);
    yb[2] -= v[2] * (gm + p[P_dyfp]);
    yb[3] -= v[3] * (gm + p[P_dcfp]);
    yb[4] -= v[4] * gm;
    yb[5] -= v[5] * gm;
    pb[P_rc] += v[1] + v[2] * p[P_aYFP] + v[3] * p[P_aCFP] + v[4] * p[P_a530] + v[5] * p[P_a480];
    pb[P_drfp] -= v[1] * y[1];
    pb[P_dyfp] -= v[2] * y[2];
    pb[P_dcfp] -= v[3] * y[3];
    pb[P_aYFP] += v[2] * rc;
    pb[P_aCFP] += v[3] * rc;
    pb[P_a530] += v[4] * rc;
    pb[P_a480] += v[5] * rc;
    growth_vjp(G, y[0], p[P_r], p[P_K], gammab, yb[0], pb[P_r], pb[P_K], pb[P_tlag]);
  }
};

// ---------------------------------------------------------------------------------------------
// relay_constant                           reference: models/relay_constant.py:13-180 (equations only:
// the reference class raises at construction, SURVEY.md 2.1 -- parity vs own restatement)
// ---------------------------------------------------------------------------------------------
struct RelayConstant {
  static constexpr int N = 12;
  static constexpr int NS = 12;  // species handed to observe()
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;   // shared neural weights
  static constexpr int NC = 2;
  static constexpr int OBS = OBS_DEFAULT;
  enum Slot {
    S_r, S_K, S_tlag, S_rc, S_drfp, S_dyfp, S_dcfp, S_dR, S_dS, S_e76, S_e81, S_KGR76, S_KGS76, S_KGR81, S_KGS81,
    S_aYFP, S_aCFP, S_a530, S_a480, S_aR, S_aS, S_dluxI, S_dlasI, S_KC6, S_KC12, S_Klux, S_Klas,
    S_nR, S_nS, S_H0, S_H1, S_H2, S_H3,
    S_init_x, S_init_rfp, S_init_yfp, S_init_cfp, S_init_luxR, S_init_lasR, S_init_luxI, S_init_lasI, NSLOT
  };
  enum Par {
    P_r, P_K, P_tlag, P_rc, P_drfp, P_dyfp, P_dcfp, P_dR, P_dS, P_e76, P_e81, P_KGR76, P_KGS76, P_KGR81, P_KGS81,
    P_aYFP, P_aCFP, P_a530, P_a480, P_aR, P_aS, P_dluxI, P_dlasI, P_KC6, P_KC12, P_Klux, P_Klas, P_fR, P_fS, NP
  };
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"r", "K", "tlag", "rc", "drfp", "dyfp", "dcfp", "dR", "dS", "e76", "e81", "KGR_76",
                              "KGS_76", "KGR_81", "KGS_81", "aYFP", "aCFP", "a530", "a480", "aR", "aS", "dluxI",
                              "dlasI", "KC6", "KC12", "Klux", "Klas", "nR", "nS", "KR6", "KR12", "KS6", "KS12",
                              "init_x", "init_rfp", "init_yfp", "init_cfp", "init_luxR", "init_lasR", "init_luxI",
                              "init_lasI"};
    return n[s];
  }
  __device__ static void prepare(const float* th, const float* c, float* p) {
    p[P_r] = clampf(th[S_r], 0.f, 4.f);
    p[P_K] = clampf(th[S_K], 0.f, 4.f);
    p[P_tlag] = th[S_tlag];
    p[P_rc] = th[S_rc];
    p[P_drfp] = clampf(th[S_drfp], 1e-12f, 2.f);
    p[P_dyfp] = clampf(th[S_dyfp], 1e-12f, 2.f);
    p[P_dcfp] = clampf(th[S_dcfp], 1e-12f, 2.f);
    p[P_dR] = clampf(th[S_dR], 1e-12f, 5.f);
    p[P_dS] = clampf(th[S_dS], 1e-12f, 5.f);
    VIHDS_UNROLL for (int k = P_e76; k <= P_aS; ++k) p[k] = th[S_e76 + (k - P_e76)];
    p[P_dluxI] = clampf(th[S_dluxI], 1e-12f, 5.f);
    p[P_dlasI] = clampf(th[S_dlasI], 1e-12f, 5.f);
    p[P_KC6] = th[S_KC6]; p[P_KC12] = th[S_KC12]; p[P_Klux] = th[S_Klux]; p[P_Klas] = th[S_Klas];
    p[P_fR] = hill_frac(th[S_nR], th[S_H0], th[S_H1], c[0], c[1]);
    p[P_fS] = hill_frac(th[S_nS], th[S_H2], th[S_H3], c[0], c[1]);
  }
  __device__ static void prepare_vjp(const float* th, const float* c, const float*, const float* pb, float* thb) {
    thb[S_r] = pb[P_r] * clamp_pass(th[S_r], 0.f, 4.f);
    thb[S_K] = pb[P_K] * clamp_pass(th[S_K], 0.f, 4.f);
    thb[S_tlag] = pb[P_tlag];
    thb[S_rc] = pb[P_rc];
    thb[S_drfp] = pb[P_drfp] * clamp_pass(th[S_drfp], 1e-12f, 2.f);
    thb[S_dyfp] = pb[P_dyfp] * clamp_pass(th[S_dyfp], 1e-12f, 2.f);
    thb[S_dcfp] = pb[P_dcfp] * clamp_pass(th[S_dcfp], 1e-12f, 2.f);
    thb[S_dR] = pb[P_dR] * clamp_pass(th[S_dR], 1e-12f, 5.f);
    thb[S_dS] = pb[P_dS] * clamp_pass(th[S_dS], 1e-12f, 5.f);
    VIHDS_UNROLL for (int k = P_e76; k <= P_aS; ++k) thb[S_e76 + (k - P_e76)] = pb[k];
    thb[S_dluxI] = pb[P_dluxI] * clamp_pass(th[S_dluxI], 1e-12f, 5.f);
    thb[S_dlasI] = pb[P_dlasI] * clamp_pass(th[S_dlasI], 1e-12f, 5.f);
    thb[S_KC6] = pb[P_KC6]; thb[S_KC12] = pb[P_KC12]; thb[S_Klux] = pb[P_Klux]; thb[S_Klas] = pb[P_Klas];
    hill_frac_vjp(th[S_nR], th[S_H0], th[S_H1], c[0], c[1], pb[P_fR], thb[S_nR], thb[S_H0], thb[S_H1]);
    hill_frac_vjp(th[S_nS], th[S_H2], th[S_H3], c[0], c[1], pb[P_fS], thb[S_nS], thb[S_H2], thb[S_H3]);
  }
  __device__ static void init(const float* th, const float* c, float* y) {
    y[0] = th[S_init_x]; y[1] = th[S_init_rfp]; y[2] = th[S_init_yfp]; y[3] = th[S_init_cfp];
    y[4] = 0.f; y[5] = 0.f; y[6] = th[S_init_luxR]; y[7] = th[S_init_lasR];
    y[8] = th[S_init_luxI]; y[9] = th[S_init_lasI]; y[10] = c[0]; y[11] = c[1];
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    thb[S_init_x] = yb[0]; thb[S_init_rfp] = yb[1]; thb[S_init_yfp] = yb[2]; thb[S_init_cfp] = yb[3];
    thb[S_init_luxR] = yb[6]; thb[S_init_lasR] = yb[7]; thb[S_init_luxI] = yb[8]; thb[S_init_lasI] = yb[9];
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float*, float* dy) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float bR = y[6] * y[6] * p[P_fR], bS = y[7] * y[7] * p[P_fS];
    float d76, d81;
    float P76 = promoter(p[P_e76], p[P_KGR76], p[P_KGS76], bR, bS, d76);
    float P81 = promoter(p[P_e81], p[P_KGR81], p[P_KGS81], bR, bS, d81);
    float rc = p[P_rc], gm = G.gamma;
    dy[0] = gm * y[0];
    dy[1] = rc - (gm + p[P_drfp]) * y[1];
    dy[2] = rc * p[P_aYFP] * P81 - (gm + p[P_dyfp]) * y[2];
    dy[3] = rc * p[P_aCFP] * P76 - (gm + p[P_dcfp]) * y[3];
    dy[4] = rc * p[P_a530] - gm * y[4];
    dy[5] = rc * p[P_a480] - gm * y[5];
    dy[6] = rc * p[P_aR] - (gm + p[P_dR]) * y[6];
    dy[7] = rc * p[P_aS] - (gm + p[P_dS]) * y[7];
    dy[8] = rc * P81 - (gm + p[P_dluxI]) * y[8];
    dy[9] = rc * P76 - (gm + p[P_dlasI]) * y[9];
    dy[10] = fdiv(p[P_KC6] * rc * y[0] * y[8], 1.f + fdiv(y[8], p[P_Klux]));
    dy[11] = fdiv(p[P_KC12] * rc * y[0] * y[9], 1.f + fdiv(y[9], p[P_Klas]));
  }
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float bR = y[6] * y[6] * p[P_fR], bS = y[7] * y[7] * p[P_fS];
    float d76, d81;
    float P76 = promoter(p[P_e76], p[P_KGR76], p[P_KGS76], bR, bS, d76);
    float P81 = promoter(p[P_e81], p[P_KGR81], p[P_KGS81], bR, bS, d81);
    float rc = p[P_rc], gm = G.gamma;
    float gammab = v[0] * y[0] - v[1] * y[1] - v[2] * y[2] - v[3] * y[3] - v[4] * y[4] - v[5] * y[5] - v[6] * y[6] -
                   v[7] * y[7] - v[8] * y[8] - v[9] * y[9];
    yb[0] += v[0] * gm;
    yb[1] -= v[1] * (gm + p[P_drfp]);
    yb[2] -= v[2] * (gm + p[P_dyfp]);
    yb[3] -= v[3] * (gm + p[P_dcfp]);
    yb[4] -= v[4] * gm;
    yb[5] -= v[5] * gm;
    yb[6] -= v[6] * (gm + p[P_dR]);
    yb[7] -= v[7] * (gm + p[P_dS]);
    yb[8] -= v[8] * (gm + p[P_dluxI]);
    yb[9] -= v[9] * (gm + p[P_dlasI]);
    // d_c6 = (KC6*rc*x*luxI)/(1 + luxI/Klux)
    float iKlux = frcp(p[P_Klux]), iKlas = frcp(p[P_Klas]);
    float den6 = 1.f + y[8] * iKlux, den12 = 1.f + y[9] * iKlas;
    float num6 = p[P_KC6] * rc * y[0] * y[8], num12 = p[P_KC12] * rc * y[0] * y[9];
    float n6b = fdiv(v[10], den6), n12b = fdiv(v[11], den12);
    float d6b = -n6b * fdiv(num6, den6), d12b = -n12b * fdiv(num12, den12);
    pb[P_KC6] += n6b * rc * y[0] * y[8];
    pb[P_KC12] += n12b * rc * y[0] * y[9];
    yb[0] += n6b * p[P_KC6] * rc * y[8] + n12b * p[P_KC12] * rc * y[9];
    yb[8] += n6b * p[P_KC6] * rc * y[0] + d6b * iKlux;
    yb[9] += n12b * p[P_KC12] * rc * y[0] + d12b * iKlas;
    pb[P_Klux] -= d6b * y[8] * iKlux * iKlux;
    pb[P_Klas] -= d12b * y[9] * iKlas * iKlas;
    pb[P_rc] += v[1] + v[2] * p[P_aYFP] * P81 + v[3] * p[P_aCFP] * P76 + v[4] * p[P_a530] + v[5] * p[P_a480] +
                v[6] * p[P_aR] + v[7] * p[P_aS] + v[8] * P81 + v[9] * P76 + n6b * p[P_KC6] * y[0] * y[8] +
                n12b * p[P_KC12] * y[0] * y[9];
    pb[P_drfp] -= v[1] * y[1];
    pb[P_dyfp] -= v[2] * y[2];
    pb[P_dcfp] -= v[3] * y[3];
    pb[P_dR] -= v[6] * y[6];
    pb[P_dS] -= v[7] * y[7];
    pb[P_dluxI] -= v[8] * y[8];
    pb[P_dlasI] -= v[9] * y[9];
    pb[P_aYFP] += v[2] * rc * P81;
    pb[P_aCFP] += v[3] * rc * P76;
    pb[P_a530] += v[4] * rc;
    pb[P_a480] += v[5] * rc;
    pb[P_aR] += v[6] * rc;
    pb[P_aS] += v[7] * rc;
    float P81b = v[2] * rc * p[P_aYFP] + v[8] * rc, P76b = v[3] * rc * p[P_aCFP] + v[9] * rc;
    float bRb = 0.f, bSb = 0.f;
    promoter_vjp(P76, d76, p[P_KGR76], p[P_KGS76], bR, bS, P76b, pb[P_e76], pb[P_KGR76], pb[P_KGS76], bRb, bSb);
    promoter_vjp(P81, d81, p[P_KGR81], p[P_KGS81], bR, bS, P81b, pb[P_e81], pb[P_KGR81], pb[P_KGS81], bRb, bSb);
    yb[6] += bRb * 2.f * y[6] * p[P_fR];
    yb[7] += bSb * 2.f * y[7] * p[P_fS];
    pb[P_fR] += bRb * y[6] * y[6];
    pb[P_fS] += bSb * y[7] * y[7];
    growth_vjp(G, y[0], p[P_r], p[P_K], gammab, yb[0], pb[P_r], pb[P_K], pb[P_tlag]);
  }
};

// ---------------------------------------------------------------------------------------------
// degrader_constant                        reference: models/degrader_constant.py:13-190 (equations only:
// the reference class raises at construction, SURVEY.md 2.1 -- parity vs own restatement)
// ---------------------------------------------------------------------------------------------
struct DegraderConstant {
  static constexpr int N = 11;
  static constexpr int NS = 11;  // species handed to observe()
  static constexpr bool NEURAL_PREC = false;
  static constexpr int NW = 0;   // shared neural weights
  static constexpr int NC = 3;
  static constexpr int OBS = OBS_DEFAULT;
  enum Slot {
    S_r, S_K, S_tlag, S_rc, S_drfp, S_dyfp, S_dcfp, S_dR, S_dS, S_e76, S_e81, S_KGR76, S_KGS76, S_KGR81, S_KGS81,
    S_aYFP, S_aCFP, S_a530, S_a480, S_aR, S_aS, S_aI, S_daiiA, S_dA6, S_dA12, S_nA, S_eA, S_KAra,
    S_nR, S_nS, S_H0, S_H1, S_H2, S_H3,
    S_init_x, S_init_rfp, S_init_yfp, S_init_cfp, S_init_luxR, S_init_lasR, S_init_aiiA, NSLOT
  };
  enum Par {
    P_r, P_K, P_tlag, P_rc, P_drfp, P_dyfp, P_dcfp, P_dR, P_dS, P_e76, P_e81, P_KGR76, P_KGS76, P_KGR81, P_KGS81,
    P_aYFP, P_aCFP, P_a530, P_a480, P_aR, P_aS, P_aI, P_daiiA, P_PBAD, P_rC6, P_rC12, P_fR, P_fS, NP
  };
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"r", "K", "tlag", "rc", "drfp", "dyfp", "dcfp", "dR", "dS", "e76", "e81", "KGR_76",
                              "KGS_76", "KGR_81", "KGS_81", "aYFP", "aCFP", "a530", "a480", "aR", "aS", "aI",
                              "daiiA", "dA6", "dA12", "nA", "eA", "KAra", "nR", "nS", "KR6", "KR12", "KS6", "KS12",
                              "init_x", "init_rfp", "init_yfp", "init_cfp", "init_luxR", "init_lasR", "init_aiiA"};
    return n[s];
  }
  __device__ static void prepare(const float* th, const float* c, float* p) {
    p[P_r] = clampf(th[S_r], 0.f, 4.f);
    p[P_K] = clampf(th[S_K], 0.f, 4.f);
    p[P_tlag] = th[S_tlag];
    p[P_rc] = th[S_rc];
    p[P_drfp] = clampf(th[S_drfp], 1e-12f, 2.f);
    p[P_dyfp] = clampf(th[S_dyfp], 1e-12f, 2.f);
    p[P_dcfp] = clampf(th[S_dcfp], 1e-12f, 2.f);
    p[P_dR] = clampf(th[S_dR], 1e-12f, 5.f);
    p[P_dS] = clampf(th[S_dS], 1e-12f, 5.f);
    VIHDS_UNROLL for (int k = P_e76; k <= P_daiiA; ++k) p[k] = th[S_e76 + (k - P_e76)];
    // PBAD = (ara^nA + eA*KAra^nA) / (ara^nA + KAra^nA)          degrader_constant.py:80-84
    float nA = clampf(th[S_nA], 0.5f, 3.f);
    float an = powf(c[2], nA), kn = powf(th[S_KAra], nA);
    p[P_PBAD] = (an + th[S_eA] * kn) / (an + kn);
    p[P_rC6] = th[S_dA6] * c[0];
    p[P_rC12] = th[S_dA12] * c[1];
    p[P_fR] = hill_frac(th[S_nR], th[S_H0], th[S_H1], c[0], c[1]);
    p[P_fS] = hill_frac(th[S_nS], th[S_H2], th[S_H3], c[0], c[1]);
  }
  __device__ static void prepare_vjp(const float* th, const float* c, const float*, const float* pb, float* thb) {
    thb[S_r] = pb[P_r] * clamp_pass(th[S_r], 0.f, 4.f);
    thb[S_K] = pb[P_K] * clamp_pass(th[S_K], 0.f, 4.f);
    thb[S_tlag] = pb[P_tlag];
    thb[S_rc] = pb[P_rc];
    thb[S_drfp] = pb[P_drfp] * clamp_pass(th[S_drfp], 1e-12f, 2.f);
    thb[S_dyfp] = pb[P_dyfp] * clamp_pass(th[S_dyfp], 1e-12f, 2.f);
    thb[S_dcfp] = pb[P_dcfp] * clamp_pass(th[S_dcfp], 1e-12f, 2.f);
    thb[S_dR] = pb[P_dR] * clamp_pass(th[S_dR], 1e-12f, 5.f);
    thb[S_dS] = pb[P_dS] * clamp_pass(th[S_dS], 1e-12f, 5.f);
    VIHDS_UNROLL for (int k = P_e76; k <= P_daiiA; ++k) thb[S_e76 + (k - P_e76)] = pb[k];
    float nA = clampf(th[S_nA], 0.5f, 3.f);
    float an = powf(c[2], nA), kn = powf(th[S_KAra], nA);
    float num = an + th[S_eA] * kn, den = an + kn;
    float numb = pb[P_PBAD] / den, denb = -pb[P_PBAD] * num / (den * den);
    float anb = numb + denb, knb = numb * th[S_eA] + denb;
    float nAb = 0.f, dummy = 0.f, KAb = 0.f;
    pow_vjp(c[2], nA, an, anb, dummy, nAb);
    pow_vjp(th[S_KAra], nA, kn, knb, KAb, nAb);
    thb[S_nA] = nAb * clamp_pass(th[S_nA], 0.5f, 3.f);
    thb[S_eA] = numb * kn;
    thb[S_KAra] = KAb;
    thb[S_dA6] = pb[P_rC6] * c[0];
    thb[S_dA12] = pb[P_rC12] * c[1];
    hill_frac_vjp(th[S_nR], th[S_H0], th[S_H1], c[0], c[1], pb[P_fR], thb[S_nR], thb[S_H0], thb[S_H1]);
    hill_frac_vjp(th[S_nS], th[S_H2], th[S_H3], c[0], c[1], pb[P_fS], thb[S_nS], thb[S_H2], thb[S_H3]);
  }
  __device__ static void init(const float* th, const float* c, float* y) {
    y[0] = th[S_init_x]; y[1] = th[S_init_rfp]; y[2] = th[S_init_yfp]; y[3] = th[S_init_cfp];
    y[4] = 0.f; y[5] = 0.f; y[6] = th[S_init_luxR]; y[7] = th[S_init_lasR]; y[8] = th[S_init_aiiA];
    y[9] = c[0]; y[10] = c[1];
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    thb[S_init_x] = yb[0]; thb[S_init_rfp] = yb[1]; thb[S_init_yfp] = yb[2]; thb[S_init_cfp] = yb[3];
    thb[S_init_luxR] = yb[6]; thb[S_init_lasR] = yb[7]; thb[S_init_aiiA] = yb[8];
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float*, float* dy) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float bR = y[6] * y[6] * p[P_fR], bS = y[7] * y[7] * p[P_fS];
    float d76, d81;
    float P76 = promoter(p[P_e76], p[P_KGR76], p[P_KGS76], bR, bS, d76);
    float P81 = promoter(p[P_e81], p[P_KGR81], p[P_KGS81], bR, bS, d81);
    float rc = p[P_rc], gm = G.gamma;
    dy[0] = gm * y[0];
    dy[1] = rc - (gm + p[P_drfp]) * y[1];
    dy[2] = rc * p[P_aYFP] * P81 - (gm + p[P_dyfp]) * y[2];
    dy[3] = rc * p[P_aCFP] * P76 - (gm + p[P_dcfp]) * y[3];
    dy[4] = rc * p[P_a530] - gm * y[4];
    dy[5] = rc * p[P_a480] - gm * y[5];
    dy[6] = rc * p[P_aR] - (gm + p[P_dR]) * y[6];
    dy[7] = rc * p[P_aS] - (gm + p[P_dS]) * y[7];
    dy[8] = rc * p[P_aI] * p[P_PBAD] - (p[P_daiiA] + (gm * y[8]));  // as written at degrader_constant.py:136
    dy[9] = y[0] * p[P_rC6] * y[8];
    dy[10] = y[0] * p[P_rC12] * y[8];
  }
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float*, const float* v, float* yb,
                                 float* pb) {
    Growth G = growth(t, y[0], p[P_r], p[P_K], p[P_tlag]);
    float bR = y[6] * y[6] * p[P_fR], bS = y[7] * y[7] * p[P_fS];
    float d76, d81;
    float P76 = promoter(p[P_e76], p[P_KGR76], p[P_KGS76], bR, bS, d76);
    float P81 = promoter(p[P_e81], p[P_KGR81], p[P_KGS81], bR, bS, d81);
    float rc = p[P_rc], gm = G.gamma;
    float gammab = v[0] * y[0] - v[1] * y[1] - v[2] * y[2] - v[3] * y[3] - v[4] * y[4] - v[5] * y[5] - v[6] * y[6] -
                   v[7] * y[7] - v[8] * y[8];
    yb[0] += v[0] * gm + v[9] * p[P_rC6] * y[8] + v[10] * p[P_rC12] * y[8];
    yb[1] -= v[1] * (gm + p[P_drfp]);
    yb[2] -= v[2] * (gm + p[P_dyfp]);
    yb[3] -= v[3] * (gm + p[P_dcfp]);
    yb[4] -= v[4] * gm;
    yb[5] -= v[5] * gm;
    yb[6] -= v[6] * (gm + p[P_dR]);
    yb[7] -= v[7] * (gm + p[P_dS]);
    yb[8] += -v[8] * gm + v[9] * y[0] * p[P_rC6] + v[10] * y[0] * p[P_rC12];
    pb[P_rC6] += v[9] * y[0] * y[8];
    pb[P_rC12] += v[10] * y[0] * y[8];
    pb[P_rc] += v[1] + v[2] * p[P_aYFP] * P81 + v[3] * p[P_aCFP] * P76 + v[4] * p[P_a530] + v[5] * p[P_a480] +
                v[6] * p[P_aR] + v[7] * p[P_aS] + v[8] * p[P_aI] * p[P_PBAD];
    pb[P_aI] += v[8] * rc * p[P_PBAD];
    pb[P_PBAD] += v[8] * rc * p[P_aI];
    pb[P_daiiA] -= v[8];
    pb[P_drfp] -= v[1] * y[1];
    pb[P_dyfp] -= v[2] * y[2];
    pb[P_dcfp] -= v[3] * y[3];
    pb[P_dR] -= v[6] * y[6];
    pb[P_dS] -= v[7] * y[7];
    pb[P_aYFP] += v[2] * rc * P81;
    pb[P_aCFP] += v[3] * rc * P76;
    pb[P_a530] += v[4] * rc;
    pb[P_a480] += v[5] * rc;
    pb[P_aR] += v[6] * rc;
    pb[P_aS] += v[7] * rc;
    float P81b = v[2] * rc * p[P_aYFP], P76b = v[3] * rc * p[P_aCFP];
    float bRb = 0.f, bSb = 0.f;
    promoter_vjp(P76, d76, p[P_KGR76], p[P_KGS76], bR, bS, P76b, pb[P_e76], pb[P_KGR76], pb[P_KGS76], bRb, bSb);
    promoter_vjp(P81, d81, p[P_KGR81], p[P_KGS81], bR, bS, P81b, pb[P_e81], pb[P_KGR81], pb[P_KGS81], bRb, bSb);
    yb[6] += bRb * 2.f * y[6] * p[P_fR];
    yb[7] += bSb * 2.f * y[7] * p[P_fS];
    pb[P_fR] += bRb * y[6] * y[6];
    pb[P_fS] += bSb * y[7] * y[7];
    growth_vjp(G, y[0], p[P_r], p[P_K], gammab, yb[0], pb[P_r], pb[P_K], pb[P_tlag]);
  }
};

// ---------------------------------------------------------------------------------------------
// <white-box model> + neural precisions with no hidden layer     reference: vihds/precisions.py:55-61,76-87
//   state = [core species (NS), 4 precisions];  input x = [t, species];  h = tanh(x) (activation on the INPUT)
//   d prec_j/dt = sigmoid(Wp[j].h + bp[j]) - sigmoid(Wd[j].h + bd[j]) * prec_j
// weights buffer (NeuralPrecisions.flat_weights): Wp [4][NIN], bp [4], Wd [4][NIN], bd [4]   (NIN = NS + 1)
// (every reference spec that pairs a white-box model with neural precisions sets n_hidden_decoder_precisions: 0)
// ---------------------------------------------------------------------------------------------
template <class Core>
struct WithPrec {
  static constexpr int NS = Core::N;
  static constexpr int N = Core::N + 4;
  static constexpr int NC = Core::NC;
  static constexpr int OBS = Core::OBS;
  static constexpr int NSLOT = Core::NSLOT + 4;
  static constexpr int NP = Core::NP + 1;  // + the number of hidden units H (as integer bits; set by the kernels)
  static constexpr int NIN = Core::N + 1;
  static constexpr int NW = 2 * (4 * NIN + 4);  // weights without a hidden layer (H = 0)
  static constexpr bool NEURAL_PREC = true;
  static constexpr int O_WP = 0, O_BP = 4 * NIN, O_WD = 4 * NIN + 4, O_BD = 8 * NIN + 4;
  // With a hidden layer (reference precisions.py:63-74; params.n_hidden_decoder_precisions = H >= 1, the reference's
  // default is 20, config.py:77): weights = Wh [H][NIN], bh [H], Wp [4][H], bp [4], Wd [4][H], bd [4]; the layer input is
  // [t, species] as it is, the hidden layer tanh(Wh x + bh), shared by production and degradation.
  __host__ __device__ static int n_weights(int H) { return H < 1 ? NW : H * NIN + H + 2 * (4 * H + 4); }
  __host__ __device__ static int o_bp(int H) { return H < 1 ? O_BP : H * NIN + H + 4 * H; }
  __host__ __device__ static int o_bd(int H) { return H < 1 ? O_BD : H * NIN + H + 4 * H + 4 + 4 * H; }
  __host__ __device__ static int dump_fields(int H) { return 8 + NIN + (H < 1 ? 0 : 2 * H); }
  __device__ static int hidden_units(const float* p) { return __builtin_amdgcn_readfirstlane(__float_as_int(p[Core::NP])); }
  __host__ static const char* slot_name(int s) {
    static const char* n[] = {"init_prec_x", "init_prec_rfp", "init_prec_yfp", "init_prec_cfp"};
    return s < Core::NSLOT ? Core::slot_name(s) : n[s - Core::NSLOT];
  }
  __device__ static void prepare(const float* th, const float* c, float* p) { Core::prepare(th, c, p); }
  __device__ static void prepare_vjp(const float* th, const float* c, const float* p, const float* pb, float* thb) {
    Core::prepare_vjp(th, c, p, pb, thb);
  }
  __device__ static void init(const float* th, const float* c, float* y) {
    Core::init(th, c, y);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) y[NS + j] = th[Core::NSLOT + j];
  }
  __device__ static void init_vjp(const float* yb, float* thb) {
    Core::init_vjp(yb, thb);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) thb[Core::NSLOT + j] = yb[NS + j];
  }
  __device__ static void hidden(float t, const float* y, float* h) {
    h[0] = ftanh(t);
    VIHDS_UNROLL for (int i = 0; i < NS; ++i) h[i + 1] = ftanh(y[i]);
  }
  // The weights are read through the constant address space: the index is uniform, so these are scalar loads into
  // SGPRs (s_load_dwordx16, one FMA operand each) -- no LDS staging and no 104 ds_reads per evaluation.  The buffer
  // is never written by the kernels that read it (gradients go to g_weights / the dump).
  typedef const __attribute__((address_space(4))) float* weights_ptr;
  // hidden-layer variant: pre-activations of the two output layers, one hidden unit at a time (H is a run-time size)
  __device__ static void hidden_forward(weights_ptr w, int H, const float* x, float* za, float* zd) {
    const weights_ptr Wh = w, bh = w + H * NIN, Wp = bh + H, bp = Wp + 4 * H, Wd = bp + 4, bd = Wd + 4 * H;
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) { za[j] = bp[j]; zd[j] = bd[j]; }
    for (int u = 0; u < H; ++u) {
      float z = bh[u];
      VIHDS_UNROLL for (int i = 0; i < NIN; ++i) z = fmaf(Wh[u * NIN + i], x[i], z);
      const float hu = ftanh(z);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        za[j] = fmaf(Wp[j * H + u], hu, za[j]);
        zd[j] = fmaf(Wd[j * H + u], hu, zd[j]);
      }
    }
  }
  __device__ static void rhs(float t, const float* y, const float* p, const float* wg, float* dy) {
    const weights_ptr w = (weights_ptr)wg;
    Core::rhs(t, y, p, wg, dy);
    __asm__ volatile("" ::: "memory");  // keep the weight loads inside the time loop (no hoist-and-spill)
    const int H = hidden_units(p);
    if (H > 0) {
      float x[NIN], za[4], zd[4];
      x[0] = t;
      VIHDS_UNROLL for (int i = 0; i < NS; ++i) x[i + 1] = y[i];
      hidden_forward(w, H, x, za, zd);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) dy[NS + j] = sigmoid_f(za[j]) - sigmoid_f(zd[j]) * y[NS + j];
      return;
    }
    float h[NIN];
    hidden(t, y, h);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      float za = w[O_BP + j], zd = w[O_BD + j];
      VIHDS_UNROLL for (int i = 0; i < NIN; ++i) {
        za += w[O_WP + j * NIN + i] * h[i];
        zd += w[O_WD + j * NIN + i] * h[i];
      }
      dy[NS + j] = sigmoid_f(za) - sigmoid_f(zd) * y[NS + j];
    }
  }
  template <class Ctx>
  __device__ static void rhs_vjp(float t, const float* y, const float* p, const float* wg, const float* v, float* yb,
                                 float* pb, Ctx& ctx) {
    const weights_ptr w = (weights_ptr)wg;
    Core::rhs_vjp(t, y, p, wg, v, yb, pb);
    __asm__ volatile("" ::: "memory");
    const int H = hidden_units(p);
    if (H > 0) {
      // dump fields (Ctx::DUMP): 0..3 zab, 4..7 zdb, 8.. the NIN layer inputs x, then H hidden pre-activation adjoints,
      // then the H hidden activations; the three weight matrices are rectangles of that dump (vihds_gram_blocks).
      // Without a dump only the state adjoint is produced (the API refuses weight gradients without aux).
      const weights_ptr Wh = w, bh = w + H * NIN, Wp = bh + H, Wd = Wp + 4 * H + 4;
      float x[NIN], xb[NIN], za[4], zd[4], zab[4], zdb[4];
      x[0] = t;
      VIHDS_UNROLL for (int i = 0; i < NS; ++i) x[i + 1] = y[i];
      hidden_forward(w, H, x, za, zd);
      float* D = nullptr;
      size_t fs = 0;
      if constexpr (Ctx::DUMP) {
        D = ctx.dump + (size_t)ctx.e * ctx.n;
        fs = ctx.fstride;
        ctx.e += 1;
        VIHDS_UNROLL for (int i = 0; i < NIN; ++i) D[(size_t)(8 + i) * fs] = x[i];
      }
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float a = sigmoid_f(za[j]), d = sigmoid_f(zd[j]);
        const float vj = v[NS + j];
        yb[NS + j] -= vj * d;
        zab[j] = vj * a * (1.f - a);
        zdb[j] = -vj * y[NS + j] * d * (1.f - d);
        if constexpr (Ctx::DUMP) {
          D[(size_t)j * fs] = zab[j];
          D[(size_t)(4 + j) * fs] = zdb[j];
          ctx.bsum[j] += zab[j];
          ctx.bsum[4 + j] += zdb[j];
        }
      }
      VIHDS_UNROLL for (int i = 0; i < NIN; ++i) xb[i] = 0.f;
      for (int u = 0; u < H; ++u) {
        float z = bh[u];
        VIHDS_UNROLL for (int i = 0; i < NIN; ++i) z = fmaf(Wh[u * NIN + i], x[i], z);
        const float hu = ftanh(z);
        float hub = 0.f;
        VIHDS_UNROLL for (int j = 0; j < 4; ++j) hub += Wp[j * H + u] * zab[j] + Wd[j * H + u] * zdb[j];
        const float zub = hub * (1.f - hu * hu);
        VIHDS_UNROLL for (int i = 0; i < NIN; ++i) xb[i] = fmaf(Wh[u * NIN + i], zub, xb[i]);
        if constexpr (Ctx::DUMP) {
          D[(size_t)(8 + NIN + u) * fs] = zub;
          D[(size_t)(8 + NIN + H + u) * fs] = hu;
        }
      }
      VIHDS_UNROLL for (int i = 0; i < NS; ++i) yb[i] += xb[i + 1];
      return;
    }
    float h[NIN], hb[NIN];
    hidden(t, y, h);
    VIHDS_UNROLL for (int i = 0; i < NIN; ++i) hb[i] = 0.f;
    float* D = nullptr;
    size_t fs = 0;
    if constexpr (Ctx::DUMP) {  // fields of the dump: 0..3 zab, 4..7 zdb, 8.. the layer inputs h
      D = ctx.dump + (size_t)ctx.e * ctx.n;
      fs = ctx.fstride;
      ctx.e += 1;
      VIHDS_UNROLL for (int i = 0; i < NIN; ++i) D[(size_t)(8 + i) * fs] = h[i];
    }
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      float za = w[O_BP + j], zd = w[O_BD + j];
      VIHDS_UNROLL for (int i = 0; i < NIN; ++i) {
        za += w[O_WP + j * NIN + i] * h[i];
        zd += w[O_WD + j * NIN + i] * h[i];
      }
      const float a = sigmoid_f(za), d = sigmoid_f(zd);
      const float vj = v[NS + j];
      yb[NS + j] -= vj * d;
      const float zab = vj * a * (1.f - a);
      const float zdb = -vj * y[NS + j] * d * (1.f - d);
      if constexpr (Ctx::DUMP) {
        D[(size_t)j * fs] = zab;
        D[(size_t)(4 + j) * fs] = zdb;
        ctx.bsum[j] += zab;
        ctx.bsum[4 + j] += zdb;
      } else {
        ctx.wb[O_BP + j] += zab;
        ctx.wb[O_BD + j] += zdb;
      }
      VIHDS_UNROLL for (int i = 0; i < NIN; ++i) {
        if constexpr (!Ctx::DUMP) {
          ctx.wb[O_WP + j * NIN + i] += zab * h[i];
          ctx.wb[O_WD + j * NIN + i] += zdb * h[i];
        }
        hb[i] += w[O_WP + j * NIN + i] * zab + w[O_WD + j * NIN + i] * zdb;
      }
    }
    VIHDS_UNROLL for (int i = 0; i < NS; ++i) yb[i] += hb[i + 1] * (1.f - h[i + 1] * h[i + 1]);
  }
};

}  // namespace vihds
