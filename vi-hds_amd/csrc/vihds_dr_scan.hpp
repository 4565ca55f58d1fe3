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
// two trajectories per wavefront, all 8 species of a step in one lane's registers.  Between the barrier behind the first
// stage (the block's shared time grid) and the epilogue's, a wavefront never waits for another one: its records sit in its
// own slice of LDS.
//   0. (decoder step) sampling stage: theta = clip(sample(q, u)), log q, log p, the device conditioner's rows
//   1. coefficients  |w| h r sigmoid(4 (t_s - tlag)) of this lane's steps                   (state independent, registers)
//   2. x chain       u_{k+1} = Phi_k(u_k), the one nonlinear recurrence, solved over the lanes by Newton's method: every
//                    lane walks its own steps from its current first value with the derivative, the linearised recurrence
//                    is an affine scan (round 4; before: 85 steps walked by one wavefront); stage values u[k][s] -> LDS
//   3. rfp, W (f530 = a530 W, f480 = a480 W), luxR, lasR: per-step affine maps, composed per lane, DPP scan over lanes
//   4. promoters at the stage values of luxR / lasR -> yfp, cfp the same way
//   5. log-likelihood at the grid points, segment sum
//   6. adjoint: reverse scans for yfp, cfp -> luxR, lasR (+ rfp, W) -> x, each followed by the per-step VJPs whose
//      sums over k are the parameter gradients (segment sums), then the epilogue shared with the lane kernels' algebra.
// Against the 8-lanes-per-trajectory kernel (vihds_dr_lanes.hpp): about half the instructions per trajectory, and
// 3 600 wavefronts instead of 900 at B=36, S=200.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "vihds_dr_lanes.hpp"

namespace vihds {

// ---- explicit Runge-Kutta tableaux of the five fixed-grid schemes, steps in units of h ----------
// value types of the step functions: float, or two species side by side
typedef float v2f __attribute__((ext_vector_type(2)));
template <class T> __device__ __forceinline__ T splat(float v);
template <> __device__ __forceinline__ float splat<float>(float v) { return v; }
template <> __device__ __forceinline__ v2f splat<v2f>(float v) { return v2f{v, v}; }
__device__ __forceinline__ float fm(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ v2f fm(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f fm(float a, v2f b, v2f c) { return __builtin_elementwise_fma(v2f{a, a}, b, c); }

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

  // The three step functions below are written over a value type T: float, or v2f = two species side by side, which the
  // compiler turns into packed fp32 instructions (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32: two lanes' worth of
  // arithmetic per issue slot) -- the species come in pairs with identical arithmetic, (rfp, W), (luxR, lasR),
  // (yfp, cfp), and at two wavefronts per SIMD the kernel's time is its instruction count.
  // one step of dy = F_s - a_s y as the affine map y' = A y + B
  template <class T>
  __device__ __forceinline__ static void affine(float h, const T* a_s, const T* F, T& A, T& B) {
    T kap[NS], rho[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      T al = splat<T>(1.f), be = splat<T>(0.f);
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (a(s, r) != 0.f) {
          al = fm(h * a(s, r), kap[r], al);
          be = fm(h * a(s, r), rho[r], be);
        }
      kap[s] = -a_s[s] * al;
      rho[s] = fm(-a_s[s], be, F[s]);
    }
    A = splat<T>(1.f);
    B = splat<T>(0.f);
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (b(s) != 0.f) {
        A = fm(h * b(s), kap[s], A);
        B = fm(h * b(s), rho[s], B);
      }
  }
  // the step itself: stage values Y[s], returns y'
  template <class T>
  __device__ __forceinline__ static T real(float h, const T* a_s, const T* F, T y, T* Y) {
    T k[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
      T v = y;
      VIHDS_UNROLL for (int r = 0; r < s; ++r)
        if (a(s, r) != 0.f) v = fm(h * a(s, r), k[r], v);
      Y[s] = v;
      k[s] = fm(-a_s[s], v, F[s]);
    }
    T o = y;
    VIHDS_UNROLL for (int s = 0; s < NS; ++s)
      if (b(s) != 0.f) o = fm(h * b(s), k[s], o);
    return o;
  }
  // transposed step: lam1 = adjoint of y', J[s] = adjoints arriving at the stage values from elsewhere;
  // returns the adjoint of y, kbar[s] = adjoints of the stage derivatives.  Linear in (lam1, J).
  template <class T>
  __device__ __forceinline__ static T reverse(float h, const T* a_s, T lam1, const T* J, T* kbar) {
    T Yb[NS];
    T lam = lam1;
    VIHDS_UNROLL for (int s = NS - 1; s >= 0; --s) {
      T kb = (h * b(s)) * lam1;
      VIHDS_UNROLL for (int r = s + 1; r < NS; ++r)
        if (a(r, s) != 0.f) kb = fm(h * a(r, s), Yb[r], kb);
      kbar[s] = kb;
      Yb[s] = fm(-a_s[s], kb, J[s]);
      lam += Yb[s];
    }
    return lam;
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

// ---- the x chain: x in units of K (u = x / K), du = gr u (1 - u) ------------------------------------------------------
// With p_r = u_r - u_r^2 at the stages, every later stage value (and the step's result) is
//     u_{s'} = u + sum_{r < s'} w(s', r) h gr_r p_r,       w(s', r) = a(s', r),  w(NS, r) = b(r),
// so with the products |w| h gr_r TABULATED per step (state independent: r sigmoid(4 (t - tlag)) only) a step is one FMA
// per nonzero tableau weight plus one per p_r, all of them off the dependent chain except two per stage
// (p_s, then the last FMA of u_{s+1}).  Weights of equal magnitude on the same stage share a table entry (the sign
// goes into the FMA): rk4 (3/8 rule) 8 entries for 10 weights, modeuler 3, midpoint 2, euler 1.
template <int I0, int I1, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I0 < I1) {
    f(std::integral_constant<int, I0>{});
    static_for<I0 + 1, I1>(f);
  }
}
template <int SOLVER>
struct RkChain {
  using R = Rk<SOLVER>;
  static constexpr int NS = R::NS;
  static constexpr float wgt(int s1, int r) { return s1 < NS ? R::a(s1, r) : R::b(r); }
  static constexpr float cabs(float x) { return x < 0.f ? -x : x; }
  // (s1, r) is the first weight of its magnitude on stage r, in the order s1 = 1..NS
  static constexpr bool first(int s1, int r) {
    for (int s2 = r + 1; s2 < s1; ++s2)
      if (wgt(s2, r) != 0.f && cabs(wgt(s2, r)) == cabs(wgt(s1, r))) return false;
    return true;
  }
  static constexpr int index(int s1, int r) {  // table entry of weight (s1, r); -1: the weight is zero
    if (wgt(s1, r) == 0.f) return -1;
    int idx = 0;
    for (int s = 1; s <= NS; ++s)
      for (int q = 0; q < s; ++q) {
        if (wgt(s, q) == 0.f || !first(s, q)) continue;
        if (q == r && cabs(wgt(s, q)) == cabs(wgt(s1, r))) return idx;
        ++idx;
      }
    return -1;
  }
  static constexpr int count() {
    int n = 0;
    for (int s = 1; s <= NS; ++s)
      for (int q = 0; q < s; ++q)
        if (wgt(s, q) != 0.f && first(s, q)) ++n;
    return n;
  }
  static constexpr int first_use(int r) {  // the first stage value (or the result) stage r's derivative enters
    for (int s1 = r + 1; s1 <= NS; ++s1)
      if (wgt(s1, r) != 0.f) return s1;
    return -1;
  }
};
template <int SOLVER>
struct RkChainTab : RkChain<SOLVER> {
  using C = RkChain<SOLVER>;
  static constexpr int NS = C::NS;
  static constexpr int NCOEF = C::count();
  static constexpr int NCW = NCOEF <= 1 ? 1 : (NCOEF <= 2 ? 2 : (NCOEF <= 4 ? 4 : 8));  // entries per step, padded
  static constexpr int LO = NCW < NS ? NCW : NS;  // entries kept in the first table (NS floats per step), rest in the second
  static constexpr int HI = NCW - LO;
  // table entries of one step from hgr[r] = h gr_r
  __device__ __forceinline__ static void fill(const float* hgr, float* E) {
    VIHDS_UNROLL for (int e = 0; e < NCW; ++e) E[e] = 0.f;
    static_for<1, NS + 1>([&](auto S1) {
      static_for<0, decltype(S1)::value>([&](auto Q) {
        constexpr int s1 = decltype(S1)::value, q = decltype(Q)::value;
        if constexpr (C::wgt(s1, q) != 0.f && C::first(s1, q)) {
          constexpr int e = C::index(s1, q);
          constexpr float w = C::cabs(C::wgt(s1, q));
          E[e] = w * hgr[q];
        }
      });
    });
  }
  // h gr_r back from the table
  template <int r>
  __device__ __forceinline__ static float hgr(const float* E) {
    constexpr int s1 = C::first_use(r);
    constexpr int e = C::index(s1, r);
    constexpr float inv = 1.f / C::cabs(C::wgt(s1, r));
    return E[e] * inv;
  }
  // one step: stage values us[s], returns u'
  __device__ __forceinline__ static float step(const float* E, float u, float* us) {
    float acc[NS + 1];
    VIHDS_UNROLL for (int s = 0; s <= NS; ++s) acc[s] = u;
    static_for<0, NS>([&](auto S) {
      constexpr int sidx = decltype(S)::value;
      const float v = acc[sidx];
      us[sidx] = v;
      const float p = fmaf(-v, v, v);
      static_for<sidx + 1, NS + 1>([&](auto S1) {
        constexpr int s1 = decltype(S1)::value;
        if constexpr (C::wgt(s1, sidx) != 0.f) {
          constexpr int e = C::index(s1, sidx);
          constexpr bool pos = C::wgt(s1, sidx) > 0.f;
          acc[s1] = pos ? fmaf(E[e], p, acc[s1]) : fmaf(-E[e], p, acc[s1]);
        }
      });
    });
    return acc[NS];
  }
  // the same step with its derivative: returns u' = Phi(u), d = dPhi/du, dus[s] = d us[s] / du
  __device__ __forceinline__ static float step_d(const float* E, float u, float* us, float* dus, float& d) {
    float acc[NS + 1], dac[NS + 1];
    VIHDS_UNROLL for (int s = 0; s <= NS; ++s) { acc[s] = u; dac[s] = 1.f; }
    static_for<0, NS>([&](auto S) {
      constexpr int sidx = decltype(S)::value;
      const float v = acc[sidx], dv = dac[sidx];
      us[sidx] = v;
      dus[sidx] = dv;
      const float p = fmaf(-v, v, v);
      const float dp = fmaf(-2.f * v, dv, dv);
      static_for<sidx + 1, NS + 1>([&](auto S1) {
        constexpr int s1 = decltype(S1)::value;
        if constexpr (C::wgt(s1, sidx) != 0.f) {
          constexpr int e = C::index(s1, sidx);
          constexpr bool pos = C::wgt(s1, sidx) > 0.f;
          acc[s1] = pos ? fmaf(E[e], p, acc[s1]) : fmaf(-E[e], p, acc[s1]);
          dac[s1] = pos ? fmaf(E[e], dp, dac[s1]) : fmaf(-E[e], dp, dac[s1]);
        }
      });
    });
    d = dac[NS];
    return acc[NS];
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
// The lane moves this kernel needs across the 16-lane rows, WITHOUT the LDS crossbar (round 5: each ds_bpermute was an LDS
// round trip -- ~100 cycles before its s_waitcnt lets the wavefront go on -- 42 of them per block; tests/micro/dpp_moves.hip
// checks every form against ds_bpermute).  All of them assume what holds at every call site: all 64 lanes active.
//   lane - 1 / lane + 1 of the wavefront (the trajectory's first / last lane overrides what crossed from the other half)
__device__ __forceinline__ float lane_below(float v) { return dpp_mov<0x138>(0.f, v); }  // wave_shr:1
__device__ __forceinline__ float lane_above(float v) { return dpp_mov<0x130>(0.f, v); }  // wave_shl:1
//   rows 1 and 3: lane 15 of the row below (row_bcast:15 into rows 1, 3; rows 0, 2 keep `keep`)
__device__ __forceinline__ float last_of_lower_row(float keep, float v) { return dpp_mov<0x142, 0xa>(keep, v); }
//   rows 0 and 2: lane 0 of the row above (lanes 16 / 48), by two scalar reads
__device__ __forceinline__ float first_of_upper_row(float v, int lane) {
  const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return lane < 32 ? s0 : s1;
}
//   v + v[lane ^ 16] on gfx950's row swap (through asm: the builtin's two results came back as one register)
__device__ __forceinline__ float add_other_row(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
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
  p.a = last_of_lower_row(1.f, m.a);
  p.b = last_of_lower_row(0.f, m.b);
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
  p.a = first_of_upper_row(m.a, lane);
  p.b = first_of_upper_row(m.b, lane);
  if ((lane & 31) < 16) m = after(m, p);
  return m;
}
// sum over the 32 lanes of a trajectory, result in all of them
__device__ __forceinline__ float sum32(float v, int lane) {
  v += dpp_all<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_all<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_all<0x141>(v);  // row_half_mirror
  v += dpp_all<0x140>(v);  // row_mirror
  return add_other_row(v);
}
// no code motion of LDS accesses across this point (the lanes of a wavefront exchange data through LDS; the hardware
// executes a wavefront's LDS instructions in order, so no s_barrier is needed)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#ifndef VIHDS_SCAN_TPB
#define VIHDS_SCAN_TPB 8
#endif
constexpr int DR_SCAN_TPB = VIHDS_SCAN_TPB;  // trajectories per block: 2 per wavefront, 32 lanes each
constexpr int DR_SCAN_THREADS = 32 * DR_SCAN_TPB;
static_assert(DR_SCAN_THREADS % 64 == 0 && DR_SCAN_THREADS >= 128, "two trajectories per wavefront, at least two wavefronts");
// Block barrier for data exchanged through LDS: waits for this wavefront's LDS operations only.  (__syncthreads() also
// waits for every outstanding global store and returning atomic -- a full memory round trip on the critical path of the
// block at each of its barriers; nothing in this kernel passes data between wavefronts through global memory.)
__device__ __forceinline__ void block_sync_lds() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); vmcnt / expcnt untouched
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
constexpr int DR_SCAN_NACC = 32;       // accumulators summed over the time axis in the epilogue (29 used)
// row stride of the epilogue's reduction buffer [NACC][threads]: lane l reads row l, sixteen bytes at a time -- at a stride of
// exactly `threads` floats (1 KB) all of a wavefront's lanes met in the same four banks (PMC: 3.3 M conflict cycles per
// launch, most of them here); four floats of padding move every lane to the next four banks
constexpr int DR_SCAN_RED_STRIDE = DR_SCAN_THREADS + 4;

// LDS per block (floats).  Per lane and step, as vectors: VG[NS] (sigmoid, then gamma), VU[NS] (x / K at the stages),
// VB[NS] (gamma adjoints), VQ[4] (observations, then log-likelihood injections), VY[4] + VZ[2] (states at the step's grid
// point), VA[4] (yfp / cfp maps, later reverse-scan multipliers and injections); then the time grid and x(T-1)/K per
// trajectory.  (The epilogue's reduction buffer overlays the per-step area.)
template <int SOLVER>
__host__ __device__ inline size_t dr_scan_lds_floats(int items) {
  const size_t steps = (size_t)(3 * Rk<SOLVER>::NS + 14) * items * DR_SCAN_THREADS;
  const size_t red = (size_t)DR_SCAN_NACC * DR_SCAN_RED_STRIDE + DR_SCAN_TPB * DR_SCAN_NACC + DR_SCAN_TPB * 64;
  return (steps > red ? steps : red) + (size_t)(32 * items + 4) + DR_SCAN_TPB;
}
#define VIHDS_ROLLED _Pragma("clang loop unroll(disable)")
// the loops over a lane's ITEMS steps: rolled unless VIHDS_ITEMBIT_<n> is 1 (three independent
// iterations side by side hide each other's latencies, at the price of registers: the mask is what fits 256 of them)
#ifndef VIHDS_ITEMBIT_0
#define VIHDS_ITEMBIT_0 1
#endif
#ifndef VIHDS_ITEMBIT_1
#define VIHDS_ITEMBIT_1 1
#endif
#ifndef VIHDS_ITEMBIT_2
#define VIHDS_ITEMBIT_2 1
#endif
#ifndef VIHDS_ITEMBIT_3
#define VIHDS_ITEMBIT_3 1
#endif
#ifndef VIHDS_ITEMBIT_4
#define VIHDS_ITEMBIT_4 0
#endif
#ifndef VIHDS_ITEMBIT_5
#define VIHDS_ITEMBIT_5 1
#endif
#ifndef VIHDS_ITEMBIT_6
#define VIHDS_ITEMBIT_6 0
#endif
#ifndef VIHDS_ITEMBIT_7
#define VIHDS_ITEMBIT_7 1
#endif
#ifndef VIHDS_ITEMBIT_8
#define VIHDS_ITEMBIT_8 1
#endif
#define VIHDS_ITEMLOOP_0 _Pragma("clang loop unroll(disable)")
#define VIHDS_ITEMLOOP_1 _Pragma("clang loop unroll(full)")
#define VIHDS_ITEMLOOP_SEL(B) VIHDS_ITEMLOOP_##B
#define VIHDS_ITEMLOOP_PICK(B) VIHDS_ITEMLOOP_SEL(B)
#define VIHDS_ITEMLOOP(N) VIHDS_ITEMLOOP_PICK(VIHDS_ITEMBIT_##N)
// profiling aid: kernel_variant = 3 | (phase << 8) makes the kernel return after that phase (tests/probe/scan_phases.py)
#ifdef VIHDS_SCAN_STAMPS
// profiling build (tests/probe/scan_stamps.py): every wavefront writes the 100 MHz wall clock at each phase boundary
static __device__ unsigned long long* vihds_scan_stamp_buf = nullptr;  // [blocks][waves per block][16]
#define VIHDS_SCAN_STOP(PH)                                                                                         \
  if (vihds_scan_stamp_buf && lane == 0)                                                                            \
    vihds_scan_stamp_buf[((size_t)blockIdx.x * (DR_SCAN_THREADS / 64) + wave) * 16 + (PH)] = wall_clock64();
#else
#define VIHDS_SCAN_STOP(PH) if ((a.kernel_variant >> 8) == (PH)) return;
#endif

template <int W>
__device__ __forceinline__ void ldv(const float* p, float* o) {
  if (W == 4) { const float4 v = *reinterpret_cast<const float4*>(p); o[0] = v.x; o[W > 1 ? 1 : 0] = v.y; o[W > 2 ? 2 : 0] = v.z; o[W > 3 ? 3 : 0] = v.w; }
  else if (W == 2) { const float2 v = *reinterpret_cast<const float2*>(p); o[0] = v.x; o[W > 1 ? 1 : 0] = v.y; }
  else o[0] = p[0];
}
template <int W>
__device__ __forceinline__ void stv(float* p, const float* o) {
  if (W == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[W > 1 ? 1 : 0], o[W > 2 ? 2 : 0], o[W > 3 ? 3 : 0]);
  else if (W == 2) *reinterpret_cast<float2*>(p) = make_float2(o[0], o[W > 1 ? 1 : 0]);
  else p[0] = o[0];
}

// LDS floats the conditioner's staged inputs need (relevance masks, the block's device one-hot rows, default flags:
// they sit in the VG area, which holds at least items * DR_SCAN_THREADS floats and is dead until the gamma pass)
__host__ __device__ inline size_t dr_scan_theta_floats(int E, int D) {
  return (size_t)E * D + (size_t)DR_SCAN_TPB * D + (size_t)E;
}

template <int VERSION, int SOLVER, int ITEMS, bool THETA>
__device__ __forceinline__ void dr_scan_train_body(const OdeArgs& a, float* lds, int nb_max, const ThetaStageArgs& t) {
  using M = DrConstant<VERSION>;
  using D = DrLanes<VERSION>;
  using R = Rk<SOLVER>;
  constexpr int NS = R::NS;
  constexpr int NT = DR_SCAN_THREADS;
  const int tid = threadIdx.x, lane = tid & 63, l = lane & 31, tib = tid >> 5, wave = tid >> 6;
  const int i0 = blockIdx.x * DR_SCAN_TPB + tib;
  const bool live = i0 < a.n;
  const int i = live ? i0 : a.n - 1;
  const int b = i / a.S;
  const int K = a.T - 1;
  const size_t n = a.n;
  // per-lane vector fields: field of width W, step m of thread t at base + (m * NT + t) * W
  constexpr int O_G = 0, O_U = O_G + NS * ITEMS * NT, O_B = O_U + NS * ITEMS * NT, O_Q = O_B + NS * ITEMS * NT,
                O_Y = O_Q + 4 * ITEMS * NT, O_Z = O_Y + 4 * ITEMS * NT, O_A = O_Z + 2 * ITEMS * NT,
                O_STEPS = O_A + 4 * ITEMS * NT, O_RED = DR_SCAN_NACC * DR_SCAN_RED_STRIDE + DR_SCAN_TPB * DR_SCAN_NACC + DR_SCAN_TPB * 64,
                O_T = O_STEPS > O_RED ? O_STEPS : O_RED, O_UK = O_T + 32 * ITEMS + 4, O_C = O_UK + DR_SCAN_TPB;
  auto VG = [&](int m) { return lds + O_G + (m * NT + tid) * NS; };
  // (the stage values of x are written by the chain wavefront, for all trajectories of the block at once: a trajectory's
  // 32 lane slots are rotated by its index, so that those eight stores fall into different LDS banks)
  auto VU = [&](int m) { return lds + O_U + (m * NT + tib * 32 + ((l + tib) & 31)) * NS; };
  auto VB = [&](int m) { return lds + O_B + (m * NT + tid) * NS; };
  auto VQ = [&](int m) { return lds + O_Q + (m * NT + tid) * 4; };
  auto VY = [&](int m) { return lds + O_Y + (m * NT + tid) * 4; };
  auto VZ = [&](int m) { return lds + O_Z + (m * NT + tid) * 2; };
  auto VA = [&](int m) { return lds + O_A + (m * NT + tid) * 4; };
  float* tT = lds + O_T;    // time grid [T]
  float* uK = lds + O_UK;   // x(T-1) / K per trajectory of the block
  enum { RFP, WW, LUXR, LASR, YFP, CFP, NSP };
  // theta rows of this trajectory in LDS (written once, by its 32 lanes; every lane then picks what it needs from
  // there).  Lives in the VY area of this trajectory's first lanes, which nobody writes before the parameter stage is over.
  // [0,64): theta rows; [64,66): fR, fS; [66,90): the Hill power terms (base, exponent, power) of lanes 0..7.
  // (90 floats per trajectory in use: the two trajectories of a wavefront stay inside that wavefront's own 256-float slot.)
  // (the second trajectory of a wavefront 100 floats behind the first, not 128: both halves of the wavefront read the same
  // slot at the same time, and at a distance of 512 bytes the two addresses fell into one bank -- every th() two passes)
  auto par_of = [&](int t) { return lds + O_Y + (t >> 1) * 256 + (t & 1) * 100; };
  float* par = par_of(tib);
  // (from the end of the first stage on `par` is in SLOT order -- see the permutation there: par[slot] is a constant
  // offset, neighbouring slots are read together, and no later phase waits for a scalar load of a.slot_row from the
  // kernel-argument segment: those misses were ~0.5 us each, two or three in a row ahead of the Hill stage)
  auto th = [&](int slot) { return par[slot]; };

#ifdef VIHDS_SCAN_STAMPS
  VIHDS_SCAN_STOP(0)
  if (vihds_scan_stamp_buf && lane == 0) {
    unsigned long long* sb = vihds_scan_stamp_buf + ((size_t)blockIdx.x * (DR_SCAN_THREADS / 64) + wave) * 16;
  }
#endif
  // ---- 0. Every global load of the block's first stage is requested HERE, as one batch, before anything waits for any
  //         of them: the stage is a chain of memory round trips (about 1 us each), not arithmetic, and a `lds[k] =
  //         global[k]` loop or a load under a branch makes the wavefront sit out a whole round trip by itself
  //         (s_waitcnt vmcnt(0) inside the loop / at the join).  Order: the q_rows indirection first -- q_mu / q_prec behind
  //         it are the only dependent pair --, then the time grid, the observations, the treatments, the prior's
  //         constants and the conditioner's inputs; the first pass of every staging loop is an unconditional load from a
  //         clamped index whose value is dropped by the lanes that are past the end.
  const int k0 = l * ITEMS;
  const float* ob = a.obs + (size_t)b * 4 * a.T;
  constexpr int TW = DR_SCAN_THREADS / 64 >= 3 ? 2 : DR_SCAN_THREADS / 64 - 1;  // the wavefront that takes the tickets
  int q_raw[2][2] = {{0, 0}, {0, 0}}, pqc[2] = {0, 0};
  if (THETA) {
    const int* qr = t.q_rows ? t.q_rows : t.kind;  // (no table: rows in parameter order; the loads stay in bounds)
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      pqc[q] = min(l + 32 * q, t.P - 1);
      q_raw[q][0] = qr[t.q_rows ? pqc[q] : 0];
      q_raw[q][1] = qr[t.q_rows ? t.P + pqc[q] : 0];
    }
  }
  const float t_mine = a.times[min(tid, a.T - 1)];
  float o_in[ITEMS][4], obK[4];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    const int kc = min(k0 + m, K - 1);
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) o_in[m][j] = ob[j * a.T + kc];
  }
  VIHDS_UNROLL for (int j = 0; j < 4; ++j) obK[j] = ob[j * a.T + K];
  // the treatments of this lane's trajectory and the conditioner generator's state: fetched here, used after the first barrier
  float cond_raw[1][2];
  {
    const int bb = min(blockIdx.x * DR_SCAN_TPB + tib, a.n - 1) / a.S;
    cond_raw[0][0] = a.cond[bb * a.C + 0];
    cond_raw[0][1] = a.cond[bb * a.C + 1];
  }
  // the theta rows this lane writes the gradients of in the epilogue (slots l and l + 32)
  int out_row[2];
  VIHDS_UNROLL for (int q = 0; q < 2; ++q) out_row[q] = a.slot_row[min(l + 32 * q, M::NSLOT + 3)];
  unsigned int ck0 = 0u, ck1 = 0u, cstep = 0u;
  if (THETA && t.crng) { ck0 = t.crng[0]; ck1 = t.crng[1]; cstep = t.crng[2]; }
  RngTickets tk = {0u, 0u};
  auto stage_grid_and_observations = [&]() {
    if (tid < a.T) tT[tid] = t_mine;
    VIHDS_ROLLED for (int k = tid + NT; k < a.T; k += NT) tT[k] = a.times[k];
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      stv<4>(VQ(m), o_in[m]);
    }
  };
  if (!THETA) {
    stage_grid_and_observations();
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const int slot = l + 32 * q;
      if (slot < M::NSLOT + 4) par[slot] = a.theta[(size_t)out_row[q] * n + i];  // (slot order; out_row: this lane's slots' rows)
    }
    block_sync_lds();
  } else {
    // ---- sampling stage of the decoder step (vihds_theta_ode_logp_grad): theta = clip(sample(q, u)) with log q / log p
    //      (the arithmetic of theta_fwd_lds_kernel; lane l of a trajectory owns parameters l and l + 32).  Every lane
    //      fetches the distribution constants of its own (data row, parameter) pairs itself: no per-block table, no
    //      barrier before the draws are used, and a wavefront never waits for another one in this stage.  theta / u /
    //      log_q / log_p go to global memory (the API's outputs) and theta to `par` for this kernel's own use.  The
    //      device-conditioner rows are produced later, by the last wavefront, while wavefront 0 walks the x chains.
    constexpr float LOG2PI = 1.8378770664093453f;
    const int P = t.P, B = a.B;
    float c_mu[2], c_pr[2], c_pp[2], c_lo[2], c_hi[2], c_pmu[2], uu[2] = {0.f, 0.f};
    int c_kd[2];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const int pq = pqc[q];
      c_kd[q] = t.kind[pq];
      c_pp[q] = t.p_prec[pq];
      c_lo[q] = t.clip_lo[pq];
      c_hi[q] = t.clip_hi[pq];
      c_pmu[q] = t.p_mu[pq];
    }
    if (!t.rng) {
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) uu[q] = t.u[(size_t)i * P + pqc[q]];
    }
    // the conditioner's inputs for this block (relevance masks, default flags, the device one-hot rows its tiling
    // selects: row (b S_total + s) mod B for sample (b, s)) -> LDS, for the wavefront that evaluates it later
    float* t_rel = lds + O_C;                      // [E][D]   (an area of its own behind the block's other fields:
                                                   //  read by every wavefront, at its own pace, after barrier A)
    float* t_dev = t_rel + t.E * a.D;              // [TPB][D]
    float* t_dfl = t_dev + DR_SCAN_TPB * a.D;      // [E]
    const bool cnd = t.E > 0;
    const int Dd = max(a.D, 1), ed = max(t.E * a.D, 1), td = DR_SCAN_TPB * Dd;
    auto dev_row = [&](int e) {  // element e of the block's [TPB][D] one-hot rows: its address in dev1hot
      const int tt = e / Dd, d = e - tt * Dd;
      const int it = min(blockIdx.x * DR_SCAN_TPB + tt, a.n - 1), bt = it / a.S;
      // (b S_total + s < B S_total = the number of samples of the whole job: it fits 32 bits, as a.n does)
      const unsigned int rr = ((unsigned int)bt * (unsigned int)t.S_total + (unsigned int)(t.s_off + (it - bt * a.S))) % (unsigned int)B;
      return (int)(__umul24(rr, (unsigned int)Dd) + (unsigned int)d);  // (24-bit factors: checked by the launcher)
    };
    // (no conditioner: the same loads from addresses that exist, so that nothing here sits under a branch)
    const float* relp = cnd ? t.rel : a.times;
    const float* devp = cnd ? a.dev1hot : a.times;
    const int* dflp = cnd ? t.is_default : t.kind;
    // Fast path (E rows x D devices fit a trajectory's 32 lanes, D padded to a power of two -- every shipped model): lane
    // (e, d) of the trajectory holds ITS OWN operands of row e's dot product, loaded here with everything else; the weights
    // are generated by the lanes that use them, the sum over d is four DPP adds, and nothing of the conditioner goes
    // through LDS (the general path below stages the inputs for a loop per row).
    int clog = 0;
    while ((1 << clog) < Dd) ++clog;
    const bool cfast = cnd && (t.E << clog) <= 32;
    const int ce = l >> clog, cd = l & ((1 << clog) - 1);
    const bool cmine = cfast && ce < t.E && cd < a.D;
    const unsigned int rr_own = ((unsigned int)b * (unsigned int)t.S_total + (unsigned int)(t.s_off + (i - b * a.S))) % (unsigned int)B;
    const float cf_rel = relp[cmine ? ce * a.D + cd : 0];
    const float cf_dev = devp[cmine ? (int)(__umul24(rr_own, (unsigned int)Dd) + (unsigned int)cd) : 0];
    const int cf_dfl = dflp[cmine ? ce : 0];
    const float cf_z = (cmine && !t.crng) ? t.z[ce * a.D + cd] : 0.f;
    const bool cslow = cnd && !cfast;
    float pre_rel = 0.f, pre_dev = 0.f;
    int pre_dfl = 0;
    if (cslow) {  // (uniform: the block-wide staging of the general path, with its index arithmetic -- two divisions and a modulo)
      pre_rel = relp[min(tid, ed - 1)];
      pre_dev = devp[dev_row(min(tid, td - 1))];
      pre_dfl = dflp[min(tid, t.E - 1)];
    }
    // (the dependent pair: the rows of mu_p / log-precision_p in the encoder's table)
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const int rm = t.q_rows ? q_raw[q][0] : pqc[q], rp = t.q_rows ? q_raw[q][1] : pqc[q];
      c_mu[q] = t.q_mu[rm * B + b];
      c_pr[q] = t.q_prec[rp * B + b];
    }
    const bool one_call = t.rng && t.crng && cfast && (t.E << clog) + ((P + 3) >> 2) <= 32;
    float cw_z = 0.f;
    if (t.rng) {
      // one generator call per wavefront: lane l < ceil(P / 4) draws the four normals of parameter block l (counter =
      // global sample index, block, step: vihds_rng.hpp), the trajectory's lanes pick theirs up from LDS
      const unsigned int rk0 = t.rng[0], rk1 = t.rng[1], step = t.rng[2];
      const unsigned int gidx = (unsigned int)(b * t.S_total + t.s_off + (i - b * a.S));
      float* zb = lds + O_Z + tib * 64;  // (the VZ area: 64 floats per trajectory, unused until the log-likelihood)
      float z4[4];
      // (one call for both generators where the lanes allow it: the conditioner's weights are drawn by lanes (e, d) at the
      // bottom of the trajectory's 32, the parameter blocks by lanes 31, 30, ... -- a second call would cost every lane of
      // the wavefront its ~170 instructions again)
      const bool cw_lane = one_call && cmine;
      const int ub = one_call ? 31 - l : l;
      philox_normal4(cw_lane ? (unsigned int)(ce * a.D + cd) : gidx, cw_lane ? 0xC04Du : (unsigned int)ub,
                     cw_lane ? cstep : step, 0u, cw_lane ? ck0 : rk0, cw_lane ? ck1 : rk1, z4);
      cw_z = z4[0];
      if (4 * ub < P && !cw_lane) stv<4>(zb + 4 * ub, z4);
      wave_sync();
      VIHDS_UNROLL for (int q = 0; q < 2; ++q)
        if (l + 32 * q < P) uu[q] = zb[l + 32 * q];
    }
#ifdef VIHDS_SCAN_STAMPS
    VIHDS_SCAN_STOP(14)
#endif
    stage_grid_and_observations();
    float cf_val = 0.f;
    if (cfast) {
      // device conditioner (ode.py:43-58, with its .repeat tiling), row e of this trajectory: relu(sum_d w[e,d] dev[d]
      // rel[e,d]) on top of the default flag; w = mean + std z, z from the conditioner's generator (one call per weight,
      // counter e D + d: the numbers of the general path) or from the caller's draw
      float zz = cf_z;
      if (one_call) zz = cw_z;
      else if (t.crng) zz = philox_normal((unsigned int)(ce * a.D + cd), 0xC04Du, cstep, 0u, ck0, ck1, 0);
      float sm = cmine ? (t.w_mean + t.w_std * zz) * (cf_dev * cf_rel) : 0.f;
      if (clog >= 1) sm += dpp_all<0xB1>(sm);   // quad_perm [1,0,3,2]
      if (clog >= 2) sm += dpp_all<0x4E>(sm);   // quad_perm [2,3,0,1]
      if (clog >= 3) sm += dpp_all<0x141>(sm);  // row_half_mirror
      if (clog >= 4) sm += dpp_all<0x140>(sm);  // row_mirror
      if (clog >= 5) sm = add_other_row(sm);
      cf_val = (cf_dfl ? 1.f : 0.f) + fmaxf(sm, 0.f);
      if (cmine && cd == 0) par[t.cond_row0 + ce] = cf_val;  // (its copy in theta: with the stage's other stores, below)
    }
    if (cslow) {
      asm volatile("" : "+v"(pre_dfl));  // (keeps the flag's comparison here, behind the requests above)
      if (tid < ed) t_rel[tid] = pre_rel;
      if (tid < td) t_dev[tid] = pre_dev;
      if (tid < t.E) t_dfl[tid] = pre_dfl ? 1.f : 0.f;
      VIHDS_ROLLED for (int e = tid + NT; e < ed; e += NT) t_rel[e] = t.rel[e];
      VIHDS_ROLLED for (int e = tid + NT; e < td; e += NT) t_dev[e] = a.dev1hot[dev_row(e)];
      VIHDS_ROLLED for (int e = tid + NT; e < t.E; e += NT) t_dfl[e] = t.is_default[e] ? 1.f : 0.f;
    }
    float lq = 0.f, lp = 0.f, xo[2] = {0.f, 0.f};
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const bool cst = c_kd[q] == KIND_CONSTANT, ln = c_kd[q] == KIND_LOGNORMAL;
      // (v_exp_f32 / v_log_f32 / v_rsq_f32, 1 ulp: the libm sequences of these six calls were ~300 instructions of a
      // stage every wavefront of the block goes through before the x chains can start)
      const float prc = cst ? 1.f : (t.prec_is_log ? __expf(c_pr[q]) : c_pr[q]);
      const float sigma = __builtin_amdgcn_rsqf(prc);
      const float cq = -LOG2PI + 0.34657359027997264f * __builtin_amdgcn_logf(prc + 1e-12f), cp = -LOG2PI + 0.34657359027997264f * __builtin_amdgcn_logf(c_pp[q] + 1e-12f);
      const float mu = c_mu[q];
      const float zz = mu + sigma * uu[q];
      float x = ln ? __expf(zz) : zz;
      x = x < c_lo[q] ? c_lo[q] : (x > c_hi[q] ? c_hi[q] : x);
      const float v = ln ? 0.6931471805599453f * __builtin_amdgcn_logf(x + 1e-12f) : x;
      const float jac = ln ? v : 0.f;
      const float dq = mu - v, dp = c_pmu[q] - v;
      const float tq = cq - 0.5f * prc * dq * dq - jac;
      const float tp = cp - 0.5f * c_pp[q] * dp * dp - jac;
      const bool on = l + 32 * q < P && !cst;
      lq += on ? tq : 0.f;
      lp += on ? tp : 0.f;
      xo[q] = cst ? 0.f * uu[q] + mu : x;
      if (l + 32 * q < P) par[l + 32 * q] = xo[q];
    }
    lq = sum32(lq, lane);
    lp = sum32(lp, lane);
    // Everything this stage asked memory for has been consumed: from here on the wavefront only stores (a wait for a
    // load issued before a store would also wait for the store's acknowledgement -- loads and stores share vmcnt).
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const int pq = l + 32 * q;
      if (live && pq < P) {
        if (t.rng) t.u[(size_t)i * P + pq] = uu[q];
        t.theta[(size_t)pq * n + i] = xo[q];
      }
    }
    if (live && l == 0) {
      if (t.log_q) t.log_q[i] = lq;
      if (t.log_p) t.log_p[i] = lp;
    }
    if (live && cmine && cd == 0) t.theta[(size_t)(t.cond_row0 + ce) * n + i] = cf_val;
    wave_sync();  // this trajectory's theta rows are in `par` (its own lanes wrote them)
    {
      // rows -> slots, in place: lane l picks up the rows of its slots l and l + 32 (out_row, fetched per lane at kernel
      // entry with everything else), then all write.  Rows no slot names are not needed again.
      float pv[2];
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) pv[q] = par[out_row[q]];
      wave_sync();
      VIHDS_UNROLL for (int q = 0; q < 2; ++q)
        if (l + 32 * q < M::NSLOT + 4) par[l + 32 * q] = pv[q];
      wave_sync();
    }
#ifdef VIHDS_SCAN_STAMPS
    VIHDS_SCAN_STOP(15)
#endif
  }
  const float r = clampf(th(M::S_r), 0.f, 4.f), tlag = th(M::S_tlag);
  const float h0 = tT[1] - tT[0];
  // step m of this lane: grid index (clamped for the padding steps beyond K), validity, step size
  struct Item {
    int kc;
    bool valid;
    float h, t0, dt;
  };
  auto item = [&](int m) {
    Item it;
    it.valid = k0 + m < K;
    it.kc = it.valid ? k0 + m : K - 1;
    it.t0 = tT[it.kc];
    it.dt = tT[it.kc + 1] - it.t0;
    it.h = R::FIXED_H ? h0 : it.dt;
    return it;
  };
  auto stage_sigmoid = [&](const Item& it, int s) { return sigmoid_f(4.f * (fmaf(R::c(s), it.dt, it.t0) - tlag)); };
  // ---- 1. the x chain's coefficients for this lane's steps (RkChainTab: |w| h_k r sigmoid(4 (t_{k,s} - tlag)), state
  //         independent), in registers. ---------------------------------------------------------------------------------
  using CT = RkChainTab<SOLVER>;
  float Eown[ITEMS][CT::NCW];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
    const Item it = item(m);
    float hgr[NS];
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) hgr[s] = (it.h * r) * stage_sigmoid(it, s);
    CT::fill(hgr, Eown[m]);
  }
  block_sync_lds();  // every trajectory's parameters and the conditioner's staged inputs are in place; VZ's draws are consumed
  VIHDS_SCAN_STOP(1)

  // ---- 2. the x chain u_{k+1} = Phi_k(u_k) (u = x / K), IN PARALLEL over the trajectory's 32 lanes.  The chain is the one
  //         nonlinear recurrence of the model; walked step by step (85 steps of 20 dependent instructions in one wavefront,
  //         the other three waiting) it was 6 us of a block's 30.  Here it is solved as what it is, a system of equations in
  //         the values g_l at the lanes' first steps:  g_{l+1} = F_l(g_l),  F_l = this lane's ITEMS steps, g_0 given.
  //         Newton on that system: every lane walks its own steps from its current g_l, with the derivative (F_l, F_l');
  //         the linearised recurrence  g_{l+1} = F_l + F_l' (g_l^new - g_l)  is a scan of affine maps over the lanes.  The
  //         first guess is the closed form of the continuous equation du/dt = gr(t) u (1 - u), gr = r sigmoid(4 (t -
  //         tlag)):  1/u - 1 = (1/u0 - 1) exp(-(G(t) - G(t0))),  G = r/4 softplus(4 (t - tlag)), off by the scheme's own
  //         truncation error; convergence is quadratic, and since g_0 is exact, iteration i leaves the first i lanes exact
  //         whatever the guess (even one that overflows the lanes above): 32 iterations always suffice (typically two;
  //         tests/test_hip_parity.py::test_time_parallel_x_chain_at_extreme_growth_parameters).  The loop ends when an update moved nothing
  //         by more than 1e-6 of its value; the stage values kept are those of the walk that preceded that update, so
  //         every lane's steps are exact steps of the scheme and the values where two lanes meet agree to that 1e-6 --
  //         the rounding level of the sequential walk itself.
  float us_own[ITEMS][NS];
  VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m)
    VIHDS_UNROLL for (int s = 0; s < NS; ++s) us_own[m][s] = 0.5f;  // (padding steps beyond T-1 keep this harmless value)
  {
    const float u0 = th(M::SI + 0) * frcp(clampf(th(M::S_K), 0.f, 4.f));
    auto softplus = [](float z) { return fmaxf(z, 0.f) + 0.6931471805599453f * __builtin_amdgcn_logf(1.f + __expf(-fabsf(z))); };
    const float tl = tT[min(k0, K)];
    const float dG = 0.25f * r * (softplus(4.f * (tl - tlag)) - softplus(4.f * (tT[0] - tlag)));
    float g = l == 0 ? u0 : frcp(fmaf(frcp(u0) - 1.f, __expf(-dG), 1.f));
    float uend = g;
    float dus_own[ITEMS][NS];  // d us_own[m][s] / d g
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m)
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) dus_own[m][s] = 0.f;
    int walks = 32, first_order = 0;  // (telemetry only: a.newton_hist)
    for (int iter = 0; iter < 32; ++iter) {
      float u = g, da = 1.f;
      VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
        if (k0 + m < K) {
          float d, ds[NS];
          u = CT::step_d(Eown[m], u, us_own[m], ds, d);
          VIHDS_UNROLL for (int s = 0; s < NS; ++s) dus_own[m][s] = da * ds[s];
          da *= d;
        }
      }
      uend = u;
      Aff mp;
      mp.a = da;
      mp.b = fmaf(-da, g, u);
      const Aff inc = scan_up32(mp, lane);               // lane l: this lane's map after those of the lanes below it
      const float end_new = fmaf(inc.a, u0, inc.b);      // the new value at the first step of lane l + 1
      float gn = lane_below(end_new);  // (lane l - 1 of the trajectory; its first lane takes u0 below)
      gn = l == 0 ? u0 : gn;
      // A lane whose values are not finite has NOT settled: a poor guess far above the capacity can overflow a lane's walk
      // while the lanes below it are still converging -- maps only act upwards, so those keep converging, and each
      // iteration hands at least one more lane an exact first value.  (A chain that is non-finite itself, as the
      // step-by-step walk would find it, costs the full 32 iterations; non-finite parameters leave at once.)
      const float dg = gn - g;
      const bool settled = fabsf(dg) <= 1e-6f * fabsf(gn);
      if (__builtin_amdgcn_ballot_w64(!settled && (u0 - u0 == 0.f)) == 0ull) { walks = iter + 1; break; }
      // Close enough for the first-order term to finish the job (the usual case: the closed form is off by ~1e-4, the
      // neglected second-order term is the square of that): the stage values move along the derivatives this walk carried,
      // and the second walk is saved -- every lane's steps then hold to ~1e-8 instead of exactly, the level of the rounding
      // in the walk itself.
      const bool near = fabsf(dg) <= 2e-4f * fabsf(gn);
      if (__builtin_amdgcn_ballot_w64(!near) == 0ull) {
        VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m)
          VIHDS_UNROLL for (int s = 0; s < NS; ++s) us_own[m][s] = fmaf(dus_own[m][s], dg, us_own[m][s]);
        uend = fmaf(da, dg, uend);
        walks = iter + 1;
        first_order = 1;
        break;
      }
      g = gn;
    }
    if (a.newton_hist && lane == 0) {  // (uniform; a non-returning atomic: nothing waits for it)
      atomicAdd(&a.newton_hist[walks], 1u);
      if (first_order) atomicAdd(&a.newton_hist[33], 1u);
    }
    if (l == 31) uK[tib] = uend;  // (lanes beyond the last step hold the identity: the last lane's value is x(T-1) / K)
    VIHDS_UNROLL for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      float gm[NS];
      const float invh = frcp(it.h);
      static_for<0, NS>([&](auto S) {
        constexpr int sidx = decltype(S)::value;
        const float gr = CT::template hgr<sidx>(Eown[m]) * invh;
        gm[sidx] = fmaf(-gr, us_own[m][sidx], gr);  // gamma_s = gr_s (1 - u_s)
      });
      stv<NS>(VU(m), us_own[m]);
      stv<NS>(VG(m), gm);
    }
  }
#ifdef VIHDS_SCAN_STAMPS
  VIHDS_SCAN_STOP(11)
#endif

  // ---- Hill fractions (dr_constant.py:58-73) of this wavefront's two trajectories.  Lanes 0..7 of a half-wave hold the
  //      power terms (v1: (K6 c6)^n, (K12 c12)^n, (1 + K6 c6 + K12 c12)^n for LuxR and LasR; v2: four terms), as in
  //      DrLanes::hill.
  auto treatments = [&](float* c) {
    c[0] = clampf(expf(cond_raw[0][0]) - 1.f, 1e-12f, 1e6f);
    c[1] = clampf(expf(cond_raw[0][1]) - 1.f, 1e-12f, 1e6f);
  };
  {
    // Every thread of the block read the generators' step counters before the barrier above (scalar loads, complete at its
    // lgkmcnt(0)), so the block may take its tickets.  The holder of the last ticket advances the step at the very end of
    // the kernel (rng_advance; see dr_lane_theta_stage).
    if (THETA && tid == TW * 64) {
      if (t.rng) tk.u = atomicAdd(&t.rng[3], 1u);
      if (t.crng) tk.c = atomicAdd(&t.crng[3], 1u);
    }
    {
      float* pp = par;
      auto tht = [&](int slot) { return pp[slot]; };
      float cc[2];
      treatments(cc);
      const int j = l & 7;
      const float nR = clampf(tht(M::S_nR), 0.5f, 3.f), nS = clampf(tht(M::S_nS), 0.5f, 3.f);
      float base, ex, fR_, fS_;
      if (VERSION == 1) {
        const bool isR = j < 3;
        // (both rows read, then chosen: `tht(isR ? a : b)` made the lane-dependent slot index a per-lane GLOBAL load from
        // the kernel-argument segment -- a memory round trip of ~1 us in every wavefront's path)
        const float h0 = tht(M::S_H0), h1 = tht(M::S_H1), h2 = tht(M::S_H2), h3 = tht(M::S_H3);
        const float K6 = clampf(isR ? h0 : h2, 1e-12f, 1.f);
        const float K12 = clampf(isR ? h1 : h3, 1e-12f, 1.f);
        const float ta = K6 * cc[0], tb = K12 * cc[1];
        const int k = isR ? j : j - 3;
        base = j >= 6 ? 1.f : (k == 0 ? ta : (k == 1 ? tb : 1.f + ta + tb));
        ex = j >= 6 ? 1.f : (isR ? nR : nS);
      } else {
        const float eS6 = clampf(tht(M::S_H0), 1e-12f, 1.f), eR12 = clampf(tht(M::S_H1), 1e-12f, 1.f);
        base = j == 0 ? cc[0] : (j == 1 ? eR12 * cc[1] : (j == 2 ? eS6 * cc[0] : (j == 3 ? cc[1] : 1.f)));
        ex = j < 2 ? nR : (j < 4 ? nS : 1.f);
      }
      // base^ex as 2^(ex log2 base) on v_log_f32 / v_exp_f32 (base > 0 by its clamps; the terms that matter have a base
      // near 1, where the product's rounding is below 1e-6; libm's powf was ~200 instructions on every wavefront's path)
      const float pw = __builtin_amdgcn_exp2f(ex * __builtin_amdgcn_logf(base));
      if (VERSION == 1) {
        fR_ = (bcast8<0>(pw) + bcast8<1>(pw)) / bcast8<2>(pw);
        fS_ = (bcast8<3>(pw) + bcast8<4>(pw)) / bcast8<5>(pw);
      } else {
        fR_ = bcast8<0>(pw) + bcast8<1>(pw);
        fS_ = bcast8<2>(pw) + bcast8<3>(pw);
      }
      if (l < 8) {
        pp[66 + l] = base;
        pp[74 + l] = ex;
        pp[82 + l] = pw;
      }
      if (l == 0) {
        pp[64] = fR_;
        pp[65] = fS_;
      }
    }
    if (THETA && t.E > 0 && (t.E << [&] { int c = 0; while ((1 << c) < max(a.D, 1)) ++c; return c; }()) > 32) {
      // device conditioner, general path (the rows did not fit the trajectory's lanes: see the sampling stage) for THIS
      // wavefront's two trajectories.  Every wavefront
      // generates the E x D weights itself (one generator call per weight, the same counters: the same numbers) into its
      // own corner of LDS -- its trajectories' draw slots, which it has consumed itself -- and writes the rows (aR, aS) of
      // its own trajectories only: nothing in this stage crosses wavefronts, so no workgroup barrier follows it.
      float* t_cw = lds + O_Z + wave * 128;  // [E][D], E D <= 128 (checked by the launcher)
      const float* t_rel = lds + O_C;
      const float* t_dev = t_rel + t.E * a.D;
      const float* t_dfl = t_dev + DR_SCAN_TPB * a.D;
      const int ed = t.E * a.D;
      for (int e = lane; e < ed; e += 64) {
        float zz;
        if (t.crng) zz = philox_normal((unsigned int)e, 0xC04Du, cstep, 0u, ck0, ck1, 0);
        else zz = t.z[e];
        t_cw[e] = t.w_mean + t.w_std * zz;
      }
      wave_sync();
      if (lane < 2 * t.E) {  // (E <= 32)
        const int tl = lane >= t.E ? 1 : 0, e = lane - tl * t.E, tt = 2 * wave + tl;
        const int it0 = blockIdx.x * DR_SCAN_TPB + tt;
        const bool lv = it0 < a.n;
        const int it = lv ? it0 : a.n - 1;
        float cc = 0.f;
        for (int d = 0; d < a.D; ++d) cc += t_cw[e * a.D + d] * (t_dev[tt * a.D + d] * t_rel[e * a.D + d]);
        cc = fmaxf(cc, 0.f);
        const float val = t_dfl[e] + cc;
        if (lv) t.theta[(size_t)(t.cond_row0 + e) * n + it] = val;
        for (int sl = 0; sl < M::NSLOT + 4; ++sl)  // (`par` is in slot order by now: the slots that name this row)
          if (a.slot_row[sl] == t.cond_row0 + e) par_of(tt)[sl] = val;
      }
    }
    wave_sync();  // this wavefront's Hill terms and conditioner rows are in its trajectories' parameters
#ifdef VIHDS_SCAN_STAMPS
    VIHDS_SCAN_STOP(12)
#endif
  }
  VIHDS_SCAN_STOP(2)

  // ---- parameters of this trajectory (every lane of its 32 holds them) -----------------------------------------
  float c[2];
  treatments(c);
  const float Kc = clampf(th(M::S_K), 0.f, 4.f), invK = frcp(Kc);
  const float rc = th(M::S_rc);
  typename D::HillTerm H;
  H.base = par[66 + (l & 7)];
  H.n = par[74 + (l & 7)];
  H.pw = par[82 + (l & 7)];
  const float fR = par[64], fS = par[65];
  // where torch.clamp passes the gradient (bounds included), for the epilogue
  const float pass_r = clamp_pass(th(M::S_r), 0.f, 4.f), pass_K = clamp_pass(th(M::S_K), 0.f, 4.f);
  const float pass_h[6] = {clamp_pass(th(M::S_nR), 0.5f, 3.f), clamp_pass(th(M::S_nS), 0.5f, 3.f),
                           clamp_pass(th(M::S_H0), 1e-12f, 1.f), clamp_pass(th(M::S_H1), 1e-12f, 1.f),
                           VERSION == 1 ? clamp_pass(th(M::S_H0 + 2), 1e-12f, 1.f) : 0.f,
                           VERSION == 1 ? clamp_pass(th(M::S_H0 + 3), 1e-12f, 1.f) : 0.f};
  const float pass_d[5] = {clamp_pass(th(M::S_drfp), 1e-12f, 2.f), clamp_pass(th(M::S_dyfp), 1e-12f, 2.f),
                           clamp_pass(th(M::S_dcfp), 1e-12f, 2.f), clamp_pass(th(M::S_dR), 1e-12f, 5.f),
                           clamp_pass(th(M::S_dS), 1e-12f, 5.f)};
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
  y0[RFP] = th(M::SI + 1); y0[YFP] = th(M::SI + 2); y0[CFP] = th(M::SI + 3); y0[WW] = 0.f;
  y0[LUXR] = th(M::SI + 4); y0[LASR] = th(M::SI + 5);

  // species pairs for the packed step functions: (rfp, W), (luxR, lasR), (yfp, cfp)
  const v2f delta2[3] = {{delta[RFP], delta[WW]}, {delta[LUXR], delta[LASR]}, {delta[YFP], delta[CFP]}};
  const v2f F12[2] = {{F1[RFP], F1[WW]}, {F1[LUXR], F1[LASR]}};
  const v2f pe2 = {pe[0], pe[1]}, pc2 = {pc[0], pc[1]}, pcR2 = {pcR[0], pcR[1]}, pcS2 = {pcS[0], pcS[1]};
  auto rcp2 = [](v2f v) { return v2f{frcp(v.x), frcp(v.y)}; };
  const float xK = Kc * uK[tib];  // x at the last grid point
  VIHDS_SCAN_STOP(3)

  // ---- 3. level 1 (rfp, W, luxR, lasR): per-step affine maps composed over this lane's steps, scan over lanes ----------
  float ys[NSP];  // state at this lane's first grid point
  {
    v2f la[2] = {{1.f, 1.f}, {1.f, 1.f}}, lb[2] = {{0.f, 0.f}, {0.f, 0.f}};  // composed maps of the two pairs
    VIHDS_ITEMLOOP(0) for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      float gam[NS];
      ldv<NS>(VG(m), gam);
      VIHDS_UNROLL for (int jp = 0; jp < 2; ++jp) {
        v2f as[NS], Fs[NS], A, Bb;
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta2[jp]; Fs[s] = F12[jp]; }
        R::affine(it.h, as, Fs, A, Bb);
        if (!it.valid) { A = v2f{1.f, 1.f}; Bb = v2f{0.f, 0.f}; }
        lb[jp] = fm(A, lb[jp], Bb);  // st after lm
        la[jp] = A * la[jp];
      }
    }
    const Aff lm[4] = {{la[0].x, lb[0].x}, {la[0].y, lb[0].y}, {la[1].x, lb[1].x}, {la[1].y, lb[1].y}};
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      const Aff sc = scan_up32(lm[j], lane);
      const float end = fmaf(sc.a, y0[j], sc.b);
      const float prev = lane_below(end);
      ys[j] = l == 0 ? y0[j] : prev;
    }
  }
  VIHDS_SCAN_STOP(4)
  // ---- 4. the steps themselves for level 1 (states at this lane's grid points, stage values of luxR / lasR), promoters,
  //         affine maps of yfp / cfp ----------------------------------------------------------------------------------
  float yend[NSP];  // state after this lane's last valid step
  {
    v2f cur2[2] = {{ys[RFP], ys[WW]}, {ys[LUXR], ys[LASR]}};
    v2f la = {1.f, 1.f}, lb = {0.f, 0.f};
    VIHDS_ITEMLOOP(1) for (int m = 0; m < ITEMS; ++m) {
      const Item it = item(m);
      float gam[NS];
      ldv<NS>(VG(m), gam);
      {
        const float c4[4] = {cur2[0].x, cur2[0].y, cur2[1].x, cur2[1].y};
        stv<4>(VY(m), c4);
      }
      v2f YRS[NS], dummy[NS];  // stage values of (luxR, lasR)
      VIHDS_UNROLL for (int jp = 0; jp < 2; ++jp) {
        v2f as[NS], Fs[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta2[jp]; Fs[s] = F12[jp]; }
        const v2f nx = R::real(it.h, as, Fs, cur2[jp], jp == 1 ? YRS : dummy);
        cur2[jp] = it.valid ? nx : cur2[jp];
      }
      v2f as[NS], Fs[NS], A, Bb;
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const v2f kb = fm(pcS2, splat<v2f>(YRS[s].y * YRS[s].y), pcR2 * (YRS[s].x * YRS[s].x));
        const v2f t = kb * rcp2(1.f + kb);
        Fs[s] = pc2 * fm(1.f - pe2, t, pe2);
        as[s] = gam[s] + delta2[2];
      }
      R::affine(it.h, as, Fs, A, Bb);
      if (!it.valid) { A = v2f{1.f, 1.f}; Bb = v2f{0.f, 0.f}; }
      const float ab[4] = {A.x, A.y, Bb.x, Bb.y};
      lb = fm(A, lb, Bb);
      la = A * la;
      stv<4>(VA(m), ab);
    }
    const Aff lm[2] = {{la.x, lb.x}, {la.y, lb.y}};
    const float cur[4] = {cur2[0].x, cur2[0].y, cur2[1].x, cur2[1].y};
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) yend[j] = cur[j];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const Aff sc = scan_up32(lm[q], lane);
      const float end = fmaf(sc.a, y0[YFP + q], sc.b);
      const float prev = lane_below(end);
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
    // (v_log_f32 x ln 2: 1 ulp of log2; libm's logf was ~30 instructions each.  0.5 / prec once, not per grid point.)
    float hp[4];
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
      lc[j] = LOG2PI_F - 0.6931471805599453f * __builtin_amdgcn_logf(prec[j]);
      hp[j] = 0.5f * frcp(prec[j]);
    }
    auto point = [&](const float* obs_k, bool on, float x, float rfp, float yf, float cf, float w, float* qo) {
      const float xp[4] = {x, x * rfp, x * fmaf(a530, w, yf), x * fmaf(a480, w, cf)};
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) {
        const float e = xp[j] - obs_k[j];
        lp[j] += on ? -0.5f * fmaf(prec[j] * e, e, lc[j]) : 0.f;
        qo[j] = on ? -prec[j] * e : 0.f;
        precb[j] += on ? fmaf(-0.5f * e, e, hp[j]) : 0.f;
      }
      a530b += qo[2] * x * w;  // f530 = a530 W, f480 = a480 W: their amplitudes only enter here
      a480b += qo[3] * x * w;
    };
    float cz[2] = {ys[YFP], ys[CFP]};
    VIHDS_ITEMLOOP(2) for (int m = 0; m < ITEMS; ++m) {
      const bool valid = k0 + m < K;
      float qo[4], obk[4], y4[4], ab[4], us[NS];
      ldv<4>(VQ(m), obk);
      ldv<4>(VY(m), y4);
      ldv<4>(VA(m), ab);
      ldv<NS>(VU(m), us);
      stv<2>(VZ(m), cz);
      point(obk, valid, Kc * us[0], y4[RFP], cz[0], cz[1], y4[WW], qo);
      stv<4>(VQ(m), qo);
      cz[0] = fmaf(ab[0], cz[0], ab[2]);  // (identity map on the padding steps)
      cz[1] = fmaf(ab[1], cz[1], ab[3]);
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
  //     Lambda_k = A_k (Lambda_{k+1} + post_k) + g_k,    post_k = gK at k = K-1 (the terminal injection), else 0.
  // lane_step: one more (earlier) step in front of the composition so far; lane_entry: Lambda entering this lane from the
  // steps above it (0 beyond the last lane).
  auto lane_step = [&](Aff& lm, int m, float A, float g, float gK) {
    const bool valid = k0 + m < K, last = k0 + m == K - 1;
    Aff st = {A, fmaf(A, last ? gK : 0.f, g)};
    if (!valid) st = {1.f, 0.f};
    lm = after(st, lm);
  };
  auto lane_entry = [&](const Aff& lm) {
    const Aff sc = scan_down32(lm, lane);  // applied to 0: Lambda at this lane's first grid point
    const float nxt = lane_above(sc.b);
    return l == 31 ? 0.f : nxt;
  };
  float sv[NSP], degb[NSP], lam0[NSP];
  VIHDS_UNROLL for (int j = 0; j < NSP; ++j) { sv[j] = 0.f; degb[j] = 0.f; }
  float svt[2] = {0.f, 0.f}, svr[2] = {0.f, 0.f}, c1b[2] = {0.f, 0.f}, c2b[2] = {0.f, 0.f};

  // level 2 (yfp, cfp) with the promoter adjoints -> stage injections for luxR / lasR
  {
    float lam[2];
    const float gK2[2] = {ginj(YFP, qK, xK), ginj(CFP, qK, xK)};
    {
      Aff lm[2] = {{1.f, 0.f}, {1.f, 0.f}};
      VIHDS_ITEMLOOP(3) for (int m = ITEMS - 1; m >= 0; --m) {
        float qq[4], ab[4], us[NS];
        ldv<4>(VQ(m), qq);
        ldv<4>(VA(m), ab);
        ldv<NS>(VU(m), us);
        VIHDS_UNROLL for (int q = 0; q < 2; ++q) lane_step(lm[q], m, ab[q], ginj(YFP + q, qq, Kc * us[0]), gK2[q]);
      }
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) lam[q] = lane_entry(lm[q]);
    }
    v2f dz2 = {0.f, 0.f}, sz2 = {0.f, 0.f}, drs2 = {0.f, 0.f}, srs2 = {0.f, 0.f};
    v2f svt2 = {0.f, 0.f}, svr2 = {0.f, 0.f}, c1b2 = {0.f, 0.f}, c2b2 = {0.f, 0.f};
    VIHDS_ITEMLOOP(4) for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      const bool last = k0 + m == K - 1;
      float gam[NS], us[NS], qq[4], y4[4], z2[2], ab[4];
      ldv<NS>(VG(m), gam);
      ldv<NS>(VU(m), us);
      ldv<4>(VQ(m), qq);
      ldv<4>(VY(m), y4);
      ldv<2>(VZ(m), z2);
      ldv<4>(VA(m), ab);
      // stage values of luxR / lasR again, promoters, stage values of yfp / cfp (pairs: see Rk)
      v2f YRS[NS], aRS[NS], FRS[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { aRS[s] = gam[s] + delta2[1]; FRS[s] = F12[1]; }
      R::real(it.h, aRS, FRS, v2f{y4[LUXR], y4[LASR]}, YRS);
      float gb[NS];
      v2f JRS[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { JRS[s] = v2f{0.f, 0.f}; gb[s] = 0.f; }
      {
        v2f as[NS], Fs[NS], t[NS], rd[NS], Y[NS], kbar[NS], Jz[NS];
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const v2f kb = fm(pcS2, splat<v2f>(YRS[s].y * YRS[s].y), pcR2 * (YRS[s].x * YRS[s].x));
          rd[s] = rcp2(1.f + kb);
          t[s] = kb * rd[s];
          Fs[s] = pc2 * fm(1.f - pe2, t[s], pe2);
          as[s] = gam[s] + delta2[2];
          Jz[s] = v2f{0.f, 0.f};
        }
        R::real(it.h, as, Fs, v2f{z2[0], z2[1]}, Y);
        const v2f lam2 = {lam[0], lam[1]};
        const v2f lin = lam2 + (last ? v2f{gK2[0], gK2[1]} : v2f{0.f, 0.f});  // Lambda_{k+1} (+ terminal injection)
        R::reverse(it.h, as, it.valid ? lin : v2f{0.f, 0.f}, Jz, kbar);
        if (it.valid) {
          const float xg = Kc * us[0];
          const v2f nl = fm(v2f{ab[0], ab[1]}, lin, v2f{ginj(YFP, qq, xg), ginj(CFP, qq, xg)});
          lam[0] = nl.x;
          lam[1] = nl.y;
        }
        const v2f cm = pc2 * (1.f - pe2);
        VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
          const v2f abar = -Y[s] * kbar[s];
          gb[s] += abar.x;
          gb[s] += abar.y;
          dz2 += abar;
          sz2 += kbar[s];
          svt2 = fm(kbar[s], t[s], svt2);
          svr2 = fm(kbar[s], rd[s], svr2);
          const v2f kbb = (kbar[s] * cm) * (rd[s] * rd[s]);  // d t / d kb = rd^2
          c1b2 = fm(kbb, splat<v2f>(YRS[s].x * YRS[s].x), c1b2);
          c2b2 = fm(kbb, splat<v2f>(YRS[s].y * YRS[s].y), c2b2);
          const v2f wR = kbb * (2.f * pcR2), wS = kbb * (2.f * pcS2);
          JRS[s].x = fmaf(wR.y, YRS[s].x, fmaf(wR.x, YRS[s].x, JRS[s].x));
          JRS[s].y = fmaf(wS.y, YRS[s].y, fmaf(wS.x, YRS[s].y, JRS[s].y));
        }
      }
      // the part of luxR / lasR's step adjoint that is driven by the stage injections (linear: taken here; the part
      // driven by Lambda_{k+1} follows after their scan)
      v2f kv[NS];
      const v2f oRS = R::reverse(it.h, aRS, v2f{0.f, 0.f}, JRS, kv);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const v2f abar = -YRS[s] * kv[s];
        gb[s] += abar.x;
        drs2 += abar;
        srs2 += kv[s];
      }
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) gb[s] += (-YRS[s] * kv[s]).y;
      ab[2] = it.valid ? oRS.x : 0.f;  // (the slots of the forward maps' offsets now carry luxR / lasR's injection-driven offsets)
      ab[3] = it.valid ? oRS.y : 0.f;
      stv<4>(VA(m), ab);
      stv<NS>(VB(m), gb);
    }
    degb[YFP] += dz2.x; degb[CFP] += dz2.y; sv[YFP] += sz2.x; sv[CFP] += sz2.y;
    degb[LUXR] += drs2.x; degb[LASR] += drs2.y; sv[LUXR] += srs2.x; sv[LASR] += srs2.y;
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) { svt[q] = q ? svt2.y : svt2.x; svr[q] = q ? svr2.y : svr2.x; c1b[q] = q ? c1b2.y : c1b2.x; c2b[q] = q ? c2b2.y : c2b2.x; }
    lam0[YFP] = lam[0];
    lam0[CFP] = lam[1];
  }
  VIHDS_SCAN_STOP(7)
  // level 1 in two pairs, (luxR, lasR) then (rfp, W): multipliers and injections of every step -> VA, scan, then the
  // Lambda-driven part of the step adjoints
  auto level1_pair = [&](auto JA, auto JB) {
    constexpr int jA = decltype(JA)::value, jB = decltype(JB)::value;
    constexpr bool inj = jA == LUXR;  // luxR / lasR carry the offsets left in VA by the level-2 pass
    constexpr int jp = inj ? 1 : 0;   // the pair's constants (delta2 / F12)
    const float gK[2] = {ginj(jA, qK, xK), ginj(jB, qK, xK)};
    Aff lm[2] = {{1.f, 0.f}, {1.f, 0.f}};
    VIHDS_ITEMLOOP(5) for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      float gam[NS], us[NS], qq[4], ab[4];
      ldv<NS>(VG(m), gam);
      ldv<NS>(VU(m), us);
      ldv<4>(VQ(m), qq);
      ldv<4>(VA(m), ab);
      v2f as[NS], Fz[NS], A, dB;
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { Fz[s] = v2f{0.f, 0.f}; as[s] = gam[s] + delta2[jp]; }
      R::affine(it.h, as, Fz, A, dB);
      float tag[4] = {A.x, A.y, 0.f, 0.f};
      VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
        const int j = q == 0 ? jA : jB;
        tag[2 + q] = ginj(j, qq, Kc * us[0]) + (inj ? ab[2 + q] : 0.f);
        lane_step(lm[q], m, tag[q], tag[2 + q], gK[q]);
      }
      stv<4>(VA(m), tag);
    }
    v2f lam = {lane_entry(lm[0]), lane_entry(lm[1])};
    v2f d2 = {degb[jA], degb[jB]}, s2 = {sv[jA], sv[jB]};
    VIHDS_ITEMLOOP(6) for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      const bool last = k0 + m == K - 1;
      float gam[NS], y4[4], tag[4], gb[NS];
      ldv<NS>(VG(m), gam);
      ldv<4>(VY(m), y4);
      ldv<4>(VA(m), tag);
      ldv<NS>(VB(m), gb);
      v2f as[NS], Fs[NS], Y[NS], kbar[NS], Jz[NS];
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) { as[s] = gam[s] + delta2[jp]; Fs[s] = F12[jp]; Jz[s] = v2f{0.f, 0.f}; }
      R::real(it.h, as, Fs, v2f{y4[jA], y4[jB]}, Y);
      const v2f lin = lam + (last ? v2f{gK[0], gK[1]} : v2f{0.f, 0.f});
      R::reverse(it.h, as, it.valid ? lin : v2f{0.f, 0.f}, Jz, kbar);
      if (it.valid) lam = fm(v2f{tag[0], tag[1]}, lin, v2f{tag[2], tag[3]});
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        const v2f abar = -Y[s] * kbar[s];
        d2 += abar;
        s2 += kbar[s];
        gb[s] += abar.x;
      }
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) gb[s] += (-Y[s] * kbar[s]).y;
      stv<NS>(VB(m), gb);
    }
    degb[jA] = d2.x; degb[jB] = d2.y; sv[jA] = s2.x; sv[jB] = s2.y;
    lam0[jA] = lam.x;
    lam0[jB] = lam.y;
  };
  level1_pair(std::integral_constant<int, LUXR>{}, std::integral_constant<int, LASR>{});
  level1_pair(std::integral_constant<int, RFP>{}, std::integral_constant<int, WW>{});
  VIHDS_SCAN_STOP(8)
  // x: tangent multipliers a_s = -gr_s (1 - 2 u_s), stage injections -gamma_bar gr / K, scan, then r, tlag, K
  float rb = 0.f, tlb = 0.f, gbx = 0.f, lamx;
  {
    auto x_stage = [&](int m, const Item& it, float* sg, float* us, float* ax, float* gbo, float* Jx) {
      ldv<NS>(VU(m), us);
      ldv<NS>(VB(m), gbo);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) {
        sg[s] = stage_sigmoid(it, s);
        const float g = r * sg[s];
        ax[s] = -g * fmaf(-2.f, us[s], 1.f);
        gbo[s] = it.valid ? gbo[s] : 0.f;
        Jx[s] = -gbo[s] * g * invK;
      }
    };
    const float gK = qK[0] + qK[1] * yend[RFP] + qK[2] * fmaf(a530, yend[WW], yend[YFP]) + qK[3] * fmaf(a480, yend[WW], yend[CFP]);
    Aff lm = {1.f, 0.f};
    VIHDS_ITEMLOOP(7) for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      float sg[NS], us[NS], ax[NS], gbo[NS], Jx[NS], Fz[NS], kv[NS], qq[4], y4[4], z2[2], tag[4], A, dB;
      x_stage(m, it, sg, us, ax, gbo, Jx);
      ldv<4>(VQ(m), qq);
      ldv<4>(VY(m), y4);
      ldv<2>(VZ(m), z2);
      VIHDS_UNROLL for (int s = 0; s < NS; ++s) Fz[s] = 0.f;
      R::affine(it.h, ax, Fz, A, dB);
      const float off = it.valid ? R::reverse(it.h, ax, 0.f, Jx, kv) : 0.f;
      tag[0] = A;
      tag[1] = off + qq[0] + qq[1] * y4[RFP] + qq[2] * fmaf(a530, y4[WW], z2[0]) + qq[3] * fmaf(a480, y4[WW], z2[1]);
      tag[2] = 0.f; tag[3] = 0.f;
      stv<4>(VA(m), tag);
      lane_step(lm, m, tag[0], tag[1], gK);
    }
    float lam = lane_entry(lm);
    VIHDS_ITEMLOOP(8) for (int m = ITEMS - 1; m >= 0; --m) {
      const Item it = item(m);
      const bool last = k0 + m == K - 1;
      float sg[NS], us[NS], ax[NS], gbo[NS], Jx[NS], kbar[NS], tag[4];
      x_stage(m, it, sg, us, ax, gbo, Jx);
      ldv<4>(VA(m), tag);
      const float lin = lam + (last ? gK : 0.f);
      R::reverse(it.h, ax, it.valid ? lin : 0.f, Jx, kbar);
      if (it.valid) lam = fmaf(tag[0], lin, tag[1]);
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
  // ---- epilogue: sums over the time axis through LDS (lane a of a trajectory adds up accumulator a), raw accumulators
  //      -> gradients of the theta rows ------------------------------------------------------------------------------------
  {
    // The gradients leave through LDS: lane 0 of the trajectory lays them out in slot order (constant offsets), then lane
    // l stores slots l and l + 32 -- two store instructions per wavefront instead of one per slot with two lanes at work.
    unsigned long long put_mask = 0ull;
    float* gout = lds + DR_SCAN_NACC * DR_SCAN_RED_STRIDE + DR_SCAN_TPB * DR_SCAN_NACC + tib * 64;  // [TPB][64], behind `tot`
    auto put = [&](int slot, float v) {
      gout[slot] = v;
      put_mask |= 1ull << slot;
    };
    float acc[DR_SCAN_NACC];
    VIHDS_UNROLL for (int j = 0; j < NSP; ++j) { acc[j] = sv[j]; acc[6 + j] = degb[j]; }
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) { acc[12 + q] = svt[q]; acc[14 + q] = svr[q]; acc[16 + q] = c1b[q]; acc[18 + q] = c2b[q]; }
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) acc[20 + j] = precb[j];
    acc[24] = rb; acc[25] = tlb; acc[26] = gbx; acc[27] = a530b; acc[28] = a480b; acc[29] = 0.f; acc[30] = 0.f; acc[31] = 0.f;
    block_sync_lds();  // every wavefront's per-step records are dead: the reduction buffer overlays them
#ifdef VIHDS_SCAN_STAMPS
    VIHDS_SCAN_STOP(13)
#endif
    float* red = lds;                       // [NACC][NT]
    float* tot = lds + DR_SCAN_NACC * DR_SCAN_RED_STRIDE;   // [TPB][NACC]
    VIHDS_UNROLL for (int q = 0; q < 29; ++q) red[q * DR_SCAN_RED_STRIDE + tid] = acc[q];
    wave_sync();
    {
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
      const float* row = red + l * DR_SCAN_RED_STRIDE + tib * 32;  // accumulator l of this trajectory, its 32 lanes' partial sums
      VIHDS_UNROLL for (int q = 0; q < 8; ++q) {
        float v[4];
        ldv<4>(row + 4 * q, v);
        VIHDS_UNROLL for (int e = 0; e < 4; ++e) s4[e] += v[e];
      }
      tot[tib * DR_SCAN_NACC + l] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    wave_sync();
    VIHDS_UNROLL for (int q = 0; q < DR_SCAN_NACC / 4; ++q) ldv<4>(tot + tib * DR_SCAN_NACC + 4 * q, acc + 4 * q);
    VIHDS_UNROLL for (int j = 0; j < NSP; ++j) { sv[j] = acc[j]; degb[j] = acc[6 + j]; }
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) { svt[q] = acc[12 + q]; svr[q] = acc[14 + q]; c1b[q] = acc[16 + q]; c2b[q] = acc[18 + q]; }
    VIHDS_UNROLL for (int j = 0; j < 4; ++j) precb[j] = acc[20 + j];
    rb = acc[24]; tlb = acc[25]; gbx = acc[26]; a530b = acc[27]; a480b = acc[28];
    // c P = c e + c (1 - e) t:  c_bar = sv e + svt (1 - e),  e_bar = c sum kbar (1 - t) = c svr
    float cbar[2], ebar[2];
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      cbar[q] = fmaf(svt[q], 1.f - pe[q], sv[YFP + q] * pe[q]);
      ebar[q] = pc[q] * svr[q];
    }
    const float fRb = c1b[0] * pKR[0] + c1b[1] * pKR[1], fSb = c2b[0] * pKS[0] + c2b[1] * pKS[1];
    const float rcb = sv[RFP] + sv[WW] + sv[LUXR] * aR + sv[LASR] * aS + cbar[0] * aY + cbar[1] * aC;
    const typename D::HillAdj HA = D::hill_vjp_pass(l & 7, c, H, fRb, fSb, pass_h);
    if (l == 0) {  // (the initial-state adjoints sit in lane 0 of the trajectory)
      put(M::SI + 0, lamx);
      put(M::SI + 1, lam0[RFP]); put(M::SI + 2, lam0[YFP]); put(M::SI + 3, lam0[CFP]);
      put(M::SI + 4, lam0[LUXR]); put(M::SI + 5, lam0[LASR]);
      VIHDS_UNROLL for (int j = 0; j < 4; ++j) put(M::NSLOT + j, precb[j]);
      put(M::S_r, rb * pass_r);
      put(M::S_K, gbx * invK * invK * pass_K);
      put(M::S_tlag, -4.f * tlb);
      put(M::S_rc, rcb);
      put(M::S_drfp, degb[RFP] * pass_d[0]);
      put(M::S_dyfp, degb[YFP] * pass_d[1]);
      put(M::S_dcfp, degb[CFP] * pass_d[2]);
      put(M::S_dR, degb[LUXR] * pass_d[3]);
      put(M::S_dS, degb[LASR] * pass_d[4]);
      put(M::S_e81, ebar[0]); put(M::S_KGR81, c1b[0] * fR); put(M::S_KGS81, c2b[0] * fS);
      put(M::S_e76, ebar[1]); put(M::S_KGR76, c1b[1] * fR); put(M::S_KGS76, c2b[1] * fS);
      put(M::S_aYFP, cbar[0] * rc); put(M::S_aCFP, cbar[1] * rc);
      put(M::S_a530, a530b); put(M::S_a480, a480b);
      put(M::S_aR, sv[LUXR] * rc); put(M::S_aS, sv[LASR] * rc);
      put(M::S_nR, HA.nR); put(M::S_nS, HA.nS); put(M::S_H0, HA.H0); put(M::S_H1, HA.H1);
      if (VERSION == 1) { put(M::S_H2, HA.H2); put(M::S_H3, HA.H3); }
    }
    put_mask = (unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)put_mask) |
               ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)(put_mask >> 32)) << 32);
    wave_sync();
    VIHDS_UNROLL for (int q = 0; q < 2; ++q) {
      const int slot = l + 32 * q;
      if (live && ((put_mask >> slot) & 1ull)) a.g_theta[(size_t)out_row[q] * n + i] = gout[slot];
    }
  }
  if (THETA) {
    rng_advance(t.rng, tk.u, TW * 64);
    rng_advance(t.crng, tk.c, TW * 64);
  }
#ifdef VIHDS_SCAN_STAMPS
  VIHDS_SCAN_STOP(10)
#endif
}

template <int VERSION, int SOLVER, int ITEMS>
__global__ void __launch_bounds__(DR_SCAN_THREADS) dr_scan_train_kernel(OdeArgs a) {
  extern __shared__ float lds[];
  ThetaStageArgs none = {};
  dr_scan_train_body<VERSION, SOLVER, ITEMS, false>(a, lds, 0, none);
}
// sampling stage + conditioner + everything above: the whole decoder side of a training step
template <int VERSION, int SOLVER, int ITEMS>
__global__ void __launch_bounds__(DR_SCAN_THREADS) dr_scan_train_theta_kernel(OdeArgs a, int nb_max, ThetaStageArgs t) {
  extern __shared__ float lds[];
  dr_scan_train_body<VERSION, SOLVER, ITEMS, true>(a, lds, nb_max, t);
}

// returns VIHDS_E_UNSUPPORTED when the time grid is longer than 32 lanes x 4 steps (or the sampling stage's tables do
// not fit)
template <int VERSION>
inline int launch_dr_scan_train(int solver, const OdeArgs& a, hipStream_t st, const ThetaStageArgs* ts = nullptr) {
  const int K = a.T - 1;
  const int items = (K + 31) / 32;
  if (items > 4) return VIHDS_E_UNSUPPORTED;
  const int nb_max = min(a.B, (DR_SCAN_TPB - 1) / a.S + 2);
  if (ts) {
    if (ts->P > 64 || ts->n_rows > 64 || ts->E > 32) return VIHDS_E_UNSUPPORTED;
    if (ts->E * a.D > 128) return VIHDS_E_UNSUPPORTED;  // (every wavefront keeps the conditioner's weights in 128 floats)
    if (a.B >= (1 << 24) || a.D >= (1 << 24)) return VIHDS_E_UNSUPPORTED;
  }
  for (int q = 0; q < DrConstant<VERSION>::NSLOT + 4; ++q)
    if (a.slot_row[q] >= 64) return VIHDS_E_UNSUPPORTED;
  const dim3 grid((a.n + DR_SCAN_TPB - 1) / DR_SCAN_TPB), block(DR_SCAN_THREADS);
#define VIHDS_SCASE2(SV, IT)                                                                                \
  case IT: {                                                                                                \
    const size_t lds = (dr_scan_lds_floats<SV>(IT) + (ts ? dr_scan_theta_floats(ts->E, a.D) : 0)) * sizeof(float); \
    auto kern = dr_scan_train_kernel<VERSION, SV, IT>;                                                      \
    auto kern_t = dr_scan_train_theta_kernel<VERSION, SV, IT>;                                              \
    static bool opted = false, opted_t = false;                                                             \
    bool& have = ts ? opted_t : opted;                                                                      \
    if (lds > 64 * 1024 && !have) {                                                                         \
      const void* f = ts ? reinterpret_cast<const void*>(kern_t) : reinterpret_cast<const void*>(kern);     \
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)     \
        return VIHDS_E_HIP;                                                                                 \
      have = true;                                                                                          \
    }                                                                                                       \
    if (ts) hipLaunchKernelGGL(kern_t, grid, block, lds, st, a, nb_max, *ts);                               \
    else hipLaunchKernelGGL(kern, grid, block, lds, st, a);                                                 \
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
