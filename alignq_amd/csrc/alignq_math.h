// Device-side ALIGNQ-EXP32 / ALIGNQ-NERF32 (spec: gen_erf32_coeffs.py docstring).
// Every step is one IEEE-754 single operation (fma / mul / add / rint / exact 2^k scaling), so the
// result is bit-identical to the scalar C statement of the same spec used by the test oracle.
// Translation units including this header are compiled with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alignq_erf32_coeffs.h"

namespace alignq {

__device__ __forceinline__ float pow2i(int e) { return __int_as_float((e + 127) << 23); }

__device__ __forceinline__ float exp32(float x) {
  float nf = rintf(x * ALIGNQ_LOG2E);
  float r = __fmaf_rn(nf, -ALIGNQ_LN2_HI, x);
  r = __fmaf_rn(nf, -ALIGNQ_LN2_LO, r);
  float p = ALIGNQ_PE5;
  p = __fmaf_rn(p, r, ALIGNQ_PE4);
  p = __fmaf_rn(p, r, ALIGNQ_PE3);
  p = __fmaf_rn(p, r, ALIGNQ_PE2);
  p = __fmaf_rn(p, r, ALIGNQ_PE1);
  p = __fmaf_rn(p, r, ALIGNQ_PE0);
  float r2 = r * r;
  float e = 1.0f + __fmaf_rn(r2, p, r);
  int n = (int)nf;
  int h = n >> 1;
  float res = e * pow2i(h) * pow2i(n - h);
  if (!(x >= -104.0f)) res = (x != x) ? x : 0.0f;
  if (x > 88.7f) res = __int_as_float(0x7f800000);
  return res;
}

// ---- ALIGNQ-NERF32 (round 3): nerf32(y) ~ erf(y/sqrt(2)) = 2*Phi(y)-1, ONE evaluated branch per element ------------
// Spec: gen_erf32_coeffs.py (table in include/alignq_erf32_coeffs.h), C statement oracle/alignq_oracle.c:oq_nerf32_1.
//   a = min(|y|, 5.625) (NaN-propagating: v_minimum3_f32);  u = a + 2^20 (ulp 1/8: the low mantissa bits of u ARE the node
//   index k = round(8a));  d = a - P[k] (exact);  res = C0[k] + d*(C1[k] + d*(C2[k] + d*(C3[k] + d*C4[k]))) as four fma.
// 9 vector ops + 2 LDS reads (round 1-2 erf32: both polynomial regions and an exp for every lane, 34 ops).  The table
// (46 nodes x 32 B) lives in LDS: per node a 16-byte record C1..C4 and, 736 B behind it at the same 16-byte stride,
// (C0, P): one address register, ds_read_b128 + ds_read_b64.  Lanes reading the same node broadcast; different nodes
// collide only 16 nodes apart (|y| differing by 2), so N(0,1) data reads almost conflict-free.  A NaN input gives a NaN
// address (an LDS read never faults: out of range returns 0) and a NaN result.
#define ALIGNQ_NERF_LDS_FLOATS (ALIGNQ_NERF_N * 8)
static __device__ const float g_nerf_tab[2][ALIGNQ_NERF_N][4] = {ALIGNQ_NERF_POLY, ALIGNQ_NERF_CENTRE};   // the LDS image

// every thread of the block calls it; the CALLER places a __syncthreads() before the first nerf32
__device__ __forceinline__ void nerf_tab_load(float* tab) {
  for (int i = threadIdx.x; i < ALIGNQ_NERF_LDS_FLOATS; i += blockDim.x) tab[i] = (&g_nerf_tab[0][0][0])[i];
}
// The same in two halves for the latency-bound kernels: request the image into registers first, issue the kernel's own
// loads, THEN store it (memory returns in order, so the store waits for the image only) and pass a barrier.
template <int NT>
struct NerfRegs {
  float v[(ALIGNQ_NERF_LDS_FLOATS + NT - 1) / NT];
};
template <int NT>
__device__ __forceinline__ NerfRegs<NT> nerf_tab_fetch() {
  NerfRegs<NT> r;
#pragma unroll
  for (int j = 0; j < (ALIGNQ_NERF_LDS_FLOATS + NT - 1) / NT; j++) {
    const int i = threadIdx.x + j * NT;
    r.v[j] = (&g_nerf_tab[0][0][0])[i < ALIGNQ_NERF_LDS_FLOATS ? i : ALIGNQ_NERF_LDS_FLOATS - 1];
  }
  return r;
}
template <int NT>
__device__ __forceinline__ void nerf_tab_store(float* tab, const NerfRegs<NT>& r) {
#pragma unroll
  for (int j = 0; j < (ALIGNQ_NERF_LDS_FLOATS + NT - 1) / NT; j++) {
    const int i = threadIdx.x + j * NT;
    if (i < ALIGNQ_NERF_LDS_FLOATS) tab[i] = r.v[j];
  }
}

// `tab` as the kernels pass it around: the LDS byte address of the image plus 16 * -bits(2^20) (mod 2^32), wave-uniform, so
// that ONE v_lshl_add_u32 turns the bits of u into the address of the node's first record (the second record is the
// instruction's offset field).
typedef float nerf_f4 __attribute__((ext_vector_type(4)));
typedef float nerf_f2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const nerf_f4 LdsF4;
typedef __attribute__((address_space(3))) const nerf_f2 LdsF2;
struct NerfTab {
  uint32_t base;
};
__device__ __forceinline__ NerfTab nerf_tab(const float* tab) {
  NerfTab t;
  t.base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)tab + 0x68000000u;   // -(0x49800000 << 4)
  // opaque to the optimiser: otherwise it folds the (link-time constant) address into two different 32-bit literals and
  // spends three vector instructions per element on the two addresses instead of one v_lshl_add_u32 + an offset field
  asm volatile("" : "+s"(t.base));
  return t;
}

__device__ __forceinline__ float nerf32(float y, const NerfTab tab) {
  const float a = __builtin_elementwise_minimum(__builtin_fabsf(y), ALIGNQ_NERF_YMAX);
  const float u = __fadd_rn(a, ALIGNQ_NERF_MAGIC);
  const uint32_t addr = (__float_as_uint(u) << 4) + tab.base;
  const nerf_f4 c = *reinterpret_cast<LdsF4*>(addr);
  const nerf_f2 p = *reinterpret_cast<LdsF2*>(addr + ALIGNQ_NERF_N * 16);
  const float d = __fsub_rn(a, p.y);
  float q = __fmaf_rn(c.w, d, c.z);
  q = __fmaf_rn(q, d, c.y);
  q = __fmaf_rn(q, d, c.x);
  return copysignf(__fmaf_rn(q, d, p.x), y);
}

#define ALIGNQ_LOG_SQRT_2PI_F 0.91893853320467274178f
#define ALIGNQ_TWO_OVER_SQRT_2PI 0.79788456080286535588f  // 2*phi(0)

// IEEE-754 division b/n of a level index by n = 2^k-1 through the correctly rounded reciprocal y = RN(1/n):
//   q0 = b*y; r = fma(-q0, n, b) (exact remainder); q = fma(r, y, q0)      (3 VALU ops instead of ~10)
// The arithmetic SPEC remains "IEEE division" (the C oracle divides).  tests/native/verify_div.c proves equality
// exhaustively for k<=16 over every integer-valued |b| <= 8*n+2 and -0.  The fma chain turns b = -0 into +0: the sign
// bit of b is OR-ed back (q is +0 or carries b's sign already; NaN stays NaN) - one v_and_or_b32.
__device__ __forceinline__ float div_levels(float b, float n, float y) {
  const float q0 = __fmul_rn(b, y);
  const float r = __fmaf_rn(-q0, n, b);
  const float q = __fmaf_rn(r, y, q0);
  return __uint_as_float(__float_as_uint(q) | (__float_as_uint(b) & 0x80000000u));
}

// 1 + nerf32((v-m)*rs): twice Normal(m,s).cdf (torch/distributions/normal.py; reference model/quantization.py:50-51).
// The halving and the doubling that follow in the reference (c = 0.5*(1+erf), 2c-1) are exact, so both trees continue from u1.
__device__ __forceinline__ float gauss_u1(float v, float m, float rs, const NerfTab tab) {
  return __fadd_rn(1.0f, nerf32(__fmul_rn(__fsub_rn(v, m), rs), tab));
}
__device__ __forceinline__ float gauss_cdf32(float v, float m, float rs, const NerfTab tab) {
  return __fmul_rn(0.5f, gauss_u1(v, m, rs, tab));
}

// Level-division context: n = 2^k-1 and, when every |bin| of the call is known to stay within the exhaustively
// verified range (|t| <= 8: activations with act_range <= 8, weights), yn = RN(1/n) for div_const; yn == 0 selects
// the hardware divider (arbitrary inputs, e.g. the stand-alone uniform_quantize).  yn is wave-uniform => scalar branch.
struct Levels {
  float n, yn;
};
__host__ __device__ __forceinline__ Levels make_levels(int k, bool bounded) {
  Levels L;
  L.n = (float)((1 << (k & 31)) - 1);
  L.yn = (bounded && k != 32 && k != 1) ? 1.0f / L.n : 0.0f;
  return L;
}

// uniform_quantize(k).forward (model/quantization.py:23-31) on a transformed value.
// k==32 -> identity, k==1 -> sign.  *bin receives the integer level.
__device__ __forceinline__ float round_bins(float t, int k, const Levels& L, float* bin) {
  if (k == 32) { *bin = t; return t; }
  if (k == 1) { float s = (float)((t > 0.0f) - (t < 0.0f)); *bin = s; return s; }
  float b = rintf(__fmul_rn(t, L.n));
  *bin = b;
  if (L.yn != 0.0f) return div_levels(b, L.n, L.yn);
  return __fdiv_rn(b, L.n);
}

// BOUNDED: the caller has checked L.yn != 0 (k not 1 or 32, |level index| within the verified range) for the whole launch,
// which removes the per-element scalar branches on k from the streaming loops: `ALIGNQ_BOUNDED_SWITCH(L, body)` runs
// `body` with a constexpr bool kBounded in both forms.
template <bool BOUNDED>
__device__ __forceinline__ float round_bins_t(float t, int k, const Levels& L, float* bin) {
  if (BOUNDED) {
    const float b = rintf(__fmul_rn(t, L.n));
    *bin = b;
    return div_levels(b, L.n, L.yn);
  }
  return round_bins(t, k, L, bin);
}
#define ALIGNQ_BOUNDED_SWITCH(L, ...)                          \
  if ((L).yn != 0.0f) {                                        \
    constexpr bool kBounded = true;                            \
    __VA_ARGS__                                                \
  } else {                                                     \
    constexpr bool kBounded = false;                           \
    __VA_ARGS__                                                \
  }

// activation transform + quantise for one element; returns x_q, *t_pre = pre-round transform
template <int FORMULA, bool BOUNDED = false>
__device__ __forceinline__ float act_quant1(float x, int k, const Levels& n, float r, float* t_pre, float* bin,
                                            const NerfTab tab) {
  const float u1 = __fadd_rn(1.0f, nerf32(x, tab));          // (x - 0) * 1 == x
  if (FORMULA == 0) {
    float t = __fmul_rn(__fsub_rn(u1, 1.0f), r);             // (2c - 1) * r with 2c == u1 exactly
    *t_pre = t;
    return round_bins_t<BOUNDED>(t, k, n, bin);
  } else {
    const float c = __fmul_rn(0.5f, u1);
    *t_pre = c;
    float q = round_bins_t<BOUNDED>(c, k, n, bin);
    return __fmul_rn(__fsub_rn(__fmul_rn(q, 2.0f), 1.0f), r);
  }
}

// d t / d x = r * 2*phi(x)   (tolerance-checked, not bit-checked: fast exp is fine)
__device__ __forceinline__ float act_jac(float x, float r) {
  return r * ALIGNQ_TWO_OVER_SQRT_2PI * __expf(-0.5f * x * x);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Sum over the 64 lanes on the DPP path (row_shr 1/2/4/8 inside the rows of 16, then row_bcast15 / row_bcast31 across rows:
// an inclusive scan whose last lane holds the total), ~10x cheaper than six ds_bpermute round trips per dword.  Returns the
// total in EVERY lane (v_readlane of lane 63).  Fixed summation order.
__device__ __forceinline__ double wave_sum_d_dpp(double v) {
#define ALIGNQ_DPP_STEP(CTRL, ROWMASK)                                                                      \
  {                                                                                                         \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, false);            \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, false);            \
    v += __hiloint2double(hi, lo);                                                                          \
  }
  ALIGNQ_DPP_STEP(0x111, 0xf)   // row_shr:1
  ALIGNQ_DPP_STEP(0x112, 0xf)   // row_shr:2
  ALIGNQ_DPP_STEP(0x114, 0xf)   // row_shr:4
  ALIGNQ_DPP_STEP(0x118, 0xf)   // row_shr:8
  ALIGNQ_DPP_STEP(0x142, 0xa)   // row_bcast:15 into rows 1 and 3
  ALIGNQ_DPP_STEP(0x143, 0xc)   // row_bcast:31 into rows 2 and 3
#undef ALIGNQ_DPP_STEP
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63), hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

}  // namespace alignq

// ---- weight quantiser element functions shared by the single- and multi-tensor kernels -----------------
namespace alignq {

struct WeightConsts {
  float m, rs, var2, logs;
  Levels nlev;
};

__device__ __forceinline__ WeightConsts weight_consts(float m, float s, int k) {
  WeightConsts c;
  c.m = m;
  c.rs = __fdiv_rn(1.0f, s);
  c.var2 = __fmul_rn(2.0f, __fmul_rn(s, s));
  c.logs = (float)log((double)s);
  c.nlev = make_levels(k, true);   // |weight transform| <= 1
  return c;
}

// weight_quantize_fn.forward for one element given (m, s): returns W_q, *t = the tree's weight_cdf
template <int FORMULA>
__device__ __forceinline__ float weight_quant1(float v, const WeightConsts& wc, int k, float* t, float* bin,
                                               const NerfTab tab) {
  const float u1 = gauss_u1(v, wc.m, wc.rs, tab);
  if (FORMULA == 0) {
    *t = __fsub_rn(u1, 1.0f);
    return round_bins(*t, k, wc.nlev, bin);
  }
  const float c = __fmul_rn(0.5f, u1);
  *t = c;
  return __fsub_rn(__fmul_rn(round_bins(c, k, wc.nlev, bin), 2.0f), 1.0f);
}

// weight_pdf = exp(Normal(m,s).log_prob(v)) * 2   (model/quantization.py:58)
__device__ __forceinline__ float weight_pdf2(float v, const WeightConsts& wc) {
  float d = __fsub_rn(v, wc.m);
  float lp = __fsub_rn(__fsub_rn(__fdiv_rn(-__fmul_rn(d, d), wc.var2), wc.logs), ALIGNQ_LOG_SQRT_2PI_F);
  return __fmul_rn(exp32(lp), 2.0f);
}

// backward helpers: P = 2*pdf_N(m,s)(w), z = (w-m)/s ; cs = 2/(s*sqrt(2*pi))
__device__ __forceinline__ void weight_PZ(float w, float m, float rs, float cs, float* P, float* z) {
  float zz = (w - m) * rs;
  *z = zz;
  *P = cs * __expf(-0.5f * zz * zz);
}

// block-wide sum of two doubles (result in every thread); sm: >= 2*16 doubles of LDS
__device__ __forceinline__ void block_sum2d(double& a, double& b, double* sm) {
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) { sm[w] = a; sm[16 + w] = b; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  a = 0; b = 0;
  for (int i = 0; i < nw; i++) { a += sm[i]; b += sm[16 + i]; }
}

}  // namespace alignq

// ---- fast (non-bit-specified) helpers for BACKWARD kernels: results are tolerance-checked (1e-5) -------
namespace alignq {

// Abramowitz-Stegun 7.1.26, |abs err| <= 1.5e-7 (+ fast exp / rcp rounding); ~14 VALU ops.
__device__ __forceinline__ float erf_fast(float x) {
  const float a = fabsf(x);
  const float t = __frcp_rn(__fmaf_rn(0.3275911f, a, 1.0f));
  float p = 1.061405429f;
  p = __fmaf_rn(p, t, -1.453152027f);
  p = __fmaf_rn(p, t, 1.421413741f);
  p = __fmaf_rn(p, t, -0.284496736f);
  p = __fmaf_rn(p, t, 0.254829592f);
  const float e = __expf(-a * a);
  const float r = __fmaf_rn(-p * t, e, 1.0f);
  return copysignf(r, x);
}

// t = r*(2*Phi(x)-1) = r*erf(x/sqrt(2)) and its derivative r*2*phi(x), sharing one exponential
__device__ __forceinline__ void act_transform_fast(float x, float r, float* t, float* jac) {
  const float z = x * 0.70710678118654752440f;
  const float a = fabsf(z);
  const float tt = __frcp_rn(__fmaf_rn(0.3275911f, a, 1.0f));
  float p = 1.061405429f;
  p = __fmaf_rn(p, tt, -1.453152027f);
  p = __fmaf_rn(p, tt, 1.421413741f);
  p = __fmaf_rn(p, tt, -0.284496736f);
  p = __fmaf_rn(p, tt, 0.254829592f);
  const float e = __expf(-a * a);               // = exp(-x^2/2)
  *t = r * copysignf(__fmaf_rn(-p * tt, e, 1.0f), x);
  *jac = r * ALIGNQ_TWO_OVER_SQRT_2PI * e;
}

// The same pair without the IEEE reciprocal (v_rcp_f32: 1 ulp; the rational form's own error is 1.5e-7) and with ONE exp2 shared
// by the erf tail and the Gaussian (rjac = r * 2*phi(0)): 13 full-rate + 2 quarter-rate instructions, no table (the backward
// kernels are as much LDS- as VALU-bound: the 24 B per element of nerf32's table reads cost more than the extra arithmetic).
__device__ __forceinline__ void act_transform_rcp(float x, float r, float rjac, float* t, float* jac) {
  const float a = fabsf(x) * 0.70710678118654752440f;
  float tt = __builtin_amdgcn_rcpf(__fmaf_rn(0.3275911f, a, 1.0f));
#ifdef ALIGNQ_DIAG_TRANS_NOP      /* diagnostic build: N wait states behind the transcendental (NOTES.md round 6) */
  asm volatile("s_nop %1" : "+v"(tt) : "n"(ALIGNQ_DIAG_TRANS_NOP));
#endif
  float p = 1.061405429f;
  p = __fmaf_rn(p, tt, -1.453152027f);
  p = __fmaf_rn(p, tt, 1.421413741f);
  p = __fmaf_rn(p, tt, -0.284496736f);
  p = __fmaf_rn(p, tt, 0.254829592f);
  float e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);     // exp(-x^2/2) = 2^(-x^2 * log2(e)/2)
#ifdef ALIGNQ_DIAG_TRANS_NOP      /* diagnostic build: N wait states behind the transcendental (NOTES.md round 6) */
  asm volatile("s_nop %1" : "+v"(e) : "n"(ALIGNQ_DIAG_TRANS_NOP));
#endif
  *t = r * copysignf(__fmaf_rn(-p * tt, e, 1.0f), x);
  *jac = rjac * e;
}

}  // namespace alignq
