// Device-side ALIGNQ-EXP32 / ALIGNQ-ERF32 (spec: gen_erf32_coeffs.py docstring).
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

// exp32 for arguments known to lie in [-40, 0] (the erf tail): no range guards, single 2^n scale.
__device__ __forceinline__ float exp32_neg_small(float x) {
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
  // one exact scaling by 2^n (v_ldexp_f32) == the spec's e*2^h*2^(n-h): no subnormals or overflow for -58 <= n <= 0
  return __builtin_ldexpf(e, (int)nf);
}

// Branch-free form: both polynomial regions are evaluated and selected per lane (v_cndmask).  Issue rates measured on
// MI355X (tools/src/valu_rate.hip): v_fma/v_mul/v_add ~2.7 cycles per wave64 instruction, v_rndne/v_min/v_cmp/v_cndmask/
// v_ldexp/v_cvt/v_bfi 4, v_exp/v_rcp 8, v_pk_fma_f32 ~4.9 (so packed fp32 does not pay here) - hence as few selects,
// compares and conversions as the spec allows.  On random data almost
// every wave has lanes in both regions, so a branchy form executes both anyway and pays the exec-mask/branch traffic
// on top (measured in the ISA of the site kernels: ~4 scalar/branch instructions per element).  Values are identical
// to the branchy statement of the spec (same operations per lane).
__device__ __forceinline__ float erf32(float x) {
  const float a = fabsf(x);
  // region A: a < 0.875
  const float s = a * a;
  float pa = ALIGNQ_PA6;
  pa = __fmaf_rn(pa, s, ALIGNQ_PA5);
  pa = __fmaf_rn(pa, s, ALIGNQ_PA4);
  pa = __fmaf_rn(pa, s, ALIGNQ_PA3);
  pa = __fmaf_rn(pa, s, ALIGNQ_PA2);
  pa = __fmaf_rn(pa, s, ALIGNQ_PA1);
  pa = __fmaf_rn(pa, s, ALIGNQ_PA0);
  const float ra = __fmaf_rn(a, pa, a);
  // region B: 0.875 <= a < 4 (the argument is clamped so the unselected lanes stay finite)
  const float ab = fminf(a, ALIGNQ_ERF_HI);
  float pb = ALIGNQ_PB7;
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB6);
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB5);
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB4);
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB3);
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB2);
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB1);
  pb = __fmaf_rn(pb, ab, ALIGNQ_PB0);
  const float rb = 1.0f - exp32_neg_small(-pb);
  // One select covers the spec's three cases: for a >= ERF_HI the clamped region-B value is 1 - 1.54e-8 == 1.0f exactly
  // (checked against the oracle by tests/test_oracle_c.py), and a NaN fails (a >= T), picking ra = fma(NaN, ., NaN).
  const float res = (a >= ALIGNQ_ERF_T) ? rb : ra;
  return copysignf(res, x);
}

#define ALIGNQ_SQRT2F 1.41421356237309504880f
#define ALIGNQ_LOG_SQRT_2PI_F 0.91893853320467274178f
#define ALIGNQ_TWO_OVER_SQRT_2PI 0.79788456080286535588f  // 2*phi(0)

// IEEE-754 division x/d by a fixed divisor through its correctly rounded reciprocal y = RN(1/d):
//   q0 = x*y; r = fma(-q0, d, x) (exact remainder); q = fma(r, y, q0)      (3 VALU ops instead of ~10)
// The arithmetic SPEC remains "IEEE division" (the C oracle divides).  tests/native/verify_div.c proves equality
// exhaustively for d = float(sqrt(2)) over every finite float with |x| >= 1e-30 (below that the remainder underflows;
// such z are absorbed by 0.5*(1+erf(z)) == 0.5 exactly, so no output changes), both zeros included, and for d = 2^k-1,
// k<=16, over every integer-valued |x| <= 8*d+2 and -0 (the bin indices).  +-inf is passed through like a division would.
__device__ __forceinline__ float div_const(float x, float d, float y) {
  const float q0 = __fmul_rn(x, y);
  const float r = __fmaf_rn(-q0, d, x);
  const float q = __fmaf_rn(r, y, q0);
  // q0 already is the answer for the two cases the correction step mangles: a signed zero (the fma chain turns -0
  // into +0) and +-inf (inf - inf = NaN)
  return __builtin_amdgcn_classf(q0, 0x264) ? q0 : q;     // v_cmp_class: -inf | -0 | +0 | +inf
}
#define ALIGNQ_RCP_SQRT2F 0.707106769084930419921875f   // RN(1/float(sqrt(2)))

// Normal(m,s).cdf in torch's op order (torch/distributions/normal.py; reference
// model/quantization.py:50-51): 0.5*(1+erf((v-m)*(1/s)/sqrt(2))).  rs = 1/s.
__device__ __forceinline__ float gauss_cdf32(float v, float m, float rs) {
  float z = div_const(__fmul_rn(__fsub_rn(v, m), rs), ALIGNQ_SQRT2F, ALIGNQ_RCP_SQRT2F);
  return __fmul_rn(0.5f, __fadd_rn(1.0f, erf32(z)));
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
  if (L.yn != 0.0f) return div_const(b, L.n, L.yn);
  return __fdiv_rn(b, L.n);
}

// activation transform + quantise for one element; returns x_q, *t_pre = pre-round transform
template <int FORMULA>
__device__ __forceinline__ float act_quant1(float x, int k, const Levels& n, float r, float* t_pre, float* bin) {
  float c = gauss_cdf32(x, 0.0f, 1.0f);
  if (FORMULA == 0) {
    float t = __fmul_rn(__fsub_rn(__fmul_rn(c, 2.0f), 1.0f), r);
    *t_pre = t;
    return round_bins(t, k, n, bin);
  } else {
    *t_pre = c;
    float q = round_bins(c, k, n, bin);
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
__device__ __forceinline__ float weight_quant1(float v, const WeightConsts& wc, int k, float* t, float* bin) {
  float c = gauss_cdf32(v, wc.m, wc.rs);
  if (FORMULA == 0) {
    *t = __fsub_rn(__fmul_rn(c, 2.0f), 1.0f);
    return round_bins(*t, k, wc.nlev, bin);
  }
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

}  // namespace alignq
