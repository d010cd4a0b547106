// quant_kernels.hip — elementwise CDF-quantise kernels (activations and weights) for gfx950.
//
// HBM-bound streaming kernels: 16 B/lane coalesced loads/stores (one 1 KiB wave-instruction per
// float4), grid capped at a few waves per SIMD with a grid-stride loop, no LDS.  The only
// reductions (weight mean/std, weight-backward sums) are done in double with a deterministic
// two-level tree (wave shuffle -> LDS -> per-block partial in the workspace -> every block of the
// consumer kernel re-reduces the <= ALIGNQ_WS_BLOCKS partials), so results do not depend on
// scheduling and need no atomics.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "../../include/alignq.h"
#include "alignq_math.h"

using namespace alignq;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;      // 256 CUs x 8 blocks (guide: cap the grid, stride the rest)
constexpr int kWsBlocks = 256;        // partials per reduction (one block per CU)
constexpr int kU = 4;                 // float4 per thread and tile in the streaming kernels (forward)
constexpr int kUb = 2;                // ... backward: two or three input streams each

// Streaming structure (tools/src/stream_bw.hip on MI355X, 2^26 floats, 1 read : 1 write): a block owns TILES of kU x 256
// float4 (thread t takes t, t+256, ...: kU independent 16-byte loads in flight), inputs that are read once come in with
// NON-TEMPORAL loads (they do not displace the output lines the consumer is about to read), the grid covers all tiles up to
// 64 blocks per CU: 6.7 TB/s with the NERF32 arithmetic in the loop against 5.3 TB/s for the grid-stride form this file had
// (hipMemcpy device-to-device on the same box: 5.26 TB/s).  nt on loads AND stores is slower (6.1), nt stores alone equal.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// output store: plain (the consumer may find the lines in the caches) or, for tensors far beyond the caches (`nts`: the launcher's
// choice, >= 2^25 elements), non-temporal - on cold operands loads AND stores marked as streams copy at 6.0 TB/s against 5.7-5.8
// for either alone (tools/src/stream_bw.hip with STREAM_BW_SETS=4)
__device__ __forceinline__ void st4_out(float4* p, const float4 v, const int nts) {
  if (nts) {
    typedef float f32x4_s __attribute__((ext_vector_type(4)));
    f32x4_s t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4_s*>(p));
  } else {
    *p = v;
  }
}
__device__ __forceinline__ float4 ld4_stream(const float4* p) {
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
constexpr int kTileBlocks = 256 * 64;
inline int grid_tiles(int64_t n_vec, int u) {
  int64_t b = (n_vec + (int64_t)kThreads * u - 1) / ((int64_t)kThreads * u);
  if (b < 1) b = 1;
  return (int)(b > kTileBlocks ? kTileBlocks : b);
}

inline int grid_for(int64_t n_vec) {
  int64_t b = (n_vec + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  return (int)(b > kMaxBlocks ? kMaxBlocks : b);
}

// ------------------------------------------------------------------ activations ---------------
// RELU: stores relu(x_q) — `self.relu(self.act_q(...))` of the Office bottleneck (dann_office/model/resnet.py:137-138, 142-143) in one pass
template <int FORMULA, bool BINS, bool RELU = false>
__global__ __launch_bounds__(kThreads) void act_quant_fwd_kernel(const float* __restrict__ x,
                                                                 float* __restrict__ xq,
                                                                 int32_t* __restrict__ bins, int64_t n,
                                                                 int k, float r, int nts) {
  __shared__ __attribute__((aligned(16))) float tab_lds[ALIGNQ_NERF_LDS_FLOATS];
  nerf_tab_load(tab_lds);
  __syncthreads();
  const NerfTab tab = nerf_tab(tab_lds);
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* q4 = reinterpret_cast<float4*>(xq);
  // the transform is ~19 vector instructions per element since round 3: the kernel is bound by memory (structure: top of file)
  ALIGNQ_BOUNDED_SWITCH(nlev,
  for (int64_t i0 = (int64_t)blockIdx.x * (kThreads * kU) + threadIdx.x; i0 < nvec; i0 += kU * stride) {
    float4 v[kU];
_Pragma("unroll")
    for (int u = 0; u < kU; u++) {
      const int64_t i = i0 + u * kThreads;
      v[u] = ld4_stream(x4 + (i < nvec ? i : i0));
    }
_Pragma("unroll")
    for (int u = 0; u < kU; u++) {
      const int64_t i = i0 + u * kThreads;
      float4 o;
      float t, b0, b1, b2, b3;
      o.x = act_quant1<FORMULA, kBounded>(v[u].x, k, nlev, r, &t, &b0, tab);
      o.y = act_quant1<FORMULA, kBounded>(v[u].y, k, nlev, r, &t, &b1, tab);
      o.z = act_quant1<FORMULA, kBounded>(v[u].z, k, nlev, r, &t, &b2, tab);
      o.w = act_quant1<FORMULA, kBounded>(v[u].w, k, nlev, r, &t, &b3, tab);
      if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (i < nvec) {
        st4_out(q4 + i, o, nts);
        if (BINS) {
          int4 bi = make_int4((int)b0, (int)b1, (int)b2, (int)b3);
          reinterpret_cast<int4*>(bins)[i] = bi;
        }
      }
    }
  })
  // tail (n % 4) by the first threads of block 0
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (nvec << 2) + threadIdx.x;
    float t, b;
    const float q = act_quant1<FORMULA>(x[i], k, nlev, r, &t, &b, tab);
    xq[i] = RELU ? fmaxf(q, 0.f) : q;
    if (BINS) bins[i] = (int)b;
  }
}

// uniform_quantize(k).forward alone (model/quantization.py:23-31): y = round(x*n)/n | sign(x) | x
__global__ __launch_bounds__(kThreads) void uniform_quantize_kernel(const float* __restrict__ x,
                                                                    float* __restrict__ y, int64_t n, int k) {
  const Levels nlev = make_levels(k, false);   // arbitrary inputs: hardware divider
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float b;
    y[i] = round_bins(x[i], k, nlev, &b);
  }
}

// MASK: y = the forward's relu(x_q); the ReLU's backward (g where y > 0, else 0) is applied on load
template <bool MASK>
__global__ __launch_bounds__(kThreads) void act_quant_bwd_kernel(const float* __restrict__ g,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ y,
                                                                 float* __restrict__ dx, int64_t n, float r, int nts) {
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  const float4* x4 = reinterpret_cast<const float4*>(x);
  float4* d4 = reinterpret_cast<float4*>(dx);
  for (int64_t i0 = (int64_t)blockIdx.x * (kThreads * kUb) + threadIdx.x; i0 < nvec; i0 += kUb * stride) {
    float4 gv[kUb], xv[kUb], yv[kUb];
#pragma unroll
    for (int u = 0; u < kUb; u++) {
      const int64_t i = i0 + u * kThreads, ic = i < nvec ? i : i0;
      gv[u] = ld4_stream(g4 + ic);
      xv[u] = ld4_stream(x4 + ic);
      if (MASK) yv[u] = ld4_stream(reinterpret_cast<const float4*>(y) + ic);
    }
#pragma unroll
    for (int u = 0; u < kUb; u++) {
      const int64_t i = i0 + u * kThreads;
      float4 o;
      if (MASK) {
        gv[u].x = yv[u].x > 0.f ? gv[u].x : 0.f; gv[u].y = yv[u].y > 0.f ? gv[u].y : 0.f;
        gv[u].z = yv[u].z > 0.f ? gv[u].z : 0.f; gv[u].w = yv[u].w > 0.f ? gv[u].w : 0.f;
      }
      o.x = gv[u].x * act_jac(xv[u].x, r);
      o.y = gv[u].y * act_jac(xv[u].y, r);
      o.z = gv[u].z * act_jac(xv[u].z, r);
      o.w = gv[u].w * act_jac(xv[u].w, r);
      if (i < nvec) st4_out(d4 + i, o, nts);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    int64_t i = (nvec << 2) + threadIdx.x;
    dx[i] = ((MASK && !(y[i] > 0.f)) ? 0.f : g[i]) * act_jac(x[i], r);
  }
}

// ------------------------------------------------------------------ block reduction helper ----
// Sums two doubles over the block; result valid in every thread.  LDS: 2*4 doubles.
__device__ __forceinline__ void block_sum2(double& a, double& b, double* sm /* [16] */) {
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) { sm[w] = a; sm[8 + w] = b; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  a = 0; b = 0;
  for (int i = 0; i < nw; i++) { a += sm[i]; b += sm[8 + i]; }
}

// ------------------------------------------------------------------ weights: stats ------------
// pass 1: per-block partial (sum w, sum w^2) in double -> ws[2*blk]
__global__ __launch_bounds__(kThreads) void weight_partial_sums_kernel(const float* __restrict__ w, int64_t n,
                                                                       double* __restrict__ ws) {
  __shared__ double sm[16];
  double s = 0, s2 = 0;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {       // 16-byte loads, two in flight per thread; the tail by block 0
    const int64_t nvec = n >> 2;
    const float4* w4 = reinterpret_cast<const float4*>(w);
    for (int64_t i0 = (int64_t)blockIdx.x * kThreads + threadIdx.x; i0 < nvec; i0 += 2 * stride) {
      const float4 a = w4[i0];
      const bool two = i0 + stride < nvec;
      const float4 b = w4[two ? i0 + stride : i0];
      s += (double)a.x + (double)a.y + (double)a.z + (double)a.w;
      s2 += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
      if (two) {
        s += (double)b.x + (double)b.y + (double)b.z + (double)b.w;
        s2 += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
      }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const double v = w[(nvec << 2) + threadIdx.x]; s += v; s2 += v * v; }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
      double v = w[i];
      s += v;
      s2 += v * v;
    }
  }
  block_sum2(s, s2, sm);
  if (threadIdx.x == 0) { ws[2 * blockIdx.x] = s; ws[2 * blockIdx.x + 1] = s2; }
}

// pass 2 (1 block): combine partials -> ms = {mean, unbiased std}
__global__ __launch_bounds__(kThreads) void weight_finalize_stats_kernel(const double* __restrict__ ws, int nblk,
                                                                         int64_t n, float* __restrict__ ms) {
  __shared__ double sm[16];
  double s = 0, s2 = 0;
  for (int i = threadIdx.x; i < nblk; i += kThreads) { s += ws[2 * i]; s2 += ws[2 * i + 1]; }
  block_sum2(s, s2, sm);
  if (threadIdx.x == 0) {
    double dn = (double)n;
    double mean = s / dn;
    double var = (s2 - s * s / dn) / (dn - 1.0);
    if (var < 0) var = 0;
    ms[0] = (float)mean;
    ms[1] = (float)sqrt(var);
  }
}

// ------------------------------------------------------------------ weights: forward ----------
template <int FORMULA>
__global__ __launch_bounds__(kThreads) void weight_quant_fwd_kernel(const float* __restrict__ w,
                                                                    const float* __restrict__ ms,
                                                                    float* __restrict__ q, float* __restrict__ cdf_out,
                                                                    float* __restrict__ pdf_out,
                                                                    int32_t* __restrict__ bins, int64_t n, int k) {
  __shared__ __attribute__((aligned(16))) float tab_lds[ALIGNQ_NERF_LDS_FLOATS];
  nerf_tab_load(tab_lds);
  __syncthreads();
  const NerfTab tab = nerf_tab(tab_lds);
  const WeightConsts wc = weight_consts(ms[0], ms[1], k);
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const bool vec = ((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(cdf_out) |
                     reinterpret_cast<uintptr_t>(pdf_out) | reinterpret_cast<uintptr_t>(bins)) & 15) == 0;
  int64_t done = 0;
  if (vec) {          // 16-byte accesses (ResNet-50's filters are up to 2.4 M elements: the dword form was issue-bound)
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
      const float4 v = reinterpret_cast<const float4*>(w)[i];
      const float ve[4] = {v.x, v.y, v.z, v.w};
      float qe[4], te[4], be[4], pe[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        qe[e] = weight_quant1<FORMULA>(ve[e], wc, k, &te[e], &be[e], tab);
        if (pdf_out) pe[e] = weight_pdf2(ve[e], wc);
      }
      reinterpret_cast<float4*>(q)[i] = make_float4(qe[0], qe[1], qe[2], qe[3]);
      if (cdf_out) reinterpret_cast<float4*>(cdf_out)[i] = make_float4(te[0], te[1], te[2], te[3]);
      if (bins) reinterpret_cast<int4*>(bins)[i] = make_int4((int)be[0], (int)be[1], (int)be[2], (int)be[3]);
      if (pdf_out) reinterpret_cast<float4*>(pdf_out)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    }
    done = nvec << 2;
  }
  for (int64_t i = done + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float v = w[i];
    float t, b;
    q[i] = weight_quant1<FORMULA>(v, wc, k, &t, &b, tab);
    if (cdf_out) cdf_out[i] = t;
    if (bins) bins[i] = (int)b;
    if (pdf_out) pdf_out[i] = weight_pdf2(v, wc);
  }
}

// ------------------------------------------------------------------ weights: backward ---------
__global__ __launch_bounds__(kThreads) void weight_bwd_partial_kernel(const float* __restrict__ g,
                                                                      const float* __restrict__ w,
                                                                      const float* __restrict__ ms, int64_t n,
                                                                      double* __restrict__ ws) {
  __shared__ double sm[16];
  const float m = ms[0], s = ms[1], rs = 1.0f / s, cs = ALIGNQ_TWO_OVER_SQRT_2PI * rs;
  double s1 = 0, s2 = 0;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  int64_t done = 0;
  if (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g)) & 15) == 0) {
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
      const float4 wv = reinterpret_cast<const float4*>(w)[i], gv = reinterpret_cast<const float4*>(g)[i];
      const float we[4] = {wv.x, wv.y, wv.z, wv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float P, z;
        weight_PZ(we[e], m, rs, cs, &P, &z);
        const double gp = (double)ge[e] * (double)P;
        s1 += gp;
        s2 += gp * (double)z;
      }
    }
    done = nvec << 2;
  }
  for (int64_t i = done + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float P, z;
    weight_PZ(w[i], m, rs, cs, &P, &z);
    double gp = (double)g[i] * (double)P;
    s1 += gp;
    s2 += gp * (double)z;
  }
  block_sum2(s1, s2, sm);
  if (threadIdx.x == 0) { ws[2 * blockIdx.x] = s1; ws[2 * blockIdx.x + 1] = s2; }
}

__global__ __launch_bounds__(kThreads) void weight_bwd_apply_kernel(const float* __restrict__ g,
                                                                    const float* __restrict__ w,
                                                                    const float* __restrict__ ms,
                                                                    const double* __restrict__ ws, int nblk,
                                                                    float* __restrict__ dw, int64_t n) {
  __shared__ double sm[16];
  double s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < nblk; i += kThreads) { s1 += ws[2 * i]; s2 += ws[2 * i + 1]; }
  block_sum2(s1, s2, sm);
  const float m = ms[0], s = ms[1], rs = 1.0f / s, cs = ALIGNQ_TWO_OVER_SQRT_2PI * rs;
  const float mean_gp = (float)(s1 / (double)n);
  const float dotn = (float)(s2 / (double)(n - 1));
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  int64_t done = 0;
  if (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0) {
    const int64_t nvec = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
      const float4 wv = reinterpret_cast<const float4*>(w)[i], gv = reinterpret_cast<const float4*>(g)[i];
      const float we[4] = {wv.x, wv.y, wv.z, wv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float P, z;
        weight_PZ(we[e], m, rs, cs, &P, &z);
        o[e] = ge[e] * P - mean_gp - z * dotn;
      }
      reinterpret_cast<float4*>(dw)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
    done = nvec << 2;
  }
  for (int64_t i = done + (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float P, z;
    weight_PZ(w[i], m, rs, cs, &P, &z);
    dw[i] = g[i] * P - mean_gp - z * dotn;
  }
}


// ------------------------------------------------------------------ N2: integer bin storage ----
// the quantised activation IS an integer level index; stored as int8 / int16 it costs 1-2 B instead of the
// 4 B of the dequantised fp32 value (reference format sites: model/quantization.py:23-31 `round(x*n)/n`, :109-110).
//   ADMM / Office formula: idx = round(t*n) in [-r*n, r*n]  -> SIGNED   int8 while r*n <= 127 (k <= 6 at r = 2), else int16
//                          (8-bit: 1021 levels); value = idx / n
//   CDF-tree formula     : idx = round(c*n) in [0, n]       -> UNSIGNED uint8 for k <= 8, else uint16;
//                          value = (idx / n * 2 - 1) * r
// value(idx) below is the SAME sequence of IEEE operations as round_bins / act_quant1, so a dequantised bin is bit-identical
// to the fp32 x_q the fused quantiser writes.
template <int FORMULA>
__device__ __forceinline__ float bin_value(float b, int k, const Levels& L, float r) {
  float q = b;
  if (k != 1 && k != 32) q = (L.yn != 0.0f) ? div_levels(b, L.n, L.yn) : __fdiv_rn(b, L.n);
  if (FORMULA == 0) return q;
  return __fmul_rn(__fsub_rn(__fmul_rn(q, 2.0f), 1.0f), r);
}

template <typename T> struct Vec4;
template <> struct Vec4<int8_t> { typedef char4 type; };
template <> struct Vec4<uint8_t> { typedef uchar4 type; };
template <> struct Vec4<int16_t> { typedef short4 type; };
template <> struct Vec4<uint16_t> { typedef ushort4 type; };

template <int FORMULA, typename T, bool WITH_XQ>
__global__ __launch_bounds__(kThreads) void act_quant_fwd_packed_kernel(const float* __restrict__ x, float* __restrict__ xq,
                                                                        T* __restrict__ bins, int64_t n, int k, float r,
                                                                        int relu) {
  typedef typename Vec4<T>::type V4;
  __shared__ __attribute__((aligned(16))) float tab_lds[ALIGNQ_NERF_LDS_FLOATS];
  nerf_tab_load(tab_lds);
  __syncthreads();
  const NerfTab tab = nerf_tab(tab_lds);
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  ALIGNQ_BOUNDED_SWITCH(nlev,
  for (int64_t i0 = (int64_t)blockIdx.x * (kThreads * kU) + threadIdx.x; i0 < nvec; i0 += kU * stride) {
    float4 v[kU];
_Pragma("unroll")
    for (int u = 0; u < kU; u++) {
      const int64_t i = i0 + u * kThreads;
      v[u] = ld4_stream(x4 + (i < nvec ? i : i0));
    }
_Pragma("unroll")
    for (int u = 0; u < kU; u++) {
      const int64_t i = i0 + u * kThreads;
      float4 o;
      float t, b0, b1, b2, b3;
      o.x = act_quant1<FORMULA, kBounded>(v[u].x, k, nlev, r, &t, &b0, tab);
      o.y = act_quant1<FORMULA, kBounded>(v[u].y, k, nlev, r, &t, &b1, tab);
      o.z = act_quant1<FORMULA, kBounded>(v[u].z, k, nlev, r, &t, &b2, tab);
      o.w = act_quant1<FORMULA, kBounded>(v[u].w, k, nlev, r, &t, &b3, tab);
      if (i < nvec) {
        V4 bi;
        bi.x = (T)(int)b0; bi.y = (T)(int)b1; bi.z = (T)(int)b2; bi.w = (T)(int)b3;
        reinterpret_cast<V4*>(bins)[i] = bi;
        if (WITH_XQ) {
          if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
          reinterpret_cast<float4*>(xq)[i] = o;
        }
      }
    }
  })
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    float t, b;
    const float q = act_quant1<FORMULA>(x[i], k, nlev, r, &t, &b, tab);
    bins[i] = (T)(int)b;
    if (WITH_XQ) xq[i] = relu ? fmaxf(q, 0.f) : q;
  }
}

template <int FORMULA, typename T>
__global__ __launch_bounds__(kThreads) void bins_dequant_kernel(const T* __restrict__ bins, float* __restrict__ y, int64_t n,
                                                                int k, float r, int relu) {
  typedef typename Vec4<T>::type V4;
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += stride) {
    const V4 b = reinterpret_cast<const V4*>(bins)[i];
    float4 o;
    o.x = bin_value<FORMULA>((float)b.x, k, nlev, r);
    o.y = bin_value<FORMULA>((float)b.y, k, nlev, r);
    o.z = bin_value<FORMULA>((float)b.z, k, nlev, r);
    o.w = bin_value<FORMULA>((float)b.w, k, nlev, r);
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    reinterpret_cast<float4*>(y)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    const float v = bin_value<FORMULA>((float)bins[i], k, nlev, r);
    y[i] = relu ? fmaxf(v, 0.f) : v;
  }
}

// STE backward with the ReLU mask taken from the stored bin (value(idx) > 0) instead of from an fp32 copy of relu(x_q):
// reads g 4 B + x 4 B + idx 1-2 B, writes dx 4 B.  ADMM formula: value > 0 <=> idx > 0; CDF: <=> 2 idx > n (r > 0).
template <int FORMULA, typename T>
__global__ __launch_bounds__(kThreads) void act_quant_bwd_packed_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                        const T* __restrict__ bins, float* __restrict__ dx,
                                                                        int64_t n, int k, float r, int relu) {
  typedef typename Vec4<T>::type V4;
  // value(idx) > 0 as an INTEGER test (exact: ADMM formula value = idx / n; CDF formula value = r (2 idx / n - 1) with n odd,
  // so 2 idx != n and the fp32 quotient cannot round across 1/2): idx > thr with thr = 0 | n / 2 (floor); k == 1: the CDF
  // tree's sign(c) is always 1 (value +r), the ADMM tree's sign(t) is in {-1, 0, 1}
  const int thr = (FORMULA == 0 || k == 1) ? 0 : (((1 << k) - 1) >> 1);
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i0 = (int64_t)blockIdx.x * (kThreads * kUb) + threadIdx.x; i0 < nvec; i0 += kUb * stride) {
    float4 gv[kUb], xv[kUb];
    V4 bv[kUb];
#pragma unroll
    for (int u = 0; u < kUb; u++) {
      const int64_t i = i0 + u * kThreads, ic = i < nvec ? i : i0;
      gv[u] = ld4_stream(reinterpret_cast<const float4*>(g) + ic);
      xv[u] = ld4_stream(reinterpret_cast<const float4*>(x) + ic);
      if (relu) bv[u] = reinterpret_cast<const V4*>(bins)[ic];
    }
#pragma unroll
    for (int u = 0; u < kUb; u++) {
      const int64_t i = i0 + u * kThreads;
      if (relu) {
        gv[u].x = (int)bv[u].x > thr ? gv[u].x : 0.f;
        gv[u].y = (int)bv[u].y > thr ? gv[u].y : 0.f;
        gv[u].z = (int)bv[u].z > thr ? gv[u].z : 0.f;
        gv[u].w = (int)bv[u].w > thr ? gv[u].w : 0.f;
      }
      float4 o;
      o.x = gv[u].x * act_jac(xv[u].x, r); o.y = gv[u].y * act_jac(xv[u].y, r);
      o.z = gv[u].z * act_jac(xv[u].z, r); o.w = gv[u].w * act_jac(xv[u].w, r);
      if (i < nvec) reinterpret_cast<float4*>(dx)[i] = o;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const int64_t i = (nvec << 2) + threadIdx.x;
    const bool keep = !relu || (int)bins[i] > thr;
    dx[i] = (keep ? g[i] : 0.f) * act_jac(x[i], r);
  }
}

inline int bin_bytes_of(int k, float r, int formula) {
  if (!(k >= 1 && k <= 16) || !(r > 0.0f)) return 0;
  const double n = (double)((1 << k) - 1);
  if (formula == ALIGNQ_FORMULA_ADMM) {
    const double m = (k == 1) ? 1.0 : ceil((double)r * n);
    return m <= 127.0 ? 1 : (m <= 32767.0 ? 2 : 0);
  }
  if (formula == ALIGNQ_FORMULA_CDF) return n <= 255.0 ? 1 : 2;
  return 0;
}

// ------------------------------------------------------------------ stand-alone cdf(m, s, src): backward with LIVE (m, s) --------
// cdf.forward (ADMM tree model/quantization.py:49-59, CDF tree :45-50) returns c = kc * Phi(z) [+ const] and pdf = 2 * phi_s with
// z = (x - m) / s, phi_s = N(x; m, s); all three of x, m, s sit in the autograd graph there.  With upstream gradients gc (of c) and gp (of pdf):
//   dx_j = gc_j kc phi_s - gp_j 2 phi_s z / s,   dm = -sum_j dx_j,   ds = sum_j (-gc_j kc phi_s z + gp_j 2 phi_s (z^2 - 1) / s).
// One pass writes dx and per-block partial sums (double), a one-block pass adds them in block order.
__global__ __launch_bounds__(kThreads) void cdf_bwd_kernel(const float* __restrict__ gc, const float* __restrict__ gp,
                                                           const float* __restrict__ x, const float* __restrict__ ms, float kc,
                                                           float* __restrict__ dx, int64_t n, double* __restrict__ ws) {
  __shared__ double sm[16];
  const float m = ms[0], s = ms[1], rs = 1.0f / s, cs = ALIGNQ_TWO_OVER_SQRT_2PI * rs;      // P = 2 phi_s
  double s1 = 0, s2 = 0;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kThreads) {
    float P, z;
    weight_PZ(x[i], m, rs, cs, &P, &z);
    const float a = gc ? gc[i] * (0.5f * kc) * P : 0.f;       // gc kc phi_s
    const float b = gp ? gp[i] * P * rs : 0.f;                // gp 2 phi_s / s
    const float d = a - b * z;
    if (dx) dx[i] = d;
    s1 -= (double)d;
    s2 += (double)b * ((double)z * (double)z - 1.0) - (double)a * (double)z;
  }
  block_sum2(s1, s2, sm);
  if (threadIdx.x == 0) { ws[2 * blockIdx.x] = s1; ws[2 * blockIdx.x + 1] = s2; }
}
__global__ __launch_bounds__(kThreads) void cdf_bwd_finalize_kernel(const double* __restrict__ ws, int nblk, float* __restrict__ dms) {
  __shared__ double sm[16];
  double s1 = 0, s2 = 0;
  for (int i = threadIdx.x; i < nblk; i += kThreads) { s1 += ws[2 * i]; s2 += ws[2 * i + 1]; }
  block_sum2(s1, s2, sm);
  if (threadIdx.x == 0) { dms[0] = (float)s1; dms[1] = (float)s2; }
}

inline int ws_blocks(int64_t n) {
  int64_t b = (n + (int64_t)kThreads * 8 - 1) / ((int64_t)kThreads * 8);
  if (b < 1) b = 1;
  return (int)(b > kWsBlocks ? kWsBlocks : b);
}

}  // namespace

#define LAUNCH_CHECK()                          \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

// tensors of 2^25 elements (128 MB) and more are beyond the caches whatever comes next: their outputs are stored as streams
static inline int stream_out(int64_t n) { return n >= ((int64_t)1 << 25) ? 1 : 0; }

extern "C" {

int alignq_act_quant_fwd(const float* x, float* xq, int32_t* bins, int64_t n, int k, float act_range,
                         int formula, void* stream) {
  if (!x || !xq || n <= 0) return ALIGNQ_EINVAL;
  if (!((k >= 1 && k <= 16) || k == 32)) return ALIGNQ_EINVAL;
  if (formula != ALIGNQ_FORMULA_ADMM && formula != ALIGNQ_FORMULA_CDF) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(bins)) & 15)
    return ALIGNQ_EINVAL;  // 16-byte alignment for the float4 path (torch allocations are 256-B aligned)
  hipStream_t st = (hipStream_t)stream;
  int grid = grid_tiles(n >> 2, kU);
  if (formula == ALIGNQ_FORMULA_ADMM) {
    if (bins) hipLaunchKernelGGL((act_quant_fwd_kernel<0, true>), grid, kThreads, 0, st, x, xq, bins, n, k, act_range, stream_out(n));
    else hipLaunchKernelGGL((act_quant_fwd_kernel<0, false>), grid, kThreads, 0, st, x, xq, bins, n, k, act_range, stream_out(n));
  } else {
    if (bins) hipLaunchKernelGGL((act_quant_fwd_kernel<1, true>), grid, kThreads, 0, st, x, xq, bins, n, k, act_range, stream_out(n));
    else hipLaunchKernelGGL((act_quant_fwd_kernel<1, false>), grid, kThreads, 0, st, x, xq, bins, n, k, act_range, stream_out(n));
  }
  LAUNCH_CHECK();
  return 0;
}

int alignq_uniform_quantize(const float* x, float* y, int64_t n, int k, void* stream) {
  if (!x || !y || n <= 0) return ALIGNQ_EINVAL;
  if (!((k >= 1 && k <= 16) || k == 32)) return ALIGNQ_EINVAL;
  hipLaunchKernelGGL(uniform_quantize_kernel, grid_for(n), kThreads, 0, (hipStream_t)stream, x, y, n, k);
  LAUNCH_CHECK();
  return 0;
}

int alignq_act_quant_bwd(const float* g, const float* x, float* dx, int64_t n, float act_range, void* stream) {
  if (!g || !x || !dx || n <= 0) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dx)) & 15)
    return ALIGNQ_EINVAL;
  hipLaunchKernelGGL(act_quant_bwd_kernel<false>, grid_tiles(n >> 2, kUb), kThreads, 0, (hipStream_t)stream, g, x, nullptr, dx, n, act_range, stream_out(n));
  LAUNCH_CHECK();
  return 0;
}

int alignq_act_quant_relu_fwd(const float* x, float* y, int64_t n, int k, float act_range, int formula, void* stream) {
  if (!x || !y || n <= 0) return ALIGNQ_EINVAL;
  if (!((k >= 1 && k <= 16) || k == 32)) return ALIGNQ_EINVAL;
  if (formula != ALIGNQ_FORMULA_ADMM && formula != ALIGNQ_FORMULA_CDF) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_tiles(n >> 2, kU);
  if (formula == ALIGNQ_FORMULA_ADMM)
    hipLaunchKernelGGL((act_quant_fwd_kernel<0, false, true>), grid, kThreads, 0, st, x, y, nullptr, n, k, act_range, stream_out(n));
  else
    hipLaunchKernelGGL((act_quant_fwd_kernel<1, false, true>), grid, kThreads, 0, st, x, y, nullptr, n, k, act_range, stream_out(n));
  LAUNCH_CHECK();
  return 0;
}

int alignq_act_quant_relu_bwd(const float* g, const float* x, const float* y, float* dx, int64_t n, float act_range,
                              void* stream) {
  if (!g || !x || !y || !dx || n <= 0) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dx) |
       reinterpret_cast<uintptr_t>(y)) & 15)
    return ALIGNQ_EINVAL;
  hipLaunchKernelGGL(act_quant_bwd_kernel<true>, grid_tiles(n >> 2, kUb), kThreads, 0, (hipStream_t)stream, g, x, y, dx, n, act_range, stream_out(n));
  LAUNCH_CHECK();
  return 0;
}


int alignq_bin_bytes(int k, float act_range, int formula) { return bin_bytes_of(k, act_range, formula); }

int alignq_act_quant_fwd_packed(const float* x, float* xq, void* bins, int64_t n, int k, float act_range, int formula,
                                int relu, void* stream) {
  if (!x || !bins || n <= 0) return ALIGNQ_EINVAL;
  const int bb = bin_bytes_of(k, act_range, formula);
  if (bb == 0) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(xq) | reinterpret_cast<uintptr_t>(bins)) & 15)
    return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_tiles(n >> 2, kU);
#define FWDP(F, T)                                                                                                            \
  do {                                                                                                                        \
    if (xq) hipLaunchKernelGGL((act_quant_fwd_packed_kernel<F, T, true>), grid, kThreads, 0, st, x, xq, (T*)bins, n, k,        \
                               act_range, relu);                                                                              \
    else hipLaunchKernelGGL((act_quant_fwd_packed_kernel<F, T, false>), grid, kThreads, 0, st, x, xq, (T*)bins, n, k,          \
                            act_range, relu);                                                                                 \
  } while (0)
  if (formula == ALIGNQ_FORMULA_ADMM) { if (bb == 1) FWDP(0, int8_t); else FWDP(0, int16_t); }
  else { if (bb == 1) FWDP(1, uint8_t); else FWDP(1, uint16_t); }
#undef FWDP
  LAUNCH_CHECK();
  return 0;
}

int alignq_bins_dequant(const void* bins, float* y, int64_t n, int k, float act_range, int formula, int relu, void* stream) {
  if (!bins || !y || n <= 0) return ALIGNQ_EINVAL;
  const int bb = bin_bytes_of(k, act_range, formula);
  if (bb == 0) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(bins)) & 15) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_for(n >> 2);
  if (formula == ALIGNQ_FORMULA_ADMM) {
    if (bb == 1) hipLaunchKernelGGL((bins_dequant_kernel<0, int8_t>), grid, kThreads, 0, st, (const int8_t*)bins, y, n, k, act_range, relu);
    else hipLaunchKernelGGL((bins_dequant_kernel<0, int16_t>), grid, kThreads, 0, st, (const int16_t*)bins, y, n, k, act_range, relu);
  } else {
    if (bb == 1) hipLaunchKernelGGL((bins_dequant_kernel<1, uint8_t>), grid, kThreads, 0, st, (const uint8_t*)bins, y, n, k, act_range, relu);
    else hipLaunchKernelGGL((bins_dequant_kernel<1, uint16_t>), grid, kThreads, 0, st, (const uint16_t*)bins, y, n, k, act_range, relu);
  }
  LAUNCH_CHECK();
  return 0;
}

int alignq_act_quant_bwd_packed(const float* g, const float* x, const void* bins, float* dx, int64_t n, int k,
                                float act_range, int formula, int relu, void* stream) {
  if (!g || !x || !dx || n <= 0 || (relu && !bins)) return ALIGNQ_EINVAL;
  const int bb = bin_bytes_of(k, act_range, formula);
  if (bb == 0) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dx) |
       reinterpret_cast<uintptr_t>(bins)) & 15)
    return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int grid = grid_tiles(n >> 2, kUb);
  if (formula == ALIGNQ_FORMULA_ADMM) {
    if (bb == 1) hipLaunchKernelGGL((act_quant_bwd_packed_kernel<0, int8_t>), grid, kThreads, 0, st, g, x, (const int8_t*)bins, dx, n, k, act_range, relu);
    else hipLaunchKernelGGL((act_quant_bwd_packed_kernel<0, int16_t>), grid, kThreads, 0, st, g, x, (const int16_t*)bins, dx, n, k, act_range, relu);
  } else {
    if (bb == 1) hipLaunchKernelGGL((act_quant_bwd_packed_kernel<1, uint8_t>), grid, kThreads, 0, st, g, x, (const uint8_t*)bins, dx, n, k, act_range, relu);
    else hipLaunchKernelGGL((act_quant_bwd_packed_kernel<1, uint16_t>), grid, kThreads, 0, st, g, x, (const uint16_t*)bins, dx, n, k, act_range, relu);
  }
  LAUNCH_CHECK();
  return 0;
}

size_t alignq_weight_ws_bytes(int64_t n) {
  (void)n;
  return (size_t)kWsBlocks * 2 * sizeof(double);
}

int alignq_weight_stats(const float* w, int64_t n, float* ms, void* ws, void* stream) {
  if (!w || !ms || !ws || n < 2) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int nb = ws_blocks(n);
  hipLaunchKernelGGL(weight_partial_sums_kernel, nb, kThreads, 0, st, w, n, (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(weight_finalize_stats_kernel, 1, kThreads, 0, st, (const double*)ws, nb, n, ms);
  LAUNCH_CHECK();
  return 0;
}

int alignq_weight_quant_fwd(const float* w, const float* ms, float* q, float* cdf_out, float* pdf_out,
                            int32_t* bins, int64_t n, int k, int formula, void* stream) {
  if (!w || !ms || !q || n <= 0) return ALIGNQ_EINVAL;
  if (!((k >= 1 && k <= 16) || k == 32)) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int grid = grid_for((n + 3) >> 2);
  if (formula == ALIGNQ_FORMULA_ADMM)
    hipLaunchKernelGGL((weight_quant_fwd_kernel<0>), grid, kThreads, 0, st, w, ms, q, cdf_out, pdf_out, bins, n, k);
  else if (formula == ALIGNQ_FORMULA_CDF)
    hipLaunchKernelGGL((weight_quant_fwd_kernel<1>), grid, kThreads, 0, st, w, ms, q, cdf_out, pdf_out, bins, n, k);
  else
    return ALIGNQ_EINVAL;
  LAUNCH_CHECK();
  return 0;
}

int alignq_weight_quant_bwd(const float* g, const float* w, const float* ms, float* dw, int64_t n, void* ws,
                            void* stream) {
  if (!g || !w || !ms || !dw || !ws || n < 2) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  int nb = ws_blocks(n);
  hipLaunchKernelGGL(weight_bwd_partial_kernel, nb, kThreads, 0, st, g, w, ms, n, (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(weight_bwd_apply_kernel, grid_for((n + 3) >> 2), kThreads, 0, st, g, w, ms, (const double*)ws, nb, dw, n);
  LAUNCH_CHECK();
  return 0;
}

int alignq_cdf_bwd(const float* gc, const float* gp, const float* x, const float* ms, float kc, float* dx, float* dms, int64_t n,
                   void* ws, void* stream) {
  if ((!gc && !gp) || !x || !ms || !dms || !ws || n < 1) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int nb = ws_blocks(n);
  hipLaunchKernelGGL(cdf_bwd_kernel, nb, kThreads, 0, st, gc, gp, x, ms, kc, dx, n, (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(cdf_bwd_finalize_kernel, 1, kThreads, 0, st, (const double*)ws, nb, dms);
  LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
