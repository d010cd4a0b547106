// multi_tensor_kernels.hip — one launch for ALL weight tensors / parameters of a model (gfx950).
//
// A CIFAR ResNet has 21..57 conv weights of 432..36864 elements and 65..185 parameters: per-tensor launches
// (what the reference does in eager PyTorch, ~20 kernels per tensor) are pure launch latency.  Here the
// per-tensor pointer tables travel BY VALUE in the kernel arguments (<= kChunk tensors per launch, 4 KB
// argument limit), so there are no device-side tables to keep in sync and every launch is hipGraph-capturable;
// blockIdx.y selects the tensor, blockIdx.x strides over its elements.  Every element loop issues kU independent (clamped,
// unconditional) loads per thread before the first use: with one load per iteration a 36864-element filter cost eight
// dependent memory round trips per thread (mt_sgd_kernel: 12.5 us), a 2.4 M-element ResNet-50 filter 144 of them.
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "admm_body.h"
#include "alignq_math.h"

using namespace alignq;

namespace {

constexpr int kThreads = 256;
constexpr int kChunk = 64;      // weight tensors per launch: 64 * (5 pointers + 1 size) * 8 B = 3072 B of arguments
constexpr int kSgdChunk = 72;   // parameters per SGD launch: 72 * (48 + 1) B = 3528 B (ResNet-20's 65 in one launch)
constexpr int kCopyChunk = 128; // tensors per bucket pack / unpack launch: 128 * 24 B = 3072 B
constexpr int kMaxBlk = 256;    // blocks per tensor (partials per tensor in the workspace; <= kThreads)
constexpr int kU = 8;           // loads in flight per thread and pass (= elements per thread at 2048 elements per block)

// element index of pass slot u, clamped for the (unconditional) load; `ok` tells whether the slot is real
#define MT_FOR_ELEMENTS_NT(n, NT)                                                            \
  const long stride__ = (long)gridDim.x * (NT);                                              \
  for (long i0 = (long)blockIdx.x * (NT) + threadIdx.x; i0 < (n); i0 += stride__ * kU)
#define MT_FOR_ELEMENTS(n) MT_FOR_ELEMENTS_NT(n, kThreads)
#define MT_IDX(u) (i0 + (long)(u) * stride__)
#define MT_CLAMP(i, n) ((i) < (n) ? (i) : (n) - 1)

struct WChunk {
  const float* w[kChunk];
  const float* g[kChunk];   // backward only
  float* q[kChunk];         // fwd: W_q         | bwd: dW
  float* cdf[kChunk];
  float* pdf[kChunk];
  long n[kChunk];
};

__global__ __launch_bounds__(kThreads) void mt_weight_partial_kernel(WChunk c, double* __restrict__ ws, int t0) {
  __shared__ double sm[32];
  const int t = blockIdx.y;
  const float* __restrict__ w = c.w[t];
  const long n = c.n[t];
  double s = 0, s2 = 0;
  MT_FOR_ELEMENTS(n) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) v[u] = w[MT_CLAMP(MT_IDX(u), n)];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (MT_IDX(u) < n) {
        const double d = v[u];
        s += d;
        s2 += d * d;
      }
    }
  }
  block_sum2d(s, s2, sm);
  if (threadIdx.x == 0) {
    double* p = ws + ((long)(t0 + t) * kMaxBlk + blockIdx.x) * 2;
    p[0] = s;
    p[1] = s2;
  }
}

template <int FORMULA>
__global__ __launch_bounds__(kThreads) void mt_weight_apply_kernel(WChunk c, const double* __restrict__ ws,
                                                                   float* __restrict__ ms_out, int t0, int k) {
  __shared__ double sm[32];
  __shared__ __attribute__((aligned(16))) float nerf_lds[ALIGNQ_NERF_LDS_FLOATS];
  nerf_tab_load(nerf_lds);                  // block_sum2d's barriers publish it
  const NerfTab tab = nerf_tab(nerf_lds);
  const int t = blockIdx.y;
  const long n = c.n[t];
  double s = 0, s2 = 0;
  if (threadIdx.x < gridDim.x) {
    const double* p = ws + ((long)(t0 + t) * kMaxBlk + threadIdx.x) * 2;
    s = p[0];
    s2 = p[1];
  }
  block_sum2d(s, s2, sm);
  const double dn = (double)n;
  const double mean = s / dn;
  double var = (s2 - s * s / dn) / (dn - 1.0);
  if (var < 0) var = 0;
  const float m = (float)mean, sd = (float)sqrt(var);
  if (blockIdx.x == 0 && threadIdx.x == 0) { ms_out[2 * (t0 + t)] = m; ms_out[2 * (t0 + t) + 1] = sd; }
  const WeightConsts wc = weight_consts(m, sd, k);
  const float* __restrict__ w = c.w[t];
  float* __restrict__ q = c.q[t];
  float* __restrict__ cdf = c.cdf[t];
  float* __restrict__ pdf = c.pdf[t];
  MT_FOR_ELEMENTS(n) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) v[u] = w[MT_CLAMP(MT_IDX(u), n)];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const long i = MT_IDX(u);
      if (i < n) {
        float tt, b;
        q[i] = weight_quant1<FORMULA>(v[u], wc, k, &tt, &b, tab);
        if (cdf) cdf[i] = tt;
        if (pdf) pdf[i] = weight_pdf2(v[u], wc);
      }
    }
  }
}

// Statistics + quantisation in ONE launch for filters of at most kFusedMaxN elements (every CIFAR filter: 432 .. 36864; round 6: the
// two launches above were 12.6 us of the ResNet-20 step's chain, this one 7-8): a workgroup of 1024 threads owns kFusedPerBlk
// consecutive elements of its tensor and computes the WHOLE tensor's sums itself - nine 16-byte loads per thread at most, all in
// flight at once, the same order in every workgroup of the tensor, hence the same mean / std bits in all of them - instead of
// meeting the other workgroups' partial sums in a second launch.  Needs n % 4 == 0 and 16-byte aligned filters (the launcher checks).
constexpr int kFusedThreads = 1024, kFusedPerBlk = 2048, kFusedQuads = 9, kFusedMaxN = kFusedQuads * 4 * kFusedThreads;
template <int FORMULA>
__global__ __launch_bounds__(kFusedThreads) void mt_weight_fused_kernel(WChunk c, float* __restrict__ ms_out, int t0, int k) {
  __shared__ double sm[32];
  __shared__ __attribute__((aligned(16))) float nerf_lds[ALIGNQ_NERF_LDS_FLOATS];
  const int t = blockIdx.y;
  const long n = c.n[t];
  if ((long)blockIdx.x * kFusedPerBlk >= n) return;           // block-uniform: this tensor has fewer workgroups
  nerf_tab_load(nerf_lds);                  // block_sum2d's barriers publish it
  const NerfTab tab = nerf_tab(nerf_lds);
  const float* __restrict__ w = c.w[t];
  const int n4 = (int)(n >> 2);
  float4 v4[kFusedQuads];
#pragma unroll
  for (int u = 0; u < kFusedQuads; u++) {
    const int i = threadIdx.x + u * kFusedThreads;
    v4[u] = reinterpret_cast<const float4*>(w)[i < n4 ? i : n4 - 1];
  }
  double s = 0, s2 = 0;
#pragma unroll
  for (int u = 0; u < kFusedQuads; u++) {
    if (threadIdx.x + u * kFusedThreads < n4) {
      const double d0 = v4[u].x, d1 = v4[u].y, d2 = v4[u].z, d3 = v4[u].w;
      s += d0; s2 += d0 * d0;
      s += d1; s2 += d1 * d1;
      s += d2; s2 += d2 * d2;
      s += d3; s2 += d3 * d3;
    }
  }
  block_sum2d(s, s2, sm);
  const double dn = (double)n;
  const double mean = s / dn;
  double var = (s2 - s * s / dn) / (dn - 1.0);
  if (var < 0) var = 0;
  const float m = (float)mean, sd = (float)sqrt(var);
  if (blockIdx.x == 0 && threadIdx.x == 0) { ms_out[2 * (t0 + t)] = m; ms_out[2 * (t0 + t) + 1] = sd; }
  const WeightConsts wc = weight_consts(m, sd, k);
  float* __restrict__ q = c.q[t];
  float* __restrict__ cdf = c.cdf[t];
  float* __restrict__ pdf = c.pdf[t];
  float v[kFusedPerBlk / kFusedThreads];
#pragma unroll
  for (int u = 0; u < kFusedPerBlk / kFusedThreads; u++) {
    const long i = (long)blockIdx.x * kFusedPerBlk + threadIdx.x + u * kFusedThreads;
    v[u] = w[MT_CLAMP(i, n)];
  }
#pragma unroll
  for (int u = 0; u < kFusedPerBlk / kFusedThreads; u++) {
    const long i = (long)blockIdx.x * kFusedPerBlk + threadIdx.x + u * kFusedThreads;
    if (i < n) {
      float tt, b;
      q[i] = weight_quant1<FORMULA>(v[u], wc, k, &tt, &b, tab);
      if (cdf) cdf[i] = tt;
      if (pdf) pdf[i] = weight_pdf2(v[u], wc);
    }
  }
}

__global__ __launch_bounds__(kThreads) void mt_weight_bwd_partial_kernel(WChunk c, const float* __restrict__ ms,
                                                                         double* __restrict__ ws, int t0) {
  __shared__ double sm[32];
  const int t = blockIdx.y;
  const float* __restrict__ w = c.w[t];
  const float* __restrict__ g = c.g[t];
  const long n = c.n[t];
  const float m = ms[2 * (t0 + t)], s = ms[2 * (t0 + t) + 1], rs = 1.0f / s, cs = ALIGNQ_TWO_OVER_SQRT_2PI * rs;
  double s1 = 0, s2 = 0;
  MT_FOR_ELEMENTS(n) {
    float wv[kU], gv[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const long i = MT_CLAMP(MT_IDX(u), n);
      wv[u] = w[i];
      gv[u] = g[i];
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (MT_IDX(u) < n) {
        float P, z;
        weight_PZ(wv[u], m, rs, cs, &P, &z);
        const double gp = (double)gv[u] * (double)P;
        s1 += gp;
        s2 += gp * (double)z;
      }
    }
  }
  block_sum2d(s1, s2, sm);
  if (threadIdx.x == 0) {
    double* p = ws + ((long)(t0 + t) * kMaxBlk + blockIdx.x) * 2;
    p[0] = s1;
    p[1] = s2;
  }
}

__global__ __launch_bounds__(kThreads) void mt_weight_bwd_apply_kernel(WChunk c, const float* __restrict__ ms,
                                                                       const double* __restrict__ ws, int t0) {
  __shared__ double sm[32];
  const int t = blockIdx.y;
  const long n = c.n[t];
  double s1 = 0, s2 = 0;
  if (threadIdx.x < gridDim.x) {
    const double* p = ws + ((long)(t0 + t) * kMaxBlk + threadIdx.x) * 2;
    s1 = p[0];
    s2 = p[1];
  }
  block_sum2d(s1, s2, sm);
  const float m = ms[2 * (t0 + t)], s = ms[2 * (t0 + t) + 1], rs = 1.0f / s, cs = ALIGNQ_TWO_OVER_SQRT_2PI * rs;
  const float mean_gp = (float)(s1 / (double)n);
  const float dotn = (float)(s2 / (double)(n - 1));
  const float* __restrict__ w = c.w[t];
  const float* __restrict__ g = c.g[t];
  float* __restrict__ dw = c.q[t];
  MT_FOR_ELEMENTS(n) {
    float wv[kU], gv[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const long i = MT_CLAMP(MT_IDX(u), n);
      wv[u] = w[i];
      gv[u] = g[i];
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const long i = MT_IDX(u);
      if (i < n) {
        float P, z;
        weight_PZ(wv[u], m, rs, cs, &P, &z);
        dw[i] = gv[u] * P - mean_gp - z * dotn;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ SGD
struct SChunk {
  float* p[kSgdChunk];
  float* g[kSgdChunk];
  float* buf[kSgdChunk];
  const float* cdf[kSgdChunk];   // non-NULL => tensor is in `idx`: p.grad <- dir * sigmoid_d(transform(cdf)) * pdf
  const float* pdf[kSgdChunk];
  long n[kSgdChunk];
  unsigned char first[kSgdChunk];   // 1 => momentum buffer is being created this step
};

// one parameter tensor's SGD step by the workgroups (blockIdx.x, NT threads each) of its grid row
template <int NT>
__device__ __forceinline__ void sgd_tensor(float* __restrict__ p, float* __restrict__ g, float* __restrict__ buf,
                                           const float* __restrict__ cdf, const float* __restrict__ pdf, long n, bool first,
                                           float lr, float mom, float damp, float wd, int nesterov, float nlev, float lam,
                                           float lam2) {
  const bool use_buf = mom != 0.0f && !first;
  MT_FOR_ELEMENTS_NT(n, NT) {
    float pv[kU], gv[kU], bv[kU], cv[kU], fv[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const long i = MT_CLAMP(MT_IDX(u), n);
      pv[u] = p[i];
      gv[u] = g[i];
      bv[u] = use_buf ? buf[i] : 0.0f;          // block-uniform conditions: no per-load branches
      cv[u] = cdf ? cdf[i] : 0.0f;
      fv[u] = cdf ? pdf[i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < kU; u++) {
      const long i = MT_IDX(u);
      if (i < n) {
        float d = gv[u];
        if (wd != 0.0f) d = __fmaf_rn(wd, pv[u], d);
        float dir = d;
        if (mom != 0.0f) {
          const float b = first ? d : __fmaf_rn(1.0f - damp, d, bv[u] * mom);
          buf[i] = b;
          dir = nesterov ? __fmaf_rn(mom, b, d) : b;
        }
        p[i] = __fmaf_rn(-lr, dir, pv[u]);
        float gout = dir;
        if (cdf) {
          const float a = (cv[u] + 0.5f) * nlev;
          const float fr = a - floorf(a);
          const float tr = fr * lam2 * 2.0f;
          const float sg = 1.0f / (1.0f + __expf(-tr));
          gout = dir * (sg * (1.0f - sg) * lam) * fv[u];
        }
        g[i] = gout;
      }
    }
  }
}

__global__ __launch_bounds__(kThreads) void mt_sgd_kernel(SChunk c, float lr, float mom, float damp, float wd,
                                                          int nesterov, float nlev, float lam, float lam2) {
  const int t = blockIdx.y;
  sgd_tensor<kThreads>(c.p[t], c.g[t], c.buf[t], c.cdf[t], c.pdf[t], c.n[t], c.first[t] != 0, lr, mom, damp, wd, nesterov, nlev,
                       lam, lam2);
}

// The SGD step and the ADMM primal / dual update of one iteration as ROLES of one launch (they touch disjoint tensors: the
// alterD / gamma parameters are not in the SGD list, main.py:256-260): grid rows [0, T) take the parameters, rows [T, T + S) one
// site each (workgroup 0 of the row; 1024 threads, the site in registers).  66 parameters + 22 sites = 3.8 KB of arguments.
constexpr int kSgdA = 66, kSiteA = 22;
struct SAChunk {
  float* p[kSgdA];
  float* g[kSgdA];
  float* buf[kSgdA];
  const float* cdf[kSgdA];
  const float* pdf[kSgdA];
  long n[kSgdA];
  const float* D[kSiteA];
  float* A[kSiteA];
  float* G[kSiteA];
  unsigned char first[kSgdA];
};
__global__ __launch_bounds__(kAdmmThreads) void mt_sgd_admm_kernel(SAChunk c, int T, float lr, float mom, float damp, float wd,
                                                                   int nesterov, float nlev, float lam, float lam2, int b,
                                                                   int dim, float mu, float rho) {
  __shared__ double sm[48];
  const int t = blockIdx.y;
  if (t < T) {
    sgd_tensor<kAdmmThreads>(c.p[t], c.g[t], c.buf[t], c.cdf[t], c.pdf[t], c.n[t], c.first[t] != 0, lr, mom, damp, wd, nesterov,
                             nlev, lam, lam2);
  } else if (blockIdx.x == 0) {
    const int s = t - T;
    admm_update_site(c.D[s], c.A[s], c.G[s], b, dim, mu, rho, sm);
  }
}

inline int blocks_for(long max_n) {
  long b = (max_n + 2047) / 2048;
  if (b < 1) b = 1;
  return (int)(b > kMaxBlk ? kMaxBlk : b);
}

// ------------------------------------------------------------------------------------------ flat-bucket pack / unpack
// Data-parallel bucket (alignq_amd/dp.py): gather T dense tensors into one flat buffer (dir 0) or scatter them back
// (dir 1) in one launch per 128 tensors.  Storage order is copied as it lies (any dense layout: the all-reduce is
// elementwise, every rank uses the same layouts).
struct CChunk {
  float* t[kCopyChunk];
  long off[kCopyChunk];
  long n[kCopyChunk];
};
__global__ __launch_bounds__(kThreads) void mt_copy_kernel(CChunk c, float* __restrict__ flat, int dir) {
  const int t = blockIdx.y;
  float* __restrict__ x = c.t[t];
  float* __restrict__ f = flat + c.off[t];
  const long n = c.n[t];
  const float* __restrict__ src = dir == 0 ? x : f;
  float* __restrict__ dst = dir == 0 ? f : x;
  MT_FOR_ELEMENTS(n) {
    float v[kU];
#pragma unroll
    for (int u = 0; u < kU; u++) v[u] = src[MT_CLAMP(MT_IDX(u), n)];
#pragma unroll
    for (int u = 0; u < kU; u++) {
      if (MT_IDX(u) < n) dst[MT_IDX(u)] = v[u];
    }
  }
}

}  // namespace

#define LAUNCH_CHECK()                          \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

extern "C" {

size_t alignq_weight_multi_ws_bytes(int T) { return (size_t)(T > 0 ? T : 1) * kMaxBlk * 2 * sizeof(double); }

int alignq_weight_quant_fwd_multi(int T, const float* const* w, float* const* q, float* const* cdf_out,
                                  float* const* pdf_out, const int64_t* n, float* ms, int k, int formula, void* ws,
                                  void* stream) {
  if (T <= 0 || !w || !q || !n || !ms || !ws) return ALIGNQ_EINVAL;
  if (!((k >= 1 && k <= 16) || k == 32)) return ALIGNQ_EINVAL;
  if (formula != ALIGNQ_FORMULA_ADMM && formula != ALIGNQ_FORMULA_CDF) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < T; t0 += kChunk) {
    const int cnt = (T - t0 < kChunk) ? T - t0 : kChunk;
    WChunk c;
    long max_n = 0;
    for (int i = 0; i < cnt; i++) {
      if (!w[t0 + i] || !q[t0 + i] || n[t0 + i] < 2) return ALIGNQ_EINVAL;
      c.w[i] = w[t0 + i]; c.g[i] = nullptr; c.q[i] = q[t0 + i];
      c.cdf[i] = cdf_out ? cdf_out[t0 + i] : nullptr;
      c.pdf[i] = pdf_out ? pdf_out[t0 + i] : nullptr;
      c.n[i] = (long)n[t0 + i];
      if (c.n[i] > max_n) max_n = c.n[i];
    }
    bool fused = max_n <= kFusedMaxN;
    for (int i = 0; i < cnt && fused; i++) fused = (c.n[i] & 3) == 0 && (reinterpret_cast<uintptr_t>(c.w[i]) & 15) == 0;
    if (fused) {            // small filters: statistics and quantisation in one launch (mt_weight_fused_kernel)
      dim3 fgrid((unsigned)((max_n + kFusedPerBlk - 1) / kFusedPerBlk), cnt);
      if (formula == ALIGNQ_FORMULA_ADMM)
        hipLaunchKernelGGL((mt_weight_fused_kernel<0>), fgrid, kFusedThreads, 0, st, c, ms, t0, k);
      else
        hipLaunchKernelGGL((mt_weight_fused_kernel<1>), fgrid, kFusedThreads, 0, st, c, ms, t0, k);
      LAUNCH_CHECK();
      continue;
    }
    dim3 grid(blocks_for(max_n), cnt);
    hipLaunchKernelGGL(mt_weight_partial_kernel, grid, kThreads, 0, st, c, (double*)ws, t0);
    LAUNCH_CHECK();
    if (formula == ALIGNQ_FORMULA_ADMM)
      hipLaunchKernelGGL((mt_weight_apply_kernel<0>), grid, kThreads, 0, st, c, (const double*)ws, ms, t0, k);
    else
      hipLaunchKernelGGL((mt_weight_apply_kernel<1>), grid, kThreads, 0, st, c, (const double*)ws, ms, t0, k);
    LAUNCH_CHECK();
  }
  return 0;
}

int alignq_weight_quant_bwd_multi(int T, const float* const* g, const float* const* w, const float* ms,
                                  float* const* dw, const int64_t* n, void* ws, void* stream) {
  if (T <= 0 || !g || !w || !ms || !dw || !n || !ws) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < T; t0 += kChunk) {
    const int cnt = (T - t0 < kChunk) ? T - t0 : kChunk;
    WChunk c;
    long max_n = 0;
    for (int i = 0; i < cnt; i++) {
      if (!w[t0 + i] || !g[t0 + i] || !dw[t0 + i] || n[t0 + i] < 2) return ALIGNQ_EINVAL;
      c.w[i] = w[t0 + i]; c.g[i] = g[t0 + i]; c.q[i] = dw[t0 + i]; c.cdf[i] = nullptr; c.pdf[i] = nullptr;
      c.n[i] = (long)n[t0 + i];
      if (c.n[i] > max_n) max_n = c.n[i];
    }
    dim3 grid(blocks_for(max_n), cnt);
    hipLaunchKernelGGL(mt_weight_bwd_partial_kernel, grid, kThreads, 0, st, c, ms, (double*)ws, t0);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(mt_weight_bwd_apply_kernel, grid, kThreads, 0, st, c, ms, (const double*)ws, t0);
    LAUNCH_CHECK();
  }
  return 0;
}

int alignq_sgd_step_multi(int T, float* const* p, float* const* g, float* const* buf, const int64_t* n,
                          const float* const* w_cdf, const float* const* w_pdf, const int32_t* first, float lr,
                          float mom, float damp, float wd, int nesterov, int bitW, float lam, float lam2,
                          void* stream) {
  if (T <= 0 || !p || !g || !n) return ALIGNQ_EINVAL;
  if (mom != 0.0f && !buf) return ALIGNQ_EINVAL;
  if (bitW < 1 || bitW > 30) bitW = 1;
  const float nlev = (float)((1 << bitW) - 1);
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < T; t0 += kSgdChunk) {
    const int cnt = (T - t0 < kSgdChunk) ? T - t0 : kSgdChunk;
    SChunk c;
    long max_n = 0;
    for (int i = 0; i < cnt; i++) {
      if (!p[t0 + i] || !g[t0 + i] || n[t0 + i] <= 0) return ALIGNQ_EINVAL;
      if (mom != 0.0f && !buf[t0 + i]) return ALIGNQ_EINVAL;
      c.p[i] = p[t0 + i]; c.g[i] = g[t0 + i]; c.buf[i] = buf ? buf[t0 + i] : nullptr;
      c.cdf[i] = (w_cdf && w_pdf && w_cdf[t0 + i] && w_pdf[t0 + i]) ? w_cdf[t0 + i] : nullptr;
      c.pdf[i] = c.cdf[i] ? w_pdf[t0 + i] : nullptr;
      c.n[i] = (long)n[t0 + i];
      c.first[i] = (first && first[t0 + i]) ? 1 : 0;
      if (c.n[i] > max_n) max_n = c.n[i];
    }
    dim3 grid(blocks_for(max_n), cnt);
    hipLaunchKernelGGL(mt_sgd_kernel, grid, kThreads, 0, st, c, lr, mom, damp, wd, nesterov, nlev, lam, lam2);
    LAUNCH_CHECK();
  }
  return 0;
}

int alignq_sgd_admm_step_multi(int T, float* const* p, float* const* g, float* const* buf, const int64_t* n,
                               const float* const* w_cdf, const float* const* w_pdf, const int32_t* first, float lr, float mom,
                               float damp, float wd, int nesterov, int bitW, float lam, float lam2, int S,
                               const float* const* D_tab, float* const* alterD_tab, float* const* gamma_tab, int b, int dim,
                               float mu, float rho, void* stream) {
  if (T <= 0 || !p || !g || !n || S <= 0 || !D_tab || !alterD_tab || !gamma_tab || b <= 0 || dim < b) return ALIGNQ_EINVAL;
  if (mom != 0.0f && !buf) return ALIGNQ_EINVAL;
  if (dim > 4096) return ALIGNQ_EUNSUPPORTED;
  if (T > kSgdA || S > kSiteA) {       // more tensors than one argument block holds: the two launches of the separate entry points
    if (int rc = alignq_sgd_step_multi(T, p, g, buf, n, w_cdf, w_pdf, first, lr, mom, damp, wd, nesterov, bitW, lam, lam2, stream))
      return rc;
    return alignq_admm_update(D_tab, alterD_tab, gamma_tab, S, b, dim, mu, rho, stream);
  }
  if (bitW < 1 || bitW > 30) bitW = 1;
  const float nlev = (float)((1 << bitW) - 1);
  SAChunk c;
  long max_n = 0;
  for (int i = 0; i < T; i++) {
    if (!p[i] || !g[i] || n[i] <= 0) return ALIGNQ_EINVAL;
    if (mom != 0.0f && !buf[i]) return ALIGNQ_EINVAL;
    c.p[i] = p[i]; c.g[i] = g[i]; c.buf[i] = buf ? buf[i] : nullptr;
    c.cdf[i] = (w_cdf && w_pdf && w_cdf[i] && w_pdf[i]) ? w_cdf[i] : nullptr;
    c.pdf[i] = c.cdf[i] ? w_pdf[i] : nullptr;
    c.n[i] = (long)n[i];
    c.first[i] = (first && first[i]) ? 1 : 0;
    if (c.n[i] > max_n) max_n = c.n[i];
  }
  for (int i = 0; i < S; i++) {
    if (!D_tab[i] || !alterD_tab[i] || !gamma_tab[i]) return ALIGNQ_EINVAL;
    c.D[i] = D_tab[i]; c.A[i] = alterD_tab[i]; c.G[i] = gamma_tab[i];
  }
  long bx = (max_n + (long)kAdmmThreads * kU - 1) / ((long)kAdmmThreads * kU);
  if (bx < 1) bx = 1;
  if (bx > kMaxBlk) bx = kMaxBlk;
  hipLaunchKernelGGL(mt_sgd_admm_kernel, dim3((unsigned)bx, T + S), kAdmmThreads, 0, (hipStream_t)stream, c, T, lr, mom, damp, wd,
                     nesterov, nlev, lam, lam2, b, dim, mu, rho);
  LAUNCH_CHECK();
  return 0;
}

int alignq_bucket_copy_multi(int T, float* const* tensors, const int64_t* n, float* flat, int unpack, void* stream) {
  if (T <= 0 || !tensors || !n || !flat) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  long off = 0;
  for (int t0 = 0; t0 < T; t0 += kCopyChunk) {
    const int cnt = (T - t0 < kCopyChunk) ? T - t0 : kCopyChunk;
    CChunk c;
    long max_n = 0;
    for (int i = 0; i < cnt; i++) {
      if (!tensors[t0 + i] || n[t0 + i] <= 0) return ALIGNQ_EINVAL;
      c.t[i] = tensors[t0 + i]; c.off[i] = off; c.n[i] = (long)n[t0 + i];
      off += c.n[i];
      if (c.n[i] > max_n) max_n = c.n[i];
    }
    dim3 grid(blocks_for(max_n), cnt);
    hipLaunchKernelGGL(mt_copy_kernel, grid, kThreads, 0, st, c, flat, unpack ? 1 : 0);
    LAUNCH_CHECK();
  }
  return 0;
}

// ---- ordering an eagerly enqueued collective behind a node INSIDE a replayed HIP graph (round 6, include/alignq.h) ----------------
namespace {
__global__ void dp_bump_kernel(unsigned* counter) { *counter += 1u; }
__global__ void dp_publish_kernel(unsigned* flag, const unsigned* counter) {
  __threadfence_system();
  *flag = *counter;
}
}  // namespace

int alignq_dp_counter_bump(uint32_t* counter, void* stream) {
  if (!counter) return ALIGNQ_EINVAL;
  hipLaunchKernelGGL(dp_bump_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter);
  LAUNCH_CHECK();
  return 0;
}

int alignq_dp_flag_publish(uint32_t* flag, const uint32_t* counter, void* stream) {
  if (!flag || !counter) return ALIGNQ_EINVAL;
  hipLaunchKernelGGL(dp_publish_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, counter);
  LAUNCH_CHECK();
  return 0;
}

int alignq_dp_stream_wait_ge(void* stream, uint32_t* flag, uint32_t value) {
  if (!flag) return ALIGNQ_EINVAL;
  int can = 0, dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess || !can)
    return ALIGNQ_EUNSUPPORTED;
  const hipError_t e = hipStreamWaitValue32((hipStream_t)stream, flag, value, hipStreamWaitValueGte, 0xFFFFFFFFu);
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
