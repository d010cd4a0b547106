// admm_sgd_kernels.hip — ADMM loss (+grads), batched ADMM primal/dual update, SGD step kernels (gfx950).
//
// All of these touch a few KiB..MiB per step and are launch-latency bound; the design goal is ONE launch
// per logical operation (the reference issues ~10 eager kernels per site for the loss and a Python loop
// with list.index() searches for the update) and determinism (double tree sums).
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "admm_body.h"
#include "alignq_math.h"

using namespace alignq;

namespace {

constexpr int kBig = kAdmmThreads;  // one workgroup of 16 waves per site

// utils/admm.py:24-33 and its autograd:  loss = mu*mean|A| + rho/2*sqrt(mean (D-A)^2) + mean(gamma*|D-A|)
__global__ __launch_bounds__(kBig) void admm_loss_kernel(const float* __restrict__ D, int b,
                                                         const float* __restrict__ A,
                                                         const float* __restrict__ gamma, int dim, float mu,
                                                         float rho, float* __restrict__ loss,
                                                         float* __restrict__ dD, float* __restrict__ dA,
                                                         float* __restrict__ dgamma) {
  __shared__ double sm[48];
  const int nn = b * b;
  double sabs = 0, ssq = 0, srel = 0;
  for (int e = threadIdx.x; e < nn; e += kBig) {
    int i = e / b, j = e - i * b;
    float a = A[i * dim + j];
    float d = D[e] - a;
    sabs += fabsf(a);
    ssq += (double)d * (double)d;
    srel += (double)gamma[i * dim + j] * fabsf(d);
  }
  block_sum3(sabs, ssq, srel, sm);
  const double n = (double)nn;
  const double rms = sqrt(ssq / n);
  if (threadIdx.x == 0) *loss = (float)(mu * sabs / n + 0.5 * rho * rms + srel / n);
  const float c_con = (float)(0.5 * rho / (n * rms));
  const float inv_n = (float)(1.0 / n);
  // gradients; entries of dA / dgamma outside the [:b,:b] slice are zero
  const int full = dim * dim;
  for (int e = threadIdx.x; e < full; e += kBig) {
    int i = e / dim, j = e - i * dim;
    float ga = 0.0f, gg = 0.0f;
    if (i < b && j < b) {
      float a = A[e], gm = gamma[e];
      float d = D[i * b + j] - a;
      float sg = (float)((d > 0.0f) - (d < 0.0f));
      float sa = (float)((a > 0.0f) - (a < 0.0f));
      float gD = c_con * d + gm * sg * inv_n;
      if (dD) dD[i * b + j] = gD;
      ga = mu * sa * inv_n - gD;
      gg = fabsf(d) * inv_n;
    }
    if (dA) dA[e] = ga;
    if (dgamma) dgamma[e] = gg;
  }
}

// The same above 128 rows (round 4: b up to ALIGNQ_MAX_CORR_BATCH = 1024 is a million elements - the one-workgroup kernel took 69 us at
// b = 256): kLossBlocks workgroups leave partial sums (double, fixed order), then every workgroup of the gradient pass adds them
// up for itself (3 x kLossBlocks doubles), workgroup 0 writes the loss.
constexpr int kLossBlocks = 128, kLossThreads = 256;

__global__ __launch_bounds__(kLossThreads) void admm_loss_partial_kernel(const float* __restrict__ D, int b,
                                                                         const float* __restrict__ A,
                                                                         const float* __restrict__ gamma, int dim,
                                                                         double* __restrict__ part) {
  __shared__ double sm[48];
  const int nn = b * b;
  double sabs = 0, ssq = 0, srel = 0;
  for (int e = blockIdx.x * kLossThreads + threadIdx.x; e < nn; e += kLossBlocks * kLossThreads) {
    const int i = e / b, j = e - i * b;
    const float a = A[i * dim + j];
    const float d = D[e] - a;
    sabs += fabsf(a);
    ssq += (double)d * (double)d;
    srel += (double)gamma[i * dim + j] * fabsf(d);
  }
  block_sum3(sabs, ssq, srel, sm);
  if (threadIdx.x == 0) { part[3 * blockIdx.x] = sabs; part[3 * blockIdx.x + 1] = ssq; part[3 * blockIdx.x + 2] = srel; }
}

__global__ __launch_bounds__(kLossThreads) void admm_loss_grad_kernel(const float* __restrict__ D, int b,
                                                                      const float* __restrict__ A,
                                                                      const float* __restrict__ gamma, int dim, float mu,
                                                                      float rho, const double* __restrict__ part,
                                                                      float* __restrict__ loss, float* __restrict__ dD,
                                                                      float* __restrict__ dA, float* __restrict__ dgamma) {
  __shared__ double sm[48];
  double sabs = 0, ssq = 0, srel = 0;
  if (threadIdx.x < kLossBlocks) { sabs = part[3 * threadIdx.x]; ssq = part[3 * threadIdx.x + 1]; srel = part[3 * threadIdx.x + 2]; }
  block_sum3(sabs, ssq, srel, sm);              // (the same order in every workgroup)
  const double n = (double)b * (double)b;
  const double rms = sqrt(ssq / n);
  if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(mu * sabs / n + 0.5 * rho * rms + srel / n);
  const float c_con = (float)(0.5 * rho / (n * rms));
  const float inv_n = (float)(1.0 / n);
  const int full = dim * dim;
  for (int e = blockIdx.x * kLossThreads + threadIdx.x; e < full; e += gridDim.x * kLossThreads) {
    const int i = e / dim, j = e - i * dim;
    float ga = 0.0f, gg = 0.0f;
    if (i < b && j < b) {
      const float a = A[e], gm = gamma[e];
      const float d = D[i * b + j] - a;
      const float sg = (float)((d > 0.0f) - (d < 0.0f));
      const float sa = (float)((a > 0.0f) - (a < 0.0f));
      const float gD = c_con * d + gm * sg * inv_n;
      if (dD) dD[i * b + j] = gD;
      ga = mu * sa * inv_n - gD;
      gg = fabsf(d) * inv_n;
    }
    if (dA) dA[e] = ga;
    if (dgamma) dgamma[e] = gg;
  }
}

// utils/optimizer.py:97-124, one workgroup per site; the per-site pointers travel by value in the arguments.
constexpr int kSiteChunk = 64;
struct AChunk {
  const float* D[kSiteChunk];
  float* A[kSiteChunk];
  float* G[kSiteChunk];
};

__global__ __launch_bounds__(kBig) void admm_update_kernel(AChunk c, int b, int dim, float mu, float rho) {
  __shared__ double sm[48];
  admm_update_site(c.D[blockIdx.x], c.A[blockIdx.x], c.G[blockIdx.x], b, dim, mu, rho, sm);
}

// dim > 128 (the exact-global correlation's ADMM(dim = B_g)): one workgroup per site walks dim^2 elements with
// one load in flight per thread - 210 us at dim = 512, 1.0 ms at 1024 for 21 sites (as long as a whole ResNet-20 step).  Two
// launches over kUpdBlocks workgroups per site instead: partial sums of |V|_F^2, then every workgroup adds the site's partials
// in a fixed order and updates its slice.
constexpr int kUpdBlocks = 64;
constexpr int kUpdThreads = 256;
constexpr int kUpdIn = 8;                  // elements in flight per thread

__global__ __launch_bounds__(kUpdThreads) void admm_update_partial_kernel(AChunk c, int b, int dim, float rho, double* __restrict__ parts) {
  __shared__ double sm[kUpdThreads / 64];
  const float* __restrict__ D = c.D[blockIdx.y];
  const float* __restrict__ G = c.G[blockIdx.y];
  const int full = dim * dim;
  const float inv_rho = 1.0f / rho;
  double ss = 0;
  for (int e0 = (int)blockIdx.x * kUpdThreads + threadIdx.x; e0 < full; e0 += kUpdBlocks * kUpdThreads * kUpdIn) {
    float dv[kUpdIn], gv[kUpdIn];
#pragma unroll
    for (int u = 0; u < kUpdIn; u++) {       // clamped, unconditional loads: all in flight together
      const int e = e0 + u * kUpdBlocks * kUpdThreads;
      const int ec = e < full ? e : full - 1;
      const int i = ec / dim, j = ec - i * dim;
      const bool in = i < b && j < b;
      const float d = D[in ? i * b + j : 0];
      dv[u] = in ? d : 0.0f;
      gv[u] = G[ec];
    }
#pragma unroll
    for (int u = 0; u < kUpdIn; u++) {
      if (e0 + u * kUpdBlocks * kUpdThreads < full) {
        const float v = dv[u] + inv_rho * gv[u];
        ss += (double)v * (double)v;
      }
    }
  }
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kUpdThreads / 64; w++) t += sm[w];
    parts[(size_t)blockIdx.y * kUpdBlocks + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kUpdThreads) void admm_update_apply_kernel(AChunk c, int b, int dim, float mu, float rho,
                                                                      const double* __restrict__ parts) {
  __shared__ double sm[kUpdBlocks];
  __shared__ float shrink_s;
  const float* __restrict__ D = c.D[blockIdx.y];
  float* __restrict__ A = c.A[blockIdx.y];
  float* __restrict__ G = c.G[blockIdx.y];
  if (threadIdx.x < kUpdBlocks) sm[threadIdx.x] = parts[(size_t)blockIdx.y * kUpdBlocks + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    double ss = 0;
    for (int k = 0; k < kUpdBlocks; k++) ss += sm[k];          // fixed order: every workgroup of the site forms the same bits
    const float nv = (float)sqrt(ss), thr = mu / rho;
    shrink_s = (nv > thr) ? (1.0f - thr / nv) : 0.0f;
  }
  __syncthreads();
  const float shrink = shrink_s, inv_rho = 1.0f / rho;
  const int full = dim * dim;
  for (int e0 = (int)blockIdx.x * kUpdThreads + threadIdx.x; e0 < full; e0 += kUpdBlocks * kUpdThreads * kUpdIn) {
    float dv[kUpdIn], gv[kUpdIn];
#pragma unroll
    for (int u = 0; u < kUpdIn; u++) {
      const int e = e0 + u * kUpdBlocks * kUpdThreads;
      const int ec = e < full ? e : full - 1;
      const int i = ec / dim, j = ec - i * dim;
      const bool in = i < b && j < b;
      const float d = D[in ? i * b + j : 0];
      dv[u] = in ? d : 0.0f;
      gv[u] = G[ec];
    }
#pragma unroll
    for (int u = 0; u < kUpdIn; u++) {
      const int e = e0 + u * kUpdBlocks * kUpdThreads;
      if (e < full) {
        const float a = shrink * (dv[u] + inv_rho * gv[u]);
        A[e] = a;
        G[e] = gv[u] + rho * (dv[u] - a);
      }
    }
  }
}

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void sgd_step_kernel(float* __restrict__ p, float* __restrict__ g,
                                                            float* __restrict__ buf, int64_t n, float lr,
                                                            float mom, float damp, float wd, int nesterov,
                                                            int first) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float pv = p[i];
    float d = g[i];
    if (wd != 0.0f) d = __fmaf_rn(wd, pv, d);
    float dir = d;
    if (mom != 0.0f) {
      float bv = first ? d : __fmaf_rn(1.0f - damp, d, buf[i] * mom);
      buf[i] = bv;
      dir = nesterov ? __fmaf_rn(mom, bv, d) : bv;
    }
    p[i] = __fmaf_rn(-lr, dir, pv);
    g[i] = dir;
  }
}

// utils/optimizer.py:6-13: transform(w) = (((w+0.5)*(2^bitW-1)) % 1)*lam2*2 ; sigmoid_d = s(1-s)*lam
__global__ __launch_bounds__(kThreads) void sgd_grad_approx_kernel(const float* __restrict__ dir,
                                                                   const float* __restrict__ w_cdf,
                                                                   const float* __restrict__ w_pdf,
                                                                   float* __restrict__ gout, int64_t n, float nlev,
                                                                   float lam, float lam2) {
  const int64_t stride = (int64_t)gridDim.x * kThreads;
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n; i += stride) {
    float a = (w_cdf[i] + 0.5f) * nlev;
    float fr = a - floorf(a);
    float tr = fr * lam2 * 2.0f;
    float sg = 1.0f / (1.0f + __expf(-tr));
    gout[i] = dir[i] * (sg * (1.0f - sg) * lam) * w_pdf[i];
  }
}

inline int grid_for(int64_t n) {
  int64_t b = (n + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  return (int)(b > 2048 ? 2048 : b);
}

}  // namespace

#define LAUNCH_CHECK()                          \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

extern "C" {

size_t alignq_admm_ws_bytes(int dim) {
  return dim > 128 ? (size_t)kLossBlocks * 3 * sizeof(double) : 16;     // the partial sums of the many-workgroup form
}

int alignq_admm_loss(const float* D, int b, const float* alterD, const float* gamma, int dim, float mu, float rho,
                     float* loss, float* dD, float* dalterD, float* dgamma, void* ws, void* stream) {
  if (!D || !alterD || !gamma || !loss || b <= 0 || dim < b) return ALIGNQ_EINVAL;
  if (dim > 4096) return ALIGNQ_EUNSUPPORTED;
  if (ws && dim > 128) {        // many workgroups (ws: alignq_admm_ws_bytes(dim)); without a workspace the one-workgroup kernel
    double* part = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(admm_loss_partial_kernel, kLossBlocks, kLossThreads, 0, (hipStream_t)stream, D, b, alterD, gamma, dim, part);
    int gb = (dim * dim + kLossThreads * 4 - 1) / (kLossThreads * 4);
    if (gb > 1024) gb = 1024;
    hipLaunchKernelGGL(admm_loss_grad_kernel, gb, kLossThreads, 0, (hipStream_t)stream, D, b, alterD, gamma, dim, mu, rho,
                       (const double*)part, loss, dD, dalterD, dgamma);
    LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(admm_loss_kernel, 1, kBig, 0, (hipStream_t)stream, D, b, alterD, gamma, dim, mu, rho, loss, dD,
                     dalterD, dgamma);
  LAUNCH_CHECK();
  return 0;
}

int alignq_admm_update(const float* const* D_tab, float* const* alterD_tab, float* const* gamma_tab, int S, int b,
                       int dim, float mu, float rho, void* stream) {
  if (!D_tab || !alterD_tab || !gamma_tab || S <= 0 || b <= 0 || dim < b) return ALIGNQ_EINVAL;
  if (dim > 4096) return ALIGNQ_EUNSUPPORTED;
  for (int s0 = 0; s0 < S; s0 += kSiteChunk) {
    const int cnt = (S - s0 < kSiteChunk) ? S - s0 : kSiteChunk;
    AChunk c;
    for (int i = 0; i < cnt; i++) {
      if (!D_tab[s0 + i] || !alterD_tab[s0 + i] || !gamma_tab[s0 + i]) return ALIGNQ_EINVAL;
      c.D[i] = D_tab[s0 + i]; c.A[i] = alterD_tab[s0 + i]; c.G[i] = gamma_tab[s0 + i];
    }
    hipLaunchKernelGGL(admm_update_kernel, cnt, kBig, 0, (hipStream_t)stream, c, b, dim, mu, rho);
    LAUNCH_CHECK();
  }
  return 0;
}

size_t alignq_admm_update_ws_bytes(int S, int dim) {
  return (S > 0 && dim > 128) ? (size_t)S * kUpdBlocks * sizeof(double) : 16;
}

int alignq_admm_update_ws(const float* const* D_tab, float* const* alterD_tab, float* const* gamma_tab, int S, int b, int dim,
                          float mu, float rho, void* ws, void* stream) {
  if (!ws || dim <= 128) return alignq_admm_update(D_tab, alterD_tab, gamma_tab, S, b, dim, mu, rho, stream);
  if (!D_tab || !alterD_tab || !gamma_tab || S <= 0 || b <= 0 || dim < b) return ALIGNQ_EINVAL;
  if (dim > 4096) return ALIGNQ_EUNSUPPORTED;
  double* parts = (double*)ws;
  for (int s0 = 0; s0 < S; s0 += kSiteChunk) {
    const int cnt = (S - s0 < kSiteChunk) ? S - s0 : kSiteChunk;
    AChunk c;
    for (int i = 0; i < cnt; i++) {
      if (!D_tab[s0 + i] || !alterD_tab[s0 + i] || !gamma_tab[s0 + i]) return ALIGNQ_EINVAL;
      c.D[i] = D_tab[s0 + i]; c.A[i] = alterD_tab[s0 + i]; c.G[i] = gamma_tab[s0 + i];
    }
    hipLaunchKernelGGL(admm_update_partial_kernel, dim3(kUpdBlocks, cnt), kUpdThreads, 0, (hipStream_t)stream, c, b, dim, rho,
                       parts + (size_t)s0 * kUpdBlocks);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(admm_update_apply_kernel, dim3(kUpdBlocks, cnt), kUpdThreads, 0, (hipStream_t)stream, c, b, dim, mu, rho,
                       (const double*)(parts + (size_t)s0 * kUpdBlocks));
    LAUNCH_CHECK();
  }
  return 0;
}

int alignq_sgd_step(float* p, float* g, float* buf, int64_t n, float lr, float mom, float damp, float wd,
                    int nesterov, int first, void* stream) {
  if (!p || !g || n <= 0) return ALIGNQ_EINVAL;
  if (mom != 0.0f && !buf) return ALIGNQ_EINVAL;
  hipLaunchKernelGGL(sgd_step_kernel, grid_for(n), kThreads, 0, (hipStream_t)stream, p, g, buf, n, lr, mom, damp, wd,
                     nesterov, first);
  LAUNCH_CHECK();
  return 0;
}

int alignq_sgd_grad_approx(const float* dir, const float* w_cdf, const float* w_pdf, float* grad_out, int64_t n,
                           int bitW, float lam, float lam2, void* stream) {
  if (!dir || !w_cdf || !w_pdf || !grad_out || n <= 0 || bitW < 1 || bitW > 30) return ALIGNQ_EINVAL;
  float nlev = (float)((1 << bitW) - 1);
  hipLaunchKernelGGL(sgd_grad_approx_kernel, grid_for(n), kThreads, 0, (hipStream_t)stream, dir, w_cdf, w_pdf,
                     grad_out, n, nlev, lam, lam2);
  LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
