// corr_xy_kernels.hip — the GENERAL corr(x, y) of the reference (ADMM tree model/quantization.py:134-137; Office tree
// :158-161 with +1e-5): G = Xh Yh^T / F with Xh = (x - mean_b x)/(std_b x + eps), Yh likewise, x and y two different [B,F]
// matrices, 2 <= B <= 128.  No BASELINE configuration calls corr with y != x (every call site is the SYRK corr(v, v) that the
// fused site kernels serve), so this is the plain, exact-fp32 VALU form: correct, coalesced and deterministic, not tuned.
//
//   forward : one workgroup per 32-feature tile (grid-stride): both tiles are standardised in LDS, every thread keeps an
//             8x8 block of the [128,128] product in registers over all its tiles, one partial slab per workgroup, reduced
//             in fixed order by corr_xy_reduce_kernel (x 1/F).
//   backward: dXh = dG Yh / F, dYh = dG^T Xh / F, each followed by the standardisation backward
//             dv = rho (dVh - mean_b dVh - vh sum_b(dVh vh)/(B-1) kappa), kappa = (sd+eps)/sd (0 where sd == 0, torch's
//             std backward); one launch per requested gradient.
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"

namespace {

constexpr int kTF = 32;          // features per tile
constexpr int kLD = kTF + 1;     // LDS row stride (floats): conflict-free column walks
constexpr int kNT = 256;
constexpr int kBP = 128;

__device__ __forceinline__ void tile_load(const float* __restrict__ v, int B, int64_t F, int col0, float* __restrict__ Vs) {
  // thread -> (row group of 8 rows apart, column): 32 lanes read 128 contiguous bytes of one row
  const int c = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const bool cok = (col0 + c) < F;
  for (int row = rg; row < kBP; row += 8) Vs[row * kLD + c] = (row < B && cok) ? v[(int64_t)row * F + col0 + c] : 0.0f;
}

// per-column mean and 1/(std+eps) of a raw tile (unbiased std, two-pass); threads 0..31 of `base` own one column each
__device__ __forceinline__ void tile_stats(const float* __restrict__ Vs, int B, float eps, float* __restrict__ mean_out,
                                           float* __restrict__ rho_out, int c) {
  float s = 0.f;
  for (int b = 0; b < B; b++) s += Vs[b * kLD + c];
  const float m = s / (float)B;
  float q = 0.f;
  for (int b = 0; b < B; b++) { const float d = Vs[b * kLD + c] - m; q += d * d; }
  const float sd = sqrtf(q / (float)(B - 1));
  *mean_out = m;
  *rho_out = 1.0f / (sd + eps);
}

__global__ __launch_bounds__(kNT) void corr_xy_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, int B,
                                                          int64_t F, float eps, float* __restrict__ slabs,
                                                          float* __restrict__ stats, int n_tiles) {
  __shared__ float Xs[kBP * kLD], Ys[kBP * kLD], colv[4 * kTF];
  const int tid = threadIdx.x;
  const int ti = tid >> 4, tj = tid & 15;          // output rows ti + 16a, columns tj + 16b (a, b < 8)
  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; a++)
#pragma unroll
    for (int b = 0; b < 8; b++) acc[a][b] = 0.f;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int col0 = tile * kTF;
    tile_load(x, B, F, col0, Xs);
    tile_load(y, B, F, col0, Ys);
    __syncthreads();
    if (tid < 2 * kTF) {
      const int op = tid >> 5, c = tid & 31;
      float m, rho;
      tile_stats(op ? Ys : Xs, B, eps, &m, &rho, c);
      colv[(2 * op) * kTF + c] = m;
      colv[(2 * op + 1) * kTF + c] = rho;
      if (stats && col0 + c < F) {
        stats[(int64_t)(2 * op) * F + col0 + c] = m;
        stats[(int64_t)(2 * op + 1) * F + col0 + c] = rho;
      }
    }
    __syncthreads();
    {
      const int c = tid & 31, rg = tid >> 5;
      const bool cok = (col0 + c) < F;
      const float mx = colv[c], rx = colv[kTF + c], my = colv[2 * kTF + c], ry = colv[3 * kTF + c];
      for (int row = rg; row < kBP; row += 8) {
        const bool ok = cok && row < B;
        Xs[row * kLD + c] = ok ? (Xs[row * kLD + c] - mx) * rx : 0.0f;
        Ys[row * kLD + c] = ok ? (Ys[row * kLD + c] - my) * ry : 0.0f;
      }
    }
    __syncthreads();
    for (int f = 0; f < kTF; f++) {
      float xa[8], yb[8];
#pragma unroll
      for (int a = 0; a < 8; a++) xa[a] = Xs[(ti + 16 * a) * kLD + f];
#pragma unroll
      for (int b = 0; b < 8; b++) yb[b] = Ys[(tj + 16 * b) * kLD + f];
#pragma unroll
      for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < 8; b++) acc[a][b] = fmaf(xa[a], yb[b], acc[a][b]);
    }
    __syncthreads();
  }
  float* slab = slabs + (int64_t)blockIdx.x * kBP * kBP;
#pragma unroll
  for (int a = 0; a < 8; a++)
#pragma unroll
    for (int b = 0; b < 8; b++) slab[(ti + 16 * a) * kBP + tj + 16 * b] = acc[a][b];
}

__global__ __launch_bounds__(256) void corr_xy_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int B, float scale,
                                                             float* __restrict__ G) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * B) return;
  const int i = e / B, j = e - i * B;
  float s = 0.f;
  for (int sl = 0; sl < n_slabs; sl++) s += slabs[(int64_t)sl * kBP * kBP + i * kBP + j];
  G[e] = s * scale;
}

// dv = standardisation-backward( S Wh ), S(i,j) = transpose ? dG[j][i] : dG[i][j], times 1/F
__global__ __launch_bounds__(kNT) void corr_xy_bwd_kernel(const float* __restrict__ dG, int transpose,
                                                          const float* __restrict__ v, const float* __restrict__ w,
                                                          const float* __restrict__ stats_v,
                                                          const float* __restrict__ stats_w, int B, int64_t F, float eps,
                                                          float* __restrict__ dv, int n_tiles) {
  extern __shared__ float smem[];
  float* Ss = smem;                         // [128][129]
  float* Vs = Ss + kBP * (kBP + 1);         // [128][33]
  float* Ws = Vs + kBP * kLD;
  float* red = Ws + kBP * kLD;              // [8][32][2]
  const int tid = threadIdx.x;
  const float invF = 1.0f / (float)F;
  for (int e = tid; e < kBP * kBP; e += kNT) {
    const int i = e >> 7, j = e & 127;
    float s = 0.f;
    if (i < B && j < B) s = (transpose ? dG[j * B + i] : dG[i * B + j]) * invF;
    Ss[i * (kBP + 1) + j] = s;
  }
  const int c = tid & 31, ig = tid >> 5;    // column, row group: rows ig + 8a, a < 16
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int col0 = tile * kTF;
    const bool cok = (col0 + c) < F;
    __syncthreads();
    tile_load(v, B, F, col0, Vs);
    tile_load(w, B, F, col0, Ws);
    __syncthreads();
    const float mv = cok ? stats_v[col0 + c] : 0.f, rv = cok ? stats_v[F + col0 + c] : 0.f;
    const float mw = cok ? stats_w[col0 + c] : 0.f, rw = cok ? stats_w[F + col0 + c] : 0.f;
    for (int row = ig; row < kBP; row += 8) {
      const bool ok = cok && row < B;
      Vs[row * kLD + c] = ok ? (Vs[row * kLD + c] - mv) * rv : 0.0f;
      Ws[row * kLD + c] = ok ? (Ws[row * kLD + c] - mw) * rw : 0.0f;
    }
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int a = 0; a < 16; a++) acc[a] = 0.f;
    for (int j = 0; j < B; j++) {
      const float wj = Ws[j * kLD + c];
#pragma unroll
      for (int a = 0; a < 16; a++) acc[a] = fmaf(Ss[(ig + 8 * a) * (kBP + 1) + j], wj, acc[a]);
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int a = 0; a < 16; a++) {
      const int row = ig + 8 * a;
      if (row < B) { s0 += acc[a]; s1 += acc[a] * Vs[row * kLD + c]; }
    }
    red[(ig * kTF + c) * 2] = s0;
    red[(ig * kTF + c) * 2 + 1] = s1;
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) { t0 += red[(g * kTF + c) * 2]; t1 += red[(g * kTF + c) * 2 + 1]; }
    float kap = 1.0f;
    if (eps != 0.0f) { const float d = 1.0f - eps * rv; kap = (d > 1e-12f) ? 1.0f / d : 0.0f; }
    const float mean = t0 / (float)B, proj = t1 / (float)(B - 1) * kap;
    if (cok) {
#pragma unroll
      for (int a = 0; a < 16; a++) {
        const int row = ig + 8 * a;
        if (row < B) dv[(int64_t)row * F + col0 + c] = rv * (acc[a] - mean - Vs[row * kLD + c] * proj);
      }
    }
  }
}

constexpr size_t kBwdLds = (size_t)(kBP * (kBP + 1) + 2 * kBP * kLD + 8 * kTF * 2) * sizeof(float);

inline int xy_grid(int64_t F) {
  const int64_t n_tiles = (F + kTF - 1) / kTF;
  return (int)(n_tiles < 256 ? n_tiles : 256);
}

}  // namespace

extern "C" {

size_t alignq_corr_xy_ws_bytes(int B, int64_t F) {
  if (B < 2 || B > ALIGNQ_MAX_BATCH || F <= 0) return 0;
  return (size_t)xy_grid(F) * kBP * kBP * sizeof(float);
}

int alignq_corr_xy_fwd(const float* x, const float* y, int B, int64_t F, float eps, float* G, float* stats, void* ws,
                       void* stream) {
  if (!x || !y || !G || !ws || F <= 0) return ALIGNQ_EINVAL;
  if (B < 2 || B > ALIGNQ_MAX_BATCH) return ALIGNQ_EUNSUPPORTED;
  const int n_tiles = (int)((F + kTF - 1) / kTF), grid = xy_grid(F);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(corr_xy_fwd_kernel, grid, kNT, 0, st, x, y, B, F, eps, (float*)ws, stats, n_tiles);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(corr_xy_reduce_kernel, (B * B + 255) / 256, 256, 0, st, (const float*)ws, grid, B, 1.0f / (float)F, G);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int alignq_corr_xy_bwd(const float* dG, const float* x, const float* y, const float* stats, int B, int64_t F, float eps,
                       float* dx, float* dy, void* stream) {
  if (!dG || !x || !y || !stats || F <= 0) return ALIGNQ_EINVAL;
  if (B < 2 || B > ALIGNQ_MAX_BATCH) return ALIGNQ_EUNSUPPORTED;
  static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(corr_xy_bwd_kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBwdLds);
  if (attr != hipSuccess) return (int)attr;
  const int n_tiles = (int)((F + kTF - 1) / kTF);
  const int grid = n_tiles < 512 ? n_tiles : 512;
  hipStream_t st = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL(corr_xy_bwd_kernel, grid, kNT, kBwdLds, st, dG, 0, x, y, stats, stats + 2 * F, B, F, eps, dx, n_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (dy) {
    hipLaunchKernelGGL(corr_xy_bwd_kernel, grid, kNT, kBwdLds, st, dG, 1, y, x, stats + 2 * F, stats, B, F, eps, dy, n_tiles);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // extern "C"
