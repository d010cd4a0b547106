// bn_kernels.hip — the batch-norm pieces that remain when BN is folded into the ADMM-site kernels.
//
// Reference call site: out, loss = act_q(bn(conv(x)))  (cdf_alignment_admm/resnet-56-cifar-10/model/resnet.py:87-94).
// nn.BatchNorm2d in training mode normalises with the biased batch variance and updates running_mean/var with momentum
// (unbiased variance) and num_batches_tracked.  Folded form:
//   forward : bn_stats (per-channel sum / sum of squares of the conv output z, one block per (channel, batch split)) ->
//             bn_finalize (mean, invstd, running stats, a = gamma*invstd, b = beta - mean*a) -> the site kernel applies
//             x = a*z + b on load, so the normalised activation is never written or re-read;
//   backward: the site backward writes dx (gradient w.r.t. the BN output) plus per-tile sums of dx and dx*zhat ->
//             bn_bwd_apply: dgamma = sum dx*zhat, dbeta = sum dx, dz = a*(dx - mean(dx) - zhat*mean(dx*zhat)).
// z is [B, C, HW] row-major (HW % 64 == 0 so that a 64-feature site tile lies in one channel).
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "site_internal.h"

using namespace alignq;

namespace {

constexpr int kThreads = 256;
constexpr int kSplit = alignq_site::kBnSplit;   // batch splits per channel (partials per channel)

__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const float* __restrict__ z, int B, int C, int HW,
                                                            double* __restrict__ part) {
  __shared__ double sm[32];
  const int c = blockIdx.y, s = blockIdx.x;
  const int rows = (B + kSplit - 1) / kSplit;
  const int b0 = s * rows, b1 = (b0 + rows < B) ? b0 + rows : B;
  const int nv = HW >> 2;
  double a = 0, q = 0;
  for (int b = b0; b < b1; b++) {
    const float4* p = reinterpret_cast<const float4*>(z + ((int64_t)b * C + c) * HW);
    for (int i = threadIdx.x; i < nv; i += kThreads) {
      const float4 v = p[i];
      a += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
      q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
  }
  block_sum2d(a, q, sm);
  if (threadIdx.x == 0) { part[(c * kSplit + s) * 2] = a; part[(c * kSplit + s) * 2 + 1] = q; }
}

__global__ void bn_finalize_kernel(const double* __restrict__ part, int B, int C, int HW, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, long long* __restrict__ nbt, float momentum,
                                   float eps, float* __restrict__ ab, float* __restrict__ save) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  double a = 0, q = 0;
  for (int s = 0; s < kSplit; s++) { a += part[(c * kSplit + s) * 2]; q += part[(c * kSplit + s) * 2 + 1]; }
  const double n = (double)B * (double)HW;
  const double mean = a / n;
  double var = q / n - mean * mean;
  if (var < 0) var = 0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float g = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
  const float av = g * invstd;
  ab[c] = av;
  ab[C + c] = bt - (float)mean * av;
  save[c] = (float)mean;
  save[C + c] = invstd;
  if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
  if (running_var) running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)(var * n / (n - 1.0));
}

// ---- channels-last (NHWC) statistics ---------------------------------------------------------------------------------
// z is [B, H, W, C] in memory (torch.channels_last), C = 4 * 2^k <= 256.  kNhwcParts workgroups; block g owns a contiguous
// range of pixels (all C channels of a pixel are contiguous): thread -> (pixel slot, channel quad), 8 loads in flight,
// double accumulation, butterfly over the lanes sharing a quad, waves through LDS, one {sum, sum of squares} partial per
// (channel, block): part [C][kNhwcParts][2].  The site forward finalises its tile's channels from these (a wave per channel).
constexpr int kNhwcThreads = 256;
constexpr int kNhwcParts = alignq_site::kNhwcParts;
__global__ __launch_bounds__(kNhwcThreads) void bn_stats_nhwc_kernel(const float* __restrict__ z, int B, int C, int HW,
                                                                 double* __restrict__ part) {
  __shared__ double sm[kNhwcThreads][8];
  const int tid = threadIdx.x;
  const int C4 = C >> 2;                       // channel quads per pixel
  const int slots = kNhwcThreads / C4;         // pixels in flight per block iteration (C = 4 * 2^k <= 256)
  const int cq = tid % C4, slot = tid / C4;
  const int64_t npix = (int64_t)B * HW;
  const int64_t per = (npix + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per, p1 = (p0 + per < npix) ? p0 + per : npix;
  double a[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  constexpr int U = 8;                         // loads in flight per thread
  for (int64_t base = p0 + slot; base < p1; base += (int64_t)slots * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {              // clamped address, masked below: no branch around the load
      const int64_t px = base + (int64_t)u * slots;
      v[u] = *reinterpret_cast<const float4*>(z + (px < p1 ? px : p1 - 1) * C + 4 * cq);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (base + (int64_t)u * slots < p1) {
        a[0] += v[u].x; a[1] += v[u].y; a[2] += v[u].z; a[3] += v[u].w;
        q[0] += (double)v[u].x * v[u].x; q[1] += (double)v[u].y * v[u].y;
        q[2] += (double)v[u].z * v[u].z; q[3] += (double)v[u].w * v[u].w;
      }
    }
  }
  // lanes of a wave that share a channel quad (lane = cq mod C4): butterfly over the slot bits, then waves via LDS
  const int lane = tid & 63, w = tid >> 6;
  for (int o = C4; o < 64; o <<= 1) {          // C4 < 64 always (C <= 256 would give C4 = 64: then no lane shares)
#pragma unroll
    for (int e = 0; e < 4; e++) { a[e] += __shfl_xor(a[e], o, 64); q[e] += __shfl_xor(q[e], o, 64); }
  }
  if (lane < C4 || C4 >= 64) {
    const int row = (C4 >= 64) ? tid : w * C4 + lane;      // [wave][quad]
#pragma unroll
    for (int e = 0; e < 4; e++) { sm[row][e] = a[e]; sm[row][4 + e] = q[e]; }
  }
  __syncthreads();
  if (tid < C) {                                // channel tid: quad tid/4, component tid%4, summed over the waves
    const int qd = tid >> 2, e = tid & 3;
    double sa = 0, sq = 0;
    const int nw = (C4 >= 64) ? kNhwcThreads / C4 : kNhwcThreads / 64;
    for (int ww = 0; ww < nw; ww++) { sa += sm[ww * C4 + qd][e]; sq += sm[ww * C4 + qd][4 + e]; }
    part[((int64_t)tid * kNhwcParts + blockIdx.x) * 2] = sa;
    part[((int64_t)tid * kNhwcParts + blockIdx.x) * 2 + 1] = sq;
  }
}

// grid (kSplit, C).  dx_part: per tile of `tile_f` features (tile = c*(HW/tile_f) + t) {sum dx, sum dx*zhat}
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ dx, const float* __restrict__ z,
                                                                const float* __restrict__ ab,
                                                                const float* __restrict__ save,
                                                                const float* __restrict__ dx_part, int B, int C, int HW,
                                                                float* __restrict__ dz, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int tile_f) {
  const int c = blockIdx.y, s = blockIdx.x;
  const int tpc = HW / tile_f;
  double s0 = 0, s1 = 0;
  for (int t = 0; t < tpc; t++) { s0 += dx_part[2 * (c * tpc + t)]; s1 += dx_part[2 * (c * tpc + t) + 1]; }
  if (s == 0 && threadIdx.x == 0) {
    if (dbeta) dbeta[c] = (float)s0;
    if (dgamma) dgamma[c] = (float)s1;
  }
  const double n = (double)B * (double)HW;
  const float k0 = (float)(s0 / n), k1 = (float)(s1 / n);
  const float a = ab[c], mu = save[c], is = save[C + c];
  const int rows = (B + kSplit - 1) / kSplit;
  const int b0 = s * rows, b1 = (b0 + rows < B) ? b0 + rows : B;
  const int nv = HW >> 2;
  for (int b = b0; b < b1; b++) {
    const int64_t base = ((int64_t)b * C + c) * HW;
    const float4* pd = reinterpret_cast<const float4*>(dx + base);
    const float4* pz = reinterpret_cast<const float4*>(z + base);
    float4* po = reinterpret_cast<float4*>(dz + base);
    for (int i = threadIdx.x; i < nv; i += kThreads) {
      const float4 d = pd[i], zz = pz[i];
      float4 o;
      o.x = a * (d.x - k0 - (zz.x - mu) * is * k1);
      o.y = a * (d.y - k0 - (zz.y - mu) * is * k1);
      o.z = a * (d.z - k0 - (zz.z - mu) * is * k1);
      o.w = a * (d.w - k0 - (zz.w - mu) * is * k1);
      po[i] = o;
    }
  }
}

// channels-last backward apply.  Stage 1 (every block, fixed order): per-channel totals of the site backward's per-tile
// partials part [n_tiles][cp][2] (cp = min(C, tile_f); channel c sits at entry c mod tile_f of the tiles t with
// t mod (C/cp) == c / tile_f) -> k0 = sum dx / n, k1 = sum dx*zhat / n; block 0 writes dbeta / dgamma.  Stage 2: this block's
// slice of dz = a[c] * (dx - k0[c] - zhat * k1[c]), c = f mod C, 4 float4 per thread and iteration (8 loads in flight).
__global__ __launch_bounds__(kNhwcThreads) void bn_bwd_apply_nhwc_kernel(
    const float* __restrict__ dx, const float* __restrict__ z, const float* __restrict__ ab, const float* __restrict__ save,
    const float* __restrict__ part, int n_tiles, int tile_f, int B, int C, int HW, float* __restrict__ dz,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ double sm[kNhwcThreads][2];
  __shared__ __attribute__((aligned(16))) float kk[2][256];
  const int tid = threadIdx.x;
  {
    const int cp = C < tile_f ? C : tile_f;
    const int cyc = C / cp;
    const int cnt = n_tiles / cyc;
    const int groups = kNhwcThreads / C;
    const int c = tid % C, grp = tid / C;
    const int e = c % cp, t_first = c / cp;
    double s0 = 0, s1 = 0;
    constexpr int U = 8;
    for (int i0 = grp; i0 < cnt; i0 += groups * U) {
      float2 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int i = i0 + u * groups;
        const int ic = i < cnt ? i : cnt - 1;
        v[u] = *reinterpret_cast<const float2*>(part + ((int64_t)(ic * cyc + t_first) * cp + e) * 2);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (i0 + u * groups < cnt) { s0 += v[u].x; s1 += v[u].y; }
      }
    }
    sm[tid][0] = s0;
    sm[tid][1] = s1;
    __syncthreads();
    if (grp == 0) {
      double t0 = 0, t1 = 0;
      for (int g = 0; g < groups; g++) { t0 += sm[g * C + c][0]; t1 += sm[g * C + c][1]; }
      if (blockIdx.x == 0) {
        if (dbeta) dbeta[c] = (float)t0;
        if (dgamma) dgamma[c] = (float)t1;
      }
      const double n = (double)B * (double)HW;
      kk[0][c] = (float)(t0 / n);
      kk[1][c] = (float)(t1 / n);
    }
    __syncthreads();
  }
  const int64_t nvec = (int64_t)B * C * HW / 4;
  const int64_t per = (nvec + gridDim.x - 1) / gridDim.x;
  const int64_t v0 = (int64_t)blockIdx.x * per, v1 = (v0 + per < nvec) ? v0 + per : nvec;
  const float4* pd = reinterpret_cast<const float4*>(dx);
  const float4* pz = reinterpret_cast<const float4*>(z);
  float4* po = reinterpret_cast<float4*>(dz);
  constexpr int U = 4;
  for (int64_t i0 = v0 + tid; i0 < v1; i0 += (int64_t)kNhwcThreads * U) {
    float4 d[U], zz[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = i0 + (int64_t)u * kNhwcThreads;
      const int64_t ic = i < v1 ? i : v1 - 1;
      d[u] = pd[ic];
      zz[u] = pz[ic];
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = i0 + (int64_t)u * kNhwcThreads;
      if (i < v1) {
        const int ch = (int)((i * 4) & (C - 1));
        const float4 a4 = *reinterpret_cast<const float4*>(ab + ch);
        const float4 m4 = *reinterpret_cast<const float4*>(save + ch);
        const float4 i4 = *reinterpret_cast<const float4*>(save + C + ch);
        const float4 k0 = *reinterpret_cast<const float4*>(&kk[0][ch]);
        const float4 k1 = *reinterpret_cast<const float4*>(&kk[1][ch]);
        float4 o;
        o.x = a4.x * (d[u].x - k0.x - (zz[u].x - m4.x) * i4.x * k1.x);
        o.y = a4.y * (d[u].y - k0.y - (zz[u].y - m4.y) * i4.y * k1.y);
        o.z = a4.z * (d[u].z - k0.z - (zz[u].z - m4.z) * i4.z * k1.z);
        o.w = a4.w * (d[u].w - k0.w - (zz[u].w - m4.w) * i4.w * k1.w);
        po[i] = o;
      }
    }
  }
}

// Stage 1 of bn_bwd_apply_nhwc_kernel alone: per-channel totals of the site backward's per-tile partial sums ->
// ktot = {sum g / n, sum g*zhat / n} and dgamma / dbeta.  One workgroup; used when the consumer of the batch-norm input
// gradient (alignq_conv3x3_nhwc_bwd) forms that gradient on load.
__global__ __launch_bounds__(kNhwcThreads) void bn_bwd_totals_nhwc_kernel(const float* __restrict__ part, int n_tiles,
                                                                      int tile_f, int B, int C, int HW,
                                                                      float* __restrict__ ktot, float* __restrict__ dgamma,
                                                                      float* __restrict__ dbeta) {
  __shared__ double sm[kNhwcThreads][2];
  const int tid = threadIdx.x;
  const int cp = C < tile_f ? C : tile_f;
  const int cyc = C / cp;
  const int cnt = n_tiles / cyc;
  const int groups = kNhwcThreads / C;
  const int c = tid % C, grp = tid / C;
  const int e = c % cp, t_first = c / cp;
  double s0 = 0, s1 = 0;
  constexpr int U = 8;
  for (int i0 = grp; i0 < cnt; i0 += groups * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * groups;
      const int ic = i < cnt ? i : cnt - 1;
      v[u] = *reinterpret_cast<const float2*>(part + ((int64_t)(ic * cyc + t_first) * cp + e) * 2);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (i0 + u * groups < cnt) { s0 += v[u].x; s1 += v[u].y; }
    }
  }
  sm[tid][0] = s0;
  sm[tid][1] = s1;
  __syncthreads();
  if (grp == 0) {
    double t0 = 0, t1 = 0;
    for (int g = 0; g < groups; g++) { t0 += sm[g * C + c][0]; t1 += sm[g * C + c][1]; }
    if (dbeta) dbeta[c] = (float)t0;
    if (dgamma) dgamma[c] = (float)t1;
    const double n = (double)B * (double)HW;
    ktot[c] = (float)(t0 / n);
    ktot[C + c] = (float)(t1 / n);
  }
}

}  // namespace

#define LAUNCH_CHECK()                          \
  do {                                          \
    hipError_t e__ = hipGetLastError();         \
    if (e__ != hipSuccess) return (int)e__;     \
  } while (0)

extern "C" {

size_t alignq_bn_ws_bytes(int C) { return (size_t)(C > 0 ? C : 1) * kSplit * 2 * sizeof(double); }

int alignq_bn_stats(const float* z, int B, int C, int HW, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* ab, float* save,
                    void* ws, void* stream) {
  if (!z || !ab || !save || !ws || B < 1 || C < 1 || HW < 4) return ALIGNQ_EINVAL;
  if ((HW & 3) || (reinterpret_cast<uintptr_t>(z) & 15)) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(kSplit, C), kThreads, 0, st, z, B, C, HW, (double*)ws);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, (C + 63) / 64, 64, 0, st, (const double*)ws, B, C, HW, gamma, beta, running_mean,
                     running_var, (long long*)num_batches_tracked, momentum, eps, ab, save);
  LAUNCH_CHECK();
  return 0;
}

int alignq_bn_partial_stats(const float* z, int B, int C, int HW, void* ws, void* stream) {
  if (!z || !ws || B < 1 || C < 1 || HW < 4) return ALIGNQ_EINVAL;
  if ((HW & 3) || (reinterpret_cast<uintptr_t>(z) & 15)) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(kSplit, C), kThreads, 0, (hipStream_t)stream, z, B, C, HW, (double*)ws);
  LAUNCH_CHECK();
  return 0;
}

static inline bool nhwc_channels_ok(int C) { return C >= 4 && C <= 256 && (C & (C - 1)) == 0; }

size_t alignq_bn_nhwc_ws_bytes(int C) { return (size_t)kNhwcParts * (size_t)(C > 0 ? C : 1) * 2 * sizeof(double); }

int alignq_bn_partial_stats_nhwc(const float* z, int B, int C, int HW, void* ws, void* stream) {
  if (!z || !ws || B < 1 || C < 1 || HW < 1) return ALIGNQ_EINVAL;
  if (!nhwc_channels_ok(C) || (reinterpret_cast<uintptr_t>(z) & 15)) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(bn_stats_nhwc_kernel, kNhwcParts, kNhwcThreads, 0, (hipStream_t)stream, z, B, C, HW, (double*)ws);
  LAUNCH_CHECK();
  return 0;
}

int alignq_bn_bwd_totals(const float* dx_part, int B, int C, int HW, float* ktot, float* dgamma, float* dbeta, void* stream) {
  if (!dx_part || !ktot || B < 1 || C < 1 || HW < 1) return ALIGNQ_EINVAL;
  if (!nhwc_channels_ok(C)) return ALIGNQ_EUNSUPPORTED;
  const int64_t F = (int64_t)C * HW;
  const int tf = alignq_site::bwd_tile_features(B, F);
  if (F % tf) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(bn_bwd_totals_nhwc_kernel, 1, kNhwcThreads, 0, (hipStream_t)stream, dx_part, (int)(F / tf), tf, B, C, HW,
                     ktot, dgamma, dbeta);
  LAUNCH_CHECK();
  return 0;
}

int alignq_bn_bwd_apply(const float* dx, const float* z, const float* ab, const float* save, const float* dx_part, int B,
                        int C, int HW, int nhwc, float* dz, float* dgamma, float* dbeta, void* stream) {
  if (!dx || !z || !ab || !save || !dx_part || !dz || B < 1 || C < 1) return ALIGNQ_EINVAL;
  if (nhwc) {
    if (!nhwc_channels_ok(C)) return ALIGNQ_EUNSUPPORTED;
    const int64_t F = (int64_t)C * HW;
    const int tf = alignq_site::bwd_tile_features(B, F);
    if (F % tf) return ALIGNQ_EUNSUPPORTED;
    const int n_tiles = (int)(F / tf);
    hipLaunchKernelGGL(bn_bwd_apply_nhwc_kernel, 128, kNhwcThreads, 0, (hipStream_t)stream, dx, z, ab, save, dx_part, n_tiles,
                       tf, B, C, HW, dz, dgamma, dbeta);
    LAUNCH_CHECK();
    return 0;
  }
  if (HW % 64) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(kSplit, C), kThreads, 0, (hipStream_t)stream, dx, z, ab, save, dx_part, B, C,
                     HW, dz, dgamma, dbeta, alignq_site::bwd_tile_features(B, (int64_t)C * HW));
  LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
