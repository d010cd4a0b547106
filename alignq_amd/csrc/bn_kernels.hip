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

int alignq_bn_bwd_apply(const float* dx, const float* z, const float* ab, const float* save, const float* dx_part, int B,
                        int C, int HW, float* dz, float* dgamma, float* dbeta, void* stream) {
  if (!dx || !z || !ab || !save || !dx_part || !dz || B < 1 || C < 1) return ALIGNQ_EINVAL;
  if (HW % 64) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(kSplit, C), kThreads, 0, (hipStream_t)stream, dx, z, ab, save, dx_part, B, C,
                     HW, dz, dgamma, dbeta, alignq_site::bwd_tile_features(B, (int64_t)C * HW));
  LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
