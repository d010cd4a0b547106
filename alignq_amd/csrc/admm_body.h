// admm_body.h — the ADMM primal / dual update of ONE site as a device function (utils/optimizer.py:97-124), shared by
// admm_sgd_kernels.hip (alignq_admm_update) and multi_tensor_kernels.hip (alignq_sgd_admm_step_multi: the same work as one ROLE of
// the launch that also takes the SGD step).  One workgroup of 1024 threads per site.
#pragma once
#include <hip/hip_runtime.h>

#include "alignq_math.h"

namespace alignq {

constexpr int kAdmmThreads = 1024;  // one workgroup of 16 waves per site

__device__ __forceinline__ void block_sum3(double& a, double& b, double& c, double* sm /* [48] */) {
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  c = wave_sum_d(c);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) { sm[w] = a; sm[16 + w] = b; sm[32 + w] = c; }
  __syncthreads();
  const int nw = blockDim.x >> 6;
  a = b = c = 0;
  for (int i = 0; i < nw; i++) { a += sm[i]; b += sm[16 + i]; c += sm[32 + i]; }
}

// V = pad(D) + gamma / rho; A = (1 - (mu/rho) / |V|_F) V if |V|_F > mu/rho else 0; gamma += rho (pad(D) - A).   sm: 48 doubles of LDS
__device__ __forceinline__ void admm_update_site(const float* __restrict__ D, float* __restrict__ A, float* __restrict__ G, int b,
                                                 int dim, float mu, float rho, double* sm) {
  constexpr int kBig = kAdmmThreads;
  const int full = dim * dim;
  const float inv_rho = 1.0f / rho;
  const float thr = mu / rho;
  double ss = 0, z0 = 0, z1 = 0;
  constexpr int R = 16;                       // dim <= 128: the whole site lives in registers, ONE memory round trip
  if (full <= R * kBig) {
    float dv[R], gv[R];
#pragma unroll
    for (int u = 0; u < R; u++) {             // clamped, unconditional loads: all in flight together
      const int e = threadIdx.x + u * kBig;
      const int ec = e < full ? e : full - 1;
      const int i = ec / dim, j = ec - i * dim;
      const bool in = i < b && j < b;
      const float d = D[in ? i * b + j : 0];
      dv[u] = in ? d : 0.0f;
      gv[u] = G[ec];
    }
#pragma unroll
    for (int u = 0; u < R; u++) {
      if (threadIdx.x + u * kBig < full) {
        const float v = dv[u] + inv_rho * gv[u];
        ss += (double)v * (double)v;
      }
    }
    block_sum3(ss, z0, z1, sm);
    const float nv = (float)sqrt(ss);
    const float shrink = (nv > thr) ? (1.0f - thr / nv) : 0.0f;
#pragma unroll
    for (int u = 0; u < R; u++) {
      const int e = threadIdx.x + u * kBig;
      if (e < full) {
        const float a = shrink * (dv[u] + inv_rho * gv[u]);
        A[e] = a;
        G[e] = gv[u] + rho * (dv[u] - a);
      }
    }
    return;
  }
  for (int e = threadIdx.x; e < full; e += kBig) {
    int i = e / dim, j = e - i * dim;
    float d = (i < b && j < b) ? D[i * b + j] : 0.0f;
    float v = d + inv_rho * G[e];
    ss += (double)v * (double)v;
  }
  block_sum3(ss, z0, z1, sm);
  const float nv = (float)sqrt(ss);
  const float shrink = (nv > thr) ? (1.0f - thr / nv) : 0.0f;
  for (int e = threadIdx.x; e < full; e += kBig) {
    int i = e / dim, j = e - i * dim;
    float d = (i < b && j < b) ? D[i * b + j] : 0.0f;
    float gm = G[e];
    float a = shrink * (d + inv_rho * gm);
    A[e] = a;
    G[e] = gm + rho * (d - a);
  }
}

}  // namespace alignq
