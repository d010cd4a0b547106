// wgrad_reduce_body.h — the filter-gradient slab reduction as a device function: conv_kernels.hip (wgrad_reduce[_multi]_kernel,
// the filler role of conv3x3_bwd_kernel) and site4_kernels.hip (the filler role of the narrow site backward launches) share it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alignq_wgr {

// dW[e] = sum over slabs in a fixed order.  1024 threads = G slab groups x (1024 / G) elements with G = n_slabs / 16 clipped to
// [1, 16] (a power of two): every thread owns (up to) 16 slab rows sl = g, g + G, ... and has all of them in flight at once;
// the G partial sums of an element meet in LDS and are added in group order.  Few slabs (C = 64: 16) therefore mean wide
// workgroups (1024 elements, 64 KB) instead of sixteen times as many workgroups of one load per thread.
__host__ __device__ __forceinline__ int wgrad_reduce_groups(int n_slabs) {
  int g = 1;
  while (g < 16 && g * 16 < n_slabs) g <<= 1;
  return g;
}
// (n_elem % 4 == 0, every filter shape here: a thread owns four consecutive elements, one 16-byte load per slab)
__host__ __device__ __forceinline__ int wgrad_reduce_blocks(int n_slabs, int n_elem, int nt = 1024) {
  const int per = (nt / wgrad_reduce_groups(n_slabs)) * ((n_elem & 3) ? 1 : 4);
  return (n_elem + per - 1) / per;
}
template <int EPL, int NT = 1024>
__device__ __forceinline__ void wgrad_reduce_body_t(const float* __restrict__ slabs, int n_slabs, int n_elem,
                                                    float* __restrict__ dw, int blk, float* __restrict__ part /* [NT * EPL] */) {
  const int G = wgrad_reduce_groups(n_slabs), per = NT / G;
  const int g = threadIdx.x / per, l = threadIdx.x - g * per;
  const int e = (blk * per + l) * EPL;
  const int ec = e < n_elem ? e : n_elem - EPL;
  float s[EPL];
#pragma unroll
  for (int q = 0; q < EPL; q++) s[q] = 0.f;
  constexpr int U = 16;
  for (int sl0 = g; sl0 < n_slabs; sl0 += G * U) {
    float v[U][EPL];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int sl = sl0 + G * u;
      const float* p = slabs + (int64_t)(sl < n_slabs ? sl : n_slabs - 1) * n_elem + ec;
      // non-temporal: the slabs are read once; as a filler role the stream must not evict what the host launch's consumer
      // (the next convolution's backward) is about to read from this XCD's L2
      if constexpr (EPL == 4) {
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
        const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p));
        v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
      } else {
        v[u][0] = __builtin_nontemporal_load(p);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (sl0 + G * u < n_slabs) {
#pragma unroll
        for (int q = 0; q < EPL; q++) s[q] += v[u][q];
      }
    }
  }
  if (G == 1) {                                    // block-uniform
    if (e < n_elem) {
#pragma unroll
      for (int q = 0; q < EPL; q++) dw[e + q] = s[q];
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < EPL; q++) part[threadIdx.x * EPL + q] = s[q];
  __syncthreads();
  if (g == 0 && e < n_elem) {
#pragma unroll
    for (int q = 0; q < EPL; q++) {
      float t = 0.f;
      for (int k = 0; k < G; k++) t += part[(k * per + l) * EPL + q];
      dw[e + q] = t;
    }
  }
}
template <int NT = 1024>
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slabs, int n_slabs, int n_elem,
                                                  float* __restrict__ dw, int blk, float* __restrict__ part /* [4 * NT] */) {
  if (n_elem & 3) wgrad_reduce_body_t<1, NT>(slabs, n_slabs, n_elem, dw, blk, part);      // block-uniform
  else wgrad_reduce_body_t<4, NT>(slabs, n_slabs, n_elem, dw, blk, part);
}

// Filler role of a convolution's (conv3x3_bwd_kernel) or a narrow site's (site_bwd4_kernel<32, ..>) backward launch: the slab
// reduction of up to four EARLIER convolutions' filter gradients (their partial sums were written by earlier launches; nothing
// reads the finished gradient before the weight quantiser's backward), by workgroups of 256 threads at the end of the grid.  Same groups, same summation order, same bits as
// wgrad_reduce[_multi]_kernel: only the elements per workgroup differ.
constexpr int kFill = 4;
struct RedFill {
  const float* slabs[kFill];
  float* dw[kFill];
  int n_slabs[kFill];
  int n_elem[kFill];
  int blk0[kFill + 1];    // first filler workgroup of every item, total
};

}  // namespace alignq_wgr
