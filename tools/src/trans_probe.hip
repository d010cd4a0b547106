// Probe (round 6): does compiler-generated code around the transcendental unit (v_rcp_f32 / v_exp_f32 results feeding SLP-packed
// v_pk_*_f32) give the same bits when waves of ANOTHER workgroup run matrix / LDS work on the same SIMD?  Even workgroups run the site
// backward's element transform (alignq_math.h act_transform_rcp, 16 elements per thread and pass, the load -> fma -> transform -> split
// shape of site_bwd4_kernel's first phase) and fold every result's bits into a per-thread checksum; odd workgroups are idle (mode 0),
// run the same transform out of phase (mode 1), or an MFMA + LDS loop (mode 2).  The checksums of the even workgroups must not depend
// on the mode.  Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -Ialignq_amd/csrc tools/src/trans_probe.hip -o tools/bin/trans_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "alignq_math.h"
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float v, __bf16& hi, __bf16& lo) { hi = (__bf16)v; lo = (__bf16)(v - (float)hi); }

__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ x, unsigned* __restrict__ sums, int iters, int mode, float r,
                                                float a, float b, float mx, float rx, float mt, float rt) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, blk = blockIdx.x;
  const float rjac = r * 0.7978845608f;
  __bf16* T = reinterpret_cast<__bf16*>(lds);
  if ((blk & 1) == 0 || mode == 1) {
    unsigned cs = 0;
    const int skew = (blk & 1) ? 7 : 0;
    for (int it = 0; it < iters + skew; it++) {
      float4 xr[4];
#pragma unroll
      for (int q = 0; q < 4; q++) xr[q] = reinterpret_cast<const float4*>(x)[((size_t)(blk >> 1) * iters + (it % iters)) * 1024 + q * 256 + tid];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        xr[q].x = __fmaf_rn(a, xr[q].x, b); xr[q].y = __fmaf_rn(a, xr[q].y, b);
        xr[q].z = __fmaf_rn(a, xr[q].z, b); xr[q].w = __fmaf_rn(a, xr[q].w, b);
      }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        __bf16 xh[4], xl[4], th[4], tl[4];
        float jt[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float xe = e == 0 ? xr[q].x : (e == 1 ? xr[q].y : (e == 2 ? xr[q].z : xr[q].w));
          split2((xe - mx) * rx, xh[q], xl[q]);
          float t, jac;
          alignq::act_transform_rcp(xe, r, rjac, &t, &jac);
          split2((t - mt) * rt, th[q], tl[q]);
          jt[q] = jac;
          cs = cs * 31u + __float_as_uint(t);
          cs = cs * 31u + __float_as_uint(jac);
        }
        const int o = ((4 * (tid & 7) + e) * 136 + 4 * (tid >> 3));
        *reinterpret_cast<bf4*>(T + o) = (bf4){xh[0], xh[1], xh[2], xh[3]};
        *reinterpret_cast<bf4*>(T + 32 * 136 + o) = (bf4){xl[0], xl[1], xl[2], xl[3]};
        *reinterpret_cast<bf4*>(T + 2 * 32 * 136 + o) = (bf4){th[0], th[1], th[2], th[3]};
        *reinterpret_cast<bf4*>(T + 3 * 32 * 136 + o) = (bf4){tl[0], tl[1], tl[2], tl[3]};
        cs ^= __float_as_uint(jt[0] + jt[1] + jt[2] + jt[3]);
      }
      __syncthreads();
    }
    if ((blk & 1) == 0) sums[(size_t)(blk >> 1) * 256 + tid] = cs;
    return;
  }
  if (mode == 2) {          // foreign work: MFMA with operands from LDS, like the kernel's second phase
    f16v acc = {};
    for (int i = tid; i < 4 * 32 * 136; i += 256) T[i] = (__bf16)(0.001f * (i & 63));
    __syncthreads();
    for (int it = 0; it < iters * 6; it++) {
      const bf8 av = *reinterpret_cast<const bf8*>(T + ((tid & 31) * 136 + 8 * ((tid >> 5) & 1) + 16 * (it & 7)));
      const bf8 bv = *reinterpret_cast<const bf8*>(T + 32 * 136 + ((tid & 31) * 136 + 8 * ((tid >> 5) & 1) + 16 * (it & 7)));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, av, acc, 0, 0, 0);
    }
    if (acc[0] == 12345.f) sums[0] = 1;
  }
}

int main(int argc, char** argv) {
  const int pairs = 256, iters = argc > 1 ? atoi(argv[1]) : 64, lds = 73728;
  const size_t n = (size_t)pairs * iters * 4096;
  std::vector<float> hx(n);
  unsigned s = 12345;
  for (size_t i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; hx[i] = ((int)(s >> 8) % 20001 - 10000) * 3e-4f; }
  float* dx; unsigned* ds;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, pairs * 256 * 4);
  hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<unsigned> ref(pairs * 256), got(pairs * 256);
  for (int mode = 0; mode < 3; mode++)
    for (int rep = 0; rep < 6; rep++) {
      hipMemset(ds, 0, pairs * 256 * 4);
      hipLaunchKernelGGL(probe, 2 * pairs, 256, lds, 0, dx, ds, iters, mode, 2.0f, 1.1f, -0.05f, 0.02f, 0.9f, 0.01f, 1.3f);
      hipMemcpy(got.data(), ds, pairs * 256 * 4, hipMemcpyDeviceToHost);
      if (mode == 0 && rep == 0) { ref = got; continue; }
      int bad = 0, badblk = 0;
      for (int b = 0; b < pairs; b++) { int c = 0; for (int t = 0; t < 256; t++) c += got[b * 256 + t] != ref[b * 256 + t]; bad += c; badblk += c > 0; }
      printf("mode %d rep %d: %d threads in %d workgroups differ from the first idle-partner launch\n", mode, rep, bad, badblk);
    }
  return 0;
}
