// Streaming-structure microbenchmark (developer tool): a 1-read : 1-write pass over 2^26 floats (268 MB each way) in the
// shapes the plain quantiser could take: threads per block, blocks per CU, float4 in flight per thread, grid-stride vs
// block-contiguous tiles, non-temporal loads / stores, and with / without the NERF32 quantiser arithmetic (MATH) so the
// cost of the transform (vector ALU + LDS table reads) shows beside the pure copy.  Prints GB/s per variant.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I alignq_amd/csrc tools/src/stream_bw.hip -o tools/bin/stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include "alignq_math.h"
using namespace alignq;

template <int NT, int U, int MODE, bool NTL, bool NTS, bool MATH>
__global__ __launch_bounds__(NT) void k(const float4* __restrict__ x, float4* __restrict__ y, long nvec, int kbits, float r) {
  __shared__ __attribute__((aligned(16))) float tab_lds[ALIGNQ_NERF_LDS_FLOATS];
  NerfTab tab;
  Levels nlev = make_levels(kbits, true);
  if (MATH) {
    nerf_tab_load(tab_lds);
    __syncthreads();
    tab = nerf_tab(tab_lds);
  }
  const long stride = (long)gridDim.x * NT;
  long i0, step, inner;
  if (MODE == 0) { i0 = (long)blockIdx.x * NT + threadIdx.x; step = U * stride; inner = stride; }       // grid-stride
  else { i0 = (long)blockIdx.x * NT * U + threadIdx.x; step = U * stride; inner = NT; }                  // block tiles
  for (; i0 < nvec; i0 += step) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = i0 + u * inner;
      const float4* p = x + (i < nvec ? i : i0);
      if (NTL) {
        v[u].x = __builtin_nontemporal_load(&p->x); v[u].y = __builtin_nontemporal_load(&p->y);
        v[u].z = __builtin_nontemporal_load(&p->z); v[u].w = __builtin_nontemporal_load(&p->w);
      } else v[u] = *p;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = i0 + u * inner;
      float4 o = v[u];
      if (MATH) {
        float t, b;
        o.x = act_quant1<0, true>(v[u].x, kbits, nlev, r, &t, &b, tab);
        o.y = act_quant1<0, true>(v[u].y, kbits, nlev, r, &t, &b, tab);
        o.z = act_quant1<0, true>(v[u].z, kbits, nlev, r, &t, &b, tab);
        o.w = act_quant1<0, true>(v[u].w, kbits, nlev, r, &t, &b, tab);
      }
      if (i < nvec) {
        if (NTS) {
          float4* q = y + i;
          __builtin_nontemporal_store(o.x, &q->x); __builtin_nontemporal_store(o.y, &q->y);
          __builtin_nontemporal_store(o.z, &q->z); __builtin_nontemporal_store(o.w, &q->w);
        } else y[i] = o;
      }
    }
  }
}

// The fused site kernels' access pattern as a pure copy: x is [128, F] row-major, a workgroup owns column tiles of TF features
// (each of the 128 rows contributes TF*4 contiguous bytes), thread = (column quad, row group).  Shows what that pattern alone
// can reach before any arithmetic, statistics or MFMA.
template <int NT, int TF, bool NTL>
__global__ __launch_bounds__(NT) void k_site(const float* __restrict__ x, float* __restrict__ y, long F, int n_tiles) {
  constexpr int LPR = TF / 4, RG = NT / LPR, RJ = (128 + RG - 1) / RG;
  const int c = threadIdx.x % LPR, rg = threadIdx.x / LPR;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long col = (long)tile * TF + 4 * c;
    float4 v[RJ];
#pragma unroll
    for (int j = 0; j < RJ; j++) {
      const int row = rg + RG * j;
      const float* p = x + (long)row * F + col;
      if (row < 128) {
        if (NTL) { v[j].x = __builtin_nontemporal_load(p); v[j].y = __builtin_nontemporal_load(p + 1); v[j].z = __builtin_nontemporal_load(p + 2); v[j].w = __builtin_nontemporal_load(p + 3); }
        else v[j] = *reinterpret_cast<const float4*>(p);
      }
    }
#pragma unroll
    for (int j = 0; j < RJ; j++) {
      const int row = rg + RG * j;
      if (row < 128) *reinterpret_cast<float4*>(y + (long)row * F + col) = v[j];
    }
  }
}

template <int NT, int TF, bool NTL>
void run_site(int grid) {
  const long F = (1L << 26) / 128;
  const int n_tiles = (int)(F / TF);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  extern float *dx, *dy;
  for (int it = 0; it < 12; it++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_site<NT, TF, NTL>), grid < n_tiles ? grid : n_tiles, NT, 0, 0, (const float*)dx, dy, F, n_tiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (it >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const float med = ms[ms.size() / 2];
  printf("site pattern [128, %ld]: NT=%4d TF=%3d ntl=%d grid=%5d : %7.1f us  %6.0f GB/s\n", F, NT, TF, (int)NTL, grid, med * 1e3,
         8.0 * (1L << 26) / (med * 1e-3) / 1e9);
  fflush(stdout);
}

float *dx, *dy;
static const long N = 1L << 26;
// STREAM_BW_SETS=<n>: rotate inputs and outputs over n buffer pairs (cold operands: with one pair, part of the re-read input is
// served by the 256 MB memory-side cache for non-temporal readers, see NOTES.md 5f)
static int g_sets = 1;
static float* g_xs[8];
static float* g_ys[8];

template <int NT, int U, int MODE, bool NTL, bool NTS, bool MATH>
void run(int blocks_per_cu) {
  const long nvec = N / 4;
  long need = (nvec + (long)NT * U - 1) / ((long)NT * U);
  int grid = (int)std::min<long>(need, 256L * blocks_per_cu);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  for (int it = 0; it < 12; it++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, U, MODE, NTL, NTS, MATH>), grid, NT, 0, 0, (const float4*)g_xs[it % g_sets], (float4*)g_ys[it % g_sets], nvec, 8, 2.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (it >= 2) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const float med = ms[ms.size() / 2];
  printf("NT=%4d U=%d mode=%s ntl=%d nts=%d math=%d blocks/CU=%2d grid=%6d : %7.1f us  %6.0f GB/s\n", NT, U, MODE ? "tile" : "grid",
         (int)NTL, (int)NTS, (int)MATH, blocks_per_cu, grid, med * 1e3, 8.0 * N / (med * 1e-3) / 1e9);
  fflush(stdout);
}

template <bool MATH>
void sweep() {
  for (int bpc : {4, 8, 16}) { run<256, 1, 0, false, false, MATH>(bpc); run<256, 2, 0, false, false, MATH>(bpc); run<256, 4, 0, false, false, MATH>(bpc); }
  for (int bpc : {4, 8, 16, 1 << 20}) { run<256, 2, 1, false, false, MATH>(bpc); run<256, 4, 1, false, false, MATH>(bpc); run<256, 8, 1, false, false, MATH>(bpc); }
  for (int bpc : {2, 4, 1 << 20}) { run<512, 2, 1, false, false, MATH>(bpc); run<512, 4, 1, false, false, MATH>(bpc); run<1024, 2, 1, false, false, MATH>(bpc); run<1024, 4, 1, false, false, MATH>(bpc); }
  for (int bpc : {8, 1 << 20}) {
    run<256, 4, 1, true, false, MATH>(bpc); run<256, 4, 1, false, true, MATH>(bpc); run<256, 4, 1, true, true, MATH>(bpc);
    run<256, 4, 0, true, true, MATH>(bpc);
  }
}

int main() {
  hipMalloc(&dx, N * 4); hipMalloc(&dy, N * 4);
  std::vector<float> h(N);
  unsigned s = 12345;
  for (long i = 0; i < N; i++) {   // ~N(0,1): sum of 4 uniforms, scaled
    float a = 0;
    for (int j = 0; j < 4; j++) { s = s * 1664525u + 1013904223u; a += (s >> 8) * (1.0f / 16777216.0f); }
    h[i] = (a - 2.0f) * 1.7320508f;
  }
  hipMemcpy(dx, h.data(), N * 4, hipMemcpyHostToDevice);
  hipMemcpy(dy, dx, N * 4, hipMemcpyDeviceToDevice);
  if (const char* e = getenv("STREAM_BW_SETS")) g_sets = std::max(1, std::min(8, atoi(e)));
  g_xs[0] = dx; g_ys[0] = dy;
  for (int i = 1; i < g_sets; i++) {
    hipMalloc(&g_xs[i], N * 4); hipMalloc(&g_ys[i], N * 4);
    hipMemcpy(g_xs[i], dx, N * 4, hipMemcpyDeviceToDevice);
  }
  hipDeviceSynchronize();
  {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; it++) {
      hipEventRecord(e0); hipMemcpyAsync(dy, dx, N * 4, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1); hipEventSynchronize(e1);
      float t; hipEventElapsedTime(&t, e0, e1);
      if (it == 4) printf("hipMemcpy D2D: %.1f us %.0f GB/s\n", t * 1e3, 8.0 * N / (t * 1e-3) / 1e9);
    }
  }
  printf("---- the site kernels' [128, F] column-tile pattern as a pure copy\n");
  if (!getenv("STREAM_BW_NO_SITE")) for (int grid : {256, 512, 1024, 8192}) {
    run_site<1024, 64, false>(grid); run_site<1024, 64, true>(grid); run_site<1024, 128, true>(grid);
    run_site<512, 32, true>(grid); run_site<512, 64, true>(grid); run_site<256, 64, true>(grid); run_site<1024, 32, true>(grid);
  }
  if (getenv("STREAM_BW_SITE_ONLY")) return 0;
  printf("---- operand sets: %d\n", g_sets);
  printf("---- pure copy\n");
  sweep<false>();
  printf("---- copy + NERF32 quantiser (k=8, r=2)\n");
  sweep<true>();
  return 0;
}
