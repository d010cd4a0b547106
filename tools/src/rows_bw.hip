// Access-shape microbenchmark for the small-batch site kernels (developer tool, round 6): a [G][B][F] tensor pair (x, res) is read and
// y written, a wave owning tiles of TF feature columns of ALL B rows - the pattern site1_fwd64_kernel has - in several forms:
//   form 0: lane = feature, one dword per row and lane (B wave instructions of 256 B per array)             [the kernel's form]
//   form 1: 16 lanes x 16 B per row, 4 rows per wave instruction (B / 4 instructions of 4 x 256 B per array), TF = 64
//   form 2: 64 lanes x 16 B per row, one row per instruction (B instructions of 1 KiB per array), TF = 256
//   form 3: the same bytes as a linear float4 copy (ceiling)
// mode bits: 1 = read x, 2 = read res, 4 = write y.   Prints us and GB/s of the bytes moved.
// Build: hipcc -O3 --offload-arch=gfx950 tools/src/rows_bw.hip -o tools/bin/rows_bw
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int NT = 256;

template <int FORM, int B, int MODE>
__global__ __launch_bounds__(NT, 4) void k(const float* __restrict__ x, const float* __restrict__ res, float* __restrict__ y, long F) {
  const long go = (long)blockIdx.y * B * F;
  x += go; res += go; y += go;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (FORM == 0) {
    const int n_tile = (int)(F / 64);
    for (int t = blockIdx.x * 4 + w; t < n_tile; t += gridDim.x * 4) {
      const long col = (long)t * 64 + lane;
      float a[B], b[B];
#pragma unroll
      for (int q = 0; q < B; q++) {
        a[q] = (MODE & 1) ? x[(long)q * F + col] : 1.0f;
        b[q] = (MODE & 2) ? res[(long)q * F + col] : 2.0f;
      }
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < B; q++) s += a[q];
#pragma unroll
      for (int q = 0; q < B; q++) {
        if (MODE & 4) y[(long)q * F + col] = a[q] * s + b[q];
      }
      if (!(MODE & 4) && s == 1234.5f) y[col] = s + b[0] + b[B - 1];
    }
  } else if (FORM == 1) {
    const int n_tile = (int)(F / 64);
    const int rl = lane >> 4, c4 = lane & 15;
    for (int t = blockIdx.x * 4 + w; t < n_tile; t += gridDim.x * 4) {
      const long col = (long)t * 64 + 4 * c4;
      float4 a[B / 4], b[B / 4];
#pragma unroll
      for (int q = 0; q < B / 4; q++) {
        a[q] = (MODE & 1) ? *reinterpret_cast<const float4*>(x + (long)(4 * q + rl) * F + col) : make_float4(1, 1, 1, 1);
        b[q] = (MODE & 2) ? *reinterpret_cast<const float4*>(res + (long)(4 * q + rl) * F + col) : make_float4(2, 2, 2, 2);
      }
      float s = 0.f;
#pragma unroll
      for (int q = 0; q < B / 4; q++) s += a[q].x + a[q].w;
#pragma unroll
      for (int q = 0; q < B / 4; q++) {
        if (MODE & 4)
          *reinterpret_cast<float4*>(y + (long)(4 * q + rl) * F + col) =
              make_float4(a[q].x * s + b[q].x, a[q].y * s + b[q].y, a[q].z * s + b[q].z, a[q].w * s + b[q].w);
      }
      if (!(MODE & 4) && s == 1234.5f) y[col] = s + b[0].x + b[B / 4 - 1].w;
    }
  } else if (FORM == 2) {
    const int n_tile = (int)(F / 256);
    for (int t = blockIdx.x * 4 + w; t < n_tile; t += gridDim.x * 4) {
      const long col = (long)t * 256 + 4 * lane;
      float s = 0.f, bb = 0.f;
      for (int q0 = 0; q0 < B; q0 += 7) {
        float4 a[7], b[7];
#pragma unroll
        for (int q = 0; q < 7; q++) {
          a[q] = (MODE & 1) ? *reinterpret_cast<const float4*>(x + (long)(q0 + q) * F + col) : make_float4(1, 1, 1, 1);
          b[q] = (MODE & 2) ? *reinterpret_cast<const float4*>(res + (long)(q0 + q) * F + col) : make_float4(2, 2, 2, 2);
        }
#pragma unroll
        for (int q = 0; q < 7; q++) {
          s += a[q].x + a[q].w;
          bb += b[q].y;
          if (MODE & 4)
            *reinterpret_cast<float4*>(y + (long)(q0 + q) * F + col) =
                make_float4(a[q].x + b[q].x, a[q].y + b[q].y, a[q].z + b[q].z, a[q].w + b[q].w);
        }
      }
      if (!(MODE & 4) && s == 1234.5f) y[col] = s + bb;
    }
  } else {
    const long nvec = (long)B * F / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* r4 = reinterpret_cast<const float4*>(res);
    float4* y4 = reinterpret_cast<float4*>(y);
    float s = 0.f;
    for (long i = (long)blockIdx.x * NT * 4 + threadIdx.x; i < nvec; i += (long)gridDim.x * NT * 4) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const long j = i + u * NT < nvec ? i + u * NT : i;
        a[u] = (MODE & 1) ? x4[j] : make_float4(1, 1, 1, 1);
        b[u] = (MODE & 2) ? r4[j] : make_float4(2, 2, 2, 2);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        s += a[u].x + b[u].y;
        if ((MODE & 4) && i + u * NT < nvec) y4[i + u * NT] = make_float4(a[u].x + b[u].x, a[u].y + b[u].y, a[u].z + b[u].z, a[u].w + b[u].w);
      }
    }
    if (!(MODE & 4) && s == 1234.5f) y[threadIdx.x] = s;
  }
}

static float *dx, *dr, *dy;

template <int FORM, int MODE>
void run(long F, int G, int grid) {
  constexpr int B = 28;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<float> ms;
  for (int it = 0; it < 14; it++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM, B, MODE>), dim3(grid, G), NT, 0, 0, (const float*)dx, (const float*)dr, dy, F);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float t; hipEventElapsedTime(&t, e0, e1);
    if (it >= 3) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  const float med = ms[ms.size() / 2];
  const double bytes = 4.0 * G * B * F * (((MODE & 1) ? 1 : 0) + ((MODE & 2) ? 1 : 0) + ((MODE & 4) ? 1 : 0));
  printf("F=%7ld G=%d form=%d mode=%d grid=%4d: %7.1f us %6.0f GB/s\n", F, G, FORM, MODE, grid, med * 1e3, bytes / (med * 1e-3) / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const long F = argc > 1 ? atol(argv[1]) : 802816;
  const int G = 2;
  const size_t n = (size_t)G * 28 * F;
  hipMalloc(&dx, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&dy, n * 4);
  hipMemset(dx, 0, n * 4); hipMemset(dr, 0, n * 4); hipMemset(dy, 0, n * 4);
  for (int grid : {512, 1024}) {
    run<0, 7>(F, G, grid); run<1, 7>(F, G, grid); run<2, 7>(F, G, grid); run<3, 7>(F, G, grid);
    run<0, 3>(F, G, grid); run<1, 3>(F, G, grid); run<2, 3>(F, G, grid); run<3, 3>(F, G, grid);
    run<0, 1>(F, G, grid); run<1, 1>(F, G, grid); run<2, 1>(F, G, grid); run<3, 1>(F, G, grid);
    run<0, 5>(F, G, grid); run<1, 5>(F, G, grid); run<2, 5>(F, G, grid); run<3, 5>(F, G, grid);
  }
  return 0;
}
