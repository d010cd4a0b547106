// Grid-wide barrier cost (developer tool): G co-resident workgroups of T threads run N barriers inside ONE
// kernel; the cost of a barrier = (time with N barriers - time with 0) / N, against the ~4.8 us a dependent graph node costs.
// Forms:
//   flat : one agent-scope arrival counter (monotonic: target = (i + 1) * G), thread 0 of every workgroup adds 1 and spins on it;
//   tree : 8 first-level counters (blockIdx % 8 ~ the XCD a workgroup lands on), the last arriver of each adds to the root;
//          everybody spins on the root (8 atomics on the contended line instead of G);
//   flag : as flat, but the last arriver publishes a generation word the others spin on (reads hit a line nobody adds to).
// Every spin is bounded (kMaxPolls) and sets an error flag instead of hanging: the grid always drains.
// Between barriers each thread does a little dependent work on a register so the compiler keeps the loop.
// Build: hipcc -O3 --offload-arch=gfx950 tools/src/grid_barrier.hip -o tools/bin/grid_barrier
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e__ = (x);                                                          \
    if (e__ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                     \
      exit(2);                                                                     \
    }                                                                              \
  } while (0)

constexpr unsigned kMaxPolls = 1u << 22;

struct Bar {
  unsigned* root;    // arrival counter
  unsigned* lvl1;    // [8 * 32] first-level counters, one cache line apart
  unsigned* gen;     // generation flag
  unsigned* err;
};

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int FORM>
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned it, unsigned G) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // everything this workgroup wrote is visible device-wide
    if (FORM == 0) {
      __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = (it + 1) * G;
      unsigned polls = 0;
      while (ld_agent(b.root) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > kMaxPolls) { *b.err = 1; break; }
      }
    } else if (FORM == 1) {
      const unsigned x = blockIdx.x & 7u, nx = (G + 7u - x) / 8u;          // workgroups with this residue
      const unsigned prev = __hip_atomic_fetch_add(b.lvl1 + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1 == (it + 1) * nx) __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned nres = G < 8u ? G : 8u, target = (it + 1) * nres;
      unsigned polls = 0;
      while (ld_agent(b.root) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > kMaxPolls) { *b.err = 1; break; }
      }
    } else {
      const unsigned prev = __hip_atomic_fetch_add(b.root, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1 == (it + 1) * G) {
        __hip_atomic_store(b.gen, it + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        unsigned polls = 0;
        while (ld_agent(b.gen) < it + 1) {
          __builtin_amdgcn_s_sleep(1);
          if (++polls > kMaxPolls) { *b.err = 1; break; }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <int FORM>
__global__ void bar_kernel(Bar b, int n, float* out) {
  float v = (float)threadIdx.x;
  for (int i = 0; i < n; i++) {
    v = v * 1.0001f + 0.5f;
    grid_barrier<FORM>(b, (unsigned)i, gridDim.x);
  }
  if (v == 12345.678f) out[0] = v;
}

template <int FORM>
static float run(const Bar& b, int G, int T, int n, float* out, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  std::vector<float> ts;
  for (int r = 0; r < reps + 2; r++) {
    CK(hipMemsetAsync(b.root, 0, 4, 0));
    CK(hipMemsetAsync(b.lvl1, 0, 8 * 32 * 4, 0));
    CK(hipMemsetAsync(b.gen, 0, 4, 0));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bar_kernel<FORM>, dim3(G), dim3(T), 0, 0, b, n, out);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2] * 1e3f;      // us
}

int main() {
  Bar b;
  float* out;
  CK(hipMalloc(&b.root, 256));
  CK(hipMalloc(&b.lvl1, 8 * 32 * 4));
  CK(hipMalloc(&b.gen, 256));
  CK(hipMalloc(&b.err, 256));
  CK(hipMalloc(&out, 256));
  CK(hipMemset(b.err, 0, 4));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const int N = 400;
  const char* names[3] = {"flat", "tree", "flag"};
  printf("{\"device\": \"%s\", \"cus\": %d, \"barriers_per_kernel\": %d, \"results\": [", p.gcnArchName, cus, N);
  bool first = true;
  const int cfg[][2] = {{256, 1024}, {256, 256}, {512, 512}, {512, 256}, {1024, 256}, {2048, 256}};
  for (auto& c : cfg) {
    const int G = c[0], T = c[1];
    // co-residency: G workgroups of T threads must fit at once (2048 threads per CU, no LDS, few registers)
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bar_kernel<0>, T, 0));
    if ((long)per_cu * cus < G) continue;
    float t0[3], tn[3];
    t0[0] = run<0>(b, G, T, 0, out, 9); tn[0] = run<0>(b, G, T, N, out, 9);
    t0[1] = run<1>(b, G, T, 0, out, 9); tn[1] = run<1>(b, G, T, N, out, 9);
    t0[2] = run<2>(b, G, T, 0, out, 9); tn[2] = run<2>(b, G, T, N, out, 9);
    for (int f = 0; f < 3; f++) {
      printf("%s{\"form\": \"%s\", \"workgroups\": %d, \"threads\": %d, \"empty_kernel_us\": %.2f, \"us_per_barrier\": %.3f}",
             first ? "" : ", ", names[f], G, T, t0[f], (tn[f] - t0[f]) / N);
      first = false;
    }
  }
  unsigned err = 0;
  CK(hipMemcpy(&err, b.err, 4, hipMemcpyDeviceToHost));
  printf("], \"spin_timeouts\": %u}\n", err);
  return err ? 3 : 0;
}
