// Probe (round 6): can an eagerly enqueued consumer on a side stream be ordered behind a node INSIDE a replayed HIP graph through a flag in
// device memory and hipStreamWaitValue32?  (torch refuses external events on ROCm.)  Each replay: kernels write a = step; a one-thread
// kernel publishes flag = step; more graph work follows.  The side stream waits for flag >= step, then copies a.
// Build: hipcc -O3 --offload-arch=gfx950 tools/src/probe_waitvalue.hip -o tools/bin/probe_waitvalue ; run under `timeout`.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void bump(unsigned* step) { *step += 1u; }
__global__ void fill(float* a, const unsigned* step, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) a[i] = (float)*step;
}
__global__ void publish(unsigned* flag, const unsigned* step) { __threadfence_system(); *flag = *step; }

int main() {
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
  if (!can) { printf("WAITVALUE_UNSUPPORTED\n"); return 0; }
  const long n = 1L << 26;
  float *a, *b, *out;
  unsigned *step, *flag;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&step, 4));
  if (getenv("PLAIN_FLAG")) { CK(hipMalloc(&flag, 8)); printf("flag in plain hipMalloc memory\n"); } else CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  CK(hipMemset(step, 0, 4)); CK(hipMemset(flag, 0, 8));
  hipStream_t cap, side;
  CK(hipStreamCreate(&cap)); CK(hipStreamCreate(&side));
  hipGraph_t graph; hipGraphExec_t exec;
  CK(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
  bump<<<1, 1, 0, cap>>>(step);
  for (int r = 0; r < 6; r++) fill<<<2048, 256, 0, cap>>>(a, step, n);
  publish<<<1, 1, 0, cap>>>(flag, step);
  for (int r = 0; r < 12; r++) fill<<<2048, 256, 0, cap>>>(b, step, n);       // later graph work the consumer overlaps with
  CK(hipStreamEndCapture(cap, &graph));
  CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1, e2;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  std::vector<float> h(4);
  bool ok = true;
  for (unsigned it = 1; it <= 6; it++) {
    CK(hipEventRecord(e0, cap));
    CK(hipGraphLaunch(exec, cap));
    CK(hipEventRecord(e1, cap));
    CK(hipStreamWaitValue32(side, flag, it, hipStreamWaitValueGte, 0xFFFFFFFFu));
    CK(hipMemcpyAsync(out, a, n * 4, hipMemcpyDeviceToDevice, side));
    CK(hipEventRecord(e2, side));
    CK(hipStreamSynchronize(side));
    CK(hipStreamSynchronize(cap));
    CK(hipMemcpy(h.data(), out, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h.data() + 1, out + n - 1, 4, hipMemcpyDeviceToHost));
    float tg, tc;
    CK(hipEventElapsedTime(&tg, e0, e1)); CK(hipEventElapsedTime(&tc, e0, e2));
    printf("replay %u: consumer saw %.0f / %.0f; graph %.1f us, consumer done at %.1f us after the launch\n", it, h[0], h[1], tg * 1e3, tc * 1e3);
    ok &= h[0] == (float)it && h[1] == (float)it;
  }
  printf(ok ? "WAITVALUE_OK\n" : "WAITVALUE_WRONG\n");
  return 0;
}
