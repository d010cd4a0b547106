#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "../alignq_amd/csrc/alignq_math.h"
using namespace alignq;
__global__ void kern(int k, float* out_yn, int* bad, float* firstbad) {
  Levels L = make_levels(k, true);
  if (threadIdx.x == 0 && blockIdx.x == 0) *out_yn = L.yn;
  int lim = 8 * ((1 << k) - 1) + 2;
  for (int v = -lim + (int)(blockIdx.x * blockDim.x + threadIdx.x); v <= lim; v += gridDim.x * blockDim.x) {
    float b = (float)v;
    float ref = __fdiv_rn(b, L.n), got = div_const(b, L.n, L.yn);
    if (__float_as_int(ref) != __float_as_int(got)) { if (atomicAdd(bad, 1) == 0) { firstbad[0] = b; firstbad[1] = ref; firstbad[2] = got; } }
  }
}
int main() {
  float *yn, *fb; int* bad;
  hipMalloc(&yn, 4); hipMalloc(&bad, 4); hipMalloc(&fb, 12);
  for (int k = 2; k <= 10; k++) {
    hipMemset(bad, 0, 4);
    kern<<<64, 256>>>(k, yn, bad, fb);
    float h, hf[3]; int hb;
    hipMemcpy(&h, yn, 4, hipMemcpyDeviceToHost); hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost); hipMemcpy(hf, fb, 12, hipMemcpyDeviceToHost);
    float n = (float)((1 << k) - 1);
    printf("k=%d yn_dev=%a yn_host=%a bad=%d first b=%g ref=%a got=%a\n", k, h, 1.0f / n, hb, hf[0], hf[1], hf[2]);
  }
  return 0;
}
