// Issue-rate microbenchmark for the VALU ops the quantise transform is built from (developer tool).
// Each kernel runs N dependent-chain-free iterations of ONE op on 8 independent accumulators per lane, 8 waves/SIMD.
// Reports lane-ops per cycle per CU (64 = one wave64 instruction per SIMD every 4 cycles).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(512) void k(float* out, float a, float b) {
  float r[8];
  f2 p[8];
  unsigned long long mask = __ballot(threadIdx.x & 1);
#pragma unroll
  for (int i = 0; i < 8; i++) { r[i] = a + i + threadIdx.x; p[i].x = r[i]; p[i].y = r[i] + 1.f; }
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
      if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 3) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
      if (OP == 4) asm volatile("v_rndne_f32 %0, %0" : "+v"(r[i]));
      if (OP == 5) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[i]) : "v"(1));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
      if (OP == 7) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(r[i]), "v"(0x264) : "vcc");
      if (OP == 8) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 9) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i]));
      if (OP == 10) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
      if (OP == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
      if (OP == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
      if (OP == 13) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(0x7fffffff), "v"(b));
      if (OP == 16) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(mask));
      if (OP == 17) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
      if (OP == 18) asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r[i]) : "v"(a), "v"(b), "s"(mask));
      if (OP == 19) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 20) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
      if (OP == 21) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 22) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
      if (OP == 14) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
      if (OP == 15) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) s += r[i] + p[i].x + p[i].y;
  if (s == 12345.678f) out[0] = s;
}
template <int OP>
void run(const char* name, int lanes_per_op) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int blocks = 256 * 4;   // 4 x 512 threads per CU = 8 waves/SIMD
  hipLaunchKernelGGL(k<OP>, blocks, 512, 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, blocks, 512, 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double ops = (double)blocks * 512 * ITER * 8 * lanes_per_op;     // element-ops
  int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);  // kHz
  const double cyc = ms * 1e-3 * clk * 1e3;
  printf("%-16s %8.3f ms  %6.1f element-ops/clk/CU (at %d MHz)\n", name, ms, ops / cyc / 256, clk / 1000);
  hipFree(d);
}
int main() {
  run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 2); run<2>("v_mul_f32", 1); run<3>("v_pk_mul_f32", 2);
  run<14>("v_add_f32", 1); run<15>("v_pk_add_f32", 2);
  run<4>("v_rndne_f32", 1); run<5>("v_ldexp_f32", 1); run<6>("v_cndmask_b32", 1); run<7>("v_cmp_class_f32", 1);
  run<8>("v_min_f32", 1); run<9>("v_cvt_i32_f32", 1); run<10>("v_exp_f32", 1); run<11>("v_rcp_f32", 1);
  run<12>("v_cmp_lt_f32", 1); run<13>("v_bfi_b32", 1);
  run<16>("cndmask_e64 sgpr", 1); run<17>("cmp+cndmask vcc", 2); run<18>("cndmask nodep", 1); run<19>("v_max_f32", 1);
  run<20>("v_med3_f32", 1); run<21>("v_and_b32", 1); run<22>("v_mov_b32", 1);
  return 0;
}
