// head_body.h — the classifier head's backward as a device function, shared by head_kernels.hip (alignq_head_ce_bwd) and
// site4_kernels.hip (alignq_head_ce_bwd_site_prep: the same work as one ROLE of a launch that also prepares the sites' S matrices).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace alignq_head {

constexpr int kMaxC = 256, kMaxK = 64;

// blocks [0, B): dfeat[b][p][c] = (1/HW) sum_j dl[b][j] W[j][c] for every pixel p;  blocks [B, B+K): dW[j][:], dbias[j]
// with dl[b][j] = g * (probs[b][j] - [j == target_b]) / B   (mean reduction of the loss; g = upstream scalar)
// `block`: index among the B + K workgroups of this role (256 threads each)
__device__ __forceinline__ void head_bwd_body(const float* __restrict__ g, const float* __restrict__ probs,
                                              const int64_t* __restrict__ target, const float* __restrict__ pooled,
                                              const float* __restrict__ W, int B, int HW, int C, int K,
                                              float* __restrict__ dfeat, float* __restrict__ dW,
                                              float* __restrict__ dbias, int block) {
  __shared__ float sd[kMaxK];
  const int tid = threadIdx.x;
  const float gs = g[0] / (float)B;
  __shared__ float sdl[1024];
  __shared__ float spart[256];
  const int parts = 256 / C > 0 ? 256 / C : 1;
  const int c = tid % C, part = tid / C;
  if (block < B) {
    const int b = block;
    if (tid < K) sd[tid] = gs * (probs[(int64_t)b * K + tid] - (target[b] == tid ? 1.f : 0.f));
    __syncthreads();
    if (part < parts) {
      float s = 0.f;
      for (int j = 0; j < K; j++) s = __fmaf_rn(sd[j], W[(int64_t)j * C + c], s);
      s = s / (float)HW;
      float* p = dfeat + (int64_t)b * HW * C + c;
      const int per = (HW + parts - 1) / parts;
      const int i0 = part * per, i1 = (i0 + per < HW) ? i0 + per : HW;
      for (int i = i0; i < i1; i++) p[(int64_t)i * C] = s;
    }
  } else {
    const int j = block - B;
    // dl[b] for this class (B <= 1024 staged in LDS; larger batches fall back to recomputing in the loop)
    const bool staged = B <= 1024;
    if (staged)
      for (int b = tid; b < B; b += 256) sdl[b] = gs * (probs[(int64_t)b * K + j] - (target[b] == j ? 1.f : 0.f));
    __syncthreads();
    float dw = 0.f;
    if (part < parts) {
      const int per = (B + parts - 1) / parts;
      const int b0 = part * per, b1 = (b0 + per < B) ? b0 + per : B;
      int b = b0;
      for (; b + 8 <= b1; b += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = pooled[(int64_t)(b + u) * C + c];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const float d = staged ? sdl[b + u] : gs * (probs[(int64_t)(b + u) * K + j] - (target[b + u] == j ? 1.f : 0.f));
          dw = __fmaf_rn(d, v[u], dw);
        }
      }
      for (; b < b1; b++) {
        const float d = staged ? sdl[b] : gs * (probs[(int64_t)b * K + j] - (target[b] == j ? 1.f : 0.f));
        dw = __fmaf_rn(d, pooled[(int64_t)b * C + c], dw);
      }
    }
    spart[tid] = dw;
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
      for (int q = 0; q < parts; q++) t += spart[q * C + tid];
      dW[(int64_t)j * C + tid] = t;
    }
    if (tid == 0 && dbias) {
      float db = 0.f;
      for (int b = 0; b < B; b++) db += staged ? sdl[b] : gs * (probs[(int64_t)b * K + j] - (target[b] == j ? 1.f : 0.f));
      dbias[j] = db;
    }
  }
}


}  // namespace alignq_head
