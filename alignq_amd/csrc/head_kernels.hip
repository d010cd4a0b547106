// head_kernels.hip — the classifier head of the training harness fused with its loss (channels-last features):
//   out = avgpool(out); out = out.view(B, -1); logits = logit(out)        (model/resnet.py:127-129 of the reference)
//   loss = CrossEntropyLoss()(logits, targets)                             (main.py: criterion)
// as ONE forward and ONE backward launch instead of ~12 small PyTorch / rocBLAS kernels (pool, addmm, log-softmax, nll and
// their backward GEMMs) that are pure launch latency at batch 128 x 64 features x 10 classes.  Plain fp32 arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alignq.h"

namespace {

constexpr int kMaxC = 256, kMaxK = 64;

// one workgroup per sample: pooled[c] = mean_p feat[b][p][c]; logits = W pooled + bias; log-softmax; loss_b; probs
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const int64_t* __restrict__ target,
                                                       int HW, int C, int K, float* __restrict__ pooled,
                                                       float* __restrict__ logits, float* __restrict__ probs,
                                                       float* __restrict__ loss) {
  __shared__ float sp[kMaxC];
  __shared__ float sl[kMaxK];
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ float spart[256];
  {   // thread -> (channel, pixel part): 256 / C parts, 8 loads in flight, then a fixed-order sum over the parts
    const int parts = 256 / C > 0 ? 256 / C : 1;
    const int c = tid % C, part = tid / C;
    float s = 0.f;
    if (part < parts) {
      const float* p = feat + (int64_t)b * HW * C + c;
      const int per = (HW + parts - 1) / parts;
      const int i0 = part * per, i1 = (i0 + per < HW) ? i0 + per : HW;
      int i = i0;
      for (; i + 8 <= i1; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(int64_t)(i + u) * C];
#pragma unroll
        for (int u = 0; u < 8; u++) s += v[u];
      }
      for (; i < i1; i++) s += p[(int64_t)i * C];
    }
    spart[tid] = s;
    __syncthreads();
    if (tid < C) {
      float t = 0.f;
      for (int q = 0; q < parts; q++) t += spart[q * C + tid];
      t = t / (float)HW;
      sp[tid] = t;
      pooled[(int64_t)b * C + tid] = t;
    }
  }
  __syncthreads();
  if (tid < K) {
    float s = bias ? bias[tid] : 0.f;
    const float* w = W + (int64_t)tid * C;
    for (int c = 0; c < C; c++) s = __fmaf_rn(sp[c], w[c], s);
    sl[tid] = s;
    logits[(int64_t)b * K + tid] = s;
  }
  __syncthreads();
  if (tid < K) {
    float mx = sl[0];
    for (int j = 1; j < K; j++) mx = fmaxf(mx, sl[j]);
    float se = 0.f;
    for (int j = 0; j < K; j++) se += expf(sl[j] - mx);
    const float lse = mx + logf(se);
    probs[(int64_t)b * K + tid] = expf(sl[tid] - lse);
    if (tid == 0) {
      const int64_t y = target[b];
      loss[b] = (y >= 0 && y < K) ? lse - sl[y] : 0.f;
    }
  }
}

// blocks [0, B): dfeat[b][p][c] = (1/HW) sum_j dl[b][j] W[j][c] for every pixel p;  blocks [B, B+K): dW[j][:], dbias[j]
// with dl[b][j] = g * (probs[b][j] - [j == target_b]) / B   (mean reduction of the loss; g = upstream scalar)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ probs,
                                                       const int64_t* __restrict__ target, const float* __restrict__ pooled,
                                                       const float* __restrict__ W, int B, int HW, int C, int K,
                                                       float* __restrict__ dfeat, float* __restrict__ dW,
                                                       float* __restrict__ dbias) {
  __shared__ float sd[kMaxK];
  const int tid = threadIdx.x;
  const float gs = g[0] / (float)B;
  __shared__ float sdl[1024];
  __shared__ float spart[256];
  const int parts = 256 / C > 0 ? 256 / C : 1;
  const int c = tid % C, part = tid / C;
  if ((int)blockIdx.x < B) {
    const int b = blockIdx.x;
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
    const int j = blockIdx.x - B;
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

}  // namespace

extern "C" {

int alignq_head_ce_fwd(const float* feat, const float* W, const float* bias, const int64_t* target, int B, int HW, int C, int K,
                       float* pooled, float* logits, float* probs, float* loss, void* stream) {
  if (!feat || !W || !target || !pooled || !logits || !probs || !loss || B < 1 || HW < 1) return ALIGNQ_EINVAL;
  if (C < 1 || C > kMaxC || K < 1 || K > kMaxK) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(head_fwd_kernel, B, 256, 0, (hipStream_t)stream, feat, W, bias, target, HW, C, K, pooled, logits, probs,
                     loss);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

int alignq_head_ce_bwd(const float* g, const float* probs, const int64_t* target, const float* pooled, const float* W, int B,
                       int HW, int C, int K, float* dfeat, float* dW, float* dbias, void* stream) {
  if (!g || !probs || !target || !pooled || !W || !dfeat || !dW || B < 1 || HW < 1) return ALIGNQ_EINVAL;
  if (C < 1 || C > kMaxC || K < 1 || K > kMaxK) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(head_bwd_kernel, B + K, 256, 0, (hipStream_t)stream, g, probs, target, pooled, W, B, HW, C, K, dfeat, dW,
                     dbias);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
