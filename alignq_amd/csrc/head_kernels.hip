// head_kernels.hip — the classifier head of the training harness fused with its loss (channels-last features):
//   out = avgpool(out); out = out.view(B, -1); logits = logit(out)        (model/resnet.py:127-129 of the reference)
//   loss = CrossEntropyLoss()(logits, targets)                             (main.py: criterion)
// as ONE forward and ONE backward launch instead of ~12 small PyTorch / rocBLAS kernels (pool, addmm, log-softmax, nll and
// their backward GEMMs) that are pure launch latency at batch 128 x 64 features x 10 classes.  Plain fp32 arithmetic.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "head_body.h"

namespace {

using alignq_head::kMaxC;
using alignq_head::kMaxK;

// one workgroup per sample: pooled[c] = mean_p feat[b][p][c]; logits = W pooled + bias; log-softmax; loss_b; probs
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const int64_t* __restrict__ target,
                                                       int HW, int C, int K, float* __restrict__ pooled,
                                                       float* __restrict__ logits, float* __restrict__ probs,
                                                       float* __restrict__ loss, float* __restrict__ ce_mean,
                                                       unsigned* __restrict__ counter, const float* __restrict__ site_scal,
                                                       int n_sites, float* __restrict__ trans_total) {
  __shared__ float sp[kMaxC];
  __shared__ float sl[kMaxK];
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ float spart[256];
  __shared__ int is_last;
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
      const float lb = (y >= 0 && y < K) ? lse - sl[y] : 0.f;
      if (!ce_mean) {
        loss[b] = lb;
      } else {
        // mean over the batch (and the sum of the sites' trans losses) without a launch of their own: the workgroup whose
        // ticket is last adds the per-sample losses in index order.  Hand-off as in slab_reduce_body: write-through store,
        // drain, one relaxed agent-scope ticket; the last workgroup reads with agent-scope loads.
        __hip_atomic_store(&loss[b], lb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ALIGNQ_TICKET_ACQREL
        const unsigned tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
#else
        const unsigned tk = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
        is_last = (tk == gridDim.x - 1);
      }
    }
  }
  if (!ce_mean) return;
  __syncthreads();
  if (!is_last || tid >= 64) return;
  {
    const int B = gridDim.x;
    double s = 0.0;
    for (int i = tid; i < B; i += 64) s += (double)__hip_atomic_load(&loss[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = alignq::wave_sum_d_dpp(s);
    double t = 0.0;
    if (site_scal)
      for (int i = tid; i < n_sites; i += 64) t += (double)site_scal[4 * i];   // scal = {loss, c_con, 1/n, rms} per site
    t = alignq::wave_sum_d_dpp(t);
    if (tid == 0) {
      ce_mean[0] = (float)(s / (double)B);
      if (trans_total) trans_total[0] = (float)t;
      __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm
    }
  }
}

__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ g, const float* __restrict__ probs,
                                                       const int64_t* __restrict__ target, const float* __restrict__ pooled,
                                                       const float* __restrict__ W, int B, int HW, int C, int K,
                                                       float* __restrict__ dfeat, float* __restrict__ dW,
                                                       float* __restrict__ dbias) {
  alignq_head::head_bwd_body(g, probs, target, pooled, W, B, HW, C, K, dfeat, dW, dbias, blockIdx.x);
}

}  // namespace

extern "C" {

int alignq_head_ce_fwd(const float* feat, const float* W, const float* bias, const int64_t* target, int B, int HW, int C, int K,
                       float* pooled, float* logits, float* probs, float* loss, float* ce_mean, unsigned* counter,
                       const float* site_scal, int n_sites, float* trans_total, void* stream) {
  if (!feat || !W || !target || !pooled || !logits || !probs || !loss || B < 1 || HW < 1) return ALIGNQ_EINVAL;
  if ((ce_mean && !counter) || (site_scal && (!ce_mean || !trans_total || n_sites < 1))) return ALIGNQ_EINVAL;
  if (C < 1 || C > kMaxC || K < 1 || K > kMaxK) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(head_fwd_kernel, B, 256, 0, (hipStream_t)stream, feat, W, bias, target, HW, C, K, pooled, logits, probs,
                     loss, ce_mean, counter, site_scal, n_sites, trans_total);
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
