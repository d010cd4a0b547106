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

// one workgroup per sample: see head_fwd_body (head_body.h)
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                       const float* __restrict__ bias, const int64_t* __restrict__ target,
                                                       int HW, int C, int K, float* __restrict__ pooled,
                                                       float* __restrict__ logits, float* __restrict__ probs,
                                                       float* __restrict__ loss, float* __restrict__ ce_mean,
                                                       unsigned* __restrict__ counter, const float* __restrict__ site_scal,
                                                       int n_sites, float* __restrict__ trans_total) {
  alignq_head::head_fwd_body(feat, W, bias, target, HW, C, K, pooled, logits, probs, loss, ce_mean, counter, site_scal, n_sites,
                             trans_total, blockIdx.x, gridDim.x);
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
