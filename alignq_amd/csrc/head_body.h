// head_body.h — the classifier head's forward and backward as device functions, shared by head_kernels.hip (alignq_head_ce_fwd /
// _bwd) and site4_kernels.hip (alignq_head_ce_bwd_site_prep: the backward as one ROLE of a launch that also prepares the sites' S
// matrices; alignq_site_reduce_loss_multi_head: the forward as one role of the launch that closes the sites' slab reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "alignq_math.h"

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


// one workgroup per sample: pooled[c] = mean_p feat[b][p][c]; logits = W pooled + bias; log-softmax; loss_b; probs
// `b`: index among the `nb` workgroups of this role (256 threads each; a role of a launch with wider workgroups lets only its first
// four waves in)
__device__ __forceinline__ void head_fwd_body(const float* __restrict__ feat, const float* __restrict__ W,
                                              const float* __restrict__ bias, const int64_t* __restrict__ target,
                                              int HW, int C, int K, float* __restrict__ pooled,
                                              float* __restrict__ logits, float* __restrict__ probs,
                                              float* __restrict__ loss, float* __restrict__ ce_mean,
                                              unsigned* __restrict__ counter, const float* __restrict__ site_scal,
                                              int n_sites, float* __restrict__ trans_total, const int b, const int nb) {
  __shared__ float sp[kMaxC];
  __shared__ float sl[kMaxK];
  const int tid = threadIdx.x;
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
        is_last = (tk == (unsigned)nb - 1);
      }
    }
  }
  if (!ce_mean) return;
  __syncthreads();
  if (!is_last || tid >= 64) return;
  {
    const int B = nb;
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


}  // namespace alignq_head
