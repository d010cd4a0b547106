// conv_kernels.hip — Conv2d_Q's convolution (reference: model/quantization.py:149-154, F.conv2d(input, weight_q, ...)) for the
// shapes that carry the ResNet-20/56 hot loop: 3x3, stride 1, padding 1, C_in == C_out == C in {16, 32, 64}, channels-last
// (torch.channels_last) fp32 tensors.  Forward, and the data gradient as the same kernel on the flipped / transposed filter.
//
// MI355X design: an implicit GEMM  Y[co][pixel] = sum_k W[co][k] * X[k][pixel],  k = (tap, ci), on v_mfma_f32_16x16x32_bf16
// (gfx950 has no TF32 and its fp32 MFMA rate is 1/16 of bf16), made EXACT in its products by using what Conv2d_Q knows:
//   * the filter is weight_quantize_fn's output, W_q = b / n with integer bins |b| <= n = 2^k - 1 <= 255: the A operand is
//     the integer b (exact in bf16's 8 significant bits), recovered as rint(W_q * n) and kept in registers, and the sum is
//     divided by n once at the end;
//   * an fp32 activation is split into THREE bf16 terms hi + mid + lo (8 + 8 + 8 significant bits: the split is exact),
//     staged in an LDS image of the input tile with its halo ([row][col][channel], zero padding materialised): one
//     ds_read_b128 per fragment (8 consecutive channels), three MFMAs per k step.
//   Every product b * x_term is exact in fp32 and the accumulation is fp32, so the result carries fp32 accumulation error
//   only (measured 2-4e-6 on outputs of magnitude 5-10, the level of MIOpen's fp32 kernels) at the bf16 matrix rate.
//   * C/D: lane = pixel (lane & 15), 4 consecutive output channels per lane -> one float4 store per lane.
// One workgroup (256 threads) = PT pixels (whole image rows) x all C output channels; wave -> (16-channel group, pixel groups).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/alignq.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// C channels, WD image width, PT pixels per workgroup (TR = PT / WD whole image rows, TR divides H so a tile never straddles
// two images).  LDS pixel stride is C + 8 bf16: the 16 lanes of a ds_read_b128 phase then hit 16 distinct 16-byte slots.
template <int C, int WD, int PT, bool DGRAD>
__global__ __launch_bounds__(256) void conv3x3_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int H, int total_rows, float nlev) {
  constexpr int TR = PT / WD;                 // image rows per workgroup
  constexpr int NS = (9 * C + 31) / 32;       // k steps of 32
  constexpr int NCG = C / 16;                 // 16-channel output groups
  constexpr int NPP = 4 / NCG;                // pixel partitions (waves per channel group)
  constexpr int NG = PT / 16;                 // 16-pixel groups per tile
  constexpr int LW = WD + 2;                  // LDS row: WD pixels + left/right zero padding
  constexpr int CP = C + 8;                   // padded pixel stride (bf16 elements)
  constexpr int LROWS = TR + 2;               // + halo row above / below
  constexpr int ARR = LROWS * LW * CP;        // bf16 elements per array
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * ARR];
  __bf16* Xhi = lds;
  __bf16* Xmi = lds + ARR;
  __bf16* Xlo = lds + 2 * ARR;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row0 = blockIdx.x * TR;           // first (image*H + h) row of this tile
  const int img_lo = (row0 / H) * H, img_hi = img_lo + H;       // rows of the tile's image
  // ---- stage the input tile (+ halo) as three bf16 terms; padding and out-of-image rows are zeros --------------------
  {
    constexpr int C4 = C / 4;
    constexpr int N4 = LROWS * LW * C4;        // float4 slots
    constexpr int NIT = (N4 + 255) / 256;
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {         // all loads in flight first (clamped address, masked below)
      const int i = tid + 256 * it;
      const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
      const int grow = row0 + lr - 1;
      const bool ok = i < N4 && col >= 1 && col <= WD && grow >= img_lo && grow < img_hi && grow < total_rows;
      const int64_t off = ok ? ((int64_t)grow * WD + (col - 1)) * C + 4 * c4 : 0;
      v[it] = *reinterpret_cast<const float4*>(x + off);
      if (!ok) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      if (i < N4) {
        const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
        bf16x4 h4, m4, l4;
        const float vv[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {          // exact three-way split: v == hi + mid + lo
          const __bf16 hi = (__bf16)vv[e];
          const float r1 = vv[e] - (float)hi;
          const __bf16 mi = (__bf16)r1;
          const float r2 = r1 - (float)mi;
          h4[e] = hi; m4[e] = mi; l4[e] = (__bf16)r2;
        }
        const int o = (lr * LW + col) * CP + 4 * c4;
        *reinterpret_cast<bf16x4*>(Xhi + o) = h4;
        *reinterpret_cast<bf16x4*>(Xmi + o) = m4;
        *reinterpret_cast<bf16x4*>(Xlo + o) = l4;
      }
    }
  }
  // ---- filter fragments of this wave's 16-channel group: A[m = out channel][k = (tap, in channel)] ------------------
  const int cog = wv % NCG, pp = wv / NCG;
  const int m = lane & 15, q = lane >> 4;
  bf16x8 ab[NS];               // integer bins of the quantised filter: exact in bf16
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int k0 = 32 * s + 8 * q;
    const int tap = k0 / C, c0 = k0 % C;
    float v[8];
    if (tap < 9) {
      if (!DGRAD) {          // forward: W[co][tap][ci], 8 consecutive ci
        const float* p = w + ((int64_t)(cog * 16 + m) * 9 + tap) * C + c0;
        const float4 a4 = *reinterpret_cast<const float4*>(p), b4 = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a4.x; v[1] = a4.y; v[2] = a4.z; v[3] = a4.w; v[4] = b4.x; v[5] = b4.y; v[6] = b4.z; v[7] = b4.w;
      } else {               // data gradient: A[m = ci][k = (tap', co)] = W[co][8 - tap'][ci]
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = w[((int64_t)(c0 + j) * 9 + (8 - tap)) * C + cog * 16 + m];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) v[j] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) ab[s][j] = (__bf16)rintf(v[j] * nlev);
  }
  __syncthreads();
  // ---- MFMA over this wave's pixel groups ----------------------------------------------------------------------------
  for (int g = pp; g < NG; g += NPP) {
    const int p = g * 16 + (lane & 15);          // pixel of the tile owned by this lane (B column)
    const int r = p / WD, c = p % WD;            // tile row / column
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int k0 = 32 * s + 8 * q;
      const int tap = k0 / C, c0 = k0 % C;
      bf16x8 bh, bm, bl;
      if (tap < 9) {
        const int ky = tap / 3, kx = tap % 3;
        const int o = ((r + ky) * LW + (c + kx)) * CP + c0;
        bh = *reinterpret_cast<const bf16x8*>(Xhi + o);
        bm = *reinterpret_cast<const bf16x8*>(Xmi + o);
        bl = *reinterpret_cast<const bf16x8*>(Xlo + o);
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) { bh[j] = (__bf16)0.f; bm[j] = (__bf16)0.f; bl[j] = (__bf16)0.f; }
      }
      // smallest terms first: the fp32 accumulator then loses the least
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bh, acc, 0, 0, 0);
    }
    const int grow = row0 + r;
    if (grow < total_rows)
      *reinterpret_cast<float4*>(y + ((int64_t)grow * WD + c) * C + cog * 16 + 4 * q) =
          make_float4(acc[0] / nlev, acc[1] / nlev, acc[2] / nlev, acc[3] / nlev);
  }
}

template <int C, int WD, int PT>
int launch(const float* x, const float* w, float* y, int B, int H, int dgrad, float nlev, hipStream_t st) {
  constexpr int TR = PT / WD;
  const int total_rows = B * H;
  if (H % TR) return ALIGNQ_EUNSUPPORTED;
  const int grid = total_rows / TR;
  if (dgrad) hipLaunchKernelGGL((conv3x3_nhwc_kernel<C, WD, PT, true>), grid, 256, 0, st, x, w, y, H, total_rows, nlev);
  else hipLaunchKernelGGL((conv3x3_nhwc_kernel<C, WD, PT, false>), grid, 256, 0, st, x, w, y, H, total_rows, nlev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

extern "C" {

// y[b,h,w,co] = sum x[b,h+ky-1,w+kx-1,ci] * wt[co,ky,kx,ci]   (dgrad = 0)
// dx[b,h,w,ci] = sum dy[b,h-ky+1,w-kx+1,co] * wt[co,ky,kx,ci] (dgrad = 1: x := dy)
// wt must hold k-bit quantised values b / (2^k - 1), 1 <= k <= 8 (weight_quantize_fn's output).
int alignq_conv3x3_nhwc(const float* x, const float* wt, float* y, int B, int H, int W, int C, int w_bit, int dgrad,
                        void* stream) {
  if (!x || !wt || !y || B < 1 || H < 1) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(y)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  if (C == 16 && W == 32) return launch<16, 32, 128>(x, wt, y, B, H, dgrad, nlev, st);
  if (C == 32 && W == 16) return launch<32, 16, 128>(x, wt, y, B, H, dgrad, nlev, st);
  if (C == 64 && W == 8) return launch<64, 8, 32>(x, wt, y, B, H, dgrad, nlev, st);
  return ALIGNQ_EUNSUPPORTED;
}

}  // extern "C"
