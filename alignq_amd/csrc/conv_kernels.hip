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
#include <stdlib.h>

#include "../../include/alignq.h"
#include "site_internal.h"
#include "wgrad_reduce_body.h"

namespace {

using namespace alignq_wgr;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// 4 pixels x 16 channels block at `p` (this lane's row q = (lane & 15) >> 2, columns 4 * (lane & 3)), transposed: the lane
// receives channel (lane & 15) of the 4 pixels.  EXEC must be all ones (it is: no divergence around the calls).
__device__ __forceinline__ s16x4 tr_read(const __bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ bf16x8 join8(s16x4 a, s16x4 b) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}


// N2: the activation operand given as its integer level index instead of fp32 x_q.  xb = bytes per index (0: the
// tensor is fp32).  Returns 4 consecutive channels at ELEMENT offset `off` as floats: the raw index for xb != 0 (the kernels
// then divide the accumulated sum by the level count once: the products index x filter bin are exact integers).
template <int xb>
__device__ __forceinline__ f32x4 fetch_act4(const float* __restrict__ x, int64_t off) {
  if constexpr (xb == 0) return *reinterpret_cast<const f32x4*>(x + off);
  if constexpr (xb == 2) {
    const s16x4 b = *reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(x) + off);
    return (f32x4){(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
  const char4 b = *reinterpret_cast<const char4*>(reinterpret_cast<const signed char*>(x) + off);
  return (f32x4){(float)b.x, (float)b.y, (float)b.z, (float)b.w};
}

// C channels, WD image width, PT pixels per workgroup (TR = PT / WD whole image rows, TR divides H so a tile never straddles
// two images).  LDS pixel stride: conv_cp<C>() bf16 (below).
// The gradient w.r.t. a folded batch-norm's input, formed ON LOAD from the site backward's output g (gradient w.r.t. the BN
// output) instead of by a separate elementwise kernel:  dy = a[c] * (g - k0[c] - (z - mean[c]) * invstd[c] * k1[c]).
struct BnLazy {
  const float* z;      // the convolution's forward output (= BN input), same layout as g; nullptr => g already is dy
  const float* ab;     // [2][C]: a = gamma * invstd, (b)
  const float* save;   // [2][C]: batch mean, invstd
  const float* ktot;   // [2][C]: k0 = sum g / n, k1 = sum g * zhat / n   (or nullptr: formed from `part` by every workgroup)
  // the site backward's per-tile sums [n_tiles][min(C, tile_f)][2] = {sum g, sum g * zhat}; tile t covers the features
  // t * tile_f .. of the channels-last [pixel][C] row (channel = feature mod C)
  const float* part = nullptr;
  int n_tiles = 0, tile_f = 0;
  double inv_n = 1.0;          // 1 / (B * H * W)
  float* dgamma = nullptr;     // published by one workgroup (batch-norm parameter gradients = the totals)
  float* dbeta = nullptr;
};

// Per-channel totals of the site backward's per-tile sums, formed by EVERY workgroup (256 threads) for itself into LDS
// kt[2 * C] = {sum g / n, sum g * zhat / n}: the one-workgroup launch that otherwise sits between the site backward and the
// convolution backward (4.7 us per layer) is gone.  Thread -> (channel, group); up to 16 partials per thread in flight at
// once; double accumulation in a fixed order.  Call with all 256 threads; ends with a barrier.
template <int C>
__device__ __forceinline__ void bn_totals_lds(const BnLazy& t, float* __restrict__ kt, bool publish) {
  static_assert(256 % C == 0, "thread -> (channel, group)");
  __shared__ double red[256][2];
  constexpr int G = 256 / C;
  const int tid = threadIdx.x;
  const int cp = C < t.tile_f ? C : t.tile_f;            // channels per tile
  const int cyc = C / cp, cnt = t.n_tiles / cyc;         // tiles per channel cycle, partials per channel
  const int c = tid % C, grp = tid / C;
  const int e = c % cp, t_first = c / cp;
  double s0 = 0, s1 = 0;
  constexpr int U = 16;
  for (int i0 = grp; i0 < cnt; i0 += G * U) {
    float2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int i = i0 + u * G;
      const int ic = i < cnt ? i : cnt - 1;
      v[u] = *reinterpret_cast<const float2*>(t.part + ((int64_t)(ic * cyc + t_first) * cp + e) * 2);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (i0 + u * G < cnt) { s0 += v[u].x; s1 += v[u].y; }
    }
  }
  red[tid][0] = s0;
  red[tid][1] = s1;
  __syncthreads();
  if (tid < C) {
    double t0 = 0, t1 = 0;
#pragma unroll
    for (int g = 0; g < G; g++) { t0 += red[g * C + tid][0]; t1 += red[g * C + tid][1]; }
    kt[tid] = (float)(t0 * t.inv_n);        // (a double division costs about a microsecond on the critical path here)
    kt[C + tid] = (float)(t1 * t.inv_n);
    if (publish) {
      if (t.dbeta) t.dbeta[tid] = (float)t0;
      if (t.dgamma) t.dgamma[tid] = (float)t1;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float4 bn_lazy4(const float4& g, const float4& z, const float4& a, const float4& m, const float4& is,
                                           const float4& k0, const float4& k1) {
  float4 o;
  o.x = a.x * (g.x - k0.x - (z.x - m.x) * is.x * k1.x);
  o.y = a.y * (g.y - k0.y - (z.y - m.y) * is.y * k1.y);
  o.z = a.z * (g.z - k0.z - (z.z - m.z) * is.z * k1.z);
  o.w = a.w * (g.w - k0.w - (z.w - m.w) * is.w * k1.w);
  return o;
}

// bf16 elements per pixel of the LDS image.  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
// (+32): with 16 consecutive pixels per 16 lanes and 16 bytes per k quarter, 32-byte pixels (C = 16, no padding) and 96-byte
// pixels (C = 32, 32 bytes of padding) put the sixteen pieces of a group on sixteen different slots of the 256-byte bank row
// (tools: the bank model of qgemm_kernels.hip's LDX); C + 8 costs two to four extra cycles per read.
template <int C> constexpr int conv_cp() { return C == 16 ? 16 : (C == 32 ? 48 : C + 8); }
template <int C, int WD, int PT>
struct ConvLds {
  static constexpr int kBf16 = 3 * ((PT / WD) + 2) * (WD + 2) * conv_cp<C>();      // three bf16 images of the tile with halo
};

template <int C, int WD, int PT, bool DGRAD, int XB = 0>
__device__ __forceinline__ void conv3x3_body(const float* __restrict__ x, const float* __restrict__ w,
                                             float* __restrict__ y, int H, int total_rows, float nlev, __bf16* lds,
                                             int block, const float* __restrict__ add,
                                             float* __restrict__ bn_part = nullptr, int n_wg = 0,
                                             BnLazy lazy = BnLazy{nullptr, nullptr, nullptr, nullptr}, float xlev = 1.0f) {
  // XB != 0 (forward only; a template parameter so that the fp32 form carries no trace of it): x holds int16 / int8 level
  // indices idx (value idx / xlev): two exact bf16 terms instead of three, the result is divided by nlev * xlev
  static_assert(XB == 0 || !DGRAD, "the data gradient reads fp32 dy");
  constexpr int xb = XB;
  constexpr int TR = PT / WD;                 // image rows per workgroup
  constexpr int NS = (9 * C + 31) / 32;       // k steps of 32
  constexpr int NCG = C / 16;                 // 16-channel output groups
  constexpr int NPP = 4 / NCG;                // pixel partitions (waves per channel group)
  constexpr int NG = PT / 16;                 // 16-pixel groups per tile
  constexpr int LW = WD + 2;                  // LDS row: WD pixels + left/right zero padding
  constexpr int CP = conv_cp<C>();            // pixel stride (bf16 elements)
  constexpr int LROWS = TR + 2;               // + halo row above / below
  constexpr int ARR = LROWS * LW * CP;        // bf16 elements per array
  static_assert(3 * ARR == ConvLds<C, WD, PT>::kBf16, "LDS size");
  __bf16* Xhi = lds;
  __bf16* Xmi = lds + ARR;
  __bf16* Xlo = lds + 2 * ARR;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row0 = block * TR;                // first (image*H + h) row of this tile
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
      const f32x4 ld = fetch_act4<XB>(x, off);
      v[it] = ok ? make_float4(ld[0], ld[1], ld[2], ld[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (DGRAD && lazy.z) {       // x is g: turn it into the batch-norm input gradient here (zeros stay zeros: padding)
      float4 zz[NIT];
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int i = tid + 256 * it;
        const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
        const int grow = row0 + lr - 1;
        const bool ok = i < N4 && col >= 1 && col <= WD && grow >= img_lo && grow < img_hi && grow < total_rows;
        zz[it] = *reinterpret_cast<const float4*>(lazy.z + (ok ? ((int64_t)grow * WD + (col - 1)) * C + 4 * c4 : 0));
      }
      const int c4t = tid % C4;             // 256 % C4 == 0: a thread always handles the same channel quad
      const float4 a4 = *reinterpret_cast<const float4*>(lazy.ab + 4 * c4t);
      const float4 m4 = *reinterpret_cast<const float4*>(lazy.save + 4 * c4t);
      const float4 i4 = *reinterpret_cast<const float4*>(lazy.save + C + 4 * c4t);
      float4 k0, k1;
      if (lazy.part) {           // (workgroup-uniform) totals from the site backward's per-tile sums, every workgroup for itself
        __shared__ __attribute__((aligned(16))) float kt[2 * C];
        bn_totals_lds<C>(lazy, kt, block == 0);
        k0 = *reinterpret_cast<const float4*>(kt + 4 * c4t);
        k1 = *reinterpret_cast<const float4*>(kt + C + 4 * c4t);
      } else {
        k0 = *reinterpret_cast<const float4*>(lazy.ktot + 4 * c4t);
        k1 = *reinterpret_cast<const float4*>(lazy.ktot + C + 4 * c4t);
      }
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int i = tid + 256 * it;
        const int col = (i / C4) % LW, lr = i / (C4 * LW);
        const int grow = row0 + lr - 1;
        const bool ok = i < N4 && col >= 1 && col <= WD && grow >= img_lo && grow < img_hi && grow < total_rows;
        if (ok) v[it] = bn_lazy4(v[it], zz[it], a4, m4, i4, k0, k1);
      }
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
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};   // batch-norm statistics of this lane's outputs
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
      // smallest terms first: the fp32 accumulator then loses the least (an integer index is exact in two terms: its third
      // term is zero and its MFMA is skipped, a wave-uniform branch)
      if (DGRAD || xb == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bh, acc, 0, 0, 0);
    }
    const int grow = row0 + r;
    if (grow < total_rows) {
      const int64_t o = ((int64_t)grow * WD + c) * C + cog * 16 + 4 * q;
      const float den = (!DGRAD && xb != 0) ? nlev * xlev : nlev;        // integers <= 255 * 65535: exact in fp32 up to 2^24
      float4 v = make_float4(acc[0] / den, acc[1] / den, acc[2] / den, acc[3] / den);
      if (add) {       // e.g. the identity shortcut's gradient joining the data gradient (saves an accumulation kernel)
        const float4 r = *reinterpret_cast<const float4*>(add + o);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
      }
      *reinterpret_cast<float4*>(y + o) = v;
      bs[0] += v.x; bs[1] += v.y; bs[2] += v.z; bs[3] += v.w;
      bq[0] += v.x * v.x; bq[1] += v.y * v.y; bq[2] += v.z * v.z; bq[3] += v.w * v.w;
    }
  }
  // ---- optional epilogue: per-workgroup, per-channel {sum z, sum z^2} of the tile just produced, so that the batch-norm
  //      that follows needs no pass of its own over z.  Row (16-lane) sums on the DPP path, waves through LDS, fixed order.
  if (!DGRAD && bn_part) {
    float* red = reinterpret_cast<float*>(lds);
#define ROW_SHR_ADD(V, CTRL) V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, 0xf, 0xf, false))
#pragma unroll
    for (int e = 0; e < 4; e++) {                // row_shr 1, 2, 4, 8: lane 15 of each 16-lane row ends with the row total
      ROW_SHR_ADD(bs[e], 0x111); ROW_SHR_ADD(bs[e], 0x112); ROW_SHR_ADD(bs[e], 0x114); ROW_SHR_ADD(bs[e], 0x118);
      ROW_SHR_ADD(bq[e], 0x111); ROW_SHR_ADD(bq[e], 0x112); ROW_SHR_ADD(bq[e], 0x114); ROW_SHR_ADD(bq[e], 0x118);
    }
#undef ROW_SHR_ADD
    __syncthreads();                             // every wave is done with the LDS image
    if ((lane & 15) == 15) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        red[(wv * 16 + 4 * q + e) * 2] = bs[e];
        red[(wv * 16 + 4 * q + e) * 2 + 1] = bq[e];
      }
    }
    __syncthreads();
    if (tid < C) {                               // channel tid: group tid / 16, summed over that group's NPP waves
      const int cg = tid / 16, cl = tid % 16;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int ppi = 0; ppi < NPP; ppi++) {
        const int w2 = ppi * NCG + cg;
        s0 += red[(w2 * 16 + cl) * 2];
        s1 += red[(w2 * 16 + cl) * 2 + 1];
      }
      bn_part[((int64_t)tid * n_wg + block) * 2] = s0;
      bn_part[((int64_t)tid * n_wg + block) * 2 + 1] = s1;
    }
  }
}

template <int C, int WD, int PT, bool DGRAD, int XB = 0>
__global__ __launch_bounds__(256) void conv3x3_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ y, int H, int total_rows, float nlev,
                                                           const float* __restrict__ add, float* __restrict__ bn_part,
                                                           float xlev) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[ConvLds<C, WD, PT>::kBf16];
  conv3x3_body<C, WD, PT, DGRAD, XB>(x, w, y, H, total_rows, nlev, lds, blockIdx.x, add, bn_part, gridDim.x,
                                     BnLazy{nullptr, nullptr, nullptr, nullptr}, xlev);
}

// ---- forward of the remaining Conv2d_Q shapes of the ResNet body: stride 2 (3x3, padding 1) and the 1x1 stride-2 shortcut
// convolutions, C_in != C_out.  Same scheme as conv3x3_body (integer filter bins x three exact bf16 terms of the activation,
// 16x16x32 bf16 MFMA, optional batch-norm partial statistics in the epilogue); the LDS image holds the input rows / columns the
// tile's taps touch ((KS = 3) all of them with halo; (KS = 1) only the strided pixels) and a lane's B address is
// (S*r + ky, S*c + kx).  H is the OUTPUT height, the input is [B, S*H, WDI, CIN], the output [B, H, WDI / S, COUT].
template <int CIN, int COUT, int WDI, int KS, int S, int PT>
struct ConvGen {
  static constexpr int WDO = WDI / S, TR = PT / WDO;
  static constexpr int LROWS = KS == 3 ? TR * S + (S == 1 ? 2 : 1) : TR;
  static constexpr int LW = KS == 3 ? WDI + 2 : WDO;
  static constexpr int CP = CIN + 8;
  static constexpr int ARR = LROWS * LW * CP;
};

// block / n_wg: index and count of the workgroups of THIS convolution (a launch may carry several roles)
template <int CIN, int COUT, int WDI, int KS, int S, int PT>
__device__ __forceinline__ void convgen_fwd_body(const float* __restrict__ x, const float* __restrict__ w,
                                                 float* __restrict__ y, int H, int total_rows, float nlev,
                                                 float* __restrict__ bn_part, __bf16* lds, int block, int n_wg) {
  using G = ConvGen<CIN, COUT, WDI, KS, S, PT>;
  constexpr int WDO = G::WDO, TR = G::TR, LROWS = G::LROWS, LW = G::LW, CP = G::CP, ARR = G::ARR;
  constexpr int NS = (KS * KS * CIN + 31) / 32;
  constexpr int NCG = COUT / 16, NPP = 4 / NCG, NG = PT / 16;
  static_assert(NCG == 1 || NCG == 2 || NCG == 4, "output channels");
  __bf16* Xhi = lds;
  __bf16* Xmi = lds + ARR;
  __bf16* Xlo = lds + 2 * ARR;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row0 = block * TR;                                   // first OUTPUT row (image*H + h) of the tile
  const int Hin = H * S;
  const int img_lo = (row0 / H) * Hin, img_hi = img_lo + Hin;    // INPUT rows of the tile's image
  {
    constexpr int C4 = CIN / 4;
    constexpr int N4 = LROWS * LW * C4;
    constexpr int NIT = (N4 + 255) / 256;
    float4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
      int grow, gcol;
      if (KS == 3) { grow = row0 * S - 1 + lr; gcol = col - 1; }
      else { grow = (row0 + lr) * S; gcol = col * S; }
      const bool ok = i < N4 && gcol >= 0 && gcol < WDI && grow >= img_lo && grow < img_hi;
      v[it] = *reinterpret_cast<const float4*>(x + (ok ? ((int64_t)grow * WDI + gcol) * CIN + 4 * c4 : 0));
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
        for (int e = 0; e < 4; e++) {
          const __bf16 hi = (__bf16)vv[e];
          const float r1 = vv[e] - (float)hi;
          const __bf16 mi = (__bf16)r1;
          h4[e] = hi; m4[e] = mi; l4[e] = (__bf16)(r1 - (float)mi);
        }
        const int o = (lr * LW + col) * CP + 4 * c4;
        *reinterpret_cast<bf16x4*>(Xhi + o) = h4;
        *reinterpret_cast<bf16x4*>(Xmi + o) = m4;
        *reinterpret_cast<bf16x4*>(Xlo + o) = l4;
      }
    }
  }
  const int cog = wv % NCG, pp = wv / NCG;
  const int m = lane & 15, q = lane >> 4;
  bf16x8 ab[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int k0 = 32 * s + 8 * q;
    const int tap = k0 / CIN, c0 = k0 % CIN;
    if (tap < KS * KS) {
      const float* p = w + ((int64_t)(cog * 16 + m) * (KS * KS) + tap) * CIN + c0;
      const float4 a4 = *reinterpret_cast<const float4*>(p), b4 = *reinterpret_cast<const float4*>(p + 4);
      const float v[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
#pragma unroll
      for (int j = 0; j < 8; j++) ab[s][j] = (__bf16)rintf(v[j] * nlev);
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) ab[s][j] = (__bf16)0.f;
    }
  }
  __syncthreads();
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = pp; g < NG; g += NPP) {
    const int p = g * 16 + (lane & 15);
    const int r = p / WDO, c = p % WDO;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int k0 = 32 * s + 8 * q;
      const int tap = k0 / CIN, c0 = k0 % CIN;
      bf16x8 bh, bm, bl;
      if (tap < KS * KS) {
        const int ky = tap / KS, kx = tap % KS;
        const int o = (KS == 3 ? ((r * S + ky) * LW + (c * S + kx)) : (r * LW + c)) * CP + c0;
        bh = *reinterpret_cast<const bf16x8*>(Xhi + o);
        bm = *reinterpret_cast<const bf16x8*>(Xmi + o);
        bl = *reinterpret_cast<const bf16x8*>(Xlo + o);
      } else {
#pragma unroll
        for (int j = 0; j < 8; j++) { bh[j] = (__bf16)0.f; bm[j] = (__bf16)0.f; bl[j] = (__bf16)0.f; }
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bh, acc, 0, 0, 0);
    }
    const int grow = row0 + r;
    if (grow < total_rows) {
      const float4 v = make_float4(acc[0] / nlev, acc[1] / nlev, acc[2] / nlev, acc[3] / nlev);
      *reinterpret_cast<float4*>(y + ((int64_t)grow * WDO + c) * COUT + cog * 16 + 4 * q) = v;
      bs[0] += v.x; bs[1] += v.y; bs[2] += v.z; bs[3] += v.w;
      bq[0] += v.x * v.x; bq[1] += v.y * v.y; bq[2] += v.z * v.z; bq[3] += v.w * v.w;
    }
  }
  if (bn_part) {       // per-workgroup per-channel {sum y, sum y^2} for the batch-norm that follows (see conv3x3_body)
    float* red = reinterpret_cast<float*>(lds);
#define ROW_SHR_ADD(V, CTRL) V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, 0xf, 0xf, false))
#pragma unroll
    for (int e = 0; e < 4; e++) {
      ROW_SHR_ADD(bs[e], 0x111); ROW_SHR_ADD(bs[e], 0x112); ROW_SHR_ADD(bs[e], 0x114); ROW_SHR_ADD(bs[e], 0x118);
      ROW_SHR_ADD(bq[e], 0x111); ROW_SHR_ADD(bq[e], 0x112); ROW_SHR_ADD(bq[e], 0x114); ROW_SHR_ADD(bq[e], 0x118);
    }
#undef ROW_SHR_ADD
    __syncthreads();
    if ((lane & 15) == 15) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        red[(wv * 16 + 4 * q + e) * 2] = bs[e];
        red[(wv * 16 + 4 * q + e) * 2 + 1] = bq[e];
      }
    }
    __syncthreads();
    if (tid < COUT) {
      const int cg = tid / 16, cl = tid % 16;
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int ppi = 0; ppi < NPP; ppi++) {
        s0 += red[((ppi * NCG + cg) * 16 + cl) * 2];
        s1 += red[((ppi * NCG + cg) * 16 + cl) * 2 + 1];
      }
      bn_part[((int64_t)tid * n_wg + block) * 2] = s0;
      bn_part[((int64_t)tid * n_wg + block) * 2 + 1] = s1;
    }
  }
}

template <int CIN, int COUT, int WDI, int KS, int S, int PT>
__global__ __launch_bounds__(256) void convgen_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int H, int total_rows, float nlev,
                                                          float* __restrict__ bn_part) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * ConvGen<CIN, COUT, WDI, KS, S, PT>::ARR];
  convgen_fwd_body<CIN, COUT, WDI, KS, S, PT>(x, w, y, H, total_rows, nlev, bn_part, lds, blockIdx.x, gridDim.x);
}

// Both convolutions of a transition block (3x3 and 1x1, stride 2, same input) in ONE launch: the first n3 workgroups take
// the 3x3 role, the rest the 1x1 role (a launch boundary costs more than the 1x1 convolution itself).
template <int CIN, int COUT, int WDI, int PT3, int PT1>
__global__ __launch_bounds__(256) void transition_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w3,
                                                             const float* __restrict__ w1, float* __restrict__ y3,
                                                             float* __restrict__ y1, int H, int total_rows, float nlev,
                                                             float* __restrict__ part3, float* __restrict__ part1, int n3) {
  constexpr int A3 = ConvGen<CIN, COUT, WDI, 3, 2, PT3>::ARR, A1 = ConvGen<CIN, COUT, WDI, 1, 2, PT1>::ARR;
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * (A3 > A1 ? A3 : A1)];
  if ((int)blockIdx.x < n3)
    convgen_fwd_body<CIN, COUT, WDI, 3, 2, PT3>(x, w3, y3, H, total_rows, nlev, part3, lds, blockIdx.x, n3);
  else
    convgen_fwd_body<CIN, COUT, WDI, 1, 2, PT1>(x, w1, y1, H, total_rows, nlev, part1, lds, blockIdx.x - n3,
                                                (int)gridDim.x - n3);
}

// ---- data gradient of the stride-2 transition convolutions -----------------------------------------------------------
// dx[ih, iw, ci] = sum_{ky, kx, co} dy[(ih + P - ky) / 2, (iw + P - kx) / 2, co] * W[co][ky][kx][ci]  over the taps whose
// parities match (P = padding: 1 for 3x3, 0 for 1x1).  Implicit GEMM with k = (tap, co): A[m = ci][k] = W[co][tap][ci] (integer
// bins, registers), B[k][n = input pixel] = the dy pixel that tap reaches, or a zero pixel of the LDS image when the parity
// does not match / the position lies outside (address select, no branch).  dy may arrive in the lazy batch-norm form.
// H is the INPUT height (rows of dx), WDI its width; dy is [.., H/2, WDI/2, COUT].  PT input pixels (an even number of rows).
// WITH1 (KS == 3): the data gradient of the block's 1x1 stride-2 shortcut convolution (same input, own output gradient dy1,
// filter w1 and lazy batch-norm record) is accumulated in the same pass as a tenth tap (it reaches input pixel (2i, 2j) only,
// from dy1 pixel (i, j)): its output pixels sit behind the 3x3 image in the LDS arrays.
template <int CIN, int COUT, int WDI, int KS, int PT, bool WITH1>
__device__ __forceinline__ void dgrad_s2_body(const float* __restrict__ dy, const float* __restrict__ w,
                                              float* __restrict__ dx, int H, int total_rows, float nlev,
                                              const float* __restrict__ add, BnLazy lazy, __bf16* lds, int block,
                                              const float* __restrict__ dy1 = nullptr, const float* __restrict__ w1 = nullptr,
                                              BnLazy lazy1 = BnLazy{nullptr, nullptr, nullptr, nullptr}) {
  static_assert(!WITH1 || KS == 3, "the second source is the 1x1 shortcut of a 3x3 transition");
  constexpr int P = KS == 3 ? 1 : 0;
  constexpr int WDO = WDI / 2, TR = PT / WDI;
  constexpr int DROWS = TR / 2 + P, DCOLS = WDO + P;       // dy rows / columns the tile's taps can reach
  constexpr int CP = COUT + 8;
  constexpr int NPIX3 = DROWS * DCOLS + 1;                 // + one all-zero pixel
  constexpr int NPIX1 = WITH1 ? (TR / 2) * WDO : 0;        // the 1x1 convolution's dy pixels of the tile
  constexpr int NPIX = NPIX3 + NPIX1;
  constexpr int ARR = NPIX * CP;
  constexpr int NT = KS * KS + (WITH1 ? 1 : 0);
  constexpr int NS = (NT * COUT + 31) / 32;
  constexpr int NCG = CIN / 16, NPP = 4 / NCG, NG = PT / 16;
  __bf16* Xhi = lds;
  __bf16* Xmi = lds + ARR;
  __bf16* Xlo = lds + 2 * ARR;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row0 = block * TR;                              // first INPUT row (image*H + ih), even
  const int Ho = H / 2;
  const int orow0 = (row0 / H) * Ho + (row0 % H) / 2;       // first dy row the tile reaches (ky = P)
  const int oimg_hi = (row0 / H + 1) * Ho;                  // end of the image's dy rows
  {
    constexpr int C4 = COUT / 4;
    constexpr int N4 = NPIX3 * C4;
    constexpr int NIT = (N4 + 255) / 256;
    constexpr int M4 = NPIX1 * C4;
    constexpr int NI1 = WITH1 ? (M4 + 255) / 256 : 1;
    float4 v[NIT], zz[NIT], v1[NI1], z1[NI1];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int i = tid + 256 * it;
      const int c4 = i % C4, pix = i / C4;
      const int lr = pix / DCOLS, col = pix % DCOLS;
      const int grow = orow0 + lr;
      const bool ok = i < N4 && pix < DROWS * DCOLS && col < WDO && grow < oimg_hi;
      const int64_t off = ok ? ((int64_t)grow * WDO + col) * COUT + 4 * c4 : 0;
      v[it] = *reinterpret_cast<const float4*>(dy + off);
      if (lazy.z) zz[it] = *reinterpret_cast<const float4*>(lazy.z + off);
      if (!ok) v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if constexpr (WITH1) {
#pragma unroll
      for (int it = 0; it < NI1; it++) {       // rows orow0 .. orow0 + TR/2 - 1 of dy1: always inside the image
        const int i = tid + 256 * it;
        const int64_t off = i < M4 ? ((int64_t)orow0 * WDO) * COUT + 4 * (int64_t)i : 0;
        v1[it] = *reinterpret_cast<const float4*>(dy1 + off);
        if (lazy1.z) z1[it] = *reinterpret_cast<const float4*>(lazy1.z + off);
      }
    }
    const int c4t = tid % C4;
    if (lazy.z) {
      const float4 a4 = *reinterpret_cast<const float4*>(lazy.ab + 4 * c4t);
      const float4 m4 = *reinterpret_cast<const float4*>(lazy.save + 4 * c4t);
      const float4 i4 = *reinterpret_cast<const float4*>(lazy.save + COUT + 4 * c4t);
      float4 k0, k1;
      if (lazy.part) {           // (workgroup-uniform) see bn_totals_lds
        __shared__ __attribute__((aligned(16))) float kt[2 * COUT];
        bn_totals_lds<COUT>(lazy, kt, block == 0);
        k0 = *reinterpret_cast<const float4*>(kt + 4 * c4t);
        k1 = *reinterpret_cast<const float4*>(kt + COUT + 4 * c4t);
      } else {
        k0 = *reinterpret_cast<const float4*>(lazy.ktot + 4 * c4t);
        k1 = *reinterpret_cast<const float4*>(lazy.ktot + COUT + 4 * c4t);
      }
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int i = tid + 256 * it;
        const int pix = i / C4;
        const int lr = pix / DCOLS, col = pix % DCOLS;
        const bool ok = i < N4 && pix < DROWS * DCOLS && col < WDO && orow0 + lr < oimg_hi;
        if (ok) v[it] = bn_lazy4(v[it], zz[it], a4, m4, i4, k0, k1);
      }
    }
    if constexpr (WITH1) {
      if (lazy1.z) {
        const float4 a4 = *reinterpret_cast<const float4*>(lazy1.ab + 4 * c4t);
        const float4 m4 = *reinterpret_cast<const float4*>(lazy1.save + 4 * c4t);
        const float4 i4 = *reinterpret_cast<const float4*>(lazy1.save + COUT + 4 * c4t);
        float4 k0, k1;
        if (lazy1.part) {
          __shared__ __attribute__((aligned(16))) float kt1[2 * COUT];
          bn_totals_lds<COUT>(lazy1, kt1, block == 0);
          k0 = *reinterpret_cast<const float4*>(kt1 + 4 * c4t);
          k1 = *reinterpret_cast<const float4*>(kt1 + COUT + 4 * c4t);
        } else {
          k0 = *reinterpret_cast<const float4*>(lazy1.ktot + 4 * c4t);
          k1 = *reinterpret_cast<const float4*>(lazy1.ktot + COUT + 4 * c4t);
        }
#pragma unroll
        for (int it = 0; it < NI1; it++) {
          if (tid + 256 * it < M4) v1[it] = bn_lazy4(v1[it], z1[it], a4, m4, i4, k0, k1);
        }
      }
    }
#pragma unroll
    for (int it = 0; it < NIT + (WITH1 ? NI1 : 0); it++) {
      const bool second = it >= NIT;
      const int i = tid + 256 * (second ? it - NIT : it);
      if (i < (second ? M4 : N4)) {
        const int c4 = i % C4, pix = (second ? NPIX3 : 0) + i / C4;
        bf16x4 h4, m4, l4;
        const float4 src = second ? v1[second ? it - NIT : 0] : v[second ? 0 : it];
        const float vv[4] = {src.x, src.y, src.z, src.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const __bf16 hi = (__bf16)vv[e];
          const float r1 = vv[e] - (float)hi;
          const __bf16 mi = (__bf16)r1;
          h4[e] = hi; m4[e] = mi; l4[e] = (__bf16)(r1 - (float)mi);
        }
        const int o = pix * CP + 4 * c4;
        *reinterpret_cast<bf16x4*>(Xhi + o) = h4;
        *reinterpret_cast<bf16x4*>(Xmi + o) = m4;
        *reinterpret_cast<bf16x4*>(Xlo + o) = l4;
      }
    }
  }
  const int cig = wv % NCG, pp = wv / NCG;
  const int m = lane & 15, q = lane >> 4;
  bf16x8 ab[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int k0 = 32 * s + 8 * q;
    const int tap = k0 / COUT, c0 = k0 % COUT;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float v = 0.f;
      if (tap < KS * KS) v = w[((int64_t)(c0 + j) * (KS * KS) + tap) * CIN + cig * 16 + m];
      else if (WITH1 && tap == KS * KS) v = w1[(int64_t)(c0 + j) * CIN + cig * 16 + m];
      ab[s][j] = (__bf16)rintf(v * nlev);
    }
  }
  __syncthreads();
  for (int g = pp; g < NG; g += NPP) {
    const int p = g * 16 + (lane & 15);
    const int r = p / WDI, c = p % WDI;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int k0 = 32 * s + 8 * q;
      const int tap = k0 / COUT, c0 = k0 % COUT;
      int pix = DROWS * DCOLS;                              // the zero pixel
      if (tap < KS * KS) {
        const int ky = tap / KS, kx = tap % KS;
        const int a = r + P - ky, b = c + P - kx;           // 2 * (dy row - orow0 ... ) when even and >= 0
        if (a >= 0 && b >= 0 && !(a & 1) && !(b & 1)) pix = (a >> 1) * DCOLS + (b >> 1);
      } else if (WITH1 && tap == KS * KS) {
        if (!(r & 1) && !(c & 1)) pix = NPIX3 + (r >> 1) * WDO + (c >> 1);
      }
      const int o = pix * CP + c0;
      const bf16x8 bh = *reinterpret_cast<const bf16x8*>(Xhi + o);
      const bf16x8 bm = *reinterpret_cast<const bf16x8*>(Xmi + o);
      const bf16x8 bl = *reinterpret_cast<const bf16x8*>(Xlo + o);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[s], bh, acc, 0, 0, 0);
    }
    const int grow = row0 + r;
    if (grow < total_rows) {
      const int64_t o = ((int64_t)grow * WDI + c) * CIN + cig * 16 + 4 * q;
      float4 out = make_float4(acc[0] / nlev, acc[1] / nlev, acc[2] / nlev, acc[3] / nlev);
      if (add) {            // a second gradient w.r.t. the same input (the block's other branch): summed here
        const float4 a4 = *reinterpret_cast<const float4*>(add + o);
        out.x += a4.x; out.y += a4.y; out.z += a4.z; out.w += a4.w;
      }
      *reinterpret_cast<float4*>(dx + o) = out;
    }
  }
}

template <int CIN, int COUT, int WDI, int KS, int PT, bool WITH1>
struct DgradS2Lds {
  static constexpr int P = KS == 3 ? 1 : 0, TR = PT / WDI;
  static constexpr int kBf16 = 3 * (((TR / 2 + P) * (WDI / 2 + P) + 1) + (WITH1 ? (TR / 2) * (WDI / 2) : 0)) * (COUT + 8);
};

template <int CIN, int COUT, int WDI, int KS, int PT>
__global__ __launch_bounds__(256) void dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                       float* __restrict__ dx, int H, int total_rows, float nlev,
                                                       const float* __restrict__ add, BnLazy lazy) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[DgradS2Lds<CIN, COUT, WDI, KS, PT, false>::kBf16];
  dgrad_s2_body<CIN, COUT, WDI, KS, PT, false>(dy, w, dx, H, total_rows, nlev, add, lazy, lds, blockIdx.x);
}

template <int CIN, int COUT, int WDI, int KS, int PT>
int launch_dgrad_s2(const float* dy, const float* w, float* dx, int B, int H, float nlev, const float* add, BnLazy lazy,
                    hipStream_t st) {
  constexpr int TR = PT / WDI;
  static_assert(TR % 2 == 0, "even number of input rows per tile");
  if (H % TR) return ALIGNQ_EUNSUPPORTED;
  const int total_rows = B * H;
  hipLaunchKernelGGL((dgrad_s2_kernel<CIN, COUT, WDI, KS, PT>), total_rows / TR, 256, 0, st, dy, w, dx, H, total_rows, nlev,
                     add, lazy);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// ---- the stem: 3x3, stride 1, padding 1, 3 input channels -> 16 ---------------------------------------------------------
// K = 27 fits ONE 32-wide k step, so the tile is staged as an im2col image [pixel][32] (k = tap * 3 + ci, k >= 27 zero) of
// three bf16 terms; forward: B fragments are plain ds_read_b128 of it (A = integer filter bins [16][32]); filter gradient:
// dW[co][k] = sum_pixels dy[p][co] * col[p][k] with the pixel as MFMA k index, both operands through ds_read_b64_tr_b16 as in
// wgrad_body (two 16-column blocks of col).  x [B,H,32,3] (channels-last storage of [B,3,H,32]), wt [16][3][3][3], y [B,H,32,16].
constexpr int kStemPT = 128;                       // pixels per tile: 4 rows of 32
__device__ __forceinline__ void stem_stage(const float* __restrict__ x, int H, int row0, __bf16* Chi, __bf16* Cmi, __bf16* Clo) {
  // thread -> (pixel, tap-triple): 128 pixels x 2 halves of the 32-wide k vector (k 0..15, 16..31)
  const int tid = threadIdx.x;
  const int pix = tid >> 1, half = tid & 1;
  const int r = pix / 32, c = pix % 32;
  const int grow = row0 + r;
  const int ih = grow % H;
  float v[16];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const int k = 16 * half + j;
    const int tap = k / 3, ci = k % 3;
    const int ky = tap / 3, kx = tap % 3;
    const int yy = ih + ky - 1, xx = c + kx - 1;
    const bool ok = k < 27 && yy >= 0 && yy < H && xx >= 0 && xx < 32;
    v[j] = x[ok ? ((int64_t)(grow + ky - 1) * 32 + xx) * 3 + ci : 0];
    if (!ok) v[j] = 0.f;
  }
  bf16x8 h[2], m[2], l[2];
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const __bf16 hi = (__bf16)v[j];
    const float r1 = v[j] - (float)hi;
    const __bf16 mi = (__bf16)r1;
    h[j >> 3][j & 7] = hi; m[j >> 3][j & 7] = mi; l[j >> 3][j & 7] = (__bf16)(r1 - (float)mi);
  }
  const int o = pix * 32 + 16 * half;
  *reinterpret_cast<bf16x8*>(Chi + o) = h[0]; *reinterpret_cast<bf16x8*>(Chi + o + 8) = h[1];
  *reinterpret_cast<bf16x8*>(Cmi + o) = m[0]; *reinterpret_cast<bf16x8*>(Cmi + o + 8) = m[1];
  *reinterpret_cast<bf16x8*>(Clo + o) = l[0]; *reinterpret_cast<bf16x8*>(Clo + o + 8) = l[1];
}

__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int H, int total_rows, float nlev,
                                                       float* __restrict__ bn_part) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * kStemPT * 32];
  __bf16* Chi = lds;
  __bf16* Cmi = lds + kStemPT * 32;
  __bf16* Clo = lds + 2 * kStemPT * 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int row0 = blockIdx.x * 4;
  stem_stage(x, H, row0, Chi, Cmi, Clo);
  const int m = lane & 15, q = lane >> 4;
  bf16x8 ab;                                     // A[m = co][k = 8q + j] = W[co][k] (k < 27)
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const int k = 8 * q + j;
    ab[j] = (__bf16)rintf((k < 27 ? w[m * 27 + k] : 0.f) * nlev);
  }
  __syncthreads();
  float bs[4] = {0.f, 0.f, 0.f, 0.f}, bq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int g = wv; g < kStemPT / 16; g += 4) {
    const int p = g * 16 + (lane & 15);
    const int o = p * 32 + 8 * q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, *reinterpret_cast<const bf16x8*>(Clo + o), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, *reinterpret_cast<const bf16x8*>(Cmi + o), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, *reinterpret_cast<const bf16x8*>(Chi + o), acc, 0, 0, 0);
    const int grow = row0 + p / 32;
    if (grow < total_rows) {
      const float4 v = make_float4(acc[0] / nlev, acc[1] / nlev, acc[2] / nlev, acc[3] / nlev);
      *reinterpret_cast<float4*>(y + ((int64_t)grow * 32 + p % 32) * 16 + 4 * q) = v;
      bs[0] += v.x; bs[1] += v.y; bs[2] += v.z; bs[3] += v.w;
      bq[0] += v.x * v.x; bq[1] += v.y * v.y; bq[2] += v.z * v.z; bq[3] += v.w * v.w;
    }
  }
  if (bn_part) {
    float* red = reinterpret_cast<float*>(lds);
#define ROW_SHR_ADD(V, CTRL) V += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, V), CTRL, 0xf, 0xf, false))
#pragma unroll
    for (int e = 0; e < 4; e++) {
      ROW_SHR_ADD(bs[e], 0x111); ROW_SHR_ADD(bs[e], 0x112); ROW_SHR_ADD(bs[e], 0x114); ROW_SHR_ADD(bs[e], 0x118);
      ROW_SHR_ADD(bq[e], 0x111); ROW_SHR_ADD(bq[e], 0x112); ROW_SHR_ADD(bq[e], 0x114); ROW_SHR_ADD(bq[e], 0x118);
    }
#undef ROW_SHR_ADD
    __syncthreads();
    if ((lane & 15) == 15) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        red[(wv * 16 + 4 * q + e) * 2] = bs[e];
        red[(wv * 16 + 4 * q + e) * 2 + 1] = bq[e];
      }
    }
    __syncthreads();
    if (tid < 16) {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int w2 = 0; w2 < 4; w2++) { s0 += red[(w2 * 16 + tid) * 2]; s1 += red[(w2 * 16 + tid) * 2 + 1]; }
      bn_part[((int64_t)tid * gridDim.x + blockIdx.x) * 2] = s0;
      bn_part[((int64_t)tid * gridDim.x + blockIdx.x) * 2 + 1] = s1;
    }
  }
}

// filter gradient of the stem: slab [16][27] per workgroup (pixel range), reduced by wgrad_reduce[_multi]_kernel
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                         float* __restrict__ slabs, int H, int n_tiles, BnLazy lazy) {
  __shared__ __attribute__((aligned(16))) __bf16 lds[3 * kStemPT * 32 + 3 * kStemPT * 16];
  __bf16* Ci[3] = {lds, lds + kStemPT * 32, lds + 2 * kStemPT * 32};
  __bf16* Di[3] = {lds + 3 * kStemPT * 32, lds + 3 * kStemPT * 32 + kStemPT * 16, lds + 3 * kStemPT * 32 + 2 * kStemPT * 16};
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g = lane >> 4, q = (lane & 15) >> 2, pcol = 4 * (lane & 3);
  f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};      // the two 16-column blocks of k
  __shared__ __attribute__((aligned(16))) float kt[32];
  if (lazy.z && lazy.part) bn_totals_lds<16>(lazy, kt, blockIdx.x == 0);      // (workgroup-uniform) totals by this workgroup
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row0 = tile * 4;
    // the dy (and z) quads of this thread are requested BEFORE the im2col gather, whose 16 loads per thread are consumed inside
    // stem_stage: one memory round trip per tile instead of two
    f32x4 dvr[2], zvr[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
      const int i = tid + 256 * it;
      const int64_t off = ((int64_t)row0 * 32 + i / 4) * 16 + 4 * (tid % 4);
      dvr[it] = *reinterpret_cast<const f32x4*>(dy + off);
      zvr[it] = lazy.z ? *reinterpret_cast<const f32x4*>(lazy.z + off) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    stem_stage(x, H, row0, Ci[0], Ci[1], Ci[2]);
    {   // dy tile [128 pixels][16 co] as three bf16 terms (optionally the lazy batch-norm form)
      const int c4 = tid % 4;
      f32x4 la = {0, 0, 0, 0}, lm = la, li = la, lk0 = la, lk1 = la;
      if (lazy.z) {
        la = *reinterpret_cast<const f32x4*>(lazy.ab + 4 * c4);
        lm = *reinterpret_cast<const f32x4*>(lazy.save + 4 * c4);
        li = *reinterpret_cast<const f32x4*>(lazy.save + 16 + 4 * c4);
        const float* kp = lazy.part ? kt : lazy.ktot;
        lk0 = *reinterpret_cast<const f32x4*>(kp + 4 * c4);
        lk1 = *reinterpret_cast<const f32x4*>(kp + 16 + 4 * c4);
      }
#pragma unroll
      for (int it = 0; it < 2; it++) {
        const int i = tid + 256 * it;                  // 512 float4 slots
        f32x4 dv = dvr[it];
        if (lazy.z) dv = la * (dv - lk0 - (zvr[it] - lm) * li * lk1);
        bf16x4 h4, m4, l4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const __bf16 hi = (__bf16)dv[e];
          const float r1 = dv[e] - (float)hi;
          const __bf16 mi = (__bf16)r1;
          h4[e] = hi; m4[e] = mi; l4[e] = (__bf16)(r1 - (float)mi);
        }
        const int o = (i / 4) * 16 + 4 * c4;
        *reinterpret_cast<bf16x4*>(Di[0] + o) = h4;
        *reinterpret_cast<bf16x4*>(Di[1] + o) = m4;
        *reinterpret_cast<bf16x4*>(Di[2] + o) = l4;
      }
    }
    __syncthreads();
    {   // one 32-pixel step per wave: pixels 32 wv + 8 g .. + 7
      const int p0 = 32 * wv + 8 * g;
      bf16x8 a[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const __bf16* pa = Di[t] + (p0 + q) * 16 + pcol;
        a[t] = join8(tr_read(pa), tr_read(pa + 4 * 16));
      }
#pragma unroll
      for (int blk = 0; blk < 2; blk++) {
        bf16x8 b[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
          const __bf16* pb = Ci[t] + (p0 + q) * 32 + 16 * blk + pcol;
          b[t] = join8(tr_read(pb), tr_read(pb + 4 * 32));
        }
        f32x4 v = acc[blk];
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], v, 0, 0, 0);
        acc[blk] = v;
      }
    }
  }
  // four waves -> LDS -> wave 0; C/D: column (k within block) = lane & 15, rows (co) = 4 (lane >> 4) + e
  float* red = reinterpret_cast<float*>(lds);
  __syncthreads();
  if (wv > 0) {
#pragma unroll
    for (int blk = 0; blk < 2; blk++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[(((wv - 1) * 2 + blk) * 4 + e) * 64 + lane] = acc[blk][e];
  }
  __syncthreads();
  if (wv == 0) {
    float* slab = slabs + (int64_t)blockIdx.x * (16 * 27);
#pragma unroll
    for (int blk = 0; blk < 2; blk++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float v = acc[blk][e];
#pragma unroll
        for (int w2 = 0; w2 < 3; w2++) v += red[((w2 * 2 + blk) * 4 + e) * 64 + lane];
        const int k = 16 * blk + (lane & 15), co = 4 * (lane >> 4) + e;
        if (k < 27) slab[co * 27 + k] = v;
      }
    }
  }
}

template <int CIN, int COUT, int WDI, int KS, int S, int PT>
int launch_gen(const float* x, const float* w, float* y, int B, int H, float nlev, float* bn_part, hipStream_t st) {
  constexpr int TR = ConvGen<CIN, COUT, WDI, KS, S, PT>::TR;
  if (H % TR) return ALIGNQ_EUNSUPPORTED;
  const int total_rows = B * H;
  hipLaunchKernelGGL((convgen_fwd_kernel<CIN, COUT, WDI, KS, S, PT>), total_rows / TR, 256, 0, st, x, w, y, H, total_rows,
                     nlev, bn_part);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int C, int WD, int PT>
int launch(const float* x, const float* w, float* y, int B, int H, int dgrad, float nlev, const float* add, float* bn_part,
           hipStream_t st, int xb = 0, float xlev = 1.0f) {
  constexpr int TR = PT / WD;
  const int total_rows = B * H;
  if (H % TR) return ALIGNQ_EUNSUPPORTED;
  const int grid = total_rows / TR;
  if (dgrad) hipLaunchKernelGGL((conv3x3_nhwc_kernel<C, WD, PT, true, 0>), grid, 256, 0, st, x, w, y, H, total_rows, nlev, add, nullptr, 1.0f);
  else if (xb == 2) hipLaunchKernelGGL((conv3x3_nhwc_kernel<C, WD, PT, false, 2>), grid, 256, 0, st, x, w, y, H, total_rows, nlev, add, bn_part, xlev);
  else if (xb == 1) hipLaunchKernelGGL((conv3x3_nhwc_kernel<C, WD, PT, false, 1>), grid, 256, 0, st, x, w, y, H, total_rows, nlev, add, bn_part, xlev);
  else hipLaunchKernelGGL((conv3x3_nhwc_kernel<C, WD, PT, false, 0>), grid, 256, 0, st, x, w, y, H, total_rows, nlev, add, bn_part, xlev);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// =====================================================================================================================
// Filter gradient of the same convolution:  dW[co][tap][ci] = sum_pixels dy[p][co] * x[p + shift(tap)][ci].
// The contraction runs over PIXELS, so MFMA's k index is the pixel while the tiles lie channels-last in LDS.  gfx950's
// ds_read_b64_tr_b16 reads a 4-pixel x 16-channel block and hands every lane ONE channel of the 4 pixels: two such reads
// give a lane its 8 consecutive k of v_mfma_f32_16x16x32_bf16 straight from the [pixel][channel] image - for dy (A operand,
// rows = co) and for x at any tap shift (B operand, columns = ci).  Both operands are arbitrary fp32, so each is split
// exactly into three bf16 terms and a product keeps the six leading term pairs (hh, hm, mh, hl, lh, mm: everything above
// 2^-24 relative), accumulated in fp32: fp32-grade accuracy at 6/16 of the f32-MFMA time (1.4 us instead of 3.8 us per
// layer).  Workgroup = (pixel range, 32x32 block of (co, ci)) [16x16 for C = 16]:
//   C >= 32: wave = one 16x16 (co, ci) tile of the block over ALL pixels of the tile loop - no cross-wave reduction;
//   C == 16: the four waves take alternate 32-pixel steps and are summed through LDS once at the end.
// 9 per-tap accumulators live across a software-pipelined tile loop (the next tile's global loads fly under the MFMAs).  One
// partial-sum slab [C][9][C] per pixel range; wgrad_reduce[_multi]_kernel sums the slabs in fixed order (deterministic).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Geometry of the filter-gradient kernel for a convolution CIN -> COUT, KS x KS, stride S, input width WDI, PT OUTPUT pixels per
// tile (whole output rows).  The 3x3 stride-1 body convolutions are <C, C, WD, 3, 1, PT>.
template <int CIN, int COUT, int WDI, int KS, int S, int PT>
struct WgradGeo {
  static constexpr int WDO = WDI / S, TR = PT / WDO, NT = KS * KS;
  static constexpr int CB = (CIN < COUT ? CIN : COUT) >= 32 ? 32 : 16;           // channel block (co and ci) of a workgroup
  static constexpr int XROWS = KS == 3 ? TR * S + (S == 1 ? 2 : 1) : TR;
  static constexpr int LW = KS == 3 ? WDI + 2 : WDO;
  static constexpr int XA = XROWS * LW * CB;                                     // bf16 elements per x image
  static constexpr int DA = PT * CB;                                             // bf16 elements per dy image
  static constexpr int kRed = 3 * NT * 4 * 64 * 4;                               // bytes: cross-wave reduction (CB == 16)
  static constexpr int kBytes = (3 * (XA + DA) * 2) > kRed ? (3 * (XA + DA) * 2) : kRed;
  static constexpr int kFloats = kBytes / 4;
};
template <int C, int WD, int PT>
struct WgradLds : WgradGeo<C, C, WD, 3, 1, PT> {};

// bx / gx: index and count of the pixel-range workgroups, by: (co block, ci block) index
// H is the OUTPUT height (the input has S*H rows); x [.., S*H, WDI, CIN], dy [.., H, WDI/S, COUT].
template <int CIN, int COUT, int WDI, int KS, int S, int PT, int XB = 0>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ x, const float* __restrict__ dy,
                                           float* __restrict__ slabs, int H, int n_tiles, float* lds_f, int bx, int gx,
                                           int by, BnLazy lazy = BnLazy{nullptr, nullptr, nullptr, nullptr}, float xlev = 1.0f) {
  // XB != 0: x holds int16 / int8 level indices (value idx / xlev): exact in two bf16 terms; the slab is scaled by 1 / xlev
  constexpr int xb = XB;
  using G = WgradGeo<CIN, COUT, WDI, KS, S, PT>;
  constexpr int WDO = G::WDO, TR = G::TR, NT = G::NT, CB = G::CB, LW = G::LW, XA = G::XA, DA = G::DA, XROWS = G::XROWS;
  constexpr int NBLK = CIN / CB;                 // ci blocks (by = co block * NBLK + ci block)
  constexpr int NSTEP = PT / 32;                 // 32-pixel k steps per tile
  static_assert(WDO % 8 == 0 && PT % 32 == 0, "a lane group's 8 pixels lie in one output row");
  __bf16* lds = reinterpret_cast<__bf16*>(lds_f);
  __bf16* Xi[3] = {lds, lds + XA, lds + 2 * XA};
  __bf16* Di[3] = {lds + 3 * XA, lds + 3 * XA + DA, lds + 3 * XA + 2 * DA};
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int bi = by / NBLK, bj = by % NBLK;      // (co block, ci block)
  // this wave's 16x16 tile inside the block, and whether the waves split the k steps (C == 16)
  const int cob = CB == 32 ? (wv >> 1) : 0, cib = CB == 32 ? (wv & 1) : 0;
  const int g = lane >> 4, q = (lane & 15) >> 2, pcol = 4 * (lane & 3);

  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

  constexpr int C4 = CB / 4;
  constexpr int N4 = XROWS * LW * C4;              // float4 slots of the x tile (with halo for KS == 3)
  constexpr int M4 = PT * C4;                      // float4 slots of the dy tile
  constexpr int NIX = (N4 + 255) / 256, NID = (M4 + 255) / 256;
  f32x4 rx[NIX], rd[NID], rz[NID];     // plain vector registers (HIP's float4 struct here ends up in scratch)
  // (fetch / park are spelled out twice below rather than hidden in a lambda or macro: the register arrays must be indexed
  // by unrolled constants or they end up in scratch)
  // pixel range of this workgroup: `per` CONSECUTIVE tiles (the data-gradient workgroups of the same launch that read the same
  // rows of dy are placed on this workgroup's XCD, see conv3x3_bwd_kernel: the second reader finds them in that XCD's L2)
  const int per = (n_tiles + gx - 1) / gx;
  const int t_begin = bx * per, t_end = (t_begin + per < n_tiles) ? t_begin + per : n_tiles;
  int nxt = t_begin;
  if (nxt < t_end) {
    const int row0_ = nxt * TR;
    const int img_lo_ = (row0_ / H) * (H * S), img_hi_ = img_lo_ + H * S;      // INPUT rows of the tile's image
#pragma unroll
    for (int it = 0; it < NIX; it++) {
      const int i = tid + 256 * it;
      const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
      const int grow = KS == 3 ? row0_ * S - 1 + lr : (row0_ + lr) * S;
      const int gcol = KS == 3 ? col - 1 : col * S;
      const bool ok = i < N4 && gcol >= 0 && gcol < WDI && grow >= img_lo_ && grow < img_hi_;
      const int64_t off = ok ? ((int64_t)grow * WDI + gcol) * CIN + bj * CB + 4 * c4 : 0;
      rx[it] = fetch_act4<XB>(x, off);
      if (!ok) rx[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int it = 0; it < NID; it++) {
      const int i = tid + 256 * it;
      const int64_t off = i < M4 ? ((int64_t)row0_ * WDO + i / C4) * COUT + bi * CB + 4 * (i % C4) : 0;
      rd[it] = *reinterpret_cast<const f32x4*>(dy + off);
      if (lazy.z) rz[it] = *reinterpret_cast<const f32x4*>(lazy.z + off);
    }
  }
  // per-thread channel quad of the dy tile (256 % C4 == 0) and its batch-norm constants for the lazy form
  f32x4 la = {0, 0, 0, 0}, lm = la, li = la, lk0 = la, lk1 = la;
  if (lazy.z) {
    const int cq = bi * CB + 4 * (tid % C4);
    la = *reinterpret_cast<const f32x4*>(lazy.ab + cq);
    lm = *reinterpret_cast<const f32x4*>(lazy.save + cq);
    li = *reinterpret_cast<const f32x4*>(lazy.save + COUT + cq);
    if (lazy.part) {             // (workgroup-uniform) see bn_totals_lds; the first tile's loads above are already in flight
      __shared__ __attribute__((aligned(16))) float kt[2 * COUT];
      bn_totals_lds<COUT>(lazy, kt, false);
      lk0 = *reinterpret_cast<const f32x4*>(kt + cq);
      lk1 = *reinterpret_cast<const f32x4*>(kt + COUT + cq);
    } else {
      lk0 = *reinterpret_cast<const f32x4*>(lazy.ktot + cq);
      lk1 = *reinterpret_cast<const f32x4*>(lazy.ktot + COUT + cq);
    }
  }
  for (int tile = t_begin; tile < t_end; tile++) {
    __syncthreads();                               // previous tile's readers are done
#pragma unroll
    for (int it = 0; it < NIX; it++) {             // park the fetched tile in LDS as three bf16 terms
      const int i = tid + 256 * it;
      if (i < N4) {
        const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
        bf16x4 h4, m4, l4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float v = rx[it][e];
          const __bf16 hi = (__bf16)v;
          const float r1 = v - (float)hi;
          const __bf16 mi = (__bf16)r1;
          h4[e] = hi; m4[e] = mi; l4[e] = (__bf16)(r1 - (float)mi);
        }
        const int o = (lr * LW + col) * CB + 4 * c4;
        *reinterpret_cast<bf16x4*>(Xi[0] + o) = h4;
        *reinterpret_cast<bf16x4*>(Xi[1] + o) = m4;
        *reinterpret_cast<bf16x4*>(Xi[2] + o) = l4;
      }
    }
#pragma unroll
    for (int it = 0; it < NID; it++) {
      const int i = tid + 256 * it;
      if (i < M4) {
        bf16x4 h4, m4, l4;
        f32x4 dv = rd[it];
        if (lazy.z) dv = la * (dv - lk0 - (rz[it] - lm) * li * lk1);     // batch-norm input gradient formed on load
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float v = dv[e];
          const __bf16 hi = (__bf16)v;
          const float r1 = v - (float)hi;
          const __bf16 mi = (__bf16)r1;
          h4[e] = hi; m4[e] = mi; l4[e] = (__bf16)(r1 - (float)mi);
        }
        const int o = (i / C4) * CB + 4 * (i % C4);
        *reinterpret_cast<bf16x4*>(Di[0] + o) = h4;
        *reinterpret_cast<bf16x4*>(Di[1] + o) = m4;
        *reinterpret_cast<bf16x4*>(Di[2] + o) = l4;
      }
    }
    __syncthreads();
    nxt = tile + 1;
    if (nxt < t_end) {                           // next tile's loads fly under this tile's MFMA phase
      const int row0_ = nxt * TR;
      const int img_lo_ = (row0_ / H) * (H * S), img_hi_ = img_lo_ + H * S;      // INPUT rows of the tile's image
#pragma unroll
      for (int it = 0; it < NIX; it++) {
        const int i = tid + 256 * it;
        const int c4 = i % C4, col = (i / C4) % LW, lr = i / (C4 * LW);
        const int grow = KS == 3 ? row0_ * S - 1 + lr : (row0_ + lr) * S;
        const int gcol = KS == 3 ? col - 1 : col * S;
        const bool ok = i < N4 && gcol >= 0 && gcol < WDI && grow >= img_lo_ && grow < img_hi_;
        const int64_t off = ok ? ((int64_t)grow * WDI + gcol) * CIN + bj * CB + 4 * c4 : 0;
        rx[it] = fetch_act4<XB>(x, off);
        if (!ok) rx[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int it = 0; it < NID; it++) {
        const int i = tid + 256 * it;
        const int64_t off = i < M4 ? ((int64_t)row0_ * WDO + i / C4) * COUT + bi * CB + 4 * (i % C4) : 0;
        rd[it] = *reinterpret_cast<const f32x4*>(dy + off);
        if (lazy.z) rz[it] = *reinterpret_cast<const f32x4*>(lazy.z + off);
      }
    }
    // ---- MFMA phase: 32 pixels per step; lane group g owns pixels p0 + 8g .. + 7 (8 consecutive columns of one row) ---
    for (int s = (CB == 16 ? wv : 0); s < NSTEP; s += (CB == 16 ? 4 : 1)) {
      const int p0 = 32 * s + 8 * g;
      const int r = p0 / WDO, c0 = p0 % WDO;
      // A = dy: lane address = pixel p0 + 4*half + q, channels cob*16 + pcol
      bf16x8 a[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const __bf16* pa = Di[t] + (p0 + q) * CB + cob * 16 + pcol;
        a[t] = join8(tr_read(pa), tr_read(pa + 4 * CB));
      }
#pragma unroll
      for (int tap = 0; tap < NT; tap++) {
        const int ky = tap / KS, kx = tap % KS;
        bf16x8 b[3];
#pragma unroll
        for (int t = 0; t < 3; t++) {      // the lane's block row q is output pixel c0 + q (+4): input column S * that + kx
          const __bf16* pb = Xi[t] + (KS == 3 ? ((r * S + ky) * LW + ((c0 + q) * S + kx)) : (r * LW + c0 + q)) * CB +
                             cib * 16 + pcol;
          b[t] = join8(tr_read(pb), tr_read(pb + 4 * (KS == 3 ? S : 1) * CB));
        }
        // six leading term pairs, smallest first
        f32x4 v = acc[tap];
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], v, 0, 0, 0);
        if (xb == 0) v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], v, 0, 0, 0);      // (an index has no third term)
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], v, 0, 0, 0);
        v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], v, 0, 0, 0);
        acc[tap] = v;
      }
    }
  }
  // ---- results: C/D layout of the 16x16 MFMA: column (ci) = lane & 15, rows (co) = 4 (lane >> 4) + e ---------------------
  float* slab = slabs + (int64_t)bx * (NT * CIN * COUT);
  if (xb != 0) {
    const float inv = 1.0f / xlev;
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = acc[t] * inv;
  }
  if (CB == 16) {      // the four waves hold partial sums over alternate steps: fixed-order sum through LDS, wave 0 writes
    float* red = lds_f;
    __syncthreads();
    if (wv > 0) {
#pragma unroll
      for (int t = 0; t < NT; t++)
#pragma unroll
        for (int e = 0; e < 4; e++) red[(((wv - 1) * NT + t) * 4 + e) * 64 + lane] = acc[t][e];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
      for (int t = 0; t < NT; t++) {
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float v = acc[t][e];
#pragma unroll
          for (int w2 = 0; w2 < 3; w2++) v += red[((w2 * NT + t) * 4 + e) * 64 + lane];
          // non-temporal: the slab is read once, launches later; it must not displace dx (the next launch's operand) from the caches
          __builtin_nontemporal_store(v, &slab[((int64_t)(bi * CB + 4 * (lane >> 4) + e) * NT + t) * CIN + bj * CB + (lane & 15)]);
        }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
      for (int e = 0; e < 4; e++)
        __builtin_nontemporal_store(acc[t][e], &slab[((int64_t)(bi * CB + cob * 16 + 4 * (lane >> 4) + e) * NT + t) * CIN + bj * CB + cib * 16 + (lane & 15)]);
  }
}

template <int C, int WD, int PT, int XB = 0>
__device__ __forceinline__ void wgrad3x3_body(const float* __restrict__ x, const float* __restrict__ dy,
                                              float* __restrict__ slabs, int H, int n_tiles, float* lds_f, int bx, int gx,
                                              int by, BnLazy lazy = BnLazy{nullptr, nullptr, nullptr, nullptr}, float xlev = 1.0f) {
  wgrad_body<C, C, WD, 3, 1, PT, XB>(x, dy, slabs, H, n_tiles, lds_f, bx, gx, by, lazy, xlev);
}

// stand-alone filter gradient of the transition convolutions (stride 2: 3x3 and 1x1)
template <int CIN, int COUT, int WDI, int KS, int S, int PT>
__global__ __launch_bounds__(256) void wgradgen_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       float* __restrict__ slabs, int H, int n_tiles, BnLazy lazy) {
  __shared__ __attribute__((aligned(16))) float lds[WgradGeo<CIN, COUT, WDI, KS, S, PT>::kFloats];
  wgrad_body<CIN, COUT, WDI, KS, S, PT>(x, dy, slabs, H, n_tiles, lds, blockIdx.x, gridDim.x, blockIdx.y, lazy);
}

template <int C, int WD, int PT, int XB>
__global__ __launch_bounds__(256) void wgrad3x3_nhwc_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ slabs, int H, int n_tiles, float xlev) {
  __shared__ __attribute__((aligned(16))) float lds[WgradLds<C, WD, PT>::kFloats];
  wgrad3x3_body<C, WD, PT, XB>(x, dy, slabs, H, n_tiles, lds, blockIdx.x, gridDim.x, blockIdx.y,
                               BnLazy{nullptr, nullptr, nullptr, nullptr}, xlev);
}

// Backward of one convolution in ONE launch: the first n_wg workgroups take the filter-gradient role (the longer one, so
// it starts first), the rest the data-gradient role; the two are independent and fill the chip together.
template <int C, int WD, int PTD, int PTW, int XB>
__global__ __launch_bounds__(256) void conv3x3_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ w, float* __restrict__ dx,
                                                          float* __restrict__ slabs, int H, int total_rows, float nlev,
                                                          int n_tiles_w, int splits, int nblk2,
                                                          const float* __restrict__ add, BnLazy lazy, float xlev, int dg_R,
                                                          RedFill fill, int n_dg) {
  constexpr int kBytesD = ConvLds<C, WD, PTD>::kBf16 * 2, kBytesW = WgradLds<C, WD, PTW>::kFloats * 4;
  static_assert(kBytesD >= 4096 || kBytesW >= 4096, "the filler role's 4 KB");
  __shared__ __attribute__((aligned(16))) unsigned char lds[kBytesD > kBytesW ? kBytesD : kBytesW];
  const int n_wg = splits * nblk2;
  if ((int)blockIdx.x < n_wg) {
    wgrad3x3_body<C, WD, PTW, XB>(x, dy, slabs, H, n_tiles_w, reinterpret_cast<float*>(lds), blockIdx.x % splits, splits,
                                  blockIdx.x / splits, lazy, xlev);
  } else if ((int)blockIdx.x >= n_wg + n_dg) {
    // dispatched last: the data-gradient tiles finish before the (one per CU, longer) filter-gradient workgroups do
    const int fb = blockIdx.x - n_wg - n_dg;
    int it = 0;
    while (it + 1 < kFill && fb >= fill.blk0[it + 1]) it++;          // block-uniform (scalar) search
    wgrad_reduce_body<256>(fill.slabs[it], fill.n_slabs[it], fill.n_elem[it], fill.dw[it], fb - fill.blk0[it],
                           reinterpret_cast<float*>(lds));
  } else {
    // XCD placement (blocks are dealt round-robin over the 8 XCDs; for speed only): data-gradient tile t reads the dy rows
    // that filter-gradient workgroup t / dg_R reads, so it goes to a block on that workgroup's XCD — the later of the two reads
    // is then an L2 hit instead of a second trip over the fabric
    int m = blockIdx.x - n_wg;
    if (dg_R > 0) {
      const int i = (m / (8 * dg_R)) * 8 + (m & 7), r = (m >> 3) % dg_R;
      m = dg_R * i + r;
    }
    conv3x3_body<C, WD, PTD, true>(dy, w, dx, H, total_rows, nlev, reinterpret_cast<__bf16*>(lds), m, add, nullptr, 0, lazy);
  }
}

// Backward of a transition block's two stride-2 convolutions (3x3 `conv0` and the 1x1 shortcut, same input x) in ONE
// launch instead of four: workgroups [0, n_d) form the data gradient of BOTH convolutions (dgrad_s2_body<.., WITH1>; the longest
// role, so it is dispatched first), the next n_w3 the 3x3 filter gradient's partial slabs, the rest the 1x1 filter gradient's.
// The roles are independent.
template <int CIN, int COUT, int WDI, int PTW3, int PTW1, int PTD>
__global__ __launch_bounds__(256) void transition_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy3,
                                                             const float* __restrict__ dy1, const float* __restrict__ w3,
                                                             const float* __restrict__ w1, float* __restrict__ dx,
                                                             float* __restrict__ slabs3, float* __restrict__ slabs1, int Ho,
                                                             int n_tiles3, int n_tiles1, int splits3, int splits1, int n_d,
                                                             float nlev, const float* __restrict__ add, BnLazy lazy3,
                                                             BnLazy lazy1, int dg_R) {
  using G3 = WgradGeo<CIN, COUT, WDI, 3, 2, PTW3>;
  using G1 = WgradGeo<CIN, COUT, WDI, 1, 2, PTW1>;
  constexpr int NBY = (COUT / G3::CB) * (CIN / G3::CB);
  static_assert(G3::CB == G1::CB, "same channel blocking");
  constexpr int kD = DgradS2Lds<CIN, COUT, WDI, 3, PTD, true>::kBf16 * 2;
  constexpr int kW = G3::kBytes > G1::kBytes ? G3::kBytes : G1::kBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kD > kW ? kD : kW];
  const int n_w3 = splits3 * NBY;
  const int b = blockIdx.x;
  if (b < n_d) {           // (this role publishes the batch-norm parameter gradients of both records)
    // XCD placement as in conv3x3_bwd_kernel: tile t reads the dy rows of filter-gradient pixel range t / dg_R
    int m = b;
    if (dg_R > 0) {
      const int i = (m / (8 * dg_R)) * 8 + (m & 7), r = (m >> 3) % dg_R;
      m = dg_R * i + r;
    }
    dgrad_s2_body<CIN, COUT, WDI, 3, PTD, true>(dy3, w3, dx, 2 * Ho, 0x7fffffff, nlev, add, lazy3,
                                                reinterpret_cast<__bf16*>(lds), m, dy1, w1, lazy1);
  } else if (b < n_d + n_w3) {
    const int c = b - n_d;
    wgrad_body<CIN, COUT, WDI, 3, 2, PTW3>(x, dy3, slabs3, Ho, n_tiles3, reinterpret_cast<float*>(lds), c % splits3, splits3,
                                           c / splits3, lazy3);
  } else {
    const int c = b - n_d - n_w3;
    wgrad_body<CIN, COUT, WDI, 1, 2, PTW1>(x, dy1, slabs1, Ho, n_tiles1, reinterpret_cast<float*>(lds), c % splits1, splits1,
                                           c / splits1, lazy1);
  }
}

__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int n_elem,
                                                            float* __restrict__ dw) {
  __shared__ __attribute__((aligned(16))) float part[4096];
  wgrad_reduce_body(slabs, n_slabs, n_elem, dw, blockIdx.x, part);
}

// the same reduction for up to 32 convolutions in ONE launch: a whole-model step defers all filter-gradient reductions to the
// end of the backward (nothing reads a filter gradient before the optimizer step).  One-dimensional grid over the workgroups
// of all tensors (blk0 = prefix sums: no empty workgroups).
constexpr int kWgMulti = 32;
struct WgChunk {
  const float* slabs[kWgMulti];
  float* dw[kWgMulti];
  int n_slabs[kWgMulti];
  int n_elem[kWgMulti];
  int blk0[kWgMulti + 1];
};
__global__ __launch_bounds__(1024) void wgrad_reduce_multi_kernel(WgChunk c, int cnt) {
  __shared__ __attribute__((aligned(16))) float part[4096];
  int t = 0;
  while (t + 1 < cnt && (int)blockIdx.x >= c.blk0[t + 1]) t++;          // block-uniform (scalar) search
  wgrad_reduce_body(c.slabs[t], c.n_slabs[t], c.n_elem[t], c.dw[t], (int)blockIdx.x - c.blk0[t], part);
}

template <int C, int WD, int PT>
int launch_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int H, int* n_slabs_out, hipStream_t st,
                 int xb = 0, float xlev = 1.0f) {
  constexpr int TR = PT / WD;
  if (H % TR) return ALIGNQ_EUNSUPPORTED;
  const int n_tiles = B * H / TR;
  constexpr int NB = (C >= 32 ? C / 32 : 1);
  int splits = 256 / (NB * NB);                    // pixel ranges (= slabs): ~256 workgroups in total
  if (splits > n_tiles) splits = n_tiles;
  if (xb == 2) hipLaunchKernelGGL((wgrad3x3_nhwc_kernel<C, WD, PT, 2>), dim3(splits, NB * NB), 256, 0, st, x, dy, ws, H, n_tiles, xlev);
  else if (xb == 1) hipLaunchKernelGGL((wgrad3x3_nhwc_kernel<C, WD, PT, 1>), dim3(splits, NB * NB), 256, 0, st, x, dy, ws, H, n_tiles, xlev);
  else hipLaunchKernelGGL((wgrad3x3_nhwc_kernel<C, WD, PT, 0>), dim3(splits, NB * NB), 256, 0, st, x, dy, ws, H, n_tiles, xlev);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (n_slabs_out) { *n_slabs_out = splits; return 0; }      // deferred: the caller reduces (alignq_conv3x3_wgrad_reduce_multi)
  const int n_elem = 9 * C * C;
  hipLaunchKernelGGL(wgrad_reduce_kernel, wgrad_reduce_blocks(splits, n_elem), 1024, 0, st, ws, splits, n_elem, dw);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int C, int WD, int PTD, int PTW>
int launch_bwd(const float* x, const float* dy, const float* w, float* dx, float* ws, int B, int H, float nlev,
               int* n_slabs_out, const float* add, BnLazy lazy, hipStream_t st, int xb, float xlev, const RedFill& fill) {
  constexpr int TRD = PTD / WD, TRW = PTW / WD;
  if (H % TRD || H % TRW) return ALIGNQ_EUNSUPPORTED;
  const int total_rows = B * H;
  const int n_tiles_w = total_rows / TRW;
  constexpr int NB = (C >= 32 ? C / 32 : 1);
  int splits = 256 / (NB * NB);
  if (splits > n_tiles_w) splits = n_tiles_w;
  const int n_d = total_rows / TRD;
  const int grid = splits * NB * NB + n_d + fill.blk0[kFill];
  // data-gradient tiles per filter-gradient pixel range (0: no XCD placement: the ranges do not tile the batch evenly)
  const int per = (n_tiles_w + splits - 1) / splits;
  int dg_R = 0;
  if ((per * TRW) % TRD == 0 && (splits * NB * NB) % 8 == 0 && splits % 8 == 0) {
    const int R = per * TRW / TRD;
    if (R >= 1 && R * splits == n_d) dg_R = R;
  }
#define LBW(XBV) hipLaunchKernelGGL((conv3x3_bwd_kernel<C, WD, PTD, PTW, XBV>), grid, 256, 0, st, x, dy, w, dx, ws, H, total_rows, nlev, n_tiles_w, splits, NB * NB, add, lazy, xlev, dg_R, fill, n_d)
  if (xb == 2) LBW(2); else if (xb == 1) LBW(1); else LBW(0);
#undef LBW
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  *n_slabs_out = splits;
  return 0;
}

template <int CIN, int COUT, int WDI, int KS, int S, int PT>
int launch_wgradgen(const float* x, const float* dy, float* dw, float* ws, int B, int H, int* n_slabs_out, BnLazy lazy,
                    hipStream_t st) {
  using G = WgradGeo<CIN, COUT, WDI, KS, S, PT>;
  if (H % G::TR) return ALIGNQ_EUNSUPPORTED;
  const int n_tiles = B * H / G::TR;
  constexpr int NBY = (COUT / G::CB) * (CIN / G::CB);
  int splits = 256 / NBY;
  if (splits > n_tiles) splits = n_tiles;
  hipLaunchKernelGGL((wgradgen_kernel<CIN, COUT, WDI, KS, S, PT>), dim3(splits, NBY), 256, 0, st, x, dy, ws, H, n_tiles, lazy);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (n_slabs_out) { *n_slabs_out = splits; return 0; }
  const int n_elem = G::NT * CIN * COUT;
  hipLaunchKernelGGL(wgrad_reduce_kernel, wgrad_reduce_blocks(splits, n_elem), 1024, 0, st, ws, splits, n_elem, dw);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int CIN, int COUT, int WDI, int PT3, int PT1>
int launch_transition_fwd(const float* x, const float* w3, const float* w1, float* y3, float* y1, int B, int H, float nlev,
                          float* part3, float* part1, hipStream_t st) {
  constexpr int TR3 = ConvGen<CIN, COUT, WDI, 3, 2, PT3>::TR, TR1 = ConvGen<CIN, COUT, WDI, 1, 2, PT1>::TR;
  if (H % TR3 || H % TR1) return ALIGNQ_EUNSUPPORTED;
  const int total_rows = B * H, n3 = total_rows / TR3, n1 = total_rows / TR1;
  hipLaunchKernelGGL((transition_fwd_kernel<CIN, COUT, WDI, PT3, PT1>), n3 + n1, 256, 0, st, x, w3, w1, y3, y1, H, total_rows,
                     nlev, part3, part1, n3);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <int CIN, int COUT, int WDI, int PTW3, int PTW1, int PTD>
int launch_transition_bwd(const float* x, const float* dy3, const float* dy1, const float* w3, const float* w1, float* dx,
                          float* ws3, float* ws1, int B, int Ho, float nlev, int* ns3, int* ns1, const float* add,
                          BnLazy lazy3, BnLazy lazy1, hipStream_t st) {
  using G3 = WgradGeo<CIN, COUT, WDI, 3, 2, PTW3>;
  using G1 = WgradGeo<CIN, COUT, WDI, 1, 2, PTW1>;
  constexpr int TRD = PTD / WDI;
  if (Ho % G3::TR || Ho % G1::TR || (2 * Ho) % TRD) return ALIGNQ_EUNSUPPORTED;
  const int n_tiles3 = B * Ho / G3::TR, n_tiles1 = B * Ho / G1::TR;
  constexpr int NBY = (COUT / G3::CB) * (CIN / G3::CB);
  // the same pixel ranges as the stand-alone filter-gradient launches (so the slabs, hence dW, are bit-identical to theirs)
  int splits3 = 256 / NBY, splits1 = 256 / NBY;      // (measured: 128 -> 42 us, 512 -> 36 us against 33 us per launch, 16 -> 32)
  if (splits3 > n_tiles3) splits3 = n_tiles3;
  if (splits1 > n_tiles1) splits1 = n_tiles1;
  const int n_d = B * 2 * Ho / TRD;
  // data-gradient tiles per filter-gradient pixel range (both filter roles cut the output rows the same way), 0: no placement
  int dg_R = 0;
  {
    const int per3 = (n_tiles3 + splits3 - 1) / splits3, per1 = (n_tiles1 + splits1 - 1) / splits1;
    const int rows3 = per3 * G3::TR, rows1 = per1 * G1::TR;            // output rows per pixel range
    if (rows3 == rows1 && splits3 == splits1 && splits3 % 8 == 0 && n_d % 8 == 0 && (2 * rows3) % TRD == 0) {
      const int R = 2 * rows3 / TRD;
      if (R >= 1 && R * splits3 == n_d) dg_R = R;
    }
  }
  hipLaunchKernelGGL((transition_bwd_kernel<CIN, COUT, WDI, PTW3, PTW1, PTD>), (splits3 + splits1) * NBY + n_d, 256, 0, st, x,
                     dy3, dy1, w3, w1, dx, ws3, ws1, Ho, n_tiles3, n_tiles1, splits3, splits1, n_d, nlev, add, lazy3, lazy1,
                     dg_R);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  *ns3 = splits3;
  *ns1 = splits1;
  return 0;
}

}  // namespace

// lazy.ktot == nullptr: fill the in-kernel totals form from the site backward's per-tile sums of a [B, HW, C] tensor
static int lazy_parts(BnLazy& lazy, const float* dx_part, float* dgamma, float* dbeta, int B, int C, int HW) {
  if (!lazy.z || lazy.ktot || !dx_part) return 0;
  const int64_t F = (int64_t)C * HW;
  const int tf = alignq_site::bwd_tile_features(B, F);
  if (F % tf || (C > tf && C % tf) || (C < tf && tf % C)) return ALIGNQ_EUNSUPPORTED;
  lazy.part = dx_part; lazy.n_tiles = (int)(F / tf); lazy.tile_f = tf; lazy.inv_n = 1.0 / ((double)B * HW);
  lazy.dgamma = dgamma; lazy.dbeta = dbeta;
  return 0;
}

extern "C" {

// y[b,h,w,co] = sum x[b,h+ky-1,w+kx-1,ci] * wt[co,ky,kx,ci]   (dgrad = 0)
// dx[b,h,w,ci] = sum dy[b,h-ky+1,w-kx+1,co] * wt[co,ky,kx,ci] (dgrad = 1: x := dy)
// wt must hold k-bit quantised values b / (2^k - 1), 1 <= k <= 8 (weight_quantize_fn's output).
// workgroups (= batch-norm partials per channel) of the forward launch for this shape; 0 if unsupported
int alignq_conv3x3_bn_parts(int B, int H, int W, int C) {
  const int pt = (C == 16 && W == 32) ? 256 : (C == 32 && W == 16) ? 128 : (C == 64 && W == 8) ? 32 : 0;
  if (!pt || B < 1 || H < 1 || H % (pt / W)) return 0;
  return B * H / (pt / W);
}

static int act_bins_ok(const void* x_bins, int x_bin_bytes, int a_bit) {
  if (!x_bins) return 1;
  return (x_bin_bytes == 1 || x_bin_bytes == 2) && a_bit >= 1 && a_bit <= 16 && !(reinterpret_cast<uintptr_t>(x_bins) & 15);
}

int alignq_conv3x3_nhwc(const float* x, const float* wt, float* y, int B, int H, int W, int C, int w_bit, int dgrad,
                        const float* add, float* bn_part, const void* x_bins, int x_bin_bytes, int a_bit, void* stream) {
  if ((!x && !x_bins) || !wt || !y || B < 1 || H < 1) return ALIGNQ_EINVAL;
  if (x_bins && (dgrad || !act_bins_ok(x_bins, x_bin_bytes, a_bit))) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(y)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  if (x_bins) {        // N2: the activation operand as int8 / int16 level indices of an a_bit-bit ADMM-formula quantiser
    const float xlev = (float)((1 << a_bit) - 1);
    const float* xb_ = reinterpret_cast<const float*>(x_bins);
    if (C == 16 && W == 32) return launch<16, 32, 256>(xb_, wt, y, B, H, 0, nlev, add, bn_part, st, x_bin_bytes, xlev);
    if (C == 32 && W == 16) return launch<32, 16, 128>(xb_, wt, y, B, H, 0, nlev, add, bn_part, st, x_bin_bytes, xlev);
    if (C == 64 && W == 8) return launch<64, 8, 32>(xb_, wt, y, B, H, 0, nlev, add, bn_part, st, x_bin_bytes, xlev);
    return ALIGNQ_EUNSUPPORTED;
  }
  // tile sizes measured on MI355X (forward us per layer at batch 128): C=16: 256 pixels 7.8 (128: 8.7); C=32: 128 pixels 8.0
  // (256: 11.0); C=64: 32 pixels 9.3 (64: 12.3)
  if (C == 16 && W == 32) return launch<16, 32, 256>(x, wt, y, B, H, dgrad, nlev, add, bn_part, st);
  if (C == 32 && W == 16) return launch<32, 16, 128>(x, wt, y, B, H, dgrad, nlev, add, bn_part, st);
  if (C == 64 && W == 8) return launch<64, 8, 32>(x, wt, y, B, H, dgrad, nlev, add, bn_part, st);
  return ALIGNQ_EUNSUPPORTED;
}

size_t alignq_conv3x3_wgrad_ws_bytes(int C) { return (size_t)256 * 9 * (size_t)C * C * sizeof(float); }

// dW[co,ky,kx,ci] = sum_{b,h,w} dy[b,h,w,co] * x[b,h+ky-1,w+kx-1,ci]; plain fp32 (v_mfma_f32_*_f32), deterministic.
// n_slabs_out == NULL: partial sums + reduction (two launches), dw is complete on return.
// n_slabs_out != NULL: partial sums only; *n_slabs_out receives the number of slabs left in ws for
// alignq_conv3x3_wgrad_reduce_multi.
int alignq_conv3x3_nhwc_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H, int W, int C,
                              int* n_slabs_out, const void* x_bins, int x_bin_bytes, int a_bit, void* stream) {
  if ((!x && !x_bins) || !dy || !ws || B < 1 || H < 1 || (!dw && !n_slabs_out)) return ALIGNQ_EINVAL;
  if (!act_bins_ok(x_bins, x_bin_bytes, a_bit)) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dw)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int xb = x_bins ? x_bin_bytes : 0;
  const float xlev = x_bins ? (float)((1 << a_bit) - 1) : 1.0f;
  const float* xp = x_bins ? reinterpret_cast<const float*>(x_bins) : x;
  if (C == 16 && W == 32) return launch_wgrad<16, 32, 128>(xp, dy, dw, (float*)ws, B, H, n_slabs_out, st, xb, xlev);
  if (C == 32 && W == 16) return launch_wgrad<32, 16, 128>(xp, dy, dw, (float*)ws, B, H, n_slabs_out, st, xb, xlev);
  if (C == 64 && W == 8) return launch_wgrad<64, 8, 64>(xp, dy, dw, (float*)ws, B, H, n_slabs_out, st, xb, xlev);
  return ALIGNQ_EUNSUPPORTED;
}

int alignq_conv3x3_wgrad_reduce_multi(int T, const void* const* ws, float* const* dw, const int* n_slabs, const int* n_elem,
                                      void* stream) {
  if (T <= 0 || !ws || !dw || !n_slabs || !n_elem) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < T; t0 += kWgMulti) {
    const int cnt = (T - t0 < kWgMulti) ? T - t0 : kWgMulti;
    WgChunk c;
    int blocks = 0;
    for (int i = 0; i < cnt; i++) {
      if (!ws[t0 + i] || !dw[t0 + i] || n_slabs[t0 + i] < 1 || n_elem[t0 + i] < 1) return ALIGNQ_EINVAL;
      c.slabs[i] = (const float*)ws[t0 + i]; c.dw[i] = dw[t0 + i]; c.n_slabs[i] = n_slabs[t0 + i];
      c.n_elem[i] = n_elem[t0 + i];
      c.blk0[i] = blocks;
      blocks += wgrad_reduce_blocks(c.n_slabs[i], c.n_elem[i]);
    }
    for (int i = cnt; i <= kWgMulti; i++) c.blk0[i] = blocks;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, blocks, 1024, 0, st, c, cnt);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

// Both gradients of one convolution in a single launch (data gradient as alignq_conv3x3_nhwc(dgrad = 1), filter-gradient
// partial sums as alignq_conv3x3_nhwc_wgrad with a deferred reduction: *n_slabs_out slabs are left in ws).
int alignq_conv3x3_nhwc_bwd_fill(const float* x, const float* dy, const float* wt, float* dx, void* ws, int B, int H, int W,
                                 int C, int w_bit, int* n_slabs_out, const float* add, const float* bn_z, const float* bn_ab,
                            const float* bn_save, const float* bn_ktot, const float* bn_dx_part, float* bn_dgamma,
                            float* bn_dbeta, const void* x_bins, int x_bin_bytes, int a_bit, int n_fill,
                                 const void* const* fill_ws, float* const* fill_dw, const int* fill_n_slabs,
                                 const int* fill_n_elem, void* stream) {
  if (bn_z && (!bn_ab || !bn_save || (!bn_ktot && !bn_dx_part))) return ALIGNQ_EINVAL;
  BnLazy lazy{bn_z, bn_ab, bn_save, bn_ktot};
  if (int rc = lazy_parts(lazy, bn_dx_part, bn_dgamma, bn_dbeta, B, C, H * W)) return rc;   // totals formed inside the kernel
  if ((!x && !x_bins) || !dy || !wt || !dx || !ws || !n_slabs_out || B < 1 || H < 1) return ALIGNQ_EINVAL;
  if (!act_bins_ok(x_bins, x_bin_bytes, a_bit)) return ALIGNQ_EINVAL;
  const int xb = x_bins ? x_bin_bytes : 0;                 // N2: the filter-gradient role reads the level indices of x
  const float xlev = x_bins ? (float)((1 << a_bit) - 1) : 1.0f;
  if (x_bins) x = reinterpret_cast<const float*>(x_bins);
  if (w_bit < 1 || w_bit > 8) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(wt) |
       reinterpret_cast<uintptr_t>(dx)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  RedFill fill{};
  if (n_fill < 0 || n_fill > kFill || (n_fill && (!fill_ws || !fill_dw || !fill_n_slabs || !fill_n_elem))) return ALIGNQ_EINVAL;
  for (int i = 0; i < n_fill; i++) {
    if (!fill_ws[i] || !fill_dw[i] || fill_n_slabs[i] < 1 || fill_n_elem[i] < 1) return ALIGNQ_EINVAL;
    fill.slabs[i] = (const float*)fill_ws[i]; fill.dw[i] = fill_dw[i]; fill.n_slabs[i] = fill_n_slabs[i];
    fill.n_elem[i] = fill_n_elem[i];
    fill.blk0[i + 1] = fill.blk0[i] + wgrad_reduce_blocks(fill_n_slabs[i], fill_n_elem[i], 256);
  }
  for (int i = n_fill; i < kFill; i++) fill.blk0[i + 1] = fill.blk0[i];
  if (C == 16 && W == 32) return launch_bwd<16, 32, 256, 128>(x, dy, wt, dx, (float*)ws, B, H, nlev, n_slabs_out, add, lazy, st, xb, xlev, fill);
  if (C == 32 && W == 16) return launch_bwd<32, 16, 128, 128>(x, dy, wt, dx, (float*)ws, B, H, nlev, n_slabs_out, add, lazy, st, xb, xlev, fill);
  if (C == 64 && W == 8) return launch_bwd<64, 8, 32, 64>(x, dy, wt, dx, (float*)ws, B, H, nlev, n_slabs_out, add, lazy, st, xb, xlev, fill);
  return ALIGNQ_EUNSUPPORTED;
}

int alignq_conv3x3_nhwc_bwd(const float* x, const float* dy, const float* wt, float* dx, void* ws, int B, int H, int W,
                            int C, int w_bit, int* n_slabs_out, const float* add, const float* bn_z, const float* bn_ab,
                            const float* bn_save, const float* bn_ktot, const float* bn_dx_part, float* bn_dgamma,
                            float* bn_dbeta, const void* x_bins, int x_bin_bytes, int a_bit, void* stream) {
  return alignq_conv3x3_nhwc_bwd_fill(x, dy, wt, dx, ws, B, H, W, C, w_bit, n_slabs_out, add, bn_z, bn_ab, bn_save, bn_ktot,
                                      bn_dx_part, bn_dgamma, bn_dbeta, x_bins, x_bin_bytes, a_bit, 0, nullptr, nullptr, nullptr,
                                      nullptr, stream);
}

// Forward of the ResNet body's transition convolutions (see convgen_fwd_kernel): (KS, stride) = (3, 2) padding 1 or (1, 2)
// padding 0, (CIN, COUT, W_in) in {(16, 32, 32), (32, 64, 16)}.  x [B, H_in, W_in, CIN], wt [COUT, KS, KS, CIN] (channels-last
// storage), y [B, H_in/2, W_in/2, COUT].  bn_part as in alignq_conv3x3_nhwc ([COUT][alignq_conv_gen_bn_parts][2] floats).
static int gen_tile_rows(int H_in, int W_in, int CIN, int COUT, int KS, int stride) {
  if (stride != 2 || (KS != 1 && KS != 3)) return 0;
  if (CIN == 16 && COUT == 32 && W_in == 32) return KS == 3 ? 4 : 8;       // output rows per workgroup
  if (CIN == 32 && COUT == 64 && W_in == 16) return KS == 3 ? 4 : 8;
  return 0;
}
int alignq_conv_gen_bn_parts(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride) {
  const int tr = gen_tile_rows(H_in, W_in, CIN, COUT, KS, stride);
  if (!tr || B < 1 || H_in < 2 || (H_in % 2) || ((H_in / 2) % tr)) return 0;
  return B * (H_in / 2) / tr;
}
int alignq_conv_gen_nhwc_fwd(const float* x, const float* wt, float* y, int B, int H_in, int W_in, int CIN, int COUT, int KS,
                             int stride, int w_bit, float* bn_part, void* stream) {
  if (!x || !wt || !y || B < 1) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8 || !alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, KS, stride)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(y)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  const int H = H_in / 2;
  if (CIN == 16 && KS == 3) return launch_gen<16, 32, 32, 3, 2, 64>(x, wt, y, B, H, nlev, bn_part, st);
  if (CIN == 16 && KS == 1) return launch_gen<16, 32, 32, 1, 2, 128>(x, wt, y, B, H, nlev, bn_part, st);
  if (CIN == 32 && KS == 3) return launch_gen<32, 64, 16, 3, 2, 32>(x, wt, y, B, H, nlev, bn_part, st);
  if (CIN == 32 && KS == 1) return launch_gen<32, 64, 16, 1, 2, 64>(x, wt, y, B, H, nlev, bn_part, st);
  return ALIGNQ_EUNSUPPORTED;
}

// Filter gradient of the transition convolutions (shapes of alignq_conv_gen_nhwc_fwd): dW [COUT,KS,KS,CIN] from x [B,H_in,W_in,CIN]
// and dy [B,H_in/2,W_in/2,COUT]; ws = alignq_conv_gen_wgrad_ws_bytes; n_slabs_out as in alignq_conv3x3_nhwc_wgrad.
size_t alignq_conv_gen_wgrad_ws_bytes(int CIN, int COUT, int KS) { return (size_t)256 * KS * KS * (size_t)CIN * COUT * sizeof(float); }
int alignq_conv_gen_nhwc_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H_in, int W_in, int CIN,
                               int COUT, int KS, int stride, int* n_slabs_out, const float* bn_z, const float* bn_ab,
                               const float* bn_save, const float* bn_ktot, const float* bn_dx_part, void* stream) {
  if (!x || !dy || !ws || B < 1 || (!dw && !n_slabs_out)) return ALIGNQ_EINVAL;
  if (bn_z && (!bn_ab || !bn_save || (!bn_ktot && !bn_dx_part))) return ALIGNQ_EINVAL;
  BnLazy lazy{bn_z, bn_ab, bn_save, bn_ktot};
  if (!alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, KS, stride)) return ALIGNQ_EUNSUPPORTED;
  if (int rc = lazy_parts(lazy, bn_dx_part, nullptr, nullptr, B, COUT, (H_in / 2) * (W_in / 2))) return rc;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(dw)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int H = H_in / 2;
  if (CIN == 16 && KS == 3) return launch_wgradgen<16, 32, 32, 3, 2, 64>(x, dy, dw, (float*)ws, B, H, n_slabs_out, lazy, st);
  if (CIN == 16 && KS == 1) return launch_wgradgen<16, 32, 32, 1, 2, 128>(x, dy, dw, (float*)ws, B, H, n_slabs_out, lazy, st);
  if (CIN == 32 && KS == 3) return launch_wgradgen<32, 64, 16, 3, 2, 64>(x, dy, dw, (float*)ws, B, H, n_slabs_out, lazy, st);
  if (CIN == 32 && KS == 1) return launch_wgradgen<32, 64, 16, 1, 2, 64>(x, dy, dw, (float*)ws, B, H, n_slabs_out, lazy, st);
  return ALIGNQ_EUNSUPPORTED;
}

// Data gradient of the transition convolutions: dx [B,H_in,W_in,CIN] from dy [B,H_in/2,W_in/2,COUT]; bn_* as in
// alignq_conv3x3_nhwc_bwd (dy given in the lazy batch-norm form when bn_z != NULL).
int alignq_conv_gen_nhwc_dgrad(const float* dy, const float* wt, float* dx, int B, int H_in, int W_in, int CIN, int COUT, int KS,
                               int stride, int w_bit, const float* add, const float* bn_z, const float* bn_ab,
                               const float* bn_save, const float* bn_ktot, const float* bn_dx_part, float* bn_dgamma,
                               float* bn_dbeta, void* stream) {
  if (!dy || !wt || !dx || B < 1) return ALIGNQ_EINVAL;
  if (add && (reinterpret_cast<uintptr_t>(add) & 15)) return ALIGNQ_EUNSUPPORTED;
  if (bn_z && (!bn_ab || !bn_save || (!bn_ktot && !bn_dx_part))) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8 || !alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, KS, stride)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(dx)) & 15) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  BnLazy lazy{bn_z, bn_ab, bn_save, bn_ktot};
  if (int rc = lazy_parts(lazy, bn_dx_part, bn_dgamma, bn_dbeta, B, COUT, (H_in / 2) * (W_in / 2))) return rc;
  if (CIN == 16 && KS == 3) return launch_dgrad_s2<16, 32, 32, 3, 128>(dy, wt, dx, B, H_in, nlev, add, lazy, st);
  if (CIN == 16 && KS == 1) return launch_dgrad_s2<16, 32, 32, 1, 128>(dy, wt, dx, B, H_in, nlev, add, lazy, st);
  if (CIN == 32 && KS == 3) return launch_dgrad_s2<32, 64, 16, 3, 128>(dy, wt, dx, B, H_in, nlev, add, lazy, st);
  if (CIN == 32 && KS == 1) return launch_dgrad_s2<32, 64, 16, 1, 128>(dy, wt, dx, B, H_in, nlev, add, lazy, st);
  return ALIGNQ_EUNSUPPORTED;
}

// Both convolutions of a transition block in one launch each way (shapes of alignq_conv_gen_nhwc_fwd; wt3 [COUT,3,3,CIN], wt1
// [COUT,1,1,CIN], the same w_bit): y3 / y1 [B,H_in/2,W_in/2,COUT]; bn_part3 / bn_part1 as alignq_conv_gen_nhwc_fwd's for KS = 3 /
// KS = 1.  Results are bit-identical to the two (forward) / four (backward) separate launches.
int alignq_transition_nhwc_fwd(const float* x, const float* wt3, const float* wt1, float* y3, float* y1, int B, int H_in,
                               int W_in, int CIN, int COUT, int w_bit, float* bn_part3, float* bn_part1, void* stream) {
  if (!x || !wt3 || !wt1 || !y3 || !y1 || B < 1) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8 || !alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, 3, 2) ||
      !alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, 1, 2))
    return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(wt3) | reinterpret_cast<uintptr_t>(wt1) |
       reinterpret_cast<uintptr_t>(y3) | reinterpret_cast<uintptr_t>(y1)) & 15)
    return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  const int H = H_in / 2;
  if (CIN == 16) return launch_transition_fwd<16, 32, 32, 64, 128>(x, wt3, wt1, y3, y1, B, H, nlev, bn_part3, bn_part1, st);
  if (CIN == 32) return launch_transition_fwd<32, 64, 16, 32, 64>(x, wt3, wt1, y3, y1, B, H, nlev, bn_part3, bn_part1, st);
  return ALIGNQ_EUNSUPPORTED;
}

// dx [B,H_in,W_in,CIN] = data gradient of BOTH convolutions (+ add), and the partial-sum slabs of both filter gradients
// (ws3 / ws1 = alignq_conv_gen_wgrad_ws_bytes for KS = 3 / 1; *n_slabs3 / *n_slabs1 slabs are left for
// alignq_conv3x3_wgrad_reduce_multi).  bn3_* / bn1_*: the lazy batch-norm form of dy3 / dy1 as in alignq_conv_gen_nhwc_dgrad.
int alignq_transition_nhwc_bwd(const float* x, const float* dy3, const float* dy1, const float* wt3, const float* wt1,
                               float* dx, void* ws3, void* ws1, int B, int H_in, int W_in, int CIN, int COUT, int w_bit,
                               int* n_slabs3, int* n_slabs1, const float* add,
                               const float* bn3_z, const float* bn3_ab, const float* bn3_save, const float* bn3_ktot,
                               const float* bn3_dx_part, float* bn3_dgamma, float* bn3_dbeta,
                               const float* bn1_z, const float* bn1_ab, const float* bn1_save, const float* bn1_ktot,
                               const float* bn1_dx_part, float* bn1_dgamma, float* bn1_dbeta, void* stream) {
  if (!x || !dy3 || !dy1 || !wt3 || !wt1 || !dx || !ws3 || !ws1 || !n_slabs3 || !n_slabs1 || B < 1) return ALIGNQ_EINVAL;
  if (bn3_z && (!bn3_ab || !bn3_save || (!bn3_ktot && !bn3_dx_part))) return ALIGNQ_EINVAL;
  if (bn1_z && (!bn1_ab || !bn1_save || (!bn1_ktot && !bn1_dx_part))) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8 || !alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, 3, 2) ||
      !alignq_conv_gen_bn_parts(B, H_in, W_in, CIN, COUT, 1, 2))
    return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy3) | reinterpret_cast<uintptr_t>(dy1) |
       reinterpret_cast<uintptr_t>(wt3) | reinterpret_cast<uintptr_t>(wt1) | reinterpret_cast<uintptr_t>(dx) |
       reinterpret_cast<uintptr_t>(add)) & 15)
    return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const float nlev = (float)((1 << w_bit) - 1);
  BnLazy lazy3{bn3_z, bn3_ab, bn3_save, bn3_ktot}, lazy1{bn1_z, bn1_ab, bn1_save, bn1_ktot};
  const int HWo = (H_in / 2) * (W_in / 2);
  if (int rc = lazy_parts(lazy3, bn3_dx_part, bn3_dgamma, bn3_dbeta, B, COUT, HWo)) return rc;
  if (int rc = lazy_parts(lazy1, bn1_dx_part, bn1_dgamma, bn1_dbeta, B, COUT, HWo)) return rc;
  const int Ho = H_in / 2;
  if (CIN == 16)
    return launch_transition_bwd<16, 32, 32, 64, 128, 256>(x, dy3, dy1, wt3, wt1, dx, (float*)ws3, (float*)ws1, B, Ho, nlev,
                                                          n_slabs3, n_slabs1, add, lazy3, lazy1, st);
  if (CIN == 32)
    return launch_transition_bwd<32, 64, 16, 64, 64, 128>(x, dy3, dy1, wt3, wt1, dx, (float*)ws3, (float*)ws1, B, Ho, nlev,
                                                         n_slabs3, n_slabs1, add, lazy3, lazy1, st);
  return ALIGNQ_EUNSUPPORTED;
}

// The stem convolution (3 -> 16 channels, 3x3, stride 1, padding 1, width 32): x [B,H,32,3], wt [16,3,3,3] (channels-last
// storage), y [B,H,32,16]; H % 4 == 0.  bn_part: [16][alignq_conv_stem_bn_parts][2] floats or NULL.
int alignq_conv_stem_bn_parts(int B, int H, int W) { return (W == 32 && B >= 1 && H >= 4 && H % 4 == 0) ? B * H / 4 : 0; }
int alignq_conv_stem_nhwc_fwd(const float* x, const float* wt, float* y, int B, int H, int W, int w_bit, float* bn_part,
                              void* stream) {
  if (!x || !wt || !y) return ALIGNQ_EINVAL;
  if (w_bit < 1 || w_bit > 8 || !alignq_conv_stem_bn_parts(B, H, W)) return ALIGNQ_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(y) & 15) return ALIGNQ_EUNSUPPORTED;
  hipLaunchKernelGGL(stem_fwd_kernel, B * H / 4, 256, 0, (hipStream_t)stream, x, wt, y, H, B * H, (float)((1 << w_bit) - 1),
                     bn_part);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
// filter gradient of the stem: dW [16,3,3,3]; ws = 256 * 16 * 27 floats; n_slabs_out / bn_* as in alignq_conv_gen_nhwc_wgrad
int alignq_conv_stem_nhwc_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H, int W, int* n_slabs_out,
                                const float* bn_z, const float* bn_ab, const float* bn_save, const float* bn_ktot,
                                const float* bn_dx_part, float* bn_dgamma, float* bn_dbeta, void* stream) {
  if (!x || !dy || !ws || (!dw && !n_slabs_out)) return ALIGNQ_EINVAL;
  if (bn_z && (!bn_ab || !bn_save || (!bn_ktot && !bn_dx_part))) return ALIGNQ_EINVAL;
  if (!alignq_conv_stem_bn_parts(B, H, W) || (reinterpret_cast<uintptr_t>(dy) & 15)) return ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int n_tiles = B * H / 4;
  const int splits = n_tiles < 256 ? n_tiles : 256;
  BnLazy lazy{bn_z, bn_ab, bn_save, bn_ktot};
  if (int rc = lazy_parts(lazy, bn_dx_part, bn_dgamma, bn_dbeta, B, 16, H * W)) return rc;
  hipLaunchKernelGGL(stem_wgrad_kernel, splits, 256, 0, st, x, dy, (float*)ws, H, n_tiles, lazy);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (n_slabs_out) { *n_slabs_out = splits; return 0; }
  hipLaunchKernelGGL(wgrad_reduce_kernel, wgrad_reduce_blocks(splits, 16 * 27), 1024, 0, st, (const float*)ws, splits, 16 * 27, dw);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
