// qgemm_kernels.hip — Conv2d_Q's convolution (reference: cdf_alignment_admm/dann_office/model/quantization.py:164-181,
// F.conv2d(input, weight_q, ...); callers model/resnet.py:31-41 conv3x3 / conv1x1, Bottleneck :104-110, :131-156) at the
// ResNet-50 shapes of BASELINE config 5: channels-last fp32 tensors, C_in / C_out multiples of 64, 1x1 (stride 1 or 2) and 3x3
// (padding 1).  Forward, data gradient and filter gradient as GEMMs on the bf16 / f16 matrix cores with EXACT products:
//   * the filter is weight_quantize_fn's output W_q = b / n, integer bins |b| <= n = 2^k - 1 <= 255: the operand is b itself
//     (exact in bf16's 8 and f16's 11 significant bits), recovered as rint(W_q * n) while the tile is staged;
//   * a general fp32 activation / gradient is split exactly into THREE bf16 terms hi + mid + lo while its tile is staged
//     (three v_mfma_f32_16x16x32_bf16 per step): every product bin x term is exact in fp32, accumulation is fp32;
//   * an activation known to be a quantiser output x_q = idx / n_a (the `relu(act_q(bn(.)))` tensors that feed conv2 / conv3
//     of a bottleneck; idx <= r * n_a <= 2048) enters as ONE f16 term idx = rint(x_q * n_a) (v_mfma_f32_16x16x32_f16): the sum is
//     an integer, divided by n * n_a once in the epilogue;
//   * filter gradient: dy in three bf16 terms, x in three (the six leading pairs: everything above 2^-24 relative) or, for a
//     level tensor, its index in two exact bf16 terms (all six pairs): fp32-grade sums, deterministic split-K slabs reduced in
//     fixed order by alignq_conv3x3_wgrad_reduce_multi (no atomics, no zero-fill).
// MFMA roles: A operand = filter side (rows of D = output channels, 4 consecutive per lane -> one float4 store per lane),
// B operand = pixel side (columns of D).  One workgroup = 256 threads = 4 waves; a wave owns 64 channels x 16*TM pixels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/alignq.h"
#include "wgrad_reduce_body.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

constexpr int BK = 64;            // k per step of the forward / data-gradient GEMM (two MFMA k-substeps)

// 4 rows x 16 columns block at `p` (this lane's row (lane & 15) >> 2, columns 4 * (lane & 3)), transposed by the LDS hardware:
// the lane receives column (lane & 15) of the 4 rows.  EXEC all ones at every call.
__device__ __forceinline__ s16x4 tr_read(const u16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
__device__ __forceinline__ s16x8 join8(s16x4 a, s16x4 b) { return (s16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }

template <bool F16>
__device__ __forceinline__ f32x4 mfma16(s16x8 a, s16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// exact three-way split of four floats: v == hi + mid + lo.  Two values per v_cvt_pk_bf16_f32 and per v_pk_add_f32 (round 5: the
// element-wise form compiled to 26 vector instructions per four values, this one to 20 - the same bits; these kernels are
// bound by vector issue, a wave64 instruction holding its SIMD for four cycles)
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_bf16x2(f32x2 v) { return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 bf16x2_as_f32(unsigned b) {
  return (f32x2){__builtin_bit_cast(float, b << 16), __builtin_bit_cast(float, b & 0xffff0000u)};
}
__device__ __forceinline__ void split3(const f32x4 v, s16x4& h, s16x4& m, s16x4& l) {
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const unsigned ha = cvt_bf16x2(a), hb = cvt_bf16x2(b);
  const f32x2 ra = a - bf16x2_as_f32(ha), rb = b - bf16x2_as_f32(hb);
  const unsigned ma = cvt_bf16x2(ra), mb = cvt_bf16x2(rb);
  const f32x2 sa = ra - bf16x2_as_f32(ma), sb = rb - bf16x2_as_f32(mb);
  h = __builtin_bit_cast(s16x4, (u32x2){ha, hb});
  m = __builtin_bit_cast(s16x4, (u32x2){ma, mb});
  l = __builtin_bit_cast(s16x4, (u32x2){cvt_bf16x2(sa), cvt_bf16x2(sb)});
}
// integer-valued floats |v| < 2^16 in two exact bf16 terms
__device__ __forceinline__ void split2(const f32x4 v, s16x4& h, s16x4& l) {
  const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
  const unsigned ha = cvt_bf16x2(a), hb = cvt_bf16x2(b);
  h = __builtin_bit_cast(s16x4, (u32x2){ha, hb});
  l = __builtin_bit_cast(s16x4, (u32x2){cvt_bf16x2(a - bf16x2_as_f32(ha)), cvt_bf16x2(b - bf16x2_as_f32(hb))});
}
template <bool F16>
__device__ __forceinline__ s16x4 to_half4(const f32x4 v) {      // exact for the integers this file feeds it
  if constexpr (F16) {
    const f16x4 h = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
    return __builtin_bit_cast(s16x4, h);
  } else {
    const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return __builtin_bit_cast(s16x4, h);
  }
}
__device__ __forceinline__ f32x4 rint4(const f32x4 v, float s) {
  return (f32x4){rintf(v[0] * s), rintf(v[1] * s), rintf(v[2] * s), rintf(v[3] * s)};
}

// Workgroups that share operand rows on one XCD (blockIdx round-robins over the 8 XCDs, each with its own L2): bijective for any n.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Forward and data gradient:  out[row m][col n] = (1 / den) * sum_k  A[m][k] * Wb[n][k]
//   rows m  : pixels of the ROW grid [img][Hr][Wr] (forward: output pixels; data gradient: input pixels, or - SCATTER, the 1x1
//             stride-2 data gradient - dy's own pixels whose result goes to input pixel (2h, 2w) with zeros around it);
//   A[m][k] : k = (tap, channel) of the activation-side tensor xa ([img][Ha][Wa][CA], forward x, data gradient dy) at pixel
//             (hr * S + sgn * (ky - PAD), wr * S + sgn * (kx - PAD)), zero outside the image;
//   Wb[n][k]: forward (WTR = false) w[n][tap][c] - rows k-contiguous; data gradient (WTR = true) w[c][tap][n]: the staged image is
//             [k][n] and the fragments come through ds_read_b64_tr_b16.
struct QG {
  const float* xa; const u16* w; float* out;       // w: the filter's integer bins as bf16 (MODE 0) / f16 (MODE 1) bit patterns
  int Mg, groups, tiles_per_group, n_tiles;       // rows per group (a tile never straddles two groups), row tiles, column tiles
  int N, CA, KC;                                   // output columns, channels of xa, KC = CA / BK k-steps per tap
  int Hr, Wr, Ha, Wa, S, sgn;                      // row grid, xa grid, stride from row grid to xa grid, tap direction
  int wrow, wtap;                                  // filter strides (elements): between rows of the staged filter image, per tap
  int Ho, Wo;                                      // SCATTER: the output grid (H_in, W_in of the convolution)
  float nlev, xlev;                                // the filter is bins / nlev; MODE 1: index = rint(x * xlev)
  double* bn_part;                                 // forward: [groups][tiles_per_group][N][2] {sum y, sum y^2} or nullptr
  // split-K (data gradient of the small-M, long-K layers): workgroup (tile, split) runs its share of the k steps and leaves RAW sums
  // in part[split][out_elems] at the output's own addresses; qgemm_ksplit_reduce_kernel adds the splits in order and divides
  int ksplit; float* part; int64_t out_elems;
  int part_splits;                                 // host only: images of the output the caller's workspace holds (the planned split)
};

// OCC: workgroups per CU the register / LDS budget is sized for (a small tile runs 3-4 of them: latency hiding comes from the
// other workgroups' waves, the staging of one overlaps the MFMAs of another)
// KM: 0 = 1x1 (one tap); 1 = 3x3, the pixel-side tile gathered per tap (any stride); 2 = 3x3 stride 1 with a HALO image: the
// workgroup's BM consecutive pixels plus W + 1 pixels on either side are staged ONCE per channel chunk (rows = flattened pixel
// index) and the nine taps read them at row offsets dy * W + dx; a lane whose tap falls outside its image reads an all-zero row
// instead (address select, no branch).  The pixel-side global loads, splits and LDS writes drop by ~9 BM / (BM + 2 W + 2).
// KM 3 = the DATA GRADIENT of a 3x3 stride-2 convolution (padding 1, even H_in and W_in) by parity class: the input pixels
// (2i + py, 2j + px) of one class (py, px) = one "group" of rows; dx[2i + py, 2j + px] = sum over the taps (ky, kx) with
// ky = py + 1 (mod 2), kx = px + 1 (mod 2) of dy[i + (py + 1 - ky) / 2, j + (px + 1 - kx) / 2] * W[ky][kx] - 1, 2, 2 or 4 taps instead of
// the 9 a gather over every tap would run with 5 to 8 of them multiplying zeros; rows = the half grid, gathered per tap like KM 1.
constexpr int kHaloW = 56;        // widest image the halo form takes (LDS is sized for it)
// XI (MODE 1 only): the level operand arrives as its int16 index (N2 on the Office path: 2 B per element through the CU's load path
// instead of 4, no rint on the way to the f16 term).
template <int WN, int TM, int MODE, bool WTR, int KM, bool SCATTER, int OCC, bool XI = false>
__global__ __launch_bounds__(256, OCC) void qgemm_kernel(const QG a) {
  static_assert(!XI || MODE == 1, "only a level operand has an index form");
  using RA = typename std::conditional<XI, s16x4, f32x4>::type;
  constexpr int WM = 4 / WN, BM = WM * 16 * TM, BN = 64 * WN;
  constexpr int TA = MODE == 0 ? 3 : 1;
  constexpr bool F16 = MODE == 1;
  constexpr bool KS3 = KM != 0, HALO = KM == 2, S2D = KM == 3;
  static_assert(!S2D || (WTR && !SCATTER && MODE == 0), "the parity-class form is a data gradient");
  constexpr int KB = (HALO && MODE == 0) ? 32 : BK;      // k per step (the three-plane halo image would not fit at 64)
  // halfwords per row of a direct image.  ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+32):
  // sixteen different rows, half of them 16 B further along k.  Rows of 96 / 160 B (32 B of padding) put those sixteen 16-byte
  // pieces on sixteen different slots of the 256-B bank row; 80 / 144 B rows (16 B of padding) cost two extra cycles per read.
  // The 128x128 three-plane tile keeps 144 B (160 B rows would not leave room for two workgroups per CU).
  // The data gradient (WTR) reads the pixel side with two ds_read_b64 per fragment (see the k order below): 80 / 144 B rows are the
  // conflict-free ones for those.
  constexpr int LDX = KB + ((WTR || (TA == 3 && BM == 128 && BN == 128)) ? 8 : 16);
  constexpr int XR = HALO ? BM + 2 * kHaloW + 3 : BM;    // rows of the pixel-side image (+ the zero row)
  constexpr int NA = (XR * (KB / 4) + 255) / 256;        // float4 per thread of the pixel-side tile
  constexpr int NB = (BN * KB / 8) / 256;                // 16-byte pieces (8 bins) per thread of the filter tile
  static_assert(NB >= 1, "filter tile");
  constexpr int LDN = BN + 16;                           // halfwords per row of the transposed filter image
  constexpr int XPL = XR * LDX;                          // halfwords per pixel-side plane
  constexpr int WSZ = WTR ? KB * LDN : BN * LDX;
  constexpr int RED_HW = BN * WM * 16 * 4;      // halfwords of the epilogue's statistics buffer [BN][WM * 16][2] floats
  constexpr int LDS_HW = (TA * XPL + WSZ) > RED_HW ? (TA * XPL + WSZ) : RED_HW;
  __shared__ __attribute__((aligned(16))) u16 lds[LDS_HW];
  u16* const Xs = lds;
  u16* const Ws = lds + TA * XPL;

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wn = wv % WN, wm = wv / WN;
  int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int ksp = pid % a.ksplit;                  // (the splits of one tile sit next to each other: same XCD, same operand rows)
  pid /= a.ksplit;
  const int nt = pid % a.n_tiles, mt_all = pid / a.n_tiles;
  const int grp = mt_all / a.tiles_per_group, mt = mt_all % a.tiles_per_group;
  const int m_lo = grp * a.Mg + mt * BM, m_end = (grp + 1) * a.Mg;
  const int n0 = nt * BN;
  const int fr = lane & 15, fg = lane >> 4, fq = (lane & 15) >> 2, fc = 4 * (lane & 3);

  // ---- this thread's share of the pixel-side tile -----------------------------------------------------------------------------
  constexpr int C4 = KB / 4;                       // float4 per row
  const int c4 = tid % C4, rr = tid / C4;          // row rr + (256 / C4) i, float4 column c4
  constexpr int RP = 256 / C4;                     // rows per pass
  int pix[HALO ? 1 : NA];      // gather forms: pixel index of xa at the centre tap, -1: row beyond the group
  int hw[HALO ? 1 : NA];       // KM == 1: (h << 16) | w of that pixel in the xa grid
  const int total_px = a.Mg * a.groups;            // HALO: pixels of xa (row grid == xa grid)
  const int halo_rows = BM + 2 * a.Wa + 2;         // HALO: staged rows; row halo_rows is the zero row
  if constexpr (!HALO) {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int m = m_lo + rr + RP * i;
      if (m < m_end) {
        const int ml = S2D ? m - grp * a.Mg : m;         // (parity classes: every group covers all images)
        const int wr = ml % a.Wr, t = ml / a.Wr, hr = t % a.Hr, img = t / a.Hr;
        pix[i] = (img * a.Ha + hr * a.S) * a.Wa + wr * a.S;
        hw[i] = ((hr * a.S) << 16) | (wr * a.S);
      } else {
        pix[i] = -1;
        hw[i] = 0;
      }
    }
  } else {
    pix[0] = hw[0] = 0;
    // the zero row (and the never-staged rows behind it) of every plane
    for (int i = tid; i < TA * (XR - halo_rows) * (LDX / 4); i += 256) {
      const int pl = i / ((XR - halo_rows) * (LDX / 4)), j = i % ((XR - halo_rows) * (LDX / 4));
      *reinterpret_cast<s16x4*>(Xs + pl * XPL + halo_rows * LDX + 4 * j) = (s16x4){0, 0, 0, 0};
    }
  }
  // k steps: KM 0: channel chunks; KM 1: (tap, chunk) tap-major; KM 2: (chunk, tap) chunk-major (the halo image serves 9 taps)
  const int KC = a.CA / KB;
  const int py = S2D ? (grp >> 1) : 0, px = S2D ? (grp & 1) : 0, ntx = 1 + px;          // S2D: this class's taps: (1 + py) x (1 + px)
  const int nk = (S2D ? (1 + py) * ntx : (KS3 ? 9 : 1)) * KC;
  RA ra[NA];
  s16x8 rw[NB];
  auto load_a = [&](int64_t off, bool ok) -> RA {
    RA v;
    // (1x1: a row beyond the group only feeds its own output pixel, which is neither stored nor counted in the statistics - its
    // loads stay in bounds by the clamped address and need no zeroing; a 3x3 tap outside the image must contribute zeros)
    if constexpr (XI) {
      v = *reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(a.xa) + (ok ? off : 0));
      if (KS3 && !ok) v = (s16x4){0, 0, 0, 0};
    } else {
      v = *reinterpret_cast<const f32x4*>(a.xa + (ok ? off : 0));
      if (KS3 && !ok) v = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    return v;
  };

  auto fetch_x = [&](int kt) {          // KM 0 / 1: the step's tile; KM 2: chunk kt's halo image
    if constexpr (HALO) {
      const int c0 = kt * KB;
#pragma unroll
      for (int i = 0; i < NA; i++) {
        const int r = rr + RP * i;
        const int f = m_lo - a.Wa - 1 + r;
        const bool ok = r < halo_rows && f >= 0 && f < total_px;
        ra[i] = load_a((int64_t)f * a.CA + c0 + 4 * c4, ok);
      }
    } else {
      const int tap = KS3 ? kt / KC : 0, c0 = (KS3 ? kt % KC : kt) * KB;
      // S2D: tap (ty, tx) of the class: the dy pixel one further down / right for ky = 0 / kx = 0 (py = 1 / px = 1, ty = 0 / tx = 0)
      const int dy = S2D ? (py ? 1 - tap / ntx : 0) : (KS3 ? a.sgn * (tap / 3 - 1) : 0);
      const int dx = S2D ? (px ? 1 - tap % ntx : 0) : (KS3 ? a.sgn * (tap % 3 - 1) : 0);
#pragma unroll
      for (int i = 0; i < NA; i++) {
        bool ok = pix[i] >= 0;
        if (KS3) ok = ok && (unsigned)((hw[i] >> 16) + dy) < (unsigned)a.Ha && (unsigned)((hw[i] & 0xffff) + dx) < (unsigned)a.Wa;
        ra[i] = load_a((int64_t)(pix[i] + dy * a.Wa + dx) * a.CA + c0 + 4 * c4, ok);
      }
    }
  };
  auto fetch_w = [&](int kt) {
    int tap = !KS3 ? 0 : (HALO ? kt % 9 : kt / KC);
    const int c0 = (!KS3 ? kt : (HALO ? kt / 9 : kt % KC)) * KB;
    if (S2D) tap = (py ? 2 * (tap / ntx) : 1) * 3 + (px ? 2 * (tap % ntx) : 1);         // (ky, kx) of the class's tap (ty, tx)
    if (!WTR) {           // rows n of the tile, KB / 8 pieces of 8 k each
      constexpr int PPR = KB / 8, RPW = 256 / PPR;
#pragma unroll
      for (int i = 0; i < NB; i++)
        rw[i] = *reinterpret_cast<const s16x8*>(a.w + (int64_t)(n0 + tid / PPR + RPW * i) * a.wrow + (int64_t)tap * a.CA + c0 +
                                                8 * (tid % PPR));
    } else {              // rows k (channels of xa), BN / 8 pieces of 8 n each
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int idx = tid + 256 * i, kr = idx / (BN / 8), n8 = idx % (BN / 8);
        rw[i] = *reinterpret_cast<const s16x8*>(a.w + (int64_t)(c0 + kr) * a.wrow + (int64_t)tap * a.wtap + n0 + 8 * n8);
      }
    }
  };
  auto park_x = [&]() {
#pragma unroll
    for (int i = 0; i < NA; i++) {
      const int r = rr + RP * i;
      if (HALO && r >= halo_rows) continue;
      if (!HALO && NA * RP > BM && r >= BM) continue;
      const int o = r * LDX + 4 * c4;
      if constexpr (MODE == 0) {
        s16x4 h, m, l;
        split3(ra[i], h, m, l);
        *reinterpret_cast<s16x4*>(Xs + o) = h;
        *reinterpret_cast<s16x4*>(Xs + XPL + o) = m;
        *reinterpret_cast<s16x4*>(Xs + 2 * XPL + o) = l;
      } else if constexpr (XI) {
        const f16x4 h = {(_Float16)ra[i][0], (_Float16)ra[i][1], (_Float16)ra[i][2], (_Float16)ra[i][3]};      // exact: |index| <= 2048
        *reinterpret_cast<s16x4*>(Xs + o) = __builtin_bit_cast(s16x4, h);
      } else {
        *reinterpret_cast<s16x4*>(Xs + o) = to_half4<true>(rint4(ra[i], a.xlev));
      }
    }
  };
  auto park_w = [&]() {
#pragma unroll
    for (int i = 0; i < NB; i++) {
      if (!WTR) {
        constexpr int PPR = KB / 8, RPW = 256 / PPR;
        *reinterpret_cast<s16x8*>(Ws + (tid / PPR + RPW * i) * LDX + 8 * (tid % PPR)) = rw[i];
      } else {
        const int idx = tid + 256 * i, kr = idx / (BN / 8), n8 = idx % (BN / 8);
        *reinterpret_cast<s16x8*>(Ws + kr * LDN + 8 * n8) = rw[i];
      }
    }
  };

  // HALO: this lane's pixel of every 16-pixel tile: its row in the halo image and (h << 16) | w (-1: beyond the group: zero row)
  int hrow[HALO ? TM : 1], hhw[HALO ? TM : 1];
  if constexpr (HALO) {
#pragma unroll
    for (int tm = 0; tm < TM; tm++) {
      const int m = m_lo + (wm * TM + tm) * 16 + fr;
      hrow[tm] = (m - m_lo) + a.Wa + 1;
      hhw[tm] = m < m_end ? ((((m / a.Wa) % a.Ha) << 16) | (m % a.Wa)) : -1;
    }
  } else {
    hrow[0] = hhw[0] = 0;
  }

  f32x4 acc[4][TM];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < TM; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this workgroup's share of the k steps (split-K: whole channel chunks for the halo form, whose image serves nine steps)
  const int units = HALO ? KC : nk;
  const int kt0 = (HALO ? 9 : 1) * (int)((int64_t)units * ksp / a.ksplit);
  const int kt1 = (HALO ? 9 : 1) * (int)((int64_t)units * (ksp + 1) / a.ksplit);
  fetch_x(HALO ? kt0 / 9 : kt0);
  fetch_w(kt0);
  for (int kt = kt0; kt < kt1; kt++) {
    const int tap = HALO ? kt % 9 : 0;
    __syncthreads();                         // the previous step's fragment reads are done
    if (!HALO || tap == 0) park_x();
    park_w();
    __syncthreads();
    if (kt + 1 < kt1) {                      // in flight under this step's MFMAs
      fetch_w(kt + 1);
      if (!HALO) fetch_x(kt + 1);
    }
    if (HALO && tap == 6 && kt + 3 < kt1) fetch_x(kt / 9 + 1);     // the next chunk's halo image: three taps of MFMAs ahead
    int xrow[TM];
    if constexpr (HALO) {
      const int dy = a.sgn * (tap / 3 - 1), dx = a.sgn * (tap % 3 - 1);
#pragma unroll
      for (int tm = 0; tm < TM; tm++) {
        const bool ok = hhw[tm] >= 0 && (unsigned)((hhw[tm] >> 16) + dy) < (unsigned)a.Ha &&
                        (unsigned)((hhw[tm] & 0xffff) + dx) < (unsigned)a.Wa;
        xrow[tm] = ok ? hrow[tm] + dy * a.Wa + dx : halo_rows;
      }
    } else {
#pragma unroll
      for (int tm = 0; tm < TM; tm++) xrow[tm] = (wm * TM + tm) * 16 + fr;
    }
#pragma unroll
    for (int ks = 0; ks < KB / 32; ks++) {
      s16x8 wf[4];
#pragma unroll
      for (int tn = 0; tn < 4; tn++) {
        if (!WTR) {
          wf[tn] = *reinterpret_cast<const s16x8*>(Ws + (wn * 64 + tn * 16 + fr) * LDX + ks * 32 + 8 * fg);
        } else {
          // k order of the transposed filter image: lane group g takes rows {4g .. 4g+3} and {16 + 4g .. 16 + 4g + 3} of the
          // 32-row substep (the contraction index is a dummy; the pixel side below uses the same order), so that the 32 lanes of one
          // LDS cycle touch 8 consecutive rows = all 64 banks (rows 8 apart share banks at these row lengths)
          const u16* p = Ws + (ks * 32 + 4 * fg + fq) * LDN + wn * 64 + tn * 16 + fc;
          wf[tn] = join8(tr_read(p), tr_read(p + 16 * LDN));
        }
      }
#pragma unroll
      for (int tm = 0; tm < TM; tm++) {
        const u16* p = Xs + xrow[tm] * LDX + ks * 32 + (WTR ? 4 : 8) * fg;
#pragma unroll
        for (int t = TA - 1; t >= 0; t--) {           // smallest term first
          s16x8 xf;
          if constexpr (WTR) xf = join8(*reinterpret_cast<const s16x4*>(p + t * XPL), *reinterpret_cast<const s16x4*>(p + t * XPL + 16));
          else xf = *reinterpret_cast<const s16x8*>(p + t * XPL);
#pragma unroll
          for (int tn = 0; tn < 4; tn++) acc[tn][tm] = mfma16<F16>(wf[tn], xf, acc[tn][tm]);
        }
      }
    }
  }

  // ---- epilogue: lane = pixel (lane & 15) of the 16-pixel tile, 4 consecutive channels 4 * (lane >> 4) + e ----------------------
  const bool raw = a.ksplit > 1;                      // split-K: raw sums into this split's image
  const float den = raw ? 1.0f : (MODE == 1 ? a.nlev * a.xlev : a.nlev);
  float* const outp = raw ? a.part + (int64_t)ksp * a.out_elems : a.out;
  float bs[4][4], bq[4][4];                           // [tn][e]: this lane's sums over its TM pixels (forward statistics)
#pragma unroll
  for (int tn = 0; tn < 4; tn++)
#pragma unroll
    for (int e = 0; e < 4; e++) { bs[tn][e] = 0.f; bq[tn][e] = 0.f; }
#pragma unroll
  for (int tm = 0; tm < TM; tm++) {
    const int m = m_lo + (wm * TM + tm) * 16 + fr;
    const bool ok = m < m_end;
    int64_t orow = (int64_t)m * a.N;
    bool z01 = false, z10 = false;
    if (S2D) {
      const int ml = (ok ? m : m_lo) - grp * a.Mg;
      const int wr = ml % a.Wr, t = ml / a.Wr, hr = t % a.Hr, img = t / a.Hr;
      orow = ((int64_t)(img * a.Ho + 2 * hr + py) * a.Wo + 2 * wr + px) * a.N;
    }
    if (SCATTER) {
      const int mm = ok ? m : m_lo;
      const int wr = mm % a.Wr, t = mm / a.Wr, hr = t % a.Hr, img = t / a.Hr;
      orow = ((int64_t)(img * a.Ho + 2 * hr) * a.Wo + 2 * wr) * a.N;
      z01 = 2 * wr + 1 < a.Wo;
      z10 = 2 * hr + 1 < a.Ho;
    }
#pragma unroll
    for (int tn = 0; tn < 4; tn++) {
      const int n = n0 + wn * 64 + tn * 16 + 4 * fg;
      f32x4 v = acc[tn][tm];
      v = (f32x4){v[0] / den, v[1] / den, v[2] / den, v[3] / den};
      if (ok) {
        *reinterpret_cast<f32x4*>(outp + orow + n) = v;
        if (SCATTER) {
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          if (z01) *reinterpret_cast<f32x4*>(outp + orow + a.N + n) = z;
          if (z10) *reinterpret_cast<f32x4*>(outp + orow + (int64_t)a.Wo * a.N + n) = z;
          if (z01 && z10) *reinterpret_cast<f32x4*>(outp + orow + (int64_t)(a.Wo + 1) * a.N + n) = z;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) { bs[tn][e] += v[e]; bq[tn][e] += v[e] * v[e]; }
      }
    }
  }
  if (a.bn_part) {
    // per-channel {sum y, sum y^2} of the tile for the batch-norm that follows: the floats of every lane (TM values each) meet in
    // LDS and one thread per channel adds them in double, in a fixed order
    float* red = reinterpret_cast<float*>(lds);       // [BN channels][WM * 16 slots][2]
    constexpr int SL = WM * 16;
    static_assert(BN * SL * 4 == RED_HW, "reduction buffer size");
    __syncthreads();
#pragma unroll
    for (int tn = 0; tn < 4; tn++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int ch = wn * 64 + tn * 16 + 4 * fg + e;
        red[(ch * SL + wm * 16 + fr) * 2] = bs[tn][e];
        red[(ch * SL + wm * 16 + fr) * 2 + 1] = bq[tn][e];
      }
    __syncthreads();
    if (tid < BN) {
      double s0 = 0, s1 = 0;
#pragma unroll 8
      for (int s = 0; s < SL; s++) { s0 += (double)red[(tid * SL + s) * 2]; s1 += (double)red[(tid * SL + s) * 2 + 1]; }
      double* p = a.bn_part + (((int64_t)grp * a.tiles_per_group + mt) * a.N + n0 + tid) * 2;
      p[0] = s0;
      p[1] = s1;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Filter gradient:  slab[split][n][tap][c] = sum over the split's pixels m of  dy[m][n] * x[pixel(m, tap)][c]
// Both tiles are staged [k = pixel][channel] (32 pixels per step) and read through ds_read_b64_tr_b16; lane group g takes pixels
// {4g .. 4g+3} and {16 + 4g .. 16 + 4g + 3} of the step (the contraction index is a dummy: both operands use the same order), so
// that the 32 lanes of one LDS cycle touch 8 different rows = all 64 banks.  D rows = c (4 consecutive per lane -> float4 store).
constexpr int WK = 32;
struct QW {
  const float* x; const float* dy; float* slabs;
  int M, per;                      // dy pixels in all, pixels per split (multiple of WK)
  int CIN, COUT, c_tiles, n_tiles, taps;
  int Hr, Wr, Ha, Wa, S;           // dy grid, x grid, stride
  float xlev;                      // TX == 2: x index = rint(x * xlev), the slab is divided by xlev
};

template <int TC, int TN, int TX, bool GATHER, bool XI = false>
__global__ __launch_bounds__(256, 2) void qgemm_wgrad_kernel(const QW a) {
  static_assert(!XI || TX == 2, "only a level operand has an index form");
  using RX = typename std::conditional<XI, s16x4, f32x4>::type;
  constexpr int BC = 32 * TC, BNO = 32 * TN;       // tile: BC input channels x BNO output channels; wave = (16 TC) x (16 TN)
  constexpr int LDC = BC + 16, LDO = BNO + 16;
  constexpr int XPL = WK * LDC, DPL = WK * LDO;
  constexpr int NX = (WK * BC / 4) / 256, ND = (WK * BNO / 4) / 256;
  __shared__ __attribute__((aligned(16))) u16 lds[TX * XPL + 3 * DPL];
  u16* const Xs = lds;
  u16* const Ds = lds + TX * XPL;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wc = wv & 1, wo = wv >> 1;
  const int tiles = a.c_tiles * a.n_tiles * a.taps;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = pid / tiles, tile = pid % tiles;
  const int tap = tile % a.taps, ct = (tile / a.taps) % a.c_tiles, ot = tile / (a.taps * a.c_tiles);
  const int c0 = ct * BC, o0 = ot * BNO;
  const int m_begin = split * a.per, m_end = (m_begin + a.per < a.M) ? m_begin + a.per : a.M;
  const int dyy = GATHER ? tap / 3 - (a.taps == 9 ? 1 : 0) : 0, dxx = GATHER ? tap % 3 - (a.taps == 9 ? 1 : 0) : 0;

  RX rx[NX];
  f32x4 rd[ND];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i, kr = idx / (BC / 4), q4 = idx % (BC / 4);
      const int m = m0 + kr;
      bool ok = m < m_end;
      int64_t p = m;
      if (GATHER) {
        const int mm = ok ? m : 0;
        const int wr = mm % a.Wr, t = mm / a.Wr, hr = t % a.Hr, img = t / a.Hr;
        const int h = hr * a.S + dyy, w = wr * a.S + dxx;
        ok = ok && (unsigned)h < (unsigned)a.Ha && (unsigned)w < (unsigned)a.Wa;
        p = (int64_t)(img * a.Ha + h) * a.Wa + w;
      }
      if constexpr (XI) {
        rx[i] = *reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(a.x) + (ok ? p * a.CIN + c0 + 4 * q4 : 0));
        if (!ok) rx[i] = (s16x4){0, 0, 0, 0};
      } else {
        rx[i] = *reinterpret_cast<const f32x4*>(a.x + (ok ? p * a.CIN + c0 + 4 * q4 : 0));
        if (!ok) rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i, kr = idx / (BNO / 4), q4 = idx % (BNO / 4);
      const int m = m0 + kr;
      const bool ok = m < m_end;
      rd[i] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? (int64_t)m * a.COUT + o0 + 4 * q4 : 0));
      if (!ok) rd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i, kr = idx / (BC / 4), q4 = idx % (BC / 4);
      const int o = kr * LDC + 4 * q4;
      if constexpr (TX == 3) {
        s16x4 h, m, l;
        split3(rx[i], h, m, l);
        *reinterpret_cast<s16x4*>(Xs + o) = h;
        *reinterpret_cast<s16x4*>(Xs + XPL + o) = m;
        *reinterpret_cast<s16x4*>(Xs + 2 * XPL + o) = l;
      } else {
        s16x4 h, l;
        if constexpr (XI) split2((f32x4){(float)rx[i][0], (float)rx[i][1], (float)rx[i][2], (float)rx[i][3]}, h, l);
        else split2(rint4(rx[i], a.xlev), h, l);
        *reinterpret_cast<s16x4*>(Xs + o) = h;
        *reinterpret_cast<s16x4*>(Xs + XPL + o) = l;
      }
    }
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i, kr = idx / (BNO / 4), q4 = idx % (BNO / 4);
      const int o = kr * LDO + 4 * q4;
      s16x4 h, m, l;
      split3(rd[i], h, m, l);
      *reinterpret_cast<s16x4*>(Ds + o) = h;
      *reinterpret_cast<s16x4*>(Ds + DPL + o) = m;
      *reinterpret_cast<s16x4*>(Ds + 2 * DPL + o) = l;
    }
  };

  f32x4 acc[TC][TN];
#pragma unroll
  for (int i = 0; i < TC; i++)
#pragma unroll
    for (int j = 0; j < TN; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fc = 4 * (lane & 3);
  const int krow = 4 * fg + fq;                       // first pixel row of this lane's block (second: + 16)
  if (m_begin < m_end) fetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += WK) {
    __syncthreads();
    park();
    __syncthreads();
    if (m0 + WK < m_end) fetch(m0 + WK);
    s16x8 xf[TC][TX];
#pragma unroll
    for (int tc = 0; tc < TC; tc++)
#pragma unroll
      for (int t = 0; t < TX; t++) {
        const u16* p = Xs + t * XPL + krow * LDC + (wc * TC + tc) * 16 + fc;
        xf[tc][t] = join8(tr_read(p), tr_read(p + 16 * LDC));
      }
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      s16x8 df[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const u16* p = Ds + t * DPL + krow * LDO + (wo * TN + tn) * 16 + fc;
        df[t] = join8(tr_read(p), tr_read(p + 16 * LDO));
      }
#pragma unroll
      for (int tc = 0; tc < TC; tc++) {
        f32x4 v = acc[tc][tn];
        if constexpr (TX == 3) {          // the six leading pairs, smallest first
          v = mfma16<false>(xf[tc][1], df[1], v);
          v = mfma16<false>(xf[tc][0], df[2], v);
          v = mfma16<false>(xf[tc][2], df[0], v);
          v = mfma16<false>(xf[tc][0], df[1], v);
          v = mfma16<false>(xf[tc][1], df[0], v);
          v = mfma16<false>(xf[tc][0], df[0], v);
        } else {                          // index = hi + lo exactly: all six pairs
          v = mfma16<false>(xf[tc][1], df[2], v);
          v = mfma16<false>(xf[tc][1], df[1], v);
          v = mfma16<false>(xf[tc][0], df[2], v);
          v = mfma16<false>(xf[tc][1], df[0], v);
          v = mfma16<false>(xf[tc][0], df[1], v);
          v = mfma16<false>(xf[tc][0], df[0], v);
        }
        acc[tc][tn] = v;
      }
    }
  }
  // D: column = output channel (lane & 15), rows = input channels 4 * (lane >> 4) + e
  float* slab = a.slabs + (int64_t)split * a.COUT * a.taps * a.CIN;
#pragma unroll
  for (int tn = 0; tn < TN; tn++) {
    const int o = o0 + (wo * TN + tn) * 16 + (lane & 15);
#pragma unroll
    for (int tc = 0; tc < TC; tc++) {
      const int c = c0 + (wc * TC + tc) * 16 + 4 * fg;
      f32x4 v = acc[tc][tn];
      if constexpr (TX == 2) v = (f32x4){v[0] / a.xlev, v[1] / a.xlev, v[2] / a.xlev, v[3] / a.xlev};
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(slab + ((int64_t)o * a.taps + tap) * a.CIN + c));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Filter gradient of the 3x3 stride-1 convolutions with a HALO image: a workgroup owns a 64 x 64 (c, n) tile for ALL nine taps.
// x's rows live in a RING of R = 2 W + 34 LDS rows indexed by the flattened pixel (slot = (pixel + W + 1) mod R): a step of 32
// pixels needs the window [m0 - W - 1, m0 + 33 + W) and only its last 32 rows are new - each step stages 32 rows of x and 32
// rows of dy, whatever W is.  Tap (dy, dx) reads x at slot offset dy * W + dx; a pixel whose tap leaves its image reads the
// all-zero row (ds_read_b64_tr_b16 takes one row address per lane: an address select, no branch).  Nine accumulator sets per wave
// (32 x 32 of the tile each).  Against the per-tap form above, the staging of x and dy is shared by the nine taps.
template <int TX, bool XI = false>
__global__ __launch_bounds__(256, TX == 2 ? 2 : 1) void qgemm_wgrad3_kernel(const QW a) {
  static_assert(!XI || TX == 2, "only a level operand has an index form");
  using RX = typename std::conditional<XI, s16x4, f32x4>::type;
  constexpr int BC = 64, BNO = 64, LDC = BC + 16;
  constexpr int XR = 2 * kHaloW + 34 + 1;            // ring rows at the widest image + the zero row
  constexpr int XPL = XR * LDC, DPL = WK * LDC;
  constexpr int NX = (WK * BC / 4) / 256, ND = (WK * BNO / 4) / 256;
  __shared__ __attribute__((aligned(16))) u16 lds[TX * XPL + 3 * DPL];
  u16* const Xs = lds;
  u16* const Ds = lds + TX * XPL;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wc = wv & 1, wo = wv >> 1;
  const int tiles = a.c_tiles * a.n_tiles;
  const int pid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = pid / tiles, tile = pid % tiles;
  const int ct = tile % a.c_tiles, ot = tile / a.c_tiles;
  const int c0 = ct * BC, o0 = ot * BNO;
  const int m_begin = split * a.per, m_end = (m_begin + a.per < a.M) ? m_begin + a.per : a.M;
  const int W = a.Wa, H = a.Ha;
  const int R = 2 * W + 34;                          // ring rows; row R is the zero row
  for (int i = tid; i < TX * (LDC / 4); i += 256)
    *reinterpret_cast<s16x4*>(Xs + (i / (LDC / 4)) * XPL + R * LDC + 4 * (i % (LDC / 4))) = (s16x4){0, 0, 0, 0};
  RX rx[NX];
  f32x4 rd[ND];
  // x rows [f0, f0 + 32) that lie below f_hi (pixels outside the tensor are zeros)
  auto fetch_x = [&](int f0, int f_hi) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i, f = f0 + idx / (BC / 4), q4 = idx % (BC / 4);
      const bool ok = f < f_hi && f >= 0 && f < a.M;
      if constexpr (XI) {
        rx[i] = *reinterpret_cast<const s16x4*>(reinterpret_cast<const short*>(a.x) + (ok ? (int64_t)f * a.CIN + c0 + 4 * q4 : 0));
        if (!ok) rx[i] = (s16x4){0, 0, 0, 0};
      } else {
        rx[i] = *reinterpret_cast<const f32x4*>(a.x + (ok ? (int64_t)f * a.CIN + c0 + 4 * q4 : 0));
        if (!ok) rx[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  auto park_x = [&](int f0, int f_hi) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
      const int idx = tid + 256 * i, f = f0 + idx / (BC / 4), q4 = idx % (BC / 4);
      if (f >= f_hi) continue;
      const int o = ((f + W + 1) % R) * LDC + 4 * q4;
      if constexpr (TX == 3) {
        s16x4 h, m, l;
        split3(rx[i], h, m, l);
        *reinterpret_cast<s16x4*>(Xs + o) = h;
        *reinterpret_cast<s16x4*>(Xs + XPL + o) = m;
        *reinterpret_cast<s16x4*>(Xs + 2 * XPL + o) = l;
      } else {
        s16x4 h, l;
        if constexpr (XI) split2((f32x4){(float)rx[i][0], (float)rx[i][1], (float)rx[i][2], (float)rx[i][3]}, h, l);
        else split2(rint4(rx[i], a.xlev), h, l);
        *reinterpret_cast<s16x4*>(Xs + o) = h;
        *reinterpret_cast<s16x4*>(Xs + XPL + o) = l;
      }
    }
  };
  auto fetch_d = [&](int m0) {
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i, kr = idx / (BNO / 4), q4 = idx % (BNO / 4);
      const int m = m0 + kr;
      const bool ok = m < m_end;
      rd[i] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? (int64_t)m * a.COUT + o0 + 4 * q4 : 0));
      if (!ok) rd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park_d = [&]() {
#pragma unroll
    for (int i = 0; i < ND; i++) {
      const int idx = tid + 256 * i, kr = idx / (BNO / 4), q4 = idx % (BNO / 4);
      const int o = kr * LDC + 4 * q4;
      s16x4 h, m, l;
      split3(rd[i], h, m, l);
      *reinterpret_cast<s16x4*>(Ds + o) = h;
      *reinterpret_cast<s16x4*>(Ds + DPL + o) = m;
      *reinterpret_cast<s16x4*>(Ds + 2 * DPL + o) = l;
    }
  };

  f32x4 acc[9][2][2];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fc = 4 * (lane & 3);
  const int kp = 4 * fg + fq;                         // this lane's pixel of the step (second block: + 16)
  if (m_begin < m_end) {
    // prologue: the window's old part [m_begin - W - 1, m_begin + W + 1); the loop adds 32 rows per step
    const int f_hi = m_begin + W + 1;
    for (int f0 = m_begin - W - 1; f0 < f_hi; f0 += WK) {
      fetch_x(f0, f_hi);
      park_x(f0, f_hi);
    }
    fetch_x(f_hi, f_hi + WK);
    fetch_d(m_begin);
  }
  for (int m0 = m_begin; m0 < m_end; m0 += WK) {
    __syncthreads();                                  // the previous step's reads are done (its oldest 32 rows may go)
    park_x(m0 + W + 1, m0 + W + 1 + WK);
    park_d();
    __syncthreads();
    if (m0 + WK < m_end) {
      fetch_x(m0 + WK + W + 1, m0 + 2 * WK + W + 1);
      fetch_d(m0 + WK);
    }
    // (h, w) of this lane's two pixels, and the ring slot of pixel m0 + kp
    const int ma = m0 + kp, mb = ma + 16;
    const int wa = ma % W, ha = (ma / W) % H, wb = mb % W, hb = (mb / W) % H;
    const int sa = (ma + W + 1) % R;
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
      const bool oka = (unsigned)(ha + dy) < (unsigned)H && (unsigned)(wa + dx) < (unsigned)W;
      const bool okb = (unsigned)(hb + dy) < (unsigned)H && (unsigned)(wb + dx) < (unsigned)W;
      int ra = sa + dy * W + dx, rb = ra + 16;
      ra += ra < 0 ? R : 0; ra -= ra >= R ? R : 0;
      rb += rb < 0 ? R : 0; rb -= rb >= R ? R : 0;
      ra = oka ? ra : R;
      rb = okb ? rb : R;
#pragma unroll
      for (int tc = 0; tc < 2; tc++) {
        s16x8 xf[TX];
#pragma unroll
        for (int t = 0; t < TX; t++) {
          const u16* p = Xs + t * XPL + (wc * 2 + tc) * 16 + fc;
          xf[t] = join8(tr_read(p + ra * LDC), tr_read(p + rb * LDC));
        }
#pragma unroll
        for (int tn = 0; tn < 2; tn++) {
          s16x8 df[3];
#pragma unroll
          for (int t = 0; t < 3; t++) {
            const u16* p = Ds + t * DPL + kp * LDC + (wo * 2 + tn) * 16 + fc;
            df[t] = join8(tr_read(p), tr_read(p + 16 * LDC));
          }
          f32x4 v = acc[tap][tc][tn];
          if constexpr (TX == 3) {
            v = mfma16<false>(xf[1], df[1], v);
            v = mfma16<false>(xf[0], df[2], v);
            v = mfma16<false>(xf[2], df[0], v);
            v = mfma16<false>(xf[0], df[1], v);
            v = mfma16<false>(xf[1], df[0], v);
            v = mfma16<false>(xf[0], df[0], v);
          } else {
            v = mfma16<false>(xf[1], df[2], v);
            v = mfma16<false>(xf[1], df[1], v);
            v = mfma16<false>(xf[0], df[2], v);
            v = mfma16<false>(xf[1], df[0], v);
            v = mfma16<false>(xf[0], df[1], v);
            v = mfma16<false>(xf[0], df[0], v);
          }
          acc[tap][tc][tn] = v;
        }
      }
    }
  }
  float* slab = a.slabs + (int64_t)split * a.COUT * 9 * a.CIN;
#pragma unroll
  for (int tap = 0; tap < 9; tap++)
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
      const int o = o0 + (wo * 2 + tn) * 16 + (lane & 15);
#pragma unroll
      for (int tc = 0; tc < 2; tc++) {
        const int c = c0 + (wc * 2 + tc) * 16 + 4 * fg;
        f32x4 v = acc[tap][tc][tn];
        if constexpr (TX == 2) v = (f32x4){v[0] / a.xlev, v[1] / a.xlev, v[2] / a.xlev, v[3] / a.xlev};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(slab + ((int64_t)o * 9 + tap) * a.CIN + c));
      }
    }
}

// stand-alone reduction of the split-K slabs (a whole-model step defers it to alignq_conv3x3_wgrad_reduce_multi: same body, same order)
__global__ __launch_bounds__(1024) void qgemm_slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs, int n_elem,
                                                                 float* __restrict__ dw) {
  __shared__ __attribute__((aligned(16))) float part[4096];
  alignq_wgr::wgrad_reduce_body(slabs, n_slabs, n_elem, dw, blockIdx.x, part);
}

// The integer bins of quantised filters, b = rint(W_q * n), as bf16 and as f16 bit patterns (both exact for |b| <= 255): the
// operands of the GEMMs above.  Multi-tensor: blockIdx.y = filter.
constexpr int kPackMax = 64;
struct PackChunk {
  const float* w[kPackMax];
  u16* bf[kPackMax];
  u16* hf[kPackMax];
  int64_t n[kPackMax];
};
__global__ __launch_bounds__(256) void qgemm_pack_kernel(const PackChunk c, float nlev) {
  const int t = blockIdx.y;
  const float* __restrict__ w = c.w[t];
  u16* __restrict__ bf = c.bf[t];
  u16* __restrict__ hf = c.hf[t];
  const int64_t n4 = c.n[t] >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const f32x4 b = rint4(*reinterpret_cast<const f32x4*>(w + 4 * i), nlev);
    *reinterpret_cast<s16x4*>(bf + 4 * i) = to_half4<false>(b);
    *reinterpret_cast<s16x4*>(hf + 4 * i) = to_half4<true>(b);
  }
}

// out = (part[0] + part[1] + ... in split order) / den, elementwise (float4): the closing pass of a split-K data gradient
__global__ __launch_bounds__(256) void qgemm_ksplit_reduce_kernel(const float* __restrict__ part, int S, int64_t n4, int64_t stride,
                                                                  float den, float* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part) + i);
    for (int s2 = 1; s2 < S; s2++) v += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(part + s2 * stride) + i);
    *(reinterpret_cast<f32x4*>(out) + i) = (f32x4){v[0] / den, v[1] / den, v[2] / den, v[3] / den};
  }
}

// Split-K choice: `blocks` workgroups of `units` sequential k units each on `slots` resident workgroup slots.  Modelled time =
// rounds x (units per split + 2 for a workgroup's prologue / epilogue); a split is taken when it saves a quarter (it costs a
// closing pass over S + 1 images of the output).  The layers this finds: M = B H W of layer4 / layer3 with K = 9 C or 2048.
constexpr int kMaxKSplit = 4;
inline int pick_ksplit(int64_t blocks, int slots, int units) {
  auto cost = [&](int S) { return (double)((blocks * S + slots - 1) / slots) * ((double)((units + S - 1) / S) + 2.0); };
  int best = 1;
  for (int S = 2; S <= kMaxKSplit && units / S >= 2; S++)
    if (cost(S) < cost(best)) best = S;
  return cost(best) <= 0.75 * cost(1) ? best : 1;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The stem of the Office ResNet-50 (dann_office/model/resnet.py:193-195: Conv2d_Q(3, 64, kernel_size = 7, stride = 2, padding = 3)):
// y[m][co] = (1 / n) sum_{ky, j} bins[co][ky][j] * x(m; ky, j),  j = kx * 3 + c < 21.  Channels-last x makes the 21 values of one
// filter row CONTIGUOUS in memory ((2 ow - 3) * 3 + j of input row 2 oh - 3 + ky), so the contraction index is laid out as
// k = ky * 24 + j (three zero-bin pads per filter row, 168 -> 192 = six MFMA k steps): a lane's B fragment - 8 consecutive k of ONE
// pixel - is then ONE 32-byte piece of one input row, fetched straight from global memory as two dword-aligned 16-byte loads (the
// 16 pixels of a wave's tile sit 24 B apart; the input is 34 MB and lives in L2); no im2col image, no LDS for the pixel side.  The
// three values behind j = 20 belong to the pixel's right neighbours and meet zero bins.  Each value is split into three exact bf16
// terms in registers; the filter's bins [64][192] stay in LDS for the whole workgroup.  A wave owns 16-pixel tiles (all 64
// channels, 4 accumulators) and walks over them; four workgroups per CU cover the gather's latency.  Measured (B = 56, 224 x 224):
// 112 us against MIOpen's 159-207; the counters say vector issue (753 instructions per tile at 4 cycles each = 54 us of it, 290 of
// them the split) + 21 us of MFMA, not memory.  Epilogue: y = sum / n, float4
// stores; per-workgroup per-channel {sum y, sum y^2} in double for the batch-norm that follows ([groups][workgroups per group][64][2]).
struct QS {
  const float* x; const u16* w; float* y;
  int H, W, Ho, Wo;            // input / output grids
  int tiles_pg, wg_pg;         // 16-pixel tiles per group, workgroups per group (grid = groups * wg_pg)
  int Mg;                      // output pixels per group
  float nlev;
  double* bn_part;
  uint64_t magic_w, magic_h;   // ceil(2^40 / Wo), ceil(2^40 / Ho)
};
constexpr int kStemK = 147, kStemLDW = 208;      // (192 k per filter) 416-byte rows (= 32 mod 64): conflict-free ds_read_b128 (see LDX above)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // a dword-aligned 16-byte global load

// the 8 values of k group g of a BORDER pixel, every element's bounds checked.  Not inlined: the compiler otherwise turns the rare
// path into straight-line predicated code that every tile executes (80 of 149 vector instructions per step).
typedef float f32x8 __attribute__((ext_vector_type(8)));
__device__ __attribute__((noinline)) f32x8 stem_border8(const char* xbytes, int org, int oh, int ow, int H, int W, int g) {
  const int ky = g / 3, jb = 8 * (g - 3 * ky);
  f32x8 v;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int j = jb + e, ih = 2 * oh - 3 + ky, iw = 2 * ow - 3 + j / 3;
    const bool in = g < 21 && j < 21 && (unsigned)ih < (unsigned)H && (unsigned)iw < (unsigned)W;
    const float ld = *reinterpret_cast<const float*>(xbytes + (in ? (unsigned)(org + ky * W * 3 + j) * 4u : 0u));
    v[e] = in ? ld : 0.f;
  }
  return v;
}

__global__ __launch_bounds__(256, 4) void qstem7_fwd_kernel(const QS a) {
  __shared__ __attribute__((aligned(16))) u16 Ws[64 * kStemLDW];
  __shared__ float red[64][16][2];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int fr = lane & 15, fg = lane >> 4;
  const int grp = blockIdx.x / a.wg_pg, wg = blockIdx.x % a.wg_pg;
  // the filter's 64 x 147 bins into the padded [64][192] image: 19 dword loads per thread, ALL in flight before the first is used (a
  // loop of single 2-byte loads waited for each one: ~48 L2 round trips = 70 us per workgroup, more than the rest of the kernel)
  {
    constexpr int ND = 64 * kStemK / 2, NL = (ND + 255) / 256;          // 4704 dwords
    unsigned wd[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int d = tid + 256 * i;
      wd[i] = reinterpret_cast<const unsigned*>(a.w)[d < ND ? d : 0];
    }
    for (int i = tid; i < 64 * kStemLDW / 2; i += 256) reinterpret_cast<unsigned*>(Ws)[i] = 0u;       // pads (and everything else) zero
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NL; i++) {
      const int d = tid + 256 * i;
      if (d < ND) {
#pragma unroll
        for (int hlf = 0; hlf < 2; hlf++) {
          const int f = 2 * d + hlf, co = f / kStemK, kk = f - co * kStemK, ky = kk / 21, j = kk - 21 * ky;
          Ws[co * kStemLDW + ky * 24 + j] = (u16)(hlf ? wd[i] >> 16 : wd[i] & 0xffffu);
        }
      }
    }
  }
  __syncthreads();
  const int rs = a.W * 3;
  const int m_base = grp * a.Mg;
  const char* const xbytes = reinterpret_cast<const char*>(a.x);
  const float rn = 1.0f / a.nlev;
  float bs[4][4], bq[4][4];
#pragma unroll
  for (int tn = 0; tn < 4; tn++)
#pragma unroll
    for (int e = 0; e < 4; e++) { bs[tn][e] = 0.f; bq[tn][e] = 0.f; }
  for (int t = wg * 4 + wv; t < a.tiles_pg; t += a.wg_pg * 4) {
    const int ml = t * 16 + fr;
    const int m = m_base + (ml < a.Mg ? ml : a.Mg - 1);       // (stem7_ok: every index below 2^30)
    // m = (img * Ho + oh) * Wo + ow by multiplication (q = m * ceil(2^40 / d) >> 40 is exact while m * d < 2^40: stem7_ok)
    const int r = (int)(((uint64_t)(unsigned)m * a.magic_w) >> 40), ow = m - r * a.Wo;
    const int img = (int)(((uint64_t)(unsigned)r * a.magic_h) >> 40), oh = r - img * a.Ho;
    // interior: the patch and the three elements read behind each of its rows lie inside the image's rows
    const bool inner = oh >= 2 && 2 * oh + 3 < a.H && ow >= 2 && 2 * ow + 4 < a.W;
    const int org = ((img * a.H + 2 * oh - 3) * a.W + (2 * ow - 3)) * 3;
    const bool plain = __builtin_amdgcn_ballot_w64(!inner) == 0;
    int fgv = fg;
    asm volatile("" : "+v"(fgv));      // (the per-step tap coordinates are re-derived per tile: hoisted out of the tile loop they spill)
    // the 8 values of step ks: k = 32 ks + 8 fg + e = 24 ky + jb + e
#define STEM_LOAD(KS, V)                                                                                                          \
    {                                                                                                                             \
      const int g = 4 * (KS) + fgv, ky = g / 3, jb = 8 * (g - 3 * ky);                                                           \
      if (plain) {                                                                                                                \
        const unsigned o = g < 21 ? (unsigned)(org + ky * rs + jb) * 4u : 0u;      /* (groups 21..23: zero bins) */               \
        const f32x4 v0 = *reinterpret_cast<const f32x4u*>(xbytes + o), v1 = *reinterpret_cast<const f32x4u*>(xbytes + o + 16);   \
        V[0] = v0[0]; V[1] = v0[1]; V[2] = v0[2]; V[3] = v0[3]; V[4] = v1[0]; V[5] = v1[1]; V[6] = v1[2]; V[7] = v1[3];           \
      } else {      /* a border tile (3 of 112 rows, the first and last tile of a row) */                                         \
        const f32x8 b8 = stem_border8(xbytes, org, oh, ow, a.H, a.W, g);                                                          \
        _Pragma("unroll") for (int e = 0; e < 8; e++) V[e] = b8[e];                                                               \
      }                                                                                                                           \
    }
    f32x4 acc[4];
#pragma unroll
    for (int tn = 0; tn < 4; tn++) acc[tn] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float cur[8], nxt[8];
    STEM_LOAD(0, cur)
#pragma unroll
    for (int ks = 0; ks < 6; ks++) {
      if (ks < 5) STEM_LOAD(ks + 1, nxt)              // in flight under this step's splits and MFMAs
      s16x4 h0, m0, l0, h1, m1, l1;
      split3((f32x4){cur[0], cur[1], cur[2], cur[3]}, h0, m0, l0);
      split3((f32x4){cur[4], cur[5], cur[6], cur[7]}, h1, m1, l1);
      const s16x8 xh = join8(h0, h1), xm = join8(m0, m1), xl = join8(l0, l1);
#pragma unroll
      for (int tn = 0; tn < 4; tn++) {
        const s16x8 wf = *reinterpret_cast<const s16x8*>(Ws + (tn * 16 + fr) * kStemLDW + 32 * ks + 8 * fg);
        acc[tn] = mfma16<false>(wf, xl, acc[tn]);       // smallest term first
        acc[tn] = mfma16<false>(wf, xm, acc[tn]);
        acc[tn] = mfma16<false>(wf, xh, acc[tn]);
      }
#pragma unroll
      for (int e = 0; e < 8; e++) cur[e] = nxt[e];
      __builtin_amdgcn_sched_barrier(0);      // (else every step's fragments and split terms are formed up front: 250+ registers)
    }
#undef STEM_LOAD
    if (ml < a.Mg) {
      float* yo = a.y + (int64_t)(m_base + ml) * 64 + 4 * fg;
#pragma unroll
      for (int tn = 0; tn < 4; tn++) {
        // sum / n as q = s * (1/n) with one residual correction (within an ulp of the quotient; a full division is ten vector
        // instructions per value, 160 per tile in a kernel that is short of issue slots)
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float q = acc[tn][e] * rn;
          v[e] = __fmaf_rn(__fmaf_rn(-q, a.nlev, acc[tn][e]), rn, q);
        }
        *reinterpret_cast<f32x4*>(yo + tn * 16) = v;
#pragma unroll
        for (int e = 0; e < 4; e++) { bs[tn][e] += v[e]; bq[tn][e] += v[e] * v[e]; }
      }
    }
  }
  if (a.bn_part) {      // channel ch = tn * 16 + 4 fg + e: this lane's sums over its pixels; 16 lanes (fr) x 4 waves per channel
#pragma unroll
    for (int w2 = 0; w2 < 4; w2++) {
      __syncthreads();
      if (wv == w2) {
#pragma unroll
        for (int tn = 0; tn < 4; tn++)
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int ch = tn * 16 + 4 * fg + e;
            if (w2 == 0) { red[ch][fr][0] = bs[tn][e]; red[ch][fr][1] = bq[tn][e]; }
            else { red[ch][fr][0] += bs[tn][e]; red[ch][fr][1] += bq[tn][e]; }
          }
      }
    }
    __syncthreads();
    if (tid < 64) {
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) { s0 += (double)red[tid][j][0]; s1 += (double)red[tid][j][1]; }
      double* p = a.bn_part + ((int64_t)blockIdx.x * 64 + tid) * 2;
      p[0] = s0;
      p[1] = s1;
    }
  }
}

// Filter gradient of the stem: slab[split][co][ky * 21 + j] = sum over the split's output pixels m of dy[m][co] * x(m; ky, j).  The
// contraction runs over PIXELS (32 per step), so both tiles are staged [pixel][column] and read through ds_read_b64_tr_b16 as in
// qgemm_wgrad_kernel: the dy tile [32][64] (three bf16 terms) and the gathered patch tile [32][192] (k' = ky * 24 + j, the 32-byte
// pieces of the forward; three bf16 terms; the pad columns j = 21..23 and k' >= 168 hold whatever lies there - every output column
// depends on its own k' only and the pads are never stored).  Wave (wc, wo) owns 6 of the 12 k' tiles x 2 of the 4 channel tiles:
// 12 accumulators; six leading term pairs per product.  Deterministic split slabs, reduced by the caller's closing reduction.
struct QSW {
  const float* x; const float* dy; float* slabs;
  int H, W, Ho, Wo;
  int M, per;                  // output pixels in all, per split (a multiple of 32)
  uint64_t magic_w, magic_h;
};
__global__ __launch_bounds__(256, 2) void qstem7_wgrad_kernel(const QSW a) {
  constexpr int LDC = 192 + 16, LDO = 64 + 16, XPL = WK * LDC, DPL = WK * LDO;
  __shared__ __attribute__((aligned(16))) u16 lds[3 * XPL + 3 * DPL];
  u16* const Xs = lds;
  u16* const Ds = lds + 3 * XPL;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wc = wv & 1, wo = wv >> 1;
  const int m_begin = blockIdx.x * a.per, m_end = (m_begin + a.per < a.M) ? m_begin + a.per : a.M;
  const char* const xbytes = reinterpret_cast<const char*>(a.x);
  const int rs = a.W * 3;
  float rx[3][8];
  f32x4 rd[2];
  auto fetch = [&](int m0) {
#pragma unroll
    for (int i = 0; i < 3; i++) {          // (pixel, 8-column group) items of the patch tile: 32 x 24
      const int idx = tid + 256 * i, p = idx / 24, g = idx - 24 * p;
      const int m = m0 + p;
      const bool ok = m < m_end;
      const int mm = ok ? m : m_end - 1;
      const int r = (int)(((uint64_t)(unsigned)mm * a.magic_w) >> 40), ow = mm - r * a.Wo;
      const int img = (int)(((uint64_t)(unsigned)r * a.magic_h) >> 40), oh = r - img * a.Ho;
      const bool inner = oh >= 2 && 2 * oh + 3 < a.H && ow >= 2 && 2 * ow + 4 < a.W;
      const int org = ((img * a.H + 2 * oh - 3) * a.W + (2 * ow - 3)) * 3;
      const int ky = g / 3, jb = 8 * (g - 3 * ky);
      if (inner) {
        const unsigned o = g < 21 ? (unsigned)(org + ky * rs + jb) * 4u : 0u;
        const f32x4 v0 = *reinterpret_cast<const f32x4u*>(xbytes + o), v1 = *reinterpret_cast<const f32x4u*>(xbytes + o + 16);
        rx[i][0] = v0[0]; rx[i][1] = v0[1]; rx[i][2] = v0[2]; rx[i][3] = v0[3];
        rx[i][4] = v1[0]; rx[i][5] = v1[1]; rx[i][6] = v1[2]; rx[i][7] = v1[3];
      } else {
        const f32x8 b8 = stem_border8(xbytes, org, oh, ow, a.H, a.W, g);
#pragma unroll
        for (int e = 0; e < 8; e++) rx[i][e] = b8[e];
      }
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; e++) rx[i][e] = 0.f;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + 256 * i, kr = idx >> 4, q4 = idx & 15;
      const int m = m0 + kr;
      const bool ok = m < m_end;
      rd[i] = *reinterpret_cast<const f32x4*>(a.dy + (ok ? (int64_t)m * 64 + 4 * q4 : 0));
      if (!ok) rd[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  auto park = [&]() {
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const int idx = tid + 256 * i, p = idx / 24, g = idx - 24 * p;
      const int o = p * LDC + 8 * g;
      s16x4 h0, m0, l0, h1, m1, l1;
      split3((f32x4){rx[i][0], rx[i][1], rx[i][2], rx[i][3]}, h0, m0, l0);
      split3((f32x4){rx[i][4], rx[i][5], rx[i][6], rx[i][7]}, h1, m1, l1);
      *reinterpret_cast<s16x8*>(Xs + o) = join8(h0, h1);
      *reinterpret_cast<s16x8*>(Xs + XPL + o) = join8(m0, m1);
      *reinterpret_cast<s16x8*>(Xs + 2 * XPL + o) = join8(l0, l1);
    }
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int idx = tid + 256 * i, kr = idx >> 4, q4 = idx & 15;
      const int o = kr * LDO + 4 * q4;
      s16x4 h, m, l;
      split3(rd[i], h, m, l);
      *reinterpret_cast<s16x4*>(Ds + o) = h;
      *reinterpret_cast<s16x4*>(Ds + DPL + o) = m;
      *reinterpret_cast<s16x4*>(Ds + 2 * DPL + o) = l;
    }
  };
  f32x4 acc[6][2];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fg = lane >> 4, fq = (lane & 15) >> 2, fc = 4 * (lane & 3);
  const int krow = 4 * fg + fq;                       // first pixel row of this lane's block (second: + 16), as qgemm_wgrad_kernel
  if (m_begin < m_end) fetch(m_begin);
  for (int m0 = m_begin; m0 < m_end; m0 += WK) {
    __syncthreads();
    park();
    __syncthreads();
    if (m0 + WK < m_end) fetch(m0 + WK);
    s16x8 df[2][3];
#pragma unroll
    for (int tn = 0; tn < 2; tn++)
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const u16* p = Ds + t * DPL + krow * LDO + (wo * 2 + tn) * 16 + fc;
        df[tn][t] = join8(tr_read(p), tr_read(p + 16 * LDO));
      }
#pragma unroll
    for (int tc = 0; tc < 6; tc++) {
      s16x8 xf[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const u16* p = Xs + t * XPL + krow * LDC + (wc * 6 + tc) * 16 + fc;
        xf[t] = join8(tr_read(p), tr_read(p + 16 * LDC));
      }
#pragma unroll
      for (int tn = 0; tn < 2; tn++) {
        f32x4 v = acc[tc][tn];            // the six leading pairs, smallest first
        v = mfma16<false>(xf[1], df[tn][1], v);
        v = mfma16<false>(xf[0], df[tn][2], v);
        v = mfma16<false>(xf[2], df[tn][0], v);
        v = mfma16<false>(xf[0], df[tn][1], v);
        v = mfma16<false>(xf[1], df[tn][0], v);
        v = mfma16<false>(xf[0], df[tn][0], v);
        acc[tc][tn] = v;
      }
    }
  }
  // D: column = output channel (lane & 15), rows = k' = 16 tile + 4 * (lane >> 4) + e -> (ky, j); j < 21 are the filter's elements
  float* slab = a.slabs + (int64_t)blockIdx.x * (64 * kStemK);
#pragma unroll
  for (int tn = 0; tn < 2; tn++) {
    const int co = (wo * 2 + tn) * 16 + (lane & 15);
#pragma unroll
    for (int tc = 0; tc < 6; tc++) {
      const int kp = (wc * 6 + tc) * 16 + 4 * fg, ky = kp / 24, j = kp - 24 * ky;
      if (ky < 7) {
#pragma unroll
        for (int e = 0; e < 4; e++)
          if (j + e < 21) slab[co * kStemK + ky * 21 + j + e] = acc[tc][tn][e];
      }
    }
  }
}

inline bool stem7_ok(int B, int H, int W, int groups) {
  return B >= 1 && groups >= 1 && B % groups == 0 && H >= 8 && W >= 8 && H <= 4096 && W <= 4096 && (int64_t)B * H * W * 3 < ((int64_t)1 << 30);
}
inline int stem7_wgs(int64_t Mg) {          // workgroups per group: <= 512, every wave at least one tile
  const int64_t tiles = (Mg + 15) / 16;
  const int64_t w = (tiles + 3) / 4;
  return (int)(w < 512 ? w : 512);
}

bool shape_ok(int B, int H, int W, int CIN, int COUT, int KS, int stride) {
  if (B < 1 || H < 1 || W < 1 || CIN < 64 || COUT < 64 || CIN % 64 || COUT % 64) return false;
  if (!((KS == 1 && (stride == 1 || stride == 2)) || (KS == 3 && (stride == 1 || stride == 2)))) return false;
  if (H > 32767 || W > 32767) return false;
  const int64_t Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  if ((int64_t)B * H * W * (CIN > COUT ? CIN : COUT) >= (int64_t)1 << 40) return false;
  if ((int64_t)B * H * W >= (int64_t)1 << 30 || (int64_t)B * Ho * Wo < 1) return false;
  return true;
}

template <int WN, int TM, int MODE, bool WTR, int KM, bool SCATTER, int OCC, bool XI = false>
int launch_g(QG a, hipStream_t st) {
  constexpr int BM = (4 / WN) * 16 * TM, BN = 64 * WN;
  a.tiles_per_group = (a.Mg + BM - 1) / BM;
  a.n_tiles = a.N / BN;
  const int blocks = a.groups * a.tiles_per_group * a.n_tiles;
  constexpr int KBh = (KM == 2 && MODE == 0) ? 32 : BK;
  // k units per workgroup (the parity-class form: 1 to 4 taps, 2.25 on average)
  const int units = KM == 2 ? a.CA / KBh : (KM == 1 ? 9 : (KM == 3 ? 2 : 1)) * (a.CA / BK);
  a.ksplit = (a.part && !a.bn_part) ? pick_ksplit(blocks, 256 * OCC, units) : 1;
  if (KM == 3 && a.ksplit > a.CA / BK) a.ksplit = 1;          // (a one-tap class must have a k unit for every split)
  if (a.part && a.ksplit > a.part_splits) return ALIGNQ_EINVAL;   // (the workspace query and this launch disagree on the plan: never write past it)
  hipLaunchKernelGGL((qgemm_kernel<WN, TM, MODE, WTR, KM, SCATTER, OCC, XI>), dim3(blocks * a.ksplit), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (a.ksplit > 1) {
    const int64_t n4 = a.out_elems / 4;
    const int gx = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(qgemm_ksplit_reduce_kernel, dim3(gx), dim3(256), 0, st, a.part, a.ksplit, n4, a.out_elems,
                       MODE == 1 ? a.nlev * a.xlev : a.nlev, a.out);
    e = hipGetLastError();
  }
  return e == hipSuccess ? 0 : (int)e;
}

// Tile choice: the largest of 128x128, 64x128, 64x64 (pixels x channels) that still gives the chip >= `want` workgroups
// (rows = all groups' rows; alignq_qconv_bn_parts reports the row tiles per group of the same choice).  The halo form of a level
// operand (64 channels per step) keeps to 64-row tiles: its image of 128 + 2 W + 2 rows would not fit the prefetch registers.
constexpr int kWantBlocks = 1024;        // (round 5 sweep 512 / 768 / 1024 / 1536: forward + data gradient 4717 / 4651 / 4656 / 4733 us per iteration)
inline int pick_tile(int64_t rows, int N, bool rows64_only = false, int want = kWantBlocks) {        // 0: 128x128, 1: 64x128, 2: 128x64, 3: 64x64
  const bool n128 = N % 128 == 0;
  auto blocks = [&](int bm, int bn) { return ((rows + bm - 1) / bm) * (N / bn); };
  if (!rows64_only && n128 && blocks(128, 128) >= want) return 0;
  if (n128 && (blocks(64, 128) >= want || rows64_only)) return (blocks(64, 128) >= want) ? 1 : 3;
  if (!rows64_only && blocks(128, 64) >= 2 * want) return 2;
  return 3;
}
inline int tile_rows(int cfg) { return (cfg == 0 || cfg == 2) ? 128 : 64; }
// (KM, MODE) -> whether the halo form runs and what pick_tile gets for it
inline bool halo_ok(int KS, int stride, int W) { return KS == 3 && stride == 1 && W <= kHaloW; }
inline int pick_tile_conv(int64_t rows, int N, bool halo, bool level) {
  return halo ? pick_tile(rows, N, level, kWantBlocks / 2) : pick_tile(rows, N);
}

template <int MODE, bool WTR, int KM, bool SCATTER, bool XI = false>
int launch_g_tiles(const QG& a, hipStream_t st) {
  switch (pick_tile_conv((int64_t)a.Mg * a.groups, a.N, KM == 2, MODE == 1)) {
    case 0: return launch_g<2, 4, MODE, WTR, KM, SCATTER, 2, XI>(a, st);
    case 1: return launch_g<2, 2, MODE, WTR, KM, SCATTER, (KM == 2 ? 2 : 3), XI>(a, st);
    case 2: return launch_g<1, 2, MODE, WTR, KM, SCATTER, 2, XI>(a, st);
    default: return launch_g<1, 1, MODE, WTR, KM, SCATTER, (KM == 2 ? (MODE == 0 ? 2 : 3) : 4), XI>(a, st);
  }
}

int wgrad_splits(int64_t M, int tiles, int want = 512) {
  int s = (want + tiles - 1) / tiles;
  const int64_t max_s = (M + 2 * WK - 1) / (2 * WK);       // at least two steps per split
  if (s > max_s) s = (int)max_s;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  return s;
}

}  // namespace

extern "C" {

int alignq_qconv_supported(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride) {
  return shape_ok(B, H_in, W_in, CIN, COUT, KS, stride) ? 1 : 0;
}

int alignq_qconv_bn_parts(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride, int groups, float x_levels) {
  if (!shape_ok(B, H_in, W_in, CIN, COUT, KS, stride) || groups < 1 || B % groups) return 0;
  const int Ho = (H_in - 1) / stride + 1, Wo = (W_in - 1) / stride + 1;
  // (the row tiles of alignq_qconv_fwd's tile choice for the same arguments)
  const int bm = tile_rows(pick_tile_conv((int64_t)B * Ho * Wo, COUT, halo_ok(KS, stride, W_in), x_levels != 0.0f));
  return (int)(((int64_t)(B / groups) * Ho * Wo + bm - 1) / bm);
}

int alignq_qconv_pack_weights(int T, const float* const* wt, const int64_t* n, int w_bit, void* const* bins_bf16, void* const* bins_f16,
                              void* stream) {
  if (T <= 0 || !wt || !n || !bins_bf16 || !bins_f16 || w_bit < 1 || w_bit > 8) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  for (int t0 = 0; t0 < T; t0 += kPackMax) {
    const int cnt = T - t0 < kPackMax ? T - t0 : kPackMax;
    PackChunk c;
    int64_t max_n = 0;
    for (int i = 0; i < cnt; i++) {
      if (!wt[t0 + i] || !bins_bf16[t0 + i] || !bins_f16[t0 + i] || n[t0 + i] < 4 || (n[t0 + i] & 3)) return ALIGNQ_EINVAL;
      c.w[i] = wt[t0 + i]; c.bf[i] = (u16*)bins_bf16[t0 + i]; c.hf[i] = (u16*)bins_f16[t0 + i]; c.n[i] = n[t0 + i];
      if (n[t0 + i] > max_n) max_n = n[t0 + i];
    }
    int gx = (int)((max_n / 4 + 1023) / 1024);
    if (gx < 1) gx = 1;
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(qgemm_pack_kernel, dim3(gx, cnt), dim3(256), 0, st, c, (float)((1 << w_bit) - 1));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return 0;
}

int alignq_qconv_fwd(const void* x, const void* w_bins, float* y, int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride,
                     int w_bit, float x_levels, int x_bin_bytes, int groups, double* bn_part, void* stream) {
  if (!x || !w_bins || !y || w_bit < 1 || w_bit > 8 || groups < 1) return ALIGNQ_EINVAL;
  if (x_bin_bytes != 0 && (x_bin_bytes != 2 || x_levels == 0.0f)) return ALIGNQ_EINVAL;      // int16 indices of a level tensor, or fp32
  if (!shape_ok(B, H_in, W_in, CIN, COUT, KS, stride) || B % groups) return ALIGNQ_EUNSUPPORTED;
  if (x_levels != 0.0f && !(x_levels >= 1.0f)) return ALIGNQ_EINVAL;
  const int Ho = (H_in - 1) / stride + 1, Wo = (W_in - 1) / stride + 1;
  QG a{};
  a.xa = (const float*)x; a.w = (const u16*)w_bins; a.out = y;
  a.Mg = (B / groups) * Ho * Wo; a.groups = groups;
  a.N = COUT; a.CA = CIN; a.KC = CIN / BK;
  a.Hr = Ho; a.Wr = Wo; a.Ha = H_in; a.Wa = W_in; a.S = stride; a.sgn = 1;
  a.wrow = KS * KS * CIN; a.wtap = CIN;
  a.Ho = Ho; a.Wo = Wo;
  a.nlev = (float)((1 << w_bit) - 1); a.xlev = x_levels;
  a.bn_part = bn_part;
  hipStream_t st = (hipStream_t)stream;
  if (x_bin_bytes == 2) {
    if (KS == 3 && halo_ok(KS, stride, W_in)) return launch_g_tiles<1, false, 2, false, true>(a, st);
    if (KS == 3) return launch_g_tiles<1, false, 1, false, true>(a, st);
    return launch_g_tiles<1, false, 0, false, true>(a, st);
  }
  if (KS == 3 && halo_ok(KS, stride, W_in))
    return x_levels != 0.0f ? launch_g_tiles<1, false, 2, false>(a, st) : launch_g_tiles<0, false, 2, false>(a, st);
  if (KS == 3) return x_levels != 0.0f ? launch_g_tiles<1, false, 1, false>(a, st) : launch_g_tiles<0, false, 1, false>(a, st);
  return x_levels != 0.0f ? launch_g_tiles<1, false, 0, false>(a, st) : launch_g_tiles<0, false, 0, false>(a, st);
}

int alignq_qconv_stem7_bn_parts(int B, int H_in, int W_in, int groups) {
  if (!stem7_ok(B, H_in, W_in, groups)) return 0;
  const int Ho = (H_in - 1) / 2 + 1, Wo = (W_in - 1) / 2 + 1;
  return stem7_wgs((int64_t)(B / groups) * Ho * Wo);
}

int alignq_qconv_stem7_fwd(const float* x, const void* w_bins, float* y, int B, int H_in, int W_in, int w_bit, int groups,
                           double* bn_part, void* stream) {
  if (!x || !w_bins || !y || w_bit < 1 || w_bit > 8) return ALIGNQ_EINVAL;
  if (!stem7_ok(B, H_in, W_in, groups)) return ALIGNQ_EUNSUPPORTED;
  QS a{};
  a.x = x; a.w = (const u16*)w_bins; a.y = y;
  a.H = H_in; a.W = W_in; a.Ho = (H_in - 1) / 2 + 1; a.Wo = (W_in - 1) / 2 + 1;
  a.Mg = (B / groups) * a.Ho * a.Wo;
  a.tiles_pg = (a.Mg + 15) / 16;
  a.wg_pg = stem7_wgs(a.Mg);
  a.nlev = (float)((1 << w_bit) - 1);
  a.bn_part = bn_part;
  a.magic_w = (((uint64_t)1 << 40) + a.Wo - 1) / a.Wo;
  a.magic_h = (((uint64_t)1 << 40) + a.Ho - 1) / a.Ho;
  hipLaunchKernelGGL(qstem7_fwd_kernel, dim3(groups * a.wg_pg), dim3(256), 0, (hipStream_t)stream, a);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

static int stem7_wgrad_splits(int64_t M) {
  const int64_t steps = (M + WK - 1) / WK;
  return (int)(steps < 512 ? steps : 512);
}

size_t alignq_qconv_stem7_wgrad_ws_bytes(int B, int H_in, int W_in) {
  if (!stem7_ok(B, H_in, W_in, 1)) return 0;
  const int Ho = (H_in - 1) / 2 + 1, Wo = (W_in - 1) / 2 + 1;
  return (size_t)stem7_wgrad_splits((int64_t)B * Ho * Wo) * 64 * kStemK * sizeof(float);
}

int alignq_qconv_stem7_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H_in, int W_in, int* n_slabs_out,
                             void* stream) {
  if (!x || !dy || !ws || (!dw && !n_slabs_out)) return ALIGNQ_EINVAL;
  if (!stem7_ok(B, H_in, W_in, 1)) return ALIGNQ_EUNSUPPORTED;
  QSW a{};
  a.x = x; a.dy = dy; a.slabs = (float*)ws;
  a.H = H_in; a.W = W_in; a.Ho = (H_in - 1) / 2 + 1; a.Wo = (W_in - 1) / 2 + 1;
  a.M = B * a.Ho * a.Wo;
  const int splits = stem7_wgrad_splits(a.M);
  const int64_t steps = ((int64_t)a.M + WK - 1) / WK;
  a.per = (int)((steps + splits - 1) / splits) * WK;
  a.magic_w = (((uint64_t)1 << 40) + a.Wo - 1) / a.Wo;
  a.magic_h = (((uint64_t)1 << 40) + a.Ho - 1) / a.Ho;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(qstem7_wgrad_kernel, dim3(splits), dim3(256), 0, st, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (n_slabs_out) { *n_slabs_out = splits; return 0; }      // deferred: the caller reduces (alignq_conv3x3_wgrad_reduce_multi)
  const int n_elem = 64 * kStemK;
  hipLaunchKernelGGL(qgemm_slab_reduce_kernel, dim3(alignq_wgr::wgrad_reduce_blocks(splits, n_elem)), dim3(1024), 0, st,
                     (const float*)ws, splits, n_elem, dw);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

// The split-K factor alignq_qconv_dgrad will choose for this layer: the SAME tile choice (pick_tile_conv) and cost model (pick_ksplit)
// as launch_g_tiles / launch_g take for the data gradient (MODE 0).  Round 6: the workspace query used to answer
// kMaxKSplit images of dx for every layer with dx <= 32 MB, although most of them never split - up to 128 MB allocated and dropped
// per convolution backward and reserved in the captured graph's pool.
static int dgrad_ksplit(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride) {
  const int Ho = (H_in - 1) / stride + 1, Wo = (W_in - 1) / stride + 1;
  int km, groups = 1;
  int64_t Mg;
  if (stride == 2 && KS == 3) { km = 3; groups = 4; Mg = (int64_t)B * Ho * Wo; }
  else if (stride == 2) { km = 0; Mg = (int64_t)B * Ho * Wo; }
  else { Mg = (int64_t)B * H_in * W_in; km = KS == 3 ? (halo_ok(KS, stride, W_in) ? 2 : 1) : 0; }
  const int N = CIN, CA = COUT;
  int bm, bn, occ;
  switch (pick_tile_conv(Mg * groups, N, km == 2, false)) {
    case 0: bm = 128; bn = 128; occ = 2; break;
    case 1: bm = 64; bn = 128; occ = km == 2 ? 2 : 3; break;
    case 2: bm = 128; bn = 64; occ = 2; break;
    default: bm = 64; bn = 64; occ = km == 2 ? 2 : 4; break;
  }
  const int64_t blocks = (int64_t)groups * ((Mg + bm - 1) / bm) * (N / bn);
  const int kbh = km == 2 ? 32 : BK;
  const int units = km == 2 ? CA / kbh : (km == 1 ? 9 : (km == 3 ? 2 : 1)) * (CA / BK);
  int ks = pick_ksplit(blocks, 256 * occ, units);
  if (km == 3 && ks > CA / BK) ks = 1;
  return ks;
}

size_t alignq_qconv_dgrad_ws_bytes(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride) {
  if (!shape_ok(B, H_in, W_in, CIN, COUT, KS, stride)) return 0;
  // split-K is for the layers with few row tiles: their dx is small (layer3 / layer4 of ResNet-50 at B = 56: 2.8 - 11 MB)
  const size_t out = (size_t)B * H_in * W_in * CIN * sizeof(float);
  if (out > ((size_t)32 << 20)) return 0;
  const int ks = dgrad_ksplit(B, H_in, W_in, CIN, COUT, KS, stride);
  return ks > 1 ? (size_t)ks * out : 0;
}

int alignq_qconv_dgrad(const float* dy, const void* w_bins, float* dx, int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride,
                       int w_bit, void* ws, void* stream) {
  if (!dy || !w_bins || !dx || w_bit < 1 || w_bit > 8) return ALIGNQ_EINVAL;
  if (!shape_ok(B, H_in, W_in, CIN, COUT, KS, stride)) return ALIGNQ_EUNSUPPORTED;
  if (KS == 3 && stride != 1 && ((H_in | W_in) & 1)) return ALIGNQ_EUNSUPPORTED;       // (the parity classes need even H_in, W_in)
  const int Ho = (H_in - 1) / stride + 1, Wo = (W_in - 1) / stride + 1;
  QG a{};
  a.xa = dy; a.w = (const u16*)w_bins; a.out = dx;
  a.groups = 1;
  a.N = CIN; a.CA = COUT; a.KC = COUT / BK;
  a.Ha = Ho; a.Wa = Wo; a.S = 1; a.sgn = -1;
  a.wrow = KS * KS * CIN; a.wtap = CIN;
  a.Ho = H_in; a.Wo = W_in;
  a.nlev = (float)((1 << w_bit) - 1); a.xlev = 0.f;
  a.bn_part = nullptr;
  a.out_elems = (int64_t)B * H_in * W_in * CIN;
  {
    const size_t wsb = alignq_qconv_dgrad_ws_bytes(B, H_in, W_in, CIN, COUT, KS, stride);
    a.part = (ws && wsb) ? (float*)ws : nullptr;
    a.part_splits = a.part ? (int)(wsb / ((size_t)a.out_elems * sizeof(float))) : 1;
  }
  hipStream_t st = (hipStream_t)stream;
  if (stride == 2 && KS == 3) {      // rows = the half grid, once per parity class of the input pixels (see KM 3)
    a.groups = 4;
    a.Mg = B * Ho * Wo; a.Hr = Ho; a.Wr = Wo;
    return launch_g_tiles<0, true, 3, false>(a, st);
  }
  if (stride == 2) {         // 1x1: rows = dy's pixels, scattered to (2h, 2w) with zeros at the other three pixels of the 2x2 cell
    a.Mg = B * Ho * Wo; a.Hr = Ho; a.Wr = Wo;
    return launch_g_tiles<0, true, 0, true>(a, st);
  }
  a.Mg = B * H_in * W_in; a.Hr = H_in; a.Wr = W_in;
  if (KS == 3) return halo_ok(KS, stride, W_in) ? launch_g_tiles<0, true, 2, false>(a, st) : launch_g_tiles<0, true, 1, false>(a, st);
  return launch_g_tiles<0, true, 0, false>(a, st);
}

static int wgrad_geometry(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride, int* tc, int* tn, int* tiles, int* splits,
                          int64_t* M) {
  const int Ho = (H_in - 1) / stride + 1, Wo = (W_in - 1) / stride + 1;
  *M = (int64_t)B * Ho * Wo;
  *tc = CIN % 128 == 0 ? 4 : 2;
  *tn = COUT % 128 == 0 ? 4 : 2;
  if (halo_ok(KS, stride, W_in)) {        // qgemm_wgrad3_kernel: 64 x 64 tiles, nine taps per workgroup
    *tc = *tn = 2;
    *tiles = (CIN / 64) * (COUT / 64);
    *splits = wgrad_splits(*M, *tiles, 512);       // (256 / 384: the kernel +6 / +9 us, its slab reduction -9 / -2 us per convolution: a wash)
    return 1;
  }
  *tiles = (CIN / (32 * *tc)) * (COUT / (32 * *tn)) * KS * KS;
  *splits = wgrad_splits(*M, *tiles, KS == 3 ? 1024 : 512);
  return 0;
}

size_t alignq_qconv_wgrad_ws_bytes(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride) {
  if (!shape_ok(B, H_in, W_in, CIN, COUT, KS, stride)) return 0;
  int tc, tn, tiles, splits;
  int64_t M;
  wgrad_geometry(B, H_in, W_in, CIN, COUT, KS, stride, &tc, &tn, &tiles, &splits, &M);
  return (size_t)splits * (size_t)COUT * KS * KS * CIN * sizeof(float);
}

int alignq_qconv_wgrad(const void* x, const float* dy, float* dw, void* ws, int B, int H_in, int W_in, int CIN, int COUT, int KS,
                       int stride, float x_levels, int x_bin_bytes, int* n_slabs_out, void* stream) {
  if (!x || !dy || !ws || (!dw && !n_slabs_out)) return ALIGNQ_EINVAL;
  if (x_bin_bytes != 0 && (x_bin_bytes != 2 || x_levels == 0.0f)) return ALIGNQ_EINVAL;
  if (!shape_ok(B, H_in, W_in, CIN, COUT, KS, stride)) return ALIGNQ_EUNSUPPORTED;
  if (x_levels != 0.0f && !(x_levels >= 1.0f)) return ALIGNQ_EINVAL;
  int tc, tn, tiles, splits;
  int64_t M;
  const int halo = wgrad_geometry(B, H_in, W_in, CIN, COUT, KS, stride, &tc, &tn, &tiles, &splits, &M);
  QW a{};
  a.x = (const float*)x; a.dy = dy; a.slabs = (float*)ws;
  a.M = (int)M;
  int64_t per = (M + splits - 1) / splits;
  per = (per + WK - 1) / WK * WK;
  a.per = (int)per;
  a.CIN = CIN; a.COUT = COUT; a.c_tiles = CIN / (32 * tc); a.n_tiles = COUT / (32 * tn); a.taps = KS * KS;
  a.Hr = (H_in - 1) / stride + 1; a.Wr = (W_in - 1) / stride + 1; a.Ha = H_in; a.Wa = W_in; a.S = stride;
  a.xlev = x_levels;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid(tiles * splits), blk(256);
  const bool gather = KS == 3 || stride != 1;
  const bool lev = x_levels != 0.0f, xi = x_bin_bytes == 2;
#define QW_LAUNCH(TC_, TN_)                                                                                              \
  do {                                                                                                                   \
    if (xi) {                                                                                                            \
      if (gather) hipLaunchKernelGGL((qgemm_wgrad_kernel<TC_, TN_, 2, true, true>), grid, blk, 0, st, a);                 \
      else hipLaunchKernelGGL((qgemm_wgrad_kernel<TC_, TN_, 2, false, true>), grid, blk, 0, st, a);                       \
    } else if (lev) {                                                                                                    \
      if (gather) hipLaunchKernelGGL((qgemm_wgrad_kernel<TC_, TN_, 2, true>), grid, blk, 0, st, a);                       \
      else hipLaunchKernelGGL((qgemm_wgrad_kernel<TC_, TN_, 2, false>), grid, blk, 0, st, a);                             \
    } else {                                                                                                             \
      if (gather) hipLaunchKernelGGL((qgemm_wgrad_kernel<TC_, TN_, 3, true>), grid, blk, 0, st, a);                       \
      else hipLaunchKernelGGL((qgemm_wgrad_kernel<TC_, TN_, 3, false>), grid, blk, 0, st, a);                             \
    }                                                                                                                    \
  } while (0)
  if (halo) {
    if (xi) hipLaunchKernelGGL((qgemm_wgrad3_kernel<2, true>), grid, blk, 0, st, a);
    else if (lev) hipLaunchKernelGGL((qgemm_wgrad3_kernel<2>), grid, blk, 0, st, a);
    else hipLaunchKernelGGL((qgemm_wgrad3_kernel<3>), grid, blk, 0, st, a);
  } else if (tc == 4 && tn == 4) QW_LAUNCH(4, 4);
  else if (tc == 2 && tn == 4) QW_LAUNCH(2, 4);
  else if (tc == 4 && tn == 2) QW_LAUNCH(4, 2);
  else QW_LAUNCH(2, 2);
#undef QW_LAUNCH
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (n_slabs_out) { *n_slabs_out = splits; return 0; }      // deferred: the caller reduces (alignq_conv3x3_wgrad_reduce_multi)
  const int n_elem = COUT * KS * KS * CIN;
  hipLaunchKernelGGL(qgemm_slab_reduce_kernel, dim3(alignq_wgr::wgrad_reduce_blocks(splits, n_elem)), dim3(1024), 0, st,
                     (const float*)ws, splits, n_elem, dw);
  e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

}  // extern "C"
