// corr_large_kernels.hip — corr(x, x) for batches ABOVE the 128 rows the fused site kernels hold on chip
// (128 < B <= ALIGNQ_MAX_CORR_BATCH = 1024): the exact-global-batch correlation of SURVEY.md §8f-N4 at the BASELINE global
// batches (224 / 512 / 1024) and a single GPU run at batch 256.
//
// Reference semantics: corr (ADMM tree model/quantization.py:134-137; Office tree :158-161 with eps = 1e-5); the reference
// takes any batch (utils/admm.py:17-27 sizes alterD / gamma by train_batch_size).
//
// Blocked form (the one-pass kernels need all rows of a feature tile in ONE workgroup, which stops at 128 rows):
//   1. corrl_stats_kernel  : per feature column mean and 1/(std + eps) over ALL B rows (one sweep, double accumulation)
//   2. corrl_gram_kernel   : the B x B Gram in 128 x 128 output blocks, upper block triangle only; a workgroup owns one block
//                            pair (I <= J) and a K range of 64-feature tiles: rows of block I and block J are standardised on
//                            load into LDS, the product runs on v_mfma_f32_32x32x2_f32 (exact fp32: this path is about
//                            reach, not speed; the bf16-split pipe of site4_kernels.hip is 5x faster per flop);
//                            one 128 x 128 partial slab per (pair, K split)
//   3. corrl_reduce_kernel : fixed-order sum of a pair's slabs, x 1/F, written to G[I][J] and mirrored to G[J][I]
// Backward (dG any, G symmetric in x):  dXh = S Xh,  S = (dG + dG^T)/F  [B,B];  dx = (dXh - mean_b dXh) rho
//                                       - xh * sum_b(dXh o xh) / ((B-1) std)          (torch's std backward: 0 where std == 0)
//   4. corrl_sym_kernel    : S
//   5. corrl_bwd_kernel<RB>: a workgroup owns a 32-feature tile with ALL rows: Xh tile in LDS (B x 33 floats), wave w the
//                            32-row blocks w, w+4, ... (RB of them: B <= 128 RB); A fragments of S from L2 (S is symmetric,
//                            so S[k][i] is read: coalesced in i), column projections over all rows through LDS, 128-byte
//                            row-segment stores.
// Deterministic (no atomics); HBM traffic of the forward: x once for the statistics plus (nb+1)/2 x for the Gram (nb = B/128).
//
// Round 4: the ADMM SITE above 128 rows on the same blocked form (the module layer composed it from four stand-alone passes before:
// 5x the fused site's cost per element): PAIR instantiations of the kernels above -
//   1'. sitel_stats_kernel : one sweep over x leaves the statistics of x AND of t = r (2 Phi(x) - 1) (stats [4][F]) and x_q;
//   2'. corrl_gram_kernel<true> : a tile's rows are staged twice - Xh, contracted with a NEGATED A operand, then Th (the transform
//       re-formed from the registers that hold x) - into ONE accumulator: the slabs hold D = corr(t,t) - corr(x,x) directly;
//   5'. corrl_bwd_kernel<RB, NW, true> : both operands in one launch: dx = -dcorr_x + (g + dcorr_t) dt/dx.
// Forward = x read (nb+1)/2 + 1 times and written once (was 4 reads + 2 writes + two Gram passes), backward one kernel (was five).
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "site_internal.h"

using namespace alignq;

namespace alignq_site {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kT = 256;          // threads per workgroup (4 waves)
constexpr int kTF = 64;          // features per Gram tile
constexpr int kLD = kTF + 1;     // LDS row stride (odd: the strided fragment reads are conflict-free)
constexpr int kBlk = 128;        // rows per output block
constexpr int kTFb = 32;         // features per backward tile
constexpr int kLDb = kTFb + 1;

__device__ __forceinline__ float4 ldq(const float* __restrict__ x, int64_t off, int col, int64_t F, bool row_ok, bool aligned) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!row_ok) return v;
  if (aligned) {
    if (col < F) v = *reinterpret_cast<const float4*>(x + off);
  } else {
    if (col + 0 < F) v.x = x[off + 0];
    if (col + 1 < F) v.y = x[off + 1];
    if (col + 2 < F) v.z = x[off + 2];
    if (col + 3 < F) v.w = x[off + 3];
  }
  return v;
}

// ---------------------------------------------------------------------------------------------- 1. column statistics
// grid = feature tiles of kTFs = 32 (two workgroups per CU at F = 16384; 64-feature tiles with four rows in flight had 16 KB per CU on
// request: 19 us for 2 x 16 MB at 256 rows); thread = (column quad c, row group rg of 32); rows rg, rg + 32, ...: eight in flight
constexpr int kTFs = 32;
constexpr int kRGs = kT / (kTFs / 4);      // 32 row groups
constexpr int kRows = 8;                   // rows in flight per thread

__global__ __launch_bounds__(kT) void corrl_stats_kernel(const float* __restrict__ x, int B, int64_t F, float eps,
                                                         float* __restrict__ stats, int aligned) {
  __shared__ double red[kRGs][kTFs][2];
  const int tid = threadIdx.x, c = tid & (kTFs / 4 - 1), rg = tid / (kTFs / 4);
  const int col = blockIdx.x * kTFs + 4 * c;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  for (int r0 = rg; r0 < B; r0 += kRGs * kRows) {
    float4 v[kRows];
#pragma unroll
    for (int u = 0; u < kRows; u++) {
      const int r = r0 + kRGs * u;
      v[u] = ldq(x, (int64_t)r * F + col, col, F, r < B, aligned);
    }
#pragma unroll
    for (int u = 0; u < kRows; u++) {
      const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
      for (int j = 0; j < 4; j++) { s[j] += (double)e[j]; q[j] += (double)e[j] * (double)e[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { red[rg][4 * c + j][0] = s[j]; red[rg][4 * c + j][1] = q[j]; }
  __syncthreads();
  if (tid < kTFs) {
    double a = 0, b = 0;
    for (int g = 0; g < kRGs; g++) { a += red[g][tid][0]; b += red[g][tid][1]; }     // fixed order
    const int64_t f = (int64_t)blockIdx.x * kTFs + tid;
    if (f < F) {
      const double mean = a / (double)B;
      double var = (b - a * mean) / (double)(B - 1);
      if (var < 0) var = 0;
      stats[f] = (float)mean;
      stats[F + f] = 1.0f / ((float)sqrt(var) + eps);
    }
  }
}

// ---------------------------------------------------------------------------------------------- 1'. pair statistics + x_q
// As corrl_stats_kernel, for the ADMM site: every element also goes through the quantiser (alignq_math.h: x_q bit-identical to
// alignq_act_quant_fwd) whose pre-round transform t is accumulated like x.  stats: [4][F] = mean_x, 1/(std_x+eps), mean_t, 1/(std_t+eps).
__global__ __launch_bounds__(kT) void sitel_stats_kernel(const float* __restrict__ x, int B, int64_t F, int k, float r, float eps,
                                                         float* __restrict__ xq, float* __restrict__ stats, int aligned) {
  __shared__ double red[kRGs][kTFs][4];
  __shared__ __attribute__((aligned(16))) float nerf_lds[ALIGNQ_NERF_LDS_FLOATS];
  nerf_tab_load(nerf_lds);
  __syncthreads();
  const NerfTab tab = nerf_tab(nerf_lds);
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  const int tid = threadIdx.x, c = tid & (kTFs / 4 - 1), rg = tid / (kTFs / 4);
  const int col = blockIdx.x * kTFs + 4 * c;
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0}, qt[4] = {0, 0, 0, 0};
  for (int r0 = rg; r0 < B; r0 += kRGs * kRows) {
    float4 v[kRows];
#pragma unroll
    for (int u = 0; u < kRows; u++) {
      const int rr = r0 + kRGs * u;
      v[u] = ldq(x, (int64_t)rr * F + col, col, F, rr < B, aligned);
    }
#pragma unroll
    for (int u = 0; u < kRows; u++) {
      const int rr = r0 + kRGs * u;
      const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float t, b;
        o[j] = act_quant1<0>(e[j], k, nlev, r, &t, &b, tab);
        if (rr < B) {
          s[j] += (double)e[j]; q[j] += (double)e[j] * (double)e[j];
          st[j] += (double)t; qt[j] += (double)t * (double)t;
        }
      }
      if (xq && rr < B) {
        const int64_t off = (int64_t)rr * F + col;
        if (aligned) {
          if (col < F) *reinterpret_cast<float4*>(xq + off) = make_float4(o[0], o[1], o[2], o[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++)
            if (col + j < F) xq[off + j] = o[j];
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    red[rg][4 * c + j][0] = s[j]; red[rg][4 * c + j][1] = q[j];
    red[rg][4 * c + j][2] = st[j]; red[rg][4 * c + j][3] = qt[j];
  }
  __syncthreads();
  if (tid < 2 * kTFs) {
    const int cc = tid & (kTFs - 1), which = tid / kTFs;          // which: 0 = x, 1 = t
    double a = 0, b = 0;
    for (int g = 0; g < kRGs; g++) { a += red[g][cc][2 * which]; b += red[g][cc][2 * which + 1]; }     // fixed order
    const int64_t f = (int64_t)blockIdx.x * kTFs + cc;
    if (f < F) {
      const double mean = a / (double)B;
      double var = (b - a * mean) / (double)(B - 1);
      if (var < 0) var = 0;
      stats[(2 * which) * F + f] = (float)mean;
      stats[(2 * which + 1) * F + f] = 1.0f / ((float)sqrt(var) + eps);
    }
  }
}

// ---------------------------------------------------------------------------------------------- 2. blocked Gram
__device__ __forceinline__ void pair_ij(int p, int nb, int& I, int& J) {
  int t = p;
  I = 0;
  while (t >= nb - I) { t -= nb - I; I++; }
  J = I + t;
}

// grid = (ksplit, npairs).  slabs: [npairs][ksplit][128][128].  PAIR: see the file header (stats is then [4][F], r the act_range)
template <bool PAIR>
__global__ __launch_bounds__(kT) void corrl_gram_kernel(const float* __restrict__ x, const float* __restrict__ stats, int B,
                                                        int64_t F, float* __restrict__ slabs, int n_tiles, int nb,
                                                        int aligned, float r) {
  __shared__ float As[kBlk * kLD];
  __shared__ float Bs[kBlk * kLD];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = tid & 15, rg = tid >> 4;
  const int h = lane >> 5, l31 = lane & 31;
  int I, J;
  pair_ij(blockIdx.y, nb, I, J);
  const bool diag = I == J;
  const int wr = w >> 1, wc = w & 1;          // the wave's 64 x 64 quadrant
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
  const float* Bsel = diag ? As : Bs;

  const float rjac = r * ALIGNQ_TWO_OVER_SQRT_2PI;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int col = tile * kTF + 4 * c;
    float m[4], rho[4], mt[4], rhot[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const bool ok = col + e < F;
      m[e] = ok ? stats[col + e] : 0.0f;
      rho[e] = ok ? stats[F + col + e] : 0.0f;
      mt[e] = (PAIR && ok) ? stats[2 * F + col + e] : 0.0f;
      rhot[e] = (PAIR && ok) ? stats[3 * F + col + e] : 0.0f;
    }
    // the rows of both blocks stay in registers: staged as Xh first and (PAIR) as Th afterwards
    float4 v[2][8];
#pragma unroll
    for (int half = 0; half < 2; half++) {
      if (half == 1 && diag) break;
      const int blk = half == 0 ? I : J;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int grow = blk * kBlk + rg + 16 * j;
        v[half][j] = ldq(x, (int64_t)grow * F + col, col, F, grow < B, aligned);
      }
    }
    const int rowA = wr * 64 + l31, rowB = wc * 64 + l31;
#pragma unroll
    for (int op = 0; op < (PAIR ? 2 : 1); op++) {
#pragma unroll
      for (int half = 0; half < 2; half++) {
        if (half == 1 && diag) break;
        const int blk = half == 0 ? I : J;
        float* dst = half == 0 ? As : Bs;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int row = rg + 16 * j;
          const bool ok = blk * kBlk + row < B;
          const float e4[4] = {v[half][j].x, v[half][j].y, v[half][j].z, v[half][j].w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float val;
            if (op == 0) val = (e4[e] - m[e]) * rho[e];
            else {
              float t, jac;
              act_transform_rcp(e4[e], r, rjac, &t, &jac);
              val = (t - mt[e]) * rhot[e];
            }
            dst[row * kLD + 4 * c + e] = (ok && col + e < F) ? val : 0.0f;
          }
        }
      }
      __syncthreads();
      const bool neg = PAIR && op == 0;            // D = corr(t,t) - corr(x,x): the x operand enters with a minus sign
#pragma unroll 4
      for (int k0 = 0; k0 < kTF; k0 += 2) {
        float a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const float av = As[(rowA + 32 * i) * kLD + k0 + h];
          a[i] = neg ? -av : av;
          b[i] = Bsel[(rowB + 32 * i) * kLD + k0 + h];
        }
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
  }
  float* slab = slabs + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (kBlk * kBlk);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int row = wr * 64 + 32 * i + (e & 3) + 8 * (e >> 2) + 4 * h;
        slab[row * kBlk + wc * 64 + 32 * j + l31] = acc[i][j][e];
      }
}

// ---------------------------------------------------------------------------------------------- 3. slab reduction
// grid = (64, npairs): thread = one element of the pair's 128 x 128 block
__global__ __launch_bounds__(kT) void corrl_reduce_kernel(const float* __restrict__ slabs, int ksplit, int nb, int B, float scale,
                                                          float* __restrict__ G) {
  int I, J;
  pair_ij(blockIdx.y, nb, I, J);
  const int e = blockIdx.x * kT + threadIdx.x;
  const float* p = slabs + (int64_t)blockIdx.y * ksplit * (kBlk * kBlk) + e;
  float s = 0.f;
  int k = 0;
  for (; k + 16 <= ksplit; k += 16) {        // sixteen slabs in flight per thread (192 workgroups per launch at two row blocks); the
    float a[16];                              // additions stay in slab order
#pragma unroll
    for (int u = 0; u < 16; u++) a[u] = p[(int64_t)(k + u) * (kBlk * kBlk)];
#pragma unroll
    for (int u = 0; u < 16; u++) s += a[u];
  }
  for (; k + 4 <= ksplit; k += 4) {
    const float a0 = p[(int64_t)(k + 0) * (kBlk * kBlk)], a1 = p[(int64_t)(k + 1) * (kBlk * kBlk)];
    const float a2 = p[(int64_t)(k + 2) * (kBlk * kBlk)], a3 = p[(int64_t)(k + 3) * (kBlk * kBlk)];
    s += a0; s += a1; s += a2; s += a3;
  }
  for (; k < ksplit; k++) s += p[(int64_t)k * (kBlk * kBlk)];
  const int i = I * kBlk + (e >> 7), j = J * kBlk + (e & 127);
  if (i < B && j < B) {
    G[(int64_t)i * B + j] = s * scale;
    if (I != J) G[(int64_t)j * B + i] = s * scale;
  }
}

// ---------------------------------------------------------------------------------------------- 4. S = (dG + dG^T) / F
// S is stored PADDED: [BP = 32 ceil(B/32)][WP = the 32 NW RB columns the backward's waves cover], zero outside [B][B], so the backward's
// K loop reads it without a condition (corrl_s_width)
__global__ __launch_bounds__(kT) void corrl_sym_kernel(const float* __restrict__ dG, int B, float scale, float* __restrict__ S,
                                                       const float* __restrict__ dscale, int BP, int WP) {
  if (dscale) scale *= *dscale;                 // optional DEVICE scalar (the upstream gradient of a scalar loss)
  const int n = BP * WP;
  for (int e = (int)blockIdx.x * kT + threadIdx.x; e < n; e += (int)gridDim.x * kT) {
    const int i = e / WP, j = e - i * WP;
    S[e] = (i < B && j < B) ? (dG[(int64_t)i * B + j] + dG[(int64_t)j * B + i]) * scale : 0.0f;
  }
}

// ---------------------------------------------------------------------------------------------- 5. backward
// dynamic LDS: Xh [nblk*32][33] floats + red [NW][2][32] + tot [2][32].  NW waves share the row blocks (wave w: blocks w + NW q,
// q < RB); B <= 512: 4 waves x RB <= 4; above: 8 waves x 4 (8 accumulators per wave spilled 54 registers at 256 + 256)
// PAIR (round 4, the ADMM site above 128 rows): both operands in one launch.  Pass 0 is the corr(x,x) part (it enters D with a minus
// sign): dx = -dcorr_x is stored; pass 1 stages Th (the transform re-formed from x, read again from L2), contracts, projects with the
// t statistics (stats [4][F]) and finishes dx = -dcorr_x + (g + dcorr_t) * dt/dx, g = the upstream gradient of x_q (or nullptr).
// assemble: rows written per group of requests (x / g / -dcorr_x in flight together); measured 4 / 8 / 16 at 256, 512, 1024 rows x 16384
#ifndef CORRL_AG
#define CORRL_AG 16
#endif
template <int RB, int NW = 4, bool PAIR = false>
__global__ __launch_bounds__(64 * NW) void corrl_bwd_kernel(const float* __restrict__ S, const float* __restrict__ x,
                                                       const float* __restrict__ stats, int B, int64_t F, float eps,
                                                       float* __restrict__ dx, int n_tiles, int aligned,
                                                       const float* __restrict__ gup = nullptr, float r = 1.0f) {
  extern __shared__ __attribute__((aligned(16))) float lds_dyn[];
  const int nblk = (B + 31) >> 5, BP = nblk * 32;
  float* Xs = lds_dyn;
  float* red = lds_dyn + BP * kLDb;            // [NW waves][2][32]
  float* tot = red + NW * 2 * 32;              // [2][32]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int c = tid & 7, rg = tid >> 3;        // load mapping: column quad, 8 NW row groups
  constexpr int RG = 8 * NW;
  const float invB = 1.0f / (float)B, invBm1 = 1.0f / (float)(B - 1);
  const float rjac = r * ALIGNQ_TWO_OVER_SQRT_2PI;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int col0 = tile * kTFb, col = col0 + 4 * c;
#pragma unroll
    for (int op = 0; op < (PAIR ? 2 : 1); op++) {
      const int so = 2 * op;                     // statistics rows of this operand: (mean, rho) at [so], [so + 1]
      float m[4], rho[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const bool ok = col + e < F;
        m[e] = ok ? stats[so * F + col + e] : 0.0f;
        rho[e] = ok ? stats[(so + 1) * F + col + e] : 0.0f;
      }
      for (int r0 = rg; r0 < BP; r0 += RG * 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int rr = r0 + RG * u;
          v[u] = ldq(x, (int64_t)rr * F + col, col, F, rr < B, aligned);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int rr = r0 + RG * u;
          if (rr < BP) {
            const float e4[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
              float val = e4[e];
              if (PAIR && op == 1) {
                float t, jac;
                act_transform_rcp(e4[e], r, rjac, &t, &jac);
                val = t;
              }
              Xs[rr * kLDb + 4 * c + e] = (rr < B && col + e < F) ? (val - m[e]) * rho[e] : 0.0f;
            }
          }
        }
      }
      __syncthreads();
      // ---- dVh block rows of this wave: acc[q] = S[rows of block w + NW q][:] Vh ------------------------------------
      f32x16 acc[RB];
#pragma unroll
      for (int q = 0; q < RB; q++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[q][e] = 0.0f;
      // software pipeline, two register sets: the S fragments and Vh values of the NEXT KD K steps are requested before this
      // chunk's matrix instructions are issued (always: the last chunk re-reads the one before it), so an L2 round trip hides
      // under 64 KD RB cycles of the matrix pipe.  No branch around the matrix instruction (a row block past the last one
      // reads zero padding): under `if (block < nblk)` the accumulators travelled VGPR -> AGPR -> VGPR around EVERY instruction
      {
        constexpr int KD = RB <= 2 ? 2 : 1;      // K steps per request: four S loads in flight per wave and register set either way
        static_assert(32 % (4 * KD) == 0, "two chunks of K steps must divide the 32-row blocks");
        float a0[KD][RB], b0[KD], a1[KD][RB], b1[KD];
        const float* Sw = S + (unsigned)(w * 32 + l31);
        const float* Xw = Xs + h * kLDb + l31;
        auto request = [&](int kb, float (&a)[KD][RB], float (&b)[KD]) {
#pragma unroll
          for (int s2 = 0; s2 < KD; s2++) {
            b[s2] = Xw[(kb + 2 * s2) * kLDb];
#pragma unroll
            for (int q = 0; q < RB; q++)
              a[s2][q] = Sw[(unsigned)(kb + 2 * s2 + h) * (unsigned)(32 * NW * RB) + (unsigned)(NW * q * 32)];
          }
        };
        auto contract = [&](const float (&a)[KD][RB], const float (&b)[KD]) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int s2 = 0; s2 < KD; s2++)
#pragma unroll
            for (int q = 0; q < RB; q++) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s2][q], b[s2], acc[q], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        };
        request(0, a0, b0);
        for (int k0 = 0; k0 < BP; k0 += 4 * KD) {
          request(k0 + 2 * KD, a1, b1);
          contract(a0, b0);
          request(min(k0 + 4 * KD, BP - 2 * KD), a0, b0);
          contract(a1, b1);
        }
      }
      // ---- column projections over ALL rows: sum_b dVh, sum_b dVh * vh ----------------------------------------------
      float sd = 0.f, sdx = 0.f;
#pragma unroll
      for (int q = 0; q < RB; q++) {
        if (w + NW * q < nblk) {
#pragma unroll
          for (int e = 0; e < 16; e++) {
            const int row = (w + NW * q) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float d = acc[q][e];            // rows >= B: S rows are zero there, so d == 0
            sd += d;
            sdx += d * Xs[row * kLDb + l31];
          }
        }
      }
      sd += __shfl_xor(sd, 32, 64);
      sdx += __shfl_xor(sdx, 32, 64);
      if (h == 0) { red[(w * 2 + 0) * 32 + l31] = sd; red[(w * 2 + 1) * 32 + l31] = sdx; }
      __syncthreads();
      if (tid < 64) {
        const int which = tid >> 5, cc = tid & 31;
        float t4 = (red[(0 * 2 + which) * 32 + cc] + red[(1 * 2 + which) * 32 + cc]) +
                   (red[(2 * 2 + which) * 32 + cc] + red[(3 * 2 + which) * 32 + cc]);
        if (NW == 8)
          t4 += (red[(4 * 2 + which) * 32 + cc] + red[(5 * 2 + which) * 32 + cc]) +
                (red[(6 * 2 + which) * 32 + cc] + red[(7 * 2 + which) * 32 + cc]);
        tot[which * 32 + cc] = t4;
      }
      __syncthreads();
      // ---- assemble and store (lanes l31 -> 32 consecutive features of one row: 128-byte segments) -------------------
      {
        const int f = col0 + l31;
        const bool fok = f < F;
        const float rr = fok ? stats[(so + 1) * F + f] : 0.0f;
        const float sdev = fok ? (1.0f / rr - eps) : 0.0f;
        const float mean_d = tot[l31] * invB;
        // torch's std backward gives no gradient through a zero std (masked_fill), DESIGN.md §7
        const float kdot = (sdev > 0.0f) ? tot[32 + l31] * invBm1 / sdev : 0.0f;
#pragma unroll
        for (int q = 0; q < RB; q++) {
          if (w + NW * q < nblk) {
#pragma unroll
            for (int grp = 0; grp < 16 / CORRL_AG; grp++) {
              // CORRL_AG rows at a time (their x / g / -dcorr_x loads in flight together, then the stores)
              __builtin_amdgcn_sched_barrier(0);
              int rowb = (w + NW * q) * 32 + 4 * h;
              // opaque: row * F is the same for every tile, and the optimiser otherwise keeps all 16 RB (x 2 for the pair) 64-bit
              // row offsets in registers across the tile loop (256 + 246 of them at RB = 4; 134 spilled with 8 waves)
              asm volatile("" : "+v"(rowb));
              float xv[CORRL_AG], gv[CORRL_AG], cx[CORRL_AG];
              if (PAIR && op == 1) {
#pragma unroll
                for (int j = 0; j < CORRL_AG; j++) {
                  const int e = CORRL_AG * grp + j, row = rowb + (e & 3) + 8 * (e >> 2);
                  const int64_t o = (int64_t)min(row, B - 1) * F + (fok ? f : 0);
                  xv[j] = x[o];
                  gv[j] = gup ? gup[o] : 0.0f;
                  cx[j] = dx[o];                                  // pass 0 left -dcorr_x here
                }
              }
#pragma unroll
              for (int j = 0; j < CORRL_AG; j++) {
                const int e = CORRL_AG * grp + j, row = rowb + (e & 3) + 8 * (e >> 2);
                const float cv = (acc[q][e] - mean_d) * rr - Xs[row * kLDb + l31] * kdot;
                if (row < B && fok) {
                  float* o = dx + (int64_t)row * F + f;
                  if (!PAIR) *o = cv;
                  else if (op == 0) *o = -cv;                    // corr(x,x) enters D with a minus sign
                  else *o = cx[j] + (gv[j] + cv) * (rjac * __builtin_amdgcn_exp2f(xv[j] * xv[j] * -0.72134752044448170368f));
                }
              }
            }
          }
        }
      }
      __syncthreads();      // Xs is overwritten by the next operand / tile
    }
  }
}

inline int n_pairs(int nb) { return nb * (nb + 1) / 2; }

}  // namespace

// columns of the padded S image = 32 x the row blocks the backward's workgroup covers (launch_corrl_bwd's choice of <RB, NW>)
int corrl_s_width(int B) {
  const int nblk = (B + 31) / 32, rb = (nblk + 3) / 4;
  return rb <= 2 ? 32 * 4 * 2 : (rb <= 4 ? 32 * 4 * 4 : 32 * 8 * 4);
}
size_t corrl_s_bytes(int B) { return (size_t)((B + 31) / 32 * 32) * corrl_s_width(B) * sizeof(float); }

// K splits of the Gram launch.  512 workgroups are resident at once (two per CU: 66 KB of LDS each), a workgroup costs its
// ceil(n_tiles / ks) tiles, the launch ceil(np ks / 512) rounds of them; every slab (64 KB written, read once by the reduction)
// costs about 0.0064 of a tile round.  Rounds 3 took ceil(1024 / np) splits: 1044 workgroups of 9 tiles at 1024 x 16384, i.e. three
// rounds of 9 where 14 splits make ONE round of 19.
int corrl_ksplit(int B, int64_t F) {
  const int nb = (B + kBlk - 1) / kBlk, np = n_pairs(nb);
  const int64_t n_tiles = (F + kTF - 1) / kTF;
  const int kmax = (int)(n_tiles < 512 ? n_tiles : 512);
  int best = 1;
  double best_cost = 1e30;
  for (int ks = 1; ks <= kmax; ks++) {
    const int64_t rounds = ((int64_t)np * ks + 511) / 512, per_wg = (n_tiles + ks - 1) / ks;
    const double cost = (double)(rounds * per_wg) + 0.0064 * np * ks;
    if (cost < best_cost) { best_cost = cost; best = ks; }
  }
  return best;
}

size_t corrl_ws_bytes(int B, int64_t F) {
  const int nb = (B + kBlk - 1) / kBlk;
  return (size_t)n_pairs(nb) * corrl_ksplit(B, F) * kBlk * kBlk * sizeof(float);
}

int launch_corrl_fwd(const float* x, int B, int64_t F, float eps, float* G, float* stats, float* ws, hipStream_t st) {
  const int aligned = ((F & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
  const int nb = (B + kBlk - 1) / kBlk, np = n_pairs(nb), ks = corrl_ksplit(B, F);
  const int n_tiles = (int)((F + kTF - 1) / kTF);
  hipLaunchKernelGGL(corrl_stats_kernel, dim3((unsigned)((F + kTFs - 1) / kTFs)), dim3(kT), 0, st, x, B, F, eps, stats, aligned);
  hipLaunchKernelGGL((corrl_gram_kernel<false>), dim3(ks, np), dim3(kT), 0, st, x, (const float*)stats, B, F, ws, n_tiles, nb, aligned, 1.0f);
  hipLaunchKernelGGL(corrl_reduce_kernel, dim3(kBlk * kBlk / kT, np), dim3(kT), 0, st, (const float*)ws, ks, nb, B,
                     1.0f / (float)F, G);
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

// The ADMM site above 128 rows (round 4): x_q, stats [4][F] and D = corr(t,t) - corr(x,x) in three launches
int launch_sitel_fwd(const float* x, int B, int64_t F, int k, float r, float eps, float* xq, float* D, float* stats, float* ws,
                     hipStream_t st) {
  const int aligned = ((F & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(xq) & 15) == 0) ? 1 : 0;
  const int nb = (B + kBlk - 1) / kBlk, np = n_pairs(nb), ks = corrl_ksplit(B, F);
  const int n_tiles = (int)((F + kTF - 1) / kTF);
  hipLaunchKernelGGL(sitel_stats_kernel, dim3((unsigned)((F + kTFs - 1) / kTFs)), dim3(kT), 0, st, x, B, F, k, r, eps, xq, stats, aligned);
  hipLaunchKernelGGL((corrl_gram_kernel<true>), dim3(ks, np), dim3(kT), 0, st, x, (const float*)stats, B, F, ws, n_tiles, nb, aligned, r);
  hipLaunchKernelGGL(corrl_reduce_kernel, dim3(kBlk * kBlk / kT, np), dim3(kT), 0, st, (const float*)ws, ks, nb, B,
                     1.0f / (float)F, D);
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

int launch_corrl_bwd(const float* dG, const float* x, const float* stats, int B, int64_t F, float eps, float* dx, float* S,
                     hipStream_t st, bool pair, const float* gup, float r, const float* dG_scale) {
  const int aligned = ((F & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) ? 1 : 0;
  const int nblk = (B + 31) / 32, rb = (nblk + 3) / 4;
  const int BP = nblk * 32, WP = corrl_s_width(B);
  int gs = (BP * WP + kT - 1) / kT;
  if (gs > 2048) gs = 2048;
  hipLaunchKernelGGL(corrl_sym_kernel, dim3(gs), dim3(kT), 0, st, dG, B, 1.0f / (float)F, S, dG_scale, BP, WP);
  const int n_tiles = (int)((F + kTFb - 1) / kTFb);
  const size_t lds = ((size_t)nblk * 32 * kLDb + 8 * 2 * 32 + 2 * 32) * sizeof(float);
  int grid = n_tiles < 2048 ? n_tiles : 2048;
#define CORRL_BWD(RB, NWV)                                                                                                     \
  do {                                                                                                                         \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&corrl_bwd_kernel<RB, NWV>),                                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess)                     \
        return ALIGNQ_EINVAL;                                                                                                  \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL((corrl_bwd_kernel<RB, NWV>), dim3(grid), dim3(64 * NWV), lds, st, (const float*)S, x, stats, B, F, eps, \
                       dx, n_tiles, aligned, nullptr, 1.0f);                                                                   \
  } while (0)
#define SITEL_BWD(RB, NWV)                                                                                                     \
  do {                                                                                                                         \
    static bool attr_set = false;                                                                                              \
    if (!attr_set) {                                                                                                           \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&corrl_bwd_kernel<RB, NWV, true>),                                 \
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess)                     \
        return ALIGNQ_EINVAL;                                                                                                  \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL((corrl_bwd_kernel<RB, NWV, true>), dim3(grid), dim3(64 * NWV), lds, st, (const float*)S, x, stats, B, F, \
                       eps, dx, n_tiles, aligned, gup, r);                                                                     \
  } while (0)
  if (pair) {
    if (rb <= 2) SITEL_BWD(2, 4);
    else if (rb <= 4) SITEL_BWD(4, 4);
    else SITEL_BWD(4, 8);
    return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
  }
  if (rb <= 2) CORRL_BWD(2, 4);
  else if (rb <= 4) CORRL_BWD(4, 4);
  else CORRL_BWD(4, 8);                   // 512 < B <= 1024: eight waves, four row blocks each
#undef CORRL_BWD
#undef SITEL_BWD
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

}  // namespace alignq_site
