// site_kernels.hip — fused ADMM-site kernels for gfx950: activation quantise + the pair of sample-
// correlation (Gram) matrices, forward and backward, on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// Reference semantics: activation_quantize_fn.forward (ADMM tree model/quantization.py:102-132), corr
// (:134-137; Office tree :158-161), with x viewed as [B, F], B <= 128.
//
// Data layout / tiling:
//   * x is row-major [B,F]; a workgroup (256 threads = 4 waves) owns column tiles of TF=64 features:
//     each row contributes 256 contiguous bytes -> 16 lanes x float4, fully coalesced; x is read ONCE.
//   * per-feature batch statistics (mean, unbiased std) of x and of t = r(2Phi(x)-1) are reduced over the
//     rows through LDS; standardised tiles Xh, Th are staged in LDS [BP][65] (odd stride => the strided
//     MFMA A/B fragment reads `row = lane&31` are bank-conflict free).
//   * D = Th Th^T - Xh Xh^T is accumulated in ONE set of MFMA accumulators per 32x32 output tile
//     (the x contribution enters with a negated A operand), K = the tile's 64 features.
//   * cross-workgroup reduction is deterministic: each workgroup writes its partial [BP,BP] slab, a second
//     kernel sums the slabs (16 slab groups x 64 elements per 1024-thread block) and scales by 1/F.
//   * backward is tile-local (no cross-workgroup traffic): dXh = S Xh with S = (dD+dD^T)*scale/F held as
//     MFMA A fragments in registers, Xh/Th from LDS as B fragments; the per-feature projections of the
//     standardisation backward are wave-shuffle + LDS reductions over the batch dimension.
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "site_internal.h"

using namespace alignq;
using namespace alignq_site;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int TF = 64;        // features per tile
constexpr int LD = TF + 1;    // LDS row stride in floats
constexpr int kThreads = 256;

__device__ __forceinline__ float4 load4(const float* __restrict__ x, int64_t off, int col, int64_t F, bool row_ok,
                                        bool aligned) {
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

__device__ __forceinline__ void store4(float* __restrict__ y, int64_t off, int col, int64_t F, bool row_ok, bool aligned,
                                       float4 v) {
  if (!row_ok) return;
  if (aligned) {
    if (col < F) *reinterpret_cast<float4*>(y + off) = v;
  } else {
    if (col + 0 < F) y[off + 0] = v.x;
    if (col + 1 < F) y[off + 1] = v.y;
    if (col + 2 < F) y[off + 2] = v.z;
    if (col + 3 < F) y[off + 3] = v.w;
  }
}

// ================================================================================================
// forward: per column tile -> quantise, statistics, standardise, MFMA Gram difference
//   NB   : number of 32-row blocks (BP = 32*NB >= B)
//   PAIR : true  -> D = corr(t,t) - corr(x,x) and x_q;  false -> G = corr(x,x) only
template <int NB, bool PAIR>
__global__ __launch_bounds__(kThreads) void site_fwd_kernel(const float* __restrict__ x, int B, int64_t F, int k,
                                                            float r, float eps, float* __restrict__ xq,
                                                            float* __restrict__ slabs, float* __restrict__ stats,
                                                            int n_tiles, int aligned, unsigned* __restrict__ counter) {
  constexpr int BP = 32 * NB;
  constexpr int RJ = BP / 16;                 // rows per thread in the load mapping
  constexpr int NOP = PAIR ? 2 : 1;           // operands staged in LDS (x, t)
  constexpr int RI = (NB == 4) ? 2 : 1;       // 32x32 output tiles per wave per dimension
  constexpr int KS = (NB == 1) ? 4 : 1;       // K-split across waves when there is a single output tile
  constexpr int LDS_FLOATS = (2 * BP * LD > 4096 ? 2 * BP * LD : 4096) + 4 * TF;
  __shared__ float lds[LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float nerf_lds[PAIR ? ALIGNQ_NERF_LDS_FLOATS : 4];
  if (PAIR) {
    nerf_tab_load(nerf_lds);
    __syncthreads();
  }
  const NerfTab tab = nerf_tab(nerf_lds);
  float* Xs = lds;                            // [BP][LD]
  float* Ts = lds + BP * LD;                  // [BP][LD]
  float* red = Ts;                            // [2][16][TF]   (aliased: used before Ts is written)
  float* colv = lds + (2 * BP * LD > 4096 ? 2 * BP * LD : 4096);  // mean_x, rho_x, mean_t, rho_t : [4][TF]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = tid & 15, rg = tid >> 4;
  const int h = lane >> 5, l31 = lane & 31;
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  const float invBm1 = 1.0f / (float)(B - 1);
  if (blockIdx.x == 0 && tid == 0 && counter) *counter = 0u;   // arrival counter of the reduce epilogue

  // wave -> output tiles
  const int wr = (NB == 1) ? 0 : (w >> 1), wc = (NB == 1) ? 0 : (w & 1);
  f32x16 acc[RI][RI];
#pragma unroll
  for (int i = 0; i < RI; i++)
#pragma unroll
    for (int j = 0; j < RI; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int col0 = tile * TF;
    const int col = col0 + 4 * c;
    float4 xv[RJ], tv[RJ];
    // ---- load + transform + quantise --------------------------------------------------------
#pragma unroll
    for (int j = 0; j < RJ; j++) {
      const int row = rg + 16 * j;
      const bool ok = row < B;
      const int64_t off = (int64_t)row * F + col;
      xv[j] = load4(x, off, col, F, ok, aligned);
      if (PAIR) {
        float4 q;
        float b;
        q.x = act_quant1<0>(xv[j].x, k, nlev, r, &tv[j].x, &b, tab);
        q.y = act_quant1<0>(xv[j].y, k, nlev, r, &tv[j].y, &b, tab);
        q.z = act_quant1<0>(xv[j].z, k, nlev, r, &tv[j].z, &b, tab);
        q.w = act_quant1<0>(xv[j].w, k, nlev, r, &tv[j].w, &b, tab);
        if (xq) store4(xq, off, col, F, ok, aligned, q);
      }
    }
    // ---- column means -----------------------------------------------------------------------
    {
      float4 sx = make_float4(0, 0, 0, 0), st = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        if (rg + 16 * j < B) {
          sx.x += xv[j].x; sx.y += xv[j].y; sx.z += xv[j].z; sx.w += xv[j].w;
          if (PAIR) { st.x += tv[j].x; st.y += tv[j].y; st.z += tv[j].z; st.w += tv[j].w; }
        }
      }
      float* p0 = red + rg * TF + 4 * c;
      p0[0] = sx.x; p0[1] = sx.y; p0[2] = sx.z; p0[3] = sx.w;
      if (PAIR) {
        float* p1 = red + 16 * TF + rg * TF + 4 * c;
        p1[0] = st.x; p1[1] = st.y; p1[2] = st.z; p1[3] = st.w;
      }
    }
    __syncthreads();
    if (tid < NOP * TF) {
      const int op = tid >> 6, cc = tid & 63;
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; g++) s += red[op * 16 * TF + g * TF + cc];
      colv[(2 * op) * TF + cc] = s / (float)B;      // true division, like torch.mean (a constant column: exactly its value)
    }
    __syncthreads();
    // ---- column variances (two-pass) ----------------------------------------------------------
    {
      const float4 mx = make_float4(colv[4 * c], colv[4 * c + 1], colv[4 * c + 2], colv[4 * c + 3]);
      float4 mt = make_float4(0, 0, 0, 0);
      if (PAIR) mt = make_float4(colv[2 * TF + 4 * c], colv[2 * TF + 4 * c + 1], colv[2 * TF + 4 * c + 2],
                                 colv[2 * TF + 4 * c + 3]);
      float4 sx = make_float4(0, 0, 0, 0), st = make_float4(0, 0, 0, 0);
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        if (rg + 16 * j < B) {
          float d;
          d = xv[j].x - mx.x; sx.x += d * d;
          d = xv[j].y - mx.y; sx.y += d * d;
          d = xv[j].z - mx.z; sx.z += d * d;
          d = xv[j].w - mx.w; sx.w += d * d;
          if (PAIR) {
            d = tv[j].x - mt.x; st.x += d * d;
            d = tv[j].y - mt.y; st.y += d * d;
            d = tv[j].z - mt.z; st.z += d * d;
            d = tv[j].w - mt.w; st.w += d * d;
          }
        }
      }
      float* p0 = red + rg * TF + 4 * c;
      p0[0] = sx.x; p0[1] = sx.y; p0[2] = sx.z; p0[3] = sx.w;
      if (PAIR) {
        float* p1 = red + 16 * TF + rg * TF + 4 * c;
        p1[0] = st.x; p1[1] = st.y; p1[2] = st.z; p1[3] = st.w;
      }
    }
    __syncthreads();
    if (tid < NOP * TF) {
      const int op = tid >> 6, cc = tid & 63;
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 16; g++) s += red[op * 16 * TF + g * TF + cc];
      const float sd = sqrtf(s * invBm1);
      const float rho = 1.0f / (sd + eps);
      colv[(2 * op + 1) * TF + cc] = rho;
      if (stats && col0 + cc < F) {
        stats[(int64_t)(2 * op) * F + col0 + cc] = colv[(2 * op) * TF + cc];
        stats[(int64_t)(2 * op + 1) * F + col0 + cc] = rho;
      }
    }
    __syncthreads();
    // ---- standardise into LDS -------------------------------------------------------------------
    {
      float mx[4], rx[4], mt[4], rt[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        mx[e] = colv[4 * c + e];
        rx[e] = colv[TF + 4 * c + e];
        if (PAIR) { mt[e] = colv[2 * TF + 4 * c + e]; rt[e] = colv[3 * TF + 4 * c + e]; }
      }
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        const int row = rg + 16 * j;
        const bool ok = row < B;
        const float xe[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
        const float te[4] = {tv[j].x, tv[j].y, tv[j].z, tv[j].w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool okc = ok && (col + e < F);
          Xs[row * LD + 4 * c + e] = okc ? (xe[e] - mx[e]) * rx[e] : 0.0f;
          if (PAIR) Ts[row * LD + 4 * c + e] = okc ? (te[e] - mt[e]) * rt[e] : 0.0f;
        }
      }
    }
    __syncthreads();
    // ---- MFMA: acc += Th Th^T - Xh Xh^T over this tile's 64 features ------------------------------
    {
      const int kbeg = (KS == 1) ? 0 : w * (TF / KS);
      const int kend = (KS == 1) ? TF : kbeg + TF / KS;
      const int rowA = (NB == 4 ? wr * 64 : wr * 32) + l31;
      const int rowB = (NB == 4 ? wc * 64 : wc * 32) + l31;
#pragma unroll 4
      for (int k0 = kbeg; k0 < kend; k0 += 2) {
        float ax[RI], bx[RI], at[RI], bt[RI];
#pragma unroll
        for (int i = 0; i < RI; i++) {
          ax[i] = Xs[(rowA + 32 * i) * LD + k0 + h];
          bx[i] = Xs[(rowB + 32 * i) * LD + k0 + h];
          if (PAIR) {
            at[i] = Ts[(rowA + 32 * i) * LD + k0 + h];
            bt[i] = Ts[(rowB + 32 * i) * LD + k0 + h];
          }
        }
#pragma unroll
        for (int i = 0; i < RI; i++)
#pragma unroll
          for (int j = 0; j < RI; j++) {
            if (PAIR) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(at[i], bt[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(-ax[i], bx[j], acc[i][j], 0, 0, 0);
            } else {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax[i], bx[j], acc[i][j], 0, 0, 0);
            }
          }
      }
    }
    __syncthreads();   // LDS tiles are overwritten by the next iteration
  }

  // ---- write this workgroup's partial slab [BP][BP] ----------------------------------------------
  float* slab = slabs + (int64_t)blockIdx.x * BP * BP;
  if (KS == 1) {
#pragma unroll
    for (int i = 0; i < RI; i++)
#pragma unroll
      for (int j = 0; j < RI; j++) {
        const int I = (NB == 4 ? wr * 2 : wr) + i, J = (NB == 4 ? wc * 2 : wc) + j;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = I * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          slab[row * BP + J * 32 + l31] = acc[i][j][e];
        }
      }
  } else {
    // single 32x32 tile, four K-slices (one per wave): combine through LDS
    float* buf = lds;  // [4][32][32]
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
      buf[w * 1024 + row * 32 + l31] = acc[0][0][e];
    }
    __syncthreads();
    for (int e = tid; e < 1024; e += kThreads) slab[e] = buf[e] + buf[1024 + e] + buf[2048 + e] + buf[3072 + e];
  }
}

// ================================================================================================
// backward.  dx = g*jac + jac * d(corr(t,t))/dt [with +dD] + d(corr(x,x))/dx [with -dD]      (PAIR)
//            dx = d(corr(x,x))/dx [with +dG]                                                   (!PAIR)
// Wave w owns output row block I (32 batch rows) and PX column blocks of 32 features; for every owned
// column block it holds the accumulator tile of BOTH operands, so the final assembly is wave-local.
//   NB=4: I = w,    column blocks {0,1}
//   NB=2: I = w>>1, column block  w&1
//   NB=1: I = 0,    column block  w (waves 0,1; waves 2,3 only help with loads and LDS staging)
template <int NB, bool PAIR>
__global__ __launch_bounds__(kThreads) void site_bwd_kernel(const float* __restrict__ g, const float* __restrict__ S,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ stats, int B, int64_t F, float r,
                                                            float eps, float* __restrict__ dx, int n_tiles,
                                                            int aligned) {
  constexpr int BP = 32 * NB;
  constexpr int RJ = BP / 16;
  constexpr int NOP = PAIR ? 2 : 1;
  constexpr int PX = (NB == 4) ? 2 : 1;
  constexpr int KSTEPS = BP / 2;
  __shared__ float lds[2 * BP * LD + 4 * TF + NB * NOP * 2 * TF];
  __shared__ __attribute__((aligned(16))) float nerf_lds[PAIR ? ALIGNQ_NERF_LDS_FLOATS : 4];
  if (PAIR) nerf_tab_load(nerf_lds);           // published by the tile loop's first barrier
  const NerfTab tab = nerf_tab(nerf_lds);
  float* Xs = lds;
  float* Ts = lds + BP * LD;
  float* colv = lds + 2 * BP * LD;             // mean_x, rho_x, mean_t, rho_t  [4][TF]
  float* red = colv + 4 * TF;                  // [NB row blocks][NOP][2][TF]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = tid & 15, rg = tid >> 4;
  const int h = lane >> 5, l31 = lane & 31;

  const int I = (NB == 4) ? w : (NB == 2 ? (w >> 1) : 0);
  const bool wave_active = (NB != 1) || (w < 2);

  // S fragments (MFMA A operand): S = sym(dD)*scale/F prepared by site_prep_kernel (symmetric => read as S[k][i],
  // coalesced in i):  A[i][k], i = I*32 + l31, k = 2*s + h
  float sfrag[KSTEPS];
  {
    const int i = I * 32 + l31;
#pragma unroll
    for (int s = 0; s < KSTEPS; s++) {
      const int kk = 2 * s + h;
      sfrag[s] = (i < B && kk < B) ? S[kk * B + i] : 0.0f;
    }
  }
  const float invB = 1.0f / (float)B, invBm1 = 1.0f / (float)(B - 1);

  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int col0 = tile * TF;
    const int col = col0 + 4 * c;
    // ---- column constants -------------------------------------------------------------------------
    if (tid < 2 * NOP * TF) {
      const int a = tid >> 6, cc = tid & 63;
      colv[a * TF + cc] = (col0 + cc < F) ? stats[(int64_t)a * F + col0 + cc] : 0.0f;
    }
    __syncthreads();
    // ---- load x, recompute t, standardise into LDS ---------------------------------------------
    {
      float mx[4], rx[4], mt[4], rt[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        mx[e] = colv[4 * c + e];
        rx[e] = colv[TF + 4 * c + e];
        if (PAIR) { mt[e] = colv[2 * TF + 4 * c + e]; rt[e] = colv[3 * TF + 4 * c + e]; }
      }
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        const int row = rg + 16 * j;
        const bool ok = row < B;
        const int64_t off = (int64_t)row * F + col;
        const float4 xv = load4(x, off, col, F, ok, aligned);
        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const bool okc = ok && (col + e < F);
          Xs[row * LD + 4 * c + e] = okc ? (xe[e] - mx[e]) * rx[e] : 0.0f;
          if (PAIR) {
            const float t = __fmul_rn(__fsub_rn(gauss_u1(xe[e], 0.0f, 1.0f, tab), 1.0f), r);
            Ts[row * LD + 4 * c + e] = okc ? (t - mt[e]) * rt[e] : 0.0f;
          }
        }
      }
    }
    __syncthreads();
    // ---- MFMA: accX[p] = S[I,:] Xh[:, cb(p)],  accT[p] = S[I,:] Th[:, cb(p)] ------------------------
    f32x16 accX[PX], accT[PX];
    int cb[PX];
#pragma unroll
    for (int p = 0; p < PX; p++) {
      cb[p] = ((NB == 4) ? p : (w & 1)) * 32 + l31;      // this lane's feature column inside the tile
#pragma unroll
      for (int e = 0; e < 16; e++) { accX[p][e] = 0.0f; accT[p][e] = 0.0f; }
    }
    if (wave_active) {
#pragma unroll
      for (int s = 0; s < KSTEPS; s++) {
        const int kk = 2 * s + h;
#pragma unroll
        for (int p = 0; p < PX; p++) {
          accX[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(sfrag[s], Xs[kk * LD + cb[p]], accX[p], 0, 0, 0);
          if (PAIR) accT[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(sfrag[s], Ts[kk * LD + cb[p]], accT[p], 0, 0, 0);
        }
      }
      // ---- per-feature projections over this wave's 32 batch rows: sum dVh, sum dVh*Vh ------------
#pragma unroll
      for (int p = 0; p < PX; p++) {
        float x0 = 0.f, x1 = 0.f, t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = I * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          x0 += accX[p][e];
          x1 += accX[p][e] * Xs[row * LD + cb[p]];
          if (PAIR) { t0 += accT[p][e]; t1 += accT[p][e] * Ts[row * LD + cb[p]]; }
        }
        x0 += __shfl_xor(x0, 32, 64);
        x1 += __shfl_xor(x1, 32, 64);
        if (PAIR) { t0 += __shfl_xor(t0, 32, 64); t1 += __shfl_xor(t1, 32, 64); }
        if (h == 0) {
          float* rp = red + (I * NOP * 2) * TF + cb[p];
          rp[0] = x0;
          rp[TF] = x1;
          if (PAIR) { rp[2 * TF] = t0; rp[3 * TF] = t1; }
        }
      }
    }
    __syncthreads();
    // ---- assemble and write dx (accumulator layout: 32 consecutive features per half-wave) ----------
    if (wave_active) {
#pragma unroll
      for (int p = 0; p < PX; p++) {
        float sx0 = 0.f, sx1 = 0.f, st0 = 0.f, st1 = 0.f;
#pragma unroll
        for (int rb = 0; rb < NB; rb++) {
          const float* rp = red + (rb * NOP * 2) * TF + cb[p];
          sx0 += rp[0];
          sx1 += rp[TF];
          if (PAIR) { st0 += rp[2 * TF]; st1 += rp[3 * TF]; }
        }
        const float rho_x = colv[TF + cb[p]];
        const float rho_t = PAIR ? colv[3 * TF + cb[p]] : 0.0f;
        // through-std factor (sd+eps)/sd = 1/(1-eps*rho); torch's std backward is 0 where sd == 0
        float kap_x = 1.0f, kap_t = 1.0f;
        if (eps != 0.0f) {
          const float dxn = 1.0f - eps * rho_x, dtn = 1.0f - eps * rho_t;
          kap_x = (dxn > 1e-12f) ? 1.0f / dxn : 0.0f;
          kap_t = (dtn > 1e-12f) ? 1.0f / dtn : 0.0f;
        }
        const float mean_x = sx0 * invB, proj_x = sx1 * invBm1 * kap_x;
        const float mean_t = st0 * invB, proj_t = st1 * invBm1 * kap_t;
        const bool colok = (col0 + cb[p]) < F;
#pragma unroll
        for (int e = 0; e < 16; e++) {
          const int row = I * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < B && colok) {
            const int64_t off = (int64_t)row * F + col0 + cb[p];
            const float cx = rho_x * (accX[p][e] - mean_x - Xs[row * LD + cb[p]] * proj_x);
            float out;
            if (PAIR) {
              const float ct = rho_t * (accT[p][e] - mean_t - Ts[row * LD + cb[p]] * proj_t);
              const float jac = act_jac(x[off], r);
              const float gv = g ? g[off] : 0.0f;
              out = (gv + ct) * jac - cx;       // corr(x,x) enters D with a minus sign
            } else {
              out = cx;
            }
            dx[off] = out;
          }
        }
      }
    }
    __syncthreads();   // LDS is overwritten by the next tile
  }
}

template <bool PAIR>
int launch_partials(const Geom& g, const float* x, int B, int64_t F, int k, float r, float eps, float* xq, float* stats,
                    float* ws, hipStream_t st) {
  if (g.nb == 4) return launch_partials4(PAIR, g, x, B, F, k, r, eps, xq, stats, ws, st);
  if (g.nb == 1) return launch_partials1(PAIR, g, x, B, F, k, r, eps, xq, stats, ws, st);
  const int aligned = ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                      (!xq || (reinterpret_cast<uintptr_t>(xq) & 15) == 0);
  unsigned* counter = reinterpret_cast<unsigned*>(ws + (size_t)g.grid * g.slab_floats + kPartFloats);
  if (g.nb == 1)
    hipLaunchKernelGGL((site_fwd_kernel<1, PAIR>), g.grid, kThreads, 0, st, x, B, F, k, r, eps, xq, ws, stats, g.n_tiles, aligned, counter);
  else
    hipLaunchKernelGGL((site_fwd_kernel<2, PAIR>), g.grid, kThreads, 0, st, x, B, F, k, r, eps, xq, ws, stats, g.n_tiles, aligned, counter);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <bool PAIR>
int launch_bwd(const Geom& g, const float* gup, const float* S, const float* x, const float* stats, int B, int64_t F,
               float r, float eps, float* dx, hipStream_t st) {
  if (g.nb == 4) return launch_bwd4(PAIR, g, gup, S, x, stats, B, F, r, eps, dx, st);
  if (g.nb == 1) return launch_bwd1(PAIR, gup, S, x, stats, B, F, r, eps, dx, st);
  const int n_tiles = (int)((F + TF - 1) / TF);
  const int grid = n_tiles < 2048 ? n_tiles : 2048;
  const int aligned = ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (g.nb == 1)
    hipLaunchKernelGGL((site_bwd_kernel<1, PAIR>), grid, kThreads, 0, st, gup, S, x, stats, B, F, r, eps, dx, n_tiles, aligned);
  else
    hipLaunchKernelGGL((site_bwd_kernel<2, PAIR>), grid, kThreads, 0, st, gup, S, x, stats, B, F, r, eps, dx, n_tiles, aligned);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

inline bool bad_shape(int B, int64_t F) { return B < 2 || B > ALIGNQ_MAX_BATCH || F <= 0; }
inline bool large_corr(int B, int64_t F) { return B > ALIGNQ_MAX_BATCH && B <= ALIGNQ_MAX_CORR_BATCH && F > 0; }
inline bool bad_k(int k) { return !((k >= 1 && k <= 16) || k == 32); }

}  // namespace

extern "C" {

size_t alignq_site_ws_bytes(int B, int64_t F) {
  if (large_corr(B, F)) return corrl_ws_bytes(B, F);      // alignq_corr_fwd only
  if (bad_shape(B, F)) return 0;
  return ws_floats(geom(B, F)) * sizeof(float);
}

// the S buffer of the backward: [B,B] fp32 (sym(dD) * scale / F) in its first 64 KB, followed at byte offset 65536 by the
// bf16 hi / lo fragment image of the same matrix that the prep kernels leave for the B in (64,128] backward (64 KB)
size_t alignq_site_bwd_ws_bytes(int B) {
  if (B > ALIGNQ_MAX_BATCH) return corrl_s_bytes(B);     // large batch: S only, zero-padded to the backward's tiling (<= 4 MB)
  return (size_t)2 * 128 * 128 * sizeof(float);
}

int alignq_site_partials(const float* x, int B, int64_t F, int k, float act_range, float eps, float* xq, float* stats,
                         void* ws, void* stream) {
  if (!x || !ws) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (bad_k(k)) return ALIGNQ_EINVAL;
  return launch_partials<true>(geom(B, F), x, B, F, k, act_range, eps, xq, stats, (float*)ws, (hipStream_t)stream);
}

// Small batches (B <= 32, the Office tree's 28): the same launch with the bottleneck's `out += identity; out = relu(out)`
// (dann_office/model/resnet.py:153-154) folded into the store of x_q.
int alignq_site_partials_res(const float* x, int B, int64_t F, int k, float act_range, float eps, const float* residual,
                             int relu, float* y, float* stats, void* ws, void* stream) {
  if (!x || !ws || !y) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (bad_k(k)) return ALIGNQ_EINVAL;
  const Geom g = geom(B, F);
  if (g.nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_partials1(true, g, x, B, F, k, act_range, eps, y, stats, (float*)ws, (hipStream_t)stream, residual, relu);
}

// The same with the training-mode batch-norm in front of the quantiser folded in (channels-last: channel = f mod C, C a power
// of two): the kernel reads the convolution's output z and applies x = a*z + b on load (ab from alignq_bnq_stats).
int alignq_site_partials_res_ab(const float* z, const float* ab, int C, int B, int64_t F, int k, float act_range, float eps,
                                const float* residual, int relu, float* y, float* stats, void* ws, void* stream) {
  if (!z || !ab || !ws || !y || C < 1 || (C & (C - 1)) != 0 || F % C != 0) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (bad_k(k)) return ALIGNQ_EINVAL;
  const Geom g = geom(B, F);
  if (g.nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_partials1(true, g, z, B, F, k, act_range, eps, y, stats, (float*)ws, (hipStream_t)stream, residual, relu, ab, C);
}

int alignq_site_reduce(const void* ws, int B, int64_t F, float* D, void* stream) {
  if (!ws || !D) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  return launch_reduce_any(geom(B, F), (const float*)ws, nullptr, B, F, D, false, nullptr, nullptr, 0, 0.f, 0.f, nullptr,
                           (hipStream_t)stream);
}

int alignq_site_reduce_loss(void* ws, int B, int64_t F, float* D, const float* alterD, const float* gamma, int dim,
                            float mu, float rho, float* scal, void* stream) {
  if (!ws || !D || !alterD || !gamma || !scal || dim < B) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  return launch_reduce_any(geom(B, F), (const float*)ws, (float*)ws, B, F, D, true, alterD, gamma, dim, mu, rho, scal,
                           (hipStream_t)stream);
}

int alignq_site_fwd(const float* x, int B, int64_t F, int k, float act_range, float eps, float* xq, float* D,
                    float* stats, void* ws, void* stream) {
  if (!x || !D || !ws) return ALIGNQ_EINVAL;
  if (large_corr(B, F)) {       // 128 < B <= 1024: the pair kernels on the blocked Gram (corr_large_kernels.hip); stats required
    if (!stats || bad_k(k)) return ALIGNQ_EINVAL;
    return launch_sitel_fwd(x, B, F, k, act_range, eps, xq, D, stats, (float*)ws, (hipStream_t)stream);
  }
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (bad_k(k)) return ALIGNQ_EINVAL;
  const Geom g = geom(B, F);
  int rc = launch_partials<true>(g, x, B, F, k, act_range, eps, xq, stats, (float*)ws, (hipStream_t)stream);
  if (rc) return rc;
  return launch_reduce_any(g, (const float*)ws, nullptr, B, F, D, false, nullptr, nullptr, 0, 0.f, 0.f, nullptr,
                           (hipStream_t)stream);
}

int alignq_site_bwd(const float* g, const float* dD, const float* dD_scale, const float* x, const float* stats,
                    int B, int64_t F, float act_range, float eps, float* dx, void* ws, void* stream) {
  if (!dD || !x || !stats || !dx || !ws) return ALIGNQ_EINVAL;
  if (large_corr(B, F))         // ws: alignq_site_bwd_ws_bytes(B) (the padded S)
    return launch_corrl_bwd(dD, x, stats, B, F, eps, dx, (float*)ws, (hipStream_t)stream, true, g, act_range, dD_scale);
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_prep(false, dD, nullptr, nullptr, nullptr, 0, nullptr, 0.f, dD_scale, B, F, (float*)ws, nullptr,
                       nullptr, st);
  if (rc) return rc;
  return launch_bwd<true>(geom(B, F), g, (const float*)ws, x, stats, B, F, act_range, eps, dx, st);
}

// ---- batch-norm folded forms (B in (64,128] only) ------------------------------------------------
static inline bool bn_shape_ok(int B, int64_t F, int C, int HW, int nhwc) {
  if (!(B > 64 && B <= ALIGNQ_MAX_BATCH && C >= 1 && (int64_t)C * HW == F)) return false;
  if (nhwc) return C >= 4 && C <= 256 && (C & (C - 1)) == 0 && (F % 64) == 0;   // channel = f mod C
  return HW >= 64 && (HW % 64) == 0;                                             // one channel per 64-feature tile
}

int alignq_site_partials_bn(const float* z, const void* bn_part, const float* bn_gamma, const float* bn_beta,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            float bn_eps, float* ab, float* save, int C, int HW, int B, int64_t F, int k, float act_range,
                            float eps, int relu, const float* residual, int nhwc, int conv_parts, float* xq, void* bins_out,
                            float* stats, void* ws, void* stream) {
  return alignq_site_partials_bn_fill(z, bn_part, bn_gamma, bn_beta, running_mean, running_var, num_batches_tracked, momentum,
                                      bn_eps, ab, save, C, HW, B, F, k, act_range, eps, relu, residual, nhwc, conv_parts, xq,
                                      bins_out, stats, ws, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0.f, 0.f,
                                      stream);
}

int alignq_site_fill_slots(int B, int64_t F) { return site_fill_slots(B, F); }

int alignq_site_partials_bn_fill(const float* z, const void* bn_part, const float* bn_gamma, const float* bn_beta,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                 float bn_eps, float* ab, float* save, int C, int HW, int B, int64_t F, int k,
                                 float act_range, float eps, int relu, const float* residual, int nhwc, int conv_parts,
                                 float* xq, void* bins_out, float* stats, void* ws, int n_fill, void* const* fill_ws,
                                 float* const* fill_D, const float* const* fill_alterD, const float* const* fill_gamma,
                                 float* const* fill_scal, const int64_t* fill_F, int fill_dim, float fill_mu, float fill_rho,
                                 void* stream) {
  if (!z || !ab || !save || !ws) return ALIGNQ_EINVAL;
  if (n_fill < 0 || n_fill > site_fill_slots(B, F)) return ALIGNQ_EINVAL;
  if (n_fill && (!fill_ws || !fill_D || !fill_alterD || !fill_gamma || !fill_scal || !fill_F || fill_dim < B)) return ALIGNQ_EINVAL;
  const SiteFillArgs fa{n_fill, fill_ws, fill_D, fill_alterD, fill_gamma, fill_scal, fill_F, fill_dim, fill_mu, fill_rho};
  if (bad_k(k)) return ALIGNQ_EINVAL;
  if (!bn_shape_ok(B, F, C, HW, nhwc)) return ALIGNQ_EUNSUPPORTED;
  BnFold bn = no_bn();
  bn.ab = ab; bn.save = save; bn.HW = HW; bn.C = C; bn.nhwc = nhwc;
  bn.part = (const double*)bn_part; bn.gamma = bn_gamma; bn.beta = bn_beta;
  bn.running_mean = running_mean; bn.running_var = running_var; bn.nbt = (long long*)num_batches_tracked;
  bn.momentum = momentum; bn.bn_eps = bn_eps; bn.relu = relu; bn.res = residual;
  if (bins_out) {                  // N2: the stored activation's level index in alignq_bin_bytes(k, act_range, ADMM) bytes
    bn.bins = bins_out;
    bn.bin_bytes = alignq_bin_bytes(k, act_range, ALIGNQ_FORMULA_ADMM);
    if (bn.bin_bytes == 0 || residual) return ALIGNQ_EINVAL;
  }
  if (conv_parts > 0) {          // bn_part holds the producing convolution's per-workgroup float partials
    if (!nhwc || !bn_part) return ALIGNQ_EINVAL;
    bn.n_parts = conv_parts; bn.part_f32 = 1;
  }
  return launch_partials4(true, geom(B, F), z, B, F, k, act_range, eps, xq, stats, (float*)ws, (hipStream_t)stream, bn,
                          n_fill ? &fa : nullptr);
}

// Two sites of one shape in one launch (include/alignq.h): each described as alignq_site_partials_bn's arguments.
static int site_bn_fold(const alignq_site_bn_args& a, BnFold* out) {
  if (!a.z || !a.ab || !a.save || !a.ws || !a.stats) return ALIGNQ_EINVAL;
  if (bad_k(a.k)) return ALIGNQ_EINVAL;
  if (!bn_shape_ok(a.B, a.F, a.C, a.HW, a.nhwc)) return ALIGNQ_EUNSUPPORTED;
  BnFold bn = no_bn();
  bn.ab = a.ab; bn.save = a.save; bn.HW = a.HW; bn.C = a.C; bn.nhwc = a.nhwc;
  bn.part = (const double*)a.bn_part; bn.gamma = a.bn_gamma; bn.beta = a.bn_beta;
  bn.running_mean = a.running_mean; bn.running_var = a.running_var; bn.nbt = (long long*)a.num_batches_tracked;
  bn.momentum = a.momentum; bn.bn_eps = a.bn_eps; bn.relu = a.relu; bn.res = a.residual;
  if (a.bins_out) {
    bn.bins = a.bins_out;
    bn.bin_bytes = alignq_bin_bytes(a.k, a.act_range, ALIGNQ_FORMULA_ADMM);
    if (bn.bin_bytes == 0 || a.residual) return ALIGNQ_EINVAL;
  }
  if (a.conv_parts > 0) {
    if (!a.nhwc || !a.bn_part) return ALIGNQ_EINVAL;
    bn.n_parts = a.conv_parts; bn.part_f32 = 1;
  }
  *out = bn;
  return 0;
}

int alignq_site_partials_bn_twin(const alignq_site_bn_args* a, const alignq_site_bn_args* b, void* stream) {
  if (!a || !b) return ALIGNQ_EINVAL;
  if (a->B != b->B || a->F != b->F || a->k != b->k || a->act_range != b->act_range || a->eps != b->eps) return ALIGNQ_EUNSUPPORTED;
  if (a->ws == b->ws || a->stats == b->stats || (a->xq && a->xq == b->xq) || a->ab == b->ab) return ALIGNQ_EINVAL;   // two sites, two sets of buffers
  BnFold bna, bnb;
  int rc = site_bn_fold(*a, &bna);
  if (rc) return rc;
  rc = site_bn_fold(*b, &bnb);
  if (rc) return rc;
  return launch_partials4_twin(geom(a->B, a->F), a->B, a->F, a->k, a->act_range, a->eps, a->z, a->xq, a->stats, (float*)a->ws, bna,
                               b->z, b->xq, b->stats, (float*)b->ws, bnb, (hipStream_t)stream);
}

size_t alignq_site_bn_part_bytes(int64_t F, int nhwc) {
  return (nhwc ? (size_t)F : (size_t)((F + 31) / 32)) * 2 * sizeof(float);   // <= per column | per smallest backward tile
}

int alignq_site_bwd_apply_bn(const float* g, const float* S, const float* z, const float* ab, const float* save, int C,
                             int HW, int nhwc, const float* y_relu, const void* y_bins, int y_bin_bytes, float* dresidual,
                             const float* stats, int B, int64_t F, float act_range, float eps, float* dx, float* dx_part,
                             void* stream) {
  return alignq_site_bwd_apply_bn_fill(g, S, z, ab, save, C, HW, nhwc, y_relu, y_bins, y_bin_bytes, dresidual, stats, B, F,
                                       act_range, eps, dx, dx_part, 0, nullptr, nullptr, nullptr, nullptr, stream);
}

int alignq_site_bwd_fill_slots(int B, int64_t F) { return bwd_fill_ok(B, F) ? alignq_wgr::kFill : 0; }

int alignq_site_bwd_apply_bn_fill(const float* g, const float* S, const float* z, const float* ab, const float* save, int C,
                                  int HW, int nhwc, const float* y_relu, const void* y_bins, int y_bin_bytes,
                                  float* dresidual, const float* stats, int B, int64_t F, float act_range, float eps,
                                  float* dx, float* dx_part, int n_fill, const void* const* fill_ws, float* const* fill_dw,
                                  const int* fill_n_slabs, const int* fill_n_elem, void* stream) {
  if (!S || !z || !ab || !save || !stats || !dx || !dx_part) return ALIGNQ_EINVAL;
  if (n_fill < 0 || n_fill > alignq_site_bwd_fill_slots(B, F)) return ALIGNQ_EINVAL;
  if (n_fill && (!fill_ws || !fill_dw || !fill_n_slabs || !fill_n_elem)) return ALIGNQ_EINVAL;
  alignq_wgr::RedFill fill{};
  for (int i = 0; i < n_fill; i++) {
    if (!fill_ws[i] || !fill_dw[i] || fill_n_slabs[i] < 1 || fill_n_elem[i] < 1) return ALIGNQ_EINVAL;
    fill.slabs[i] = (const float*)fill_ws[i]; fill.dw[i] = fill_dw[i]; fill.n_slabs[i] = fill_n_slabs[i];
    fill.n_elem[i] = fill_n_elem[i];
    fill.blk0[i + 1] = fill.blk0[i] + alignq_wgr::wgrad_reduce_blocks(fill_n_slabs[i], fill_n_elem[i], 256);
  }
  for (int i = n_fill; i < alignq_wgr::kFill; i++) fill.blk0[i + 1] = fill.blk0[i];
  if (y_bins && (y_relu || (y_bin_bytes != 1 && y_bin_bytes != 2))) return ALIGNQ_EINVAL;
  if (!bn_shape_ok(B, F, C, HW, nhwc)) return ALIGNQ_EUNSUPPORTED;
  BnFold bn = no_bn();
  bn.ab = ab; bn.save = save; bn.HW = HW; bn.C = C; bn.nhwc = nhwc;
  bn.dx_part = dx_part; bn.y = y_relu; bn.dres = dresidual; bn.ybins = y_bins; bn.bin_bytes = y_bin_bytes;
  return launch_bwd4(true, geom(B, F), g, S, z, stats, B, F, act_range, eps, dx, (hipStream_t)stream, bn, n_fill ? &fill : nullptr);
}

// The backward of two sites of one shape in one launch (include/alignq.h): each described as alignq_site_bwd_apply_bn's arguments.
int alignq_site_bwd_apply_bn_twin(const alignq_site_bwd_bn_args* a, const alignq_site_bwd_bn_args* b, void* stream) {
  if (!a || !b) return ALIGNQ_EINVAL;
  if (a->B != b->B || a->F != b->F || a->act_range != b->act_range || a->eps != b->eps) return ALIGNQ_EUNSUPPORTED;
  BnFold bn[2];
  const alignq_site_bwd_bn_args* two[2] = {a, b};
  for (int i = 0; i < 2; i++) {
    const alignq_site_bwd_bn_args& q = *two[i];
    if (!q.S || !q.z || !q.ab || !q.save || !q.stats || !q.dx || !q.dx_part) return ALIGNQ_EINVAL;
    if (q.y_bins && (q.y_relu || (q.y_bin_bytes != 1 && q.y_bin_bytes != 2))) return ALIGNQ_EINVAL;
    if (!bn_shape_ok(q.B, q.F, q.C, q.HW, q.nhwc)) return ALIGNQ_EUNSUPPORTED;
    bn[i] = no_bn();
    bn[i].ab = q.ab; bn[i].save = q.save; bn[i].HW = q.HW; bn[i].C = q.C; bn[i].nhwc = q.nhwc;
    bn[i].dx_part = q.dx_part; bn[i].y = q.y_relu; bn[i].dres = q.dresidual; bn[i].ybins = q.y_bins; bn[i].bin_bytes = q.y_bin_bytes;
  }
  if (a->dx == b->dx || a->dx_part == b->dx_part) return ALIGNQ_EINVAL;
  return launch_bwd4_twin(a->B, a->F, a->act_range, a->eps, a->g, a->S, a->z, a->stats, a->dx, bn[0], b->g, b->S, b->z, b->stats, b->dx,
                          bn[1], (hipStream_t)stream);
}

int alignq_site_prep_fused(const float* D, const float* alterD, const float* gamma, int dim, const float* scal, float mu,
                           const float* dD_scale, int B, int64_t F, float* S, float* dalterD, float* dgamma,
                           void* stream) {
  if (!D || !alterD || !gamma || !scal || !S || dim < B) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  return launch_prep(true, nullptr, D, alterD, gamma, dim, scal, mu, dD_scale, B, F, S, dalterD, dgamma,
                     (hipStream_t)stream);
}

int alignq_site_reduce_loss_multi(int S, void* const* ws, float* const* D, const float* const* alterD,
                                  const float* const* gamma, float* const* scal, const int64_t* F, int B, int dim,
                                  float mu, float rho, void* stream) {
  if (S <= 0 || !ws || !D || !alterD || !gamma || !scal || !F || dim < B) return ALIGNQ_EINVAL;
  if (B <= 64 || B > ALIGNQ_MAX_BATCH) return ALIGNQ_EUNSUPPORTED;
  for (int i = 0; i < S; i++)
    if (!ws[i] || !D[i] || !alterD[i] || !gamma[i] || !scal[i] || F[i] <= 0) return ALIGNQ_EINVAL;
  return launch_reduce_loss_multi(S, ws, D, alterD, gamma, scal, F, B, dim, mu, rho, (hipStream_t)stream);
}

int alignq_site_reduce_loss_multi_head(int S, void* const* ws, float* const* D, const float* const* alterD,
                                       const float* const* gamma, float* const* scal, const int64_t* F, int B, int dim, float mu,
                                       float rho, const float* feat, const float* W, const float* bias, const int64_t* target, int HB,
                                       int HW, int C, int K, float* pooled, float* logits, float* probs, float* loss, float* ce_mean,
                                       unsigned* head_counter, const float* scal_all, int n_sites, float* trans_total,
                                       unsigned* site_counter, void* stream) {
  if (S <= 0 || !ws || !D || !alterD || !gamma || !scal || !F || dim < B) return ALIGNQ_EINVAL;
  if (B <= 64 || B > ALIGNQ_MAX_BATCH) return ALIGNQ_EUNSUPPORTED;
  for (int i = 0; i < S; i++)
    if (!ws[i] || !D[i] || !alterD[i] || !gamma[i] || !scal[i] || F[i] <= 0) return ALIGNQ_EINVAL;
  if (!feat || !W || !target || !pooled || !logits || !probs || !loss || !ce_mean || !head_counter || HB < 1 || HW < 1) return ALIGNQ_EINVAL;
  if (!scal_all || n_sites < S || !trans_total || !site_counter) return ALIGNQ_EINVAL;
  if (C < 1 || C > 256 || K < 1 || K > 64) return ALIGNQ_EUNSUPPORTED;      // (alignq_head::kMaxC / kMaxK, as alignq_head_ce_fwd)
  return launch_reduce_loss_multi_head(S, ws, D, alterD, gamma, scal, F, B, dim, mu, rho, feat, W, bias, target, HB, HW, C, K, pooled,
                                       logits, probs, loss, ce_mean, head_counter, scal_all, n_sites, trans_total, site_counter,
                                       (hipStream_t)stream);
}

int alignq_site_prep_fused_multi(int S, const float* const* D, const float* const* alterD, const float* const* gamma,
                                 const float* const* scal, const float* dD_scale, const int64_t* F, int B, int dim,
                                 float mu, float* const* S_out, float* const* dalterD, float* const* dgamma,
                                 void* stream) {
  if (S <= 0 || !D || !alterD || !gamma || !scal || !F || !S_out || !dalterD || !dgamma || dim < B) return ALIGNQ_EINVAL;
  for (int i = 0; i < S; i++)
    if (!D[i] || !alterD[i] || !gamma[i] || !scal[i] || !S_out[i] || !dalterD[i] || !dgamma[i] || F[i] <= 0)
      return ALIGNQ_EINVAL;
  return launch_prep_multi(S, D, alterD, gamma, scal, dD_scale, F, B, dim, mu, S_out, dalterD, dgamma, (hipStream_t)stream);
}

int alignq_head_ce_bwd_site_prep(const float* g_ce, const float* probs, const int64_t* target, const float* pooled,
                                 const float* W, int HB, int HW, int C, int K, float* dfeat, float* dW, float* dbias, int S,
                                 const float* const* D, const float* const* alterD, const float* const* gamma,
                                 const float* const* scal, const float* dD_scale, const int64_t* F, int B, int dim, float mu,
                                 float* const* S_out, float* const* dalterD, float* const* dgamma, void* stream) {
  if (!g_ce || !probs || !target || !pooled || !W || !dfeat || !dW || HB < 1 || HW < 1) return ALIGNQ_EINVAL;
  if (S <= 0 || !D || !alterD || !gamma || !scal || !F || !S_out || !dalterD || !dgamma || dim < B) return ALIGNQ_EINVAL;
  for (int i = 0; i < S; i++)
    if (!D[i] || !alterD[i] || !gamma[i] || !scal[i] || !S_out[i] || !dalterD[i] || !dgamma[i] || F[i] <= 0)
      return ALIGNQ_EINVAL;
  return launch_head_bwd_prep_multi(g_ce, probs, target, pooled, W, HB, HW, C, K, dfeat, dW, dbias, S, D, alterD, gamma, scal,
                                    dD_scale, F, B, dim, mu, S_out, dalterD, dgamma, (hipStream_t)stream);
}

int alignq_site_bwd_apply(const float* g, const float* S, const float* x, const float* stats, int B, int64_t F,
                          float act_range, float eps, float* dx, void* stream) {
  if (!S || !x || !stats || !dx) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  return launch_bwd<true>(geom(B, F), g, S, x, stats, B, F, act_range, eps, dx, (hipStream_t)stream);
}

// backward of alignq_site_partials_res_ab: dx w.r.t. the batch-norm OUTPUT x = a*z + b (B <= 32); alignq_bnq_bwd_dx turns it
// into dz, dgamma, dbeta
int alignq_site_bwd_apply_ab(const float* g, const float* S, const float* z, const float* ab, int C, const float* stats, int B,
                             int64_t F, float act_range, float eps, float* dx, void* stream) {
  if (!S || !z || !ab || !stats || !dx || C < 1 || (C & (C - 1)) != 0 || F % C != 0) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (geom(B, F).nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_bwd1(true, g, S, z, stats, B, F, act_range, eps, dx, (hipStream_t)stream, ab, C);
}

// the same with the ReLU behind the site folded in: g is the gradient w.r.t. relu(x_q + residual), y that output (mask y > 0);
// dres (optional) receives the masked gradient = the gradient of the residual operand
int alignq_site_bwd_apply_ab_relu(const float* g, const float* y, const float* S, const float* z, const float* ab, int C,
                                  const float* stats, int B, int64_t F, float act_range, float eps, float* dx, float* dres,
                                  void* stream) {
  if (!g || !y || !S || !z || !ab || !stats || !dx || C < 1 || (C & (C - 1)) != 0 || F % C != 0) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (geom(B, F).nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_bwd1(true, g, S, z, stats, B, F, act_range, eps, dx, (hipStream_t)stream, ab, C, y, dres);
}

// ---- the B <= 32 site for `groups` batch slices stacked along the batch (the Office step's merged source + target pass):
// ONE launch per kernel, blockIdx.y = slice.  z / residual / y / g / dx / dres: [groups][B][F]; ab: [groups][2][C]; stats:
// [groups][4][F]; D: [groups][B][B]; scal: [groups][4]; ws: groups regions of alignq_site_ws_bytes(B, F); S: groups regions of
// alignq_site_bwd_ws_bytes(B) (alignq_site_prep_fused_multi fills them, one "site" per slice).
int alignq_site1_groups_fwd(const float* z, const float* ab, int C, int B, int64_t F, int groups, int k, float act_range,
                            float eps, const float* residual, int relu, float* y, float* stats, void* ws, void* stream) {
  if (!z || !ab || !ws || !y || groups < 1 || C < 1 || (C & (C - 1)) != 0 || F % C != 0) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (bad_k(k)) return ALIGNQ_EINVAL;
  const Geom g = geom(B, F);
  if (g.nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_partials1(true, g, z, B, F, k, act_range, eps, y, stats, (float*)ws, (hipStream_t)stream, residual, relu, ab, C,
                          groups, (int64_t)(alignq_site_ws_bytes(B, F) / 4));
}

// (round 5) the same with the stored output's sign bits for the backward (alignq_site1_groups_bwd_bn_m): one bit per element,
// [groups][ceil(F / 32)][32 features] words, bit = row - the backward then reads 0.14 B per element for its ReLU mask instead of 4 B of y
size_t alignq_site1_mask_bytes(int B, int64_t F, int groups) {
  if (bad_shape(B, F) || groups < 1) return 0;
  return (size_t)groups * (size_t)((F + 31) / 32) * 32 * sizeof(unsigned);
}

int alignq_site1_groups_fwd_m(const float* z, const float* ab, int C, int B, int64_t F, int groups, int k, float act_range,
                              float eps, const float* residual, int relu, float* y, float* stats, void* ws, void* relu_mask,
                              void* stream) {
  if (!z || !ab || !ws || !y || !relu_mask || groups < 1 || C < 1 || (C & (C - 1)) != 0 || F % C != 0) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (bad_k(k)) return ALIGNQ_EINVAL;
  const Geom g = geom(B, F);
  if (g.nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_partials1(true, g, z, B, F, k, act_range, eps, y, stats, (float*)ws, (hipStream_t)stream, residual, relu, ab, C,
                          groups, (int64_t)(alignq_site_ws_bytes(B, F) / 4), (unsigned*)relu_mask);
}

int alignq_site1_groups_reduce_loss(void* ws, int B, int64_t F, int groups, float* D, const float* alterD, const float* gamma,
                                    int dim, float mu, float rho, float* scal, void* stream) {
  if (!ws || !D || !alterD || !gamma || !scal || groups < 1 || dim < B) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  const Geom g = geom(B, F);
  if (g.nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_reduce_loss_groups(g, (float*)ws, B, F, groups, D, alterD, gamma, dim, mu, rho, scal,
                                   (int64_t)(alignq_site_ws_bytes(B, F) / 4), (hipStream_t)stream);
}

// the backward's preparation for all slices in one launch: S per slice + dalterD / dgamma SUMMED over the slices (slice order)
int alignq_site1_groups_prep(const float* D, const float* alterD, const float* gamma, int dim, const float* scal, float mu,
                             const float* dD_scale, int dD_scale_stride, int B, int64_t F, int groups, float* S, float* dalterD,
                             float* dgamma, void* stream) {
  if (!D || !alterD || !gamma || !scal || !S || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS || dim < B || dD_scale_stride < 0)
    return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (geom(B, F).nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_prep_groups(D, alterD, gamma, dim, scal, mu, dD_scale, B, F, groups, S, (int64_t)(alignq_site_bwd_ws_bytes(B) / 4),
                            dalterD, dgamma, (hipStream_t)stream, dD_scale_stride);
}

// (round 5) the two launches above for SEVERAL sites at once: the Office step's bottleneck tails defer them to the end of the
// forward / the start of the backward (one launch each instead of 16); bit-identical to the per-site calls
int alignq_site1_groups_reduce_loss_multi(int T, void* const* ws, const int64_t* F, int B, int groups, float* const* D,
                                          const float* const* alterD, const float* const* gamma, int dim, float mu, float rho,
                                          float* const* scal, void* stream) {
  if (T < 1 || !ws || !F || !D || !alterD || !gamma || !scal || groups < 1 || dim < B) return ALIGNQ_EINVAL;
  for (int i = 0; i < T; i++) {
    if (!ws[i] || !D[i] || !alterD[i] || !gamma[i] || !scal[i]) return ALIGNQ_EINVAL;
    if (bad_shape(B, F[i])) return F[i] <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
    if (geom(B, F[i]).nb != 1) return ALIGNQ_EUNSUPPORTED;
  }
  return launch_reduce_loss_groups_multi(T, (float* const*)ws, F, B, groups, D, alterD, gamma, dim, mu, rho, scal, (hipStream_t)stream);
}

int alignq_site1_groups_prep_multi(int T, const float* const* D, const float* const* alterD, const float* const* gamma, int dim,
                                   const float* const* scal, float mu, const float* dD_scale, int B, const int64_t* F, int groups,
                                   float* const* S, float* const* dalterD, float* const* dgamma, void* stream) {
  if (T < 1 || !D || !alterD || !gamma || !scal || !S || !F || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS || dim < B)
    return ALIGNQ_EINVAL;
  for (int i = 0; i < T; i++) {
    if (!D[i] || !alterD[i] || !gamma[i] || !scal[i] || !S[i]) return ALIGNQ_EINVAL;
    if (bad_shape(B, F[i])) return F[i] <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
    if (geom(B, F[i]).nb != 1) return ALIGNQ_EUNSUPPORTED;
  }
  return launch_prep_groups_multi(T, D, alterD, gamma, dim, scal, mu, dD_scale, B, F, groups, S,
                                  (int64_t)(alignq_site_bwd_ws_bytes(B) / 4), dalterD, dgamma, (hipStream_t)stream);
}

int alignq_site1_groups_bwd(const float* g, const float* g2, const float* y, const float* S, const float* z, const float* ab, int C,
                            const float* stats, int B, int64_t F, int groups, float act_range, float eps, float* dx,
                            float* dres, void* stream) {
  if (!S || !z || !ab || !stats || !dx || groups < 1 || (g && !y) || (g2 && !g) || C < 1 || (C & (C - 1)) != 0 || F % C != 0)
    return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (geom(B, F).nb != 1) return ALIGNQ_EUNSUPPORTED;
  return launch_bwd1(true, g, S, z, stats, B, F, act_range, eps, dx, (hipStream_t)stream, ab, C, g ? y : nullptr, g ? dres : nullptr,
                     groups, (int64_t)(alignq_site_bwd_ws_bytes(B) / 4), g2);
}

// alignq_site1_groups_bwd + alignq_bnq_bwd_dx with the batch-norm backward's first pass replaced by per-column sums the site
// kernel leaves (round 4): cols = alignq_site1_cols_bytes(F, groups) bytes of scratch
size_t alignq_site1_cols_bytes(int64_t F, int groups) {
  if (F <= 0 || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS) return 0;
  return (size_t)2 * groups * F * sizeof(float);
}

int alignq_site1_groups_bwd_bn(const float* g, const float* g2, const float* y, const float* S, const float* z, const float* ab,
                               const float* save, int C, const float* stats, int B, int64_t F, int groups, float act_range, float eps,
                               float* dz, float* dres, float* dgamma, float* dbeta, void* cols, void* ws_bn, void* stream) {
  if (!S || !z || !ab || !save || !stats || !dz || !cols || !ws_bn || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS || (g && !y) ||
      (g2 && !g) || C < 1 || (C & (C - 1)) != 0 || F % C != 0)
    return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (geom(B, F).nb != 1) return ALIGNQ_EUNSUPPORTED;
  const int64_t HW = F / C, P = (int64_t)B * HW;
  if (P < 2) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_bwd1(true, g, S, z, stats, B, F, act_range, eps, dz, st, ab, C, g ? y : nullptr, g ? dres : nullptr, groups,
                             (int64_t)(alignq_site_bwd_ws_bytes(B) / 4), g2, save, (float*)cols);
  if (rc) return rc;
  return launch_bnq_bwd_from_cols((const float*)cols, dz, z, ab, save, P, HW, C, groups, dz, dgamma, dbeta, ws_bn, st);
}

int alignq_site1_groups_bwd_bn_m(const float* g, const float* g2, const void* relu_mask, const float* S, const float* z,
                                 const float* ab, const float* save, int C, const float* stats, int B, int64_t F, int groups,
                                 float act_range, float eps, float* dz, float* dres, float* dgamma, float* dbeta, void* cols,
                                 void* ws_bn, void* stream) {
  if (!S || !z || !ab || !save || !stats || !dz || !cols || !ws_bn || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS ||
      (g && !relu_mask) || (g2 && !g) || C < 1 || (C & (C - 1)) != 0 || F % C != 0)
    return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  if (geom(B, F).nb != 1) return ALIGNQ_EUNSUPPORTED;
  const int64_t HW = F / C, P = (int64_t)B * HW;
  if (P < 2) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int rc = launch_bwd1(true, g, S, z, stats, B, F, act_range, eps, dz, st, ab, C, nullptr, g ? dres : nullptr, groups,
                             (int64_t)(alignq_site_bwd_ws_bytes(B) / 4), g2, save, (float*)cols, g ? (const unsigned*)relu_mask : nullptr);
  if (rc) return rc;
  return launch_bnq_bwd_from_cols((const float*)cols, dz, z, ab, save, P, HW, C, groups, dz, dgamma, dbeta, ws_bn, st);
}

int alignq_site_bwd_fused(const float* g, const float* D, const float* alterD, const float* gamma, int dim,
                          const float* scal, float mu, const float* dD_scale, const float* x, const float* stats,
                          int B, int64_t F, float act_range, float eps, float* dx, float* dalterD, float* dgamma,
                          void* ws, void* stream) {
  if (!D || !alterD || !gamma || !scal || !x || !stats || !dx || !ws || dim < B) return ALIGNQ_EINVAL;
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_prep(true, nullptr, D, alterD, gamma, dim, scal, mu, dD_scale, B, F, (float*)ws, dalterD, dgamma, st);
  if (rc) return rc;
  return launch_bwd<true>(geom(B, F), g, (const float*)ws, x, stats, B, F, act_range, eps, dx, st);
}

int alignq_corr_fwd(const float* x, int B, int64_t F, float eps, float* G, float* stats, void* ws, void* stream) {
  if (!x || !G || !ws) return ALIGNQ_EINVAL;
  if (large_corr(B, F)) {       // 128 < B <= 1024: blocked Gram (corr_large_kernels.hip); stats is required there
    if (!stats) return ALIGNQ_EINVAL;
    return launch_corrl_fwd(x, B, F, eps, G, stats, (float*)ws, (hipStream_t)stream);
  }
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  const Geom g = geom(B, F);
  int rc = launch_partials<false>(g, x, B, F, 32, 1.0f, eps, nullptr, stats, (float*)ws, (hipStream_t)stream);
  if (rc) return rc;
  return launch_reduce_any(g, (const float*)ws, nullptr, B, F, G, false, nullptr, nullptr, 0, 0.f, 0.f, nullptr,
                           (hipStream_t)stream);
}

int alignq_corr_bwd(const float* dG, const float* x, const float* stats, int B, int64_t F, float eps, float* dx,
                    void* ws, void* stream) {
  if (!dG || !x || !stats || !dx || !ws) return ALIGNQ_EINVAL;
  if (large_corr(B, F)) return launch_corrl_bwd(dG, x, stats, B, F, eps, dx, (float*)ws, (hipStream_t)stream);
  if (bad_shape(B, F)) return F <= 0 ? ALIGNQ_EINVAL : ALIGNQ_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_prep(false, dG, nullptr, nullptr, nullptr, 0, nullptr, 0.f, nullptr, B, F, (float*)ws, nullptr, nullptr,
                       st);
  if (rc) return rc;
  return launch_bwd<false>(geom(B, F), nullptr, (const float*)ws, x, stats, B, F, 1.0f, eps, dx, st);
}

}  // extern "C"
