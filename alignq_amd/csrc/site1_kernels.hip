// site1_kernels.hip — ADMM-site kernels for SMALL batches (2 <= B <= 32: Office-31's train batch 28, short last
// batches) on gfx950.  These sites are HBM-bound (B flop/byte is below the machine balance) and can be
// huge (config 5: 28 x 802816 = 90 MB), so the design goal is streaming without workgroup barriers:
//
//   * each WAVE owns 32-feature sub-tiles, two lanes per feature column (16 batch rows each): a lane loads its rows
//     (every wave instruction reads two 128-byte row segments), so the per-feature batch statistics are register
//     reductions plus ONE cross-half shuffle — no LDS, no __syncthreads anywhere in the main loop; 16 rows per lane
//     keep the kernels at <=~100 VGPRs (a one-lane-per-column variant needed 128/203 and ran at 3/2 waves per SIMD);
//   * the standardised column is packed as (bf16 hi << 16 | bf16 lo) words and written to a wave-private LDS buffer
//     [feature][row] with 16-byte stores; the same wave reads it back in MFMA fragment order (lane = row, 8 consecutive
//     features: conflict-free 4-byte reads, hi and lo in one word) and runs the 3-term split-bf16 Gram
//     (v_mfma_f32_32x32x16_bf16) into ONE 32x32 accumulator per wave, D = T-Gram - X-Gram (x A-operand sign-flipped);
//   * backward (rewritten in round 4): S*Xh / S*Th on the FP32 matrix instruction with operands and results in registers only - a
//     lane owns exactly the rows the accumulator hands it, so nothing is split into bf16 halves, staged or transposed (see
//     site1_bwd_kernel); the standardisation backward (batch projections) is register arithmetic, dx leaves as row segments.
//   * the four waves of a workgroup only meet at the very end of the forward to add their accumulators into the slab.
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "site_internal.h"

using namespace alignq;

namespace alignq_site {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWaves = 4;
constexpr int kThreads1 = 64 * kWaves;
constexpr int SUBF = 32;                // features per wave iteration: lane = (feature l&31, row half l>>5)
constexpr int RPL = 16;                 // rows per lane (two lanes share a feature column: halves of 16 rows)
constexpr int LDW = 36;                 // words per feature row of the wave buffer (32 rows + pad; 144 B: 16-B aligned)
constexpr int WBUF = SUBF * LDW;        // words per wave buffer

__device__ __forceinline__ unsigned pack_hi_lo(float v) {
  const __bf16 hi = (__bf16)v;
  const __bf16 lo = (__bf16)(v - (float)hi);
  return ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16) | (unsigned)__builtin_bit_cast(unsigned short, lo);
}

// 8 packed words -> bf16x8 of the high halves / of the low halves (element j from word j)
__device__ __forceinline__ void unpack8(const unsigned (&wd)[8], bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    h[q] = (wd[2 * q] >> 16) | (wd[2 * q + 1] & 0xffff0000u);
    l[q] = (wd[2 * q] & 0xffffu) | (wd[2 * q + 1] << 16);
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

__device__ __forceinline__ bf16x8 neg8(bf16x8 v) {
  u32x4 u = __builtin_bit_cast(u32x4, v);
  u ^= (u32x4){0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
  return __builtin_bit_cast(bf16x8, u);
}

// wave-private LDS hand-off between lanes of ONE wave: LDS operations of a wave complete in order; the fence keeps
// the compiler from moving accesses across the phase boundary.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// stage this lane's 16 standardised rows of feature l31 (zeros beyond B) into the wave buffer: W[feature][row]
__device__ __forceinline__ void stage_rows(unsigned* __restrict__ W, int l31, int hh, const unsigned (&wd)[RPL]) {
  u32x4* dst = reinterpret_cast<u32x4*>(W + l31 * LDW + RPL * hh);
#pragma unroll
  for (int q = 0; q < RPL / 4; q++) dst[q] = (u32x4){wd[4 * q], wd[4 * q + 1], wd[4 * q + 2], wd[4 * q + 3]};
}

// Gram of the staged 32 features: acc (+/-)= V V^T, lane = (row l&31, feature group h): 2 K-steps of 16 features
template <bool NEG>
__device__ __forceinline__ void gram32(const unsigned* __restrict__ W, int l31, int h, f32x16& acc) {
#pragma unroll
  for (int ks = 0; ks < SUBF / 16; ks++) {
    unsigned wd[8];
#pragma unroll
    for (int j = 0; j < 8; j++) wd[j] = W[(16 * ks + 8 * h + j) * LDW + l31];
    bf16x8 hi, lo;
    unpack8(wd, hi, lo);
    const bf16x8 ahi = NEG ? neg8(hi) : hi, alo = NEG ? neg8(lo) : lo;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, lo, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, hi, acc, 0, 0, 0);
  }
}

// ================================================================================================ forward
// res (or nullptr): a [B,F] tensor added to x_q before it is stored (the bottleneck's `out += identity`); relu: store relu(.)
// (`out = self.relu(out)`): dann_office/model/resnet.py:153-154 — the stored tensor is what the next layer reads.
// BND (PAIR only): the launch's (k, act_range) keep every level index in the verified range (Levels.yn != 0, alignq_math.h): the
// quantiser is the straight-line form.  Without it every ROW carried the scalar switch on k (k == 32 / k == 1 / bounded) that the
// generic form needs - five scalar branches per row, 80 per sub-tile, in a kernel that lives on issue slots.
template <bool PAIR, bool RES, bool BND = false>
__global__ __launch_bounds__(kThreads1, PAIR ? 4 : 5) void site1_fwd_kernel(const float* __restrict__ x, int B, int64_t F, int k,
                                                              float r, float eps, float* __restrict__ xq,
                                                              float* __restrict__ slabs, float* __restrict__ stats,
                                                              int n_sub, unsigned* __restrict__ counter,
                                                              const float* __restrict__ res, int relu,
                                                              const float* __restrict__ ab, int nch, int64_t ws_gstride,
                                                              unsigned* __restrict__ rmask = nullptr) {
  // blockIdx.y = group: batch slices of a merged multi-pass tensor ([groups][B][F]), each with its own statistics, slabs
  // (workspace regions ws_gstride floats apart) and (a, b)
  {
    const int64_t gi = blockIdx.y, go = gi * (int64_t)B * F;
    x += go;
    if (xq) xq += go;
    if (res) res += go;
    if (stats) stats += gi * 4 * F;
    slabs += gi * ws_gstride;
    if (counter) counter += gi * ws_gstride;
    if (ab) ab += gi * 2 * nch;
    if (rmask) rmask += gi * (int64_t)n_sub * 32;
  }
  __shared__ __attribute__((aligned(16))) unsigned lds[(kWaves * WBUF > 4096) ? kWaves * WBUF : 4096];
  __shared__ __attribute__((aligned(16))) float nerf_lds[PAIR ? ALIGNQ_NERF_LDS_FLOATS : 4];
  if (PAIR) {
    nerf_tab_load(nerf_lds);
    __syncthreads();
  }
  const NerfTab tab = nerf_tab(nerf_lds);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;      // h doubles as the row half of the load mapping
  unsigned* W = lds + w * WBUF;
  // launch-uniform floats: into scalar registers (computed on the vector unit they would each hold a vector register of a kernel
  // that sits at the 128 four waves per SIMD allow)
  auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  nlev.n = uni(nlev.n);
  nlev.yn = uni(nlev.yn);
  const float invBm1 = uni(1.0f / (float)(B - 1));
  if (blockIdx.x == 0 && tid == 0 && counter) *counter = 0u;

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.0f;

  // software pipeline: the rows of the NEXT sub-tile are requested before this one is transformed (a wave's 16 loads used to
  // be waited for in full before its ~500 vector instructions: ~10 us per sub-tile at four waves per SIMD)
  // Loads: clamped addresses (kernel-argument base + one 32-bit byte offset; the launcher guarantees B*F*4 < 2^32), issued
  // unconditionally - a conditional load is a branch each, and whatever waits for a register loaded before it waits for
  // everything in flight (the counters are in order); rows >= B and columns >= F are masked where the values are used.
  const unsigned rowB = (unsigned)F * 4u;
  float xn[RPL];
  {
    const int sub0 = min(blockIdx.x * kWaves + w, n_sub - 1);
    const unsigned colB = (unsigned)min((int64_t)sub0 * SUBF + l31, F - 1) * 4u;
#pragma unroll
    for (int q = 0; q < RPL; q++)
      xn[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + ((unsigned)min(RPL * h + q, B - 1) * rowB + colB));
  }
  for (int sub = blockIdx.x * kWaves + w; sub < n_sub; sub += gridDim.x * kWaves) {
    const int64_t col = (int64_t)sub * SUBF + l31;
    const bool cok = col < F;
    // this sub-tile's clamped (row, column) byte offsets are recomputed where they are used (2 instructions each)
    const unsigned colBc = (unsigned)min(col, F - 1) * 4u;
#define S1_OFF(q) ((unsigned)min(RPL * h + (q), B - 1) * rowB + colBc)
    float xr[RPL], tr[RPL], rr[RES ? 4 : 1];
#pragma unroll
    for (int q = 0; q < RPL; q++) xr[q] = xn[q];       // (non-temporal dword loads measured slower here: 72.6 vs 63.8 us)
    {
      const int subn = min(sub + (int)(gridDim.x * kWaves), n_sub - 1);       // (the last round re-reads its own sub-tile)
      const unsigned colB = (unsigned)min((int64_t)subn * SUBF + l31, F - 1) * 4u;
#pragma unroll
      for (int q = 0; q < RPL; q++)
        xn[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + ((unsigned)min(RPL * h + q, B - 1) * rowB + colB));
    }
    if (ab) {      // folded batch-norm (channels-last: channel = column mod nch, nch a power of two): x = a*z + b on load
      // lanes beyond F work on the CLAMPED column F - 1 (loads above, the x_q store below): they take ITS channel, so what they
      // store is the owning lane's value (round-3 advisor finding: F % 32 != 0 stored a wrong-channel value there)
      const int ch = (int)(min(col, F - 1) & (int64_t)(nch - 1));
      const float av = ab[ch], bv = ab[nch + ch];
#pragma unroll
      for (int q = 0; q < RPL; q++) xr[q] = __fmaf_rn(av, xr[q], bv);
    }
    // ---- transform + quantise; batch statistics: registers + one cross-half shuffle ---------------------------
    // (round 3: the same as ONE straight-line block over the 16 rows - selects instead of the per-row exec-mask branches, so
    //  that the rows' table reads and dependent chains overlap - takes 141 instead of 88 VGPRs, three waves per SIMD instead
    //  of four: 64.6-65.6 us against 61.6-62.4 at [28, 802816].  The kernel is bound by neither bytes nor issue slots: without
    //  the x_q store it runs 59-61 us, the x-only correlation 26 us; what it lives on is waves per SIMD.)
    // Order (round 3): x statistics -> stage x -> x Gram, THEN the transform in place (the row registers take t as soon as
    // x_q is stored), t statistics -> stage t -> t Gram: one 16-row array is live instead of two.
    float sx = 0.f;
#pragma unroll
    for (int q = 0; q < RPL; q++)
      if (RPL * h + q < B) sx += xr[q];
    sx += __shfl_xor(sx, 32, 64);
    const float mx = sx / (float)B;          // true division, like torch.mean: a constant column gives EXACTLY its value
    float vx = 0.f;
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      if (RPL * h + q < B) {
        const float d = xr[q] - mx;
        vx += d * d;
      }
    }
    vx += __shfl_xor(vx, 32, 64);
    const float rx = 1.0f / (sqrtf(vx * invBm1) + eps);
    if (stats && cok && h == 0) {
      stats[col] = mx;
      stats[F + col] = rx;
    }
    // ---- x operand: stage, Gram (negated when it is subtracted from the t Gram) --------------------------------
    {
      unsigned wd[RPL];
#pragma unroll
      for (int q = 0; q < RPL; q++) wd[q] = (cok && RPL * h + q < B) ? pack_hi_lo((xr[q] - mx) * rx) : 0u;
      stage_rows(W, l31, h, wd);
    }
    wave_lds_sync();
    if (PAIR && RES) {        // the first four shortcut rows: requested here, in flight under the x Gram; the rest one group ahead
#pragma unroll
      for (int q = 0; q < 4; q++) rr[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(res) + S1_OFF(q));
    }
    gram32<PAIR>(W, l31, h, acc);
    wave_lds_sync();
    if (PAIR) {
      // ---- transform + quantise (x_q stored, the row register takes t), t statistics, stage, Gram ------------------
      // Straight-line, four rows at a time (their table reads and dependent chains overlap; all sixteen at once cost 141
      // registers, a branch per row serialises them).  Rows >= B and columns >= F carry the clamped row's / column's values:
      // they are stored too - the same value to the same address as the lane that owns it - and masked out of the sums.
      float st = 0.f;
      unsigned mbits = 0u;         // rmask: this lane's 16 stored-output sign bits, row 16 h + q at bit 15 - q while they are collected
#pragma unroll
      for (int q4 = 0; q4 < RPL; q4 += 4) {
        float qq[4], rn[RES ? 4 : 1];
        if (RES && q4 + 4 < RPL) {
#pragma unroll
          for (int j = 0; j < 4; j++) rn[RES ? j : 0] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(res) + S1_OFF(q4 + 4 + j));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = q4 + j;
          float b;
          qq[j] = act_quant1<0, BND>(xr[q], k, nlev, r, &tr[q], &b, tab);
          if (RES) qq[j] += rr[RES ? j : 0];
          if (relu) qq[j] = fmaxf(qq[j], 0.0f);
          if (RPL * h + q >= B) tr[q] = 0.f;
          st += tr[q];
        }
        if (xq) {
#pragma unroll
          for (int j = 0; j < 4; j++) *reinterpret_cast<float*>(reinterpret_cast<char*>(xq) + S1_OFF(q4 + j)) = qq[j];
        }
        if (BND && rmask) {  // (launch-uniform; the bounded-quantiser form only: the generic one has no register left for it)
#pragma unroll
          for (int j = 0; j < 4; j++) mbits = mbits + mbits + (qq[j] > 0.0f ? 1u : 0u);      // (a compare and an add-with-carry per row)
        }
        if (RES && q4 + 4 < RPL) {
#pragma unroll
          for (int j = 0; j < 4; j++) rr[RES ? j : 0] = rn[RES ? j : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (BND && rmask) {
        // one bit per stored element for the backward's ReLU mask: word [sub-tile][feature], bit R = sign of row R (both row halves
        // of a feature in one word: the backward's lanes own rows of both)
        const unsigned mine = __builtin_bitreverse32(mbits) >> 16;                 // row 16 h + q at bit q
        const unsigned other = __builtin_amdgcn_permlane32_swap(mine, mine, false, false)[1];      // lanes < 32: the other half's (no LDS)
        if (h == 0) rmask[(int64_t)sub * 32 + l31] = mine | (other << 16);
      }
      st += __shfl_xor(st, 32, 64);
      const float mt = st / (float)B;
      float vt = 0.f;
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        if (RPL * h + q < B) { const float d2 = tr[q] - mt; vt += d2 * d2; }
      }
      vt += __shfl_xor(vt, 32, 64);
      const float rt = 1.0f / (sqrtf(vt * invBm1) + eps);
      if (stats && cok && h == 0) { stats[2 * F + col] = mt; stats[3 * F + col] = rt; }
      unsigned wd[RPL];
#pragma unroll
      for (int q = 0; q < RPL; q++) wd[q] = (cok && RPL * h + q < B) ? pack_hi_lo((tr[q] - mt) * rt) : 0u;
      stage_rows(W, l31, h, wd);
      wave_lds_sync();
      gram32<false>(W, l31, h, acc);
      wave_lds_sync();
    }
  }
#undef S1_OFF
  // ---- the only workgroup-wide step: add the four wave accumulators, write the [32][32] slab -------------------
  __syncthreads();
  float* Cw = reinterpret_cast<float*>(lds) + w * 1024;
#pragma unroll
  for (int e = 0; e < 16; e++) Cw[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + l31] = acc[e];
  __syncthreads();
  const float* C = reinterpret_cast<const float*>(lds);
  float* slab = slabs + (int64_t)blockIdx.x * 1024;
  for (int e = tid; e < 1024; e += kThreads1) slab[e] = C[e] + C[1024 + e] + C[2048 + e] + C[3072 + e];
}

// ================================================================================================ forward, 64-feature form (round 6)
// VERDICT r5 item 2a asked for >= 256-byte row segments without the registers that two feature columns per lane cost.  This form
// gets them by turning the wave's tile around: ONE LANE PER FEATURE COLUMN (64 features = 256-byte row segments for every load
// and store of the kernel), the lane holding ALL B rows of its column:
//   * no row is padded: 28 rows cost 28 rows (the 32-feature form computes 16 rows in both half-waves: 4 of 32 wasted at B = 28),
//     the batch statistics are plain register sums - no cross-half shuffle, no per-row "row < B" selects, no clamped addresses;
//   * the standardised columns go to a wave-private fp32 image [row][feature] with ONE ds_write_b32 per element; the Gram reads it
//     back in MFMA fragment order (lane = row, 8 consecutive features = two ds_read_b128) and only THERE splits into bf16 hi / lo -
//     on pairs (v_cvt_pk_bf16_f32), 3 vector instructions per value instead of ~10 for the packed (hi << 16 | lo) words plus their
//     unpacking;
//   * image layout: [32 rows][68 floats] (272-byte rows): the lane = feature accesses of one row touch 64 consecutive dwords
//     (conflict-free b32, the row a compile-time offset of the instruction), the lane = row b128 reads of one 16-byte feature
//     granule start (17 r + g) granules in: 16 distinct bank quads within each of the instruction's 16-lane groups
//     (MI355X_MICROARCH.md, LDS table);
//   * every row address is a scalar base (x + q * rowB) plus ONE vector offset (the lane's column): no vector add per access;
//   * nothing is prefetched: measured on one box (NOTES.md, round 6) a second register set for the next tile's x rows (3 waves per SIMD,
//     or 4 with spills), all 28 shortcut rows requested with the x rows (one wait per tile instead of seven) and both together are
//     equal or slower - the kernel sits at 82-86 % of what a LINEAR copy of the same 2 reads + 1 write reaches on this chip
//     (tools/src/rows_bw.hip: 5.2-5.4 TB/s), and a compute-free build of it is only 8 us faster.  An LDS-DMA landing pad for the next
//     tile (global_load_lds_dwordx4) was priced, not built: with the staging image it is 15 KiB per wave = 10 waves per CU.
// Preconditions (launcher): PAIR, bounded quantiser, B = 28 or 32 (straight-line code over the rows: a template parameter; with the
// row count a run-time value the register allocation degenerated); everything else runs the 32-feature form above.
constexpr int TF1 = 64;
constexpr int kRowB = 272;                 // bytes per image row: 64 floats + 16 B (17 granules: b128 reads down a column spread over all banks)
constexpr int kStageB = 32 * kRowB;        // bytes of one wave's image (rows >= B stay zero)

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t1 __attribute__((ext_vector_type(4)));

// 8 fp32 values -> their bf16 high parts and the bf16 of the remainders (element j of the vectors = value j)
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const f32x2_t two = {v[2 * q], v[2 * q + 1]};
    const unsigned pw = __builtin_bit_cast(unsigned, __builtin_convertvector(two, bf16x2_t));
    const float h0 = __uint_as_float(pw << 16), h1 = __uint_as_float(pw & 0xffff0000u);
    const f32x2_t rem = {v[2 * q] - h0, v[2 * q + 1] - h1};
    h[q] = pw;
    l[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(rem, bf16x2_t));
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}


// Gram of the staged 64 features: acc (+/-)= V V^T; lane = (row l & 31, feature group l >> 5), 4 K-steps of 16 features
template <bool NEG>
__device__ __forceinline__ void gram64(const char* __restrict__ Wr, f32x16& acc) {       // Wr: this lane's row, at its K-half
#pragma unroll
  for (int ks = 0; ks < TF1 / 16; ks++) {
    const f32x4_t1 a = *reinterpret_cast<const f32x4_t1*>(Wr + 64 * ks);
    const f32x4_t1 b = *reinterpret_cast<const f32x4_t1*>(Wr + 64 * ks + 16);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    bf16x8 hi, lo;
    split8(v, hi, lo);
    if (NEG) {          // - V V^T in three terms from hi, lo and ONE negated vector: (-hi) hi + (-hi) lo + lo (-hi)
      const bf16x8 nhi = neg8(hi);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nhi, hi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(nhi, lo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, nhi, acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, hi, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hi, lo, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(lo, hi, acc, 0, 0, 0);
    }
  }
}

template <bool RES, int NG>
__global__ __launch_bounds__(kThreads1, 4) void site1_fwd64_kernel(const float* __restrict__ x, int B, int64_t F, int k, float r,
                                                                   float eps, float* __restrict__ xq, float* __restrict__ slabs,
                                                                   float* __restrict__ stats, int n_sub32,
                                                                   unsigned* __restrict__ counter, const float* __restrict__ res,
                                                                   int relu, const float* __restrict__ ab, int nch,
                                                                   int64_t ws_gstride, unsigned* __restrict__ rmask) {
  {      // blockIdx.y = group, as in site1_fwd_kernel
    const int64_t gi = blockIdx.y, go = gi * (int64_t)B * F;
    x += go;
    if (xq) xq += go;
    if (res) res += go;
    if (stats) stats += gi * 4 * F;
    slabs += gi * ws_gstride;
    if (counter) counter += gi * ws_gstride;
    if (ab) ab += gi * 2 * nch;
    if (rmask) rmask += gi * (int64_t)n_sub32 * 32;
  }
  __shared__ __attribute__((aligned(16))) char img[kWaves * kStageB];
  __shared__ __attribute__((aligned(16))) float nerf_lds[ALIGNQ_NERF_LDS_FLOATS];
  nerf_tab_load(nerf_lds);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  char* Wb = img + w * kStageB;
  {      // rows >= B of the image stay zero for the whole launch (never written again)
    for (int j = lane * 16; j < kStageB; j += 1024) *reinterpret_cast<u32x4*>(Wb + j) = (u32x4){0u, 0u, 0u, 0u};
  }
  __syncthreads();
  const NerfTab tab = nerf_tab(nerf_lds);
  auto uni = [](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); };
  Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  nlev.n = uni(nlev.n);
  nlev.yn = uni(nlev.yn);
  const float fB = uni((float)B), invBm1 = uni(1.0f / (float)(B - 1));
  constexpr int ng = NG;                      // 4-row groups: B = 4 NG (a template parameter: straight-line code over the rows)
  if (blockIdx.x == 0 && tid == 0 && counter) *counter = 0u;
  char* Wl = Wb + lane * 4;                   // lane = feature: this lane's column (row q at q * kRowB)
  // Gram layout: row l & 31, features 8 h .. 8 h + 7 of every 16-feature K-step
  const int r31 = lane & 31, hh = lane >> 5;
  const char* Wr = Wb + r31 * kRowB + hh * 32;
  const unsigned rowB = (unsigned)F * 4u;
  const int n_tile = (int)((F + TF1 - 1) / TF1);

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.0f;

  for (int tile = blockIdx.x * kWaves + w; tile < n_tile; tile += gridDim.x * kWaves) {
    const int64_t col = (int64_t)tile * TF1 + lane;
    const bool cok = col < F;
    const unsigned colB = (unsigned)min(col, F - 1) * 4u;       // lanes beyond F work on the clamped column (masked below)
    // rows: a scalar base per row (x + q * rowB in scalar registers) + ONE vector offset for the lane's column - no vector add per access
#define S64_ROW(base, q) (reinterpret_cast<const char*>(base) + (size_t)(q) * rowB + colB)
    float xr[32];
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < ng) {
#pragma unroll
        for (int j = 0; j < 4; j++) xr[4 * g + j] = *reinterpret_cast<const float*>(S64_ROW(x, 4 * g + j));
      }
    }
    float rr[RES ? 4 : 1];
    if (RES) {        // the first four shortcut rows: in flight under the x statistics and the x Gram
#pragma unroll
      for (int j = 0; j < 4; j++) rr[RES ? j : 0] = *reinterpret_cast<const float*>(S64_ROW(res, j));
    }
    if (ab) {         // folded batch-norm (channels-last: channel = column mod nch): x = a*z + b on load
      const int ch = (int)(min(col, F - 1) & (int64_t)(nch - 1));
      const float av = ab[ch], bv = ab[nch + ch];
#pragma unroll
      for (int g = 0; g < 8; g++) {
        if (g < ng) {
#pragma unroll
          for (int j = 0; j < 4; j++) xr[4 * g + j] = __fmaf_rn(av, xr[4 * g + j], bv);
        }
      }
    }
    // ---- x statistics: register sums over the lane's own column -------------------------------------------------------------
    float sx = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < ng) {
#pragma unroll
        for (int j = 0; j < 4; j++) sx += xr[4 * g + j];
      }
    }
    const float mx = sx / fB;                 // true division, like torch.mean
    float vx = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < ng) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float d = xr[4 * g + j] - mx;
          vx += d * d;
        }
      }
    }
    const float rx = 1.0f / (sqrtf(vx * invBm1) + eps);
    if (stats && cok) {
      stats[col] = mx;
      stats[F + col] = rx;
    }
    {
      // xhat = x * r - m * r as ONE fma per element (|error| ~ 6e-8 |x r|: far below the 1e-5 bar on D); columns beyond F contribute zeros
      const float rxm = cok ? rx : 0.0f, nmx = -mx * rxm;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        if (g < ng) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            *reinterpret_cast<float*>(Wl + (4 * g + j) * kRowB) = __fmaf_rn(xr[4 * g + j], rxm, nmx);
        }
      }
    }
    wave_lds_sync();
    gram64<true>(Wr, acc);
    wave_lds_sync();
    // ---- transform + quantise, four rows at a time; the row register takes t; x_q (+ shortcut, ReLU) leaves as 256-byte segments --
    float st = 0.f;
    unsigned mbits = 0u;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < ng) {
        float qq[4], rn[RES ? 4 : 1];
        if (RES && g + 1 < ng) {
#pragma unroll
          for (int j = 0; j < 4; j++) rn[RES ? j : 0] = *reinterpret_cast<const float*>(S64_ROW(res, 4 * g + 4 + j));
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = 4 * g + j;
          float b, t;
          qq[j] = act_quant1<0, true>(xr[q], k, nlev, r, &t, &b, tab);
          xr[q] = t;
          if (RES) qq[j] += rr[RES ? j : 0];
          if (relu) qq[j] = fmaxf(qq[j], 0.0f);
          st += t;
        }
        if (xq) {
#pragma unroll
          for (int j = 0; j < 4; j++) *reinterpret_cast<float*>(const_cast<char*>(S64_ROW(xq, 4 * g + j))) = qq[j];
        }
        if (rmask) {
#pragma unroll
          for (int j = 0; j < 4; j++) mbits = mbits + mbits + (qq[j] > 0.0f ? 1u : 0u);
        }
        if (RES && g + 1 < ng) {
#pragma unroll
          for (int j = 0; j < 4; j++) rr[RES ? j : 0] = rn[RES ? j : 0];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // bit R = [stored row R > 0]; bits B..31 repeat row B - 1 (arithmetic shift): the backward's lanes that own rows >= B work on the
    // clamped row B - 1 and must see ITS bit (they store its dres value to its address)
    if (rmask && cok) rmask[col] = (unsigned)((int)__builtin_bitreverse32(mbits) >> (32 - B));
    const float mt = st / fB;
    float vt = 0.f;
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < ng) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float d2 = xr[4 * g + j] - mt;
          vt += d2 * d2;
        }
      }
    }
    const float rt = 1.0f / (sqrtf(vt * invBm1) + eps);
    if (stats && cok) {
      stats[2 * F + col] = mt;
      stats[3 * F + col] = rt;
    }
    {
      const float rtm = cok ? rt : 0.0f, nmt = -mt * rtm;
#pragma unroll
      for (int g = 0; g < 8; g++) {
        if (g < ng) {
#pragma unroll
          for (int j = 0; j < 4; j++)
            *reinterpret_cast<float*>(Wl + (4 * g + j) * kRowB) = __fmaf_rn(xr[4 * g + j], rtm, nmt);
        }
      }
    }
    wave_lds_sync();
    gram64<false>(Wr, acc);
    wave_lds_sync();
#undef S64_ROW
  }
  // ---- the only workgroup-wide step: add the four wave accumulators, write the [32][32] slab (as site1_fwd_kernel) -----------
  __syncthreads();
  float* Cw = reinterpret_cast<float*>(img) + w * 1024;
#pragma unroll
  for (int e = 0; e < 16; e++) Cw[((e & 3) + 8 * (e >> 2) + 4 * hh) * 32 + r31] = acc[e];
  __syncthreads();
  const float* C = reinterpret_cast<const float*>(img);
  float* slab = slabs + (int64_t)blockIdx.x * 1024;
  for (int e = tid; e < 1024; e += kThreads1) slab[e] = C[e] + C[1024 + e] + C[2048 + e] + C[3072 + e];
}

// ================================================================================================ backward
// site1_bwd_kernel (rewritten in round 4): S * Vh on the FP32 matrix instruction v_mfma_f32_32x32x2_f32, operands and results in
// registers only.  Round 3's kernel is vector-ALU bound (~1500 vector instructions per 32-feature sub-tile and wave: 61 of its
// 68 us at [28, 802816] are issue slots); a fifth of them split every standardised value into bf16 hi / lo halves, staged the
// packed words through LDS into MFMA fragment order, unpacked them, and carried the 32x32 result back through LDS into the
// lanes' row layout.  None of that is needed when a lane owns the ROWS the accumulator hands it:
//   * lane (feature f = l & 31, half h = l >> 5) owns rows R(q, h) = (q & 3) + 8 (q >> 2) + 4 h, q = 0..15 - exactly the rows of
//     accumulator element q of v_mfma_f32_32x32x* in that lane (loads and stores address these rows; a wave instruction still
//     reads two 128-byte row segments);
//   * D[i][f] = sum_k S[i][k] Vh[k][f]: step m of 16 contracts k = R(m, half): the B operand of lane (f, h) is ITS OWN register
//     Vh[R(m, h)][f], the A operand is S[i = l & 31][R(m, l >> 5)] - sixteen registers loaded once per launch (fp32: no split);
//     rows >= B contribute nothing because S is zero there (the clamped loads keep every operand finite);
//   * the result element q of lane (f, h) is D[R(q, h)][f]: the lane's own rows - no transposition.
// 32 fp32 MFMAs per sub-tile (2048 matrix-pipe cycles per SIMD and sub-tile against ~4600 vector cycles): the matrix pipe is
// not the limit, and exact fp32 products replace the 3-term split.  No LDS at all.
__device__ __forceinline__ int row_of(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

template <bool PAIR>
__global__ __launch_bounds__(kThreads1, PAIR ? 3 : 5) void site1_bwd_kernel(const float* __restrict__ gup, const float* __restrict__ S,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ stats, int B, int64_t F, float r,
                                                               float eps, float* __restrict__ dx, int n_sub,
                                                               const float* __restrict__ ab, int C,
                                                               const float* __restrict__ ymask, float* __restrict__ dres,
                                                               int64_t s_gstride, const float* __restrict__ gup2,
                                                               const float* __restrict__ save = nullptr, float* __restrict__ colsum = nullptr,
                                                               const unsigned* __restrict__ rmask = nullptr) {
  {        // blockIdx.y = group (see site1_fwd_kernel); S matrices s_gstride floats apart
    const int64_t gi = blockIdx.y, go = gi * (int64_t)B * F;
    x += go; dx += go;
    if (gup) gup += go;
    if (gup2) gup2 += go;
    if (ymask) ymask += go;
    if (dres) dres += go;
    S += gi * s_gstride;
    stats += gi * 4 * F;
    if (ab) ab += gi * 2 * C;
    if (save) save += gi * 2 * C;
    if (colsum) colsum += gi * F;               // [2][groups][F]: the second array starts gridDim.y * F floats further
    if (rmask) rmask += gi * (int64_t)n_sub * 32;
  }
  const int64_t cs2 = (int64_t)gridDim.y * F;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const float invB = 1.0f / (float)B, invBm1 = 1.0f / (float)(B - 1);
  const float rjac = r * ALIGNQ_TWO_OVER_SQRT_2PI;
  // A operands: S[i = l31][k = R(m, h)], zero outside the batch (clamped addresses, all requests first)
  float sA[RPL];
#pragma unroll
  for (int m = 0; m < RPL; m++) sA[m] = S[min(l31, B - 1) * B + min(row_of(m, h), B - 1)];
#pragma unroll
  for (int m = 0; m < RPL; m++) {
    asm volatile("" : "+v"(sA[m]));
    sA[m] = (l31 < B && row_of(m, h) < B) ? sA[m] : 0.0f;
  }
  const unsigned rowB = (unsigned)F * 4u;
  const bool has_g = PAIR && gup;
  // the upstream gradient is needed in the last loop of a sub-tile only: each lane parks its 16 rows in LDS meanwhile ([4][thread]
  // quads: conflict-free 16-byte accesses, a slot of its own, no synchronisation) - with it the kernel fits 128 registers = four
  // waves per SIMD
  __shared__ __attribute__((aligned(16))) float4 g_park[PAIR ? 4 * kThreads1 : 1];
  for (int sub = blockIdx.x * kWaves + w; sub < n_sub; sub += gridDim.x * kWaves) {
    const int64_t col = (int64_t)sub * SUBF + l31;
    const bool cok = col < F;
    const int64_t colc = cok ? col : F - 1;            // loads: clamped addresses, unconditional, values selected afterwards
    const unsigned colB = (unsigned)colc * 4u;
#define S2_OFF(q) ((unsigned)min(row_of(q, h), B - 1) * rowB + colB)
    float xr[RPL], gr[RPL], out[RPL];
    const char* gsrc = reinterpret_cast<const char*>(has_g ? gup : x);   // (no upstream gradient: x stands in, values dropped)
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      xr[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + S2_OFF(q));
      gr[q] = *reinterpret_cast<const float*>(gsrc + S2_OFF(q));
    }
    if (PAIR && gup2) {      // the upstream gradient as two addends (fused.GradFork): g + g2, the sum autograd would have formed
      float g2[RPL];
#pragma unroll
      for (int q = 0; q < RPL; q++) g2[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(gup2) + S2_OFF(q));
#pragma unroll
      for (int q = 0; q < RPL; q++) gr[q] += g2[q];
    }
    const float mx_l = stats[colc], rx_l = stats[F + colc];
    const float mt_l = PAIR ? stats[2 * F + colc] : 0.f, rt_l = PAIR ? stats[3 * F + colc] : 0.f;
    float av = 1.0f, bv = 0.0f;
    if (ab) {      // folded batch-norm: x = a*z + b on load (dx is the gradient w.r.t. x; alignq_bnq_bwd_dx takes it to z)
      const int ch = (int)(colc & (int64_t)(C - 1));
      av = ab[ch];
      bv = ab[C + ch];
    }
    if (PAIR && rmask) {
      // fused ReLU backward from the forward's ONE-BIT mask (round 5: 4 bytes per element of y were read for its sign): one word
      // per feature of the sub-tile, bit R = row R.  Rows >= B carry row B - 1's bits (the forward's clamped rows hold its values).
      const unsigned mw = rmask[(int64_t)sub * 32 + l31] >> (4 * h);       // bit R(q, 0) of it = row R(q, h) of this lane's feature
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const bool pos = (mw & (1u << row_of(q, 0))) != 0u;
        gr[q] = pos ? gr[q] : 0.0f;
        if (dres) *reinterpret_cast<float*>(reinterpret_cast<char*>(dres) + S2_OFF(q)) = has_g ? gr[q] : 0.0f;
      }
    } else if (PAIR && ymask) {
      // fused ReLU backward: the mask from the forward's output; the masked gradient is also the shortcut's gradient (dres).
      // Rows >= B / columns >= F carry the clamped element's values and store them to its address (the owning lane's value).
      float yr[RPL];
#pragma unroll
      for (int q = 0; q < RPL; q++) yr[q] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(ymask) + S2_OFF(q));
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        gr[q] = yr[q] > 0.0f ? gr[q] : 0.0f;
        if (dres) *reinterpret_cast<float*>(reinterpret_cast<char*>(dres) + S2_OFF(q)) = has_g ? gr[q] : 0.0f;
      }
    }
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      if (ab) xr[q] = __fmaf_rn(av, xr[q], bv);
      gr[q] = has_g ? gr[q] : 0.0f;
    }
    if (PAIR) {
#pragma unroll
      for (int q4 = 0; q4 < RPL / 4; q4++)
        g_park[q4 * kThreads1 + tid] = make_float4(gr[4 * q4], gr[4 * q4 + 1], gr[4 * q4 + 2], gr[4 * q4 + 3]);
    }
    const float mx = mx_l, rx = rx_l, mt = mt_l, rt = rt_l;        // (columns >= F: the clamped column's, results never stored)
    float kap_x = 1.0f, kap_t = 1.0f;       // (sd+eps)/sd = 1/(1-eps*rho); torch's std backward is 0 where sd == 0
    if (eps != 0.0f) {
      const float dxn = 1.0f - eps * rx, dtn = 1.0f - eps * rt;
      kap_x = (dxn > 1e-12f) ? 1.0f / dxn : 0.0f;
      kap_t = (dtn > 1e-12f) ? 1.0f / dtn : 0.0f;
    }
    // ---- x operand: S * Xh, operands and result in registers ----------------------------------------------------
    f32x16 ax;
#pragma unroll
    for (int e = 0; e < 16; e++) ax[e] = 0.f;
#pragma unroll
    for (int m = 0; m < RPL; m++) ax = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[m], (xr[m] - mx) * rx, ax, 0, 0, 0);
    // ---- t operand's transform: vector-ALU work while the matrix pipe contracts the x operand ---------------------------
    float th[PAIR ? RPL : 1];
    if (PAIR) {
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        float t, jac;
        act_transform_rcp(xr[q], r, rjac, &t, &jac);        // (jac is recomputed below: 16 registers for ~5 instructions)
        th[q] = (t - mt) * rt;
      }
    }
    {
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        if (row_of(q, h) < B) { s0 += ax[q]; s1 += ax[q] * ((xr[q] - mx) * rx); }
      }
      s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean_d = s0 * invB, proj = s1 * invBm1 * kap_x;
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const float cx = rx * (ax[q] - mean_d - ((xr[q] - mx) * rx) * proj);
        out[q] = PAIR ? -cx : cx;          // corr(x,x) enters D with a minus sign
      }
    }
    if (PAIR) {
      f32x16 at;                            // (the x operand's accumulator is dead: the same registers)
#pragma unroll
      for (int e = 0; e < 16; e++) at[e] = 0.f;
#pragma unroll
      for (int m = 0; m < RPL; m++) at = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[m], th[PAIR ? m : 0], at, 0, 0, 0);
      float s0 = 0.f, s1 = 0.f;
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        if (row_of(q, h) < B) { s0 += at[q]; s1 += at[q] * th[q]; }
      }
      s0 += __shfl_xor(s0, 32, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mean_d = s0 * invB, proj = s1 * invBm1 * kap_t;
#pragma unroll
      for (int q4 = 0; q4 < RPL / 4; q4++) {
        const float4 g4 = g_park[q4 * kThreads1 + tid];
        const float ge[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int q = 4 * q4 + j;
          const float ct = rt * (at[q] - mean_d - th[q] * proj);
          const float xv = xr[q];
          out[q] += (ge[j] + ct) * (rjac * __builtin_amdgcn_exp2f(xv * xv * -0.72134752044448170368f));
        }
      }
    }
    if (cok) {
#pragma unroll
      for (int q = 0; q < RPL; q++)
        if (row_of(q, h) < B)
          *reinterpret_cast<float*>(reinterpret_cast<char*>(dx) + ((unsigned)row_of(q, h) * rowB + colB)) = out[q];
    }
    if (PAIR && colsum) {
      // The folded batch-norm's backward needs sum dx and sum dx * zhat per channel over batch AND pixels: the batch part is formed
      // here, per feature column, from the registers that still hold this sub-tile's dx and x (zhat = (z - mean) * invstd =
      // (x - beta) / gamma up to the rounding of x = a*z + b; an ill-conditioned channel (alignq_bn_col_ill: gamma == 0 or
      // |gamma| < 1e-2 |beta|) leaves 0 and bnq_finalize_bwd_kernel forms it from dx and z itself): 8 bytes per column instead of alignq_bnq_bwd_dx's own pass over dx and z (8 B per ELEMENT).  A pass
      // of its own behind the stores and scheduling barriers, a and the statistics re-read through pointers the optimiser cannot
      // match with the earlier ones: inside the loop above (the kernel's register peak) every form of it spilled (NOTES.md 5g).
      __builtin_amdgcn_sched_barrier(0);
      const float* ab2 = ab;
      const float* sv2 = save;
      asm volatile("" : "+s"(ab2), "+s"(sv2));
      const int ch = (int)(colc & (int64_t)(C - 1));
      const float a2 = ab2[ch];
      const float bt = __fmaf_rn(sv2[ch], a2, ab2[C + ch]);
      // (an ill-conditioned channel - gamma == 0 or |gamma| << |beta| - contributes 0 here: bnq_finalize_bwd_kernel sums it itself)
      const float rg = alignq_bn_col_ill(a2, ab2[C + ch], sv2[ch], sv2[C + ch]) ? 0.0f : sv2[C + ch] / a2;
      float f0 = 0.f, f1 = 0.f;
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const float o = row_of(q, h) < B ? out[q] : 0.0f;
        f0 += o;
        f1 = __fmaf_rn(o, (xr[q] - bt) * rg, f1);
      }
      f0 += __shfl_xor(f0, 32, 64);
      f1 += __shfl_xor(f1, 32, 64);
      if (h == 0 && cok) { colsum[col] = f0; colsum[cs2 + col] = f1; }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef S2_OFF
  }
}

#define RET_ON_ERR1()                                 \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)

}  // namespace

int launch_partials1(bool pair, const Geom& g, const float* x, int B, int64_t F, int k, float r, float eps, float* xq,
                     float* stats, float* ws, hipStream_t st, const float* res, int relu, const float* ab, int C, int groups,
                     int64_t ws_gstride, unsigned* rmask) {
  unsigned* counter = reinterpret_cast<unsigned*>(ws + (size_t)g.grid * g.slab_floats + kPartFloats);
  if ((int64_t)B * F * 4 >= ((int64_t)1 << 32)) return ALIGNQ_EUNSUPPORTED;   // 32-bit byte offsets
  const int n_sub = (int)((F + SUBF - 1) / SUBF);
  const bool bnd = make_levels(k, fabsf(r) <= 8.0f).yn != 0.0f;       // as the kernel forms its Levels
  if (rmask && !(pair && bnd)) return ALIGNQ_EUNSUPPORTED;            // (k == 1 / k == 32 / a huge act_range: the caller keeps y for the mask)
#ifndef ALIGNQ_S1_FWD32_ONLY
  if (pair && bnd && (B == 28 || B == 32)) {       // round 6: one lane per feature column, 256-byte row segments (Office batch 28; 32)
#define S1_L64(RESV, NGV) hipLaunchKernelGGL((site1_fwd64_kernel<RESV, NGV>), dim3(g.grid, groups), kThreads1, 0, st, x, B, F, k, r, eps, xq, ws, stats, n_sub, counter, res, relu, ab, C, ws_gstride, rmask)
    if (res && B == 28) S1_L64(true, 7);
    else if (res) S1_L64(true, 8);
    else if (B == 28) S1_L64(false, 7);
    else S1_L64(false, 8);
#undef S1_L64
    RET_ON_ERR1();
    return 0;
  }
#endif
  if (pair && res && bnd) hipLaunchKernelGGL((site1_fwd_kernel<true, true, true>), dim3(g.grid, groups), kThreads1, 0, st, x, B, F, k, r, eps, xq, ws, stats, n_sub, counter, res, relu, ab, C, ws_gstride, rmask);
  else if (pair && res) hipLaunchKernelGGL((site1_fwd_kernel<true, true>), dim3(g.grid, groups), kThreads1, 0, st, x, B, F, k, r, eps, xq, ws, stats, n_sub, counter, res, relu, ab, C, ws_gstride, rmask);
  else if (pair && bnd) hipLaunchKernelGGL((site1_fwd_kernel<true, false, true>), dim3(g.grid, groups), kThreads1, 0, st, x, B, F, k, r, eps, xq, ws, stats, n_sub, counter, nullptr, relu, ab, C, ws_gstride, rmask);
  else if (pair) hipLaunchKernelGGL((site1_fwd_kernel<true, false>), dim3(g.grid, groups), kThreads1, 0, st, x, B, F, k, r, eps, xq, ws, stats, n_sub, counter, nullptr, relu, ab, C, ws_gstride, rmask);
  else hipLaunchKernelGGL((site1_fwd_kernel<false, false>), dim3(g.grid, groups), kThreads1, 0, st, x, B, F, k, r, eps, xq, ws, stats, n_sub, counter, nullptr, 0, ab, C, ws_gstride);
  RET_ON_ERR1();
  return 0;
}

int launch_bwd1(bool pair, const float* gup, const float* S, const float* x, const float* stats, int B, int64_t F,
                float r, float eps, float* dx, hipStream_t st, const float* ab, int C, const float* ymask, float* dres, int groups,
                int64_t s_gstride, const float* gup2, const float* save, float* colsum, const unsigned* rmask) {
  if (gup2 && (!gup || !pair)) return ALIGNQ_EINVAL;
  if (colsum && (!pair || !ab || !save)) return ALIGNQ_EINVAL;
  if ((int64_t)B * F * 4 >= ((int64_t)1 << 32)) return ALIGNQ_EUNSUPPORTED;   // 32-bit byte offsets
  const int n_sub = (int)((F + SUBF - 1) / SUBF);
  int grid = (n_sub + kWaves - 1) / kWaves;
  // three 4-wave workgroups per CU (167 VGPRs) = 768 resident; a grid of exactly that (every wave loops over ~8 sub-tiles at
  // [28, 802816]) ran 82-84 us against 86-87 for 2048 and 17.2 against 19.6 at [28, 100352] (tools/s1_grid_sweep.sh)
  constexpr int capb = 768;
  if (grid > capb) grid = capb;
  if (groups > 1 && grid * groups > capb) grid = (capb + groups - 1) / groups;      // the groups share the resident round
  if (pair) hipLaunchKernelGGL((site1_bwd_kernel<true>), dim3(grid, groups), kThreads1, 0, st, gup, S, x, stats, B, F, r, eps, dx, n_sub, ab, C, ymask, dres, s_gstride, gup2, save, colsum, rmask);
  else hipLaunchKernelGGL((site1_bwd_kernel<false>), dim3(grid, groups), kThreads1, 0, st, gup, S, x, stats, B, F, r, eps, dx, n_sub, ab, C, nullptr, nullptr, s_gstride, nullptr, nullptr, nullptr);
  RET_ON_ERR1();
  return 0;
}

}  // namespace alignq_site
