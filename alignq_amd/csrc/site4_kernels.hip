// site4_kernels.hip — the ADMM-site kernels for batches of 65..128 rows (the reference's train batch 128 and
// eval batch 100) on gfx950, plus the slab reduction with its ADMM-loss epilogue and the backward "prep" kernel.
//
// Why a second set of kernels: at B=128 the Gram pair of one 64-feature tile is 1024 v_mfma_f32_32x32x2_f32
// (6.8 us of one CU's matrix pipes) and the CDF transform is ~90 VALU ops per element; a 256-thread workgroup
// (one wave per SIMD) runs load -> erf -> statistics -> LDS -> MFMA -> store as one serial chain (measured
// 25 us forward / 52 us backward per launch).  Here a workgroup is 16 waves (4 per SIMD):
//   forward : D is symmetric, so only the 10 upper-triangular 32x32 tiles are computed; the 20 work items
//             (tile, K-half) are dealt to the 16 waves so that every SIMD gets exactly 5 items whatever the
//             wave->SIMD rotation is (waves 0-3 take two items); the two K-halves of a tile are combined in
//             LDS in a fixed order (deterministic) and the slab shrinks from 64 KB to 40 KB;
//             small sites use 32- or 16-feature tiles so that all 256 CUs still get a tile.
//   backward: 16 units (row block, operand, column block) -> one 32x32 accumulator per wave, S = sym(dD)*c as
//             register-resident MFMA A fragments (prepared once per site by site_prep_kernel), the activation
//             transform re-evaluated with a cheap 1.5e-7 erf (tolerance-checked, not bit-checked), output
//             assembled in LDS and written back as full 256-byte rows.
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "site_internal.h"
#include "head_body.h"

using namespace alignq;

namespace alignq_site {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 1024;

// Memory order of the arrival ticket of the last-workgroup epilogues (slab reduction + ADMM loss here, the head's batch mean
// in head_kernels.hip).  Product: relaxed - the partials are write-through (sc1) stores drained with s_waitcnt vmcnt(0) by the
// ONE lane that also takes the ticket, the last workgroup reads them with sc1 loads behind a workgroup barrier: the form
// MI355X_MICROARCH.md lists as measured-valid (its hand-off table, row 1).  -DALIGNQ_TICKET_ACQREL builds the formally
// ordered variant (release fence = write-back of the XCD's dirty L2 lines before the ticket, acquire after it) for the A/B
// measurement in NOTES.md.
#ifdef ALIGNQ_TICKET_ACQREL
#define ALIGNQ_TICKET_ORDER __ATOMIC_ACQ_REL
#else
#define ALIGNQ_TICKET_ORDER __ATOMIC_RELAXED
#endif

// Diagnostic build only (-DALIGNQ_STAMPS, never shipped): wall-clock stamps (100 MHz) of workgroup 0 at phase
// boundaries, written to a buffer no kernel reads.
#ifdef ALIGNQ_STAMPS
__device__ unsigned long long g_stamps[64];
#define STAMP(i)                                                             \
  do {                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_stamps[i] = wall_clock64();   \
  } while (0)
// every workgroup's entry / exit time: [kernel (0 fwd, 1 bwd)][entry, exit][block]
__device__ unsigned long long g_blk[2][2][2048];
#define BSTAMP(kern, which)                                                                        \
  do {                                                                                             \
    if (threadIdx.x == 0 && blockIdx.x < 2048) g_blk[kern][which][blockIdx.x] = wall_clock64();    \
  } while (0)
#else
#define STAMP(i) do { } while (0)
#define BSTAMP(kern, which) do { } while (0)
#endif
#ifdef ALIGNQ_DIAG_DUMP
#define ALIGNQ_DIAG_DUMP_BITS (ALIGNQ_DIAG_DUMP)
// diagnostic build (round 6, tools/diag_twin_dump.py): intermediates of the 32-feature backward twin launch, by stage
__device__ float* g_dump = nullptr;       // [4 stages][2 sites][256 tiles][128 rows][32 columns][2]
#define DUMP2(stage, site, tile, row, col, v0, v1)                                                                        \
  do {                                                                                                                    \
    if (((((ALIGNQ_DIAG_DUMP) >> (stage)) & 1) || ((stage) == 0 && ((ALIGNQ_DIAG_DUMP) & 16)) || ((stage) == 2 && ((ALIGNQ_DIAG_DUMP) & 32)) || ((stage) == 3 && ((ALIGNQ_DIAG_DUMP) & 64))) && g_dump) { \
      float* p__ = g_dump + ((((((size_t)(stage) * 2 + (site)) * 256 + (tile)) * 128 + (row)) * 32 + (col)) * 2);         \
      p__[0] = (v0); p__[1] = (v1);                                                                                       \
    }                                                                                                                     \
  } while (0)
#else
#define ALIGNQ_DIAG_DUMP_BITS 0
#define DUMP2(stage, site, tile, row, col, v0, v1) do { } while (0)
#endif


// Addressing of the forward kernel: kernel-argument base (SGPR pair) + ONE unsigned 32-bit element offset per (row, column
// quad), shared by x / residual / x_q (global_load / global_store saddr + voffset form: no 64-bit address arithmetic in
// VGPRs; the launcher guarantees B*F*4 < 2^32).
// Loads are UNCONDITIONAL on clamped addresses (offset 0 stands in for anything outside the tensor) and the values are selected
// afterwards: a conditional load is a branch, and the compiler must assume it was not issued - so the next use of a register
// loaded BEFORE it waits for everything in flight (vmcnt counts in order).  Both paths of the launch-uniform `aligned` branch
// issue a fixed number of loads, so that the waits stay exact behind it.
__device__ __forceinline__ float4 ld4(const float* __restrict__ x, unsigned off, int col, int64_t F, bool row_ok,
                                      bool aligned) {
  const char* xb = reinterpret_cast<const char*>(x);
  float4 v;
  if (aligned) {
    const bool ok = row_ok && col < F;
    v = *reinterpret_cast<const float4*>(xb + (ok ? 4u * off : 0u));
    if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    const bool o0 = row_ok && col + 0 < F, o1 = row_ok && col + 1 < F, o2 = row_ok && col + 2 < F, o3 = row_ok && col + 3 < F;
    v.x = *reinterpret_cast<const float*>(xb + (o0 ? 4u * off : 0u));
    v.y = *reinterpret_cast<const float*>(xb + (o1 ? 4u * off + 4u : 0u));
    v.z = *reinterpret_cast<const float*>(xb + (o2 ? 4u * off + 8u : 0u));
    v.w = *reinterpret_cast<const float*>(xb + (o3 ? 4u * off + 12u : 0u));
    v.x = o0 ? v.x : 0.f; v.y = o1 ? v.y : 0.f; v.z = o2 ? v.z : 0.f; v.w = o3 ? v.w : 0.f;
  }
  return v;
}

// the same for an input that is read exactly once (the forward's x): non-temporal, so the stream does not displace the
// lines written for the consumer (tools/src/stream_bw.hip: +25 % on a 1:1 read/write stream)
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* __restrict__ x, unsigned off, int col, int64_t F, bool row_ok,
                                             bool aligned) {
  if (!aligned) return ld4(x, off, col, F, row_ok, aligned);
  const bool ok = row_ok && col < F;
  const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(reinterpret_cast<const char*>(x) + (ok ? 4u * off : 0u)));
  return ok ? make_float4(t.x, t.y, t.z, t.w) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ void st4(float* __restrict__ y, unsigned off, int col, int64_t F, bool row_ok, bool aligned,
                                    float4 v) {
  if (!row_ok) return;
  if (aligned) {
    if (col < F) *reinterpret_cast<float4*>(reinterpret_cast<char*>(y) + 4u * off) = v;
  } else {
    char* yb = reinterpret_cast<char*>(y);
    if (col + 0 < F) *reinterpret_cast<float*>(yb + 4u * off) = v.x;
    if (col + 1 < F) *reinterpret_cast<float*>(yb + 4u * off + 4u) = v.y;
    if (col + 2 < F) *reinterpret_cast<float*>(yb + 4u * off + 8u) = v.z;
    if (col + 3 < F) *reinterpret_cast<float*>(yb + 4u * off + 12u) = v.w;
  }
}

// upper-triangular tile index -> (I, J), I <= J, order (0,0)(0,1)(0,2)(0,3)(1,1)(1,2)(1,3)(2,2)(2,3)(3,3)
__device__ __forceinline__ void tile_ij(int tile, int& I, int& J) {
  int t = tile;
  I = 0;
  while (t >= 4 - I) { t -= 4 - I; I++; }
  J = I + t;
}

// ================================================================================================ forward
// Split-bf16 Gram: every standardised value v is stored as hi = bf16(v), lo = bf16(v - hi) and a product u*v is
// evaluated as hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  bf16 products are exact
// in fp32, the dropped lo*lo term is < 2^-16 relative and the representation error of hi+lo is 2^-17: the Gram
// difference D comes out within 6e-7 of an fp64 evaluation (plain fp32 sgemm: 2e-7; tolerance 1e-5) while the
// matrix pipe does 16x the K per instruction at half the cycles of v_mfma_f32_32x32x2_f32 (3 instead of 8 MFMAs
// per 16 features, 32 instead of 64 cycles each).  Measured MFMA phase at F=16384: 6.0 us -> see DESIGN.md §4.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// flip the sign of 8 packed bf16 (4 VGPRs, 4 v_xor): lets T-Gram minus X-Gram share ONE accumulator
__device__ __forceinline__ bf16x8 neg8(bf16x8 v) {
  u32x4 u = __builtin_bit_cast(u32x4, v);
  u ^= (u32x4){0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
  return __builtin_bit_cast(bf16x8, u);
}

// component e of a float4 (e is a compile-time constant after unrolling: folds to the register, no scratch array)
__device__ __forceinline__ float f4get(const float4& v, int e) { return e == 0 ? v.x : (e == 1 ? v.y : (e == 2 ? v.z : v.w)); }

// Sum per-lane values over the row groups of a wave (lanes with equal lane % LPR) on the VALU, no LDS crossbar: the two
// cross-row levels use v_permlane32_swap / v_permlane16_swap, which exchange halves / odd-even rows of TWO registers at once, so
// one swap + one add reduces two values and leaves each in half of the lanes (a reduce-scatter): 8 values -> 4 -> 2 registers
// with 6 swaps + 6 adds.  Result: in DPP row r (lanes 16r..16r+15), t0 holds the total of v[kRowVal[r]] and t1 that of
// v[4 + kRowVal[r]], kRowVal = {0, 2, 1, 3}; levels inside a row (LPR < 16) are row_ror all-reduces.
__device__ __forceinline__ float swap_add32(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);     // lanes < 32: a(l) + a(l+32); lanes >= 32: b(l-32) + b(l)
}
__device__ __forceinline__ float swap_add16(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);     // rows 0, 2: a(r) + a(r+1); rows 1, 3: b(r-1) + b(r)
}
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int LPR, bool PAIR>
__device__ __forceinline__ void rowgroup_sums(const float (&vx)[4], const float (&vt)[4], float& t0, float& t1) {
  t0 = swap_add16(swap_add32(vx[0], vx[1]), swap_add32(vx[2], vx[3]));
  t1 = PAIR ? swap_add16(swap_add32(vt[0], vt[1]), swap_add32(vt[2], vt[3])) : 0.0f;
  if (LPR <= 4) { t0 = dpp_add<0x124>(t0); if (PAIR) t1 = dpp_add<0x124>(t1); }      // row_ror:4
  if (LPR <= 8) { t0 = dpp_add<0x128>(t0); if (PAIR) t1 = dpp_add<0x128>(t1); }      // row_ror:8
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
// LDS byte address of a __shared__ object as a VECTOR register the optimiser cannot see through: accesses at base + constant
// then use the instruction's 16-bit offset field.  (Left to itself the compiler folds the region's own offset - beyond 64 KB for
// the backward's fp32 tiles - into every constant, materialises one address register per access and hoists them all out of the
// tile loop: 64 registers, spilled.)
__device__ __forceinline__ uint32_t lds_base(const void* p) {
  uint32_t a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
  asm volatile("" : "+v"(a));
  return a;
}
// site_bwd4's transposed staging: batch-row XOR of feature column `col` (multiples of 8 only: the 8-row MFMA fragments and the
// 4-row chunks stay contiguous).  Modelled against the LDS banking rules of MI355X_MICROARCH.md, plain [column][row] costs
// 32 / 4 / 4 LDS cycles per staging write / MFMA operand read / projection read (704 per wave and tile), mode 1 16 / 8 / 4 (576),
// mode 3 (found by enumeration) 8 / 8 / 4 (448) - and on the GPU all six modes run the [128, 524288] backward in 242-256 us,
// inside the run-to-run spread: the staging conflicts are not on the critical path (PMC: 48 % of the LDS-active cycles are
// conflict cycles, but waves wait on LDS for 3 % of their time).  Mode 1 stays; -DALIGNQ_XSWZ_MODE=n builds the others.
#ifndef ALIGNQ_XSWZ_MODE
#define ALIGNQ_XSWZ_MODE 1
#endif
#if ALIGNQ_XSWZ_MODE == 0
#define XSWZ(col) 0
#elif ALIGNQ_XSWZ_MODE == 1
#define XSWZ(col) ((((col) >> 4) & 1) << 4)
#elif ALIGNQ_XSWZ_MODE == 2
#define XSWZ(col) (((((col) >> 4) & 1) << 4) | ((((col) >> 3) & 1) << 3))
#elif ALIGNQ_XSWZ_MODE == 3
#define XSWZ(col) (((((col) >> 3) & 1) << 3) | ((((col) >> 5) & 1) << 4) | ((((col) >> 4) & 1) << 6))
#elif ALIGNQ_XSWZ_MODE == 4
#define XSWZ(col) (((((col) >> 4) & 1) << 4) | ((((col) >> 5) & 1) << 5))
#elif ALIGNQ_XSWZ_MODE == 5
#define XSWZ(col) (((((col) >> 4) & 1) << 4) | ((((col) >> 2) & 1) << 3))
#endif
#define LDS_F32(addr) (*reinterpret_cast<__attribute__((address_space(3))) float*>(addr))
#define LDS_F32X4(addr) (*reinterpret_cast<__attribute__((address_space(3))) f32x4_nt*>(addr))
// the two bf16 of one dword as floats (element 0 in the low half)
__device__ __forceinline__ f32x2 bf16_pair(uint32_t d) {
  return f32x2{__uint_as_float(d << 16), __uint_as_float(d & 0xffff0000u)};
}

__device__ __forceinline__ void split_bf16(float v, __bf16& hi, __bf16& lo) {
  hi = (__bf16)v;
  lo = (__bf16)(v - (float)hi);
}

template <bool SYM>
constexpr int kSlabReduceLds = 16 * 64 * (SYM ? 4 : 1) * 4 + 48 * 8 + 16;
template <bool SYM, bool LOSS>
__device__ __forceinline__ bool slab_reduce_body(const float* __restrict__ slabs, int n_slabs, int slab_floats, int BP, int B,
                                                 float scale, float* __restrict__ out, const float* __restrict__ A,
                                                 const float* __restrict__ gamma, int dim, float mu, float rho,
                                                 float* __restrict__ parts, unsigned* __restrict__ counter,
                                                 float* __restrict__ scal, const int blk, const int nblk,
                                                 unsigned char* __restrict__ lds);

// Filler role of a one-tile forward launch that leaves CUs idle (F <= 8192: 128 workgroups): the slab reduction + ADMM loss of
// up to three EARLIER sites of the same step (same B, dim, mu, rho), kSlabRedBlocks workgroups each behind the site's own tiles.
// Nothing reads those sites' D or loss before the end of the forward, and the reduction is the same code with the same workgroup
// partition as slab_reduce_multi_kernel: same bits.
constexpr int kSiteFill = 3;
constexpr int kSlabRedBlocks = (kSlab4Floats + 255) / 256;
struct SFill {
  const float* slabs[kSiteFill];
  float* out[kSiteFill];
  const float* A[kSiteFill];
  const float* gamma[kSiteFill];
  float* scal[kSiteFill];
  float scale[kSiteFill];
  int n_slabs[kSiteFill];
  int n, dim;
  float mu, rho;
};

// Twin launch (round 6): TWO sites of the same shape (B, F, k, act_range, eps) in one launch of the one-tile form - the workgroups from
// `split` on work on the second site's tensors.  For the two sites behind a transition block's convolutions (conv0's and skip_conv's
// outputs, model/resnet.py PreActBlock_conv_Q.forward): each of them is a 128-workgroup launch that leaves half the chip idle, and
// they do not depend on each other - one launch saves a node of the step's chain.  Same code per workgroup: bit-identical results.
struct Twin {
  const float* x; float* xq; float* slabs; float* stats; unsigned* counter; BnFold bn; int split;
};

// SINGLE: one tile per workgroup (n_tiles <= grid, every CIFAR-size site): no tile loop, so nothing is hoisted out of it and
// kept alive across the phases (88 instead of 128 VGPRs), which buys the early requests of the batch-norm finalisation.
// NTv: 1024 threads (16 waves: one workgroup per CU, the latency-tuned CIFAR form) or 512 (8 waves, 32-feature tiles, 45 KB of
// LDS: TWO workgroups per CU whose phases interleave - the multi-tile form for large F, where a tile's load -> transform ->
// statistics -> stage -> MFMA chain with its six barriers is otherwise exposed in full; plain sites only, no batch-norm fold).
template <int TFv, bool PAIR, bool SINGLE, int NTv = NT, bool FULLP = false>
__global__ __launch_bounds__(NTv, 4) void site_fwd4_kernel(const float* __restrict__ x, int B, int64_t F, int k, float r,
                                                       float eps, float* __restrict__ xq, float* __restrict__ slabs,
                                                       float* __restrict__ stats, int n_tiles, int aligned,
                                                       unsigned* __restrict__ counter, BnFold bn, SFill fill, Twin twin) {
  BSTAMP(0, 0);
  int bid = blockIdx.x;
  if constexpr (SINGLE && NTv == 1024) {
    if (twin.split && bid >= twin.split) {        // block-uniform: the second site of a twin launch
      bid -= twin.split;
      x = twin.x; xq = twin.xq; slabs = twin.slabs; stats = twin.stats; counter = twin.counter; bn = twin.bn;
    }
  }
  constexpr int LDB = TFv + 8;                    // bf16 elements per LDS row (row bytes multiple of 16, see bank note)
  constexpr int LPR = TFv / 4;                    // lanes per row (float4 each)
  constexpr int NW = NTv / 64;                    // waves
  constexpr bool kBnCode = NTv == NT;             // the batch-norm fold's wave assignment is written for 16 waves
  constexpr int RG = NTv / LPR;                   // row groups: 64 / 128 / 256
  constexpr int RJ = (128 + RG - 1) / RG;         // rows per thread: 2 / 1 / 1
  constexpr int NOP = PAIR ? 2 : 1;
  constexpr int ARR = 128 * LDB;                  // bf16 elements per array
  constexpr int STAGE_BYTES = (4 * ARR * 2 > 40960) ? 4 * ARR * 2 : 40960;
  constexpr int KSPLIT = (TFv >= 32 && NTv == NT) ? 2 : 1;   // K-halves per tile (a half must hold >= 16 features); the
                                                             // 8-wave form keeps 10 whole-K items (2 accumulators per wave)
  constexpr int KSTEPS = TFv / 16 / KSPLIT;       // 16-feature MFMA steps per item
  // the 512-thread multi-tile form is launched for COMPLETE tiles only (B == 128, F % 64 == 0, 16-byte aligned tensors): no
  // row / column masks, no selects behind the loads (54 -> vector instructions per element counted in the masks' favour: the
  // kernel runs at 62 % VALUBusy at [128, 524288])
  constexpr bool kPlain = !SINGLE && NTv == 512;     // no shortcut / ReLU / index / batch-norm paths at all
  // complete tiles: no masks (FULLP: the launcher's promise for the one-tile forms).  A complete tile has 128 rows: with 16-feature
  // tiles the 1024 threads form 256 row groups, half of them beyond the batch, so that geometry (ALIGNQ_FWD_WIDE=0 reaches it) keeps
  // the row mask whatever the launcher promised
  static_assert(!FULLP || (SINGLE && NTv == NT && RG <= 128),
                "FULLP is the one-tile 1024-thread form over complete 128-row tiles: 16-feature tiles have 256 row groups");
  static_assert(!kPlain || TFv == 64, "the 512-thread multi-tile form is written for 64-feature tiles");
  constexpr bool kFull = FULLP || kPlain;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE_BYTES + (4 * TFv + 2 * NW * TFv) * 4];
  __shared__ __attribute__((aligned(16))) float nerf_lds[PAIR ? ALIGNQ_NERF_LDS_FLOATS : 4];
  if constexpr (SINGLE && NTv == 1024) {
    if (bid >= n_tiles) {          // filler workgroups (block-uniform): see SFill (a twin launch has none)
      static_assert(STAGE_BYTES >= kSlabReduceLds<true>, "the reduction's LDS lies in the staging area");
      const int fb = bid - n_tiles, it = fb / kSlabRedBlocks;
      float* wsf = const_cast<float*>(fill.slabs[it]);
      float* parts = wsf + (size_t)fill.n_slabs[it] * kSlab4Floats;
      slab_reduce_body<true, true>(fill.slabs[it], fill.n_slabs[it], kSlab4Floats, 128, B, fill.scale[it], fill.out[it], fill.A[it],
                                   fill.gamma[it], fill.dim, fill.mu, fill.rho, parts,
                                   reinterpret_cast<unsigned*>(parts + kPartFloats), fill.scal[it], fb - it * kSlabRedBlocks,
                                   kSlabRedBlocks, lds_raw);
      return;
    }
  }
  // the transform's table (alignq_math.h): requested first, stored behind the tile loads of the first iteration
  NerfRegs<NTv> nerf_regs;
  if (PAIR) nerf_regs = nerf_tab_fetch<NTv>();
  const NerfTab tab = nerf_tab(nerf_lds);
  __bf16* Xhi = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* Xlo = Xhi + ARR;
  __bf16* Thi = Xhi + 2 * ARR;
  __bf16* Tlo = Xhi + 3 * ARR;
  float* colv = reinterpret_cast<float*>(lds_raw + STAGE_BYTES);   // mean_x, rho_x, mean_t, rho_t : [4][TFv]
  float* red = colv + 4 * TFv;                                     // [2 operands][NW waves][TFv]

  const int tid = threadIdx.x, lane = tid & 63;
  // the wave index is wave-uniform: readfirstlane puts it (and the work-item bookkeeping derived from it: tile, K-half,
  // I/J blocks, LDS fragment bases) into SGPRs instead of long-lived VGPRs of a kernel that runs at its 128-VGPR cap
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = tid % LPR, rg = tid / LPR;
  const int h = lane >> 5, l31 = lane & 31;
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  const float invBm1 = 1.0f / (float)(B - 1);

  if (bid == 0 && tid == 0 && counter) *counter = 0u;   // arrival counter of the reduce kernel's epilogue

  // work items: 10 upper-triangular output tiles x KSPLIT K-halves, item q = (tile q % 10, K-half q / 10); wave w takes
  // q = w, w + NW, ...: 16 waves: 20 items -> q = w and (waves 0..3) 16 + w; 10 items -> waves 0..9;  8 waves: 3 / 3 / 2.
  constexpr int NITEMS = 10 * KSPLIT;
  constexpr int NI = (NITEMS + NW - 1) / NW;
  int it_tile[NI], it_kh[NI], it_I[NI], it_J[NI];
  bool it_on[NI];
  f32x16 acc[NI];       // PAIR: T-Gram minus X-Gram (the x A-operand enters negated); else the X-Gram
#pragma unroll
  for (int i = 0; i < NI; i++) {
    const int q = w + NW * i;
    it_on[i] = q < NITEMS;
    it_tile[i] = it_on[i] ? q % 10 : 0;
    it_kh[i] = it_on[i] ? q / 10 : 0;
    tile_ij(it_tile[i], it_I[i], it_J[i]);
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
  }

  STAMP(0);
  float4 xv[RJ];      // this tile's rows; in the multi-tile form refilled with the NEXT tile's rows once they are staged
  for (int tile = bid; tile < n_tiles; tile += gridDim.x) {
    const int col0 = tile * TFv;
    const int col = col0 + 4 * c;
    float4 tv[RJ];
    // ---- channels-last batch-norm finalisation, part 1 (one-tile launches): everything it reads is requested BEFORE the
    // tile loads (memory returns in order, and no element can be transformed before (a, b) exist).  Wave w owns channel
    // chbase + w (and + 16 on the narrow tiles); lane l of every wave fetches gamma / beta (/ the running statistics in the
    // publishing workgroups) of channel chbase + l with one coalesced load; the convolution epilogue's float partials
    // [C][n_parts][2] come as float4 = two partials per lane and load (16 registers in flight cover 512 partials of one
    // channel, or 256 of two).  The multi-tile variant runs at its 128-register cap and keeps the plain order.
    constexpr bool kPairCh = TFv < 64;
    constexpr int PU = kPairCh ? 2 : 4;
    const bool fin = kBnCode && bn.ab && bn.nhwc && bn.part;
    const bool fin4 = SINGLE && fin && bn.part_f32 && !(bn.n_parts & 1);
    const int nch = bn.C < TFv ? bn.C : TFv;
    const int chbase = col0 & (bn.C - 1);
    const bool publish = col0 < bn.C;       // the tiles of the first pixel cover every channel exactly once
    float gam_l = 1.0f, bet_l = 0.0f, rm_l = 0.0f, rv_l = 0.0f;
    float4 pv0[SINGLE ? PU : 1], pv1[SINGLE ? PU : 1];
    if constexpr (SINGLE) {
      if (fin) {
        const int chl = chbase + (lane & (nch - 1));
        if (bn.gamma) gam_l = bn.gamma[chl];
        if (bn.beta) bet_l = bn.beta[chl];
        if (publish) {
          if (bn.running_mean) rm_l = bn.running_mean[chl];
          if (bn.running_var) rv_l = bn.running_var[chl];
        }
      }
      if (fin4 && w < nch) {
        const int nq = bn.n_parts >> 1;
        const bool two = kPairCh && w + 16 < nch;
        const float4* pq0 = reinterpret_cast<const float4*>(bn.part) + (int64_t)(chbase + w) * nq;
        const float4* pq1 = reinterpret_cast<const float4*>(bn.part) + (int64_t)(chbase + (two ? w + 16 : w)) * nq;
#pragma unroll
        for (int u = 0; u < PU; u++) {
          const int pi = lane + 64 * u, pc = pi < nq ? pi : nq - 1;
          pv0[u] = pq0[pc];
          if (kPairCh) pv1[u] = pq1[pc];
        }
      }
    }
    // ---- load + transform + quantise ----------------------------------------------------------------
    // (multi-tile launches: every tile but the first was requested by the previous iteration, in front of its MFMA phase)
    if (SINGLE || tile == bid) {
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        const int row = rg + RG * j;
        const bool ok = kFull || row < B;
        const unsigned off = (unsigned)row * (unsigned)F + (unsigned)col;
        if constexpr (kFull && SINGLE) {
          xv[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(x) + 4u * off);
        } else if constexpr (kFull) {
          const f32x4_nt t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(reinterpret_cast<const char*>(x) + 4u * off));
          xv[j] = make_float4(t4.x, t4.y, t4.z, t4.w);
        } else {
          xv[j] = SINGLE ? ld4(x, off, col, F, ok, aligned) : ld4_stream(x, off, col, F, ok, aligned);
        }
      }
    }
    if (PAIR && tile == bid) {     // first iteration (block-uniform): publish the transform's table
      nerf_tab_store<NTv>(nerf_lds, nerf_regs);
      __syncthreads();
    }
    // ---- folded batch-norm: x = a*z + b with (a, b) of this tile's channel (HW % 64 == 0: one channel per tile) -------
    if (kBnCode && bn.ab && bn.nhwc) {
      // channels-last: the float4 at column `col` covers channels (col mod C) .. +3 (C % 4 == 0); a, b are inputs
      const int ch = col & (bn.C - 1);
      float4 a4, b4;
      if (bn.part) {
        {
        // finalise the batch statistics of this tile's channels here (saves a launch and a grid-wide hand-off): wave w
        // reduces the partials of channel chbase + w (+16, ...) with a fixed butterfly
        const int C = bn.C;
        const double n = (double)B * (double)bn.HW;
        // 1/n is wave-uniform: kept in a scalar register pair
        const double inv_nv = 1.0 / n;
        const double inv_n = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(inv_nv)),
                                              __builtin_amdgcn_readfirstlane(__double2loint(inv_nv)));
        for (int cl0 = w; cl0 < nch; cl0 += kPairCh ? 32 : 16) {
          const int cl1 = cl0 + 16;
          const bool two_p = kPairCh && cl1 < nch;
          const int c0 = chbase + cl0, c1 = chbase + (two_p ? cl1 : cl0);
          double sa0 = 0, sq0 = 0, sa1 = 0, sq1 = 0;
          if (fin4) {
            const int nq = bn.n_parts >> 1;
            const float4* pq0 = reinterpret_cast<const float4*>(bn.part) + (int64_t)c0 * nq;
            const float4* pq1 = reinterpret_cast<const float4*>(bn.part) + (int64_t)c1 * nq;
            for (int p0 = lane; p0 < nq; p0 += 64 * PU) {
              float4 v0[PU], v1[PU];
              if (SINGLE && cl0 == w && p0 == lane) {    // first round of the first pass: requested in front of the tile loads
#pragma unroll
                for (int u = 0; u < PU; u++) { v0[u] = pv0[SINGLE ? u : 0]; v1[u] = pv1[SINGLE ? u : 0]; }
              } else {
#pragma unroll
                for (int u = 0; u < PU; u++) {
                  const int pi = p0 + 64 * u, pc = pi < nq ? pi : nq - 1;
                  v0[u] = pq0[pc];
                  if (kPairCh) v1[u] = pq1[pc];
                }
              }
#pragma unroll
              for (int u = 0; u < PU; u++) {
                if (p0 + 64 * u < nq) {
                  sa0 += v0[u].x; sq0 += v0[u].y; sa0 += v0[u].z; sq0 += v0[u].w;
                  if (kPairCh) { sa1 += v1[u].x; sq1 += v1[u].y; sa1 += v1[u].z; sq1 += v1[u].w; }
                }
              }
            }
          } else if (bn.part_f32) {  // odd partial count or a multi-tile launch: one partial per lane and load
            const float2* pf0 = reinterpret_cast<const float2*>(bn.part) + (int64_t)c0 * bn.n_parts;
            const float2* pf1 = reinterpret_cast<const float2*>(bn.part) + (int64_t)c1 * bn.n_parts;
            constexpr int PU2 = kPairCh ? 4 : 8;     // 16 registers of loads either way
            for (int p0 = lane; p0 < bn.n_parts; p0 += 64 * PU2) {
              float2 v0[PU2], v1[PU2];
#pragma unroll
              for (int u = 0; u < PU2; u++) {
                const int pi = p0 + 64 * u, pc = pi < bn.n_parts ? pi : bn.n_parts - 1;
                v0[u] = pf0[pc];
                if (kPairCh) v1[u] = pf1[pc];
              }
#pragma unroll
              for (int u = 0; u < PU2; u++) {
                if (p0 + 64 * u < bn.n_parts) {
                  sa0 += v0[u].x; sq0 += v0[u].y;
                  if (kPairCh) { sa1 += v1[u].x; sq1 += v1[u].y; }
                }
              }
            }
          } else {
            sa0 = bn.part[((int64_t)c0 * kNhwcParts + lane) * 2];
            sq0 = bn.part[((int64_t)c0 * kNhwcParts + lane) * 2 + 1];
            if (kPairCh) {
              sa1 = bn.part[((int64_t)c1 * kNhwcParts + lane) * 2];
              sq1 = bn.part[((int64_t)c1 * kNhwcParts + lane) * 2 + 1];
            }
          }
          sa0 = wave_sum_d_dpp(sa0);
          sq0 = wave_sum_d_dpp(sq0);
          if (kPairCh) { sa1 = wave_sum_d_dpp(sa1); sq1 = wave_sum_d_dpp(sq1); }
          // per-channel inputs of the owner lanes: fetched in front of the tile loads by lane cl of every wave (cl0, cl1 are
          // wave-uniform: v_readlane); the multi-tile variant has no registers to hold them and loads at the point of use
          float g0 = 1.0f, g1 = 1.0f, e0 = 0.0f, e1 = 0.0f, rm0 = 0.0f, rm1 = 0.0f, rv0 = 0.0f, rv1 = 0.0f;
          if constexpr (SINGLE) {
            const int cla = cl0, clb = two_p ? cl1 : cl0;
            g0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gam_l), cla));
            g1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gam_l), clb));
            e0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bet_l), cla));
            e1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bet_l), clb));
            if (publish) {
              rm0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm_l), cla));
              rm1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rm_l), clb));
              rv0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv_l), cla));
              rv1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rv_l), clb));
            }
          }
          if (lane < 2 && (lane == 0 || two_p)) {            // lane 0: channel c0, lane 1: channel c1
            const int cl = lane ? cl1 : cl0, cc = lane ? c1 : c0;
            const double sa = lane ? sa1 : sa0, sq = lane ? sq1 : sq0;
            const double mean = sa * inv_n;
            double var = sq * inv_n - mean * mean;
            if (var < 0) var = 0;
            const float invstd = 1.0f / sqrtf((float)(var + (double)bn.bn_eps));     // fp32 like torch's batch-norm
            float gv = lane ? g1 : g0, ev = lane ? e1 : e0;
            if constexpr (!SINGLE) {
              gv = bn.gamma ? bn.gamma[cc] : 1.0f;
              ev = bn.beta ? bn.beta[cc] : 0.0f;
            }
            const float av = gv * invstd;
            const float bv = ev - (float)mean * av;
            colv[cl] = av;
            colv[TFv + cl] = bv;
            if (publish) {
              float* abo = const_cast<float*>(bn.ab);
              float* svo = const_cast<float*>(bn.save);
              abo[cc] = av; abo[C + cc] = bv;
              svo[cc] = (float)mean; svo[C + cc] = invstd;
              float rm = lane ? rm1 : rm0, rv = lane ? rv1 : rv0;
              if constexpr (!SINGLE) {
                rm = bn.running_mean ? bn.running_mean[cc] : 0.0f;
                rv = bn.running_var ? bn.running_var[cc] : 0.0f;
              }
              if (bn.running_mean) bn.running_mean[cc] = (1.0f - bn.momentum) * rm + bn.momentum * (float)mean;
              if (bn.running_var) bn.running_var[cc] = (1.0f - bn.momentum) * rv + bn.momentum * (float)(var * n / (n - 1.0));
              if (cc == 0 && bn.nbt) *bn.nbt += 1;
            }
          }
        }
        }
        __syncthreads();
        const int jl = (4 * c) & (nch - 1);                // this thread's columns -> local channel index
        a4 = *reinterpret_cast<const float4*>(colv + jl);
        b4 = *reinterpret_cast<const float4*>(colv + TFv + jl);
        // (no barrier needed before colv is reused: its next writer, the column statistics, writes only after a barrier
        // that every thread reaches after these reads)
      } else {
        a4 = *reinterpret_cast<const float4*>(bn.ab + ch);
        b4 = *reinterpret_cast<const float4*>(bn.ab + bn.C + ch);
      }
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        if (kFull || (rg + RG * j < B && col < F)) {
          xv[j].x = __fmaf_rn(a4.x, xv[j].x, b4.x); xv[j].y = __fmaf_rn(a4.y, xv[j].y, b4.y);
          xv[j].z = __fmaf_rn(a4.z, xv[j].z, b4.z); xv[j].w = __fmaf_rn(a4.w, xv[j].w, b4.w);
        }
      }
    } else if (kBnCode && bn.ab) {
      const int ch = col0 / bn.HW;
      float bn_a, bn_b;
      if (bn.part) {
        // finalise the batch statistics here (saves a launch): 16 lanes reduce the 16 split partials of the channel
        if (w == 0) {
          double sa = 0, sq = 0;
          if (lane < kBnSplit) { sa = bn.part[(ch * kBnSplit + lane) * 2]; sq = bn.part[(ch * kBnSplit + lane) * 2 + 1]; }
          sa = wave_sum_d(sa);
          sq = wave_sum_d(sq);
          if (lane == 0) {
            const double n = (double)B * (double)bn.HW;
            const double mean = sa / n;
            double var = sq / n - mean * mean;
            if (var < 0) var = 0;
            const float invstd = (float)(1.0 / sqrt(var + (double)bn.bn_eps));
            const float av = (bn.gamma ? bn.gamma[ch] : 1.0f) * invstd;
            const float bv = (bn.beta ? bn.beta[ch] : 0.0f) - (float)mean * av;
            colv[0] = av;
            colv[1] = bv;
            if (col0 % bn.HW == 0) {          // first tile of the channel publishes the per-channel results
              float* abo = const_cast<float*>(bn.ab);
              float* svo = const_cast<float*>(bn.save);
              abo[ch] = av; abo[bn.C + ch] = bv;
              svo[ch] = (float)mean; svo[bn.C + ch] = invstd;
              if (bn.running_mean) bn.running_mean[ch] = (1.0f - bn.momentum) * bn.running_mean[ch] + bn.momentum * (float)mean;
              if (bn.running_var) bn.running_var[ch] = (1.0f - bn.momentum) * bn.running_var[ch] + bn.momentum * (float)(var * n / (n - 1.0));
              if (col0 == 0 && bn.nbt) *bn.nbt += 1;
            }
          }
        }
        __syncthreads();
        bn_a = colv[0];
        bn_b = colv[1];       // (colv's next writer runs behind a later barrier: no second barrier here)
      } else {
        bn_a = bn.ab[ch];
        bn_b = bn.ab[bn.C + ch];
      }
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        if (kFull || (rg + RG * j < B && col < F)) {
          xv[j].x = __fmaf_rn(bn_a, xv[j].x, bn_b); xv[j].y = __fmaf_rn(bn_a, xv[j].y, bn_b);
          xv[j].z = __fmaf_rn(bn_a, xv[j].z, bn_b); xv[j].w = __fmaf_rn(bn_a, xv[j].w, bn_b);
        }
      }
    }
    // fused residual add: loads issued here and consumed after the erf work below; the 64-feature tile has no registers
    // to spare (128-VGPR budget at 1024 threads) and loads row by row inside the loop instead
    constexpr bool kEarlyRes = TFv < 64;
    float4 rv[RJ];
    if (!kPlain && PAIR && kEarlyRes && bn.res) {
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        const int row = rg + RG * j;
        if constexpr (kFull) rv[j] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bn.res) + 4u * ((unsigned)row * (unsigned)F + (unsigned)col));
        else rv[j] = ld4(bn.res, (unsigned)row * (unsigned)F + (unsigned)col, col, F, row < B, aligned);
      }
    }
#pragma unroll
    for (int j = 0; j < RJ; j++) {
      const int row = rg + RG * j;
      const bool ok = kFull || row < B;
      const unsigned off = (unsigned)row * (unsigned)F + (unsigned)col;
      tv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (PAIR) {
        float4 q, rl = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!kPlain && !kEarlyRes && bn.res) {                 // (kPlain: the plain site - no shortcut, ReLU, indices)
          if constexpr (kFull) rl = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bn.res) + 4u * off);
          else rl = ld4(bn.res, off, col, F, ok, aligned);
        }
        float b0, b1, b2, b3;
        if (nlev.yn != 0.0f) {      // launch-uniform: no per-element branches on k in the common case
          q.x = act_quant1<0, true>(xv[j].x, k, nlev, r, &tv[j].x, &b0, tab);
          q.y = act_quant1<0, true>(xv[j].y, k, nlev, r, &tv[j].y, &b1, tab);
          q.z = act_quant1<0, true>(xv[j].z, k, nlev, r, &tv[j].z, &b2, tab);
          q.w = act_quant1<0, true>(xv[j].w, k, nlev, r, &tv[j].w, &b3, tab);
        } else {
          q.x = act_quant1<0>(xv[j].x, k, nlev, r, &tv[j].x, &b0, tab);
          q.y = act_quant1<0>(xv[j].y, k, nlev, r, &tv[j].y, &b1, tab);
          q.z = act_quant1<0>(xv[j].z, k, nlev, r, &tv[j].z, &b2, tab);
          q.w = act_quant1<0>(xv[j].w, k, nlev, r, &tv[j].w, &b3, tab);
        }
        if (!kPlain && bn.bins && (kFull || (ok && col < F))) {
          // N2: the level index of the stored value (no residual on this path; the fused ReLU clamps the index at 0), narrow:
          // 8 or 4 bytes per quad at the element offset (the launcher requires the aligned float4 path: F % 4 == 0)
          if (bn.relu) { b0 = fmaxf(b0, 0.f); b1 = fmaxf(b1, 0.f); b2 = fmaxf(b2, 0.f); b3 = fmaxf(b3, 0.f); }
          if (bn.bin_bytes == 2) {
            short4 bi; bi.x = (short)(int)b0; bi.y = (short)(int)b1; bi.z = (short)(int)b2; bi.w = (short)(int)b3;
            *reinterpret_cast<short4*>(reinterpret_cast<char*>(bn.bins) + 2u * off) = bi;
          } else {
            char4 bi; bi.x = (signed char)(int)b0; bi.y = (signed char)(int)b1; bi.z = (signed char)(int)b2; bi.w = (signed char)(int)b3;
            *reinterpret_cast<char4*>(reinterpret_cast<char*>(bn.bins) + off) = bi;
          }
        }
        if (!kPlain && bn.res) {
          const float4 rr = kEarlyRes ? rv[j] : rl;
          q.x += rr.x; q.y += rr.y; q.z += rr.z; q.w += rr.w;
        }
        if (!kPlain && bn.relu) { q.x = fmaxf(q.x, 0.f); q.y = fmaxf(q.y, 0.f); q.z = fmaxf(q.z, 0.f); q.w = fmaxf(q.w, 0.f); }
        if constexpr (kFull) {
          if (xq) *reinterpret_cast<float4*>(reinterpret_cast<char*>(xq) + 4u * off) = q;
        } else {
          if (xq) st4(xq, off, col, F, ok, aligned, q);
        }
      }
    }
    STAMP(1);
    // ---- column means: registers -> row-group sums inside the wave (VALU lane swaps) -> LDS over waves ----
    // after rowgroup_sums, lane (row r, column quad c = lane % LPR) holds the wave's sum of column 4c + kRowVal[r]
    const int rsel = ((lane >> 4) & 1) * 2 + (lane >> 5);     // kRowVal[lane / 16]
    const bool rwrite = (lane & 15) < LPR;
    {
      float sx[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0};
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        if (kFull || rg + RG * j < B) {
          sx[0] += xv[j].x; sx[1] += xv[j].y; sx[2] += xv[j].z; sx[3] += xv[j].w;
          if (PAIR) { st[0] += tv[j].x; st[1] += tv[j].y; st[2] += tv[j].z; st[3] += tv[j].w; }
        }
      }
      float t0, t1;
      rowgroup_sums<LPR, PAIR>(sx, st, t0, t1);
      if (rwrite) {
        red[w * TFv + 4 * c + rsel] = t0;
        if (PAIR) red[NW * TFv + w * TFv + 4 * c + rsel] = t1;
      }
    }
    __syncthreads();
    if (tid < NOP * TFv) {
      const int op = tid / TFv, cc = tid % TFv;
      float pw[NW];
#pragma unroll
      for (int g = 0; g < NW; g++) pw[g] = red[op * NW * TFv + g * TFv + cc];      // NW reads in flight, then a fixed tree
#pragma unroll
      for (int o = 1; o < NW; o <<= 1) {
#pragma unroll
        for (int g = 0; g < NW; g += 2 * o) pw[g] += pw[g + o];
      }
      colv[(2 * op) * TFv + cc] = pw[0] / (float)B;      // true division, like torch.mean (a constant column: exactly its value)
    }
    __syncthreads();
    // ---- column variances (two-pass) --------------------------------------------------------------
    {
      float mx[4], mt[4], sx[4] = {0, 0, 0, 0}, st[4] = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < 4; e++) { mx[e] = colv[4 * c + e]; mt[e] = PAIR ? colv[2 * TFv + 4 * c + e] : 0.f; }
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        if (kFull || rg + RG * j < B) {
          const float xe[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
          const float te[4] = {tv[j].x, tv[j].y, tv[j].z, tv[j].w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float d = xe[e] - mx[e];
            sx[e] += d * d;
            if (PAIR) { d = te[e] - mt[e]; st[e] += d * d; }
          }
        }
      }
      float t0, t1;
      rowgroup_sums<LPR, PAIR>(sx, st, t0, t1);
      if (rwrite) {
        red[w * TFv + 4 * c + rsel] = t0;
        if (PAIR) red[NW * TFv + w * TFv + 4 * c + rsel] = t1;
      }
    }
    __syncthreads();
    if (tid < NOP * TFv) {
      const int op = tid / TFv, cc = tid % TFv;
      float pw[NW];
#pragma unroll
      for (int g = 0; g < NW; g++) pw[g] = red[op * NW * TFv + g * TFv + cc];
#pragma unroll
      for (int o = 1; o < NW; o <<= 1) {
#pragma unroll
        for (int g = 0; g < NW; g += 2 * o) pw[g] += pw[g + o];
      }
      const float sd = sqrtf(pw[0] * invBm1);
      const float rho = 1.0f / (sd + eps);
      colv[(2 * op + 1) * TFv + cc] = rho;
      if (stats && (kFull || col0 + cc < F)) {
        stats[(int64_t)(2 * op) * F + col0 + cc] = colv[(2 * op) * TFv + cc];
        stats[(int64_t)(2 * op + 1) * F + col0 + cc] = rho;
      }
    }
    __syncthreads();
    STAMP(2);
    // ---- standardise, split into bf16 hi/lo, stage in LDS (one 8-byte store per array and row) -------------
    {
      float mx[4], rx[4], mt[4], rt[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        mx[e] = colv[4 * c + e];
        rx[e] = colv[TFv + 4 * c + e];
        mt[e] = PAIR ? colv[2 * TFv + 4 * c + e] : 0.f;
        rt[e] = PAIR ? colv[3 * TFv + 4 * c + e] : 0.f;
      }
#pragma unroll
      for (int j = 0; j < RJ; j++) {
        const int row = rg + RG * j;
        if (row < 128) {
          const bool ok = kFull || row < B;
          const float xe[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
          const float te[4] = {tv[j].x, tv[j].y, tv[j].z, tv[j].w};
          bf16x4 xh, xl, th, tl;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const bool okc = kFull || (ok && (col + e < F));
            __bf16 a, b2;
            split_bf16(okc ? (xe[e] - mx[e]) * rx[e] : 0.0f, a, b2);
            xh[e] = a; xl[e] = b2;
            if (PAIR) {
              split_bf16(okc ? (te[e] - mt[e]) * rt[e] : 0.0f, a, b2);
              th[e] = a; tl[e] = b2;
            }
          }
          const int o = row * LDB + 4 * c;
          *reinterpret_cast<bf16x4*>(Xhi + o) = xh;
          *reinterpret_cast<bf16x4*>(Xlo + o) = xl;
          if (PAIR) {
            *reinterpret_cast<bf16x4*>(Thi + o) = th;
            *reinterpret_cast<bf16x4*>(Tlo + o) = tl;
          }
        }
      }
    }
    __syncthreads();
    STAMP(3);
    if constexpr (!SINGLE) {
      // software pipeline: x and t of this tile now live in LDS, so the registers take the next tile's rows; the loads fly
      // under the MFMA phase and the loop's closing barrier instead of in front of the next transform.  (The 8-wave form
      // reports 12 B of scratch per lane: ONE 8-byte prologue value stored before the tile loop and reloaded after it -
      // no scratch traffic inside the loop; requesting half of the rows behind the MFMA phase did not change that.)
      if constexpr (kPlain) {     // always issued (the last iteration re-reads its own tile): no branch around the loads
        const int tn = min(tile + (int)gridDim.x, n_tiles - 1);
        const int coln = tn * TFv + 4 * c;
#pragma unroll
        for (int j = 0; j < RJ; j++) {
          const int row = rg + RG * j;
          const f32x4_nt t4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(
              reinterpret_cast<const char*>(x) + 4u * ((unsigned)row * (unsigned)F + (unsigned)coln)));
          xv[j] = make_float4(t4.x, t4.y, t4.z, t4.w);
        }
      } else {
        const int tn = tile + (int)gridDim.x;
        if (tn < n_tiles) {
          const int coln = tn * TFv + 4 * c;
#pragma unroll
          for (int j = 0; j < RJ; j++) {
            const int row = rg + RG * j;
            xv[j] = ld4_stream(x, (unsigned)row * (unsigned)F + (unsigned)coln, coln, F, row < B, aligned);
          }
        }
      }
    }
    // ---- MFMA: upper-triangular tiles of Th Th^T and Xh Xh^T (3 bf16 MFMAs each per 16 features) ------------
    if constexpr (SINGLE) {      // one tile per workgroup: the accumulators start their life here, not in front of the erf work
#pragma unroll
      for (int i = 0; i < NI; i++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[i][e] = 0.0f;
    }
    // A operand: lane -> row I*32 + l31, 8 consecutive features k0 + 8h..; B operand: row J*32 + l31, same features
#pragma unroll
    for (int i = 0; i < NI; i++) {
      if (it_on[i]) {
        const int ra = (it_I[i] * 32 + l31) * LDB + 8 * h, rb = (it_J[i] * 32 + l31) * LDB + 8 * h;
        constexpr int kUnrollK = (NTv == NT) ? KSTEPS : 1;   // (the 8-wave form runs at its 128-register cap: no fragment
#pragma unroll kUnrollK                                     //  loads hoisted across K steps)
        for (int s = 0; s < KSTEPS; s++) {
          const int k0 = (it_kh[i] * KSTEPS + s) * 16;
          bf16x8 axh = *reinterpret_cast<const bf16x8*>(Xhi + ra + k0);
          bf16x8 axl = *reinterpret_cast<const bf16x8*>(Xlo + ra + k0);
          const bf16x8 bxh = *reinterpret_cast<const bf16x8*>(Xhi + rb + k0);
          const bf16x8 bxl = *reinterpret_cast<const bf16x8*>(Xlo + rb + k0);
          if (PAIR) { axh = neg8(axh); axl = neg8(axl); }
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(axh, bxh, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(axh, bxl, acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(axl, bxh, acc[i], 0, 0, 0);
          if (PAIR) {
            const bf16x8 ath = *reinterpret_cast<const bf16x8*>(Thi + ra + k0);
            const bf16x8 atl = *reinterpret_cast<const bf16x8*>(Tlo + ra + k0);
            const bf16x8 bth = *reinterpret_cast<const bf16x8*>(Thi + rb + k0);
            const bf16x8 btl = *reinterpret_cast<const bf16x8*>(Tlo + rb + k0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ath, bth, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ath, btl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(atl, bth, acc[i], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();   // LDS tiles are overwritten by the next iteration / by the combine below
    if constexpr (SINGLE) break;
  }

  STAMP(4);
  // ---- combine the K-halves of every tile in LDS in a fixed order (deterministic); write the slab ---------------
  float* C = reinterpret_cast<float*>(lds_raw);   // [10][32][32]
#pragma unroll
  for (int i = 0; i < NI; i++) {
    if (it_on[i] && it_kh[i] == 0) {
#pragma unroll
      for (int e = 0; e < 16; e++)
        C[it_tile[i] * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + l31] = acc[i][e];
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NI; i++) {
    if (it_on[i] && it_kh[i] == 1) {
#pragma unroll
      for (int e = 0; e < 16; e++)
        C[it_tile[i] * 1024 + ((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + l31] += acc[i][e];
    }
  }
  __syncthreads();
  // packed slab (site_internal.h, kSlab4Floats): six off-diagonal tiles as they are, the four diagonal tiles' upper
  // triangles two to a [32][33] block
  float* slab = slabs + (int64_t)bid * kSlab4Floats;
  float4* slab4 = reinterpret_cast<float4*>(slab);
  const float4* C4 = reinterpret_cast<const float4*>(C);
  for (int e = tid; e < kSlab4Off / 4; e += NTv) {
    const int t6 = e >> 8;                                      // C tiles 1, 2, 3, 5, 6, 8
    const int tl = t6 < 3 ? t6 + 1 : (t6 < 5 ? t6 + 2 : 8);
    slab4[e] = C4[tl * 256 + (e & 255)];
  }
  for (int idx = tid; idx < kSlab4Packed; idx += NTv) {
    const int blk = idx / 1056, rem = idx - blk * 1056;
    const int row = rem / 33, col = rem - row * 33;
    const bool second = col <= row;                             // tile 2*blk + 1, stored transposed
    const int rr = second ? col : row, cc2 = second ? row : col - 1;
    const int d = 2 * blk + (second ? 1 : 0);
    const int tl = d == 0 ? 0 : (d == 1 ? 4 : (d == 2 ? 7 : 9));
    slab[kSlab4Off + idx] = C[tl * 1024 + rr * 32 + cc2];
  }
  STAMP(5);
  BSTAMP(0, 1);
}

// ================================================================================================ reduce
// out[i][j] = scale * sum_s slab[s](i,j), i,j < B.  1024 threads = 16 slab groups x 64 elements.
// SYM : slabs hold the 10 upper-triangular tiles (site4 geometry) instead of a full BPxBP matrix.
// LOSS: additionally the ADMM loss scalar (utils/admm.py:24-33) through a last-block epilogue:
//       scal = {loss, c_con = rho/2/(n*rms), 1/n, rms}.
template <bool SYM, bool LOSS>
__device__ __forceinline__ bool slab_reduce_body(const float* __restrict__ slabs, int n_slabs,
                                                           int slab_floats, int BP, int B, float scale,
                                                           float* __restrict__ out, const float* __restrict__ A,
                                                           const float* __restrict__ gamma, int dim, float mu,
                                                           float rho, float* __restrict__ parts,
                                                           unsigned* __restrict__ counter, float* __restrict__ scal,
                                                           const int blk, const int nblk, unsigned char* __restrict__ lds) {
  // SYM: a lane owns FOUR consecutive stored elements (one 16-byte load per slab: the slab stream is the HBM-bound part of a
  // step, 215 MB for ResNet-20); else one element of the full BPxBP slab.
  // blk / nblk: this workgroup's index among the workgroups of THIS reduction (a launch of its own: blockIdx.x / gridDim.x; a
  // filler role of another launch: see SFill);  lds: kSlabReduceLds<SYM> bytes, 16-byte aligned
  constexpr int EPL = SYM ? 4 : 1;
  float (*part)[64 * EPL] = reinterpret_cast<float (*)[64 * EPL]>(lds);
  double* fin = reinterpret_cast<double*>(lds + 16 * 64 * EPL * 4);
  int& is_last = *reinterpret_cast<int*>(lds + 16 * 64 * EPL * 4 + 48 * 8);
  const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
  const int e = (blk * 64 + lane) * EPL;
  // SYM: iterate over the STORED elements (10 tiles x 32 x 32, coalesced) and mirror the off-diagonal tiles;
  // else: over the output elements of the full BPxBP slab.
  int iq[EPL], jq[EPL], off;
  bool ok[EPL], mir[EPL], any_ok = false;
  if (SYM) {
#pragma unroll
    for (int q = 0; q < EPL; q++) {
      const int eq = e + q;
      int ti, tj, r, c2;
      if (eq < kSlab4Off) {
        const int t6 = eq >> 10, within = eq & 1023;
        ti = t6 < 3 ? 0 : (t6 < 5 ? 1 : 2);
        tj = t6 < 3 ? t6 + 1 : (t6 < 5 ? t6 - 1 : 3);
        r = within >> 5; c2 = within & 31;
      } else {
        const int idx = eq - kSlab4Off;
        const int blk = idx / 1056, rem = idx - blk * 1056;
        const int row = rem / 33, col = rem - row * 33;
        const bool second = col <= row;
        r = second ? col : row; c2 = second ? row : col - 1;
        ti = tj = 2 * blk + (second ? 1 : 0);
      }
      iq[q] = ti * 32 + r;
      jq[q] = tj * 32 + c2;
      ok[q] = eq < kSlab4Floats && iq[q] < B && jq[q] < B;
      mir[q] = ok[q] && iq[q] != jq[q];
      any_ok = any_ok || ok[q];
    }
    off = e < kSlab4Floats ? e : 0;
  } else {
    ok[0] = e < B * B;
    any_ok = ok[0];
    iq[0] = ok[0] ? e / B : 0;
    jq[0] = ok[0] ? e - iq[0] * B : 0;
    mir[0] = false;
    off = iq[0] * BP + jq[0];
  }
  const float* p = slabs + off;
  // the loss operands do not depend on the slab sums: issue their loads first so they fly under the reduction
  float a_ij[EPL], g_ij[EPL], a_ji[EPL], g_ji[EPL];
#pragma unroll
  for (int q = 0; q < EPL; q++) { a_ij[q] = 0.f; g_ij[q] = 0.f; a_ji[q] = 0.f; g_ji[q] = 0.f; }
  if (LOSS && sg == 0) {
#pragma unroll
    for (int q = 0; q < EPL; q++) {
      if (ok[q]) { a_ij[q] = A[iq[q] * dim + jq[q]]; g_ij[q] = gamma[iq[q] * dim + jq[q]]; }
      if (mir[q]) { a_ji[q] = A[jq[q] * dim + iq[q]]; g_ji[q] = gamma[jq[q] * dim + iq[q]]; }
    }
  }
  float s[EPL];
#pragma unroll
  for (int q = 0; q < EPL; q++) s[q] = 0.f;
  if (any_ok) {
    if constexpr (SYM) {
#pragma unroll 16
      for (int sl = sg; sl < n_slabs; sl += 16) {     // non-temporal: read once, and not to evict the host launch's operands
        const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p + (int64_t)sl * slab_floats));
        s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      }
    } else {
#pragma unroll 16
      for (int sl = sg; sl < n_slabs; sl += 16) s[0] += p[(int64_t)sl * slab_floats];
    }
  }
#pragma unroll
  for (int q = 0; q < EPL; q++) part[sg][lane * EPL + q] = s[q];
  __syncthreads();
  if (sg == 0) {
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int q = 0; q < EPL; q++) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < 16; g++) t += part[g][lane * EPL + q];
      const float d = t * scale;
      if (ok[q]) out[iq[q] * B + jq[q]] = d;
      if (mir[q]) out[jq[q] * B + iq[q]] = d;
      if (LOSS) {
        if (ok[q]) {
          const float dd = d - a_ij[q];
          v0 += fabsf(a_ij[q]);
          v1 += dd * dd;
          v2 += g_ij[q] * fabsf(dd);
        }
        if (mir[q]) {
          const float dd = d - a_ji[q];
          v0 += fabsf(a_ji[q]);
          v1 += dd * dd;
          v2 += g_ji[q] * fabsf(dd);
        }
      }
    }
    if (LOSS) {
      v0 = wave_sum(v0);
      v1 = wave_sum(v1);
      v2 = wave_sum(v2);
      if (lane == 0) {
        // Hand-off without an L2 write-back (a release fence here would flush what the previous kernel left dirty
        // in this XCD's L2): the partials are stored write-through (sc1), drained, then one relaxed agent-scope
        // ticket; the block whose ticket is last reads every partial with sc1 loads.
        // ISA ASSUMPTION (gfx950, not the C++ memory model): agent-scope atomic stores/loads are emitted with sc1 and
        // bypass the non-coherent per-XCD L2 for these addresses; s_waitcnt vmcnt(0) orders this wave's three stores
        // before its ticket.  Formally this is a relaxed hand-off; tests/test_gpu_bench_path.py::
        // test_reduce_loss_is_idempotent_and_matches_the_oracle runs it 20x against the oracle's loss.
        __hip_atomic_store(&parts[blk * 4 + 0], v0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&parts[blk * 4 + 1], v1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&parts[blk * 4 + 2], v2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(counter, 1u, ALIGNQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
        is_last = (tk == (unsigned)nblk - 1);
      }
    }
  }
  if (!LOSS) return false;
  __syncthreads();
  if (!is_last) return false;
  double s0 = 0, s1 = 0, s2 = 0;
  if ((int)threadIdx.x < nblk) {
    s0 = __hip_atomic_load(&parts[threadIdx.x * 4 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s1 = __hip_atomic_load(&parts[threadIdx.x * 4 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s2 = __hip_atomic_load(&parts[threadIdx.x * 4 + 2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  if (lane == 0) { fin[sg] = s0; fin[16 + sg] = s1; fin[32 + sg] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a0 = 0, a1 = 0, a2 = 0;
    for (int g = 0; g < 16; g++) { a0 += fin[g]; a1 += fin[16 + g]; a2 += fin[32 + g]; }
    const double n = (double)B * (double)B;
    const double rms = sqrt(a1 / n);
    // (the loss itself write-through: slab_reduce_multi_head_kernel's last site reads every site's loss inside the launch)
    __hip_atomic_store(&scal[0], (float)(mu * a0 / n + 0.5 * rho * rms + a2 / n), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    scal[1] = (float)(0.5 * rho / (n * rms));
    scal[2] = (float)(1.0 / n);
    scal[3] = (float)rms;
    // re-arm the ticket: the entry point is idempotent (a second reduction of the same workspace finds counter == 0 again;
    // the partials kernel also zeroes it, which covers the very first use of an uninitialised workspace)
    __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  return true;       // (block-uniform) this workgroup closed the reduction: its thread 0 has written scal
}

template <bool SYM, bool LOSS>
__global__ __launch_bounds__(1024) void slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs,
                                                           int slab_floats, int BP, int B, float scale,
                                                           float* __restrict__ out, const float* __restrict__ A,
                                                           const float* __restrict__ gamma, int dim, float mu,
                                                           float rho, float* __restrict__ parts,
                                                           unsigned* __restrict__ counter, float* __restrict__ scal) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSlabReduceLds<SYM>];
  slab_reduce_body<SYM, LOSS>(slabs, n_slabs, slab_floats, BP, B, scale, out, A, gamma, dim, mu, rho, parts, counter, scal,
                              blockIdx.x, gridDim.x, lds);
}

// All sites of a model in ONE launch (blockIdx.y = site): the per-site reductions are off the critical path of the
// network's forward (only x_q feeds the next layer), so a whole-model step defers them to the end of the forward.
constexpr int kMultiSites = 32;
struct RChunk {
  const float* slabs[kMultiSites];
  float* out[kMultiSites];
  const float* A[kMultiSites];
  const float* gamma[kMultiSites];
  float* scal[kMultiSites];
  float scale[kMultiSites];
  int n_slabs[kMultiSites];
};
__global__ __launch_bounds__(1024) void slab_reduce_multi_kernel(RChunk c, int B, int dim, float mu, float rho) {
  const int s = blockIdx.y;
  float* ws = const_cast<float*>(c.slabs[s]);
  float* parts = ws + (size_t)c.n_slabs[s] * kSlab4Floats;
  unsigned* counter = reinterpret_cast<unsigned*>(parts + kPartFloats);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSlabReduceLds<true>];
  slab_reduce_body<true, true>(c.slabs[s], c.n_slabs[s], kSlab4Floats, 128, B, c.scale[s], c.out[s], c.A[s], c.gamma[s], dim, mu,
                               rho, parts, counter, c.scal[s], blockIdx.x, gridDim.x, lds);
}

// The same launch with the classifier head's forward as a second ROLE (round 6: the head reads the last site's x_q, the reductions
// read the sites' slabs - neither needs the other, and each alone is a 9-10 us node of the step's chain).  Workgroups [0, cnt *
// kSlabRedBlocks): slab_reduce_body of site blk / kSlabRedBlocks, same partition as slab_reduce_multi_kernel (same bits); the rest:
// head_fwd_body, one workgroup per sample, first four waves.  The sum of ALL sites' losses (trans_total; the head's last workgroup
// formed it when the reductions were a launch of their own) is formed by the workgroup that closes the LAST of this launch's sites:
// second-level ticket over the cnt sites, hand-off as in slab_reduce_body (write-through loss, drain, relaxed agent-scope ticket,
// agent-scope loads by the last arriver); the sites reduced by earlier launches are visible by launch order.
struct HeadFwd {
  const float* feat; const float* W; const float* bias; const int64_t* target; int HW, C, K, B;
  float* pooled; float* logits; float* probs; float* loss; float* ce_mean; unsigned* counter;
};
struct TransTail {
  const float* scal_all; int n_sites; float* trans_total; unsigned* counter;
};
__global__ __launch_bounds__(1024) void slab_reduce_multi_head_kernel(RChunk c, int B, int dim, float mu, float rho, int cnt, HeadFwd h,
                                                                      TransTail tt) {
  const int nred = cnt * kSlabRedBlocks;
  if ((int)blockIdx.x >= nred) {          // block-uniform: the head role
    if (threadIdx.x >= 256) return;
    alignq_head::head_fwd_body(h.feat, h.W, h.bias, h.target, h.HW, h.C, h.K, h.pooled, h.logits, h.probs, h.loss, h.ce_mean, h.counter,
                               nullptr, 0, nullptr, (int)blockIdx.x - nred, h.B);
    return;
  }
  const int s = blockIdx.x / kSlabRedBlocks, blk = blockIdx.x - s * kSlabRedBlocks;
  float* ws = const_cast<float*>(c.slabs[s]);
  float* parts = ws + (size_t)c.n_slabs[s] * kSlab4Floats;
  unsigned* counter = reinterpret_cast<unsigned*>(parts + kPartFloats);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSlabReduceLds<true>];
  __shared__ int all_sites;
  const bool closed = slab_reduce_body<true, true>(c.slabs[s], c.n_slabs[s], kSlab4Floats, 128, B, c.scale[s], c.out[s], c.A[s], c.gamma[s],
                                                   dim, mu, rho, parts, counter, c.scal[s], blk, kSlabRedBlocks, lds);
  if (!closed) return;
  if (threadIdx.x == 0) {                 // (the thread that stored this site's loss)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned tk = __hip_atomic_fetch_add(tt.counter, 1u, ALIGNQ_TICKET_ORDER, __HIP_MEMORY_SCOPE_AGENT);
    all_sites = (tk == (unsigned)cnt - 1);
  }
  __syncthreads();
  if (!all_sites || threadIdx.x >= 64) return;
  double t = 0.0;
  for (int i = threadIdx.x; i < tt.n_sites; i += 64)       // scal = {loss, c_con, 1/n, rms} per site
    t += (double)__hip_atomic_load(&tt.scal_all[4 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  t = alignq::wave_sum_d_dpp(t);
  if (threadIdx.x == 0) {
    tt.trans_total[0] = (float)t;
    __hip_atomic_store(tt.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm
  }
}

// The B <= 64 geometries (full [BP][BP] slabs) for several batch slices of one site in ONE launch (blockIdx.y = group; the
// slices' workspace regions lie ws_gstride floats apart, D and scal are [groups][B][B] / [groups][4]): the merged source +
// target pass of the Office step.
__global__ __launch_bounds__(1024) void slab_reduce_groups_kernel(float* __restrict__ ws0, int n_slabs, int slab_floats, int BP,
                                                                  int B, float scale, float* __restrict__ out,
                                                                  const float* __restrict__ A, const float* __restrict__ gamma,
                                                                  int dim, float mu, float rho, float* __restrict__ scal,
                                                                  int64_t ws_gstride) {
  const int64_t gi = blockIdx.y;
  float* ws = ws0 + gi * ws_gstride;
  float* parts = ws + (size_t)n_slabs * slab_floats;
  unsigned* counter = reinterpret_cast<unsigned*>(parts + kPartFloats);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSlabReduceLds<false>];
  slab_reduce_body<false, true>(ws, n_slabs, slab_floats, BP, B, scale, out + gi * B * B, A, gamma, dim, mu, rho, parts, counter,
                                scal + gi * 4, blockIdx.x, gridDim.x, lds);
}

// The same for SEVERAL sites in one launch (round 5: the Office step's 16 bottleneck tails leave their reductions to the end of the
// forward, as the CIFAR step's sites do): blockIdx.y = site * groups + slice.  Same body, same workgroup partition and
// summation order as slab_reduce_groups_kernel => same bits.
struct GRChunk {
  float* ws[kMultiSites];
  float* D[kMultiSites];
  const float* A[kMultiSites];
  const float* gamma[kMultiSites];
  float* scal[kMultiSites];
  int64_t ws_gstride[kMultiSites];
  float scale[kMultiSites];
  int n_slabs[kMultiSites];
};
__global__ __launch_bounds__(1024) void slab_reduce_groups_multi_kernel(GRChunk c, int groups, int slab_floats, int BP, int B, int dim,
                                                                        float mu, float rho) {
  const int s = blockIdx.y / groups;
  const int64_t gi = blockIdx.y - s * groups;
  float* ws = c.ws[s] + gi * c.ws_gstride[s];
  float* parts = ws + (size_t)c.n_slabs[s] * slab_floats;
  unsigned* counter = reinterpret_cast<unsigned*>(parts + kPartFloats);
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSlabReduceLds<false>];
  slab_reduce_body<false, true>(ws, c.n_slabs[s], slab_floats, BP, B, c.scale[s], c.D[s] + gi * B * B, c.A[s], c.gamma[s], dim, mu, rho,
                                parts, counter, c.scal[s] + gi * 4, blockIdx.x, gridDim.x, lds);
}

// ================================================================================================ backward prep
// S = (gD + gD^T) * gscale / F  [B,B]   (MFMA A operand of the backward), and (FUSED) the scaled parameter
// gradients dA_out = gscale*(mu*sign(A)/n - gD), dG_out = gscale*|D-A|/n  (zero outside [:B,:B]).
//   FUSED : gD = c_con*(D-A) + gamma*sign(D-A)/n from the forward's scal = {loss, c_con, 1/n, rms}
//   !FUSED: gD = dD (explicit upstream gradient)
// S image for the B in (64,128] backward (site_bwd4_kernel's MFMA A operand), written next to the fp32 S at byte offset
// kSImageOffset of the S buffer: every element split ONCE into bf16 hi / lo (the backward used to split its 64 elements per
// lane in every workgroup: ~400 VALU instructions per lane of a kernel that issues ~1100), in fragment order
//   img[ks][h][i][0..7] = hi(S[i][16 ks + 8 h + 0..7]),  img[ks][h][i][8..15] = lo(...)        (32 B per (ks, h, i))
// so that the 32 lanes l31 of a half-wave read 1 KB contiguously.  Rows / columns >= B are zero.
constexpr int kSImageOffset = 128 * 128 * 4;      // bytes; the fp32 S occupies at most the first 64 KB
__device__ __forceinline__ void s_image_store(float* __restrict__ S, int i, int j, float v) {
  __bf16* img = reinterpret_cast<__bf16*>(reinterpret_cast<char*>(S) + kSImageOffset);
  const __bf16 hi = (__bf16)v;
  const __bf16 lo = (__bf16)(v - (float)hi);
  const int o = (((j >> 4) * 2 + ((j >> 3) & 1)) * 128 + i) * 16 + (j & 7);
  img[o] = hi;
  img[o + 8] = lo;
}

template <bool FUSED>
__device__ __forceinline__ void site_prep_body(const float* __restrict__ dD, const float* __restrict__ D,
                                                        const float* __restrict__ A, const float* __restrict__ gamma,
                                                        int dim, const float* __restrict__ scal, float mu,
                                                        const float* __restrict__ gscale, int B, float invF,
                                                        float* __restrict__ S, float* __restrict__ dA_out,
                                                        float* __restrict__ dG_out, int bx, int gx) {
  // bx / gx: index and count of the 256-thread workgroups working on THIS site (a launch may carry other roles)
  const float gs = gscale ? gscale[0] : 1.0f;
  const int total = FUSED ? dim * dim : B * B;
  const int side = FUSED ? dim : B;
  const float c_con = FUSED ? scal[1] : 0.f, inv_n = FUSED ? scal[2] : 0.f;
  const bool image = B > 64;                 // only the B in (64,128] backward reads it
  for (int e = bx * 256 + threadIdx.x; e < total; e += gx * 256) {
    const int i = e / side, j = e - i * side;
    if (i < B && j < B) {
      float gij, gji;
      if (FUSED) {
        const float a = A[i * dim + j], gm = gamma[i * dim + j];
        const float d = D[i * B + j] - a;
        gij = c_con * d + gm * (float)((d > 0.f) - (d < 0.f)) * inv_n;
        const float a2 = A[j * dim + i], gm2 = gamma[j * dim + i];
        const float d2 = D[j * B + i] - a2;
        gji = c_con * d2 + gm2 * (float)((d2 > 0.f) - (d2 < 0.f)) * inv_n;
        if (dA_out) dA_out[e] = gs * (mu * (float)((a > 0.f) - (a < 0.f)) * inv_n - gij);
        if (dG_out) dG_out[e] = gs * fabsf(d) * inv_n;
      } else {
        gij = dD[i * B + j];
        gji = dD[j * B + i];
      }
      const float sv = (gij + gji) * gs * invF;
      S[i * B + j] = sv;
      if (image) s_image_store(S, i, j, sv);
    } else if (FUSED) {
      if (dA_out) dA_out[e] = 0.f;
      if (dG_out) dG_out[e] = 0.f;
    }
  }
  if (image && B < 128) {                    // zero padding of the image (rows / columns B..127)
    for (int e = bx * 256 + threadIdx.x; e < 128 * 128; e += gx * 256) {
      const int i = e >> 7, j = e & 127;
      if (i >= B || j >= B) s_image_store(S, i, j, 0.0f);
    }
  }
}

template <bool FUSED>
__global__ __launch_bounds__(256) void site_prep_kernel(const float* __restrict__ dD, const float* __restrict__ D,
                                                        const float* __restrict__ A, const float* __restrict__ gamma,
                                                        int dim, const float* __restrict__ scal, float mu,
                                                        const float* __restrict__ gscale, int B, float invF,
                                                        float* __restrict__ S, float* __restrict__ dA_out,
                                                        float* __restrict__ dG_out) {
  site_prep_body<FUSED>(dD, D, A, gamma, dim, scal, mu, gscale, B, invF, S, dA_out, dG_out, blockIdx.x, gridDim.x);
}

struct PChunk {
  const float* D[kMultiSites];
  const float* A[kMultiSites];
  const float* gamma[kMultiSites];
  const float* scal[kMultiSites];
  float* S[kMultiSites];
  float* dA[kMultiSites];
  float* dG[kMultiSites];
  float invF[kMultiSites];
};
__global__ __launch_bounds__(256) void site_prep_multi_kernel(PChunk c, int dim, float mu, const float* __restrict__ gscale,
                                                              int B) {
  const int s = blockIdx.y;
  site_prep_body<true>(nullptr, c.D[s], c.A[s], c.gamma[s], dim, c.scal[s], mu, gscale, B, c.invF[s], c.S[s], c.dA[s],
                       c.dG[s], blockIdx.x, gridDim.x);
}

// The B <= 32 site's preparation for `groups` batch slices that share ONE ADMM module (the Office step's merged source + target
// pass): S of every slice (blockIdx.y = slice) and - by the workgroups of slice 0 - the parameter gradients of ALL slices added
// in slice order (dalterD = sum_g dalterD_g: the sum autograd forms when the module is called once per pass; round 4: it was an
// elementwise launch of its own behind the per-slice gradients).  D: [groups][B][B], scal: [groups][4], S regions s_gstride floats apart.
__device__ __forceinline__ void site_prep_groups_body(const float* __restrict__ D, const float* __restrict__ A,
                                                      const float* __restrict__ gamma, int dim,
                                                      const float* __restrict__ scal, float mu,
                                                      const float* __restrict__ gscale, int B, float invF,
                                                      float* __restrict__ S, int64_t s_gstride,
                                                      float* __restrict__ dA_out, float* __restrict__ dG_out, int groups,
                                                      int gs_stride, int gi) {
  // gs_stride: elements between the slices' upstream loss gradients (0: one scalar for all of them)
  // S of this slice (no parameter gradients from here: dA_out / dG_out nullptr)
  site_prep_body<true>(nullptr, D + (int64_t)gi * B * B, A, gamma, dim, scal + 4 * gi, mu, gscale ? gscale + (int64_t)gi * gs_stride : nullptr,
                       B, invF, S + gi * s_gstride, nullptr, nullptr, blockIdx.x, gridDim.x);
  if (gi != 0 || (!dA_out && !dG_out)) return;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < dim * dim; e += gridDim.x * 256) {
    const int i = e / dim, j = e - i * dim;
    float da = 0.f, dg = 0.f;
    if (i < B && j < B) {
      const float a = A[e], gm = gamma[e];
      const float sa = (float)((a > 0.f) - (a < 0.f));
      for (int g = 0; g < groups; g++) {                      // slice order: the same sum as pass after pass
        const float c_con = scal[4 * g + 1], inv_n = scal[4 * g + 2];
        const float d = D[(int64_t)g * B * B + i * B + j] - a;
        const float gij = c_con * d + gm * (float)((d > 0.f) - (d < 0.f)) * inv_n;
        const float gs = gscale ? gscale[(int64_t)g * gs_stride] : 1.0f;
        const float va = gs * (mu * sa * inv_n - gij), vg = gs * fabsf(d) * inv_n;
        da = g == 0 ? va : da + va;
        dg = g == 0 ? vg : dg + vg;
      }
    }
    if (dA_out) dA_out[e] = da;
    if (dG_out) dG_out[e] = dg;
  }
}
__global__ __launch_bounds__(256) void site_prep_groups_kernel(const float* __restrict__ D, const float* __restrict__ A,
                                                               const float* __restrict__ gamma, int dim,
                                                               const float* __restrict__ scal, float mu,
                                                               const float* __restrict__ gscale, int B, float invF,
                                                               float* __restrict__ S, int64_t s_gstride,
                                                               float* __restrict__ dA_out, float* __restrict__ dG_out, int groups,
                                                               int gs_stride) {
  site_prep_groups_body(D, A, gamma, dim, scal, mu, gscale, B, invF, S, s_gstride, dA_out, dG_out, groups, gs_stride, blockIdx.y);
}
// ... of SEVERAL sites in one launch (blockIdx.y = site * groups + slice; one upstream scalar for every site and slice: the
// gradient of the sum of all their losses)
__global__ __launch_bounds__(256) void site_prep_groups_multi_kernel(PChunk c, int groups, int dim, float mu,
                                                                     const float* __restrict__ gscale, int B, int64_t s_gstride) {
  const int s = blockIdx.y / groups, gi = blockIdx.y - s * groups;
  site_prep_groups_body(c.D[s], c.A[s], c.gamma[s], dim, c.scal[s], mu, gscale, B, c.invF[s], c.S[s], s_gstride, c.dA[s], c.dG[s],
                        groups, 0, gi);
}

// The classifier head's backward and the preparation of every site's S / dalterD / dgamma in ONE launch (two independent roles:
// the first n_head workgroups are alignq_head::head_bwd_body's B + K, the rest the sites' gx each).
struct HeadBwdArgs {
  const float* g; const float* probs; const int64_t* target; const float* pooled; const float* W;
  int B, HW, C, K;
  float* dfeat; float* dW; float* dbias;
};
__global__ __launch_bounds__(256) void head_bwd_prep_multi_kernel(HeadBwdArgs h, int n_head, PChunk c, int dim, float mu,
                                                                  const float* __restrict__ gscale, int B, int gx) {
  const int b = blockIdx.x;
  if (b < n_head) {
    alignq_head::head_bwd_body(h.g, h.probs, h.target, h.pooled, h.W, h.B, h.HW, h.C, h.K, h.dfeat, h.dW, h.dbias, b);
  } else {
    const int r = b - n_head, s = r / gx, bx = r - s * gx;
    site_prep_body<true>(nullptr, c.D[s], c.A[s], c.gamma[s], dim, c.scal[s], mu, gscale, B, c.invF[s], c.S[s], c.dA[s],
                         c.dG[s], bx, gx);
  }
}

// ================================================================================================ backward
// 512 threads = 8 waves; wave w: row block I = w>>1 (32 batch rows of dVh), column block cj = w&1 of the 64-feature
// tile, BOTH operands, so the final assembly is wave-local.
// dVh = S * Vh on split-bf16 MFMA (same 3-term scheme as the forward; relative error 5e-6, tolerance 1e-4):
//   A = S[i][j] (K = batch index j): 8 k-steps x (hi, lo) fragments live in registers for the whole kernel;
//   B = Vh[j][c] needs 8 CONSECUTIVE j per lane, so the standardised tiles are staged TRANSPOSED in LDS
//       ([feature][batch], bf16 hi/lo): the load mapping gives each thread one feature column and 16 consecutive
//       batch rows (256-byte coalesced row segments per wave instruction), two 16-byte LDS stores per array.
// TFv = 64: 512 threads, one workgroup per CU (138 KB LDS);  TFv = 32: 256 threads, 69 KB LDS, twice as many tiles for the small-F
// sites; launched ONE workgroup per CU (kBwd32OnePerCuLds at the launcher: two co-resident ones were not reproducible to the bit).
// VEC (F % 4 == 0, 16-byte aligned tensors): a thread owns FOUR feature columns x FOUR batch rows and moves them with 16-byte
// global accesses (12 loads + 4-8 stores per thread instead of 48 + 16-32 dword ones: the dword form kept the texture
// addresser busy for ~4 us per launch at F = 16384); the transposed staging is then one 8-byte LDS store per column and
// array.  !VEC: one column x 16 rows per thread, dword accesses with clamped offsets (any F, any alignment).
// LOOP (plain site, VEC, 64-feature tiles, many tiles per CU): a workgroup walks over tiles and requests the next tile's x / g
// rows as soon as this tile's are staged, so they fly under the MFMA, projection, assembly and copy-out phases.
// Twin launch (round 6; 32-feature one-tile form only): the backward of TWO sites of one shape - the workgroups from `split` on work on
// the second site's tensors (see Twin at the forward kernel).  No filler role in such a launch.
struct BwdTwin {
  const float* gup; const float* S; const float* x; const float* stats; float* dx; BnFold bn; int split;
};
template <int TFv, bool PAIR, bool BN, bool VEC, bool LOOP = false, bool FULLP = false>
__global__ __launch_bounds__(TFv * 8, 2) void site_bwd4_kernel(const float* __restrict__ gup, const float* __restrict__ S,
                                                        const float* __restrict__ x, const float* __restrict__ stats,
                                                        int B, int64_t F, float r, float eps, float* __restrict__ dx,
                                                        int n_tiles, int aligned, BnFold bn, alignq_wgr::RedFill fill, BwdTwin twin) {
  BSTAMP(1, 0);
  (void)aligned;
  int bid = blockIdx.x;
  [[maybe_unused]] int dsite = 0;
  if constexpr (!LOOP && TFv == 32) {
    if (twin.split && bid >= twin.split) {        // block-uniform: the second site of a twin launch
      dsite = 1;
      bid -= twin.split;
      gup = twin.gup; S = twin.S; x = twin.x; stats = twin.stats; dx = twin.dx; bn = twin.bn;
    }
  }
  constexpr int LDv = TFv + 4, TILE = 128 * LDv;    // fp32 tiles: 16-byte aligned rows (copied out with 16-byte LDS reads)
  constexpr int NWv = TFv / 8;                       // waves per workgroup
  constexpr int CB = TFv / 32;                       // 32-column blocks per tile
  constexpr int LDT = 128 + 8;                       // bf16 elements per transposed row (272 B: 16-B aligned, 4-bank skew)
  constexpr int TARR = TFv * LDT;                    // bf16 elements per transposed array
  constexpr int NT_ARR = PAIR ? 4 : 2;
  constexpr int NF_ARR = PAIR ? 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[NT_ARR * TARR * 2 + NF_ARR * TILE * 4 + 4 * 2 * 2 * TFv * 4];
  if constexpr (!LOOP && TFv == 32) {
    // filler workgroups (block-uniform; the narrow sites' launches leave half the chip idle): filter-gradient slab reductions
    // of convolutions whose backward already ran, see wgrad_reduce_body.h
    if (bid >= n_tiles) {
      const int fb = bid - n_tiles;
      int it = 0;
      while (it + 1 < alignq_wgr::kFill && fb >= fill.blk0[it + 1]) it++;
      alignq_wgr::wgrad_reduce_body<256>(fill.slabs[it], fill.n_slabs[it], fill.n_elem[it], fill.dw[it], fb - fill.blk0[it],
                                         reinterpret_cast<float*>(lds_raw));
      return;
    }
  }
  __bf16* XThi = reinterpret_cast<__bf16*>(lds_raw);
  __bf16* XTlo = XThi + TARR;
  __bf16* TThi = XThi + 2 * TARR;                    // (PAIR)
  __bf16* TTlo = XThi + 3 * TARR;                    // (PAIR)
  // dt/dx goes to the accumulator layout through Js (16-byte stores), jac*ct - cx comes back through Os (16-byte reads); g * dt/dx
  // stays in the registers of the thread that loaded the element (round 3: the tiles used to carry g*jac too, with dword
  // stores that collided 4-way, and the sum was formed by a read-modify-write in LDS)
  float* Os = reinterpret_cast<float*>(lds_raw + NT_ARR * TARR * 2);   // [128][LDv] jac*ct - cx (PAIR) | cx
  // Js = Os + TILE: [128][LDv] dt/dx (PAIR), addressed as os_acc / os_ld + kJs below
  float* red = Os + NF_ARR * TILE;                                     // [4 row blocks][2 operands][2][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: I, cj and the LDS bases derived from it in SGPRs
  const int h = lane >> 5, l31 = lane & 31;
  const int I = w / CB, cj = w % CB;
  const int cc = cj * 32 + l31;            // this lane's feature column inside the tile (accumulator layout)
  const int lcol = tid % TFv, lrow0 = (tid / TFv) * 16;   // !VEC load mapping: one feature column, 16 consecutive batch rows
  constexpr int NCQ = TFv / 4;                             // VEC load mapping: column quad lc4, rows lrow4 .. lrow4 + 3
  const int lc4 = tid % NCQ, lrow4 = (tid / NCQ) * 4;

  const float invB = 1.0f / (float)B, invBm1 = 1.0f / (float)(B - 1);
  const float rjac = r * ALIGNQ_TWO_OVER_SQRT_2PI;
  // Os / Js as this thread sees them: accumulator layout (rows (I*32 + 4h) + {0..3} + 8 g4, column cc) and load layout
  const uint32_t os_acc = lds_base(Os + (I * 32 + 4 * h) * LDv + cc);
  const uint32_t os_ld = lds_base(VEC ? Os + lrow4 * LDv + 4 * lc4 : Os + lrow0 * LDv + lcol);
  constexpr uint32_t kJs = TILE * 4, kRow = LDv * 4;       // byte offsets: Os -> Js, row -> row + 1
  STAMP(10);

  float4 xr[4], gr[4];                    // VEC: this thread's x / g quads (LOOP: refilled one tile ahead)
  bf16x8 sh[8], sl[8];                    // S fragments
  // (round 3: requesting the 64-feature form's S fragments FIRST, under the tile loads - it has the registers: one workgroup
  //  per CU, 256-register budget - changed nothing: 1.172 / 1.176 / 1.173 against 1.173 / 1.168 / 1.174 ms per step)
  // LOOP: the per-column statistics (mean x, 1/std x, mean t, 1/std t) of a tile travel one tile ahead too, through a small
  // double-buffered LDS image [2][4][TFv]: loaded at the top of a tile they would expose a full memory latency per tile
  __shared__ __attribute__((aligned(16))) float stt[LOOP ? 2 * 4 * TFv : 4];
  const int st_arr = (tid / NCQ) & 3, st_q = tid % NCQ;       // the float4 of the image this thread fetches (waves 1.. repeat wave 0)
  if constexpr (LOOP) {     // the first tile's rows (grid <= n_tiles) and statistics
    if (tid < 4 * NCQ)
      *reinterpret_cast<float4*>(stt + st_arr * TFv + 4 * st_q) =
          *reinterpret_cast<const float4*>(stats + (int64_t)st_arr * F + (int)blockIdx.x * TFv + 4 * st_q);
    __syncthreads();
    const bool has_g = PAIR && gup != nullptr;
    const char* gsrc = reinterpret_cast<const char*>(has_g ? gup : x);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned off = (unsigned)(lrow4 + q) * ((unsigned)F * 4u) + (unsigned)((int)blockIdx.x * TFv + 4 * lc4) * 4u;
      xr[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(x) + off);
      gr[q] = *reinterpret_cast<const float4*>(gsrc + off);
      if (!has_g) gr[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // !LOOP: one tile per workgroup (grid == n_tiles): no tile loop, nothing to hoist
  int st_buf = 0;
  for (int tile = LOOP ? (int)blockIdx.x : bid; tile < n_tiles; tile += (int)gridDim.x) {
    const int col0 = tile * TFv;
    const bool lcol_ok = (col0 + lcol) < F;
    float gj[PAIR ? 16 : 1];   // g * dt/dx of this thread's 16 elements (VEC: [4*row + column])
    float4 xn[LOOP ? 4 : 1], gn[LOOP ? 4 : 1];       // (LOOP) the next tile's rows, in flight for the whole of this tile
    float rho_x = 0.f, rho_t = 0.f;                  // 1/std of this lane's accumulator column
    float4 st_next = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (LOOP) {      // requested BEFORE the next tile's rows: the wait for it (vmcnt is in order) leaves those in flight
      const int tn = min(tile + (int)gridDim.x, n_tiles - 1);
      st_next = *reinterpret_cast<const float4*>(stats + (int64_t)st_arr * F + tn * TFv + 4 * st_q);
      rho_x = stt[(st_buf * 4 + 1) * TFv + cc];
      if (PAIR) rho_t = stt[(st_buf * 4 + 3) * TFv + cc];
    }

    // Addressing: kernel-argument base (SGPR pair) + one 32-bit byte offset per element, shared by x / g / y / dx /
    // dres (global_load saddr+voffset form; the launcher guarantees B*F*4 < 2^32).  Loads use offsets CLAMPED into the
    // tensor and are unconditional (a conditional load costs a branch each); out-of-range lanes are zeroed afterwards.
    const unsigned rowB = (unsigned)F * 4u;
    const unsigned colc = (unsigned)(lcol_ok ? col0 + lcol : (int)F - 1) * 4u;
    unsigned boff[16];
#pragma unroll
    for (int q = 0; q < 16; q++) boff[q] = (unsigned)min(lrow0 + q, B - 1) * rowB + colc;
#define AT(ptr, q) (*reinterpret_cast<const float*>(reinterpret_cast<const char*>(ptr) + boff[q]))
#define ATW(ptr, q) (*reinterpret_cast<float*>(reinterpret_cast<char*>(ptr) + boff[q]))
    // VEC addressing: one 32-bit byte offset per row for the thread's column quad (clamped like the dword form)
    // LOOP: only complete tiles (B == 128, F % TFv == 0 - the launcher's condition): no masks, no branches around memory
    // operations (a conditional load or store makes the compiler wait for EVERYTHING in flight at the next use of a register
    // that was loaded before it: the counters are in order, and it must assume the younger operation was not issued)
    // FULLP: the launcher's promise of complete tiles for the one-tile VEC form (B == 128, F % TFv == 0): the masks fold away
    static_assert(!FULLP || (VEC && !LOOP), "FULLP names the one-tile VEC form over complete tiles");
    static_assert(!LOOP || (VEC && !BN), "the looped form is the plain site's: 16-byte accesses, no batch-norm fold");
    constexpr bool kFullB = LOOP || FULLP;
    const bool q_ok = kFullB || (col0 + 4 * lc4) < F;          // F % 4 == 0: a quad lies inside or outside as a whole
    const unsigned colq = (unsigned)(q_ok ? col0 + 4 * lc4 : (int)F - 4) * 4u;
    const uint4 bo4 = make_uint4((unsigned)min(lrow4 + 0, B - 1) * rowB + colq, (unsigned)min(lrow4 + 1, B - 1) * rowB + colq,
                                 (unsigned)min(lrow4 + 2, B - 1) * rowB + colq, (unsigned)min(lrow4 + 3, B - 1) * rowB + colq);
#define BO4(q) ((q) == 0 ? bo4.x : ((q) == 1 ? bo4.y : ((q) == 2 ? bo4.z : bo4.w)))
#define AT4(ptr, q) (*reinterpret_cast<const float4*>(reinterpret_cast<const char*>(ptr) + BO4(q)))
#define ATW4(ptr, q) (*reinterpret_cast<float4*>(reinterpret_cast<char*>(ptr) + BO4(q)))
    if constexpr (VEC) {
      // ---- load x / g / y quads of 4 rows, recompute t / jac, standardise, stage transposed ---------------------------
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const int fq = col0 + 4 * lc4;
      // (clamped address + value select: `ok ? *p : zero` would be turned into a select of POINTERS with the zero on the stack)
      const int fqc = q_ok ? fq : 0;
      const bool has_g = PAIR && gup != nullptr;
      // Request order of the one-tile form: the tile's rows first (x, g: unconditional - without an upstream gradient the x
      // rows stand in and are dropped), then the per-column statistics and (a, b), the ReLU mask's source last, as RAW words
      // (fp32 y, or the 2- / 1-byte level indices: one launch-uniform branch, all four requests of a kind together - the
      // compiler turns them into predicates on the spot, which waits for everything older, so nothing may follow them).
      float4 yraw[(!LOOP && BN) ? 4 : 1];
      uint2 ybraw[(!LOOP && BN) ? 4 : 1];
      int ykind = 0;                       // 0 none, 1 fp32 y, 2 int16 indices, 3 int8 indices
      if constexpr (!LOOP) {
        const char* gsrc = reinterpret_cast<const char*>(has_g ? gup : x);
#pragma unroll
        for (int q = 0; q < 4; q++) xr[q] = AT4(x, q);
#pragma unroll
        for (int q = 0; q < 4; q++) gr[q] = *reinterpret_cast<const float4*>(gsrc + BO4(q));
      }
      float4 mx4, rx4, mt4 = z4, rt4 = z4;
      if constexpr (LOOP) {
        mx4 = *reinterpret_cast<const float4*>(stt + (st_buf * 4 + 0) * TFv + 4 * lc4);
        rx4 = *reinterpret_cast<const float4*>(stt + (st_buf * 4 + 1) * TFv + 4 * lc4);
        if (PAIR) {
          mt4 = *reinterpret_cast<const float4*>(stt + (st_buf * 4 + 2) * TFv + 4 * lc4);
          rt4 = *reinterpret_cast<const float4*>(stt + (st_buf * 4 + 3) * TFv + 4 * lc4);
        }
      } else {
        mx4 = *reinterpret_cast<const float4*>(stats + fqc);
        rx4 = *reinterpret_cast<const float4*>(stats + F + fqc);
        if (PAIR) {
          mt4 = *reinterpret_cast<const float4*>(stats + 2 * F + fqc);
          rt4 = *reinterpret_cast<const float4*>(stats + 3 * F + fqc);
        }
      }
      float4 a4 = make_float4(1.f, 1.f, 1.f, 1.f), b4 = z4;
      if (BN && q_ok) {
        if (bn.nhwc) {             // channels-last: the quad covers 4 consecutive channels (C % 4 == 0)
          const int ch = fq & (bn.C - 1);
          a4 = *reinterpret_cast<const float4*>(bn.ab + ch);
          b4 = *reinterpret_cast<const float4*>(bn.ab + bn.C + ch);
        } else {                   // one channel for the whole quad (HW % 4 == 0)
          const int ch = fq / bn.HW;
          const float av = bn.ab[ch], bv = bn.ab[bn.C + ch];
          a4 = make_float4(av, av, av, av);
          b4 = make_float4(bv, bv, bv, bv);
        }
      }
      if constexpr (!LOOP) {
        if constexpr (BN) {
          ykind = bn.y ? 1 : (bn.ybins ? (bn.bin_bytes == 2 ? 2 : 3) : 0);
          if (ykind == 1) {
#pragma unroll
            for (int q = 0; q < 4; q++) yraw[q] = AT4(bn.y, q);
          } else if (ykind == 2) {
#pragma unroll
            for (int q = 0; q < 4; q++) ybraw[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(bn.ybins) + (BO4(q) >> 1));
          } else if (ykind == 3) {
#pragma unroll
            for (int q = 0; q < 4; q++) ybraw[q].x = *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(bn.ybins) + (BO4(q) >> 2));
          }
        }
        if (!has_g) {
#pragma unroll
          for (int q = 0; q < 4; q++) gr[q] = z4;
        }
      }
      // LOOP, software pipeline: the NEXT tile's rows are requested before this tile's are touched (read once: non-temporal),
      // into registers of their own - a whole tile period in flight, so a CU always has 64 KB on request (requested after
      // the staging, into the same registers, they had only the MFMA .. copy-out phases to arrive: the staging waited ~3 us)
      if constexpr (LOOP) {
        // always issued (the last iteration re-reads its own tile; without an upstream gradient the x rows stand in for g's)
        const int tn = min(tile + (int)gridDim.x, n_tiles - 1);
        const unsigned colqn = (unsigned)(tn * TFv + 4 * lc4) * 4u;
        const char* gsrc = reinterpret_cast<const char*>(has_g ? gup : x);
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const unsigned off = (unsigned)(lrow4 + q) * rowB + colqn;
          const f32x4_nt xv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(reinterpret_cast<const char*>(x) + off));
          const f32x4_nt gv = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(gsrc + off));
          xn[q] = make_float4(xv.x, xv.y, xv.z, xv.w);
          gn[q] = make_float4(gv.x, gv.y, gv.z, gv.w);
        }
      }
      if (!q_ok) { mx4 = z4; rx4 = z4; mt4 = z4; rt4 = z4; }      // (a use: behind every request)
      if constexpr (BN && !LOOP) {
        // the mask as four booleans per row (N2: the stored level index is positive exactly where y is: sign tests on the
        // packed words, no conversion)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          xr[q].x = __fmaf_rn(a4.x, xr[q].x, b4.x); xr[q].y = __fmaf_rn(a4.y, xr[q].y, b4.y);
          xr[q].z = __fmaf_rn(a4.z, xr[q].z, b4.z); xr[q].w = __fmaf_rn(a4.w, xr[q].w, b4.w);
          if (ykind != 0) {                                          // fused ReLU backward
            bool p0, p1, p2, p3;
            if (ykind == 1) {
              p0 = yraw[q].x > 0.0f; p1 = yraw[q].y > 0.0f; p2 = yraw[q].z > 0.0f; p3 = yraw[q].w > 0.0f;
            } else if (ykind == 2) {
              p0 = (short)(ybraw[q].x & 0xffffu) > 0; p1 = (short)(ybraw[q].x >> 16) > 0;
              p2 = (short)(ybraw[q].y & 0xffffu) > 0; p3 = (short)(ybraw[q].y >> 16) > 0;
            } else {
              const unsigned wv = ybraw[q].x;
              p0 = (signed char)(wv & 0xffu) > 0; p1 = (signed char)((wv >> 8) & 0xffu) > 0;
              p2 = (signed char)((wv >> 16) & 0xffu) > 0; p3 = (signed char)(wv >> 24) > 0;
            }
            gr[q].x = p0 ? gr[q].x : 0.0f; gr[q].y = p1 ? gr[q].y : 0.0f;
            gr[q].z = p2 ? gr[q].z : 0.0f; gr[q].w = p3 ? gr[q].w : 0.0f;
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (!kFullB && !(q_ok && (lrow4 + q) < B)) { xr[q] = z4; gr[q] = z4; }
      }
      if (BN && bn.dres && q_ok) {            // the masked gradient is also the residual branch's gradient
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (kFullB || lrow4 + q < B) ATW4(bn.dres, q) = gr[q];
      }
      float jt[PAIR ? 16 : 1];                 // dt/dx, on its way to Js
#pragma unroll
      for (int e = 0; e < 4; e++) {            // column 4*lc4 + e: its 4 rows are 4 consecutive entries of a transposed row
        // (scalars + one braced vector construction per array: element-wise insertion into four live bf16x4 put them on the
        // stack in the PAIR instantiations)
        __bf16 xh0, xh1, xh2, xh3, xl0, xl1, xl2, xl3, th0, th1, th2, th3, tl0, tl1, tl2, tl3;
#define ALIGNQ_ROW(q, XH, XL, TH, TL)                                                              \
        {                                                                                          \
          const int row = lrow4 + q;                                                               \
          const bool ok = kFullB || (q_ok && row < B);                                             \
          const float xe = f4get(xr[q], e);                                                        \
          split_bf16(ok ? (xe - f4get(mx4, e)) * f4get(rx4, e) : 0.0f, XH, XL);                    \
          if (PAIR) {                                                                              \
            float t, jac;                                                                          \
            act_transform_rcp(xe, r, rjac, &t, &jac);                                         \
            split_bf16(ok ? (t - f4get(mt4, e)) * f4get(rt4, e) : 0.0f, TH, TL);                   \
            jt[PAIR ? 4 * q + e : 0] = jac;                                                        \
            gj[(PAIR ? 4 * q + e : 0)] = f4get(gr[q], e) * jac;                                    \
            if constexpr (TFv == 32 && !LOOP && ((ALIGNQ_DIAG_DUMP_BITS) & 1)) DUMP2(0, dsite, tile, row, 4 * lc4 + e, t, jac); \
            if constexpr (TFv == 32 && !LOOP && ((ALIGNQ_DIAG_DUMP_BITS) & 32)) {                  \
              if (q == 2 && e == 1) DUMP2(2, dsite, tile, 64 + row / 2, 4 * lc4 + e, t, (t - f4get(mt4, e)) * f4get(rt4, e)); \
            }                                                                                      \
          } else {                                                                                 \
            TH = (__bf16)0.0f; TL = (__bf16)0.0f;                                                  \
          }                                                                                        \
        }
        ALIGNQ_ROW(0, xh0, xl0, th0, tl0)
        ALIGNQ_ROW(1, xh1, xl1, th1, tl1)
        ALIGNQ_ROW(2, xh2, xl2, th2, tl2)
        ALIGNQ_ROW(3, xh3, xl3, th3, tl3)
#undef ALIGNQ_ROW
        // 8-byte aligned: LDT * 2 and lrow4 * 2 are multiples of 8.  The row index is XOR-swizzled per column (XSWZ above): the
        // 16 column quads of a wave's 8-byte stores would otherwise fall on one pair of the 32 write banks 8-way
        const int o = (4 * lc4 + e) * LDT + (lrow4 ^ XSWZ(4 * lc4 + e));
        *reinterpret_cast<bf16x4*>(XThi + o) = (bf16x4){xh0, xh1, xh2, xh3};
        *reinterpret_cast<bf16x4*>(XTlo + o) = (bf16x4){xl0, xl1, xl2, xl3};
        if (PAIR) {
          *reinterpret_cast<bf16x4*>(TThi + o) = (bf16x4){th0, th1, th2, th3};
          *reinterpret_cast<bf16x4*>(TTlo + o) = (bf16x4){tl0, tl1, tl2, tl3};
        }
      }
#ifdef ALIGNQ_DIAG_DUMP
      if constexpr (TFv == 32 && !LOOP && PAIR) {      // stage 6 (slot 3): the thread's own staged t operands, read back right behind its stores
        if ((ALIGNQ_DIAG_DUMP) & 64) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int o = (4 * lc4 + e) * LDT + (lrow4 ^ XSWZ(4 * lc4 + e));
            const uint2 hh = *reinterpret_cast<const uint2*>(TThi + o), ll = *reinterpret_cast<const uint2*>(TTlo + o);
            DUMP2(3, dsite, tile, lrow4 + 0, 4 * lc4 + e, bf16_pair(hh.x).x, bf16_pair(ll.x).x);
            DUMP2(3, dsite, tile, lrow4 + 1, 4 * lc4 + e, bf16_pair(hh.x).y, bf16_pair(ll.x).y);
            DUMP2(3, dsite, tile, lrow4 + 2, 4 * lc4 + e, bf16_pair(hh.y).x, bf16_pair(ll.y).x);
            DUMP2(3, dsite, tile, lrow4 + 3, 4 * lc4 + e, bf16_pair(hh.y).y, bf16_pair(ll.y).y);
          }
        }
      }
#endif
      if (PAIR) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          LDS_F32X4(os_ld + kJs + q * kRow) =
              f32x4_nt{jt[PAIR ? 4 * q : 0], jt[PAIR ? 4 * q + 1 : 0], jt[PAIR ? 4 * q + 2 : 0], jt[PAIR ? 4 * q + 3 : 0]};
      }
    } else
    // ---- load x (and g) for (feature lcol, rows lrow0..+15), recompute t / jac, standardise ---------------
    {
      const float mx = lcol_ok ? stats[col0 + lcol] : 0.f;
      const float rx = lcol_ok ? stats[F + col0 + lcol] : 0.f;
      const float mt = (PAIR && lcol_ok) ? stats[2 * F + col0 + lcol] : 0.f;
      const float rt = (PAIR && lcol_ok) ? stats[3 * F + col0 + lcol] : 0.f;
      float bn_a = 1.f, bn_b = 0.f;
      if (BN && lcol_ok) {
        const int ch = bn.nhwc ? ((col0 + lcol) & (bn.C - 1)) : (col0 + lcol) / bn.HW;
        bn_a = bn.ab[ch];
        bn_b = bn.ab[bn.C + ch];
      }
      float xr[16], gr[16];
#pragma unroll
      for (int q = 0; q < 16; q++) xr[q] = AT(x, q);          // all loads in flight before the first use
      const bool has_g = PAIR && gup != nullptr;
      if (has_g) {
#pragma unroll
        for (int q = 0; q < 16; q++) gr[q] = AT(gup, q);
      } else {
#pragma unroll
        for (int q = 0; q < 16; q++) gr[q] = 0.0f;
      }
      if (BN) {      // separate loops: a use inside a load loop would serialise the loads on their latency
        float yr[16];
        const bool masked = bn.y != nullptr || bn.ybins != nullptr;
        if (bn.y) {
#pragma unroll
          for (int q = 0; q < 16; q++) yr[q] = AT(bn.y, q);
        } else if (bn.ybins) {
#pragma unroll
          for (int q = 0; q < 16; q++) {
            const char* bp = reinterpret_cast<const char*>(bn.ybins);
            yr[q] = bn.bin_bytes == 2 ? (float)*reinterpret_cast<const short*>(bp + (boff[q] >> 1))
                                      : (float)*reinterpret_cast<const signed char*>(bp + (boff[q] >> 2));
          }
        }
#pragma unroll
        for (int q = 0; q < 16; q++) {
          xr[q] = __fmaf_rn(bn_a, xr[q], bn_b);
          if (masked) gr[q] = (yr[q] > 0.0f) ? gr[q] : 0.0f;     // fused ReLU backward
        }
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const bool ok = lcol_ok && (lrow0 + q) < B;
        xr[q] = ok ? xr[q] : 0.0f;
        gr[q] = ok ? gr[q] : 0.0f;
      }
      if (BN && bn.dres && lcol_ok) {         // the masked gradient is also the residual branch's gradient
#pragma unroll
        for (int q = 0; q < 16; q++) {
          const int row = lrow0 + q;
          if (row < B) ATW(bn.dres, q) = gr[q];
        }
      }
#pragma unroll
      for (int half = 0; half < 2; half++) {
        bf16x8 xh, xl, th, tl;
#pragma unroll
        for (int q = 0; q < 8; q++) {
          const int row = lrow0 + 8 * half + q;
          const bool ok = lcol_ok && row < B;
          const float xe = xr[8 * half + q];
          __bf16 a, b2;
          split_bf16(ok ? (xe - mx) * rx : 0.0f, a, b2);
          xh[q] = a; xl[q] = b2;
          if (PAIR) {
            float t, jac;
            act_transform_rcp(xe, r, rjac, &t, &jac);
            split_bf16(ok ? (t - mt) * rt : 0.0f, a, b2);
            th[q] = a; tl[q] = b2;
            LDS_F32(os_ld + kJs + (8 * half + q) * kRow) = jac;
            gj[PAIR ? 8 * half + q : 0] = gr[8 * half + q] * jac;
          }
        }
        const int o = lcol * LDT + ((lrow0 + 8 * half) ^ XSWZ(lcol));
        *reinterpret_cast<bf16x8*>(XThi + o) = xh;
        *reinterpret_cast<bf16x8*>(XTlo + o) = xl;
        if (PAIR) {
          *reinterpret_cast<bf16x8*>(TThi + o) = th;
          *reinterpret_cast<bf16x8*>(TTlo + o) = tl;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the 16 fragment loads below out of the register-hungry load phase
    // S fragments (already scaled, symmetric, split into bf16 hi / lo by the prep kernel: s_image_store): A[i][k],
    // i = I*32 + l31, k = 16*ks + 8h + jj.  Loaded here, after the load phase's registers are dead (a workgroup owns ONE
    // tile: grid == n_tiles); two 16-byte loads per k step, the 32 lanes of a half-wave read 1 KB contiguously.
    {     // (LOOP: fetched again for every tile - keeping the 64 registers alive across the load phase next to the prefetched
          //  rows overflowed the 256-register budget of the 8-wave workgroup: 36 B of scratch)
      const char* img = reinterpret_cast<const char*>(S) + kSImageOffset;
  #pragma unroll
      for (int ks = 0; ks < 8; ks++) {
        const char* q = img + ((((ks * 2 + h) * 128) + I * 32 + l31) * 32);
        sh[ks] = *reinterpret_cast<const bf16x8*>(q);
        sl[ks] = *reinterpret_cast<const bf16x8*>(q + 16);
      }
    }
    __syncthreads();
    STAMP(11);
    // ---- MFMA: accX = S[I-block, :] * Xh[:, column block], accT likewise (3 bf16 MFMAs per 16 batch rows) ---------
    f32x16 accX, accT;
#pragma unroll
    for (int e = 0; e < 16; e++) { accX[e] = 0.0f; accT[e] = 0.0f; }
#pragma unroll
    for (int ks = 0; ks < 8; ks++) {
      const int o = cc * LDT + ((16 * ks + 8 * h) ^ XSWZ(cc));
      const bf16x8 bxh = *reinterpret_cast<const bf16x8*>(XThi + o);
      const bf16x8 bxl = *reinterpret_cast<const bf16x8*>(XTlo + o);
      accX = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[ks], bxh, accX, 0, 0, 0);
      accX = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[ks], bxl, accX, 0, 0, 0);
      accX = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl[ks], bxh, accX, 0, 0, 0);
      if (PAIR) {
        const bf16x8 bth = *reinterpret_cast<const bf16x8*>(TThi + o);
        const bf16x8 btl = *reinterpret_cast<const bf16x8*>(TTlo + o);
        accT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[ks], bth, accT, 0, 0, 0);
        accT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sh[ks], btl, accT, 0, 0, 0);
        accT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(sl[ks], bth, accT, 0, 0, 0);
      }
      // (LOOP: next to the prefetched rows and the per-element factors there is no room for every k step's B operands at once)
      if constexpr (LOOP && PAIR) { if (ks & 1) __builtin_amdgcn_sched_barrier(0); }
    }
    STAMP(12);
    if constexpr (TFv == 32 && !LOOP) {
#pragma unroll
      for (int e = 0; e < 16; e++) DUMP2(1, dsite, tile, I * 32 + 4 * h + (e & 3) + 8 * (e >> 2), cc, accX[e], PAIR ? accT[e] : 0.f);
    }
    // folded batch-norm backward needs zhat of the elements this thread copies out below (the ones it loaded above):
    // re-issue those 16 loads now (L2 hits), while no other large register array is live, so that their latency
    // overlaps the projection / assemble phases
    float zr[VEC ? 1 : 16];
    float4 zr4[VEC ? 4 : 1];
    if (BN) {
      if constexpr (VEC) {
#pragma unroll
        for (int q = 0; q < 4; q++) zr4[q] = AT4(x, q);
      } else {
#pragma unroll
        for (int q = 0; q < 16; q++) zr[q] = AT(x, q);   // clamped offset; out-of-range elements are not used below
      }
    }
    // per-column constants of the assemble / copy-out phases: requested here too, so that their round trip is over by then
    if constexpr (!LOOP) {
      const bool cok = kFullB || (col0 + cc) < F;
      rho_x = cok ? stats[F + col0 + cc] : 0.f;
      rho_t = (PAIR && cok) ? stats[3 * F + col0 + cc] : 0.0f;
    }
    float4 sv_m = make_float4(0.f, 0.f, 0.f, 0.f), sv_i = sv_m;      // (VEC, channels-last) batch mean / invstd of the column quad
    if constexpr (VEC && BN) {
      if (bn.nhwc && q_ok) {
        const int ch = (col0 + 4 * lc4) & (bn.C - 1);
        sv_m = *reinterpret_cast<const float4*>(bn.save + ch);
        sv_i = *reinterpret_cast<const float4*>(bn.save + bn.C + ch);
      }
    }
    // ---- projections over this wave's 32 batch rows: sum dVh, sum dVh*Vh.  Vh of this lane's accumulator cells
    //      (rows (e&3)+8(e>>2)+4h of block I, column cc) are 4 consecutive entries of the transposed row: 8-byte reads.
    //      Two rows per instruction (v_pk_add/fma_f32); the reassembled Vh stay in registers for the assembly below.
    f32x2 xv[8], tv[PAIR ? 8 : 1];
    {
      f32x2 x0 = {0.f, 0.f}, x1 = {0.f, 0.f}, t0 = {0.f, 0.f}, t1 = {0.f, 0.f};
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
        const int o = cc * LDT + ((I * 32 + 8 * g4 + 4 * h) ^ XSWZ(cc));
        const uint2 a = *reinterpret_cast<const uint2*>(XThi + o);
        const uint2 b2 = *reinterpret_cast<const uint2*>(XTlo + o);
        xv[2 * g4] = bf16_pair(a.x) + bf16_pair(b2.x);
        xv[2 * g4 + 1] = bf16_pair(a.y) + bf16_pair(b2.y);
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const f32x2 av = {accX[4 * g4 + 2 * q], accX[4 * g4 + 2 * q + 1]};
          x0 += av;
          x1 = __builtin_elementwise_fma(av, xv[2 * g4 + q], x1);
        }
        if (PAIR) {
          const uint2 c2 = *reinterpret_cast<const uint2*>(TThi + o);
          const uint2 d2 = *reinterpret_cast<const uint2*>(TTlo + o);
          tv[PAIR ? 2 * g4 : 0] = bf16_pair(c2.x) + bf16_pair(d2.x);
          tv[PAIR ? 2 * g4 + 1 : 0] = bf16_pair(c2.y) + bf16_pair(d2.y);
#ifdef ALIGNQ_DIAG_DUMP
          if constexpr (TFv == 32 && !LOOP && PAIR) {           // stage 4 (stored in slot 0): the standardised operands as the MFMAs read them
            if ((ALIGNQ_DIAG_DUMP) & 16) {
              const int r0 = I * 32 + 8 * g4 + 4 * h;
              DUMP2(0, dsite, tile, r0 + 0, cc, xv[2 * g4].x, tv[2 * g4].x);
              DUMP2(0, dsite, tile, r0 + 1, cc, xv[2 * g4].y, tv[2 * g4].y);
              DUMP2(0, dsite, tile, r0 + 2, cc, xv[2 * g4 + 1].x, tv[2 * g4 + 1].x);
              DUMP2(0, dsite, tile, r0 + 3, cc, xv[2 * g4 + 1].y, tv[2 * g4 + 1].y);
            }
          }
#endif
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const f32x2 av = {accT[4 * g4 + 2 * q], accT[4 * g4 + 2 * q + 1]};
            t0 += av;
            t1 = __builtin_elementwise_fma(av, tv[PAIR ? 2 * g4 + q : 0], t1);
          }
        }
      }
      // the two half-waves hold the two 16-row halves of every column: one lane swap + add sums both quantities at once and
      // leaves sum dVh in the lower half-wave, sum dVh*Vh in the upper one (no LDS crossbar round trips)
      const float px = swap_add32(x0.x + x0.y, x1.x + x1.y);
      red[((I * 2 + 0) * 2 + h) * TFv + cc] = px;
      if (PAIR) {
        const float pt = swap_add32(t0.x + t0.y, t1.x + t1.y);
        red[((I * 2 + 1) * 2 + h) * TFv + cc] = pt;
      }
    }
    __syncthreads();
    STAMP(13);
    if constexpr (LOOP) {      // the next tile's statistics -> the other half of the image (read behind the third barrier)
      if (tid < 4 * NCQ) *reinterpret_cast<float4*>(stt + ((st_buf ^ 1) * 4 + st_arr) * TFv + 4 * st_q) = st_next;
    }
    // ---- in accumulator layout: cx = rho_x (dVh - mean - Vh proj), ct likewise; jac * ct - cx (PAIR) | cx  ->  Os ----
    {
      float sx0 = 0.f, sx1 = 0.f, st0 = 0.f, st1 = 0.f;
#pragma unroll
      for (int rb = 0; rb < 4; rb++) {
        sx0 += red[((rb * 2 + 0) * 2 + 0) * TFv + cc];
        sx1 += red[((rb * 2 + 0) * 2 + 1) * TFv + cc];
        if (PAIR) {
          st0 += red[((rb * 2 + 1) * 2 + 0) * TFv + cc];
          st1 += red[((rb * 2 + 1) * 2 + 1) * TFv + cc];
        }
      }
      // through-std factor (sd+eps)/sd = 1/(1-eps*rho); torch's std backward is 0 where sd == 0
      float kap_x = 1.0f, kap_t = 1.0f;
      if (eps != 0.0f) {
        const float dxn = 1.0f - eps * rho_x, dtn = 1.0f - eps * rho_t;
        kap_x = (dxn > 1e-12f) ? 1.0f / dxn : 0.0f;
        kap_t = (dtn > 1e-12f) ? 1.0f / dtn : 0.0f;
      }
      const float mean_x = sx0 * invB, proj_x = sx1 * invBm1 * kap_x;
      const float mean_t = st0 * invB, proj_t = st1 * invBm1 * kap_t;
      if constexpr (TFv == 32 && !LOOP) {
        if (h == 0) { DUMP2(2, dsite, tile, 2 * I, cc, sx0, sx1); DUMP2(2, dsite, tile, 2 * I + 1, cc, st0, st1); }
      }
      const f32x2 mx2 = {mean_x, mean_x}, px2 = {proj_x, proj_x}, rx2 = {rho_x, rho_x};
      const f32x2 mt2 = {mean_t, mean_t}, pt2 = {proj_t, proj_t}, rt2 = {rho_t, rho_t};
#pragma unroll
      for (int g4 = 0; g4 < 4; g4++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int e = 4 * g4 + 2 * q;
          const uint32_t a = os_acc + (8 * g4 + 2 * q) * kRow;
          const f32x2 ax = {accX[e], accX[e + 1]};
          f32x2 o2 = rx2 * (ax - mx2 - xv[2 * g4 + q] * px2);             // cx
          if (PAIR) {
            const f32x2 at = {accT[e], accT[e + 1]};
            const f32x2 ct = rt2 * (at - mt2 - tv[PAIR ? 2 * g4 + q : 0] * pt2);
            const f32x2 jp = {LDS_F32(a + kJs), LDS_F32(a + kJs + kRow)};
            o2 = jp * ct - o2;                                              // corr(x,x) enters D with a minus sign
          }
          LDS_F32(a) = o2.x;
          LDS_F32(a + kRow) = o2.y;
          if constexpr (TFv == 32 && !LOOP) DUMP2(3, dsite, tile, I * 32 + 4 * h + 8 * g4 + 2 * q, cc, o2.x, o2.y);
        }
      }
    }
    __syncthreads();
    STAMP(14);
    if constexpr (VEC) {
      // ---- copy out: one 16-byte store per (row, column quad); folded BN: per-column sums of dx and dx * zhat -----------
      float bp0[4] = {0.f, 0.f, 0.f, 0.f}, bp1[4] = {0.f, 0.f, 0.f, 0.f};
      if (q_ok) {
        float bmu[4] = {0.f, 0.f, 0.f, 0.f}, bis[4] = {0.f, 0.f, 0.f, 0.f};
        if (BN) {
          const int fq = col0 + 4 * lc4;
          if (bn.nhwc) {
            (void)fq;
            bmu[0] = sv_m.x; bmu[1] = sv_m.y; bmu[2] = sv_m.z; bmu[3] = sv_m.w;
            bis[0] = sv_i.x; bis[1] = sv_i.y; bis[2] = sv_i.z; bis[3] = sv_i.w;
          } else {
            const int ch = fq / bn.HW;
#pragma unroll
            for (int e = 0; e < 4; e++) { bmu[e] = bn.save[ch]; bis[e] = bn.save[bn.C + ch]; }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int row = lrow4 + q;
          if (kFullB || row < B) {
            // out = g*jac + (jac*ct - cx)  (PAIR)   |   out = cx  (!PAIR)
            const f32x4_nt c4 = LDS_F32X4(os_ld + q * kRow);
            float oo[4] = {c4.x, c4.y, c4.z, c4.w};
            if (PAIR) {
#pragma unroll
              for (int e = 0; e < 4; e++) oo[e] += gj[PAIR ? 4 * q + e : 0];
            }
            ATW4(dx, q) = make_float4(oo[0], oo[1], oo[2], oo[3]);
            if (BN) {
#pragma unroll
              for (int e = 0; e < 4; e++) {      // x is the conv output z here
                bp0[e] += oo[e];
                bp1[e] += oo[e] * ((f4get(zr4[q], e) - bmu[e]) * bis[e]);
              }
            }
          }
        }
      }
      // red is free again (all its reads finished before the barrier above)
      if (BN && bn.nhwc) {
        // sum over the wave's row groups (lanes NCQ apart; VALU lane swaps, see rowgroup_sums), then one row of `red` per
        // wave: [2 * wave + {0,1}][column]; lane (DPP row r, quad lc4) ends with column 4*lc4 + kRowVal[r]
        float s0, s1;
        rowgroup_sums<NCQ, true>(bp0, bp1, s0, s1);
        if ((lane & 15) < NCQ) {
          const int rsel = ((lane >> 4) & 1) * 2 + (lane >> 5);
          red[(2 * w) * TFv + 4 * lc4 + rsel] = s0;
          red[(2 * w + 1) * TFv + 4 * lc4 + rsel] = s1;
        }
      } else if (BN) {                 // one channel per tile
        float s0 = wave_sum(bp0[0] + bp0[1] + bp0[2] + bp0[3]);
        float s1 = wave_sum(bp1[0] + bp1[1] + bp1[2] + bp1[3]);
        if (lane == 0) { red[2 * w] = s0; red[2 * w + 1] = s1; }
      }
      __syncthreads();
      if (BN && bn.nhwc) {
        const int C = bn.C;
        const int cp = C < TFv ? C : TFv;                      // distinct channels in a tile
        if (tid < cp) {
          float t0 = 0.f, t1 = 0.f;
          for (int j = tid; j < TFv; j += cp) {
#pragma unroll
            for (int q = 0; q < NWv; q++) { t0 += red[(2 * q) * TFv + j]; t1 += red[(2 * q + 1) * TFv + j]; }
          }
          bn.dx_part[((int64_t)tile * cp + tid) * 2] = t0;
          bn.dx_part[((int64_t)tile * cp + tid) * 2 + 1] = t1;
        }
      } else if (BN && tid == 0) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int q = 0; q < NWv; q++) { t0 += red[2 * q]; t1 += red[2 * q + 1]; }
        bn.dx_part[2 * tile] = t0;
        bn.dx_part[2 * tile + 1] = t1;
      }
    } else {
    // ---- copy out: 256-byte row segments, one feature column per lane -----------------------------------
    float bp0 = 0.f, bp1 = 0.f;     // folded batch-norm backward: this tile's sum dx and sum dx*zhat (one channel per tile)
    if (lcol_ok) {
      float bmu = 0.f, bis = 0.f;
      if (BN) {
        const int ch = bn.nhwc ? ((col0 + lcol) & (bn.C - 1)) : (col0 + lcol) / bn.HW;
        bmu = bn.save[ch];
        bis = bn.save[bn.C + ch];
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int row = lrow0 + q;
        if (row < B) {
          float o = LDS_F32(os_ld + q * kRow);
          if (PAIR) o += gj[PAIR ? q : 0];
          ATW(dx, q) = o;
          if (BN) {
            const float zh = (zr[q] - bmu) * bis;      // x is the conv output z here
            bp0 += o;
            bp1 += o * zh;
          }
        }
      }
    }
    // red is free again (all its reads finished before the barrier above)
    if (BN && bn.nhwc) {             // channels-last: column j of the tile belongs to channel (col0 + j) mod C
      red[(2 * (tid / TFv)) * TFv + lcol] = bp0;
      red[(2 * (tid / TFv) + 1) * TFv + lcol] = bp1;
    } else if (BN) {                 // one channel per tile
      bp0 = wave_sum(bp0);
      bp1 = wave_sum(bp1);
      if (lane == 0) { red[2 * w] = bp0; red[2 * w + 1] = bp1; }
    }
    __syncthreads();
    if (BN && bn.nhwc) {
      // per-tile, per-channel partial sums [n_tiles][min(C,TF)][2]; alignq_bn_bwd_apply (nhwc) reduces them per channel
      const int C = bn.C;
      const int cp = C < TFv ? C : TFv;                      // distinct channels in a tile
      if (tid < cp) {
        float t0 = 0.f, t1 = 0.f;
        for (int j = tid; j < TFv; j += cp) {
#pragma unroll
          for (int q = 0; q < 8; q++) { t0 += red[(2 * q) * TFv + j]; t1 += red[(2 * q + 1) * TFv + j]; }
        }
        bn.dx_part[((int64_t)tile * cp + tid) * 2] = t0;
        bn.dx_part[((int64_t)tile * cp + tid) * 2 + 1] = t1;
      }
    } else if (BN && tid == 0) {
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int q = 0; q < NWv; q++) { t0 += red[2 * q]; t1 += red[2 * q + 1]; }
      bn.dx_part[2 * tile] = t0;
      bn.dx_part[2 * tile + 1] = t1;
    }
    }
    STAMP(15);
    if constexpr (!LOOP) break;                // no back edge: the one-tile form keeps the register allocation of a plain `if`
    if constexpr (LOOP) {
      const bool has_g = PAIR && gup != nullptr;
#pragma unroll
      for (int q = 0; q < 4; q++) { xr[q] = xn[q]; gr[q] = has_g ? gn[q] : make_float4(0.f, 0.f, 0.f, 0.f); }
      st_buf ^= 1;
    }
    // (no barrier here: the next tile's staging writes XT / TT / Js, whose last reads - projection, assembly - lie before this
    //  tile's second and third barrier; Os and red are rewritten only behind the next tile's first barrier)
#undef AT
#undef ATW
#undef AT4
#undef ATW4
#undef BO4
  }
  BSTAMP(1, 1);
}

#define RET_ON_ERR()                                  \
  do {                                                \
    hipError_t e__ = hipGetLastError();               \
    if (e__ != hipSuccess) return (int)e__;           \
  } while (0)


// One forward launch.  Which instantiation runs is decided HERE, next to the conditions it was written for, so a promise of
// complete tiles (FULLP, the 512-thread multi-tile form) cannot be made for a geometry the kernel masks differently: the
// unmasked one-tile form exists for 32- and 64-feature tiles only (static_assert in the kernel), and is chosen only when
// B == 128, F % TFV == 0 and the tensors are 16-byte aligned.
struct FwdLaunch {
  const float* x; int B; int64_t F; int k; float r, eps; float* xq; float* ws; float* stats; int n_tiles, aligned;
  unsigned* counter; BnFold bn; SFill fill; int grid, fgrid; bool full64; hipStream_t st; Twin twin;
};
template <int TFV, bool P, bool SG, int NTV, bool FULLP>
inline void launch_fwd4_inst(const FwdLaunch& a) {
  hipLaunchKernelGGL((site_fwd4_kernel<TFV, P, SG, NTV, FULLP>), (SG && NTV == NT) ? a.fgrid : a.grid, NTV, 0, a.st, a.x, a.B, a.F,
                     a.k, a.r, a.eps, a.xq, a.ws, a.stats, a.n_tiles, a.aligned, a.counter, a.bn, a.fill, a.twin);
}
template <int TFV, bool P>
inline int launch_fwd4_tf(const FwdLaunch& a) {
  if (a.n_tiles <= a.grid) {                                  // one tile per workgroup
    const bool complete = a.B == 128 && a.F % TFV == 0 && a.aligned;
    if constexpr (TFV >= 32) {
      if (complete) { launch_fwd4_inst<TFV, P, true, NT, true>(a); return 0; }
    }
    launch_fwd4_inst<TFV, P, true, NT, false>(a);
    return 0;
  }
  if constexpr (TFV == 64) {                                  // the tile loop
    const BnFold& bn = a.bn;
    if (bn.ab || bn.res || bn.relu || bn.bins || !a.full64) launch_fwd4_inst<64, P, false, NT, false>(a);
    else launch_fwd4_inst<64, P, false, 512, false>(a);
    return 0;
  }
  return ALIGNQ_EINVAL;
}

// One backward launch, same rule as the forward's: the unmasked instantiations (FULLP one-tile, LOOP) are chosen only by the
// functions that check what they assume.
// site_bwd4_kernel<32, ..> with more than 256 workgroups: 16 KB of unused dynamic LDS per workgroup keep the workgroups ONE per CU.
// Two co-resident site workgroups of that kernel (69 KB of LDS, 196 VGPRs: they fit) gave results that differed by one ulp in scattered
// elements of a few tiles from launch to launch - 8 of 20 launches behind a large GEMM, 2 of 8 training steps, found by the
// graph-replay-equals-eager test (tools/diag_cold_determinism.py; NOTES.md round 6: never with one workgroup per CU, never in a build
// of this file without SLP vectorisation; cause not established).  The twin launch's step time is the same within the spread
// (1.017 / 1.018 ms per step), so nothing is given up.
constexpr int kBwd32OnePerCuLds = 16384;
#ifdef ALIGNQ_DIAG_CORESIDENT
static int g_diag_one_per_cu = 0;
#endif
struct BwdLaunch {
  const float* gup; const float* S; const float* x; const float* stats; int B; int64_t F; float r, eps; float* dx;
  int n_tiles, aligned; BnFold bn; alignq_wgr::RedFill fill; int grid; bool vec; hipStream_t st; BwdTwin twin; int dyn_lds = 0;
};
template <int TFV, bool P, bool N, bool VEC, bool LOOP, bool FULLP>
inline void launch_bwd4_inst(const BwdLaunch& a) {
  hipLaunchKernelGGL((site_bwd4_kernel<TFV, P, N, VEC, LOOP, FULLP>), a.grid, TFV * 8, a.dyn_lds, a.st, a.gup, a.S, a.x, a.stats, a.B, a.F,
                     a.r, a.eps, a.dx, a.n_tiles, a.aligned, a.bn, a.fill, a.twin);
}
template <int TFV, bool P, bool N>
inline void launch_bwd4_tile(const BwdLaunch& a) {             // one tile per workgroup
  if (a.vec && a.B == 128 && a.F % TFV == 0) launch_bwd4_inst<TFV, P, N, true, false, true>(a);
  else if (a.vec) launch_bwd4_inst<TFV, P, N, true, false, false>(a);
  else launch_bwd4_inst<TFV, P, N, false, false, false>(a);
}
template <int TFV, bool P>
inline int launch_bwd4_loop(const BwdLaunch& a) {              // complete tiles only, no fold
  const BnFold& bn = a.bn;
  if (!a.vec || a.B != 128 || a.F % TFV != 0 || bn.ab || bn.y || bn.ybins || bn.dres) return ALIGNQ_EINVAL;
  if (a.grid < 1 || a.n_tiles != (int)(a.F / TFV)) return ALIGNQ_EINVAL;
  launch_bwd4_inst<TFV, P, false, true, true, false>(a);
  return 0;
}
}  // namespace

int launch_partials4(bool pair, const Geom& g, const float* x, int B, int64_t F, int k, float r, float eps, float* xq,
                     float* stats, float* ws, hipStream_t st, BnFold bn, const SiteFillArgs* fa) {
  SFill fill{};
  if (fa && fa->n > 0) {
    // only the one-tile launches that leave half the chip idle take fillers; the caller asked alignq_site_fill_slots first
    if (fa->n > kSiteFill || g.nb != 4 || g.n_tiles > g.grid || g.grid > 128) return ALIGNQ_EINVAL;
    for (int i = 0; i < fa->n; i++) {
      if (!fa->ws[i] || !fa->D[i] || !fa->A[i] || !fa->G[i] || !fa->scal[i] || fa->F[i] < 1) return ALIGNQ_EINVAL;
      const Geom gi = geom(B, fa->F[i]);
      if (gi.nb != 4) return ALIGNQ_EINVAL;
      fill.slabs[i] = (const float*)fa->ws[i]; fill.out[i] = fa->D[i]; fill.A[i] = fa->A[i]; fill.gamma[i] = fa->G[i];
      fill.scal[i] = fa->scal[i]; fill.scale[i] = 1.0f / (float)fa->F[i]; fill.n_slabs[i] = gi.grid;
    }
    fill.n = fa->n; fill.dim = fa->dim; fill.mu = fa->mu; fill.rho = fa->rho;
  }
  const int fgrid = g.grid + fill.n * kSlabRedBlocks;
  if ((int64_t)B * F * 4 >= ((int64_t)1 << 32)) return ALIGNQ_EUNSUPPORTED;   // 32-bit byte offsets (see ld4 / st4)
  const int aligned = ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                      (!xq || (reinterpret_cast<uintptr_t>(xq) & 15) == 0);
  if (bn.bins && (!aligned || bn.res || (reinterpret_cast<uintptr_t>(bn.bins) & 15) || (bn.bin_bytes != 1 && bn.bin_bytes != 2)))
    return ALIGNQ_EINVAL;             // the index is stored per aligned column quad and only for a value that IS a level
  unsigned* counter = reinterpret_cast<unsigned*>(ws + (size_t)g.grid * g.slab_floats + kPartFloats);
  // geom(): one tile per workgroup up to F = 16384; beyond that the 64-feature tile loop runs in up to 512 workgroups of 512
  // threads, two per CU, for the plain site; with the batch-norm fold (no configuration has one at such F) in 1024-thread ones
  const bool full64 = B == 128 && F % 64 == 0 && aligned;     // the 512-thread multi-tile form takes complete tiles only
  const FwdLaunch fl{x, B, F, k, r, eps, xq, ws, stats, g.n_tiles, aligned, counter, bn, fill, g.grid, fgrid, full64, st, Twin{}};
  int rc;
  if (pair) rc = g.tf == 64 ? launch_fwd4_tf<64, true>(fl) : launch_fwd4_tf<32, true>(fl);
  else rc = g.tf == 64 ? launch_fwd4_tf<64, false>(fl) : launch_fwd4_tf<32, false>(fl);
  if (rc) return rc;
  RET_ON_ERR();
  return 0;
}

// Two sites of one shape in ONE launch of the one-tile form (see Twin): only where a single site's launch leaves half the chip
// idle (at most 128 workgroups), complete or masked tiles alike; no filler roles.  ALIGNQ_EUNSUPPORTED otherwise (the caller then
// launches the sites one after the other).
int launch_partials4_twin(const Geom& g, int B, int64_t F, int k, float r, float eps, const float* xa, float* xqa, float* statsa,
                          float* wsa, BnFold bna, const float* xb, float* xqb, float* statsb, float* wsb, BnFold bnb, hipStream_t st) {
  if (g.nb != 4 || g.n_tiles > g.grid || g.grid > 128) return ALIGNQ_EUNSUPPORTED;
  if ((int64_t)B * F * 4 >= ((int64_t)1 << 32)) return ALIGNQ_EUNSUPPORTED;
  auto al = [&](const float* x, float* xq) {
    return ((F & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) && (!xq || (reinterpret_cast<uintptr_t>(xq) & 15) == 0);
  };
  const int aligned = al(xa, xqa) && al(xb, xqb);
  for (const BnFold* bn : {&bna, &bnb})
    if (bn->bins && (!aligned || bn->res || (reinterpret_cast<uintptr_t>(bn->bins) & 15) || (bn->bin_bytes != 1 && bn->bin_bytes != 2)))
      return ALIGNQ_EINVAL;
  auto ctr = [&](float* ws) { return reinterpret_cast<unsigned*>(ws + (size_t)g.grid * g.slab_floats + kPartFloats); };
  const Twin tw{xb, xqb, wsb, statsb, ctr(wsb), bnb, g.grid};
  const FwdLaunch fl{xa, B, F, k, r, eps, xqa, wsa, statsa, g.n_tiles, aligned, ctr(wsa), bna, SFill{}, 2 * g.grid, 2 * g.grid, false, st, tw};
  const int rc = g.tf == 64 ? launch_fwd4_tf<64, true>(fl) : launch_fwd4_tf<32, true>(fl);
  if (rc) return rc;
  RET_ON_ERR();
  return 0;
}

int launch_reduce_any(const Geom& g, const float* ws_c, float* ws_mut, int B, int64_t F, float* out, bool with_loss,
                      const float* alterD, const float* gamma, int dim, float mu, float rho, float* scal,
                      hipStream_t st) {
  const int BP = 32 * g.nb;
  const int blocks = (g.nb == 4) ? (kSlab4Floats + 255) / 256 : (B * B + 63) / 64;   // site4: four STORED elements per lane
  float* parts = ws_mut ? ws_mut + (size_t)g.grid * g.slab_floats : nullptr;
  unsigned* counter = ws_mut ? reinterpret_cast<unsigned*>(ws_mut + (size_t)g.grid * g.slab_floats + kPartFloats) : nullptr;
  const float scale = 1.0f / (float)F;
#define LR(SYM, LOSS) hipLaunchKernelGGL((slab_reduce_kernel<SYM, LOSS>), blocks, 1024, 0, st, ws_c, g.grid, g.slab_floats, BP, B, scale, out, alterD, gamma, dim, mu, rho, parts, counter, scal)
  if (g.nb == 4) { if (with_loss) LR(true, true); else LR(true, false); }
  else { if (with_loss) LR(false, true); else LR(false, false); }
#undef LR
  RET_ON_ERR();
  return 0;
}

int launch_reduce_loss_groups(const Geom& g, float* ws, int B, int64_t F, int groups, float* D, const float* alterD,
                              const float* gamma, int dim, float mu, float rho, float* scal, int64_t ws_gstride, hipStream_t st) {
  if (g.nb == 4) return ALIGNQ_EUNSUPPORTED;
  const int BP = 32 * g.nb, blocks = (B * B + 63) / 64;
  hipLaunchKernelGGL(slab_reduce_groups_kernel, dim3(blocks, groups), 1024, 0, st, ws, g.grid, g.slab_floats, BP, B,
                     1.0f / (float)F, D, alterD, gamma, dim, mu, rho, scal, ws_gstride);
  RET_ON_ERR();
  return 0;
}

int launch_prep(bool fused, const float* dD, const float* D, const float* alterD, const float* gamma, int dim,
                const float* scal, float mu, const float* gscale, int B, int64_t F, float* S, float* dA_out,
                float* dG_out, hipStream_t st) {
  const int total = fused ? dim * dim : B * B;
  const int blocks = (total + 255) / 256;
  const float invF = 1.0f / (float)F;
  if (fused)
    hipLaunchKernelGGL((site_prep_kernel<true>), blocks, 256, 0, st, dD, D, alterD, gamma, dim, scal, mu, gscale, B, invF, S, dA_out, dG_out);
  else
    hipLaunchKernelGGL((site_prep_kernel<false>), blocks, 256, 0, st, dD, D, alterD, gamma, dim, scal, mu, gscale, B, invF, S, dA_out, dG_out);
  RET_ON_ERR();
  return 0;
}

int launch_reduce_loss_multi(int S, void* const* ws, float* const* D, const float* const* alterD,
                             const float* const* gamma, float* const* scal, const int64_t* F, int B, int dim, float mu,
                             float rho, hipStream_t st) {
  for (int s0 = 0; s0 < S; s0 += kMultiSites) {
    const int cnt = (S - s0 < kMultiSites) ? S - s0 : kMultiSites;
    RChunk c;
    for (int i = 0; i < cnt; i++) {
      const Geom g = geom(B, F[s0 + i]);
      c.slabs[i] = (const float*)ws[s0 + i]; c.out[i] = D[s0 + i]; c.A[i] = alterD[s0 + i]; c.gamma[i] = gamma[s0 + i];
      c.scal[i] = scal[s0 + i]; c.scale[i] = 1.0f / (float)F[s0 + i]; c.n_slabs[i] = g.grid;
    }
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3((kSlab4Floats + 255) / 256, cnt), 1024, 0, st, c, B, dim, mu, rho);
    RET_ON_ERR();
  }
  return 0;
}

int launch_reduce_loss_multi_head(int S, void* const* ws, float* const* D, const float* const* alterD, const float* const* gamma,
                                  float* const* scal, const int64_t* F, int B, int dim, float mu, float rho, const float* feat,
                                  const float* W, const float* bias, const int64_t* target, int HB, int HW, int C, int K, float* pooled,
                                  float* logits, float* probs, float* loss, float* ce_mean, unsigned* head_counter,
                                  const float* scal_all, int n_sites, float* trans_total, unsigned* site_counter, hipStream_t st) {
  // all but the last chunk of sites as plain launches; the last chunk takes the head along
  const int s_last = ((S - 1) / kMultiSites) * kMultiSites;
  if (s_last > 0)
    if (int rc = launch_reduce_loss_multi(s_last, ws, D, alterD, gamma, scal, F, B, dim, mu, rho, st)) return rc;
  const int cnt = S - s_last;
  RChunk c;
  for (int i = 0; i < cnt; i++) {
    const Geom g = geom(B, F[s_last + i]);
    c.slabs[i] = (const float*)ws[s_last + i]; c.out[i] = D[s_last + i]; c.A[i] = alterD[s_last + i]; c.gamma[i] = gamma[s_last + i];
    c.scal[i] = scal[s_last + i]; c.scale[i] = 1.0f / (float)F[s_last + i]; c.n_slabs[i] = g.grid;
  }
  const HeadFwd h{feat, W, bias, target, HW, C, K, HB, pooled, logits, probs, loss, ce_mean, head_counter};
  const TransTail tt{scal_all, n_sites, trans_total, site_counter};
  hipLaunchKernelGGL(slab_reduce_multi_head_kernel, cnt * kSlabRedBlocks + HB, 1024, 0, st, c, B, dim, mu, rho, cnt, h, tt);
  RET_ON_ERR();
  return 0;
}

int launch_prep_groups(const float* D, const float* alterD, const float* gamma, int dim, const float* scal, float mu,
                       const float* gscale, int B, int64_t F, int groups, float* S, int64_t s_gstride, float* dA, float* dG,
                       hipStream_t st, int gs_stride) {
  const int gx = (dim * dim + 255) / 256;
  hipLaunchKernelGGL(site_prep_groups_kernel, dim3(gx, groups), 256, 0, st, D, alterD, gamma, dim, scal, mu, gscale, B, 1.0f / (float)F,
                     S, s_gstride, dA, dG, groups, gs_stride);
  RET_ON_ERR();
  return 0;
}

int launch_reduce_loss_groups_multi(int T, float* const* ws, const int64_t* F, int B, int groups, float* const* D,
                                    const float* const* alterD, const float* const* gamma, int dim, float mu, float rho,
                                    float* const* scal, hipStream_t st) {
  const int blocks = (B * B + 63) / 64;
  for (int s0 = 0; s0 < T; s0 += kMultiSites) {
    const int cnt = (T - s0 < kMultiSites) ? T - s0 : kMultiSites;
    GRChunk c;
    int slab_floats = 0, BP = 0;
    for (int i = 0; i < cnt; i++) {
      const Geom g = geom(B, F[s0 + i]);
      if (g.nb == 4) return ALIGNQ_EUNSUPPORTED;
      if (i == 0) { slab_floats = g.slab_floats; BP = 32 * g.nb; }
      else if (slab_floats != g.slab_floats || BP != 32 * g.nb) return ALIGNQ_EUNSUPPORTED;       // (one B: one slab geometry)
      c.ws[i] = ws[s0 + i]; c.D[i] = D[s0 + i]; c.A[i] = alterD[s0 + i]; c.gamma[i] = gamma[s0 + i]; c.scal[i] = scal[s0 + i];
      c.ws_gstride[i] = (int64_t)(alignq_site_ws_bytes(B, F[s0 + i]) / 4);
      c.scale[i] = 1.0f / (float)F[s0 + i]; c.n_slabs[i] = g.grid;
    }
    hipLaunchKernelGGL(slab_reduce_groups_multi_kernel, dim3(blocks, cnt * groups), 1024, 0, st, c, groups, slab_floats, BP, B, dim,
                       mu, rho);
    RET_ON_ERR();
  }
  return 0;
}

int launch_prep_groups_multi(int T, const float* const* D, const float* const* alterD, const float* const* gamma, int dim,
                             const float* const* scal, float mu, const float* gscale, int B, const int64_t* F, int groups,
                             float* const* S, int64_t s_gstride, float* const* dA, float* const* dG, hipStream_t st) {
  const int gx = (dim * dim + 255) / 256;
  for (int s0 = 0; s0 < T; s0 += kMultiSites) {
    const int cnt = (T - s0 < kMultiSites) ? T - s0 : kMultiSites;
    PChunk c;
    for (int i = 0; i < cnt; i++) {
      c.D[i] = D[s0 + i]; c.A[i] = alterD[s0 + i]; c.gamma[i] = gamma[s0 + i]; c.scal[i] = scal[s0 + i];
      c.S[i] = S[s0 + i]; c.dA[i] = dA ? dA[s0 + i] : nullptr; c.dG[i] = dG ? dG[s0 + i] : nullptr;
      c.invF[i] = 1.0f / (float)F[s0 + i];
    }
    hipLaunchKernelGGL(site_prep_groups_multi_kernel, dim3(gx, cnt * groups), 256, 0, st, c, groups, dim, mu, gscale, B, s_gstride);
    RET_ON_ERR();
  }
  return 0;
}

int launch_prep_multi(int S, const float* const* D, const float* const* alterD, const float* const* gamma,
                      const float* const* scal, const float* gscale, const int64_t* F, int B, int dim, float mu,
                      float* const* Sout, float* const* dA, float* const* dG, hipStream_t st) {
  for (int s0 = 0; s0 < S; s0 += kMultiSites) {
    const int cnt = (S - s0 < kMultiSites) ? S - s0 : kMultiSites;
    PChunk c;
    for (int i = 0; i < cnt; i++) {
      c.D[i] = D[s0 + i]; c.A[i] = alterD[s0 + i]; c.gamma[i] = gamma[s0 + i]; c.scal[i] = scal[s0 + i];
      c.S[i] = Sout[s0 + i]; c.dA[i] = dA[s0 + i]; c.dG[i] = dG[s0 + i]; c.invF[i] = 1.0f / (float)F[s0 + i];
    }
    hipLaunchKernelGGL(site_prep_multi_kernel, dim3((dim * dim + 255) / 256, cnt), 256, 0, st, c, dim, mu, gscale, B);
    RET_ON_ERR();
  }
  return 0;
}

int launch_head_bwd_prep_multi(const float* g_ce, const float* probs, const int64_t* target, const float* pooled, const float* W,
                               int HB, int HW, int C, int K, float* dfeat, float* dW, float* dbias, int S, const float* const* D,
                               const float* const* alterD, const float* const* gamma, const float* const* scal,
                               const float* gscale, const int64_t* F, int B, int dim, float mu, float* const* Sout,
                               float* const* dA, float* const* dG, hipStream_t st) {
  if (C < 1 || C > alignq_head::kMaxC || K < 1 || K > alignq_head::kMaxK) return ALIGNQ_EUNSUPPORTED;
  const int cnt = S < kMultiSites ? S : kMultiSites;          // the first chunk of sites rides with the head
  PChunk c;
  for (int i = 0; i < cnt; i++) {
    c.D[i] = D[i]; c.A[i] = alterD[i]; c.gamma[i] = gamma[i]; c.scal[i] = scal[i];
    c.S[i] = Sout[i]; c.dA[i] = dA[i]; c.dG[i] = dG[i]; c.invF[i] = 1.0f / (float)F[i];
  }
  const HeadBwdArgs h{g_ce, probs, target, pooled, W, HB, HW, C, K, dfeat, dW, dbias};
  const int gx = (dim * dim + 255) / 256, n_head = HB + K;
  hipLaunchKernelGGL(head_bwd_prep_multi_kernel, n_head + gx * cnt, 256, 0, st, h, n_head, c, dim, mu, gscale, B, gx);
  RET_ON_ERR();
  if (S > cnt)
    return launch_prep_multi(S - cnt, D + cnt, alterD + cnt, gamma + cnt, scal + cnt, gscale, F + cnt, B, dim, mu, Sout + cnt,
                             dA + cnt, dG + cnt, st);
  return 0;
}

int launch_bwd4(bool pair, const Geom& g, const float* gup, const float* S, const float* x, const float* stats, int B,
                int64_t F, float r, float eps, float* dx, hipStream_t st, BnFold bn, const alignq_wgr::RedFill* fa) {
  (void)g;
  alignq_wgr::RedFill fill{};
  if (fa) fill = *fa;
  if ((int64_t)B * F * 4 >= ((int64_t)1 << 32)) return ALIGNQ_EUNSUPPORTED;   // 32-bit byte offsets inside a tile column
  const int tf = bwd_tile_features(B, F);
  const int n_tiles = (int)((F + tf - 1) / tf);
  int grid = n_tiles;                     // one tile per workgroup (the looped form below: one workgroup per CU)
  if (fill.blk0[alignq_wgr::kFill] > 0) {
    if (!bwd_fill_ok(B, F)) return ALIGNQ_EINVAL;      // only the 32-feature one-tile launches carry the filler role
    grid += fill.blk0[alignq_wgr::kFill];
  }
  const int aligned = 0;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  // 16-byte accesses need whole column quads and aligned rows (channels-last BN: C % 4 == 0 holds, C is a power of two >= 4)
  const bool vec = (F % 4 == 0) && al16(gup) && al16(x) && al16(stats) && al16(dx) && al16(bn.y) && al16(bn.dres) && al16(bn.ybins) &&
                   al16(bn.ab) && al16(bn.save) && (!bn.ab || bn.nhwc || bn.HW % 4 == 0) && (!bn.nhwc || bn.C % 4 == 0);
  BwdLaunch bl{gup, S, x, stats, B, F, r, eps, dx, n_tiles, aligned, bn, fill, grid, vec, st, BwdTwin{}};
  if (tf == 32 && n_tiles > 256) bl.dyn_lds = kBwd32OnePerCuLds;      // (8192 < F < 16384: no configuration has such a site)
  const bool plain = !bn.ab && !bn.y && !bn.ybins && !bn.dres;
  if (tf == 64 && plain && n_tiles > 2 * 256 && vec && B == 128 && F % 64 == 0) {
    // plain site with many tiles per CU (F > 32768): the looped, software-pipelined form (138 KB of LDS: one workgroup per CU)
    BwdLaunch b2 = bl;
    b2.grid = 256;
    const int rc = pair ? launch_bwd4_loop<64, true>(b2) : launch_bwd4_loop<64, false>(b2);
    if (rc) return rc;
  } else if (tf == 64) {
    if (pair && bn.ab) launch_bwd4_tile<64, true, true>(bl); else if (pair) launch_bwd4_tile<64, true, false>(bl); else launch_bwd4_tile<64, false, false>(bl);
  } else {
    if (pair && bn.ab) launch_bwd4_tile<32, true, true>(bl); else if (pair) launch_bwd4_tile<32, true, false>(bl); else launch_bwd4_tile<32, false, false>(bl);
  }
  RET_ON_ERR();
  return 0;
}

// The backward of two sites of one shape in ONE launch (see BwdTwin): only the 32-feature one-tile form whose single launch leaves
// half of the chip's workgroup slots free (bwd_fill_ok); both sites with the folded batch-norm and the same access form.
int launch_bwd4_twin(int B, int64_t F, float r, float eps, const float* ga, const float* Sa, const float* xa, const float* statsa,
                     float* dxa, BnFold bna, const float* gb, const float* Sb, const float* xb, const float* statsb, float* dxb,
                     BnFold bnb, hipStream_t st) {
  if (!bwd_fill_ok(B, F) || !bna.ab || !bnb.ab) return ALIGNQ_EUNSUPPORTED;
  if ((int64_t)B * F * 4 >= ((int64_t)1 << 32)) return ALIGNQ_EUNSUPPORTED;
  const int n_tiles = (int)((F + 31) / 32);
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  auto vec_of = [&](const float* g, const float* x, const float* stats, const float* dx, const BnFold& bn) {
    return (F % 4 == 0) && al16(g) && al16(x) && al16(stats) && al16(dx) && al16(bn.y) && al16(bn.dres) && al16(bn.ybins) &&
           al16(bn.ab) && al16(bn.save) && (bn.nhwc || bn.HW % 4 == 0) && (!bn.nhwc || bn.C % 4 == 0);
  };
  const bool va = vec_of(ga, xa, statsa, dxa, bna), vb = vec_of(gb, xb, statsb, dxb, bnb);
  if (va != vb) return ALIGNQ_EUNSUPPORTED;
  alignq_wgr::RedFill fill{};
  const BwdTwin tw{gb, Sb, xb, statsb, dxb, bnb, n_tiles};
  BwdLaunch bl{ga, Sa, xa, statsa, B, F, r, eps, dxa, n_tiles, 0, bna, fill, 2 * n_tiles, va, st, tw};
#ifndef ALIGNQ_DIAG_CORESIDENT
  if (2 * n_tiles > 256) bl.dyn_lds = kBwd32OnePerCuLds;      // the F = 8192 pair: 512 workgroups, two rounds of 256
#else
  if (2 * n_tiles > 256 && g_diag_one_per_cu) bl.dyn_lds = kBwd32OnePerCuLds;      // (diagnostic build: alignq_debug_one_per_cu)
#endif
  launch_bwd4_tile<32, true, true>(bl);
  RET_ON_ERR();
  return 0;
}

#ifdef ALIGNQ_DIAG_CORESIDENT
extern "C" void alignq_debug_one_per_cu(int on) { g_diag_one_per_cu = on; }
#endif
#ifdef ALIGNQ_DIAG_DUMP
extern "C" int alignq_debug_set_dump(float* device_buffer) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dump), &device_buffer, sizeof(float*));
}
#endif
#ifdef ALIGNQ_STAMPS
extern "C" int alignq_debug_read_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 64);
}
extern "C" int alignq_debug_read_block_stamps(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_blk), sizeof(unsigned long long) * 2 * 2 * 2048);
}
#endif

}  // namespace alignq_site
