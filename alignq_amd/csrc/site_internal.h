// site_internal.h — geometry and launcher declarations shared by site_kernels.hip (generic B<=64 kernels,
// C ABI) and site4_kernels.hip (the B in (64,128] kernels: 1024-thread workgroups, symmetric tiles).
#pragma once
#include "wgrad_reduce_body.h"
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace alignq_site {

// Workspace layout (floats): [grid * slab_floats partial slabs][kPartFloats loss partials][counter, pad]
constexpr int kPartFloats = 1024;   // up to 256 blocks x 4 floats of ADMM-loss partial sums
constexpr int kTailFloats = kPartFloats + 16;
// Partial-Gram slab of the B in (64,128] kernels: the 10 upper-triangular 32x32 tiles of the symmetric 128x128 partial, packed:
// the six off-diagonal tiles (0,1)(0,2)(0,3)(1,2)(1,3)(2,3) row-major (6144 floats), then two [32][33] blocks that each hold
// the upper triangles (r <= c) of TWO diagonal tiles: tile 2*blk at [r][c + 1], tile 2*blk + 1 transposed at [c][r].
constexpr int kSlab4Off = 6 * 1024, kSlab4Packed = 2 * 32 * 33, kSlab4Floats = kSlab4Off + kSlab4Packed;   // 8256 (was 10240)

// Optional batch-norm fold: the site kernels read the CONV output z and apply x = a[c]*z + b[c] on load
// (c = feature / HW), a = gamma*invstd, b = beta - mean*a; ab == nullptr means the input already is x.
constexpr int kBnSplit = 16;   // batch splits per channel in bn_stats (partials per channel)
constexpr int kNhwcParts = 64; // channels-last statistics: partials per channel (one per workgroup of the statistics kernel)

struct BnFold {
  const float* ab;      // [2][C]: a then b.  Forward with `part`: OUTPUT (written by the first tile of each channel)
  const float* save;    // [2][C]: batch mean then invstd.  Forward with `part`: OUTPUT
  int HW, C;
  float* dx_part;       // backward only: per-tile partial sums (sum dx, sum dx*zhat) [F / bwd_tile_features][2]
  const float* y;       // backward only: the forward's relu(x_q) output when the ReLU is fused (mask y > 0), else nullptr
  // forward, in-kernel finalisation of the batch statistics (bn_stats partials -> mean/invstd/a/b + running statistics)
  const double* part;   // [C][kBnSplit][2] {sum z, sum z^2}; nullptr => `ab` is an input
  const float* gamma;
  const float* beta;
  float* running_mean;
  float* running_var;
  long long* nbt;
  float momentum, bn_eps;
  int relu;             // forward: store relu(x_q [+ res])
  const float* res;     // forward: optional residual (shortcut) added to x_q before the ReLU, same [B,F] layout
  float* dres;          // backward: optional output, the upstream gradient after the ReLU mask (= gradient of `res`)
  int nhwc;             // 0: z is [B,C,HW] (channel = f / HW);  1: channels-last [B,HW,C] (channel = f mod C, C a power of
                        // two in [4,256]): forward `part` is [C][n_parts][2]; backward dx_part is [n_tiles][min(C,TF)][2]
  int n_parts;          // channels-last forward: partials per channel (kNhwcParts doubles from alignq_bn_partial_stats_nhwc,
  int part_f32;         //   or, part_f32 = 1, the convolution epilogue's per-workgroup FLOAT partials)
  // N2: the stored activation as its integer level index, int8 (bin_bytes 1) or int16 (2); only without a residual
  void* bins;           // forward: optional output, [B,F] indices of the stored value (clamped at 0 when the ReLU is fused)
  const void* ybins;    // backward: the forward's indices; the ReLU mask is idx > 0 (instead of y > 0 on an fp32 y)
  int bin_bytes;
};
inline BnFold no_bn() {
  return BnFold{nullptr, nullptr, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0.f, 0,
                nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0};
}

// Features per tile of the B in (64,128] backward kernel (also the granularity of BnFold::dx_part).
inline int bwd_tile_features(int B, int64_t F) {
  (void)B;
  return F >= 16384 ? 64 : 32;
}

// the backward launches that carry a filler role (site4_kernels.hip: site_bwd4_kernel<32, ..>, one tile per workgroup, at most
// 256 workgroups of 256 threads: half the chip's wave slots stay free)
inline bool bwd_fill_ok(int B, int64_t F) { return B > 64 && B <= 128 && F >= 32 && bwd_tile_features(B, F) == 32 && F <= 8192; }

struct Geom {
  int nb;           // 32-row blocks: 1, 2 (generic kernels) or 4 (site4 kernels)
  int tf;           // features per tile
  int n_tiles;
  int grid;         // workgroups of the partials kernel == number of slabs
  int slab_floats;  // nb<4: BP*BP (full matrix);  nb==4: kSlab4Floats (packed upper triangle)
};

inline Geom geom(int B, int64_t F) {
  Geom g;
  g.nb = B <= 32 ? 1 : (B <= 64 ? 2 : 4);
  if (g.nb == 4) {
    // tile width by F (only the 64-feature kernel ever loops over tiles; the one-tile forms are the latency-tuned ones)
    // F <= 8192 takes tiles twice as wide as "one tile per CU" would (128 workgroups instead of 256): half as many partial-Gram
    // slabs are written and reduced (118 instead of 173 MB per ResNet-20 step), which outweighs the longer per-tile chain
    g.tf = (F > 16 * 256) ? 64 : 32;
    // beyond one 64-feature tile per CU the kernel loops over tiles: two 512-thread workgroups per CU (80 KB of LDS each)
    // (32-feature tiles were tried for this form: 128-byte row segments copy at 4.7-5.3 TB/s against 5.9-6.3 for 256-byte
    // ones, tools/src/stream_bw.hip, and the kernel ran 203 us against 186 at [128, 524288])
    const bool looped = F > 64 * 256;
    g.n_tiles = (int)((F + g.tf - 1) / g.tf);
    const int cap = looped ? 512 : 256;
    g.grid = g.n_tiles < cap ? g.n_tiles : cap;
    g.slab_floats = kSlab4Floats;
  } else {
    g.tf = 64;
    g.n_tiles = (int)((F + 63) / 64);
    // 8 workgroups (of 4 waves) per CU keep enough loads in flight for the HBM-bound small-batch sites; the slabs
    // are only (32*nb)^2 floats each
    g.grid = g.n_tiles < 2048 ? g.n_tiles : 2048;
    if (g.nb == 1) {                 // site1 kernels: four wave-autonomous 32-feature sub-tiles per workgroup
      g.grid = (int)((F + 127) / 128);
      // 1024 workgroups = one resident round at four waves per SIMD (round 3: 2048 ran the kernel no faster - 57.6 vs 58.2 us at
      // [28, 802816] - and doubled the slabs the reduction reads: config 5 22.36 -> 22.24 ms)
#ifndef ALIGNQ_S1_CAP
#define ALIGNQ_S1_CAP 1024
#endif
      constexpr int cap1 = ALIGNQ_S1_CAP;
      if (g.grid > cap1) g.grid = cap1;
    }
    g.slab_floats = (32 * g.nb) * (32 * g.nb);
  }
  return g;
}

inline size_t ws_floats(const Geom& g) { return (size_t)g.grid * g.slab_floats + kTailFloats; }

// ---- launchers defined in site1_kernels.hip (2 <= B <= 32) ---------------------------------------------------
int launch_partials1(bool pair, const Geom& g, const float* x, int B, int64_t F, int k, float r, float eps, float* xq,
                     float* stats, float* ws, hipStream_t st, const float* res = nullptr, int relu = 0,
                     const float* ab = nullptr, int C = 1, int groups = 1, int64_t ws_gstride = 0,       // ab: folded batch-norm (channels-last, C a power of two)
                     unsigned* rmask = nullptr);      // (round 5) one sign bit per stored element [groups][sub-tiles][32 rows]: the backward's ReLU mask
int launch_reduce_loss_groups(const Geom& g, float* ws, int B, int64_t F, int groups, float* D, const float* alterD,
                              const float* gamma, int dim, float mu, float rho, float* scal, int64_t ws_gstride, hipStream_t st);
int launch_bwd1(bool pair, const float* gup, const float* S, const float* x, const float* stats, int B, int64_t F,
                float r, float eps, float* dx, hipStream_t st, const float* ab = nullptr, int C = 1,
                const float* ymask = nullptr, float* dres = nullptr, int groups = 1, int64_t s_gstride = 0,
                const float* gup2 = nullptr,       // a second addend of the upstream gradient (fused.GradFork), or nullptr
                // round 4: per feature column sum_b dx and sum_b dx * zhat for the folded batch-norm's backward: save = [groups][2][C]
                // (mean, invstd), colsum = [2][groups][F] floats (written)
                const float* save = nullptr, float* colsum = nullptr,
                const unsigned* rmask = nullptr);      // the forward's one-bit ReLU mask instead of ymask (4 B per element)
// bnq_kernels.hip: the batch-norm backward from those per-column sums (HW columns per channel and group): a small reduction over
// the columns, the finalisation (channels with gamma == 0 are summed from dx and z directly) and dz = a (dx - k0 - zhat k1)
int launch_bnq_bwd_from_cols(const float* cols, const float* dx, const float* z, const float* ab, const float* save, int64_t P,
                             int64_t HW, int C, int groups, float* dz, float* dgamma, float* dbeta, void* ws, hipStream_t st);

// ---- launchers defined in site4_kernels.hip ----------------------------------------------------------------
// earlier sites whose slab reduction + ADMM loss ride in this forward launch as a filler role (site4_kernels.hip: SFill)
struct SiteFillArgs {
  int n;
  void* const* ws;
  float* const* D;
  const float* const* A;
  const float* const* G;
  float* const* scal;
  const int64_t* F;
  int dim;
  float mu, rho;
};
inline int site_fill_slots(int B, int64_t F) {       // fillers a forward launch at (B, F) takes: the launches that leave CUs idle
  if (B <= 64 || B > 128 || F < 1) return 0;
  const Geom g = geom(B, F);
  return (g.nb == 4 && g.n_tiles <= g.grid && g.grid <= 128) ? 3 : 0;
}
int launch_partials4(bool pair, const Geom& g, const float* x, int B, int64_t F, int k, float r, float eps, float* xq,
                     float* stats, float* ws, hipStream_t st, BnFold bn = no_bn(), const SiteFillArgs* fa = nullptr);
// two sites of one shape in one launch of the one-tile form (ALIGNQ_EUNSUPPORTED unless a single site leaves half the chip idle)
int launch_partials4_twin(const Geom& g, int B, int64_t F, int k, float r, float eps, const float* xa, float* xqa, float* statsa,
                          float* wsa, BnFold bna, const float* xb, float* xqb, float* statsb, float* wsb, BnFold bnb, hipStream_t st);
int launch_bwd4(bool pair, const Geom& g, const float* gup, const float* S, const float* x, const float* stats, int B,
                int64_t F, float r, float eps, float* dx, hipStream_t st, BnFold bn = no_bn(), const alignq_wgr::RedFill* fill = nullptr);
int launch_bwd4_twin(int B, int64_t F, float r, float eps, const float* ga, const float* Sa, const float* xa, const float* statsa,
                     float* dxa, BnFold bna, const float* gb, const float* Sb, const float* xb, const float* statsb, float* dxb,
                     BnFold bnb, hipStream_t st);
// S = sym(gD) * gscale / F (and, fused, the scaled ADMM parameter gradients) — first launch of every backward
int launch_prep(bool fused, const float* dD, const float* D, const float* alterD, const float* gamma, int dim,
                const float* scal, float mu, const float* gscale, int B, int64_t F, float* S, float* dA_out,
                float* dG_out, hipStream_t st);
// `groups` batch slices of ONE small-batch site: S per slice, the shared module's parameter gradients summed over the slices
int launch_reduce_loss_groups_multi(int T, float* const* ws, const int64_t* F, int B, int groups, float* const* D,
                                    const float* const* alterD, const float* const* gamma, int dim, float mu, float rho,
                                    float* const* scal, hipStream_t st);
int launch_prep_groups_multi(int T, const float* const* D, const float* const* alterD, const float* const* gamma, int dim,
                             const float* const* scal, float mu, const float* gscale, int B, const int64_t* F, int groups,
                             float* const* S, int64_t s_gstride, float* const* dA, float* const* dG, hipStream_t st);
int launch_prep_groups(const float* D, const float* alterD, const float* gamma, int dim, const float* scal, float mu,
                       const float* gscale, int B, int64_t F, int groups, float* S, int64_t s_gstride, float* dA, float* dG,
                       hipStream_t st, int gs_stride = 0);     // gs_stride: elements between the slices' loss gradients (0: one for all)
// all sites of a model in one launch each (64 < B <= 128 only)
int launch_reduce_loss_multi(int S, void* const* ws, float* const* D, const float* const* alterD,
                             const float* const* gamma, float* const* scal, const int64_t* F, int B, int dim, float mu,
                             float rho, hipStream_t st);
// the same with the classifier head's forward as a second role of the (last) launch: see slab_reduce_multi_head_kernel
int launch_reduce_loss_multi_head(int S, void* const* ws, float* const* D, const float* const* alterD, const float* const* gamma,
                                  float* const* scal, const int64_t* F, int B, int dim, float mu, float rho, const float* feat,
                                  const float* W, const float* bias, const int64_t* target, int HB, int HW, int C, int K, float* pooled,
                                  float* logits, float* probs, float* loss, float* ce_mean, unsigned* head_counter,
                                  const float* scal_all, int n_sites, float* trans_total, unsigned* site_counter, hipStream_t st);
int launch_prep_multi(int S, const float* const* D, const float* const* alterD, const float* const* gamma,
                      const float* const* scal, const float* gscale, const int64_t* F, int B, int dim, float mu,
                      float* const* Sout, float* const* dA, float* const* dG, hipStream_t st);
// the same plus the classifier head's backward (alignq_head_ce_bwd's arguments) as a second role of the launch
int launch_head_bwd_prep_multi(const float* g_ce, const float* probs, const int64_t* target, const float* pooled, const float* W,
                               int HB, int HW, int C, int K, float* dfeat, float* dW, float* dbias, int S, const float* const* D,
                               const float* const* alterD, const float* const* gamma, const float* const* scal,
                               const float* gscale, const int64_t* F, int B, int dim, float mu, float* const* Sout,
                               float* const* dA, float* const* dG, hipStream_t st);
// slab reduction for both geometries (+ optional ADMM-loss scalar through a last-block epilogue)
int launch_reduce_any(const Geom& g, const float* ws_c, float* ws_mut, int B, int64_t F, float* out, bool with_loss,
                      const float* alterD, const float* gamma, int dim, float mu, float rho, float* scal,
                      hipStream_t st);

// ---- corr_large_kernels.hip: corr(x, x) for 128 < B <= ALIGNQ_MAX_CORR_BATCH (blocked Gram, exact fp32) ------------------
size_t corrl_ws_bytes(int B, int64_t F);
int launch_corrl_fwd(const float* x, int B, int64_t F, float eps, float* G, float* stats, float* ws, hipStream_t st);
int corrl_s_width(int B);
size_t corrl_s_bytes(int B);      // the backward's zero-padded S image
int launch_corrl_bwd(const float* dG, const float* x, const float* stats, int B, int64_t F, float eps, float* dx, float* S,
                     hipStream_t st, bool pair = false, const float* gup = nullptr, float r = 1.0f, const float* dG_scale = nullptr);
// the ADMM site for 128 < B <= ALIGNQ_MAX_CORR_BATCH on the blocked Gram (stats [4][F]; ws: corrl_ws_bytes)
int launch_sitel_fwd(const float* x, int B, int64_t F, int k, float r, float eps, float* xq, float* D, float* stats, float* ws,
                     hipStream_t st);

}  // namespace alignq_site

// The small-batch site backward leaves sum dx * zhat per feature column with zhat rebuilt from x = a*z + b as (x - beta) / gamma
// (site1_kernels.hip); x carries z only to about eps_f32 * |beta + gamma zhat|, so for |gamma| << |beta| the rebuilt zhat is
// noise (gamma = 1e-8, beta = 0.1: steps of ~0.7).  Such channels - gamma == 0 included - are summed from dx and z directly by
// bnq_finalize_bwd_kernel instead (at 1e-2 the column form keeps zhat to 6e-6); BOTH kernels decide with this predicate on the same (a, b, mean, invstd) floats:
// |gamma| < 1e-2 |beta|  <=>  |a| < 1e-2 |beta| invstd  (a = gamma * invstd, beta = mean * a + b as the forward formed b).
__host__ __device__ __forceinline__ bool alignq_bn_col_ill(float a, float b, float mean, float invstd) {
  const float beta = fmaf(mean, a, b);
  return a == 0.0f || fabsf(a) < 1e-2f * fabsf(beta) * invstd;
}
