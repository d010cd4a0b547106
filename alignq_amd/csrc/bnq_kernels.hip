// bnq_kernels.hip — training-mode batch-norm folded into the PLAIN CDF quantiser (+ the ReLU behind it), any batch, for
// channels-last tensors: the Office tree's bottleneck sites without an ADMM term,
//     out = relu(act_q1(bn1(conv1(x))));  out = relu(act_q2(bn2(conv2(out))))      (cdf_alignment_admm/dann_office/model/
//     resnet.py:134-143, stem :230-233; activation_quantize_fn :87-110)
// SURVEY.md §8f-N1 on configuration 5.  The convolution is MIOpen's (out of scope) and hands over z; what is folded is
// everything between z and the next convolution's input:
//   forward : bnq_sums<STATS> (per-channel sum / sum of squares of z: one read of z) -> bnq_finalize (mean, invstd, running
//             statistics, a = gamma*invstd, b = beta - mean*a) -> bnq_apply_fwd: y = relu(quantise(a*z + b)) (read z, write y)
//             = 12 B/element instead of 20 (BatchNorm forward: two reads + one write, then the quantiser's read + write); the
//             normalised activation is never written;
//   backward: bnq_sums<BWD> (dx = g * [y > 0] * dt/dx(a*z + b); per-channel sum dx, sum dx*zhat: reads g, z, y) ->
//             bnq_finalize_bwd (k0 = mean dx, k1 = mean dx*zhat, dgamma, dbeta) -> bnq_apply_bwd: dz = a*(dx - k0 - zhat*k1)
//             (reads g, z, y, writes dz) = 28 B/element instead of 16 (quantiser + ReLU backward) + ~20 (BatchNorm backward).
// z viewed as [P, C], P = B*H*W pixels, C channels fastest (torch.channels_last); C = 4 * 2^j <= 2048 so that a thread's
// channel quad is fixed for the whole launch (256 threads = slots x C/4 quads; 512 at C = 2048).  Arithmetic: statistics in double, the
// affine as ONE fma (the form oracle/alignq_oracle.c: oq_bn_fold_ab / oq_bn_apply pin to torch.nn.BatchNorm2d), then the
// quantiser of alignq_math.h: y is bit-identical to quantising the oracle's BN output.  Deterministic (fixed-order sums).
#include <hip/hip_runtime.h>

#include "../../include/alignq.h"
#include "alignq_math.h"
#include "site_internal.h"

using namespace alignq;

namespace {

constexpr int kT = 256;
constexpr int kParts = 512;          // workgroups of the two reduction kernels = partials per channel
constexpr int kUs = 4;               // pixels per thread and round the partial-count rule assumes (parts_for)
constexpr int kUa = 4;               // float4 per thread and tile in the elementwise kernels

typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldnt(const float* p) {
  const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---- per-channel sums over the pixels -----------------------------------------------------------------------------------
// MODE 0: {sum z, sum z^2};  MODE 1: {sum dx, sum dx*zhat} with dx = g * [y > 0] * jac(a*z + b) (the quantiser's backward in
// front of the batch-norm's);  MODE 2: the same sums for a GIVEN dx = g (plain batch-norm backward: the ADMM sites' bn3 and the
// downsample branch's batch-norm).  part: [gridDim.x][C][2] doubles.
// ReLU mask (round 4): `mask` != nullptr replaces the fp32 y (4 B/element in each backward pass) by ONE BIT per element, written by
// the forward's apply pass (bnq_apply_fwd_kernel).  Layout per group: for every chunk of 64 consecutive float4 (vec index i,
// chunk i >> 6) four 64-bit words, one per float4 component, bit (i & 63) = [y > 0] of that component: the wave-wide compare
// results of the forward, stored as they come.  mask_gstride: 64-bit words per group.
__device__ __forceinline__ void mask_bits4(const unsigned long long* __restrict__ mask, int64_t i, bool (&m)[4]) {
  const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(mask + (i >> 6) * 4);
  const ulonglong2 a = p2[0], b = p2[1];
  const int sh = (int)(i & 63);
  m[0] = (a.x >> sh) & 1ull; m[1] = (a.y >> sh) & 1ull; m[2] = (b.x >> sh) & 1ull; m[3] = (b.y >> sh) & 1ull;
}
__host__ __device__ __forceinline__ int64_t mask_words(int64_t nvec) { return ((nvec + 63) >> 6) * 4; }

// MODE 3 (round 4): the plain per-channel sums of TWO arrays {sum z, sum g} - the second stage over the per-column sums the small-batch
// site backward leaves (site1_bwd_kernel: P is then the number of columns per channel).
template <int MODE>
struct SumsDepth { static constexpr int U = MODE == 0 ? 8 : (MODE == 1 ? 4 : 6); };   // float4 per thread and array in flight

template <int MODE, int NT>
__global__ __launch_bounds__(NT) void bnq_sums_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                      const float* __restrict__ y, const float* __restrict__ ab,
                                                      const float* __restrict__ save, int64_t P, int C, float r, int relu,
                                                      double* __restrict__ part,
                                                      const unsigned long long* __restrict__ mask = nullptr) {
  constexpr int kUs = SumsDepth<MODE>::U;        // (round 4: one array of 64 B per thread kept 8 MB in flight on the chip: 4.9 TB/s)
  __shared__ double sm[NT][8];
  // blockIdx.y = group: the batch slices of a merged multi-pass tensor ([groups][P][C] back to back), each with its own
  // statistics (ab / save: [groups][2][C]; part: [groups][gridDim.x][C][2])
  {
    const int64_t go = (int64_t)blockIdx.y * P * C;
    z += go;
    if (g) g += go;
    if (y) y += go;
    if (mask) mask += (int64_t)blockIdx.y * mask_words(P * (C >> 2));
    if (ab) ab += (int64_t)blockIdx.y * 2 * C;
    if (save) save += (int64_t)blockIdx.y * 2 * C;
    part += (int64_t)blockIdx.y * gridDim.x * C * 2;
  }
  const int tid = threadIdx.x;
  const int C4 = C >> 2, slots = NT / C4;
  const int cq = tid % C4, slot = tid / C4;
  const int64_t per = (P + gridDim.x - 1) / gridDim.x;
  const int64_t p0 = (int64_t)blockIdx.x * per, p1 = (p0 + per < P) ? p0 + per : P;
  constexpr bool BWD = MODE != 0;
  constexpr bool STAT = MODE == 1 || MODE == 2;      // forms using (a, b, mean, invstd)
  float4 a4, b4, m4, i4;
  if (STAT) {
    a4 = *reinterpret_cast<const float4*>(ab + 4 * cq);
    b4 = *reinterpret_cast<const float4*>(ab + C + 4 * cq);
    m4 = *reinterpret_cast<const float4*>(save + 4 * cq);
    i4 = *reinterpret_cast<const float4*>(save + C + 4 * cq);
  }
  double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};
  for (int64_t base = p0 + slot; base < p1; base += (int64_t)slots * kUs) {
    float4 zv[kUs], gv[kUs], yv[kUs];
    bool mk[kUs][4];
#pragma unroll
    for (int u = 0; u < kUs; u++) {            // clamped address, masked below
      const int64_t px = base + (int64_t)u * slots;
      const int64_t off = (px < p1 ? px : p1 - 1) * C + 4 * cq;
      zv[u] = *reinterpret_cast<const float4*>(z + off);
      if (BWD) {
        gv[u] = *reinterpret_cast<const float4*>(g + off);
        if (MODE == 1 && relu) {
          if (mask) mask_bits4(mask, off >> 2, mk[u]);
          else yv[u] = *reinterpret_cast<const float4*>(y + off);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kUs; u++) {
      if (base + (int64_t)u * slots < p1) {
        const float ze[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
        if (!BWD) {
#pragma unroll
          for (int e = 0; e < 4; e++) { s0[e] += (double)ze[e]; s1[e] += (double)ze[e] * (double)ze[e]; }
        } else if (MODE == 3) {
          const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
          for (int e = 0; e < 4; e++) { s0[e] += (double)ze[e]; s1[e] += (double)ge[e]; }
        } else {
          const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
          const float ye[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
          const float ae[4] = {a4.x, a4.y, a4.z, a4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
          const float me[4] = {m4.x, m4.y, m4.z, m4.w}, ie[4] = {i4.x, i4.y, i4.z, i4.w};
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float dx = ge[e];
            if (MODE == 1) {
              const float x = __fmaf_rn(ae[e], ze[e], be[e]);
              const bool pos = mask ? mk[u][e] : (ye[e] > 0.f);
              const float gm = (relu && !pos) ? 0.f : ge[e];
              dx = gm * act_jac(x, r);
            }
            const float zh = (ze[e] - me[e]) * ie[e];
            s0[e] += (double)dx;
            s1[e] += (double)dx * (double)zh;
          }
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; e++) { sm[tid][e] = s0[e]; sm[tid][4 + e] = s1[e]; }
  __syncthreads();
  for (int c = tid; c < C; c += NT) {
    const int qd = c >> 2, e = c & 3;
    double a = 0, b = 0;
    for (int s = 0; s < slots; s++) { a += sm[s * C4 + qd][e]; b += sm[s * C4 + qd][4 + e]; }      // fixed order
    part[((int64_t)blockIdx.x * C + c) * 2] = a;
    part[((int64_t)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

// ---- finalisation: ONE WAVE per channel (4 channels per 256-thread block, grid = C / 4) ------------------------------------------
// Every lane takes every 64th partial of its channel - of TWO groups at a time, all loads in flight together: the kernel is a
// chain of memory round trips (a one-thread-per-channel loop over 512 partials measured ~100 us, 16 lanes per channel and one
// group after the other ~9 us: 4 + 4 dependent batches) - then a fixed butterfly over the 64 lanes.  Totals valid in every lane.
// U: partials per lane and round.  8 covers the <= kParts partials of bnq_sums_kernel in ONE round of loads; a convolution's epilogue
// leaves one partial per row tile (686 per batch slice at layer1 of configuration 5): U = 12 keeps those in one round too (round 6:
// with two rounds - two dependent trips to memory the previous kernel has only just written - the layer1 nodes took 11-14 us
// against 6-7 for the others)
template <int U = kParts / 64>
__device__ __forceinline__ void bnq_wave_totals2(const double* __restrict__ p0, const double* __restrict__ p1, int nparts, int C,
                                                 int c, int lane, double& a0, double& q0, double& a1, double& q1) {
  a0 = 0; q0 = 0; a1 = 0; q1 = 0;
  for (int base = 0; base < nparts; base += 64 * U) {
    double va0[U], vq0[U], va1[U], vq1[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int s = base + lane + 64 * u, sc = s < nparts ? s : nparts - 1;
      const double2 v0 = *reinterpret_cast<const double2*>(p0 + ((int64_t)sc * C + c) * 2);
      const double2 v1 = *reinterpret_cast<const double2*>(p1 + ((int64_t)sc * C + c) * 2);
      va0[u] = v0.x; vq0[u] = v0.y; va1[u] = v1.x; vq1[u] = v1.y;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (base + lane + 64 * u < nparts) { a0 += va0[u]; q0 += vq0[u]; a1 += va1[u]; q1 += vq1[u]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o, 64); q0 += __shfl_xor(q0, o, 64);
    a1 += __shfl_xor(a1, o, 64); q1 += __shfl_xor(q1, o, 64);
  }
}

// groups: the slices' statistics one after the other (running statistics updated in slice order, like successive forward
// passes of the module).
template <int U>
__global__ __launch_bounds__(kT) void bnq_finalize_kernel(const double* __restrict__ part, int nparts, int64_t P, int C,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          long long* __restrict__ nbt, float momentum, float eps,
                                                          float* __restrict__ ab, float* __restrict__ save, int groups) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += groups;
  if (c >= C) return;
  float rm = running_mean ? running_mean[c] : 0.f, rv = running_var ? running_var[c] : 0.f;
  const float gm = gamma ? gamma[c] : 1.0f, bt = beta ? beta[c] : 0.0f;
  const int64_t gstride = (int64_t)nparts * C * 2;
  for (int g0 = 0; g0 < groups; g0 += 2) {
    const bool two = g0 + 1 < groups;
    double a[2], q[2];
    bnq_wave_totals2<U>(part + g0 * gstride, part + (two ? g0 + 1 : g0) * gstride, nparts, C, c, lane, a[0], q[0], a[1], q[1]);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (j == 1 && !two) break;
      const double n = (double)P;
      const double mean = a[j] / n;
      double var = q[j] / n - mean * mean;
      if (var < 0) var = 0;
      const float invstd = (float)(1.0 / sqrt(var + (double)eps));
      const float av = gm * invstd;
      if (lane == 0) {
        float* abg = ab + (int64_t)(g0 + j) * 2 * C;
        float* svg = save + (int64_t)(g0 + j) * 2 * C;
        abg[c] = av;
        abg[C + c] = bt - (float)mean * av;
        svg[c] = (float)mean;
        svg[C + c] = invstd;
      }
      rm = (1.0f - momentum) * rm + momentum * (float)mean;
      rv = (1.0f - momentum) * rv + momentum * (float)(var * n / (n - 1.0));
    }
  }
  if (lane == 0) {
    if (running_mean) running_mean[c] = rm;
    if (running_var) running_var[c] = rv;
  }
}

// dgamma / dbeta: the SUM over the groups (one parameter, several slices); ktot: [groups][2][C]
// fix (or nullptr; round 4): the second sums came from x = a*z + b as sum dx * (x - beta) / gamma (the small-batch site backward's
// per-column sums); a channel with gamma == 0 carries no trace of z there and one with |gamma| << |beta| only noise (round 5,
// alignq_bn_col_ill in site_internal.h), so the wave forms its sum dx * zhat from dx and z directly (P strided reads: slow, and
// only for such channels - e.g. a zero-initialised last batch-norm of a residual branch and the step after it, dead channels of a
// pretrained network).
struct ZeroGammaFix {
  const float* dx; const float* z; const float* ab; const float* save;
};
__global__ __launch_bounds__(kT) void bnq_finalize_bwd_kernel(const double* __restrict__ part, int nparts, int64_t P, int C,
                                                              float* __restrict__ ktot, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int groups, int64_t Pfull = 0,
                                                              ZeroGammaFix fix = ZeroGammaFix{nullptr, nullptr, nullptr, nullptr}) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= C) return;
  if (Pfull == 0) Pfull = P;                    // (P: what the partials were summed over; Pfull: elements per channel and group)
  double ta = 0, tq = 0;
  const int64_t gstride = (int64_t)nparts * C * 2;
  for (int g0 = 0; g0 < groups; g0 += 2) {
    const bool two = g0 + 1 < groups;
    double a[2], q[2];
    bnq_wave_totals2(part + g0 * gstride, part + (two ? g0 + 1 : g0) * gstride, nparts, C, c, lane, a[0], q[0], a[1], q[1]);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (j == 1 && !two) break;
      if (fix.dx && alignq_bn_col_ill(fix.ab[(int64_t)(g0 + j) * 2 * C + c], fix.ab[(int64_t)(g0 + j) * 2 * C + C + c],
                                      fix.save[(int64_t)(g0 + j) * 2 * C + c], fix.save[(int64_t)(g0 + j) * 2 * C + C + c])) {
        const float* dxg = fix.dx + (int64_t)(g0 + j) * Pfull * C;
        const float* zg = fix.z + (int64_t)(g0 + j) * Pfull * C;
        const float mm = fix.save[(int64_t)(g0 + j) * 2 * C + c], ii = fix.save[(int64_t)(g0 + j) * 2 * C + C + c];
        double qq = 0;
        for (int64_t px = lane; px < Pfull; px += 64) qq += (double)dxg[px * C + c] * (double)((zg[px * C + c] - mm) * ii);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qq += __shfl_xor(qq, o, 64);
        q[j] = qq;
      }
      if (lane == 0) {
        ktot[(int64_t)(g0 + j) * 2 * C + c] = (float)(a[j] / (double)Pfull);
        ktot[(int64_t)(g0 + j) * 2 * C + C + c] = (float)(q[j] / (double)Pfull);
      }
      ta += a[j];
      tq += q[j];
    }
  }
  if (lane != 0) return;
  if (dbeta) dbeta[c] = (float)ta;
  if (dgamma) dgamma[c] = (float)tq;
}

// ---- in-kernel finalisation (round 4, "small" sites: one group, C <= kFinC, a few MB - configuration 1's CIFAR-size tensors) ----
// At these sizes bnq_finalize* are launches at the ~5 us node floor between two 5-10 us kernels.  With at most kFinParts partials
// per channel every workgroup of the APPLY pass can afford to finalise for itself (C x partials x 16 B <= 64 KB of L2 reads per
// workgroup, all of them the same lines, read by all its threads at once): the statistics kernel is followed by the apply kernel directly; workgroup 0 also writes
// what the module keeps (a, b, mean, invstd, running statistics / dgamma, dbeta).  Same arithmetic and summation order as the
// finalisation kernels (partials in index order, double).
constexpr int kFinC = 64, kFinParts = 128;      // C x partials x 16 B <= 64 KB per workgroup (fin_parts)

struct FinFwd {                 // bnq_finalize_kernel's arguments
  const double* part; int nparts; int64_t P;
  const float* gamma; const float* beta; float* running_mean; float* running_var; long long* nbt; float momentum, eps;
  float* ab_out; float* save_out;
};
struct FinBwd {                 // bnq_finalize_bwd_kernel's arguments
  const double* part; int nparts; int64_t P; float* dgamma; float* dbeta;
};

// Totals of channel c = tid % C over the partials, by ALL threads of the workgroup: thread (c, slice = tid / C) adds every
// (NT / C)-th partial (all its loads in flight together, rounds of 8), the slices are then added in slice order by the threads of
// slice 0: one or two memory round trips per workgroup instead of nparts / 8.  Valid in threads tid < C after the call.
template <int NT>
__device__ __forceinline__ void fin_totals(const double* __restrict__ part, int nparts, int C, double* __restrict__ sm, double& a,
                                           double& q) {
  const int tid = threadIdx.x, c = tid % C, sl = tid / C, slices = NT / C;
  double pa = 0, pq = 0;
  for (int s0 = sl; s0 < nparts; s0 += 8 * slices) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const double2*>(part + ((int64_t)min(s0 + u * slices, nparts - 1) * C + c) * 2);
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (s0 + u * slices < nparts) { pa += v[u].x; pq += v[u].y; }
  }
  sm[2 * tid] = pa;
  sm[2 * tid + 1] = pq;
  __syncthreads();
  a = 0; q = 0;
  if (tid < C)
    for (int j = 0; j < slices; j++) { a += sm[2 * (j * C + tid)]; q += sm[2 * (j * C + tid) + 1]; }
}

// ---- elementwise passes: tiles of kUa x 256 float4 per block; 256 % (C/4) == 0 keeps a thread on one channel quad ----------
// res (or nullptr): a tensor of z's shape added to the quantised value before the ReLU (the CDF-only block's `out += shortcut;
// out = F.relu(out)`, cdf_alignment/resnet-20-cifar-10/model/resnet.py:76-78); the mask bits are those of the stored value.
template <int FORMULA, int NT, bool FIN = false>
__global__ __launch_bounds__(NT) void bnq_apply_fwd_kernel(const float* __restrict__ z, const float* __restrict__ ab, int64_t nvec,
                                                           int C, int k, float r, int relu, float* __restrict__ y,
                                                           unsigned long long* __restrict__ mask = nullptr,
                                                           const float* __restrict__ res = nullptr, FinFwd fin = FinFwd{},
                                                           short* __restrict__ ybins = nullptr) {
  __shared__ __attribute__((aligned(16))) float nerf_lds[ALIGNQ_NERF_LDS_FLOATS];
  __shared__ __attribute__((aligned(16))) float ab_s[FIN ? 2 * kFinC : 4];
  nerf_tab_load(nerf_lds);
  __shared__ double fin_sm[FIN ? 2 * NT : 2];
  if (FIN) {
    double a, q;
    fin_totals<NT>(fin.part, fin.nparts, C, fin_sm, a, q);
    if ((int)threadIdx.x < C) {
      const int c = threadIdx.x;
      const double n = (double)fin.P;
      const double mean = a / n;
      double var = q / n - mean * mean;
      if (var < 0) var = 0;
      const float invstd = (float)(1.0 / sqrt(var + (double)fin.eps));
      const float av = (fin.gamma ? fin.gamma[c] : 1.0f) * invstd;
      const float bv = (fin.beta ? fin.beta[c] : 0.0f) - (float)mean * av;
      ab_s[c] = av;
      ab_s[C + c] = bv;
      if (blockIdx.x == 0) {
        fin.ab_out[c] = av; fin.ab_out[C + c] = bv;
        fin.save_out[c] = (float)mean; fin.save_out[C + c] = invstd;
        if (fin.running_mean) fin.running_mean[c] = (1.0f - fin.momentum) * fin.running_mean[c] + fin.momentum * (float)mean;
        if (fin.running_var) fin.running_var[c] = (1.0f - fin.momentum) * fin.running_var[c] + fin.momentum * (float)(var * n / (n - 1.0));
      }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && fin.nbt) *fin.nbt += 1;
  }
  __syncthreads();
  const NerfTab tab = nerf_tab(nerf_lds);
  const Levels nlev = make_levels(k, fabsf(r) <= 8.0f);
  z += (int64_t)blockIdx.y * nvec * 4;           // blockIdx.y = group (see bnq_sums_kernel)
  if (y) y += (int64_t)blockIdx.y * nvec * 4;
  if (ybins) ybins += (int64_t)blockIdx.y * nvec * 4;   // N2 on the Office path: the (ReLU-clamped) level index as int16
  if (res) res += (int64_t)blockIdx.y * nvec * 4;
  if (mask) mask += (int64_t)blockIdx.y * mask_words(nvec);
  if (!FIN) ab += (int64_t)blockIdx.y * 2 * C;
  const float* abp = FIN ? ab_s : ab;
  const int cq = threadIdx.x % (C >> 2);
  const float4 a4 = *reinterpret_cast<const float4*>(abp + 4 * cq);
  const float4 b4 = *reinterpret_cast<const float4*>(abp + C + 4 * cq);
  const float4* z4 = reinterpret_cast<const float4*>(z);
  float4* y4 = reinterpret_cast<float4*>(y);
  const int64_t stride = (int64_t)gridDim.x * NT * kUa;
  ALIGNQ_BOUNDED_SWITCH(nlev,
  for (int64_t i0 = (int64_t)blockIdx.x * (NT * kUa) + threadIdx.x; i0 < nvec; i0 += stride) {
    float4 v[kUa], rv[kUa];
_Pragma("unroll")
    for (int u = 0; u < kUa; u++) {
      const int64_t i = i0 + u * NT;
      v[u] = z4[i < nvec ? i : i0];
      if (res) rv[u] = reinterpret_cast<const float4*>(res)[i < nvec ? i : i0];
    }
_Pragma("unroll")
    for (int u = 0; u < kUa; u++) {
      const int64_t i = i0 + u * NT;
      float4 o;
      float t;
      if (FORMULA == 2) {          // no quantiser: the batch-norm output itself (downsample branch)
        o = make_float4(__fmaf_rn(a4.x, v[u].x, b4.x), __fmaf_rn(a4.y, v[u].y, b4.y), __fmaf_rn(a4.z, v[u].z, b4.z),
                        __fmaf_rn(a4.w, v[u].w, b4.w));
      } else {
        constexpr int FQ = FORMULA == 2 ? 0 : FORMULA;
        float bx, by, bz, bw;
        o.x = act_quant1<FQ, kBounded>(__fmaf_rn(a4.x, v[u].x, b4.x), k, nlev, r, &t, &bx, tab);
        o.y = act_quant1<FQ, kBounded>(__fmaf_rn(a4.y, v[u].y, b4.y), k, nlev, r, &t, &by, tab);
        o.z = act_quant1<FQ, kBounded>(__fmaf_rn(a4.z, v[u].z, b4.z), k, nlev, r, &t, &bz, tab);
        o.w = act_quant1<FQ, kBounded>(__fmaf_rn(a4.w, v[u].w, b4.w), k, nlev, r, &t, &bw, tab);
        if (ybins && i < nvec) {       // (host: only without a residual, ADMM formula, index range within int16)
          if (relu) { bx = fmaxf(bx, 0.f); by = fmaxf(by, 0.f); bz = fmaxf(bz, 0.f); bw = fmaxf(bw, 0.f); }
          reinterpret_cast<short4*>(ybins)[i] = make_short4((short)bx, (short)by, (short)bz, (short)bw);
        }
      }
      if (res) { o.x += rv[u].x; o.y += rv[u].y; o.z += rv[u].z; o.w += rv[u].w; }
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (y && i < nvec) y4[i] = o;
      if (mask) {
        // the wave's 64 float4 are the consecutive vec indices of ONE chunk (tile bases and NT are multiples of 64): four
        // wave-wide compares = the chunk's four words, stored by lane 0 (a chunk is written whole or - beyond nvec - not at all)
        const bool in = i < nvec;
        const unsigned long long bx = __ballot(in && o.x > 0.f), by = __ballot(in && o.y > 0.f);
        const unsigned long long bz = __ballot(in && o.z > 0.f), bw = __ballot(in && o.w > 0.f);
        const int lane = threadIdx.x & 63;
        const int64_t i_w = i - lane;                 // the wave's first vec index
        if (lane == 0 && i_w < nvec) {                // (lane 0 is in the loop whenever any lane of its wave is)
          ulonglong2* mw = reinterpret_cast<ulonglong2*>(mask + (i_w >> 6) * 4);
          mw[0] = make_ulonglong2(bx, by);
          mw[1] = make_ulonglong2(bz, bw);
        }
      }
    }
  })
}

// g and dz may be the SAME buffer (alignq_bnq_bwd_dx in place: every thread reads its own elements before it writes them),
// hence no __restrict__ on the two.  dres (or nullptr): receives the masked upstream gradient g * [y > 0] = the gradient of the
// forward's residual operand.  FIN: see above (ktot is then formed here from the sums kernel's partials).
template <int NT, bool FIN = false>
__global__ __launch_bounds__(NT) void bnq_apply_bwd_kernel(const float* g, const float* __restrict__ z,
                                                           const float* __restrict__ y, const float* __restrict__ ab,
                                                           const float* __restrict__ save, const float* __restrict__ ktot,
                                                           int64_t nvec, int C, float r, int relu, int from_dx,
                                                           float* dz, const unsigned long long* __restrict__ mask = nullptr,
                                                           float* __restrict__ dres = nullptr, FinBwd fin = FinBwd{}) {
  constexpr int U = 2;
  __shared__ __attribute__((aligned(16))) float k_s[FIN ? 2 * kFinC : 4];
  __shared__ double fin_sm[FIN ? 2 * NT : 2];
  if (FIN) {
    double a, q;
    fin_totals<NT>(fin.part, fin.nparts, C, fin_sm, a, q);
    if ((int)threadIdx.x < C) {
      const int c = threadIdx.x;
      k_s[c] = (float)(a / (double)fin.P);
      k_s[C + c] = (float)(q / (double)fin.P);
      if (blockIdx.x == 0) {
        if (fin.dbeta) fin.dbeta[c] = (float)a;
        if (fin.dgamma) fin.dgamma[c] = (float)q;
      }
    }
    __syncthreads();
    ktot = k_s;
  }
  {
    const int64_t go = (int64_t)blockIdx.y * nvec * 4;       // blockIdx.y = group
    g += go; z += go; dz += go;
    if (y) y += go;
    if (dres) dres += go;
    if (mask) mask += (int64_t)blockIdx.y * mask_words(nvec);
    ab += (int64_t)blockIdx.y * 2 * C; save += (int64_t)blockIdx.y * 2 * C;
    if (!FIN) ktot += (int64_t)blockIdx.y * 2 * C;
  }
  const int cq = threadIdx.x % (C >> 2);
  const float4 a4 = *reinterpret_cast<const float4*>(ab + 4 * cq), b4 = *reinterpret_cast<const float4*>(ab + C + 4 * cq);
  const float4 m4 = *reinterpret_cast<const float4*>(save + 4 * cq), i4 = *reinterpret_cast<const float4*>(save + C + 4 * cq);
  const float4 k0 = *reinterpret_cast<const float4*>(ktot + 4 * cq), k1 = *reinterpret_cast<const float4*>(ktot + C + 4 * cq);
  const float ae[4] = {a4.x, a4.y, a4.z, a4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
  const float me[4] = {m4.x, m4.y, m4.z, m4.w}, ie[4] = {i4.x, i4.y, i4.z, i4.w};
  const float k0e[4] = {k0.x, k0.y, k0.z, k0.w}, k1e[4] = {k1.x, k1.y, k1.z, k1.w};
  const int64_t stride = (int64_t)gridDim.x * NT * U;
  for (int64_t i0 = (int64_t)blockIdx.x * (NT * U) + threadIdx.x; i0 < nvec; i0 += stride) {
    float4 gv[U], zv[U], yv[U];
    bool mk[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = i0 + u * NT, ic = i < nvec ? i : i0;
      gv[u] = ldnt(g + 4 * ic);                      // the upstream gradient is read here for the last time
      zv[u] = *reinterpret_cast<const float4*>(z + 4 * ic);
      if (relu) {
        if (mask) mask_bits4(mask, ic, mk[u]);
        else yv[u] = *reinterpret_cast<const float4*>(y + 4 * ic);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = i0 + u * NT;
      const float ge[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, ze[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
      const float ye[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
      float o[4], gmv[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float dx = ge[e];
        gmv[e] = ge[e];
        if (!from_dx) {            // launch-uniform
          const float x = __fmaf_rn(ae[e], ze[e], be[e]);
          const bool pos = mask ? mk[u][e] : (ye[e] > 0.f);
          const float gm = (relu && !pos) ? 0.f : ge[e];
          gmv[e] = gm;
          dx = gm * act_jac(x, r);
        }
        const float zh = (ze[e] - me[e]) * ie[e];
        o[e] = ae[e] * (dx - k0e[e] - zh * k1e[e]);
      }
      if (i < nvec) *reinterpret_cast<float4*>(dz + 4 * i) = make_float4(o[0], o[1], o[2], o[3]);
      if (dres && i < nvec) *reinterpret_cast<float4*>(dres + 4 * i) = make_float4(gmv[0], gmv[1], gmv[2], gmv[3]);
    }
  }
}

inline bool bad_c(int C) { return C < 4 || C > 2048 || (C & (C - 1)) != 0; }   // threads = slots x C/4 quads (512 at C = 2048)
inline int threads_for(int C) { return (C >> 2) > kT ? 512 : kT; }
inline int tiles(int64_t nvec, int u, int nt = kT) {
  int64_t b = (nvec + (int64_t)nt * u - 1) / ((int64_t)nt * u);
  if (b < 1) b = 1;
  return (int)(b > 256 * 64 ? 256 * 64 : b);
}
// the reductions run 512-thread workgroups (round 4): at most 512 of them exist (kParts), i.e. two per CU - with 256 threads
// that is two waves per SIMD for a kernel that is one long chain of load rounds
inline int sums_threads(int C) {
  (void)C;
  return 512;
}
inline int parts_for(int64_t P, int C) {
  // every block should own at least a few pixel rounds: slots * kUs pixels per round
  const int slots = sums_threads(C) / (C >> 2);
  int64_t n = P / ((int64_t)slots * kUs);
  if (n < 1) n = 1;
  // ... and the partial image [parts][C][2] doubles is written once and read once: at most 4 MB of it per group (wide layers:
  // 512 partials of 2048 channels are 16.8 MB for a 22 MB tensor)
  const int64_t cap = ((int64_t)4 << 20) / ((int64_t)C * 16);
  if (n > cap) n = cap < 64 ? 64 : cap;
  return (int)(n > kParts ? kParts : n);
}

// launch with 256 threads, or 512 where a pixel has more than 256 channel quads (C = 2048: ResNet-50's last stage)
#define BNQ_NT(C, ...)                                   \
  do {                                                   \
    if (threads_for(C) == 512) {                         \
      constexpr int NTV = 512;                           \
      __VA_ARGS__;                                       \
    } else {                                             \
      constexpr int NTV = kT;                            \
      __VA_ARGS__;                                       \
    }                                                    \
  } while (0)

#define BNQ_SUMS_NT(C, ...)                              \
  do {                                                   \
    if (sums_threads(C) == 512) {                        \
      constexpr int NTV = 512;                           \
      __VA_ARGS__;                                       \
    } else {                                             \
      constexpr int NTV = kT;                            \
      __VA_ARGS__;                                       \
    }                                                    \
  } while (0)

// the finalisation with every partial of a channel in flight at once whenever they fit one round (see bnq_wave_totals2)
inline void launch_finalize(int np, int C, hipStream_t st, const double* part, int64_t P, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, long long* nbt, float momentum, float bn_eps, float* ab,
                            float* save, int groups) {
  const dim3 grid((C + 3) / 4), blk(kT);
  if (np > 64 * 8 && np <= 64 * 12)
    hipLaunchKernelGGL((bnq_finalize_kernel<12>), grid, blk, 0, st, part, np, P, C, gamma, beta, running_mean, running_var, nbt, momentum,
                       bn_eps, ab, save, groups);
  else
    hipLaunchKernelGGL((bnq_finalize_kernel<8>), grid, blk, 0, st, part, np, P, C, gamma, beta, running_mean, running_var, nbt, momentum,
                       bn_eps, ab, save, groups);
}

}  // namespace

extern "C" {

// groups: the tensor is [groups][P][C] -- batch slices of a merged multi-pass traversal, each normalised with its own batch
// statistics (ab / save are [groups][2][C]); one launch per kernel covers all of them (blockIdx.y = group)
size_t alignq_bnq_ws_bytes(int C, int groups) {
  if (C <= 0 || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS) return 0;
  return (size_t)groups * ((size_t)kParts * C * 2 * sizeof(double) + (size_t)2 * C * sizeof(float));
}

namespace {
inline bool bad_groups(int groups) { return groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS; }
inline float* ktot_of(void* ws, int C, int groups) {
  return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (size_t)groups * kParts * C * 2 * sizeof(double));
}
}  // namespace

size_t alignq_bnq_mask_bytes(int64_t P, int C, int groups) {
  if (P < 1 || C < 4 || groups < 1 || groups > ALIGNQ_BNQ_MAX_GROUPS) return 0;
  return (size_t)groups * (size_t)mask_words(P * (C >> 2)) * sizeof(unsigned long long);
}

namespace {
// "small" sites finalise inside the apply kernels (see kFinC): one group, few channels, a tensor of a few MB
inline bool fin_small(int64_t P, int C, int groups) {
  return groups == 1 && C <= kFinC && P * C <= ((int64_t)4 << 20);
}
inline int fin_parts(int64_t P, int C) {
  const int n = parts_for(P, C), cap = (64 * 1024) / (C * 16) < kFinParts ? (64 * 1024) / (C * 16) : kFinParts;
  return n < cap ? n : cap;
}
inline int fin_grid(int64_t nvec, int u) { const int t = tiles(nvec, u); return t < 512 ? t : 512; }
}  // namespace

int alignq_bnq_fwd(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, int k, float act_range,
                   int formula, int relu, const float* residual, float* ab, float* save, float* y, void* mask, void* ws,
                   void* stream) {
  return alignq_bnq_fwd_parts(z, P, C, groups, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, bn_eps, k,
                              act_range, formula, relu, residual, ab, save, y, mask, ws, nullptr, 0, nullptr, stream);
}

int alignq_bnq_fwd_parts(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, int k, float act_range,
                         int formula, int relu, const float* residual, float* ab, float* save, float* y, void* mask, void* ws,
                         const double* conv_part, int conv_parts, void* bins_out, void* stream) {
  if (!z || !ab || !save || (!y && !bins_out) || (!ws && !conv_part) || P < 2 || bad_groups(groups) || (conv_part && conv_parts < 1))
    return ALIGNQ_EINVAL;
  if (bins_out) {      // the int16 level index: ADMM / Office formula, no residual, |index| <= act_range * (2^k - 1) within int16
    if (formula != ALIGNQ_FORMULA_ADMM || residual || k < 1 || k > 14 || !(fabsf(act_range) * (float)((1 << k) - 1) <= 32767.0f) ||
        (reinterpret_cast<uintptr_t>(bins_out) & 7))
      return ALIGNQ_EINVAL;
  }
  short* yb = reinterpret_cast<short*>(bins_out);
  if ((reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(residual)) & 15) return ALIGNQ_EINVAL;
  unsigned long long* mk = reinterpret_cast<unsigned long long*>(mask);
  if (!((k >= 1 && k <= 16) || k == 32)) return ALIGNQ_EINVAL;
  if (formula != ALIGNQ_FORMULA_ADMM && formula != ALIGNQ_FORMULA_CDF) return ALIGNQ_EINVAL;
  if (bad_c(C)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y)) & 15) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  const int64_t nvec = P * (C >> 2);
  if (!conv_part && fin_small(P, C, groups)) {
    const int np = fin_parts(P, C);
    BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<0, NTV>), dim3(np, 1), dim3(NTV), 0, st, z, nullptr, nullptr, nullptr, nullptr, P, C,
                       act_range, 0, part));
    const FinFwd fin{part, np, P, gamma, beta, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), momentum,
                     bn_eps, ab, save};
    const dim3 grid(fin_grid(nvec, kUa), 1);
    if (formula == ALIGNQ_FORMULA_ADMM)
      hipLaunchKernelGGL((bnq_apply_fwd_kernel<0, kT, true>), grid, dim3(kT), 0, st, z, (const float*)ab, nvec, C, k, act_range, relu, y, mk,
                         residual, fin, yb);
    else
      hipLaunchKernelGGL((bnq_apply_fwd_kernel<1, kT, true>), grid, dim3(kT), 0, st, z, (const float*)ab, nvec, C, k, act_range, relu, y, mk,
                         residual, fin);
    return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
  }
  int np = parts_for(P, C);
  if (conv_part) {      // the producing convolution's epilogue already summed z and z^2 per row tile (alignq_qconv_fwd bn_part)
    part = const_cast<double*>(conv_part);
    np = conv_parts;
  } else {
    BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<0, NTV>), dim3(np, groups), dim3(NTV), 0, st, z, nullptr, nullptr, nullptr, nullptr, P, C,
                       act_range, 0, part));
  }
  launch_finalize(np, C, st, (const double*)part, P, gamma, beta, running_mean, running_var,
                  reinterpret_cast<long long*>(num_batches_tracked), momentum, bn_eps, ab, save, groups);
  if (formula == ALIGNQ_FORMULA_ADMM)
    BNQ_NT(C, hipLaunchKernelGGL((bnq_apply_fwd_kernel<0, NTV>), dim3(tiles(nvec, kUa, NTV), groups), dim3(NTV), 0, st, z, (const float*)ab, nvec, C, k,
                       act_range, relu, y, mk, residual, FinFwd{}, yb));
  else
    BNQ_NT(C, hipLaunchKernelGGL((bnq_apply_fwd_kernel<1, NTV>), dim3(tiles(nvec, kUa, NTV), groups), dim3(NTV), 0, st, z, (const float*)ab, nvec, C, k,
                       act_range, relu, y, mk, residual));
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

int alignq_bnq_bwd(const float* g, const float* z, const float* y, const void* mask, const float* ab, const float* save, int64_t P,
                   int C, int groups, float act_range, int relu, float* dz, float* dres, float* dgamma, float* dbeta, void* ws,
                   void* stream) {
  if (!g || !z || !ab || !save || !dz || !ws || P < 2 || (relu && !y && !mask) || bad_groups(groups)) return ALIGNQ_EINVAL;
  if ((reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(dres)) & 15) return ALIGNQ_EINVAL;
  const unsigned long long* mk = reinterpret_cast<const unsigned long long*>(mask);
  if (mk) y = nullptr;          // one source for the ReLU mask: the bits when they are given
  if (bad_c(C)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dz) |
       reinterpret_cast<uintptr_t>(y)) & 15)
    return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  float* ktot = ktot_of(ws, C, groups);
  const int64_t nvec = P * (C >> 2);
  if (fin_small(P, C, groups)) {
    const int np = fin_parts(P, C);
    BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<1, NTV>), dim3(np, 1), dim3(NTV), 0, st, z, g, y, ab, save, P, C, act_range, relu, part, mk));
    const FinBwd fin{part, np, P, dgamma, dbeta};
    hipLaunchKernelGGL((bnq_apply_bwd_kernel<kT, true>), dim3(fin_grid(nvec, 2), 1), dim3(kT), 0, st, g, z, y, ab, save, (const float*)ktot, nvec,
                       C, act_range, relu, 0, dz, mk, dres, fin);
    return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
  }
  const int np = parts_for(P, C);
  BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<1, NTV>), dim3(np, groups), dim3(NTV), 0, st, z, g, y, ab, save, P, C, act_range, relu, part, mk));
  hipLaunchKernelGGL(bnq_finalize_bwd_kernel, dim3((C + 3) / 4), dim3(kT), 0, st, (const double*)part, np, P, C, ktot,
                     dgamma, dbeta, groups);
  BNQ_NT(C, hipLaunchKernelGGL((bnq_apply_bwd_kernel<NTV>), dim3(tiles(nvec, 2, NTV), groups), dim3(NTV), 0, st, g, z, y, ab, save, (const float*)ktot, nvec,
                     C, act_range, relu, 0, dz, mk, dres));
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

// The batch-norm alone (no quantiser behind it in the same chain): statistics -> (a, b); y = a*z + b; backward from a given dx.
int alignq_bnq_stats(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, float* ab, float* save,
                     void* ws, void* stream) {
  return alignq_bnq_stats_parts(z, P, C, groups, gamma, beta, running_mean, running_var, num_batches_tracked, momentum, bn_eps, ab,
                                save, ws, nullptr, 0, stream);
}

int alignq_bnq_stats_parts(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, float* ab, float* save,
                           void* ws, const double* conv_part, int conv_parts, void* stream) {
  if (!ab || !save || P < 2 || bad_groups(groups)) return ALIGNQ_EINVAL;
  if (conv_part ? conv_parts < 1 : (!z || !ws)) return ALIGNQ_EINVAL;
  if (bad_c(C)) return ALIGNQ_EUNSUPPORTED;
  if (reinterpret_cast<uintptr_t>(z) & 15) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  int np = parts_for(P, C);
  if (conv_part) {
    part = const_cast<double*>(conv_part);
    np = conv_parts;
  } else {
    BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<0, NTV>), dim3(np, groups), dim3(NTV), 0, st, z, nullptr, nullptr, nullptr, nullptr, P, C, 0.f, 0, part));
  }
  launch_finalize(np, C, st, (const double*)part, P, gamma, beta, running_mean, running_var,
                  reinterpret_cast<long long*>(num_batches_tracked), momentum, bn_eps, ab, save, groups);
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

int alignq_bnq_affine(const float* z, const float* ab, int64_t P, int C, int groups, float* y, void* stream) {
  if (!z || !ab || !y || P < 1 || bad_groups(groups)) return ALIGNQ_EINVAL;
  if (bad_c(C)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(y)) & 15) return ALIGNQ_EINVAL;
  const int64_t nvec = P * (C >> 2);
  BNQ_NT(C, hipLaunchKernelGGL((bnq_apply_fwd_kernel<2, NTV>), dim3(tiles(nvec, kUa, NTV), groups), dim3(NTV), 0, (hipStream_t)stream, z, ab, nvec, C, 32, 1.0f,
                     0, y));
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

int alignq_bnq_bwd_dx(const float* dx, const float* z, const float* ab, const float* save, int64_t P, int C, int groups,
                      float* dz, float* dgamma, float* dbeta, void* ws, void* stream) {
  if (!dx || !z || !ab || !save || !dz || !ws || P < 2 || bad_groups(groups)) return ALIGNQ_EINVAL;
  if (bad_c(C)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dz)) & 15) return ALIGNQ_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  double* part = reinterpret_cast<double*>(ws);
  float* ktot = ktot_of(ws, C, groups);
  const int np = parts_for(P, C);
  BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<2, NTV>), dim3(np, groups), dim3(NTV), 0, st, z, dx, nullptr, ab, save, P, C, 0.f, 0, part));
  hipLaunchKernelGGL(bnq_finalize_bwd_kernel, dim3((C + 3) / 4), dim3(kT), 0, st, (const double*)part, np, P, C, ktot,
                     dgamma, dbeta, groups);
  const int64_t nvec = P * (C >> 2);
  BNQ_NT(C, hipLaunchKernelGGL((bnq_apply_bwd_kernel<NTV>), dim3(tiles(nvec, 2, NTV), groups), dim3(NTV), 0, st, dx, z, nullptr, ab, save, (const float*)ktot,
                     nvec, C, 0.f, 0, 1, dz));
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}

}  // extern "C"

// The batch-norm backward when the small-batch site backward has left per-column sums (site1_bwd_kernel): [2][groups][HW * C] floats
namespace alignq_site {
int launch_bnq_bwd_from_cols(const float* cols, const float* dx, const float* z, const float* ab, const float* save, int64_t P,
                             int64_t HW, int C, int groups, float* dz, float* dgamma, float* dbeta, void* ws, hipStream_t st) {
  if (!cols || !dx || !z || !ab || !save || !dz || !ws || P < 2 || HW < 1 || bad_groups(groups)) return ALIGNQ_EINVAL;
  if (bad_c(C)) return ALIGNQ_EUNSUPPORTED;
  if ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(dz) |
       reinterpret_cast<uintptr_t>(cols)) & 15)
    return ALIGNQ_EINVAL;
  double* part = reinterpret_cast<double*>(ws);
  float* ktot = ktot_of(ws, C, groups);
  const int np = parts_for(HW, C);
  const float* cols1 = cols + (int64_t)groups * HW * C;
  BNQ_SUMS_NT(C, hipLaunchKernelGGL((bnq_sums_kernel<3, NTV>), dim3(np, groups), dim3(NTV), 0, st, cols, cols1, nullptr, nullptr, nullptr, HW, C,
                     0.f, 0, part));
  const ZeroGammaFix fix{dx, z, ab, save};
  hipLaunchKernelGGL(bnq_finalize_bwd_kernel, dim3((C + 3) / 4), dim3(kT), 0, st, (const double*)part, np, HW, C, ktot, dgamma, dbeta,
                     groups, P, fix);
  const int64_t nvec = P * (C >> 2);
  BNQ_NT(C, hipLaunchKernelGGL((bnq_apply_bwd_kernel<NTV>), dim3(tiles(nvec, 2, NTV), groups), dim3(NTV), 0, st, dx, z, nullptr, ab, save,
                     (const float*)ktot, nvec, C, 0.f, 0, 1, dz));
  return hipGetLastError() == hipSuccess ? 0 : ALIGNQ_EINVAL;
}
}  // namespace alignq_site
