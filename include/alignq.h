/* alignq.h — C ABI of libalignq_hip.so: AlignQ's quantize / correlation / ADMM hot path on MI355X (gfx950).
 *
 * The reference has no FFI for this path: it is pure Python on PyTorch and its
 * boundary is a Python module API (model/quantization.py, utils/admm.py, utils/optimizer.py).  The
 * package alignq_amd/ mirrors that API; underneath, its autograd Functions call THIS library through
 * ctypes (INTEGRATION.md shows the binding).  Each entry point cites the reference lines it replaces;
 * paths are relative to the reference root, "ADMM tree" = cdf_alignment_admm/resnet-56-cifar-10,
 * "CDF tree" = cdf_alignment/resnet-20-cifar-10, "Office tree" = cdf_alignment_admm/dann_office.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types, no exceptions across the boundary;
 *   - every function returns int: 0 = ok, >0 = hipError_t of the launch, <0 = ALIGNQ_E* argument error;
 *   - the CALLER owns every buffer (inputs, outputs, workspaces); all pointers are DEVICE pointers to
 *     contiguous fp32 unless stated; `*_ws_bytes` functions size the scratch buffers;
 *   - functions only ENQUEUE work on `stream` (a hipStream_t, may be NULL = default stream): no
 *     allocation, no synchronisation, no global mutable state => re-entrant, callable from the autograd
 *     thread, capturable into a hipGraph;
 *   - `formula`: ALIGNQ_FORMULA_ADMM (ADMM/Office trees: transform to [-1,1]*(r) first, then round with
 *     n=2^k-1) or ALIGNQ_FORMULA_CDF (CDF tree: round the cdf in [0,1], then map) — SURVEY.md §0-F5;
 *   - erf/exp are the repo's ALIGNQ-ERF32/EXP32: bit-identical to oracle/alignq_oracle.c.
 */
#ifndef ALIGNQ_H
#define ALIGNQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALIGNQ_ABI_VERSION 23

#define ALIGNQ_FORMULA_ADMM 0
#define ALIGNQ_FORMULA_CDF 1

#define ALIGNQ_EINVAL (-1)       /* bad argument (null pointer, k out of range, n <= 0 ...) */
#define ALIGNQ_EUNSUPPORTED (-2) /* shape outside what the kernels handle (e.g. batch > ALIGNQ_MAX_BATCH) */

#define ALIGNQ_MAX_BATCH 128 /* rows of a correlation site the FUSED kernels (alignq_site_*) hold on chip */
#define ALIGNQ_MAX_CORR_BATCH 1024 /* rows alignq_corr_fwd / _bwd take: above 128 a blocked Gram (column statistics over all
                                      rows, 128x128 output blocks, exact fp32) - the reference takes any batch
                                      (model/quantization.py:134-137, utils/admm.py:17-27); the module layer composes the
                                      ADMM site from it for B > 128 */

int alignq_abi_version(void);
/* message for a return code of this library (static string) */
const char* alignq_strerror(int code);

/* ---- R1: uniform_quantize(k).forward alone (model/quantization.py:19-31, identical in all trees):
 * y = round_half_even(x*n)/n with n = 2^k-1; k==1 -> sign(x); k==32 -> x.  Backward is the identity
 * (straight-through, :34-36) and needs no kernel.                                                    */
int alignq_uniform_quantize(const float* x, float* y, int64_t n, int k, void* stream);

/* ---- R1/R2/R4: activation CDF transform + uniform quantize/dequantize --------------------------
 * forward: ADMM tree model/quantization.py:109-110 (+cdf :49-56, qfn :23-31); CDF tree :97-98;
 *          Office tree :103-104.  x,xq: [n].  k in {1..16, 32}; k==32 writes the transform itself.
 * bins (optional, may be NULL): int32 bin index round(t*n) per element (parity instrumentation).   */
int alignq_act_quant_fwd(const float* x, float* xq, int32_t* bins, int64_t n, int k, float act_range,
                         int formula, void* stream);
/* backward (STE of :34-36 chained through the transform): dx = g * act_range * 2*phi(x).            */
int alignq_act_quant_bwd(const float* g, const float* x, float* dx, int64_t n, float act_range,
                         void* stream);

/* The same pair with a ReLU after the quantiser — `self.relu(self.act_q1(self.bn1(...)))` of the Office bottleneck
 * (cdf_alignment_admm/dann_office/model/resnet.py:137-138, :142-143) — in one pass each way: y = relu(x_q); backward
 * dx = (y > 0 ? g : 0) * d t / d x. */
int alignq_act_quant_relu_fwd(const float* x, float* y, int64_t N, int k, float act_range, int formula, void* stream);
int alignq_act_quant_relu_bwd(const float* g, const float* x, const float* y, float* dx, int64_t N, float act_range,
                              void* stream);

/* ---- N2: integer bin storage.  The quantised activation is an integer level index idx = round(t*n)
 * (model/quantization.py:23-31 `torch.round(input * n) / n`; callers :109-110, Conv2d_Q.forward :149-154 consumes it): stored
 * narrow it costs 1-2 B per element instead of the 4 B of the dequantised fp32 value.
 *   ADMM / Office formula: idx in [-r*n, r*n], SIGNED: int8 while r*n <= 127 (k <= 6 at act_range 2), else int16 (8-bit:
 *   1021 levels);  CDF-tree formula: idx in [0, n], UNSIGNED: uint8 for k <= 8, else uint16.
 * alignq_bin_bytes: bytes per stored index (1 or 2; 0: no packed form for these arguments, e.g. k == 32).
 * alignq_act_quant_fwd_packed: as alignq_act_quant_fwd, but writes the (pre-ReLU) index in the narrow type; xq (optional)
 *   additionally receives [relu](x_q) in fp32.  alignq_bins_dequant: y = [relu](value(idx)), the same fp32 value as the x_q of
 *   the fused quantiser (same IEEE operations; bit-identical except that an integer 0 cannot carry the sign of x_q = -0.0).  alignq_act_quant_bwd_packed: dx = g * dt/dx with the ReLU mask (relu != 0)
 *   taken from the stored index (value(idx) > 0) instead of from an fp32 copy of relu(x_q).  16-byte aligned pointers. */
int alignq_bin_bytes(int k, float act_range, int formula);
int alignq_act_quant_fwd_packed(const float* x, float* xq, void* bins, int64_t n, int k, float act_range, int formula,
                                int relu, void* stream);
int alignq_bins_dequant(const void* bins, float* y, int64_t n, int k, float act_range, int formula, int relu, void* stream);
int alignq_act_quant_bwd_packed(const float* g, const float* x, const void* bins, float* dx, int64_t n, int k,
                                float act_range, int formula, int relu, void* stream);

/* ---- R3: weight quantisation -------------------------------------------------------------------
 * stats: torch.mean / torch.std (unbiased) over all n elements (model/quantization.py:78).
 *        ms: device float[2] = {mean, std}.  ws: alignq_weight_ws_bytes(n) bytes of scratch.        */
size_t alignq_weight_ws_bytes(int64_t n);
int alignq_weight_stats(const float* w, int64_t n, float* ms, void* ws, void* stream);
/* forward given ms: q = quantised weight; cdf_out / pdf_out (optional) = the module's weight_cdf /
 * weight_pdf attributes (ADMM tree :78-80; CDF tree :70-72); bins optional as above.               */
int alignq_weight_quant_fwd(const float* w, const float* ms, float* q, float* cdf_out, float* pdf_out,
                            int32_t* bins, int64_t n, int k, int formula, void* stream);
/* backward THROUGH mean and std (they are in the autograd graph):
 * dw_i = g_i P_i - mean(g P) - z_i/(n-1) * sum(g P z).  ws as above.                               */
int alignq_weight_quant_bwd(const float* g, const float* w, const float* ms, float* dw, int64_t n,
                            void* ws, void* stream);
/* Backward of the stand-alone cdf(m, s, quant_src).forward (ADMM tree model/quantization.py:41-59, CDF tree :37-50) when m and s
 * are tensors of the autograd graph (the reference builds cdf(torch.mean(w), torch.std(w), 'w'), :78): c = kc * Phi(z) (+ const),
 * pdf = 2 N(x; m, s), z = (x - m) / s.  gc / gp: upstream gradients of c / pdf (either may be NULL); ms = {m, s} on the device;
 * kc = 2 (ADMM / Office trees; x act_range for quant_src == 'a') or 1 (CDF tree).  dx [n] (may be NULL), dms[2] = {d/dm, d/ds};
 * ws: alignq_weight_ws_bytes(n).                                                                                               */
int alignq_cdf_bwd(const float* gc, const float* gp, const float* x, const float* ms, float kc, float* dx, float* dms, int64_t n,
                   void* ws, void* stream);

/* ---- R4+R5(+R6): fused ADMM site: quantise + both sample-correlation matrices (+ ADMM loss) ---------
 * x: [B,F] (the [B,C,H,W] activation viewed as [B,-1]); 2 <= B <= ALIGNQ_MAX_BATCH.
 * Computes xq (as alignq_act_quant_fwd, ADMM formula) and
 *   D = corr(t,t) - corr(x,x),  t = act_range*(2*Phi(x)-1)  (pre-round),
 *   corr(v,v) = Vh Vh^T / F, Vh = (v - mean_b v)/(std_b v + eps)     (model/quantization.py:115-122,
 *   corr :134-137; Office tree corr :158-161 uses eps = 1e-5).
 * stats: [4][F] out (mean_x, 1/(std_x+eps), mean_t, 1/(std_t+eps)) consumed by the backward.
 * xq may be NULL (correlation only).  ws: alignq_site_ws_bytes(B,F) bytes (partial slabs + reduction tail).
 * alignq_site_fwd = alignq_site_partials (per-tile quantise + Gram partial slabs -> ws) followed by
 * alignq_site_reduce (deterministic slab reduction ws -> D, scaled by 1/F).
 * alignq_site_reduce_loss = the same reduction plus the ADMM loss of utils/admm.py:24-33 in the same launch
 * (alterD, gamma: [dim,dim] sliced [:B,:B]): scal = device float[4] {loss, rho/2/(n*rms), 1/n, rms}, n = B*B.
 * Round 4: alignq_site_fwd and alignq_site_bwd (below) also take ALIGNQ_MAX_BATCH < B <= ALIGNQ_MAX_CORR_BATCH (the reference
 * takes any batch): pair kernels on the blocked exact-fp32 Gram of corr_large_kernels.hip (stats required; ws sizes from the
 * same two query functions); the split entry points (_partials / _reduce / _bwd_apply / _bwd_fused) stay at B <= 128.       */
size_t alignq_site_ws_bytes(int B, int64_t F);
int alignq_site_fwd(const float* x, int B, int64_t F, int k, float act_range, float eps, float* xq,
                    float* D, float* stats, void* ws, void* stream);
int alignq_site_partials(const float* x, int B, int64_t F, int k, float act_range, float eps, float* xq,
                         float* stats, void* ws, void* stream);
/* Small batches (B <= 32; ALIGNQ_EUNSUPPORTED otherwise): alignq_site_partials with the Office bottleneck's
 * `out += identity; out = self.relu(out)` (cdf_alignment_admm/dann_office/model/resnet.py:153-154) folded into the store:
 * y = [relu](x_q + residual) (residual may be NULL).  (The backward keeps the ReLU mask as its own pass: the masked gradient
 * is an output in its own right — the residual's gradient — so folding it saves no bytes; measured slower.) */
int alignq_site_partials_res(const float* x, int B, int64_t F, int k, float act_range, float eps, const float* residual,
                             int relu, float* y, float* stats, void* ws, void* stream);
int alignq_site_reduce(const void* ws, int B, int64_t F, float* D, void* stream);
int alignq_site_reduce_loss(void* ws, int B, int64_t F, float* D, const float* alterD, const float* gamma,
                            int dim, float mu, float rho, float* scal, void* stream);
/* backward: dx = g*dt/dx + d(corr pair)/dx.  g may be NULL.  ws: alignq_site_bwd_ws_bytes(B) (holds sym(dD)).
 * dD_scale: optional DEVICE scalar (the upstream gradient of the scalar loss).
 *   alignq_site_bwd       : explicit upstream gradient dD [B,B] w.r.t. D;
 *   alignq_site_bwd_fused : dD is the ADMM-loss gradient, rebuilt from (D, alterD, gamma, scal of
 *                           alignq_site_reduce_loss); also writes the parameter gradients of the loss,
 *                           dalterD / dgamma [dim,dim] (already multiplied by dD_scale; may be NULL).
 * alignq_site_bwd_ws_bytes: B <= ALIGNQ_MAX_BATCH: 128 KB (fp32 S + its bf16 image); above (alignq_site_bwd / alignq_corr_bwd on the
 * blocked form): S zero-padded to the backward's tiling, [32 ceil(B/32)][256 | 512 | 1024] floats (<= 4 MB), always ask.       */
size_t alignq_site_bwd_ws_bytes(int B);
/* second launch of both forms alone: S = the buffer (alignq_site_bwd_ws_bytes(B) bytes) the first launch (site_prep_kernel:
 * alignq_site_prep_fused[_multi], or the first half of alignq_site_bwd / _bwd_fused) leaves: the prepared, scaled,
 * symmetrised dD [B,B] in fp32, followed at byte offset 65536 by its bf16 hi/lo split in MFMA fragment order (read by the
 * 64 < B <= 128 kernels; the prep kernel splits each element once instead of every workgroup of the backward).        */
int alignq_site_bwd_apply(const float* g, const float* S, const float* x, const float* stats, int B, int64_t F,
                          float act_range, float eps, float* dx, void* stream);
int alignq_site_bwd(const float* g, const float* dD, const float* dD_scale, const float* x,
                    const float* stats, int B, int64_t F, float act_range, float eps, float* dx, void* ws,
                    void* stream);
int alignq_site_bwd_fused(const float* g, const float* D, const float* alterD, const float* gamma, int dim,
                          const float* scal, float mu, const float* dD_scale, const float* x,
                          const float* stats, int B, int64_t F, float act_range, float eps, float* dx,
                          float* dalterD, float* dgamma, void* ws, void* stream);
/* corr(x,x) alone (module-level `corr`, model/quantization.py:134-137): G [B,B]; stats [2][F];
 * 2 <= B <= ALIGNQ_MAX_CORR_BATCH (B > ALIGNQ_MAX_BATCH: blocked form, stats required);
 * ws: alignq_site_ws_bytes(B,F) forward, alignq_site_bwd_ws_bytes(B) backward.                               */
int alignq_corr_fwd(const float* x, int B, int64_t F, float eps, float* G, float* stats, void* ws,
                    void* stream);
int alignq_corr_bwd(const float* dG, const float* x, const float* stats, int B, int64_t F, float eps,
                    float* dx, void* ws, void* stream);

/* The GENERAL corr(x, y), y a different [B,F] matrix (same reference lines: `x.matmul(y.T) / x.shape[1]` after both
 * operands were standardised per feature over the batch): G = Xh Yh^T / F (not symmetric).  No BASELINE configuration
 * calls it with y != x; plain exact-fp32 kernels (corr_xy_kernels.hip).  stats [4][F] out: mean_x, 1/(std_x+eps), mean_y,
 * 1/(std_y+eps).  Backward: dx = d/dx sum(dG o G), dy likewise (either may be NULL).  ws: alignq_corr_xy_ws_bytes(B,F).  */
size_t alignq_corr_xy_ws_bytes(int B, int64_t F);
int alignq_corr_xy_fwd(const float* x, const float* y, int B, int64_t F, float eps, float* G, float* stats, void* ws,
                       void* stream);
int alignq_corr_xy_bwd(const float* dG, const float* x, const float* y, const float* stats, int B, int64_t F, float eps,
                       float* dx, float* dy, void* stream);

/* ---- R6: ADMM loss (utils/admm.py:24-33) --------------------------------------------------------
 * D: [b,b]; alterD, gamma: [dim,dim] with b <= dim (sliced [:b,:b]).  Writes loss (device scalar) and
 * the gradients dD [b,b], dalterD, dgamma [dim,dim] (zero outside the slice); any grad may be NULL.
 * ws: alignq_admm_ws_bytes(dim).                                                                    */
size_t alignq_admm_ws_bytes(int dim);
int alignq_admm_loss(const float* D, int b, const float* alterD, const float* gamma, int dim, float mu,
                     float rho, float* loss, float* dD, float* dalterD, float* dgamma, void* ws,
                     void* stream);

/* ---- R7: ADMM primal/dual update (utils/optimizer.py:97-124), batched over S sites ---------------
 * D_tab, alterD_tab, gamma_tab: HOST arrays of S DEVICE pointers (they are passed to the kernel by value,
 * so the call is graph-capturable and needs no device-side table); D_s is [b,b] (zero-padded to dim as
 * :104-105 does), alterD_s/gamma_s [dim,dim] are updated in place:
 *   V = pad(D)+gamma/rho; A = (1-(mu/rho)/|V|_F) V if |V|_F > mu/rho else 0; gamma += rho (pad(D)-A). */
int alignq_admm_update(const float* const* D_tab, float* const* alterD_tab, float* const* gamma_tab,
                       int S, int b, int dim, float mu, float rho, void* stream);
/* the same with a workspace (alignq_admm_update_ws_bytes(S, dim)): above dim = 128 (ADMM(dim = B_g) of the exact-global correlation)
 * 64 workgroups per site in two launches (partial |V|_F^2, then update) instead of one workgroup walking dim^2 elements; at
 * dim <= 128 or ws == NULL it IS alignq_admm_update.  The norm is summed in another order than there (last-bit differences).   */
size_t alignq_admm_update_ws_bytes(int S, int dim);
int alignq_admm_update_ws(const float* const* D_tab, float* const* alterD_tab, float* const* gamma_tab,
                          int S, int b, int dim, float mu, float rho, void* ws, void* stream);

/* ---- R8: SGD step (utils/optimizer.py:212-229,251,255) and the grad rewrite (:6-13,233-249) ------
 * d = g + wd*p; buf = first ? d : mom*buf + (1-damp)*d; dir = nesterov ? d + mom*buf : buf;
 * p -= lr*dir; g <- dir (what the reference leaves in p.grad for tensors outside idx).
 * buf may be NULL when mom == 0.                                                                    */
int alignq_sgd_step(float* p, float* g, float* buf, int64_t n, float lr, float mom, float damp, float wd,
                    int nesterov, int first, void* stream);
/* grad_out = dir * sigmoid_d(transform(w_cdf)) * w_pdf for tensors in idx.                          */
int alignq_sgd_grad_approx(const float* dir, const float* w_cdf, const float* w_pdf, float* grad_out,
                           int64_t n, int bitW, float lam, float lam2, void* stream);

/* ---- multi-tensor forms: ALL conv weights / ALL parameters of a model in one or two launches -------
 * Every `*_tab`/array argument is a HOST array of T entries (device pointers / sizes); entries travel to the
 * kernels by value in chunks, so these calls are hipGraph-capturable.  Semantics per tensor are exactly those
 * of the single-tensor entry points above.
 * weight fwd: ms is a DEVICE float[T][2] output (mean, std per tensor); cdf_out/pdf_out may be NULL.
 * ws: alignq_weight_multi_ws_bytes(T).                                                                 */
size_t alignq_weight_multi_ws_bytes(int T);
int alignq_weight_quant_fwd_multi(int T, const float* const* w, float* const* q, float* const* cdf_out,
                                  float* const* pdf_out, const int64_t* n, float* ms, int k, int formula,
                                  void* ws, void* stream);
int alignq_weight_quant_bwd_multi(int T, const float* const* g, const float* const* w, const float* ms,
                                  float* const* dw, const int64_t* n, void* ws, void* stream);
/* SGD over T parameters (utils/optimizer.py:212-255): w_cdf[t]/w_pdf[t] non-NULL marks tensor t as a member
 * of `idx` (its p.grad receives the sigmoid_d(transform(w_cdf))*w_pdf rewrite); first[t] != 0 marks a
 * momentum buffer created in this step.                                                               */
int alignq_sgd_step_multi(int T, float* const* p, float* const* g, float* const* buf, const int64_t* n,
                          const float* const* w_cdf, const float* const* w_pdf, const int32_t* first, float lr,
                          float mom, float damp, float wd, int nesterov, int bitW, float lam, float lam2,
                          void* stream);

/* The SGD step (alignq_sgd_step_multi) and the ADMM update (alignq_admm_update) of one iteration in ONE launch: the two touch
 * disjoint tensors when alterD / gamma are not SGD parameters (the CIFAR drivers: main.py:256-260 split them by name), so the
 * order utils' main.py:330-340 calls them in does not matter.  Same arguments and results as the two calls; more than 66
 * parameters or 22 sites fall back to exactly those two calls.                                                                */
int alignq_sgd_admm_step_multi(int T, float* const* p, float* const* g, float* const* buf, const int64_t* n,
                               const float* const* w_cdf, const float* const* w_pdf, const int32_t* first, float lr,
                               float mom, float damp, float wd, int nesterov, int bitW, float lam, float lam2, int S,
                               const float* const* D_tab, float* const* alterD_tab, float* const* gamma_tab, int b,
                               int dim, float mu, float rho, void* stream);

/* ---- all ADMM sites of a model in one launch each (64 < B <= 128): the slab reduction + loss of every site is off the
 * network's critical path (only x_q feeds the next layer), so a whole-model step can defer them to the end of the forward;
 * likewise ONE prep launch at the start of the backward.  Arrays are HOST arrays of S entries (passed by value to the
 * kernels in chunks of 32); per-site semantics are those of alignq_site_reduce_loss / alignq_site_prep_fused.          */
int alignq_site_reduce_loss_multi(int S, void* const* ws, float* const* D, const float* const* alterD,
                                  const float* const* gamma, float* const* scal, const int64_t* F, int B, int dim,
                                  float mu, float rho, void* stream);
int alignq_site_prep_fused_multi(int S, const float* const* D, const float* const* alterD, const float* const* gamma,
                                 const float* const* scal, const float* dD_scale, const int64_t* F, int B, int dim,
                                 float mu, float* const* S_out, float* const* dalterD, float* const* dgamma,
                                 void* stream);
/* alignq_site_reduce_loss_multi (the S sites still open at the end of the forward) and alignq_head_ce_fwd (head batch HB, with
 * ce_mean) as two roles of ONE launch (round 6, ABI 23): the head reads the last site's x_q, the reductions the sites' slabs -
 * neither needs the other (main.py:300-312: `outputs, trans_loss = net(inputs)`; `loss = criterion(outputs, targets)`).  Same
 * code and workgroup partition per role: same bits as the two calls.  *trans_total = sum_i scal_all[4 i] over ALL n_sites sites
 * of the step (scal_all: the consecutive `scal` rows, these S sites' among them), formed by the workgroup that closes the last of
 * the S reductions (`site_counter`: one zero-initialised unsigned the kernel re-arms, like `head_counter`).                  */
int alignq_site_reduce_loss_multi_head(int S, void* const* ws, float* const* D, const float* const* alterD,
                                       const float* const* gamma, float* const* scal, const int64_t* F, int B, int dim, float mu,
                                       float rho, const float* feat, const float* W, const float* bias, const int64_t* target, int HB,
                                       int HW, int C, int K, float* pooled, float* logits, float* probs, float* loss, float* ce_mean,
                                       unsigned* head_counter, const float* scal_all, int n_sites, float* trans_total,
                                       unsigned* site_counter, void* stream);

/* ---- Conv2d_Q's convolution on the matrix cores (model/quantization.py:149-154: F.conv2d(input, weight_q, bias, stride,
 * padding, dilation, groups)) for the ResNet-20/56 body: 3x3, stride 1, padding 1, groups 1, no bias, C_in == C_out == C,
 * channels-last fp32 tensors x [B,H,W,C], wt [C,3,3,C] (torch.channels_last storage of a [C,C,3,3] weight), y [B,H,W,C].
 * wt MUST be weight_quantize_fn's output for w_bit <= 8, i.e. values b / (2^w_bit - 1) with integer |b| <= 2^w_bit - 1
 * (both trees' formulas give that): the kernel multiplies the exact integers b on the bf16 matrix pipe with an exact
 * three-way bf16 split of the fp32 activations and divides by 2^w_bit - 1 at the end, so products are exact and only fp32
 * accumulation error remains.  Supported (C, W): (16, 32), (32, 16), (64, 8), H a multiple or divisor of the tile rows;
 * anything else returns ALIGNQ_EUNSUPPORTED and the caller keeps MIOpen.
 * dgrad = 0: y = conv(x, wt);  dgrad = 1: x is dy and y receives dx (the same kernel on the flipped, transposed filter).
 * add != NULL: a [B,H,W,C] tensor added to the result in the epilogue (the identity shortcut's gradient joining dx).
 * bn_part != NULL (forward only): per-workgroup per-channel {sum y, sum y^2} as floats [C][alignq_conv3x3_bn_parts][2] for the
 * batch-norm that follows (alignq_site_partials_bn, conv_parts).                                                          */
int alignq_conv3x3_bn_parts(int B, int H, int W, int C);   /* workgroups of the forward launch (0: unsupported shape) */
int alignq_conv3x3_nhwc(const float* x, const float* wt, float* y, int B, int H, int W, int C, int w_bit, int dgrad,
                        const float* add, float* bn_part, const void* x_bins, int x_bin_bytes, int a_bit, void* stream);
/* x_bins (N2, forward only, may be NULL; x may then be NULL): the activation operand as the int8 / int16 (x_bin_bytes 1 / 2)
 * level indices of an a_bit-bit ADMM-formula activation quantiser (alignq_site_partials_bn bins_out: value = idx / (2^a_bit - 1),
 * idx already clamped by the fused ReLU): the index is exact in TWO bf16 terms (two instead of three MFMAs per k step, 1-2 B
 * instead of 4 B read per activation) and the integer sum is divided by (2^w_bit - 1)(2^a_bit - 1) once.                   */

/* Forward of the body's transition convolutions (stride 2: 3x3 padding 1, and the 1x1 shortcut), C_in != C_out:
 * (CIN, COUT, W_in) in {(16, 32, 32), (32, 64, 16)}; x [B,H_in,W_in,CIN], wt [COUT,KS,KS,CIN], y [B,H_in/2,W_in/2,COUT], all
 * channels-last; same exact-product scheme and optional bn_part ([COUT][alignq_conv_gen_bn_parts][2]) as alignq_conv3x3_nhwc.
 * Gradients: alignq_conv_gen_nhwc_wgrad, alignq_conv_gen_nhwc_dgrad.                                                     */
int alignq_conv_gen_bn_parts(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride);
int alignq_conv_gen_nhwc_fwd(const float* x, const float* wt, float* y, int B, int H_in, int W_in, int CIN, int COUT, int KS,
                             int stride, int w_bit, float* bn_part, void* stream);

/* data gradient of the transition convolutions; bn_* as in alignq_conv3x3_nhwc_bwd (lazy batch-norm form of dy, incl. the
 * in-kernel totals form bn_ktot == NULL + bn_dx_part; the filter-gradient entry points take bn_dx_part too but leave the
 * parameter gradients to the data-gradient kernel — the stem has none, so its filter gradient writes them); add (or NULL):
 * a tensor of dx's shape added in the epilogue (the gradient the block's other branch sends to the same input) */
int alignq_conv_gen_nhwc_dgrad(const float* dy, const float* wt, float* dx, int B, int H_in, int W_in, int CIN, int COUT, int KS,
                               int stride, int w_bit, const float* add, const float* bn_z, const float* bn_ab,
                               const float* bn_save, const float* bn_ktot, const float* bn_dx_part, float* bn_dgamma,
                               float* bn_dbeta, void* stream);

/* Both convolutions of a transition block (the 3x3 `conv0` and the 1x1 `skip_conv`, stride 2, reading the SAME input;
 * reference: model/resnet.py PreActBlock_conv_Q.forward, `self.skip_conv(x)` and `self.conv0(x)`) in ONE launch each way: a launch
 * boundary costs more than the 1x1 convolution.  Shapes of alignq_conv_gen_nhwc_fwd; wt3 [COUT,3,3,CIN], wt1 [COUT,1,1,CIN], one
 * w_bit.  Forward: workgroup roles (3x3 | 1x1).  Backward: roles (data gradient of both convolutions, the 1x1 one entering as a
 * tenth tap | 3x3 filter-gradient slabs | 1x1 filter-gradient slabs); ws3 / ws1 = alignq_conv_gen_wgrad_ws_bytes(.., 3 / 1), the
 * slabs are left for alignq_conv3x3_wgrad_reduce_multi (*n_slabs3 / *n_slabs1); bn3_* / bn1_*: lazy batch-norm form of dy3 / dy1
 * as in alignq_conv_gen_nhwc_dgrad; add (or NULL) joins dx.  Results equal the separate launches' (y, bn_part, dW bit for bit;
 * dx to fp32 accumulation order).                                                                                          */
int alignq_transition_nhwc_fwd(const float* x, const float* wt3, const float* wt1, float* y3, float* y1, int B, int H_in,
                               int W_in, int CIN, int COUT, int w_bit, float* bn_part3, float* bn_part1, void* stream);
int alignq_transition_nhwc_bwd(const float* x, const float* dy3, const float* dy1, const float* wt3, const float* wt1,
                               float* dx, void* ws3, void* ws1, int B, int H_in, int W_in, int CIN, int COUT, int w_bit,
                               int* n_slabs3, int* n_slabs1, const float* add,
                               const float* bn3_z, const float* bn3_ab, const float* bn3_save, const float* bn3_ktot,
                               const float* bn3_dx_part, float* bn3_dgamma, float* bn3_dbeta,
                               const float* bn1_z, const float* bn1_ab, const float* bn1_save, const float* bn1_ktot,
                               const float* bn1_dx_part, float* bn1_dgamma, float* bn1_dbeta, void* stream);

/* The stem (3 -> 16 channels, 3x3, stride 1, padding 1, width 32; x [B,H,32,3], wt [16,3,3,3], y [B,H,32,16], channels-last):
 * forward with the optional batch-norm partials, and its filter gradient (ws: 256 * 432 floats); K = 27 is one MFMA k step
 * over an im2col image of the tile.                                                                                      */
int alignq_conv_stem_bn_parts(int B, int H, int W);
int alignq_conv_stem_nhwc_fwd(const float* x, const float* wt, float* y, int B, int H, int W, int w_bit, float* bn_part,
                              void* stream);
int alignq_conv_stem_nhwc_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H, int W, int* n_slabs_out,
                                const float* bn_z, const float* bn_ab, const float* bn_save, const float* bn_ktot,
                                const float* bn_dx_part, float* bn_dgamma, float* bn_dbeta, void* stream);

/* Filter gradient of the same convolution, dW [C,3,3,C] (channels-last weight storage) from x and dy: plain fp32 on the f32
 * MFMAs (products and accumulation bit-for-bit an fmaf chain), per-pixel-range partial sums in ws
 * (alignq_conv3x3_wgrad_ws_bytes(C)) reduced in fixed order by a second launch: deterministic, no zero-fill, no atomics.  */
size_t alignq_conv3x3_wgrad_ws_bytes(int C);
int alignq_conv3x3_nhwc_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H, int W, int C,
                              int* n_slabs_out, const void* x_bins, int x_bin_bytes, int a_bit, void* stream);
/* (x_bins / x_bin_bytes / a_bit as in alignq_conv3x3_nhwc: the x operand of the filter gradient from its level indices)       */
/* n_slabs_out (HOST pointer) != NULL defers the reduction: only the partial sums are launched, *n_slabs_out receives the
 * slab count, and ONE alignq_conv3x3_wgrad_reduce_multi launch later finishes T filters (HOST arrays ws / dw / n_slabs / C). */
int alignq_conv3x3_wgrad_reduce_multi(int T, const void* const* ws, float* const* dw, const int* n_slabs, const int* n_elem,
                                      void* stream);       /* n_elem[t] = elements of filter t (9*C*C, or KS*KS*CIN*COUT) */
/* filter gradient of the transition convolutions (shapes of alignq_conv_gen_nhwc_fwd), same deferred-reduction contract */
size_t alignq_conv_gen_wgrad_ws_bytes(int CIN, int COUT, int KS);
int alignq_conv_gen_nhwc_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H_in, int W_in, int CIN,
                               int COUT, int KS, int stride, int* n_slabs_out, const float* bn_z, const float* bn_ab,
                               const float* bn_save, const float* bn_ktot, const float* bn_dx_part, void* stream);

/* ---- Conv2d_Q's convolution at the ResNet-50 / Office-31 shapes (BASELINE config 5) as an exact-product GEMM -----------------
 * Reference: cdf_alignment_admm/dann_office/model/quantization.py:164-181 (Conv2d_Q.forward: F.conv2d(input, weight_q, bias,
 * stride, padding, dilation, groups)), built by model/resnet.py:31-41 (conv3x3 / conv1x1) and called from Bottleneck.forward
 * :131-156 and the downsample branch :122-126; the gradients are what autograd derives from that call.
 * Shapes: x [B, H_in, W_in, CIN] channels-last fp32; wt = weight_q [COUT, KS, KS, CIN] (channels-last storage of [COUT,CIN,KS,KS]),
 * w_bit <= 8; y [B, H_out, W_out, COUT] with H_out = (H_in - 1) / stride + 1; CIN, COUT multiples of 64; KS = 1 (padding 0) or
 * KS = 3 (padding 1), stride 1 or 2; no bias, no groups, no dilation.  alignq_qconv_supported says whether a shape is taken.
 * Arithmetic: integer filter bins rint(wt * (2^w_bit - 1)) times either the three exact bf16 terms of a general fp32 operand
 * (x_levels == 0) or - x_levels = n_a > 0: x is a quantiser output idx / n_a with integer |idx| <= 2048, e.g. relu(act_q(.)) of
 * activation_quantize_fn - the index itself as one f16 term; fp32 accumulation on v_mfma_f32_16x16x32_{bf16,f16}; see
 * csrc/qgemm_kernels.hip.  A wrong x_levels (a tensor that is not on that grid) silently rounds the operand: callers pass it only
 * for tensors they produced with the quantiser.
 * groups / bn_part (forward, optional): per-row-tile per-channel {sum y, sum y^2} in double,
 * [groups][alignq_qconv_bn_parts][COUT][2], the layout alignq_bnq_* finalise (the batch is `groups` equal slices with separate
 * batch-norm statistics; a tile never straddles two slices).                                                                    */
int alignq_qconv_supported(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride);
int alignq_qconv_bn_parts(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride, int groups, float x_levels);
/* The filter operand of the forward and the data gradient: the integer bins b = rint(W_q * (2^w_bit - 1)) of T quantised filters
 * (wt[i]: n[i] floats, n[i] % 4 == 0, any layout - the bins keep it) as 16-bit patterns, bf16 in bins_bf16[i] and f16 in
 * bins_f16[i] (n[i] halfwords each; both exact for |b| <= 255).  One launch per 64 filters; HOST arrays of DEVICE pointers.       */
int alignq_qconv_pack_weights(int T, const float* const* wt, const int64_t* n, int w_bit, void* const* bins_bf16, void* const* bins_f16,
                              void* stream);
/* w_bins: the filter's bins [COUT, KS, KS, CIN] from alignq_qconv_pack_weights - the bf16 patterns when x_levels == 0, the f16
 * patterns when x_levels > 0.                                                                                                    */
int alignq_qconv_fwd(const void* x, const void* w_bins, float* y, int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride,
                     int w_bit, float x_levels, int x_bin_bytes, int groups, double* bn_part, void* stream);
/* x_bin_bytes = 2 (N2; needs x_levels > 0): x holds the level tensor's int16 indices (alignq_bnq_fwd_parts bins_out)
 * instead of fp32 values; 0: fp32.                                                                                              */
/* data gradient dx [B, H_in, W_in, CIN] from dy [B, H_out, W_out, COUT] and the filter's bf16 bins (every element of dx is written;
 * the 3x3 stride-2 form runs per parity class of the input pixel and needs even H_in and W_in, else ALIGNQ_EUNSUPPORTED)          */
/* The Office ResNet-50's stem, Conv2d_Q(3, 64, kernel_size = 7, stride = 2, padding = 3, bias = False) (dann_office/model/resnet.py:193-195):
 * x [B, H_in, W_in, 3] channels-last fp32 (the image: no data gradient), w_bins = the filter's bf16 bins [64][7][7][3] from
 * alignq_qconv_pack_weights, y [B, H_out, W_out, 64]; three exact bf16 terms of x, fp32 accumulation (as alignq_qconv_fwd).
 * bn_part (or NULL): [groups][alignq_qconv_stem7_bn_parts][64][2] doubles {sum y, sum y^2} for alignq_bnq_fwd_parts.              */
int alignq_qconv_stem7_bn_parts(int B, int H_in, int W_in, int groups);
int alignq_qconv_stem7_fwd(const float* x, const void* w_bins, float* y, int B, int H_in, int W_in, int w_bit, int groups,
                           double* bn_part, void* stream);
/* its filter gradient dw [64][7][7][3] (the layout of the filter) from the image and dy [B, H_out, W_out, 64]: six leading bf16 term
 * pairs, deterministic split slabs in ws (alignq_qconv_stem7_wgrad_ws_bytes), summed in slab order by this call (dw != NULL,
 * n_slabs_out == NULL) or later by alignq_conv3x3_wgrad_reduce_multi (n_slabs_out receives the slab count; n_elem = 64 * 147). */
size_t alignq_qconv_stem7_wgrad_ws_bytes(int B, int H_in, int W_in);
int alignq_qconv_stem7_wgrad(const float* x, const float* dy, float* dw, void* ws, int B, int H_in, int W_in, int* n_slabs_out,
                             void* stream);
/* ws (or NULL; alignq_qconv_dgrad_ws_bytes, 0 = none needed): scratch for split-K - layers with few row tiles and a long
 * contraction (layer3 / layer4 at B = 56) run 2 to 4 workgroups per tile over disjoint k ranges and a closing pass adds their raw
 * sums in split order (deterministic); without ws every tile is one workgroup.                                                     */
size_t alignq_qconv_dgrad_ws_bytes(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride);
int alignq_qconv_dgrad(const float* dy, const void* w_bins, float* dx, int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride,
                       int w_bit, void* ws, void* stream);
/* filter gradient dW [COUT, KS, KS, CIN] (the layout of wt): deterministic split-K slabs in ws (alignq_qconv_wgrad_ws_bytes),
 * summed in slab order - by this call (dw != NULL, n_slabs_out == NULL) or later by alignq_conv3x3_wgrad_reduce_multi
 * (n_slabs_out receives the slab count; n_elem = COUT * KS * KS * CIN).  x_levels as in alignq_qconv_fwd.                       */
size_t alignq_qconv_wgrad_ws_bytes(int B, int H_in, int W_in, int CIN, int COUT, int KS, int stride);
int alignq_qconv_wgrad(const void* x, const float* dy, float* dw, void* ws, int B, int H_in, int W_in, int CIN, int COUT, int KS,
                       int stride, float x_levels, int x_bin_bytes, int* n_slabs_out, void* stream);

/* Data gradient AND filter-gradient partial sums of one convolution in a single launch (workgroup roles by block index; the
 * two are independent and fill the chip together).  The slabs left in ws are finished by alignq_conv3x3_wgrad_reduce_multi.
 * Lazy batch-norm form of dy (bn_z != NULL): dy is the gradient g w.r.t. the folded batch-norm's OUTPUT and the kernel forms
 * dz = a*(g - k0 - zhat*k1) on load; the totals k0, k1 come from bn_ktot (alignq_bn_bwd_totals) or, with bn_ktot == NULL, are
 * reduced by every workgroup from bn_dx_part = alignq_site_bwd_apply_bn's per-tile sums (no launch in between); one
 * workgroup then also writes the batch-norm parameter gradients bn_dgamma / bn_dbeta ([C] each, may be NULL). */
int alignq_conv3x3_nhwc_bwd(const float* x, const float* dy, const float* wt, float* dx, void* ws, int B, int H, int W,
                            int C, int w_bit, int* n_slabs_out, const float* add, const float* bn_z, const float* bn_ab,
                            const float* bn_save, const float* bn_ktot, const float* bn_dx_part, float* bn_dgamma,
                            float* bn_dbeta, const void* x_bins, int x_bin_bytes, int a_bit, void* stream);
/* The same launch with a FILLER role: workgroups behind the filter-gradient role finish the slab reduction of up to four EARLIER
 * convolutions (fill_ws[i]: their slab workspace, fill_n_slabs[i] / fill_n_elem[i] as alignq_conv3x3_wgrad_reduce_multi takes
 * them, fill_dw[i]: the finished gradient; HOST arrays of n_fill <= 4 entries).  Independent work inside a launch that is on the
 * backward's critical path anyway: the closing alignq_conv3x3_wgrad_reduce_multi is left with the last convolutions only.
 * Bit-identical to that reduction.                                                                                          */
int alignq_conv3x3_nhwc_bwd_fill(const float* x, const float* dy, const float* wt, float* dx, void* ws, int B, int H, int W,
                            int C, int w_bit, int* n_slabs_out, const float* add, const float* bn_z, const float* bn_ab,
                            const float* bn_save, const float* bn_ktot, const float* bn_dx_part, float* bn_dgamma,
                            float* bn_dbeta, const void* x_bins, int x_bin_bytes, int a_bit, int n_fill,
                                 const void* const* fill_ws, float* const* fill_dw, const int* fill_n_slabs,
                                 const int* fill_n_elem, void* stream);
/* bn_z != NULL: `dy` is not the convolution output's gradient but g, the gradient w.r.t. the OUTPUT of the training-mode
 * batch-norm that follows the convolution (what alignq_site_bwd_apply_bn writes); both roles form
 * dy = a[c] * (g - k0[c] - (z - mean[c]) * invstd[c] * k1[c]) on load from bn_z (the convolution's forward output), bn_ab,
 * bn_save and bn_ktot = {k0[C], k1[C]} (alignq_bn_bwd_totals), which replaces the elementwise pass of alignq_bn_bwd_apply. */
int alignq_bn_bwd_totals(const float* dx_part, int B, int C, int HW, float* ktot, float* dgamma, float* dbeta, void* stream);

/* ---- classifier head of the training harness fused with its loss (channels-last features feat [B,HW,C]):
 * pooled = mean over HW; logits = pooled W^T + bias (W [K,C]); loss[b] = cross-entropy(logits[b], target[b]) per sample (the
 * caller averages); probs = softmax(logits) kept for the backward.  Backward for the MEAN of loss with upstream scalar *g:
 * dfeat, dW, dbias (dbias may be NULL).  C <= 256, K <= 64.  Reference: model/resnet.py:127-129 + main.py's criterion.   */
/* ce_mean (or NULL): additionally *ce_mean = mean_b loss[b], formed in index order by the workgroup that finishes last
 * (`counter`: one zero-initialised unsigned the kernel re-arms); site_scal / n_sites / trans_total (or NULL / 0 / NULL, need
 * ce_mean): *trans_total = sum_i site_scal[4 i], the total of the sites' transition losses (the `scal` rows of
 * alignq_site_reduce_loss_multi, consecutive) — neither reduction gets a launch of its own in a training step.           */
int alignq_head_ce_fwd(const float* feat, const float* W, const float* bias, const int64_t* target, int B, int HW, int C, int K,
                       float* pooled, float* logits, float* probs, float* loss, float* ce_mean, unsigned* counter,
                       const float* site_scal, int n_sites, float* trans_total, void* stream);
int alignq_head_ce_bwd(const float* g, const float* probs, const int64_t* target, const float* pooled, const float* W, int B,
                       int HW, int C, int K, float* dfeat, float* dW, float* dbias, void* stream);
/* alignq_head_ce_bwd (head batch HB) and alignq_site_prep_fused_multi (S sites) as two roles of ONE launch: the two backward
 * roots of a training step (cross-entropy and the summed transition loss) start together.                                 */
int alignq_head_ce_bwd_site_prep(const float* g_ce, const float* probs, const int64_t* target, const float* pooled,
                                 const float* W, int HB, int HW, int C, int K, float* dfeat, float* dW, float* dbias, int S,
                                 const float* const* D, const float* const* alterD, const float* const* gamma,
                                 const float* const* scal, const float* dD_scale, const int64_t* F, int B, int dim, float mu,
                                 float* const* S_out, float* const* dalterD, float* const* dgamma, void* stream);

/* ---- data-parallel flat bucket (ONE mean all-reduce per step over gradients + stacked D matrices):
 * gather T dense device tensors (HOST array of pointers, element counts n[T]) into `flat` back to back (unpack = 0) or
 * scatter them back (unpack = 1); one launch per 128 tensors instead of one copy kernel per tensor.                      */
int alignq_bucket_copy_multi(int T, float* const* tensors, const int64_t* n, float* flat, int unpack, void* stream);
/* Overlapping the buckets' all-reduces with a CAPTURED backward (round 6; SURVEY.md §5 "bucket + overlap with backward", §8e; the
 * reference has no distributed code, cdf_alignment_admm/dann_office/main.py:28).  A collective cannot be captured into the step's HIP
 * graph and torch refuses external events on ROCm, so the order "bucket i packed -> all-reduce of bucket i"
 * goes through a flag in device memory: the graph holds  alignq_dp_counter_bump(counter)  once at its start and, behind the pack of
 * bucket i,  alignq_dp_flag_publish(flag_i, counter)  (flag_i = the replay's number, system-scope release); after launching the
 * graph the host enqueues  alignq_dp_stream_wait_ge(comm_stream, flag_i, replay number)  (hipStreamWaitValue32, >=) followed by the
 * eager all-reduce on the communication stream.  counter / flag: 4-byte aligned device words (any device allocation).  The wait
 * returns ALIGNQ_EUNSUPPORTED where the device has no stream memory operations (hipDeviceAttributeCanUseStreamWaitValue).        */
int alignq_dp_counter_bump(uint32_t* counter, void* stream);
int alignq_dp_flag_publish(uint32_t* flag, const uint32_t* counter, void* stream);
int alignq_dp_stream_wait_ge(void* stream, uint32_t* flag, uint32_t value);

/* ---- batch-norm (and the ReLU that follows) folded into the ADMM site (caller:
 * out, loss = act_q(bn(conv(x))); out = relu(out), cdf_alignment_admm/resnet-56-cifar-10/model/resnet.py:87-94) ----
 * Training-mode nn.BatchNorm2d semantics.  z = conv output [B,C,HW] (HW % 64 == 0, 64 < B <= 128 for the folded site
 * kernels).
 * alignq_bn_partial_stats: per-(channel, batch split) sums of z and z^2 into ws (alignq_bn_ws_bytes(C)); one launch.
 * alignq_site_partials_bn: as alignq_site_partials, but reads z, finalises the batch statistics of its tile's channel
 *   from bn_part in-kernel (mean, invstd, a = gamma*invstd, b = beta - mean*a) and applies x = a*z + b on load: the
 *   normalised activation is never materialised.  OUTPUTS besides xq/stats/ws: ab = {a[C], b[C]}, save = {mean[C],
 *   invstd[C]} (for the backward), running_mean / running_var (momentum, unbiased variance) and *num_batches_tracked
 *   (+1) — any of the last three may be NULL.  residual != NULL adds a [B,F] tensor to x_q (the block's shortcut,
 *   `out += shortcut`, resnet.py:95-96), relu != 0 then stores relu(.) — the stored tensor is what the next layer reads.
 * alignq_bn_stats: the stand-alone form (statistics + finalisation, two launches) producing the same ab / save.
 * Backward: alignq_site_prep_fused (first launch of alignq_site_bwd_fused alone), then alignq_site_bwd_apply_bn which
 *   writes dx (gradient w.r.t. the BN output; y_relu = the forward's output when relu was fused, else NULL; dresidual
 *   != NULL additionally receives the ReLU-masked upstream gradient, i.e. the gradient of `residual`) and per-tile
 *   sums dx_part [F/64][2] = {sum dx, sum dx*zhat}, then alignq_bn_bwd_apply: dz, dgamma, dbeta.                    */
size_t alignq_bn_ws_bytes(int C);
int alignq_bn_partial_stats(const float* z, int B, int C, int HW, void* ws, void* stream);
int alignq_bn_stats(const float* z, int B, int C, int HW, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float momentum, float eps, float* ab, float* save,
                    void* ws, void* stream);
int alignq_site_partials_bn(const float* z, const void* bn_part, const float* bn_gamma, const float* bn_beta,
                            float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                            float bn_eps, float* ab, float* save, int C, int HW, int B, int64_t F, int k, float act_range,
                            float eps, int relu, const float* residual, int nhwc, int conv_parts, float* xq, void* bins_out,
                            float* stats, void* ws, void* stream);
/* The same launch with a FILLER role (one-tile launches that leave CUs idle; alignq_site_fill_slots(B, F) says how many items
 * this shape takes, 0 = none): workgroups behind the site's own tiles finish the slab reduction + ADMM loss of n_fill EARLIER
 * sites of the step (arguments per item as alignq_site_reduce_loss_multi takes them; all items share this site's B and
 * fill_dim / fill_mu / fill_rho).  Nothing reads a site's D or loss before the end of the forward; same code, same workgroup
 * partition, same bits as alignq_site_reduce_loss[_multi].                                                              */
int alignq_site_fill_slots(int B, int64_t F);
/* TWO sites of one shape in ONE launch (round 6): the sites behind a transition block's two convolutions - `out = act_q0(bn0(conv0(x)))`
 * and `shortcut = act_skip_q(skip_bn(skip_conv(x)))`, cdf_alignment_admm/resnet-56-cifar-10/model/resnet.py:81-90 - do not depend on each
 * other and each is a 128-workgroup launch that leaves half of the chip idle; together they are one node of the step's chain instead of
 * two.  Each site is described by alignq_site_partials_bn's arguments (same meaning); both must share B, F, k, act_range, eps and have
 * their own buffers.  Same code per workgroup as two alignq_site_partials_bn launches: bit-identical outputs.  No filler role.
 * ALIGNQ_EUNSUPPORTED where a single site's launch already fills the chip (more than 128 tiles) or loops over tiles: launch them one
 * after the other then.                                                                                                          */
typedef struct alignq_site_bn_args {
  const float* z; const void* bn_part; const float* bn_gamma; const float* bn_beta; float* running_mean; float* running_var;
  int64_t* num_batches_tracked; float momentum, bn_eps; float* ab; float* save; int C, HW, B; int64_t F; int k; float act_range, eps;
  int relu; const float* residual; int nhwc, conv_parts; float* xq; void* bins_out; float* stats; void* ws;
} alignq_site_bn_args;
int alignq_site_partials_bn_twin(const alignq_site_bn_args* a, const alignq_site_bn_args* b, void* stream);
/* ... and their backward (alignq_site_bwd_apply_bn's arguments per site; both need the folded batch-norm): the two launches of 256
 * workgroups of 256 threads each leave half of the chip's workgroup slots free.  ALIGNQ_EUNSUPPORTED for other shapes.        */
typedef struct alignq_site_bwd_bn_args {
  const float* g; const float* S; const float* z; const float* ab; const float* save; int C, HW, nhwc; const float* y_relu;
  const void* y_bins; int y_bin_bytes; float* dresidual; const float* stats; int B; int64_t F; float act_range, eps; float* dx;
  float* dx_part;
} alignq_site_bwd_bn_args;
int alignq_site_bwd_apply_bn_twin(const alignq_site_bwd_bn_args* a, const alignq_site_bwd_bn_args* b, void* stream);
int alignq_site_partials_bn_fill(const float* z, const void* bn_part, const float* bn_gamma, const float* bn_beta,
                                 float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum,
                                 float bn_eps, float* ab, float* save, int C, int HW, int B, int64_t F, int k,
                                 float act_range, float eps, int relu, const float* residual, int nhwc, int conv_parts,
                                 float* xq, void* bins_out, float* stats, void* ws, int n_fill, void* const* fill_ws,
                                 float* const* fill_D, const float* const* fill_alterD, const float* const* fill_gamma,
                                 float* const* fill_scal, const int64_t* fill_F, int fill_dim, float fill_mu, float fill_rho,
                                 void* stream);
/* bins_out (N2, optional; residual must be NULL, F % 4 == 0): the level index of the STORED value (idx, clamped at 0 when the
 * ReLU is fused) in alignq_bin_bytes(k, act_range, ALIGNQ_FORMULA_ADMM) bytes per element, same [B,F] order as xq; xq may then
 * be NULL: consumers that understand the index (alignq_conv3x3_nhwc x_bins, alignq_site_bwd_apply_bn y_bins) need no fp32 copy. */
size_t alignq_site_bn_part_bytes(int64_t F, int nhwc);
int alignq_site_prep_fused(const float* D, const float* alterD, const float* gamma, int dim, const float* scal, float mu,
                           const float* dD_scale, int B, int64_t F, float* S, float* dalterD, float* dgamma,
                           void* stream);
int alignq_site_bwd_apply_bn(const float* g, const float* S, const float* z, const float* ab, const float* save, int C,
                             int HW, int nhwc, const float* y_relu, const void* y_bins, int y_bin_bytes, float* dresidual,
                             const float* stats, int B, int64_t F, float act_range, float eps, float* dx, float* dx_part,
                             void* stream);     /* y_bins (N2): the ReLU mask from the forward's level index (idx > 0) instead of y_relu */
/* alignq_site_bwd_apply_bn with a FILLER role (the narrow sites' launches, F <= 8192, fill half the chip; alignq_site_bwd_fill_slots
 * says how many items a shape takes, 0 = none): filter-gradient slab reductions of convolutions whose backward already ran, items
 * as in alignq_conv3x3_nhwc_bwd_fill.  Bit-identical to alignq_conv3x3_wgrad_reduce_multi.                                  */
int alignq_site_bwd_fill_slots(int B, int64_t F);
int alignq_site_bwd_apply_bn_fill(const float* g, const float* S, const float* z, const float* ab, const float* save, int C,
                                  int HW, int nhwc, const float* y_relu, const void* y_bins, int y_bin_bytes,
                                  float* dresidual, const float* stats, int B, int64_t F, float act_range, float eps,
                                  float* dx, float* dx_part, int n_fill, const void* const* fill_ws, float* const* fill_dw,
                                  const int* fill_n_slabs, const int* fill_n_elem, void* stream);
int alignq_bn_bwd_apply(const float* dx, const float* z, const float* ab, const float* save, const float* dx_part, int B,
                        int C, int HW, int nhwc, float* dz, float* dgamma, float* dbeta, void* stream);
/* Channels-last (torch.channels_last, memory [B,H,W,C]) form of the fold, nhwc = 1 above: channel = f mod C with C a power
 * of two in [4,256], F % 64 == 0.  alignq_bn_partial_stats_nhwc writes 64 {sum, sum of squares} partials per channel into
 * ws (alignq_bn_nhwc_ws_bytes(C)); alignq_site_partials_bn(bn_part = ws, nhwc = 1) finalises its tile's channels from
 * them exactly as in the NCHW form.  The site backward leaves per-tile per-channel sums in dx_part
 * (alignq_site_bn_part_bytes(F, 1)) and alignq_bn_bwd_apply(nhwc = 1) reduces them itself.                               */
/* conv_parts > 0 (channels-last only): bn_part is not alignq_bn_partial_stats_nhwc's buffer but the FLOAT partials
 * [C][conv_parts][2] that alignq_conv3x3_nhwc(bn_part = ...) left while it produced z (conv_parts =
 * alignq_conv3x3_bn_parts(B,H,W,C)): the batch-norm then needs no statistics pass of its own over z.                      */
size_t alignq_bn_nhwc_ws_bytes(int C);
int alignq_bn_partial_stats_nhwc(const float* z, int B, int C, int HW, void* ws, void* stream);

/* ---- batch-norm folded into the PLAIN quantiser (+ ReLU), channels-last, any batch (SURVEY.md §8f-N1 on configuration 5;
 * caller: out = relu(act_q1(bn1(conv1(x)))) of the Office bottleneck, cdf_alignment_admm/dann_office/model/resnet.py:134-143,
 * stem :230-233; quantiser model/quantization.py:87-110) ----
 * z: the convolution's output viewed as [P, C], P = B*H*W pixels, channels fastest (torch.channels_last); C = 4 * 2^j <= 2048
 * (ALIGNQ_EUNSUPPORTED otherwise).  Training-mode nn.BatchNorm2d semantics (biased batch variance for the normalisation,
 * running statistics updated with momentum and the unbiased variance, *num_batches_tracked += 1; any of the three NULL).
 * alignq_bnq_fwd (3 launches): per-channel statistics of z -> ab = {a[C], b[C]} (a = gamma*invstd, b = beta - mean*a),
 *   save = {mean[C], invstd[C]} -> y = [relu](quantise(a*z + b)) with alignq_act_quant_fwd's arithmetic; the normalised
 *   activation is never written (12 B/element instead of 20).
 * alignq_bnq_bwd (3 launches): dx = g * [y > 0] * dt/dx (y = the forward's output, required when relu), then the batch-norm
 *   backward dz = a*(dx - mean dx - zhat*mean(dx*zhat)), dgamma = sum dx*zhat, dbeta = sum dx (28 B/element instead of ~36).
 * ws: alignq_bnq_ws_bytes(C, groups).                                                                                             */
/* The batch-norm of that family WITHOUT a quantiser behind it in the same chain (C <= 2048):
 * alignq_bnq_stats: statistics + finalisation only -> ab, save (+ running statistics): what a consumer that applies
 *   x = a*z + b itself needs (alignq_site_partials_res_ab below: bn3 in front of the bottleneck's ADMM site, resnet.py:146-150);
 * alignq_bnq_affine: y = a*z + b (the downsample branch's batch-norm, resnet.py:122-126 / :151-152);
 * alignq_bnq_bwd_dx: batch-norm backward for a GIVEN dx (gradient w.r.t. the batch-norm output): dz, dgamma, dbeta; dz may
 *   alias dx.
 * alignq_site_partials_res_ab / alignq_site_bwd_apply_ab (B <= 32): alignq_site_partials_res / alignq_site_bwd_apply reading
 *   z and applying the affine on load (channel = f mod C); the backward's dx is w.r.t. x = a*z + b.                          */
/* groups (1..ALIGNQ_BNQ_MAX_GROUPS): z / g / y / dz are [groups][P][C] -- the batch slices of a merged multi-pass traversal
 * (source and target passes of the Office step, dann_office/main.py:296-330, stacked along the batch) -- each slice normalised
 * with ITS OWN batch statistics, as separate forward calls of the module would: ab and save are [groups][2][C], the running
 * statistics are updated slice after slice and *num_batches_tracked += groups; dgamma / dbeta are the sums over the slices.
 * One launch per kernel covers every slice.  P = pixels per slice.                                                          */
#define ALIGNQ_BNQ_MAX_GROUPS 8
int alignq_bnq_stats(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, float* ab, float* save,
                     void* ws, void* stream);
/* The same with the statistics pass replaced by partial sums the producing convolution left in its epilogue (alignq_qconv_fwd
 * bn_part: conv_part [groups][conv_parts][C][2] doubles, {sum z, sum z^2} per row tile): no read of z here.  conv_part == NULL: as
 * alignq_bnq_stats.                                                                                                          */
int alignq_bnq_stats_parts(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                           float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, float* ab, float* save,
                           void* ws, const double* conv_part, int conv_parts, void* stream);
int alignq_bnq_affine(const float* z, const float* ab, int64_t P, int C, int groups, float* y, void* stream);
int alignq_bnq_bwd_dx(const float* dx, const float* z, const float* ab, const float* save, int64_t P, int C, int groups,
                      float* dz, float* dgamma, float* dbeta, void* ws, void* stream);
int alignq_site_partials_res_ab(const float* z, const float* ab, int C, int B, int64_t F, int k, float act_range, float eps,
                                const float* residual, int relu, float* y, float* stats, void* ws, void* stream);
int alignq_site_bwd_apply_ab(const float* g, const float* S, const float* z, const float* ab, int C, const float* stats, int B,
                             int64_t F, float act_range, float eps, float* dx, void* stream);
/* alignq_site_bwd_apply_ab with the bottleneck's closing ReLU (resnet.py:153-154) folded in: g = gradient w.r.t.
 * relu(x_q + residual), y = that tensor (mask y > 0); dres (or NULL) receives the masked gradient (the residual's gradient).  */
int alignq_site_bwd_apply_ab_relu(const float* g, const float* y, const float* S, const float* z, const float* ab, int C,
                                  const float* stats, int B, int64_t F, float act_range, float eps, float* dx, float* dres,
                                  void* stream);
/* The B <= 32 site (alignq_site_partials_res_ab / alignq_site_reduce_loss / alignq_site_bwd_apply_ab_relu) for `groups` batch
 * slices stacked along the batch - the Office step's merged source + target pass (dann_office/main.py:296-330) - in ONE launch
 * per kernel: z / residual / y / g / dx / dres [groups][B][F]; ab [groups][2][C]; stats [groups][4][F]; D [groups][B][B];
 * scal [groups][4]; ws = groups regions of alignq_site_ws_bytes(B, F) each; S = groups regions of alignq_site_bwd_ws_bytes(B)
 * each (filled by alignq_site_prep_fused_multi with one entry per slice).  g == NULL: no upstream gradient (no mask, no dres).
 * g2 (round 4; or NULL): a second addend of the upstream gradient, same layout as g; the kernel reads g + g2 - the sum autograd
 * would form in an elementwise pass of its own where the block's output feeds the next block's convolution AND its shortcut. */
int alignq_site1_groups_fwd(const float* z, const float* ab, int C, int B, int64_t F, int groups, int k, float act_range,
                            float eps, const float* residual, int relu, float* y, float* stats, void* ws, void* stream);
int alignq_site1_groups_reduce_loss(void* ws, int B, int64_t F, int groups, float* D, const float* alterD, const float* gamma,
                                    int dim, float mu, float rho, float* scal, void* stream);
/* (round 4) the backward's preparation for all slices of ONE site in one launch: S regions as alignq_site_prep_fused_multi leaves
 * them, and dalterD / dgamma [dim,dim] = the SUM over the slices in slice order - what autograd accumulates when the reference
 * calls the module once per pass (dann_office/main.py:372,377), without the per-slice buffers and the launch that added them.   */
/* dD_scale (or NULL): the upstream gradient(s) of the slices' losses, slice g's at dD_scale[g * dD_scale_stride] (stride 0: one
 * device scalar for all slices - the gradient of their SUM; stride >= 1: the loss left as a vector over the slices).              */
int alignq_site1_groups_prep(const float* D, const float* alterD, const float* gamma, int dim, const float* scal, float mu,
                             const float* dD_scale, int dD_scale_stride, int B, int64_t F, int groups, float* S, float* dalterD,
                             float* dgamma, void* stream);
/* (round 5, ABI 20) the bottleneck tail with a ONE-BIT ReLU mask: alignq_site1_groups_fwd_m also leaves, per stored element of
 * y = relu(x_q + identity), its sign bit in relu_mask (alignq_site1_mask_bytes; [groups][ceil(F / 32)][32] words: bit `row` of word
 * (s, f) = y[row][32 s + f] > 0; rows >= B repeat row B - 1); alignq_site1_groups_bwd_bn_m takes that mask where alignq_site1_groups_bwd_bn takes y - the
 * backward of `out = self.relu(out)` (dann_office/model/resnet.py:154) then reads 0.14 B per element instead of 4.  Same results. */
size_t alignq_site1_mask_bytes(int B, int64_t F, int groups);
int alignq_site1_groups_fwd_m(const float* z, const float* ab, int C, int B, int64_t F, int groups, int k, float act_range,
                              float eps, const float* residual, int relu, float* y, float* stats, void* ws, void* relu_mask,
                              void* stream);
int alignq_site1_groups_bwd_bn_m(const float* g, const float* g2, const void* relu_mask, const float* S, const float* z,
                                 const float* ab, const float* save, int C, const float* stats, int B, int64_t F, int groups,
                                 float act_range, float eps, float* dz, float* dres, float* dgamma, float* dbeta, void* cols,
                                 void* ws_bn, void* stream);
/* (round 5, ABI 16) alignq_site1_groups_reduce_loss / _prep for T sites in ONE launch each (HOST arrays of device pointers, F[i]
 * per site; one B, groups, dim, mu, rho for all; dD_scale: ONE device scalar - the gradient of the sum of every site's and slice's
 * loss - or NULL = 1).  The Office iteration's 16 bottleneck tails (dann_office/model/resnet.py:145-154 called from main.py:372,377)
 * leave their reductions to the end of the forward and their preparations to the start of the backward; results are bit-identical
 * to the per-site calls.  Each site needs its OWN ws (alignq_site_ws_bytes(B, F) * groups) until the reduction has run.            */
int alignq_site1_groups_reduce_loss_multi(int T, void* const* ws, const int64_t* F, int B, int groups, float* const* D,
                                          const float* const* alterD, const float* const* gamma, int dim, float mu, float rho,
                                          float* const* scal, void* stream);
int alignq_site1_groups_prep_multi(int T, const float* const* D, const float* const* alterD, const float* const* gamma, int dim,
                                   const float* const* scal, float mu, const float* dD_scale, int B, const int64_t* F, int groups,
                                   float* const* S, float* const* dalterD, float* const* dgamma, void* stream);
int alignq_site1_groups_bwd(const float* g, const float* g2, const float* y, const float* S, const float* z, const float* ab, int C,
                            const float* stats, int B, int64_t F, int groups, float act_range, float eps, float* dx,
                            float* dres, void* stream);
/* alignq_site1_groups_bwd + alignq_bnq_bwd_dx as ONE entry (round 4): the backward of `relu(act_q3(bn3(z))[0] + identity)`
 * (dann_office/model/resnet.py:146-154) from the gradient of that output to dz, dgamma, dbeta, dres.  The site kernel also leaves,
 * per feature column, sum_b dx and sum_b dx * zhat (8 bytes per column, taken from the registers that hold the sub-tile); the
 * batch-norm backward's first pass over dx and z (8 B per ELEMENT) becomes a reduction over those columns.  save: [groups][2][C]
 * (mean, invstd) of alignq_bnq_stats; dz: [groups][B][F] (receives dx, then dz in place); cols: alignq_site1_cols_bytes(F, groups)
 * bytes of scratch; ws_bn: alignq_bnq_ws_bytes(C, groups).  Channels whose gamma is exactly 0 are summed from dx and z directly.  */
size_t alignq_site1_cols_bytes(int64_t F, int groups);
int alignq_site1_groups_bwd_bn(const float* g, const float* g2, const float* y, const float* S, const float* z, const float* ab,
                               const float* save, int C, const float* stats, int B, int64_t F, int groups, float act_range, float eps,
                               float* dz, float* dres, float* dgamma, float* dbeta, void* cols, void* ws_bn, void* stream);
size_t alignq_bnq_ws_bytes(int C, int groups);
/* ReLU mask as ONE BIT per element (round 4): alignq_bnq_fwd with mask != NULL also writes [y > 0] for every element
 * (alignq_bnq_mask_bytes(P, C, groups) bytes, 16-byte aligned; per group and per 64 consecutive channel quads four 64-bit words,
 * one per quad component: the forward's wave-wide compare results); alignq_bnq_bwd with mask != NULL takes the ReLU mask from
 * those bits and does not read y (y may be NULL): 20.25 instead of 28 B/element.  Same values either way.                   */
size_t alignq_bnq_mask_bytes(int64_t P, int C, int groups);
/* residual (round 4; or NULL): a tensor of z's shape added to the quantised value before the ReLU - `out = act_q1(bn1(.)); out +=
 * shortcut; out = F.relu(out)` of the CDF-only block (cdf_alignment/resnet-20-cifar-10/model/resnet.py:73-78) in the one apply
 * pass; dres (or NULL) then receives its gradient, the masked upstream gradient g * [y > 0].  Small single-group sites (C <= 64, a
 * few MB: configuration 1) skip the two finalisation launches: the apply kernels finalise the <= 32 partials per channel themselves.*/
int alignq_bnq_fwd(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, int k, float act_range,
                   int formula, int relu, const float* residual, float* ab, float* save, float* y, void* mask, void* ws,
                   void* stream);
/* alignq_bnq_fwd with the statistics pass replaced by a convolution's partial sums (see alignq_bnq_stats_parts), and - N2 on the
 * Office path - bins_out (or NULL): the quantised value's integer level index round(t n), ReLU-clamped when relu, as
 * int16 in z's layout (ADMM / Office formula, no residual, act_range * (2^k - 1) <= 32767); y may then be NULL: a consumer that
 * reads indices (alignq_qconv_fwd / _wgrad with x_bin_bytes = 2) needs no fp32 copy.  value = index / (2^k - 1) exactly as y.   */
int alignq_bnq_fwd_parts(const float* z, int64_t P, int C, int groups, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, float momentum, float bn_eps, int k, float act_range,
                         int formula, int relu, const float* residual, float* ab, float* save, float* y, void* mask, void* ws,
                         const double* conv_part, int conv_parts, void* bins_out, void* stream);
int alignq_bnq_bwd(const float* g, const float* z, const float* y, const void* mask, const float* ab, const float* save, int64_t P,
                   int C, int groups, float act_range, int relu, float* dz, float* dres, float* dgamma, float* dbeta, void* ws,
                   void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ALIGNQ_H */
