/* TEST INFRASTRUCTURE — plain-C scalar restatement of AlignQ's hot path (the kernel-level oracle).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (alignq_amd/) never links or calls it.
 *
 * Parity pinning: every function is checked against tensors captured from the reference's own
 * Python on torch-CPU (the .npz files in tests/golden, via tests/golden/gen_goldens.py) in
 * tests/test_oracle_c.py.  Elementwise fp32 results follow the reference's op ORDER exactly; the
 * only place they can differ from torch-CPU is the last ulp of erf/exp, because torch-CPU evaluates
 * those through Intel MKL VML (closed source).  This file instead implements the repo's own
 * ALIGNQ-NERF32 / ALIGNQ-EXP32 specification (alignq_amd/csrc/gen_erf32_coeffs.py),
 * which the HIP kernels implement bit-for-bit as well.  Consequence: integer
 * bins agree with torch everywhere except inside the "tie zone" |frac(t*n) - 1/2| < 1e-4, where a
 * 1-ulp erf difference may flip the rounding; tests assert exactness outside it and +-1 inside.
 *
 * Reductions (mean/std, Gram, norms) are accumulated in double here: they are compared with
 * tolerances (1e-5 / 1e-6), never bitwise.
 *
 * Reference citations are relative to /root/reference; "ADMM tree" =
 * cdf_alignment_admm/resnet-56-cifar-10, "CDF tree" = cdf_alignment/resnet-20-cifar-10,
 * "Office tree" = cdf_alignment_admm/dann_office.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -mfma).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/alignq_erf32_coeffs.h"

#define OQ_FORMULA_ADMM 0 /* ADMM/Office trees: map to [-1,1](*r) first, then round */
#define OQ_FORMULA_CDF 1  /* CDF-only tree: round c in [0,1], then map */

/* ------------------------------------------------------------------ ALIGNQ-EXP32 / ERF32 ---- */
static inline float pow2i(int e) {
  uint32_t u = (uint32_t)(e + 127) << 23;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

float oq_exp32_1(float x) {
  if (!(x >= -104.0f)) return (x != x) ? x : 0.0f;
  if (x > 88.7f) return INFINITY;
  float nf = rintf(x * ALIGNQ_LOG2E);
  int n = (int)nf;
  float r = fmaf(nf, -ALIGNQ_LN2_HI, x);
  r = fmaf(nf, -ALIGNQ_LN2_LO, r);
  float p = ALIGNQ_PE5;
  p = fmaf(p, r, ALIGNQ_PE4);
  p = fmaf(p, r, ALIGNQ_PE3);
  p = fmaf(p, r, ALIGNQ_PE2);
  p = fmaf(p, r, ALIGNQ_PE1);
  p = fmaf(p, r, ALIGNQ_PE0);
  float r2 = r * r;
  float e = 1.0f + fmaf(r2, p, r);
  int h = n >> 1;
  return e * pow2i(h) * pow2i(n - h);
}

/* ALIGNQ-NERF32 (round 3): nerf32(y) ~ erf(y/sqrt(2)) = 2*Phi(y) - 1, the function Normal(0,1).cdf needs.  ONE
 * evaluated branch: a table node per 1/8 of |y| (found by adding 2^20, whose ulp is 1/8), a degree-4 polynomial in the
 * exact difference to the node's centre.  Spec: alignq_amd/csrc/gen_erf32_coeffs.py; |error| <= 0.57 * 2^-24 over every
 * fp32 (tests/native/verify_nerf.c).  It replaces "divide by sqrt(2), then erf" of torch's Normal.cdf
 * (torch/distributions/normal.py; reference call sites model/quantization.py:50-51): the result is within 1 ulp(1) of
 * what torch computes, the same distance the round-1/2 erf32 had (MKL's erf is not restatable). */
static const float NERF_POLY[ALIGNQ_NERF_N][4] = ALIGNQ_NERF_POLY;
static const float NERF_CENTRE[ALIGNQ_NERF_N][4] = ALIGNQ_NERF_CENTRE;

float oq_nerf32_1(float y) {
  float a = fabsf(y);
  if (a != a) return y;                       /* NaN stays NaN (the device clamp is NaN-propagating) */
  a = a < ALIGNQ_NERF_YMAX ? a : ALIGNQ_NERF_YMAX;
  float u = a + ALIGNQ_NERF_MAGIC;            /* RN: u - 2^20 = a rounded to the nearest eighth, ties to even */
  uint32_t ub, mb;
  float magic = ALIGNQ_NERF_MAGIC;
  memcpy(&ub, &u, 4);
  memcpy(&mb, &magic, 4);
  const uint32_t k = ub - mb;                 /* 0 .. ALIGNQ_NERF_N-1 */
  const float* c = NERF_POLY[k];
  float d = a - NERF_CENTRE[k][1];            /* exact */
  float q = c[3];
  q = fmaf(q, d, c[2]);
  q = fmaf(q, d, c[1]);
  q = fmaf(q, d, c[0]);
  float res = fmaf(q, d, NERF_CENTRE[k][0]);
  return copysignf(res, y);
}

void oq_nerf32(const float* x, float* y, long n) {
  for (long i = 0; i < n; i++) y[i] = oq_nerf32_1(x[i]);
}
void oq_exp32(const float* x, float* y, long n) {
  for (long i = 0; i < n; i++) y[i] = oq_exp32_1(x[i]);
}

/* ------------------------------------------------------------------ R1/R2 elementwise ------- */
#define LOG_SQRT_2PI 0.91893853320467274178     /* math.log(math.sqrt(2*math.pi)) */

/* Normal(m,s).cdf: 0.5*(1+erf((v-m)*(1/s)/sqrt(2))) (torch/distributions/normal.py cdf; reference call sites
 * model/quantization.py:50-51), with erf(./sqrt 2) taken as ONE specified function, nerf32. */
static inline float gauss_cdf32(float v, float m, float rs) {
  return 0.5f * (1.0f + oq_nerf32_1((v - m) * rs));
}

/* uniform_quantize(k).forward on a value already transformed (model/quantization.py:23-31) */
static inline float round_bins(float t, int k, float* bin) {
  if (k == 32) { *bin = t; return t; }
  if (k == 1) { float s = (t > 0.0f) - (t < 0.0f); *bin = s; return s; }
  float n = (float)((1 << k) - 1);
  float b = rintf(t * n);   /* torch.round: half to even */
  *bin = b;
  return b / n;
}

/* R4 (elementwise part). ADMM tree model/quantization.py:109-110 + :53-56; CDF tree :97-98.
 * t_out (optional): the pre-round transform; bins (optional): integer bin index. */
void oq_act_quant_fwd(const float* x, float* xq, float* t_out, int32_t* bins, long n, int k, float r,
                      int formula) {
  for (long i = 0; i < n; i++) {
    float c = gauss_cdf32(x[i], 0.0f, 1.0f);
    float t, b, q;
    if (formula == OQ_FORMULA_ADMM) {
      t = (c * 2.0f - 1.0f) * r;
      q = round_bins(t, k, &b);
    } else {
      t = c;
      q = (round_bins(c, k, &b) * 2.0f - 1.0f) * r;
    }
    xq[i] = q;
    if (t_out) t_out[i] = t;
    if (bins) bins[i] = (int32_t)b;
  }
}

/* d t / d x for the activation transform = r * 2 * phi(x); autograd chain of the lines above with
 * the STE of model/quantization.py:34-36.  Same for both formulas. */
void oq_act_quant_bwd(const float* g, const float* x, float* dx, long n, float r) {
  const double c = 2.0 / sqrt(2.0 * M_PI);
  for (long i = 0; i < n; i++) {
    double xv = x[i];
    dx[i] = (float)((double)g[i] * (double)r * c * exp(-0.5 * xv * xv));
  }
}

/* ------------------------------------------------------------------ R3 weights -------------- */
/* torch.mean / torch.std (unbiased) over all elements: model/quantization.py:78 */
void oq_weight_stats(const float* w, long n, float* ms) {
  double s = 0;
  for (long i = 0; i < n; i++) s += w[i];
  double m = s / (double)n, v = 0;
  for (long i = 0; i < n; i++) { double d = w[i] - m; v += d * d; }
  ms[0] = (float)m;
  ms[1] = (float)sqrt(v / (double)(n - 1));
}

/* weight_quantize_fn.forward given (m,s): ADMM tree :78-80, CDF tree :70-72.
 * cdf_out = the tree's `weight_cdf` (ADMM: 2c-1, CDF: c), pdf_out = exp(log_prob)*2. */
void oq_weight_quant_fwd(const float* w, const float* ms, float* q, float* cdf_out, float* pdf_out,
                         int32_t* bins, long n, int k, int formula) {
  float m = ms[0], s = ms[1];
  float rs = 1.0f / s;
  float var2 = 2.0f * (s * s);
  float logs = (float)log((double)s);
  for (long i = 0; i < n; i++) {
    float c = gauss_cdf32(w[i], m, rs);
    float t, b, qq;
    if (formula == OQ_FORMULA_ADMM) {
      t = c * 2.0f - 1.0f;
      qq = round_bins(t, k, &b);
    } else {
      t = c;
      qq = round_bins(c, k, &b) * 2.0f - 1.0f;
    }
    q[i] = qq;
    if (cdf_out) cdf_out[i] = t;
    if (bins) bins[i] = (int32_t)b;
    if (pdf_out) {
      float d = w[i] - m;
      float lp = -(d * d) / var2 - logs - (float)LOG_SQRT_2PI;
      pdf_out[i] = oq_exp32_1(lp) * 2.0f;
    }
  }
}

/* Autograd of R3 through mean and std:
 *   dW_i = g_i P_i - mean_j(g_j P_j) - z_i/(N-1) * sum_j g_j P_j z_j,  P = 2*pdf_N(m,s), z=(w-m)/s */
void oq_weight_quant_bwd(const float* g, const float* w, const float* ms, float* dw, long n) {
  double m = ms[0], s = ms[1];
  double c = 2.0 / (s * sqrt(2.0 * M_PI));
  double s1 = 0, s2 = 0;
  for (long i = 0; i < n; i++) {
    double z = (w[i] - m) / s, P = c * exp(-0.5 * z * z);
    s1 += g[i] * P;
    s2 += g[i] * P * z;
  }
  for (long i = 0; i < n; i++) {
    double z = (w[i] - m) / s, P = c * exp(-0.5 * z * z);
    dw[i] = (float)(g[i] * P - s1 / (double)n - z / (double)(n - 1) * s2);
  }
}

/* ------------------------------------------------------------------ R5 corr ----------------- */
/* Column statistics over the batch (unbiased std), ADMM tree :135, Office :159 (+eps on std). */
static void col_stats(const float* x, int B, long F, double* mu, double* sd) {
  for (long f = 0; f < F; f++) {
    double s = 0;
    for (int b = 0; b < B; b++) s += x[(long)b * F + f];
    double m = s / B, v = 0;
    for (int b = 0; b < B; b++) { double d = x[(long)b * F + f] - m; v += d * d; }
    mu[f] = m;
    sd[f] = sqrt(v / (B - 1));
  }
}

/* G = Xh Xh^T / F with Xh = (x - mu)/(sd + eps) */
void oq_corr_fwd(const float* x, int B, long F, float eps, float* G) {
  double* mu = malloc(sizeof(double) * F), *sd = malloc(sizeof(double) * F);
  double* xh = malloc(sizeof(double) * B * F);
  col_stats(x, B, F, mu, sd);
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) xh[(long)b * F + f] = (x[(long)b * F + f] - mu[f]) / (sd[f] + eps);
  for (int i = 0; i < B; i++)
    for (int j = 0; j <= i; j++) {
      double a = 0;
      for (long f = 0; f < F; f++) a += xh[(long)i * F + f] * xh[(long)j * F + f];
      G[i * B + j] = G[j * B + i] = (float)(a / (double)F);
    }
  free(mu); free(sd); free(xh);
}

/* dx for upstream dG (any, not nec. symmetric), y is x (corr(x,x)):
 *   dXh = (dG + dG^T) Xh / F ;  dx = (dXh - mean_b dXh)/(sd+eps) - (x-mu)/((B-1) sd) * sum_b(dXh*Xh)/(sd+eps)
 * accumulate==1 adds into dx. `jac` (optional, per element) multiplies the result (chain rule through
 * t(x) when the corr input is the transformed activation). */
static void corr_bwd_core(const float* dG, const double* v, int B, long F, double eps, const double* jac,
                          float* dx, int accumulate, double sign) {
  double* mu = malloc(sizeof(double) * F), *sd = malloc(sizeof(double) * F);
  double* S = malloc(sizeof(double) * B * B);
  double* dxh = malloc(sizeof(double) * B), *xh = malloc(sizeof(double) * B);
  for (long f = 0; f < F; f++) {
    double s = 0;
    for (int b = 0; b < B; b++) s += v[(long)b * F + f];
    double m = s / B, q = 0;
    for (int b = 0; b < B; b++) { double d = v[(long)b * F + f] - m; q += d * d; }
    mu[f] = m; sd[f] = sqrt(q / (B - 1));
  }
  for (int i = 0; i < B; i++)
    for (int j = 0; j < B; j++) S[i * B + j] = sign * ((double)dG[i * B + j] + (double)dG[j * B + i]);
  for (long f = 0; f < F; f++) {
    double den = sd[f] + eps;
    for (int b = 0; b < B; b++) xh[b] = (v[(long)b * F + f] - mu[f]) / den;
    double mean_d = 0, dot = 0;
    for (int i = 0; i < B; i++) {
      double a = 0;
      for (int j = 0; j < B; j++) a += S[i * B + j] * xh[j];
      dxh[i] = a / (double)F;
      mean_d += dxh[i];
      dot += dxh[i] * xh[i];
    }
    mean_d /= B;
    for (int b = 0; b < B; b++) {
      /* torch's std backward defines the gradient through std as 0 where std == 0 (masked_fill) */
      double through_std = sd[f] > 0 ? (v[(long)b * F + f] - mu[f]) / ((B - 1) * sd[f]) * dot / den : 0.0;
      double d = (dxh[b] - mean_d) / den - through_std;
      if (jac) d *= jac[(long)b * F + f];
      long idx = (long)b * F + f;
      dx[idx] = accumulate ? (float)((double)dx[idx] + d) : (float)d;
    }
  }
  free(mu); free(sd); free(S); free(dxh); free(xh);
}

void oq_corr_bwd(const float* dG, const float* x, int B, long F, float eps, float* dx) {
  double* v = malloc(sizeof(double) * B * F);
  for (long i = 0; i < (long)B * F; i++) v[i] = x[i];
  corr_bwd_core(dG, v, B, F, eps, NULL, dx, 0, 1.0);
  free(v);
}

/* ------------------------------------------------------------------ R4 ADMM site ------------ */
/* activation_quantize_fn.forward with the corr pair (ADMM tree :109-123; Office :130-147):
 * x_q (elementwise, fp32 exact as above) and D = corr(t,t) - corr(x,x), t the PRE-round transform. */
void oq_site_fwd(const float* x, int B, long F, int k, float r, float eps, float* xq, float* D) {
  long N = (long)B * F;
  float* t = malloc(sizeof(float) * N);
  float* Gx = malloc(sizeof(float) * B * B), *Gt = malloc(sizeof(float) * B * B);
  oq_act_quant_fwd(x, xq, t, NULL, N, k, r, OQ_FORMULA_ADMM);
  oq_corr_fwd(x, B, F, eps, Gx);
  oq_corr_fwd(t, B, F, eps, Gt);
  for (int i = 0; i < B * B; i++) D[i] = Gt[i] - Gx[i];
  free(t); free(Gx); free(Gt);
}

/* dx = g * dt/dx  +  corr-path gradients for upstream dD (w.r.t. D): through corr(t,t) with +dD and the
 * chain dt/dx, and through corr(x,x) with -dD. */
void oq_site_bwd(const float* g, const float* dD, const float* x, int B, long F, float r, float eps,
                 float* dx) {
  long N = (long)B * F;
  double* xv = malloc(sizeof(double) * N), *tv = malloc(sizeof(double) * N), *jac = malloc(sizeof(double) * N);
  float* tf = malloc(sizeof(float) * N), *dummy = malloc(sizeof(float) * N);
  oq_act_quant_fwd(x, dummy, tf, NULL, N, 8, r, OQ_FORMULA_ADMM);
  const double c = 2.0 / sqrt(2.0 * M_PI);
  for (long i = 0; i < N; i++) {
    xv[i] = x[i];
    tv[i] = tf[i];
    jac[i] = (double)r * c * exp(-0.5 * xv[i] * xv[i]);
    dx[i] = g ? (float)((double)g[i] * jac[i]) : 0.0f;
  }
  corr_bwd_core(dD, tv, B, F, eps, jac, dx, 1, 1.0);
  corr_bwd_core(dD, xv, B, F, eps, NULL, dx, 1, -1.0);
  free(xv); free(tv); free(jac); free(tf); free(dummy);
}

/* ------------------------------------------------------------------ R6 ADMM loss ------------ */
/* utils/admm.py:24-33. D is [b,b]; A, gamma are [dim,dim] (sliced to [:b,:b]).  Outputs: loss and
 * gradients w.r.t. D [b,b], A and gamma ([dim,dim], zero outside the slice). */
void oq_admm_loss(const float* D, int b, const float* A, const float* gamma, int dim, float mu, float rho,
                  float* loss, float* dD, float* dA, float* dgamma) {
  double n = (double)b * b, sabs = 0, ssq = 0, srel = 0;
  for (int i = 0; i < b; i++)
    for (int j = 0; j < b; j++) {
      double a = A[i * dim + j], d = D[i * b + j] - a;
      sabs += fabs(a);
      ssq += d * d;
      srel += gamma[i * dim + j] * fabs(d);
    }
  double rms = sqrt(ssq / n);
  *loss = (float)(mu * sabs / n + rho / 2 * rms + srel / n);
  if (dA) memset(dA, 0, sizeof(float) * dim * dim);
  if (dgamma) memset(dgamma, 0, sizeof(float) * dim * dim);
  for (int i = 0; i < b; i++)
    for (int j = 0; j < b; j++) {
      double a = A[i * dim + j], d = D[i * b + j] - a, gm = gamma[i * dim + j];
      double sg = (d > 0) - (d < 0), sa = (a > 0) - (a < 0);
      double gD = rho / 2 * d / (n * rms) + gm * sg / n;
      if (dD) dD[i * b + j] = (float)gD;
      if (dA) dA[i * dim + j] = (float)(mu * sa / n - gD);
      if (dgamma) dgamma[i * dim + j] = (float)(fabs(d) / n);
    }
}

/* ------------------------------------------------------------------ R7 ADMM update ---------- */
/* utils/optimizer.py:97-124 for one site, in place. */
void oq_admm_update(const float* D, int b, float* A, float* gamma, int dim, float mu, float rho) {
  long n = (long)dim * dim;
  double* V = malloc(sizeof(double) * n), *Dp = malloc(sizeof(double) * n);
  double nv = 0;
  for (int i = 0; i < dim; i++)
    for (int j = 0; j < dim; j++) {
      double d = (i < b && j < b) ? D[i * b + j] : 0.0;
      Dp[i * dim + j] = d;
      V[i * dim + j] = d + (1.0 / rho) * gamma[i * dim + j];
      nv += V[i * dim + j] * V[i * dim + j];
    }
  nv = sqrt(nv);
  double thr = (double)mu / rho;
  for (long i = 0; i < n; i++) {
    double a = (nv > thr) ? (1.0 - thr / nv) * V[i] : 0.0;
    A[i] = (float)a;
    gamma[i] = (float)(gamma[i] + rho * (Dp[i] - a));
  }
  free(V); free(Dp);
}

/* ------------------------------------------------------------------ R8 SGD ------------------ */
/* utils/optimizer.py:212-229,251,255: d_p = g + wd*p; buf = mom*buf + (1-damp)*d_p (first: buf=d_p);
 * step dir = nesterov ? d_p + mom*buf : buf ; p -= lr*dir.  g is overwritten with the step direction
 * like the reference's in-place d_p (it becomes p.grad for tensors not in idx). */
void oq_sgd_step(float* p, float* g, float* buf, long n, float lr, float mom, float damp, float wd,
                 int nesterov, int first) {
  for (long i = 0; i < n; i++) {
    float d = g[i];
    if (wd != 0.0f) d = d + wd * p[i];
    float dir = d;
    if (mom != 0.0f) {
      float bv = first ? d : (buf[i] * mom + (1.0f - damp) * d);
      buf[i] = bv;
      dir = nesterov ? d + mom * bv : bv;
    }
    p[i] = p[i] - lr * dir;
    g[i] = dir;
  }
}

/* utils/optimizer.py:6-13,233-249: grad_out = d_p * sigmoid_d(transform(w_cdf)) * w_pdf */
void oq_sgd_grad_approx(const float* dir, const float* w_cdf, const float* w_pdf, float* gout, long n,
                        int bitW, float lam, float lam2) {
  double nn = (double)((1 << bitW) - 1);
  for (long i = 0; i < n; i++) {
    double a = ((double)w_cdf[i] + 0.5) * nn;
    double fr = a - floor(a);                 /* python/torch `% 1` (result in [0,1)) */
    double tr = fr * lam2 * 2.0;
    double sg = 1.0 / (1.0 + exp(-tr));
    gout[i] = (float)((double)dir[i] * (sg * (1.0 - sg) * lam) * (double)w_pdf[i]);
  }
}

/* ------------------------------------------------------------------ N1: batch-norm folded into the ADMM site ---- */
/* Caller being restated: `out, loss = self.act_q0(self.bn0(conv(x))); out += shortcut; out = relu(out)` of the ADMM
 * tree's block, cdf_alignment_admm/resnet-56-cifar-10/model/resnet.py:78-98, i.e. training-mode nn.BatchNorm2d followed
 * by activation_quantize_fn.forward (model/quantization.py:102-132).  torch-CPU's batch-norm evaluates the affine as
 * alpha = invstd*weight, beta' = bias - mean*alpha, out = z*alpha + beta' with the batch statistics accumulated in
 * double (aten/native/cpu/batch_norm_kernel.cpp); the HIP fold specifies the last step as ONE fma per element.
 * z is [B,C,HW] (nhwc == 0: feature f = c*HW + p) or channels-last [B,HW,C] (nhwc == 1: f = p*C + c); the site math
 * only sees the [B,F] matrix, F = C*HW, in the given memory order (D is invariant under feature permutations). */
static inline int bn_channel(long f, int C, long HW, int nhwc) { return nhwc ? (int)(f % C) : (int)(f / HW); }

/* ab = {a[C], b[C]}, save = {mean[C], invstd[C]}, var_unbiased (optional, [C]) for the running-variance update. */
void oq_bn_fold_ab(const float* z, int B, int C, long HW, int nhwc, const float* gamma, const float* beta, float bn_eps,
                   float* ab, float* save, float* var_unbiased) {
  long F = (long)C * HW;
  double* s = calloc(C, sizeof(double)), *q = calloc(C, sizeof(double));
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) s[bn_channel(f, C, HW, nhwc)] += z[(long)b * F + f];
  double n = (double)B * (double)HW;
  for (int c = 0; c < C; c++) s[c] /= n;
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) {
      int c = bn_channel(f, C, HW, nhwc);
      double d = z[(long)b * F + f] - s[c];
      q[c] += d * d;
    }
  for (int c = 0; c < C; c++) {
    double var = q[c] / n;
    float invstd = (float)(1.0 / sqrt(var + (double)bn_eps));
    float a = (gamma ? gamma[c] : 1.0f) * invstd;
    float bb = (beta ? beta[c] : 0.0f) - (float)s[c] * a;
    ab[c] = a; ab[C + c] = bb;
    save[c] = (float)s[c]; save[C + c] = invstd;
    if (var_unbiased) var_unbiased[c] = (float)(q[c] / (n - 1.0));
  }
  free(s); free(q);
}

/* forward given (a,b): x = fma(a,z,b); (x_q, D) as oq_site_fwd; y = [relu](x_q [+ residual]).  x_out optional. */
void oq_bn_site_fwd(const float* z, int B, int C, long HW, int nhwc, const float* ab, int k, float r, float eps,
                    const float* residual, int relu, float* y, float* D, float* x_out) {
  long F = (long)C * HW, N = (long)B * F;
  float* x = x_out ? x_out : malloc(sizeof(float) * N);
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) {
      int c = bn_channel(f, C, HW, nhwc);
      x[(long)b * F + f] = fmaf(ab[c], z[(long)b * F + f], ab[C + c]);
    }
  oq_site_fwd(x, B, F, k, r, eps, y, D);
  for (long i = 0; i < N; i++) {
    float v = y[i];
    if (residual) v = v + residual[i];
    if (relu) v = v > 0.0f ? v : 0.0f;      /* fmaxf(v, 0) of the kernel; NaN does not occur here */
    y[i] = v;
  }
  if (!x_out) free(x);
}

/* backward: g_y = gradient w.r.t. y; y_relu != NULL masks it by y > 0 (the masked gradient is also the residual's
 * gradient, dres); dx = site backward at x = fma(a,z,b) for upstream dD; then the training-mode batch-norm backward
 *   dgamma = sum dx*zhat, dbeta = sum dx, dz = a*(dx - mean(dx) - zhat*mean(dx*zhat)), zhat = (z-mean)*invstd.
 * dx_out (optional) = the gradient w.r.t. the batch-norm OUTPUT (what alignq_site_bwd_apply_bn writes). */
void oq_bn_site_bwd(const float* g_y, const float* dD, const float* z, int B, int C, long HW, int nhwc, const float* ab,
                    const float* save, const float* y_relu, float r, float eps, float* dz, float* dgamma, float* dbeta,
                    float* dres, float* dx_out) {
  long F = (long)C * HW, N = (long)B * F;
  float* x = malloc(sizeof(float) * N), *g = malloc(sizeof(float) * N);
  float* dx = dx_out ? dx_out : malloc(sizeof(float) * N);
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) {
      long i = (long)b * F + f;
      int c = bn_channel(f, C, HW, nhwc);
      x[i] = fmaf(ab[c], z[i], ab[C + c]);
      float gv = g_y ? g_y[i] : 0.0f;
      if (y_relu && !(y_relu[i] > 0.0f)) gv = 0.0f;
      g[i] = gv;
      if (dres) dres[i] = gv;
    }
  oq_site_bwd(g, dD, x, B, F, r, eps, dx);
  double* s0 = calloc(C, sizeof(double)), *s1 = calloc(C, sizeof(double));
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) {
      long i = (long)b * F + f;
      int c = bn_channel(f, C, HW, nhwc);
      double zh = ((double)z[i] - (double)save[c]) * (double)save[C + c];
      s0[c] += dx[i];
      s1[c] += (double)dx[i] * zh;
    }
  double n = (double)B * (double)HW;
  for (int c = 0; c < C; c++) {
    if (dgamma) dgamma[c] = (float)s1[c];
    if (dbeta) dbeta[c] = (float)s0[c];
  }
  if (dz)
    for (int b = 0; b < B; b++)
      for (long f = 0; f < F; f++) {
        long i = (long)b * F + f;
        int c = bn_channel(f, C, HW, nhwc);
        double zh = ((double)z[i] - (double)save[c]) * (double)save[C + c];
        dz[i] = (float)((double)ab[c] * ((double)dx[i] - s0[c] / n - zh * s1[c] / n));
      }
  free(s0); free(s1); free(x); free(g);
  if (!dx_out) free(dx);
}

/* the fold's affine alone: x = fma(a[c], z, b[c]) (cheap elementwise checker for x_q at a given (a,b)) */
void oq_bn_apply(const float* z, int B, int C, long HW, int nhwc, const float* ab, float* x) {
  long F = (long)C * HW;
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) {
      int c = bn_channel(f, C, HW, nhwc);
      x[(long)b * F + f] = fmaf(ab[c], z[(long)b * F + f], ab[C + c]);
    }
}

/* ------------------------------------------------------------------ general corr(x, y) ------ */
/* model/quantization.py:134-137 with y a different matrix (Office :158-161 adds eps): G = Xh Yh^T / F. */
void oq_corr_xy_fwd(const float* x, const float* y, int B, long F, float eps, float* G) {
  double* mx = malloc(sizeof(double) * F), *sx = malloc(sizeof(double) * F);
  double* my = malloc(sizeof(double) * F), *sy = malloc(sizeof(double) * F);
  double* xh = malloc(sizeof(double) * B * F), *yh = malloc(sizeof(double) * B * F);
  col_stats(x, B, F, mx, sx);
  col_stats(y, B, F, my, sy);
  for (int b = 0; b < B; b++)
    for (long f = 0; f < F; f++) {
      xh[(long)b * F + f] = (x[(long)b * F + f] - mx[f]) / (sx[f] + eps);
      yh[(long)b * F + f] = (y[(long)b * F + f] - my[f]) / (sy[f] + eps);
    }
  for (int i = 0; i < B; i++)
    for (int j = 0; j < B; j++) {
      double a = 0;
      for (long f = 0; f < F; f++) a += xh[(long)i * F + f] * yh[(long)j * F + f];
      G[i * B + j] = (float)(a / (double)F);
    }
  free(mx); free(sx); free(my); free(sy); free(xh); free(yh);
}

/* dv for v in {x (which=0), y (which=1)}: dVh = S Wh / F with S = dG (x) or dG^T (y), then the standardisation backward */
static void corr_xy_bwd_one(const float* dG, int transpose, const float* v, const float* w, int B, long F, double eps,
                            float* dv) {
  double* mv = malloc(sizeof(double) * F), *sv = malloc(sizeof(double) * F);
  double* mw = malloc(sizeof(double) * F), *sw = malloc(sizeof(double) * F);
  double* vh = malloc(sizeof(double) * B), *wh = malloc(sizeof(double) * B), *d = malloc(sizeof(double) * B);
  col_stats(v, B, F, mv, sv);
  col_stats(w, B, F, mw, sw);
  for (long f = 0; f < F; f++) {
    double den = sv[f] + eps;
    for (int b = 0; b < B; b++) {
      vh[b] = (v[(long)b * F + f] - mv[f]) / den;
      wh[b] = (w[(long)b * F + f] - mw[f]) / (sw[f] + eps);
    }
    double mean_d = 0, dot = 0;
    for (int i = 0; i < B; i++) {
      double a = 0;
      for (int j = 0; j < B; j++) a += (double)(transpose ? dG[j * B + i] : dG[i * B + j]) * wh[j];
      d[i] = a / (double)F;
      mean_d += d[i];
      dot += d[i] * vh[i];
    }
    mean_d /= B;
    for (int b = 0; b < B; b++) {
      double through_std = sv[f] > 0 ? (v[(long)b * F + f] - mv[f]) / ((B - 1) * sv[f]) * dot / den : 0.0;
      dv[(long)b * F + f] = (float)((d[b] - mean_d) / den - through_std);
    }
  }
  free(mv); free(sv); free(mw); free(sw); free(vh); free(wh); free(d);
}

void oq_corr_xy_bwd(const float* dG, const float* x, const float* y, int B, long F, float eps, float* dx, float* dy) {
  if (dx) corr_xy_bwd_one(dG, 0, x, y, B, F, eps, dx);
  if (dy) corr_xy_bwd_one(dG, 1, y, x, B, F, eps, dy);
}
