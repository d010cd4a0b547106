/* Exhaustive proof obligations for the fma-based division used by the HIP kernels (alignq_math.h: div_sqrt2,
 * div_levels).  The arithmetic SPEC stays "IEEE-754 division" (oracle/alignq_oracle.c divides); the kernels use
 *     q0 = x*y; r = fma(-q0, d, x); q = fma(r, y, q0),   y = RN(1/d)
 * which this program checks against x/d for
 *   (1) d = float(sqrt(2)) and EVERY finite float x (2^32 cases, both zeros; 0 < |x| < 1e-30 reported separately),
 *   (2) d = n = 2^k-1, k = 1..16, and every integer-valued x with |x| <= 8*n + 2 (bin indices).
 * Prints the number of mismatches (must be 0 outside the documented exclusions) and exits non-zero otherwise.
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp verify_div.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float fdiv_fma(float x, float d, float y) {
  float q0 = x * y;
  float r = fmaf(-q0, d, x);
  float q = fmaf(r, y, q0);
  return (q0 == 0.0f || isinf(q0)) ? q0 : q;   /* signed zero / inf: q0 is already the quotient */
}

int main(void) {
  const float d = 1.41421356237309504880f;
  const float y = 1.0f / d;
  long bad = 0, bad_tiny = 0;
#pragma omp parallel for reduction(+ : bad, bad_tiny) schedule(static)
  for (int64_t u = 0; u < (1LL << 32); u++) {
    uint32_t b = (uint32_t)u;
    float x;
    memcpy(&x, &b, 4);
    if (!isfinite(x)) continue;
    float ref = x / d, got = fdiv_fma(x, d, y);
    if (memcmp(&ref, &got, 4) != 0) {
      if (x != 0.0f && fabsf(x) < 1e-30f) bad_tiny++;   /* remainder not exact once q*d underflows: irrelevant, see caller */
      else bad++;
    }
  }
  printf("sqrt2: mismatches %ld (|x|>=1e-30), %ld (|x|<1e-30)\n", bad, bad_tiny);
  long badn = 0;
  for (int k = 1; k <= 16; k++) {
    float n = (float)((1 << k) - 1), yn = 1.0f / n;
    long lim = 8L * ((1 << k) - 1) + 2;
    for (long v = -lim; v <= lim; v++) {
      float x = (float)v, ref = x / n, got = fdiv_fma(x, n, yn);
      if (memcmp(&ref, &got, 4) != 0) badn++;
    }
    {
      float x = -0.0f, ref = x / n, got = fdiv_fma(x, n, yn);   /* rint() of a small negative value */
      if (memcmp(&ref, &got, 4) != 0) badn++;
    }
  }
  printf("levels: mismatches %ld\n", badn);
  return (bad == 0 && badn == 0) ? 0 : 1;
}
