/* Exhaustive proof obligation for the fma-based division of a level index used by the HIP kernels (alignq_math.h:
 * div_levels).  The arithmetic SPEC stays "IEEE-754 division" (oracle/alignq_oracle.c divides); the kernels use
 *     q0 = b*y; r = fma(-q0, n, b); q = fma(r, y, q0); result = q | signbit(b),   y = RN(1/n)
 * which this program checks bit for bit against b/n for n = 2^k-1, k = 1..16, and every integer-valued b with
 * |b| <= 8*n + 2 (the level indices of activations with act_range <= 8 and of weights) plus b = -0 (rint() of a small
 * negative value: the fma chain alone would give +0, hence the OR of b's sign bit).
 * (Round 1-2 also divided by sqrt(2) this way; round 3's NERF32 takes the deviate itself, so that part is gone.)
 * Prints the number of mismatches (must be 0) and exits non-zero otherwise.
 * Build: gcc -O2 -ffp-contract=off -mfma verify_div.c -lm */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float div_levels(float b, float n, float y) {
  float q0 = b * y;
  float r = fmaf(-q0, n, b);
  float q = fmaf(r, y, q0);
  uint32_t qb, bb;
  memcpy(&qb, &q, 4);
  memcpy(&bb, &b, 4);
  qb |= bb & 0x80000000u;
  memcpy(&q, &qb, 4);
  return q;
}

int main(void) {
  long badn = 0;
  for (int k = 1; k <= 16; k++) {
    float n = (float)((1 << k) - 1), yn = 1.0f / n;
    long lim = 8L * ((1 << k) - 1) + 2;
    for (long v = -lim; v <= lim; v++) {
      float x = (float)v, ref = x / n, got = div_levels(x, n, yn);
      if (memcmp(&ref, &got, 4) != 0) badn++;
    }
    {
      float x = -0.0f, ref = x / n, got = div_levels(x, n, yn);
      if (memcmp(&ref, &got, 4) != 0) badn++;
    }
  }
  printf("levels: mismatches %ld\n", badn);
  return badn == 0 ? 0 : 1;
}
