/* Exhaustive accuracy check of ALIGNQ-NERF32 (oracle/alignq_oracle.c: oq_nerf32_1, the same operation sequence the HIP
 * kernels run) against erf(y/sqrt(2)) in double, over EVERY non-negative fp32 (the function is odd by construction).
 * Prints the maximum absolute error in units of 2^-24 (= 1 ulp of a result in [0.5, 1): the unit that decides a bin,
 * because every consumer forms 1 + nerf32 first), the maximum error in ulps of the result itself, and the number of
 * places where the function decreases between neighbouring floats.
 * Build: gcc -O2 -ffp-contract=off -mfma -fopenmp verify_nerf.c ../../oracle/alignq_oracle.c -lm -o verify_nerf */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

float oq_nerf32_1(float y);

int main(void) {
  double max_abs = 0, max_ulp = 0;
  long nonmono = 0;
  float y_abs = 0, y_ulp = 0;
  const uint32_t last = 0x7f800000u;  /* +inf */
#pragma omp parallel
  {
    double m_abs = 0, m_ulp = 0;
    float ya = 0, yu = 0;
    long nm = 0;
#pragma omp for schedule(static)
    for (int64_t u = 0; u <= (int64_t)last; u++) {
      uint32_t b = (uint32_t)u;
      float y;
      memcpy(&y, &b, 4);
      float got = oq_nerf32_1(y);
      double ref = erf((double)y / sqrt(2.0));
      double e = fabs((double)got - ref);
      float rf = (float)ref;
      double ulp = (double)(nextafterf(rf, INFINITY) - rf);
      if (rf >= 1.0f) ulp = ldexp(1.0, -24);
      if (e > m_abs) { m_abs = e; ya = y; }
      if (ulp > 0 && rf > 1e-30f && e / ulp > m_ulp) { m_ulp = e / ulp; yu = y; }
      if (b > 0) {
        uint32_t pb = b - 1;
        float py;
        memcpy(&py, &pb, 4);
        if (oq_nerf32_1(py) > got) nm++;
      }
    }
#pragma omp critical
    {
      if (m_abs > max_abs) { max_abs = m_abs; y_abs = ya; }
      if (m_ulp > max_ulp) { max_ulp = m_ulp; y_ulp = yu; }
      nonmono += nm;
    }
  }
  printf("max |err| = %.4f * 2^-24 at y=%.9g ; max err = %.3f ulp(result) at y=%.9g ; decreasing steps %ld\n",
         max_abs / ldexp(1.0, -24), y_abs, max_ulp, y_ulp, nonmono);
  printf("nerf32(0)=%g nerf32(-0)=%g nerf32(inf)=%g nerf32(-7)=%g nerf32(nan)=%g\n", oq_nerf32_1(0.0f), oq_nerf32_1(-0.0f),
         oq_nerf32_1(INFINITY), oq_nerf32_1(-7.0f), oq_nerf32_1(NAN));
  return max_abs / ldexp(1.0, -24) < 0.75 ? 0 : 1;
}
