/* Exhaustive check (all 2^32 float bit patterns) that
 *   q = x*y; q' = fmaf(fmaf(-d, q, x), y, q)   with y = RN(1/d)
 * equals the IEEE quotient x / d for d = 3 and d = 6 on 2^-100 <= |x| <= 2^100, with the
 * guards of eigen_device.hpp div_by_const (zero keeps its sign; other magnitudes, inf and
 * NaN take the plain division).  The HIP solver uses this form in place of the division
 * expansion.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -mfma  (fmaf must be the fused instruction). */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

static inline float form(float x, float d, float y) {
  const float ax = fabsf(x);
  float q = x * y;
  float f = fmaf(fmaf(-d, q, x), y, q);
  f = ax == 0.0f ? x : f;
  if (!(ax <= 0x1p100f) || (ax < 0x1p-100f && ax != 0.0f)) {
    volatile float t = x / d; /* the guarded cases take the plain division */
    f = t;
  }
  return f;
}

int main(void) {
  const float ds[2] = {3.0f, 6.0f};
  const float ys[2] = {0x1.555556p-2f, 0x1.555556p-3f};
  long long bad_total = 0;
  for (int k = 0; k < 2; ++k) {
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long b = 0; b < (1LL << 32); ++b) {
      uint32_t u = (uint32_t)b;
      float x;
      memcpy(&x, &u, 4);
      volatile float ref = x / ds[k];
      float got = form(x, ds[k], ys[k]);
      float r = ref;
      if (!(got == r) && !(isnan(got) && isnan(r))) ++bad;
      /* zero sign */
      else if (got == 0.0f && signbit(got) != signbit(r)) ++bad;
    }
    printf("d=%g mismatches=%lld\n", ds[k], bad);
    bad_total += bad;
  }
  return bad_total != 0;
}
