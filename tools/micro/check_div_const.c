// Exhaustive check of the constant division used by the K1 kernels (csrc/consensus_params.hip, rc_div_fast):
//   q0 = x * RN(1/D);  q = fma(fma(-q0, D, x), RN(1/D), q0)
// against the IEEE quotient x / D for EVERY fp32 bit pattern x and every D in [lo, hi].  Reports, per D, the number of
// mismatches whose exact quotient is NORMAL (must be 0) and the largest |x| among mismatches (all of them have a
// subnormal quotient; the kernels guard |sum| < 1e-30 with the true division).
//   gcc -O2 -mfma -ffp-contract=off -fopenmp tools/micro/check_div_const.c -o /tmp/check_div -lm && /tmp/check_div 2 66
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
  int lo = argc > 1 ? atoi(argv[1]) : 2, hi = argc > 2 ? atoi(argv[2]) : 66, bad_total = 0;
  for (int D = lo; D <= hi; ++D) {
    const float d = (float)D, rcp = 1.0f / d;
    long bad_normal = 0, bad_sub = 0;
    float worst = 0.f;
#pragma omp parallel for reduction(+ : bad_normal, bad_sub) reduction(max : worst) schedule(static)
    for (long long u = 0; u < (1ll << 32); ++u) {
      uint32_t bits = (uint32_t)u;
      float x;
      memcpy(&x, &bits, 4);
      if (x != x || isinf(x)) continue;
      const float ref = x / d;
      const float q0 = x * rcp;
      const float q = fmaf(fmaf(-q0, d, x), rcp, q0);
      if (memcmp(&q, &ref, 4) != 0 && !(q == 0.f && ref == 0.f)) {
        if (fabsf(ref) >= 1.17549435e-38f) ++bad_normal; else ++bad_sub;
        if (fabsf(x) > worst) worst = fabsf(x);
      }
    }
    printf("D=%2d  mismatches with a normal quotient: %ld   with a subnormal quotient: %ld   largest |x| among them: %g\n", D,
           bad_normal, bad_sub, worst);
    fflush(stdout);
    bad_total += bad_normal != 0;
  }
  return bad_total ? 1 : 0;
}
