/* orc_math.h -- TEST INFRASTRUCTURE (oracle). Not part of the product path.
 *
 * Deterministic fp64 exp/log built only from IEEE-754 add/mul/div/fma and
 * integer bit manipulation, so that the CPU oracle ("rung B", parallel
 * schedule) and the gfx950 kernels produce bit-identical values.  The device
 * copy lives in epievo_amd/csrc/epv_math.h; both files spell out the same
 * operation sequence (this one is plain C for gcc, that one is HIP).
 *
 * The reference itself calls glibc exp/log (ContinuousTimeMarkovModel.cpp:120,
 * :149; SingleSiteSampler.cpp:207,214,299,304,524); "rung A" of the oracle
 * keeps libm so that it stays bit-identical to the reference.
 *
 * Accuracy (checked in tests/test_math.py against mpmath): < 1 ulp.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>

static inline uint64_t orc_d2u(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static inline double orc_u2d(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }

/* natural log.  x = 2^e * m, m in [sqrt(1/2), sqrt(2)); f = m-1; s = f/(2+f);
 * log(1+f) = 2 atanh(s) = 2s + s*R(z), z = s*s, R(z) = sum_{k>=1} 2 z^k/(2k+1);
 * and 2s = f - s*f exactly, so log(1+f) = f - s*(f - R). */
static inline double orc_log(double x) {
  uint64_t ux = orc_d2u(x);
  int e = 0;
  if (ux >= 0x7ff0000000000000ULL) {           /* inf, nan, negative, -0 */
    if (ux == 0x7ff0000000000000ULL) return x; /* +inf */
    if (ux == 0x8000000000000000ULL) return -INFINITY;
    return NAN;                                /* nan or negative */
  }
  if (ux < 0x0010000000000000ULL) {            /* +0 or subnormal */
    if (ux == 0) return -INFINITY;
    x *= 18014398509481984.0;                  /* 2^54 */
    ux = orc_d2u(x);
    e = -54;
  }
  /* shift so that the split point of the mantissa is sqrt(2)/2 */
  ux += 0x3ff0000000000000ULL - 0x3fe6a09e667f3bcdULL;
  e += (int)(ux >> 52) - 1023;
  ux = (ux & 0x000fffffffffffffULL) + 0x3fe6a09e667f3bcdULL;
  const double m = orc_u2d(ux);
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double r = 2.0 / 23.0;
  r = fma(z, r, 2.0 / 21.0);
  r = fma(z, r, 2.0 / 19.0);
  r = fma(z, r, 2.0 / 17.0);
  r = fma(z, r, 2.0 / 15.0);
  r = fma(z, r, 2.0 / 13.0);
  r = fma(z, r, 2.0 / 11.0);
  r = fma(z, r, 2.0 / 9.0);
  r = fma(z, r, 2.0 / 7.0);
  r = fma(z, r, 2.0 / 5.0);
  r = fma(z, r, 2.0 / 3.0);
  r = z * r;
  const double dk = (double)e;
  /* ln2 split: hi has 32 significant bits so dk*hi is exact */
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double t = s * (f - r) - dk * ln2_lo;
  return dk * ln2_hi + (f - t);
}

/* exp.  k = round(x/ln2); r = x - k ln2 (two-step fma); degree-13 Taylor
 * polynomial in Horner/fma form; scale by 2^k through the exponent field. */
static inline double orc_exp(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return INFINITY;
  if (x < -745.2) return 0.0;
  const double inv_ln2 = 1.44269504088896338700e+00;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double kr = x * inv_ln2;
  const int k = (int)(kr + (x < 0.0 ? -0.5 : 0.5));
  const double kd = (double)k;
  double r = fma(-kd, ln2_hi, x);
  r = fma(-kd, ln2_lo, r);
  double p = 1.0 / 6227020800.0;           /* 1/13! */
  p = fma(r, p, 1.0 / 479001600.0);        /* 1/12! */
  p = fma(r, p, 1.0 / 39916800.0);         /* 1/11! */
  p = fma(r, p, 1.0 / 3628800.0);          /* 1/10! */
  p = fma(r, p, 1.0 / 362880.0);           /* 1/9!  */
  p = fma(r, p, 1.0 / 40320.0);            /* 1/8!  */
  p = fma(r, p, 1.0 / 5040.0);             /* 1/7!  */
  p = fma(r, p, 1.0 / 720.0);              /* 1/6!  */
  p = fma(r, p, 1.0 / 120.0);              /* 1/5!  */
  p = fma(r, p, 1.0 / 24.0);               /* 1/4!  */
  p = fma(r, p, 1.0 / 6.0);                /* 1/3!  */
  p = fma(r, p, 0.5);
  p = fma(r, p, 1.0);
  p = fma(r, p, 1.0);
  if (k < -1021) {
    /* result may be subnormal: scale in two exact-power-of-two steps */
    const double s1 = orc_u2d((uint64_t)(k + 1000 + 1023) << 52);
    return (p * s1) * orc_u2d((uint64_t)(1023 - 1000) << 52); /* 2^-1000 */
  }
  if (k > 1023) {
    const double s1 = orc_u2d((uint64_t)(k - 1 + 1023) << 52);
    return (p * s1) * 2.0;
  }
  return p * orc_u2d((uint64_t)(k + 1023) << 52);
}

#endif
