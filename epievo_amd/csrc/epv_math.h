// epv_math.h -- deterministic fp64 exp/log for the gfx950 kernels.
//
// Built only from IEEE-754 add/mul/div/fma (v_fma_f64 etc.) and integer bit
// manipulation, and compiled with -ffp-contract=off, so every value is bit-identical
// to the CPU oracle's independent restatement (oracle/orc_math.h) -- that is what lets
// the parity tests demand exact equality of jump times, states and J/D.  OCML's
// exp/log are NOT used: they differ from glibc in the last ulp and would make
// "bit-exact for a fixed seed" unprovable (SURVEY.md section 7 "hard parts").
//
// The reference calls glibc exp/log at ContinuousTimeMarkovModel.cpp:120,149 and
// SingleSiteSampler.cpp:207,214,299,304,524; accuracy of these replacements is < 1 ulp
// (tests/test_math.py, against mpmath).
#ifndef EPV_MATH_H
#define EPV_MATH_H

#include <stdint.h>

// host + device: the ABI glue evaluates per-branch constants of the site-independent
// model with the very same functions the kernels use
#define EPV_DEV __host__ __device__ __forceinline__

EPV_DEV uint64_t epv_d2u(double x) { uint64_t u; __builtin_memcpy(&u, &x, 8); return u; }
EPV_DEV double epv_u2d(uint64_t u) { double x; __builtin_memcpy(&x, &u, 8); return x; }

// log(x), x = 2^e * m, m in [sqrt(1/2), sqrt(2)); f = m - 1; s = f/(2+f);
// log(1+f) = f - s*(f - R(s^2)), R(z) = sum_{k>=1} 2 z^k/(2k+1) (11 terms)
EPV_DEV double epv_log(double x) {
  uint64_t ux = epv_d2u(x);
  int e = 0;
  if (ux >= 0x7ff0000000000000ULL) {
    if (ux == 0x7ff0000000000000ULL) return x;
    if (ux == 0x8000000000000000ULL) return -__builtin_inf();
    return __builtin_nan("");
  }
  if (ux < 0x0010000000000000ULL) {
    if (ux == 0) return -__builtin_inf();
    x *= 18014398509481984.0;  // 2^54
    ux = epv_d2u(x);
    e = -54;
  }
  ux += 0x3ff0000000000000ULL - 0x3fe6a09e667f3bcdULL;
  e += (int)(ux >> 52) - 1023;
  ux = (ux & 0x000fffffffffffffULL) + 0x3fe6a09e667f3bcdULL;
  const double m = epv_u2d(ux);
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  double r = 2.0 / 23.0;
  r = __builtin_fma(z, r, 2.0 / 21.0);
  r = __builtin_fma(z, r, 2.0 / 19.0);
  r = __builtin_fma(z, r, 2.0 / 17.0);
  r = __builtin_fma(z, r, 2.0 / 15.0);
  r = __builtin_fma(z, r, 2.0 / 13.0);
  r = __builtin_fma(z, r, 2.0 / 11.0);
  r = __builtin_fma(z, r, 2.0 / 9.0);
  r = __builtin_fma(z, r, 2.0 / 7.0);
  r = __builtin_fma(z, r, 2.0 / 5.0);
  r = __builtin_fma(z, r, 2.0 / 3.0);
  r = z * r;
  const double dk = (double)e;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double t = s * (f - r) - dk * ln2_lo;
  return dk * ln2_hi + (f - t);
}

// exp(x): k = round(x/ln2), r = x - k ln2, degree-13 Taylor in Horner/fma form
EPV_DEV double epv_exp(double x) {
  if (x != x) return x;
  if (x > 709.782712893384) return __builtin_inf();
  if (x < -745.2) return 0.0;
  const double inv_ln2 = 1.44269504088896338700e+00;
  const double ln2_hi = 6.93147180369123816490e-01;
  const double ln2_lo = 1.90821492927058770002e-10;
  const double kr = x * inv_ln2;
  const int k = (int)(kr + (x < 0.0 ? -0.5 : 0.5));
  const double kd = (double)k;
  double r = __builtin_fma(-kd, ln2_hi, x);
  r = __builtin_fma(-kd, ln2_lo, r);
  double p = 1.0 / 6227020800.0;
  p = __builtin_fma(r, p, 1.0 / 479001600.0);
  p = __builtin_fma(r, p, 1.0 / 39916800.0);
  p = __builtin_fma(r, p, 1.0 / 3628800.0);
  p = __builtin_fma(r, p, 1.0 / 362880.0);
  p = __builtin_fma(r, p, 1.0 / 40320.0);
  p = __builtin_fma(r, p, 1.0 / 5040.0);
  p = __builtin_fma(r, p, 1.0 / 720.0);
  p = __builtin_fma(r, p, 1.0 / 120.0);
  p = __builtin_fma(r, p, 1.0 / 24.0);
  p = __builtin_fma(r, p, 1.0 / 6.0);
  p = __builtin_fma(r, p, 0.5);
  p = __builtin_fma(r, p, 1.0);
  p = __builtin_fma(r, p, 1.0);
  if (k < -1021) {
    const double s1 = epv_u2d((uint64_t)(k + 1000 + 1023) << 52);
    return (p * s1) * epv_u2d((uint64_t)(1023 - 1000) << 52);
  }
  if (k > 1023) {
    const double s1 = epv_u2d((uint64_t)(k - 1 + 1023) << 52);
    return (p * s1) * 2.0;
  }
  return p * epv_u2d((uint64_t)(k + 1023) << 52);
}

#endif
