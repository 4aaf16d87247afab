// peaq_wave.h -- wave64 primitives used by the PEAQ kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace peaq {

struct cplx {
  double re, im;
};

__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) {
  return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
}
// multiply by -i (forward-transform quarter turn)
__device__ __forceinline__ cplx cmul_mi(cplx a) { return {a.im, -a.re}; }

// 4-point forward DFT in registers
__device__ __forceinline__ void dft4(cplx& a0, cplx& a1, cplx& a2, cplx& a3) {
  const cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = cmul_mi(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// 16-point forward DFT in registers: x[n] -> X[k], natural order in and out.
// n = 4 n1 + n2, k = k1 + 4 k2.
__device__ __forceinline__ void dft16(cplx (&x)[16]) {
  // W16^m = exp(-2 pi i m / 16)
  constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508977173;   // cos/sin(pi/8)
  constexpr double c2 = 0.70710678118654752440;                                // cos(pi/4)
#pragma unroll
  for (int n2 = 0; n2 < 4; ++n2) dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2]);
  // now x[4 k1 + n2] holds y[n2][k1]; twiddle by W16^(n2 k1)
  x[5] = cmul(x[5], {c1, -s1});    // n2=1,k1=1 : m=1
  x[6] = cmul(x[6], {c2, -c2});    // n2=2,k1=1 : m=2
  x[7] = cmul(x[7], {s1, -c1});    // n2=3,k1=1 : m=3
  x[9] = cmul(x[9], {c2, -c2});    // n2=1,k1=2 : m=2
  x[10] = cmul_mi(x[10]);          // n2=2,k1=2 : m=4
  x[11] = cmul(x[11], {-c2, -c2}); // n2=3,k1=2 : m=6
  x[13] = cmul(x[13], {s1, -c1});  // n2=1,k1=3 : m=3
  x[14] = cmul(x[14], {-c2, -c2}); // n2=2,k1=3 : m=6
  x[15] = cmul(x[15], {-c1, s1});  // n2=3,k1=3 : m=9
#pragma unroll
  for (int k1 = 0; k1 < 4; ++k1) dft4(x[4 * k1], x[4 * k1 + 1], x[4 * k1 + 2], x[4 * k1 + 3]);
  // x[4 k1 + k2] = X[k1 + 4 k2]  -> transpose to natural order
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = a + 1; b < 4; ++b) {
      const cplx t = x[4 * a + b];
      x[4 * a + b] = x[4 * b + a];
      x[4 * b + a] = t;
    }
}

// ---- cross-lane reductions over the 64 lanes of a wave (result in every lane)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ double wave_prod(double v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v *= __shfl_xor(v, d, 64);
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
  return v;
}
__device__ __forceinline__ int wave_or_i(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v |= __shfl_xor(v, d, 64);
  return v;
}

// ---- logarithm and exponential for this model ------------------------------------------
// OCML's log() and exp() are correctly rounded over the whole double range at ~110 and ~54
// instructions; the model calls them a few times per band and frame, which made them a fifth of
// the front end's and half of the back end's instruction count.  The versions below (~33 / ~22
// instructions) are accurate to about 1 ulp on what the model feeds them -- finite positive
// arguments for the logarithm -- against a parity bar of 1e-7 (tools/check_math.hip measures them
// against OCML on the GPU).

// ln x for finite x > 0 (subnormals included).  x = m 2^e with m in [sqrt(1/2), sqrt(2));
// ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716: odd series up to s^19 (truncation
// < 2e-17 relative); e ln 2 is added as a 32-bit high part (exact product) plus a low part.
__device__ __forceinline__ double log_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);         // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double num = m - 1., den = m + 1.;           // both exact
  double r = __builtin_amdgcn_rcp(den);              // den in [1.7, 2.42): two Newton steps, then
  r = fma(fma(-den, r, 1.), r, r);                   // the quotient with one residual correction
  r = fma(fma(-den, r, 1.), r, r);
  double s = num * r;
  s = fma(fma(-den, s, num), r, s);
  const double z = s * s;
  double p = 1. / 19;
  p = fma(p, z, 1. / 17);
  p = fma(p, z, 1. / 15);
  p = fma(p, z, 1. / 13);
  p = fma(p, z, 1. / 11);
  p = fma(p, z, 1. / 9);
  p = fma(p, z, 1. / 7);
  p = fma(p, z, 1. / 5);
  p = fma(p, z, 1. / 3);
  const double t = s + s;
  const double lm = fma(t * z, p, t);                // 2 s (1 + z P(z))
  const double ef = (double)e;
  return fma(ef, 6.93147180369123816490e-01, fma(ef, 1.90821492927058770002e-10, lm));
}

// the same for any x >= 0 or NaN: ln 0 = -inf, ln inf = inf (digital silence reaches the
// logarithms of the error-harmonic-structure and of the filter-bank slope computation)
__device__ __forceinline__ double log_nonneg(double x) {
  const double l = log_pos(x);
  return x == 0. ? -__builtin_inf() : (x == __builtin_inf() ? __builtin_inf() : l);
}

// e^x for any finite x or -inf (underflows to 0, overflows to inf through ldexp).
// x = n ln 2 + r, |r| <= 0.3466; e^r as its Taylor polynomial of degree 12 (truncation 1.7e-16).
__device__ __forceinline__ double exp_fast(double x) {
  x = fmin(fmax(x, -1000.), 1000.);
  const double n = __builtin_rint(x * 1.44269504088896338700e+00);
  double r = fma(-n, 6.93147180369123816490e-01, x);
  r = fma(-n, 1.90821492927058770002e-10, r);
  double p = 1. / 479001600.;
  p = fma(p, r, 1. / 39916800.);
  p = fma(p, r, 1. / 3628800.);
  p = fma(p, r, 1. / 362880.);
  p = fma(p, r, 1. / 40320.);
  p = fma(p, r, 1. / 5040.);
  p = fma(p, r, 1. / 720.);
  p = fma(p, r, 1. / 120.);
  p = fma(p, r, 1. / 24.);
  p = fma(p, r, 1. / 6.);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.);
  p = fma(p, r, 1.);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// a / b for finite b of moderate magnitude (no scaling against overflow / underflow of the
// reciprocal, no special cases): reciprocal with two Newton steps, quotient with one residual
// correction; <= 1 ulp.  8 instructions instead of the 11 of the IEEE sequence.
__device__ __forceinline__ double div_fast(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.), r, r);
  r = fma(fma(-b, r, 1.), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// sqrt(x) for finite x >= 0 (0 -> 0), <= 1 ulp: reciprocal square root, two coupled Newton steps
// on (sqrt, 1/(2 sqrt)), one residual correction.  14 instructions instead of 23.
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return x == 0. ? 0. : g;
}

constexpr double kInvLn10 = 0.43429448190325182765;   // log10 x = ln x / ln 10
constexpr double kLn2 = 0.69314718055994530942;       // 2^x = e^(x ln 2)

// x^y for x > 0 as exp(y ln x): relative error about |y ln x| ulp -- a few 1e-15 at most for the
// ranges of this model.
__device__ __forceinline__ double pow_pos(double x, double y) { return exp_fast(y * log_pos(x)); }

// LDS traffic of ONE wave is executed in program order by the hardware; this
// only stops the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

}  // namespace peaq
