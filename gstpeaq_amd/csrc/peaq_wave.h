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

// x^y for x > 0 as exp(y ln x).  OCML's pow() spends ~250 instructions on a correctly
// rounded result and on special cases that cannot occur here (the arguments are positive
// pattern energies); exp(y*log(x)) is ~180 with a relative error of |y ln x| ulp -- a few
// 1e-15 at most for the ranges of this model, against a parity bar of 1e-7.
__device__ __forceinline__ double pow_pos(double x, double y) { return exp(y * log(x)); }

// LDS traffic of ONE wave is executed in program order by the hardware; this
// only stops the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

}  // namespace peaq
