// peaq_wave.h -- wave64 primitives used by the PEAQ kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

#include "peaq_device.h"

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

// 8-point forward DFT in registers, natural order in and out
__device__ __forceinline__ void dft8(cplx (&x)[8]) {
  constexpr double c = 0.70710678118654752440;
  cplx a0 = cadd(x[0], x[4]), a1 = cadd(x[1], x[5]), a2 = cadd(x[2], x[6]), a3 = cadd(x[3], x[7]);
  cplx b0 = csub(x[0], x[4]), b1 = csub(x[1], x[5]), b2 = csub(x[2], x[6]), b3 = csub(x[3], x[7]);
  b1 = {c * (b1.re + b1.im), c * (b1.im - b1.re)};      // * W8^1 = (1 - i) / sqrt 2
  b2 = cmul_mi(b2);                                       // * W8^2 = -i
  b3 = {c * (b3.im - b3.re), -c * (b3.re + b3.im)};      // * W8^3 = (-1 - i) / sqrt 2
  dft4(a0, a1, a2, a3);
  dft4(b0, b1, b2, b3);
  x[0] = a0; x[2] = a1; x[4] = a2; x[6] = a3;
  x[1] = b0; x[3] = b1; x[5] = b2; x[7] = b3;
}

// ---- cross-lane reductions over the 64 lanes of a wave (result in every lane) -----------------
// Butterfly over the lane bits 1, 2, 4, 8 with DPP row operations (VALU, no LDS traffic) and over
// 16 and 32 with gfx950's v_permlane16_swap / v_permlane32_swap.  The combining operation is
// commutative, so every lane ends with the bit-identical result.  (The generic __shfl_xor lowers to
// ds_bpermute_b32: two LDS-crossbar instructions and a round trip of ~50+ cycles per step and
// double -- the reductions were a tenth of the front end's LDS instructions.)
enum : int {
  kDppXor1 = 0xB1,         // quad_perm [1,0,3,2]
  kDppXor2 = 0x4E,         // quad_perm [2,3,0,1]
  kDppHalfMirror = 0x141,  // lane i <- lane 7 - i of its group of 8 (all lanes of a quad agree by then)
  kDppMirror = 0x140       // lane i <- lane 15 - i of its row of 16
};
// (bound_ctrl set: with every row and bank enabled the compiler then knows the destination's old value is never
// kept and does not zero it first -- two moves and a wait state per double saved; in these patterns every lane
// has a source, so the "0 for lanes without one" never shows)
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
// the values of lane ^ 16 (ROWS16) or lane ^ 32 as a pair {own half's copy, other half's}; which
// is which depends on the lane, the (commutative) caller does not care
template <bool ROWS16>
__device__ __forceinline__ void swap_halves_i(int v, int& a, int& b) {
  if (ROWS16) {
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)r[0];
    b = (int)r[1];
  } else {
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
    a = (int)r[0];
    b = (int)r[1];
  }
}
// 4 x 4 transpose between the four rows (16 lanes each) of a wave and four registers: on return x[a] holds in
// row p what x[p] held in row a.  v_permlane32_swap exchanges the upper two rows of its first operand with the
// lower two of its second, v_permlane16_swap the odd rows of the first with the even rows of the second: two of
// each per 32-bit register quadruple, no LDS.
__device__ __forceinline__ void rows_transpose4_u(unsigned& x0, unsigned& x1, unsigned& x2, unsigned& x3) {
  auto r = __builtin_amdgcn_permlane32_swap(x0, x2, false, false);
  x0 = r[0];
  x2 = r[1];
  r = __builtin_amdgcn_permlane32_swap(x1, x3, false, false);
  x1 = r[0];
  x3 = r[1];
  r = __builtin_amdgcn_permlane16_swap(x0, x1, false, false);
  x0 = r[0];
  x1 = r[1];
  r = __builtin_amdgcn_permlane16_swap(x2, x3, false, false);
  x2 = r[0];
  x3 = r[1];
}
__device__ __forceinline__ void rows_transpose4(double& x0, double& x1, double& x2, double& x3) {
  unsigned l0 = __double2loint(x0), l1 = __double2loint(x1), l2 = __double2loint(x2), l3 = __double2loint(x3);
  unsigned h0 = __double2hiint(x0), h1 = __double2hiint(x1), h2 = __double2hiint(x2), h3 = __double2hiint(x3);
  rows_transpose4_u(l0, l1, l2, l3);
  rows_transpose4_u(h0, h1, h2, h3);
  x0 = __hiloint2double(h0, l0);
  x1 = __hiloint2double(h1, l1);
  x2 = __hiloint2double(h2, l2);
  x3 = __hiloint2double(h3, l3);
}

template <bool ROWS16>
__device__ __forceinline__ void swap_halves_d(double v, double& a, double& b) {
  int alo, blo, ahi, bhi;
  swap_halves_i<ROWS16>(__double2loint(v), alo, blo);
  swap_halves_i<ROWS16>(__double2hiint(v), ahi, bhi);
  a = __hiloint2double(ahi, alo);
  b = __hiloint2double(bhi, blo);
}
// ---- neighbours, broadcasts and scans without the LDS crossbar -----------------------------------
// DPP shifts: row_shr:n / row_shl:n move data by n lanes inside a row of 16 (lane i reads lane i - n /
// i + n), wave_shr:1 / wave_shl:1 by one lane across the whole wave; lanes without a source read 0.
template <int CTRL>
__device__ __forceinline__ double dpp_d0(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true));
}
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110, kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;
__device__ __forceinline__ double lane_below(double v) { return dpp_d0<kDppWaveShr1>(v); }   // lane i <- lane i - 1
__device__ __forceinline__ double lane_above(double v) { return dpp_d0<kDppWaveShl1>(v); }   // lane i <- lane i + 1
// the value of lane l (a constant) in every lane, through the scalar registers
template <int L>
__device__ __forceinline__ double read_lane(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), L), __builtin_amdgcn_readlane(__double2loint(v), L));
}
// The carries between the four rows of 16 lanes of a scan, as two DPP broadcasts: lane 15 of rows 0 and 2 to rows
// 1 and 3 (row_bcast:15, row mask 0xA), then lane 31 -- by then the total of rows 0..1 -- to rows 2 and 3
// (row_bcast:31, row mask 0xC); the rows outside the mask read 0.  (Through the scalar registers the same took
// six lane reads, a chain of selects per lane and the scalar unit's turn-around.)
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_rows_d0(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWS, 0xF, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWS, 0xF, false));
}
// The same with the zero kept in a register of the caller's: `z` starts as 0., only the rows of ROWS are ever written
// into it, so the other rows still read 0 -- and the two moves that would set the destination to zero in front of
// every use are gone.  One such register per ROWS mask (a scan-heavy loop saves eight moves per complex scan).
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_rows_keep(double v, double& z) {
  z = __hiloint2double(__builtin_amdgcn_update_dpp(__double2hiint(z), __double2hiint(v), CTRL, ROWS, 0xF, false),
                       __builtin_amdgcn_update_dpp(__double2loint(z), __double2loint(v), CTRL, ROWS, 0xF, false));
  return z;
}
__device__ __forceinline__ double row_carry_15(double v, double& z) { return dpp_rows_keep<kDppRowBcast15, 0xA>(v, z); }
__device__ __forceinline__ double row_carry_31(double v, double& z) { return dpp_rows_keep<kDppRowBcast31, 0xC>(v, z); }
__device__ __forceinline__ double row_carry_15(double v) { return dpp_rows_d0<kDppRowBcast15, 0xA>(v); }
__device__ __forceinline__ double row_carry_31(double v) { return dpp_rows_d0<kDppRowBcast31, 0xC>(v); }

// inclusive prefix sum over the lanes: lane i gets v_0 + ... + v_i
__device__ __forceinline__ double wave_prefix_sum(double v, int lane) {
  v += dpp_d0<kDppRowShr + 1>(v);
  v += dpp_d0<kDppRowShr + 2>(v);
  v += dpp_d0<kDppRowShr + 4>(v);
  v += dpp_d0<kDppRowShr + 8>(v);
  v += row_carry_15(v);
  return v + row_carry_31(v);
}
// weighted suffix sum: lane i gets sum_{j >= i} m^(j - i) v_j (m wave-uniform)
__device__ __forceinline__ double wave_suffix_geometric(double v, double m, int lane) {
  // m^(16 - (lane & 15)): the weight of the next row's total as seen from this lane
  double wl = 1., base = m;
  const int k = 16 - (lane & 15);
#pragma unroll
  for (int bit = 0; bit < 5; ++bit) {
    wl = (k >> bit) & 1 ? wl * base : wl;
    base *= base;
  }
  const double m16 = read_lane<0>(wl);               // lane 0: k = 16
  v = fma(m, dpp_d0<kDppRowShl + 1>(v), v);
  m *= m;
  v = fma(m, dpp_d0<kDppRowShl + 2>(v), v);
  m *= m;
  v = fma(m, dpp_d0<kDppRowShl + 4>(v), v);
  m *= m;
  v = fma(m, dpp_d0<kDppRowShl + 8>(v), v);
  // suffix totals from the first lane of rows 3, 2, 1 to the end of the wave
  const double t3 = read_lane<48>(v);
  const double t2 = fma(m16, t3, read_lane<32>(v));
  const double t1 = fma(m16, t2, read_lane<16>(v));
  const int row = lane >> 4;
  return fma(wl, row == 0 ? t1 : row == 1 ? t2 : row == 2 ? t3 : 0., v);
}

// weighted prefix sum: lane i gets sum_{j <= i} m^(i - j) v_j.  m1..m8 = m, m^2, m^4, m^8 and
// wl = m^((lane & 15) + 1) are the caller's (they are loop invariants where this is used).
__device__ __forceinline__ double wave_prefix_geometric(double v, double m1, double m2, double m4, double m8, double m16,
                                                        double wl, int lane) {
  v = fma(m1, dpp_d0<kDppRowShr + 1>(v), v);
  v = fma(m2, dpp_d0<kDppRowShr + 2>(v), v);
  v = fma(m4, dpp_d0<kDppRowShr + 4>(v), v);
  v = fma(m8, dpp_d0<kDppRowShr + 8>(v), v);
  // rows 1 and 3 take in the row before them, then rows 2 and 3 the (now complete) rows 0..1: lane l of row 3 is
  // m^((l & 15) + 1 + 16) behind lane 31
  v = fma(wl, row_carry_15(v), v);
  return fma((lane >> 4) == 3 ? wl * m16 : wl, row_carry_31(v), v);
}
// the same with the row carries' zero registers kept by the caller (dpp_rows_keep)
__device__ __forceinline__ double wave_prefix_geometric(double v, double m1, double m2, double m4, double m8, double m16,
                                                        double wl, int lane, double& z15, double& z31) {
  v = fma(m1, dpp_d0<kDppRowShr + 1>(v), v);
  v = fma(m2, dpp_d0<kDppRowShr + 2>(v), v);
  v = fma(m4, dpp_d0<kDppRowShr + 4>(v), v);
  v = fma(m8, dpp_d0<kDppRowShr + 8>(v), v);
  v = fma(wl, row_carry_15(v, z15), v);
  return fma((lane >> 4) == 3 ? wl * m16 : wl, row_carry_31(v, z31), v);
}

// the same in FP32 (one DPP move per step instead of two, 2-cycle arithmetic): for recurrences that forget
// their past within a few steps
template <int CTRL>
__device__ __forceinline__ float dpp_f0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_prefix_geometric(float v, float m1, float m2, float m4, float m8, float m16, float wl,
                                                       int lane) {
  v = fmaf(m1, dpp_f0<kDppRowShr + 1>(v), v);
  v = fmaf(m2, dpp_f0<kDppRowShr + 2>(v), v);
  v = fmaf(m4, dpp_f0<kDppRowShr + 4>(v), v);
  v = fmaf(m8, dpp_f0<kDppRowShr + 8>(v), v);
  const float t0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
  const float t1 = fmaf(m16, t0, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31)));
  const float t2 = fmaf(m16, t1, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47)));
  const int row = lane >> 4;
  return fmaf(wl, row == 0 ? 0.f : row == 1 ? t0 : row == 2 ? t1 : t2, v);
}

template <typename OP>
__device__ __forceinline__ double wave_reduce_d(double v, OP op) {
  v = op(v, dpp_d<kDppXor1>(v));
  v = op(v, dpp_d<kDppXor2>(v));
  v = op(v, dpp_d<kDppHalfMirror>(v));
  v = op(v, dpp_d<kDppMirror>(v));
  double a, b;
  swap_halves_d<true>(v, a, b);
  v = op(a, b);
  swap_halves_d<false>(v, a, b);
  return op(a, b);
}
template <typename OP>
__device__ __forceinline__ int wave_reduce_i(int v, OP op) {
  v = op(v, dpp_i<kDppXor1>(v));
  v = op(v, dpp_i<kDppXor2>(v));
  v = op(v, dpp_i<kDppHalfMirror>(v));
  v = op(v, dpp_i<kDppMirror>(v));
  int a, b;
  swap_halves_i<true>(v, a, b);
  v = op(a, b);
  swap_halves_i<false>(v, a, b);
  return op(a, b);
}
__device__ __forceinline__ double wave_sum(double v) {
  return wave_reduce_d(v, [](double x, double y) { return x + y; });
}
__device__ __forceinline__ double wave_max(double v) {
  return wave_reduce_d(v, [](double x, double y) { return fmax(x, y); });
}
// Several sums for the price of one.  v_permlane32_swap hands the upper half of its first operand to the second and
// takes the second's lower half: with two DIFFERENT values as operands the sum of the two results is, in lanes 0 .. 31,
// the first value's lane + (lane + 32), and in lanes 32 .. 63 the second's -- one level of both reductions in the
// three instructions per double that one level of one reduction takes.  v_permlane16_swap does the same for the rows,
// so four values share the rest of the way (rows 0 .. 3 end as the sums of a, c, b, d).  The totals are read through
// the scalar unit: uniform, and bit-identical in every lane like wave_sum's.
__device__ __forceinline__ double fuse_halves_sum(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double fuse_rows_sum(double a, double b) {
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ double row_sum_d(double v) {      // every lane: the sum over its row of 16
  v += dpp_d<kDppXor1>(v);
  v += dpp_d<kDppXor2>(v);
  v += dpp_d<kDppHalfMirror>(v);
  v += dpp_d<kDppMirror>(v);
  return v;
}
__device__ __forceinline__ double lane_value_d(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ void wave_sum2(double a, double b, double& sa, double& sb) {
  double v = fuse_halves_sum(a, b), p, q;            // lanes 0 .. 31: a, lanes 32 .. 63: b
  swap_halves_d<true>(v, p, q);
  v = row_sum_d(p + q);
  sa = lane_value_d(v, 0);
  sb = lane_value_d(v, 32);
}
__device__ __forceinline__ void wave_sum4(double a, double b, double c, double d, double& sa, double& sb, double& sc,
                                          double& sd) {
  const double v = row_sum_d(fuse_rows_sum(fuse_halves_sum(a, b), fuse_halves_sum(c, d)));
  sa = lane_value_d(v, 0);
  sc = lane_value_d(v, 16);
  sb = lane_value_d(v, 32);
  sd = lane_value_d(v, 48);
}
__device__ __forceinline__ double wave_prod(double v) {
  return wave_reduce_d(v, [](double x, double y) { return x * y; });
}
__device__ __forceinline__ int wave_max_i(int v) {
  return wave_reduce_i(v, [](int x, int y) { return max(x, y); });
}
__device__ __forceinline__ int wave_or_i(int v) {
  return wave_reduce_i(v, [](int x, int y) { return x | y; });
}

// ---- logarithm and exponential for this model ------------------------------------------
// OCML's log() and exp() are correctly rounded over the whole double range at ~110 and ~54
// instructions; the model calls them a few times per band and frame, which made them a fifth of
// the front end's and half of the back end's instruction count.  The versions below (~33 / ~22
// instructions) are accurate to about 1 ulp on what the model feeds them -- finite positive
// arguments for the logarithm -- against a parity bar of 1e-7 (tools/check_math.hip measures them
// against OCML on the GPU).

// A double constant placed in a scalar register pair by two s_mov_b32.  The filter bank's FP64 kernel is at its
// register budget: there the compiler kept these polynomial coefficients in vector registers across the tile loop
// and spilled them to scratch; scalar copies cost no vector register and at worst a v_readlane to bring back.
#ifndef PEAQ_SK_DEFAULT
#define PEAQ_SK_DEFAULT false      // (measured on the basic version: scalar constants everywhere cost 3 % -- the
#endif                             // extra scalar instructions weigh more than the vector moves they replace)
template <long long BITS> __device__ __forceinline__ double scalar_const() {
  unsigned lo, hi;
  asm("s_mov_b32 %0, %1" : "=s"(lo) : "i"((unsigned)(BITS & 0xffffffffll)));
  asm("s_mov_b32 %0, %1" : "=s"(hi) : "i"((unsigned)((unsigned long long)BITS >> 32)));
  return __hiloint2double((int)hi, (int)lo);
}
#define PEAQ_KC(c) (SK ? scalar_const<__builtin_bit_cast(long long, (double)(c))>() : (double)(c))
__device__ __forceinline__ double fma_sgpr(double a, double b, double c_scalar) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c_scalar));
  return r;
}
#define PEAQ_SC(c) scalar_const<__builtin_bit_cast(long long, (double)(c))>()
// a * b + c with c such a scalar constant: written out because the compiler would select the two-address v_fmac_f64
// here and pay two v_mov_b32 per Horner step to move the constant into the accumulator first
template <bool SK, long long BITS> __device__ __forceinline__ double fma_const(double a, double b) {
  if constexpr (SK) {
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(scalar_const<BITS>()));
    return r;
  } else {
    return fma(a, b, __builtin_bit_cast(double, BITS));
  }
}
#define PEAQ_FMA_KC(a, b, c) fma_const<SK, __builtin_bit_cast(long long, (double)(c))>(a, b)

// ln x for finite x > 0 (subnormals included).  x = m 2^e with m in [sqrt(1/2), sqrt(2));
// ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| <= 0.1716: odd series up to s^19 (truncation
// < 2e-17 relative); e ln 2 is added as a 32-bit high part (exact product) plus a low part.
template <bool SK = PEAQ_SK_DEFAULT> __device__ __forceinline__ double log_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);         // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < PEAQ_KC(0.70710678118654752440);
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double num = m - 1., den = m + 1.;           // both exact
  double r = __builtin_amdgcn_rcp(den);              // den in [1.7, 2.42): one Newton step (the hardware's
  r = fma(fma(-den, r, 1.), r, r);                   // reciprocal is good to ~2^-26, squared: 2^-52), then the
  double s = num * r;                                // quotient with one residual correction (error x error)
  s = fma(fma(-den, s, num), r, s);
  const double z = s * s;
  double p = PEAQ_KC(1. / 19);
  p = PEAQ_FMA_KC(p, z, 1. / 17);
  p = PEAQ_FMA_KC(p, z, 1. / 15);
  p = PEAQ_FMA_KC(p, z, 1. / 13);
  p = PEAQ_FMA_KC(p, z, 1. / 11);
  p = PEAQ_FMA_KC(p, z, 1. / 9);
  p = PEAQ_FMA_KC(p, z, 1. / 7);
  p = PEAQ_FMA_KC(p, z, 1. / 5);
  p = PEAQ_FMA_KC(p, z, 1. / 3);
  const double t = s + s;
  const double lm = fma(t * z, p, t);                // 2 s (1 + z P(z))
  const double ef = (double)e;
  return fma(ef, PEAQ_KC(6.93147180369123816490e-01), fma(ef, PEAQ_KC(1.90821492927058770002e-10), lm));
}

// ln x for finite x > 0 (subnormals included) from the 129-entry table CommonTables::log_tab, which the caller
// holds in LDS (`tab`: {2 / C, ln C [- ln 2]} per bin): x = m 2^e, the bin is the nearest multiple of 1/128 to the
// fraction of 2 m, r = 2 m / C - 1 with |r| <= 2^-8, log1p(r) = r - r^2/2 + ... - r^6/6 (truncation 2^-56 / 7), and
// the power of two joins the entry in one multiply-add (relative error of that product: ln 2's, 1e-17).
// 17 vector instructions and one 16-byte LDS read against 31 of log_pos; <= 2 ulp like log_pos, <= 5 ulp for
// arguments around 1 (their bins have the centre 1: r = x - 1 exactly, ln 1 = 0 exactly); tools/check_math.hip.
// Measured (same box, 4096 pairs): +2.3 % frame-pairs/s for the basic version with the front end's and the back
// end's logarithms from the table -- the vector ALU is what the step is bound by, the LDS reads cost less.
__device__ __forceinline__ double log_tab(double x, const double* __restrict__ tab) {
  const double m = __builtin_amdgcn_frexp_mant(x);   // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  // fraction bits 51..45 of m, rounded to nearest: the carry of an all-ones fraction lands in the exponent's
  // lowest bit, which is 0 for [0.5, 1) -- so bits 20..13 of the high word read 0..128
  // (0 .. 128 for every finite positive argument; a NaN's mantissa field can say up to 255: clamped, so that the
  // read stays inside the 130-entry table whatever comes in -- the result is NaN either way)
  const unsigned idx = min(__builtin_amdgcn_ubfe((unsigned)__double2hiint(m) + 0x1000u, 13, 8), 128u);
  const double2 t = *reinterpret_cast<const double2*>(tab + 2 * idx);
  e -= idx < (unsigned)kLogTabFold ? 1 : 0;
  const double r = fma(m, t.x, -1.);
  constexpr bool SK = PEAQ_SK_DEFAULT;
  double p = -1. / 6;
  p = PEAQ_FMA_KC(p, r, 1. / 5);
  p = PEAQ_FMA_KC(p, r, -1. / 4);
  p = PEAQ_FMA_KC(p, r, 1. / 3);
  p = fma(p, r, -0.5);
  const double l1p = fma(r * r, p, r);
  return fma((double)e, PEAQ_KC(6.93147180559945286227e-01), t.y) + l1p;
}
#ifdef PEAQ_NO_LOGTAB_FE
#define FE_LOG(x, tab) log_pos(x)
#define FE_LOG_NONNEG(x, tab) log_nonneg(x)
#else
#define FE_LOG(x, tab) log_tab(x, tab)
#define FE_LOG_NONNEG(x, tab) log_tab_nonneg(x, tab)
#endif
// any x >= 0 or NaN, like log_nonneg below
__device__ __forceinline__ double log_tab_nonneg(double x, const double* __restrict__ tab) {
  const double l = log_tab(x, tab);
  return x == 0. ? -__builtin_inf() : (x == __builtin_inf() ? __builtin_inf() : l);
}

// the same for any x >= 0 or NaN: ln 0 = -inf, ln inf = inf (digital silence reaches the
// logarithms of the error-harmonic-structure and of the filter-bank slope computation)
template <bool SK = PEAQ_SK_DEFAULT> __device__ __forceinline__ double log_nonneg(double x) {
  const double l = log_pos<SK>(x);
  return x == 0. ? -__builtin_inf() : (x == __builtin_inf() ? __builtin_inf() : l);
}

// e^x for any finite x or -inf (underflows to 0, overflows to inf through ldexp).
// x = n ln 2 + r, |r| <= 0.3466; e^r as its Taylor polynomial of degree 12 (truncation 1.7e-16).
// e^x like exp_fast below, from the 64-entry table CommonTables::exp_tab (2^(j/64)) the caller holds in LDS: the
// argument is reduced to |r| <= ln 2 / 128, where e^r - 1 needs five terms instead of twelve (the next one is 3.5e-17
// relative) -- and every term of such a chain costs its multiply-add AND a move of its constant into the two-address
// accumulator.  64 ln 2 / 64 is split like ln 2 itself: the high part's product with n (17 bits) is exact.
__device__ __forceinline__ double exp_tab(double x, const double* __restrict__ tab) {
  x = fmin(fmax(x, -1000.), 1000.);
  const double n = __builtin_rint(x * 9.23324826168936580e+01);           // 64 / ln 2
  double r = fma(-n, 6.93147180369123816490e-01 / 64, x);
  r = fma(-n, 1.90821492927058770002e-10 / 64, r);
  const int ni = (int)n;
  const double t = tab[ni & 63];
  double p = fma(r, 1. / 120., 1. / 24.);
  p = fma(p, r, 1. / 6.);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.);
  return __builtin_amdgcn_ldexp(fma(t, p * r, t), ni >> 6);
}
template <bool SK = PEAQ_SK_DEFAULT> __device__ __forceinline__ double exp_fast(double x) {
  x = fmin(fmax(x, -1000.), 1000.);
  const double n = __builtin_rint(x * PEAQ_KC(1.44269504088896338700e+00));
  double r = fma(-n, PEAQ_KC(6.93147180369123816490e-01), x);
  r = fma(-n, PEAQ_KC(1.90821492927058770002e-10), r);
  double p = PEAQ_KC(1. / 479001600.);
  p = PEAQ_FMA_KC(p, r, 1. / 39916800.);
  p = PEAQ_FMA_KC(p, r, 1. / 3628800.);
  p = PEAQ_FMA_KC(p, r, 1. / 362880.);
  p = PEAQ_FMA_KC(p, r, 1. / 40320.);
  p = PEAQ_FMA_KC(p, r, 1. / 5040.);
  p = PEAQ_FMA_KC(p, r, 1. / 720.);
  p = PEAQ_FMA_KC(p, r, 1. / 120.);
  p = PEAQ_FMA_KC(p, r, 1. / 24.);
  p = PEAQ_FMA_KC(p, r, 1. / 6.);
  p = fma(p, r, 0.5);
  p = fma(p, r, 1.);
  p = fma(p, r, 1.);
  return __builtin_amdgcn_ldexp(p, (int)n);
}

// N independent arguments in lockstep with the polynomial constants in scalar registers (the filter bank's FP64
// kernel: ten bands of one time point per lane).  The same operations in the same order as log_nonneg<true> and
// exp_fast<true> on each element, so the results are theirs bit for bit; what changes is that a constant is
// set up once per Horner step rather than once per step AND element (two s_mov_b32 each, and a wave issues
// scalar and vector instructions in order), and that the N chains are independent of each other.
template <int N> __device__ __forceinline__ void log_nonneg_n(double (&x)[N]) {
  double s[N], z[N], p[N], ef[N];
  const double rt = PEAQ_SC(0.70710678118654752440);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double m = __builtin_amdgcn_frexp_mant(x[i]);
    int e = __builtin_amdgcn_frexp_exp(x[i]);
    const bool low = m < rt;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double num = m - 1., den = m + 1.;
    double r = __builtin_amdgcn_rcp(den);
    r = fma(fma(-den, r, 1.), r, r);
    double q = num * r;
    q = fma(fma(-den, q, num), r, q);
    s[i] = q;
    z[i] = q * q;
    ef[i] = (double)e;
  }
  {
    const double c = PEAQ_SC(1. / 19);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = c;
  }
#define PEAQ_STEP(v, cc)                                          \
  {                                                               \
    const double c = PEAQ_SC(cc);                                 \
    _Pragma("unroll") for (int i = 0; i < N; ++i) p[i] = fma_sgpr(p[i], v[i], c); \
  }
  PEAQ_STEP(z, 1. / 17) PEAQ_STEP(z, 1. / 15) PEAQ_STEP(z, 1. / 13) PEAQ_STEP(z, 1. / 11)
  PEAQ_STEP(z, 1. / 9) PEAQ_STEP(z, 1. / 7) PEAQ_STEP(z, 1. / 5) PEAQ_STEP(z, 1. / 3)
  const double ln2h = PEAQ_SC(6.93147180369123816490e-01), ln2l = PEAQ_SC(1.90821492927058770002e-10);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double t = s[i] + s[i];
    const double lm = fma(t * z[i], p[i], t);
    const double l = fma(ef[i], ln2h, fma(ef[i], ln2l, lm));
    x[i] = x[i] == 0. ? -__builtin_inf() : (x[i] == __builtin_inf() ? __builtin_inf() : l);
  }
}
template <int N> __device__ __forceinline__ void exp_fast_n(double (&x)[N]) {
  double r[N], p[N], n[N];
  const double il2 = PEAQ_SC(1.44269504088896338700e+00);
  const double ln2h = PEAQ_SC(6.93147180369123816490e-01), ln2l = PEAQ_SC(1.90821492927058770002e-10);
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const double xx = fmin(fmax(x[i], -1000.), 1000.);
    n[i] = __builtin_rint(xx * il2);
    r[i] = fma(-n[i], ln2l, fma(-n[i], ln2h, xx));
  }
  {
    const double c = PEAQ_SC(1. / 479001600.);
#pragma unroll
    for (int i = 0; i < N; ++i) p[i] = c;
  }
  PEAQ_STEP(r, 1. / 39916800.) PEAQ_STEP(r, 1. / 3628800.) PEAQ_STEP(r, 1. / 362880.) PEAQ_STEP(r, 1. / 40320.)
  PEAQ_STEP(r, 1. / 5040.) PEAQ_STEP(r, 1. / 720.) PEAQ_STEP(r, 1. / 120.) PEAQ_STEP(r, 1. / 24.) PEAQ_STEP(r, 1. / 6.)
#undef PEAQ_STEP
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double q = fma(p[i], r[i], 0.5);
    q = fma(q, r[i], 1.);
    q = fma(q, r[i], 1.);
    x[i] = __builtin_amdgcn_ldexp(q, (int)n[i]);
  }
}

// a / b for finite b of moderate magnitude (no scaling against overflow / underflow of the
// reciprocal, no special cases): reciprocal with ONE Newton step (2^-26 -> 2^-52), quotient with one
// residual correction, whose error is the product of the quotient's and the reciprocal's: <= 1 ulp
// (tools/check_math.hip).  6 instructions instead of the 11 of the IEEE sequence.
__device__ __forceinline__ double div_fast(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  r = fma(fma(-b, r, 1.), r, r);
  const double q = a * r;
  return fma(fma(-b, q, a), r, q);
}

// sqrt(x) for finite x >= 0 (0 -> 0), <= 1 ulp: reciprocal square root, ONE coupled Newton step
// on (sqrt, 1/(2 sqrt)), one residual correction (tools/check_math.hip).  11 instructions instead of 23.
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  g = fma(fma(-g, g, x), h, g);
  return x == 0. ? 0. : g;
}

// 1 / sqrt(x) for finite x > 0, <= 1 ulp or so; x = 0 gives NaN (inf * 0 in the Newton step), x < 0 NaN
__device__ __forceinline__ double rsqrt_pos(double x) {
  double r = __builtin_amdgcn_rsq(x);
  double e = fma(-x * r, r, 1.);                     // 1 - x r^2
  r = fma(r * e, fma(e, 0.375, 0.5), r);             // r (1 + e/2 + 3 e^2 / 8)
  e = fma(-x * r, r, 1.);
  return fma(r * e, 0.5, r);
}

constexpr double kInvLn10 = 0.43429448190325182765;   // log10 x = ln x / ln 10
constexpr double kLn2 = 0.69314718055994530942;       // 2^x = e^(x ln 2)

// x^y for x > 0 as exp(y ln x): relative error about |y ln x| ulp -- a few 1e-15 at most for the
// ranges of this model.
__device__ __forceinline__ double pow_pos(double x, double y) { return exp_fast(y * log_pos(x)); }

// Mixed-precision ledger (tools/precision_ledger.py): -DPEAQ_LEDGER_FP32_BACKEND builds
// evaluate the back end's loudness / detection transcendentals in FP32 -- an EXPERIMENT to price the
// precision, never the product.
#ifdef PEAQ_LEDGER_FP32_BACKEND
__device__ __forceinline__ double be_pow(double x, double y) { return (double)__powf((float)x, (float)y); }
__device__ __forceinline__ double be_log(double x) { return (double)__logf((float)x); }
__device__ __forceinline__ double be_exp(double x) { return (double)__expf((float)x); }
#else
__device__ __forceinline__ double be_pow(double x, double y) { return pow_pos(x, y); }
__device__ __forceinline__ double be_log(double x) { return log_pos(x); }
__device__ __forceinline__ double be_exp(double x) { return exp_fast(x); }
#endif

// LDS traffic of ONE wave is executed in program order by the hardware; this
// only stops the compiler from moving LDS accesses across the point.
__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

}  // namespace peaq
