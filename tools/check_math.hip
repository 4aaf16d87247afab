// tools/check_math.hip -- accuracy of log_pos / log_nonneg / exp_fast / pow_pos (peaq_wave.h)
// against OCML's correctly rounded log / exp / pow, measured ON the GPU.
//   hipcc -O3 --offload-arch=gfx950 -I gstpeaq_amd/csrc -I include tools/check_math.hip -x hip gstpeaq_amd/csrc/peaq_tables.cpp \
//         -o tools/check_math && tools/check_math
// (log_tab reads the engine's own table, built by peaq_tables.cpp)
// prints the worst error in ulp over 2^24 arguments per function (log-uniform over the ranges the
// model produces and beyond) and checks the special values.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "peaq_tables.h"
#include "peaq_wave.h"

using namespace peaq;

__device__ double ulp_err(double got, double want) {
  if (got == want) return 0.;
  if (isnan(got) || isnan(want) || isinf(got) || isinf(want)) return 1e300;
  int e;
  frexp(want, &e);
  return fabs(got - want) / ldexp(1., e - 53 < -1074 ? -1074 : e - 53);   // subnormal results: ulp = 2^-1074
}
__device__ double u01(uint64_t i, uint64_t salt) {   // splitmix64 -> [0, 1)
  uint64_t z = (i + salt) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (1. / 9007199254740992.);
}
__global__ void sweep(double* worst, const CommonTables* ct) {   // worst[0] log, [1] exp, [2] pow, [3] log near 1, [4] div, [5] sqrt, [6] log_tab, [7] log_tab near 1
  __shared__ __attribute__((aligned(16))) double ltab[2 * kLogTabEntries + 2];
  for (int k = threadIdx.x; k < 2 * kLogTabEntries; k += blockDim.x) ltab[k] = ct->log_tab[k >> 1][k & 1];
  __syncthreads();
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  double w0, w1, w2, w3, w6, w7;
  {
    const double x = exp2(u01(i, 1) * 2098. - 1074.);               // every finite positive magnitude
    w0 = ulp_err(log_pos(x), log(x));
    w6 = ulp_err(log_tab(x, ltab), log(x));
    const double y = 1. + (u01(i, 2) - 0.5) * exp2(-u01(i, 3) * 50.);   // around 1, down to 1 +- 2^-51
    w3 = ulp_err(log_pos(y), log(y));
    w7 = ulp_err(log_tab(y, ltab), log(y));
  }
  {
    const double x = (u01(i, 4) - 0.5) * 1480.;                     // the whole finite range of exp
    w1 = ulp_err(exp_fast(x), exp(x));
  }
  {
    const double x = exp2(u01(i, 5) * 120. - 60.), y = u01(i, 6) * 3.;   // 1e-18 .. 1e18, exponents 0 .. 3
    w2 = ulp_err(pow_pos(x, y), pow(x, y));
  }
  double w4, w5;
  {
    const double a = (u01(i, 7) - 0.5) * exp2(u01(i, 8) * 200. - 100.), b = exp2(u01(i, 9) * 400. - 200.) * (u01(i, 10) < 0.5 ? -1. : 1.);
    w4 = ulp_err(div_fast(a, b), a / b);
    const double x = exp2(u01(i, 11) * 1200. - 600.);
    w5 = ulp_err(sqrt_pos(x), sqrt(x));
  }
  double w[8] = {w0, w1, w2, w3, w4, w5, w6, w7};
  for (int k = 0; k < 8; ++k) {
    double v = w[k];
    for (int d = 32; d >= 1; d >>= 1) v = fmax(v, __shfl_xor(v, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned long long*>(worst + k), __double_as_longlong(v));
  }
}
__global__ void specials(double* out, const CommonTables* ct) {
  __shared__ __attribute__((aligned(16))) double ltab[2 * kLogTabEntries + 2];
  for (int k = 0; k < 2 * kLogTabEntries; ++k) ltab[k] = ct->log_tab[k >> 1][k & 1];
  out[10] = log_tab(1., ltab);
  out[11] = log_tab_nonneg(0., ltab);
  out[12] = log_tab(4.9406564584124654e-324, ltab);
  out[0] = log_nonneg(0.);
  out[1] = log_nonneg(__builtin_inf());
  out[2] = log_nonneg(__builtin_nan(""));
  out[3] = log_nonneg(4.9406564584124654e-324);
  out[4] = exp_fast(-__builtin_inf());
  out[5] = exp_fast(-800.);
  out[6] = exp_fast(800.);
  out[7] = exp_fast(0.);
  out[8] = log_pos(1.);
  out[9] = sqrt_pos(0.);
}
int main() {
  double *d_w, *d_s, w[8], s[13];
  CommonTables* d_ct;
  static CommonTables h_ct;
  build_common_tables(h_ct);
  if (hipMalloc(&d_w, sizeof w) != hipSuccess || hipMalloc(&d_s, sizeof s) != hipSuccess ||
      hipMalloc(&d_ct, sizeof h_ct) != hipSuccess || hipMemcpy(d_ct, &h_ct, sizeof h_ct, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(d_w, 0, sizeof w) != hipSuccess)
    return 2;
  hipLaunchKernelGGL(sweep, dim3(1 << 16), dim3(256), 0, 0, d_w, d_ct);
  hipLaunchKernelGGL(specials, dim3(1), dim3(1), 0, 0, d_s, d_ct);
  if (hipMemcpy(w, d_w, sizeof w, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(s, d_s, sizeof s, hipMemcpyDeviceToHost) != hipSuccess)
    return 2;
  printf("{\"n_per_function\": %d, \"log_pos_max_ulp\": %.3f, \"log_pos_near_1_max_ulp\": %.3f, \"exp_fast_max_ulp\": %.3f, "
         "\"pow_pos_max_ulp\": %.3f, \"div_fast_max_ulp\": %.3f, \"sqrt_pos_max_ulp\": %.3f, \"log_tab_max_ulp\": %.3f, "
         "\"log_tab_near_1_max_ulp\": %.3f, \"log_tab(1)\": %g, \"log_tab_nonneg(0)\": \"%g\", \"log_tab(denorm_min)\": %.17g, ",
         1 << 24, w[0], w[3], w[1], w[2], w[4], w[5], w[6], w[7], s[10], s[11], s[12]);
  printf("\"log_nonneg(0)\": \"%g\", \"log_nonneg(inf)\": \"%g\", \"log_nonneg(nan)\": \"%g\", \"log_nonneg(denorm_min)\": %.17g, "
         "\"exp_fast(-inf)\": %g, \"exp_fast(-800)\": %g, \"exp_fast(800)\": \"%g\", \"exp_fast(0)\": %.17g, \"log_pos(1)\": %g}\n",
         s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7], s[8]);
  const bool ok = w[0] < 2.5 && w[3] < 2.5 && w[1] < 2.5 && std::isinf(s[0]) && s[0] < 0 && std::isinf(s[1]) && std::isnan(s[2]) &&
                  s[4] == 0. && s[5] == 0. && std::isinf(s[6]) && s[7] == 1. && s[8] == 0. && s[9] == 0. && w[4] < 2.5 && w[5] < 2.5 && w[6] < 2.5 && w[7] < 5.5 && s[10] == 0. &&
                  std::isinf(s[11]) && s[11] < 0;
  return ok ? 0 : 1;
}
