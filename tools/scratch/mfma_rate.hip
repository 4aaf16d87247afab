#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(double* out, int iters, double a, double b) {
  v4d c0 = {0,0,0,0}, c1 = c0, c2 = c0, c3 = c0;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4=0,s5=0,s6=0,s7=0;
  a += threadIdx.x; b += threadIdx.x * 0.5;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    } else {
      s0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s1, 0, 0, 0);
      s2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s2, 0, 0, 0);
      s3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s3, 0, 0, 0);
      s4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s4, 0, 0, 0);
      s5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s5, 0, 0, 0);
      s6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s6, 0, 0, 0);
      s7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, s7, 0, 0, 0);
      if (MODE == 2) {   // vector FP64 adds in between (operand construction)
        a = a + b; b = b - 1e-9;
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + s0 + s1 + s2 + s3+s4+s5+s6+s7;
}
template <int MODE> void run(const char* name, int waves_per_simd) {
  double* d; hipMalloc(&d, 256 * 1024 * 8 * 8);
  const int iters = 20000, blocks = 256 * waves_per_simd, threads = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, threads>>>(d, 100, 1.0, 2.0);
  hipEventRecord(e0); k<MODE><<<blocks, threads>>>(d, iters, 1.0, 2.0); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (MODE == 0 ? 4.0 : 8.0) * iters * waves_per_simd;   // MFMAs per SIMD
  const double fma = (MODE == 0 ? 1024.0 : 256.0);
  printf("%s waves/SIMD %d: %.3f ms, %.1f cycles per MFMA at 2.4 GHz, %.1f TFLOP/s\n", name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / n,
         n * 1024 * fma * 2 / (ms * 1e-3) / 1e12);
}
int main() {
  for (int w = 1; w <= 2; ++w) { run<0>("16x16x4", w); run<1>("4x4x4_4b", w); run<2>("4x4x4_4b + 2 fp64 valu per 8", w); }
}
