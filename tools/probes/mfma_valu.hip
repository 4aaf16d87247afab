// micro-benchmark: do matrix (MFMA) and vector (FP64 FMA) instructions of DIFFERENT waves on one SIMD overlap?
// Workgroups of 8 waves (2 per SIMD).  mode 0: all waves MFMA; 1: all waves FP64 FMA; 2: waves 0-3 MFMA, 4-7 FMA
// (one of each per SIMD); 3: only waves 0-3 work (MFMA), 4-7 exit; 4: only waves 0-3 work (FMA).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <typename VT>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
  const int wv = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || mode == 3 || (mode == 2 && wv < 4);
  const bool do_fma = mode == 1 || mode == 4 || (mode == 2 && wv >= 4);
  if ((mode == 3 || mode == 4) && wv >= 4) return;
  float r = 0.f;
  if (do_mfma) {
    v4f a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0;
    const float x = threadIdx.x * 1e-3f, y = 1.0001f;
    for (int i = 0; i < iters; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
      a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a4, 0, 0, 0);
      a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a5, 0, 0, 0);
      a6 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a6, 0, 0, 0);
      a7 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a7, 0, 0, 0);
    }
    r = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1] + a6[2] + a7[3];
  }
  if (do_fma) {
    VT a = (VT)(threadIdx.x * 1e-3 + 1), b = (VT)1.0000001, c = (VT)1e-9, d = a + 1, e = a + 2, f = a + 3, g = a + 4, h = a + 5, p = a + 6, q = a + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {                  // 64 instructions per iteration = 256 cycles = the 8 MFMAs' 256
        if constexpr (sizeof(VT) == 4 && !__is_floating_point(VT)) {   // xorshift steps: no closed form for the compiler
          a ^= a << 13; d ^= d >> 17; e ^= e << 5; f ^= f << 13; g ^= g >> 17; h ^= h << 5; p ^= p << 13; q ^= q >> 17;
        } else {
          a = a * b + c; d = d * b + c; e = e * b + c; f = f * b + c;
          g = g * b + c; h = h * b + c; p = p * b + c; q = q * b + c;
        }
      }
    }
    r = (float)(a + d + e + f + g + h + p + q);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
  const int blocks = 256, iters = 100000;
  float* o; hipMalloc(&o, blocks * 512 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"2 MFMA waves per SIMD", "2 vector waves per SIMD", "1 MFMA + 1 vector wave per SIMD",
                         "1 MFMA wave per SIMD", "1 vector wave per SIMD"};
  for (int ty = 0; ty < 3; ++ty) {
    printf("vector instruction: %s\n", ty == 0 ? "v_fma_f64" : ty == 1 ? "v_fma_f32" : "v_lshl/v_xor (integer, 2 per step)");
    for (int mode = 0; mode < 5; ++mode) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (ty == 0) hipLaunchKernelGGL(k<double>, dim3(blocks), dim3(512), 0, 0, o, iters, mode);
        if (ty == 1) hipLaunchKernelGGL(k<float>, dim3(blocks), dim3(512), 0, 0, o, iters, mode);
        if (ty == 2) hipLaunchKernelGGL(k<unsigned>, dim3(blocks), dim3(512), 0, 0, o, iters, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      printf("  %-36s %.2f ms\n", names[mode], ms);
    }
  }
  return 0;
}
