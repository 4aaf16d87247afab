// cycles per v_mfma_f64_4x4x4 in the step pattern of the FIR loop: G groups x 4 instructions (2 A operands, 4 B
// operands shared by all groups), 4 FP64 adds per step
#include <hip/hip_runtime.h>
#include <cstdio>
template <int G, int ADDS>
__global__ void k(double* out, int iters, double a, double b) {
  double acc[G][4];
  double c[G][2];
  for (int g = 0; g < G; ++g) { for (int q = 0; q < 4; ++q) acc[g][q] = 0.; c[g][0] = a + g + threadIdx.x; c[g][1] = a - g; }
  double x0 = b + threadIdx.x, x1 = b * 0.5, y0 = a * 0.25, y1 = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    double bp0 = x0, bm0 = y0, bp1 = x1, bm1 = y1;
    if (ADDS) { bp0 = x0 + y0; bm0 = x0 - y0; bp1 = x1 + y1; bm1 = x1 - y1; }
    asm volatile("" : "+v"(bp0), "+v"(bm0), "+v"(bp1), "+v"(bm1));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      acc[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(c[g][0], bp0, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(c[g][1], bm0, acc[g][1], 0, 0, 0);
      acc[g][2] = __builtin_amdgcn_mfma_f64_4x4x4f64(c[g][0], bp1, acc[g][2], 0, 0, 0);
      acc[g][3] = __builtin_amdgcn_mfma_f64_4x4x4f64(c[g][1], bm1, acc[g][3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    x0 += 1.; // keep the sums live
  }
  double s = 0;
  for (int g = 0; g < G; ++g) for (int q = 0; q < 4; ++q) s += acc[g][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int G, int ADDS> void run(int wps) {
  double* d; (void)hipMalloc(&d, 256 * 1024 * 8 * 8);
  const int iters = 20000 / G, blocks = 256 * wps, threads = 256;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<G, ADDS><<<blocks, threads>>>(d, 100, 1.0, 2.0);
  (void)hipEventRecord(e0); k<G, ADDS><<<blocks, threads>>>(d, iters, 1.0, 2.0); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = 4.0 * G * iters * wps;
  printf("G %2d adds %d waves/SIMD %d: %.1f cycles per MFMA (SIMD), step %.0f cycles\n", G, ADDS, wps, ms * 1e-3 * 2.4e9 / n, ms * 1e-3 * 2.4e9 / (iters * wps));
  (void)hipFree(d);
}
int main() {
  run<1,0>(2); run<1,1>(2); run<2,0>(2); run<2,1>(2); run<3,1>(2); run<4,0>(2); run<4,1>(2); run<6,1>(2); run<10,0>(2); run<10,1>(2);
  run<1,1>(1); run<4,1>(1); run<10,1>(1);
}
