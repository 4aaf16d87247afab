// micro-benchmark: what does s_memtime count, and how fast does the shader clock run under an FP64 load?
// Each wave runs a chain of FP64 FMAs (4 independent chains) and reads s_memtime (shader clock) and
// wall_clock64 (constant 100 MHz) around it.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, unsigned long long* t, int iters) {
  double a = threadIdx.x * 1e-3, b = 1.0000001, c = 1e-9, d = a + 1, e = a + 2, f = a + 3;
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f;
  if (threadIdx.x == 0) { t[2 * blockIdx.x] = c1 - c0; t[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
  for (int blocks : {1, 256, 256 * 8, 256 * 12}) {
    const int iters = 200000;
    double* o; unsigned long long* t;
    hipMalloc(&o, blocks * 64 * 8); hipMalloc(&t, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, o, t, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, o, t, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    // 64 FMAs per iteration per wave; one wave per workgroup
    printf("%5d waves: %.2f ms, s_memtime %.1f M ticks = %.0f MHz by wall_clock64 (100 MHz), %.0f MHz by events; "
           "FMA issue %.2f ticks each\n", blocks, ms, h[0] / 1e6, h[0] * 100.0 / h[1], h[0] / (ms * 1e3),
           (double)h[0] / (64.0 * iters));
  }
  return 0;
}
