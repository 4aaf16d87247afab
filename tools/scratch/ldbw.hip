// micro-benchmark: what does a 64-lane dwordx4 load cost a CU that runs 12 such waves?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(128) void k(const float* p, float* out, int iters, size_t region_floats, size_t total_floats) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 2 + (threadIdx.x >> 6);
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    size_t off;
    if (MODE == 0) off = 0;                                           // everybody reads the same 16 KB (L1 hot)
    else if (MODE == 1) off = (wave * region_floats) % total_floats;  // every wave its own 16 KB, re-read (L2 hot)
    else off = ((wave * iters + it) * region_floats) % total_floats;  // streaming
    const float* b = p + off;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, 0x7fffffff, 0x00020000);
    u32x4 v[16];
    int lo = lane * 16;
    asm volatile("" : "+v"(lo));
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(r, lo + i * 1024, 0, 0);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += __uint_as_float(v[i].x) + __uint_as_float(v[i].w);
  }
  out[blockIdx.x * 128 + threadIdx.x] = acc;
}
int main() {
  const size_t total = (size_t)1 << 28;   // 1 GiB of floats
  float *d, *o;
  hipMalloc(&d, total * 4); hipMemset(d, 0, total * 4);
  const int blocks = 256 * 6;
  hipMalloc(&o, blocks * 128 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    const int iters = 200;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(128), 0, 0, d, o, iters, (size_t)4096, total);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(128), 0, 0, d, o, iters, (size_t)4096, total);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(128), 0, 0, d, o, iters, (size_t)4096, total);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 2 * iters * 16384.;
    printf("mode %d: %.3f ms, %.1f GB/s, %.1f B/clk/CU @2.4GHz, %.0f cycles per wave-load (12 waves/CU)\n", mode, ms, bytes / ms / 1e6,
           bytes / (ms * 1e-3) / 256 / 2.4e9, ms * 1e-3 * 2.4e9 / (iters * 16.0 * (blocks * 2 / 256.0 / 12.0)) / 12.0 * 12.0 / 1.0);
  }
  return 0;
}
