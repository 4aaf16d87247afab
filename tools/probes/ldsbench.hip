// micro-benchmark: LDS instruction costs with 12 waves per CU (6 workgroups of 2 waves, 20 KB LDS each)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(128, 3) void k(double* out, int iters) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double* u = lds + w * 1280;
  for (int i = lane; i < 1280; i += 64) u[i] = 0.;
  double v = lane * 1e-3, acc = 0.;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (MODE == 0) atomicAdd(&u[128 * (s & 1) + lane + (s >> 1)], v);            // ds_add_f64, conflict free
      if (MODE == 1) u[128 * (s & 1) + lane + (s >> 1)] = v;                         // ds_write_b64
      if (MODE == 2) acc += u[128 * (s & 1) + lane + (s >> 1)];                      // ds_read_b64
      if (MODE == 3) acc += __shfl(v, (lane * 5 + s) & 63, 64);                      // 2 x ds_bpermute_b32
      if (MODE == 4) { u[(lane * 17 + s * 67) & 1023] = v; }                         // ds_write_b64, scattered
      if (MODE == 5) acc += u[(lane * 17 + s * 67 + it) & 1023];                     // ds_read_b64, scattered
      if (MODE == 6) atomicAdd(reinterpret_cast<float*>(u) + 128 * (s & 1) + lane + (s >> 1), (float)v);   // ds_add_f32
      if (MODE == 7) atomicAdd(reinterpret_cast<float*>(u) + 128 * (s & 1) + lane + (s >> 1), (float)v * 1e-42f);   // ... denormal values
      if (MODE == 8) acc += reinterpret_cast<float*>(u)[128 * (s & 1) + lane + (s >> 1)];            // ds_read_b32
      v += 1e-9;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  out[blockIdx.x * 128 + threadIdx.x] = acc + u[lane] + v;
}
int main() {
  const int blocks = 256 * 6, iters = 2000;
  double* o; hipMalloc(&o, blocks * 128 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"ds_add_f64", "ds_write_b64", "ds_read_b64", "shfl(double)=2 bpermute", "ds_write_b64 scattered", "ds_read_b64 scattered", "ds_add_f32", "ds_add_f32 denormal", "ds_read_b32"};
  for (int mode = 0; mode < 9; ++mode) {
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      const size_t lds = 2 * 1280 * 8;
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(128), lds, 0, o, iters);
      if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(128), lds, 0, o, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // per CU: 12 waves x iters x 16 instructions
    printf("%-28s %.3f ms -> %.1f CU-cycles per wave-instruction (12 waves/CU, 2.4 GHz)\n", names[mode], ms,
           ms * 1e-3 * 2.4e9 / (12.0 * iters * 16));
  }
  return 0;
}
