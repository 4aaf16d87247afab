// experiment: range checking of raw buffer loads on gfx950 (is a dwordx4 checked per dword?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* p, float* out, int valid_bytes, int shift_bytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p + shift_bytes), 0, valid_bytes, 0x00020000);
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, threadIdx.x * 16, 0, 0);
  out[4 * threadIdx.x + 0] = __uint_as_float(v.x);
  out[4 * threadIdx.x + 1] = __uint_as_float(v.y);
  out[4 * threadIdx.x + 2] = __uint_as_float(v.z);
  out[4 * threadIdx.x + 3] = __uint_as_float(v.w);
}
int main() {
  float h[512], *d, *o, ho[256];
  for (int i = 0; i < 512; ++i) h[i] = 1000.f + i;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  const int cases[][2] = {{100, 0}, {104, 0}, {96, 0}, {100, 4}, {40, 8}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, c[0], c[1]);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    printf("valid %d bytes, base shifted %d:", c[0], c[1]);
    for (int i = 16; i < 32; ++i) printf(" %g", ho[i]);
    printf("\n");
  }
  return 0;
}
