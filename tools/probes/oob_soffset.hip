// experiment: does the range check of a raw buffer load on gfx950 include the scalar offset?
// lane i loads the dword at voffset 4 i + soffset S from a buffer of `valid` bytes over h[k] = 1000 + k.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int S>
__global__ void k(const float* p, float* out, int valid_bytes) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, valid_bytes, 0x00020000);
  out[threadIdx.x] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, threadIdx.x * 4, S, 0));
}
int main() {
  float h[1024], *d, *o, ho[64];
  for (int i = 0; i < 1024; ++i) h[i] = 1000.f + i;
  hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof ho);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  for (int valid : {1024 + 100, 1024 + 16, 1024, 512}) {
    hipLaunchKernelGGL(k<1024>, dim3(1), dim3(64), 0, 0, d, o, valid);
    hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    printf("soffset 1024, valid %4d bytes: lanes 0..7 -> %g %g %g %g %g %g %g %g ... lane 24 %g lane 25 %g lane 26 %g\n", valid,
           ho[0], ho[1], ho[2], ho[3], ho[4], ho[5], ho[6], ho[7], ho[24], ho[25], ho[26]);
  }
  return 0;
}
