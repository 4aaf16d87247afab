// lane layout of v_mfma_f64_4x4x4_4b_f64 by one-hot probing
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
      out[(la * 64 + lb) * 64 + lane] = d;
    }
}
int main() {
  double* d; (void)hipMalloc(&d, 64 * 64 * 64 * 8);
  probe<<<1, 64>>>(d);
  std::vector<double> h(64 * 64 * 64);
  (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      for (int l = 0; l < 64; ++l)
        if (h[(la * 64 + lb) * 64 + l] != 0.) printf(" B%d->D%d", lb, l);
    printf("\n");
  }
}
