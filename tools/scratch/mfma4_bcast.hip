// does v_mfma_f64_4x4x4_4b_f64 honour CBSZ / ABID (broadcast of one A block to all four)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int ABID>
__global__ void probe(double* out) {
  const int lane = threadIdx.x;
  // A[i][k] of block B in lane 16 k + 4 B + i: value encodes (block, i, k); B = identity-like: B[k][j] = (k == j) in every block
  const int k = lane >> 4, blk = (lane >> 2) & 3, i = lane & 3;
  const double a = 1000. * blk + 10. * i + k + 0.5;
  const double b = (k == (lane & 3)) ? 1. : 0.;           // B[k][j], j = lane & 3
  out[lane] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 2, ABID, 0);
}
int main() {
  double* d; (void)hipMalloc(&d, 64 * 8);
  std::vector<double> h(64);
  for (int abid = 0; abid < 4; ++abid) {
    if (abid == 0) probe<0><<<1, 64>>>(d); else if (abid == 1) probe<1><<<1, 64>>>(d); else if (abid == 2) probe<2><<<1, 64>>>(d); else probe<3><<<1, 64>>>(d);
    (void)hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
    // D[i][j] of block B in lane 16 i + 4 B + j = A[i][j] (B = identity): expect 1000 abid + 10 i + j + 0.5 in ALL blocks
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
      const int i = l >> 4, j = l & 3;
      if (h[l] != 1000. * abid + 10. * i + j + 0.5) ok = 0;
    }
    printf("cbsz 2 abid %d: %s  (lane 0..7: %g %g %g %g %g %g %g %g)\n", abid, ok ? "A block broadcast to all blocks" : "NOT a broadcast", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
  }
}
