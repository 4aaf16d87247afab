// peaq_synth.hip -- include/peaq_synth.h evaluated on the device: fills HBM with
// the seeded synthetic (reference, test) pairs of the benchmark configurations
// (4096 pairs = 31.5 GB are generated where they are consumed).
// Workload utility, not part of the PEAQ algorithm.
#include <hip/hip_runtime.h>

#include "../../include/peaq_synth.h"
#include "peaq_kernels.h"

namespace peaq {

constexpr int kRun = 8;   // consecutive samples per thread (they share 62 of 63 noise taps)

__global__ __launch_bounds__(256) void synth_kernel(uint32_t seed0, int channels, uint32_t n_samples,
                                                    size_t pair_stride, float* __restrict__ ref,
                                                    float* __restrict__ test) {
  const unsigned pair = blockIdx.y;
  const int chan = blockIdx.z;
  const uint32_t n0 = (blockIdx.x * blockDim.x + threadIdx.x) * kRun;
  if (n0 >= n_samples) return;
  peaq_synth_params p;
  peaq_synth_init(&p, seed0 + pair);
  int32_t w[PEAQ_SYNTH_NTAPS + kRun - 1];          // w[j] = noise(n0 + kRun-1 - j)
#pragma unroll
  for (int j = 0; j < PEAQ_SYNTH_NTAPS + kRun - 1; ++j)
    w[j] = peaq_synth_noise(p.chan_key[chan], (int64_t)n0 + kRun - 1 - j);
  float* r = ref + ((size_t)pair * pair_stride) * channels + chan;
  float* t = test + ((size_t)pair * pair_stride) * channels + chan;
#pragma unroll
  for (int i = 0; i < kRun; ++i) {
    const uint32_t n = n0 + i;
    if (n < n_samples) {
      int32_t vr, vt;
      // taps of sample n: noise(n - k) = w[kRun-1-i + k]
      peaq_synth_sample_from_noise(&p, chan, n, n_samples, w + (kRun - 1 - i), &vr, &vt);
      r[(size_t)n * channels] = peaq_synth_to_float(vr);
      t[(size_t)n * channels] = peaq_synth_to_float(vt);
    }
  }
}

hipError_t launch_synth(uint32_t seed0, unsigned n_pairs, int channels, uint32_t n_samples, size_t pair_stride,
                        float* ref, float* test, hipStream_t stream) {
  if (n_pairs == 0 || n_samples == 0) return hipSuccess;
  const unsigned per_block = 256 * kRun;
  // gridDim.y is limited to 65535: loop over slabs of pairs
  for (unsigned p0 = 0; p0 < n_pairs; p0 += 32768) {
    const unsigned np = n_pairs - p0 < 32768 ? n_pairs - p0 : 32768;
    dim3 grid((n_samples + per_block - 1) / per_block, np, channels);
    hipLaunchKernelGGL(synth_kernel, grid, dim3(256), 0, stream, seed0 + p0, channels, n_samples, pair_stride,
                       ref + (size_t)p0 * pair_stride * channels, test + (size_t)p0 * pair_stride * channels);
  }
  return hipGetLastError();
}

}  // namespace peaq
