// peaq_fb.hip -- advanced mode: the 40-band filter-bank ear model
// (reference fbearmodel.c:276-435), split into two kernels:
//
//  fb_hp_kernel     one THREAD per (pair, channel, signal): playback-level scaling
//                   and the two cascaded DC-rejection biquads (fbearmodel.c:289-303)
//                   are a per-sample recurrence, so the parallel axis is the signal.
//                   The same sequential walk evaluates the data-boundary detector on
//                   the 192-sample block (gstpeaq.c:971-972,1081-1099) with the
//                   reference's float running sum, bit for bit.  Output: the filtered
//                   signal in FP64, one row per signal, staged through LDS so that the
//                   stores are 64-byte runs.
//  fb_bank_kernel   one WAVEFRONT per (pair, channel, signal), walking the chunk in
//                   tiles of 60 sub-samples (= 10 blocks of 192 samples): lanes are
//                   TIME points (every 32nd sample, fbearmodel.c:314), the filtered
//                   signal window sits in LDS, the 40 complex FIR responses come in as
//                   wave-uniform scalars.  Then, still with lanes = time: level
//                   dependent spreading (slope filter as a wave scan, :327-354),
//                   rectification (:357-360); the 11-tap backward-masking FIR at block
//                   rate (:364-382), internal noise and forward masking (:385-394)
//                   with lanes = bands.
#include <hip/hip_runtime.h>

#include "peaq_device.h"
#include "peaq_kernels.h"
#include "peaq_wave.h"

namespace peaq {

// BS.1387 Table 8 (fbearmodel.c:57-61)
__device__ constexpr int kLen[kFbBands] = {1456, 1438, 1406, 1362, 1308, 1244, 1176, 1104, 1030, 956, 884, 814, 748, 686,
                                           626,  570,  520,  472,  430,  390,  354,  320,  290,  262, 238, 214, 194, 176,
                                           158,  144,  130,  118,  106,  96,   86,   78,   70,   64,  58,  52};
constexpr int coef_offset(int b) {
  int o = 0;
  for (int i = 0; i < b; ++i) o += kLen[i] / 2 + 1;
  return o;
}

constexpr double kSlopeA = 0.993355506255034;      // fbearmodel.c:49
constexpr double kLnDist = -0.08137117849224008;   // ln(0.921851456499719), DIST of fbearmodel.c:50
constexpr double kCL = 0.0802581846102741;         // fbearmodel.c:51

// ---------------------------------------------------------------------------
// kernel 1: per-signal sample recurrences
// ---------------------------------------------------------------------------
struct HpWalk {
  double x1, x2, y1a, y2a, y1b, y2b;
  // Identical operation order to the reference, no FMA contraction.
  __device__ __forceinline__ double step(double in) {
#pragma clang fp contract(off)
    const double ya = in - 2. * x1 + x2 + 1.99517 * y1a - 0.995174 * y2a;
    const double yb = ya - 2. * y1a + y2a + 1.99799 * y1b - 0.997998 * y2b;
    x2 = x1;
    x1 = in;
    y2a = y1a;
    y1a = ya;
    y2b = y1b;
    y1b = yb;
    return yb;
  }
};

__global__ __launch_bounds__(64) void fb_hp_kernel(FbFrontArgs a, unsigned n_signals) {
  __shared__ double tile[64][9];                    // [signal in wave][8 samples], padded
  const int lane = threadIdx.x;
  const unsigned g0 = blockIdx.x * 64;
  const unsigned g = g0 + lane;
  const bool live = g < n_signals;
  const unsigned gg = live ? g : n_signals - 1;
  const int sig = gg & 1;
  const int chan = (gg >> 1) % a.channels;
  const unsigned pair = gg / (2 * a.channels);
  const unsigned n_sig = sig ? (a.n_test ? a.n_test[pair] : a.n_uniform_test)
                             : (a.n_ref ? a.n_ref[pair] : a.n_uniform_ref);
  const unsigned n_blocks = a.n_blocks ? a.n_blocks[pair] : a.n_blocks_uniform;
  const float* __restrict__ x = (sig ? a.test : a.ref) + (size_t)pair * a.pair_stride * a.channels + chan;
  const long long off = sig ? a.off_test : a.off_ref;
  const size_t row_len = a.hp_row_stride;
  double* __restrict__ rows = a.hp_scratch;
  double* __restrict__ my_row = rows + (size_t)gg * row_len;
  FbSignalState* __restrict__ st = a.fbstate + gg;

  // how many blocks of this launch exist for my pair
  unsigned nb_mine = 0;
  if (live && n_blocks > a.block0) nb_mine = min(a.blocks_per_launch, n_blocks - a.block0);
  unsigned nb_max = nb_mine;                         // wave-uniform loop bound
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) nb_max = max(nb_max, (unsigned)__shfl_xor((int)nb_max, d, 64));

  // history: the newest 1456 filtered samples of the previous launch sit at the row's tail
  if (nb_mine > 0) {
    if (a.first_launch) {
      for (int i = 0; i < kFbRing; ++i) my_row[i] = 0.;
    } else {
      const size_t tail = (size_t)a.prev_blocks * kFbFrame;
      for (int i = 0; i < kFbRing; ++i) my_row[i] = my_row[tail + i];
    }
  }
  HpWalk w{st->hp[0], st->hp[1], st->hp[2], st->hp[3], st->hp[4], st->hp[5]};

  for (unsigned bl = 0; bl < nb_max; ++bl) {
    const bool mine = bl < nb_mine;
    const long long s0 = (long long)(a.block0 + bl - a.block_origin) * kFbFrame + off;
    float sum = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;   // |x| of the last five samples
    int above = 0;
    for (int k0 = 0; k0 < kFbFrame; k0 += 8) {
      double y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const long long s = s0 + k;
        const float xv = (mine && s < (long long)n_sig) ? x[s * a.channels] : 0.f;   // zero padding: gstpeaq.c:733-738
        y[j] = w.step((double)xv * a.level_factor);
        // gstpeaq.c:1083-1096: FLOAT running sum, tested from i = 5 on
        const float ax = fabsf(xv);
        if (k < 5) {
          sum = (float)((double)sum + (double)ax);
        } else {
          sum = (float)((double)sum + ((double)ax - (double)h0));
          above |= ((double)sum >= 200. / 32768);
        }
        h0 = h1; h1 = h2; h2 = h3; h3 = h4; h4 = ax;
      }
      // 64 x 8 transpose through LDS: each store instruction writes 64-byte runs
#pragma unroll
      for (int j = 0; j < 8; ++j) tile[lane][j] = y[j];
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int src = r * 8 + (lane >> 3);
        const unsigned gs = g0 + src;
        const unsigned pair_s = gs / (2 * a.channels);
        const unsigned nbs = gs < n_signals ? (a.n_blocks ? a.n_blocks[pair_s] : a.n_blocks_uniform) : 0;
        if (gs < n_signals && a.block0 + bl < nbs)
          rows[(size_t)gs * row_len + kFbRing + (size_t)bl * kFbFrame + k0 + (lane & 7)] = tile[src][lane & 7];
      }
      __syncthreads();
    }
    if (mine && sig == 0)
      a.records[((size_t)(pair * a.blocks_per_launch + bl) * a.channels + chan) * kFbRecDoubles + kFbRecFlags] =
          (double)above;
  }
  if (nb_mine > 0) {
    st->hp[0] = w.x1; st->hp[1] = w.x2; st->hp[2] = w.y1a; st->hp[3] = w.y2a; st->hp[4] = w.y1b; st->hp[5] = w.y2b;
  }
}

// ---------------------------------------------------------------------------
// kernel 2: filter bank, spreading, masking
// ---------------------------------------------------------------------------
constexpr int kTileSub = 60;                        // sub-samples per tile (10 blocks)
constexpr int kTileBlocks = 10;
constexpr int kWin = (kTileSub - 1) * 32 + kFbRing + 1;   // 3345 filtered samples
constexpr int kWinCols = (kWin + 31) / 32;          // 105
constexpr int kWinRow = kWinCols + 1;               // row stride in doubles (106)

struct BankLds {
  double win[32 * kWinRow];        // window sample w at [(w & 31)][w >> 5]
  double e1[kFbBands][kTileBlocks];
  double hist[kFbBands][10];       // the 10 newest E0 values of the previous tile, oldest first
  double e0row[kTileSub + 10];
  double cu[kFbBands];
};

// one band's complex FIR at this lane's time point (fbearmodel.c:404-434)
template <int B>
__device__ __forceinline__ void fir_band(const double* __restrict__ win, int t, const FbTables* __restrict__ fb,
                                         double& re_out, double& im_out) {
  constexpr int N = kLen[B];
  constexpr int D = 1 + (kLen[0] - N) / 2;           // (31) in BS.1387
  constexpr int H = N / 2;
  const double* __restrict__ hr = fb->h_re + coef_offset(B);
  const double* __restrict__ hi = fb->h_im + coef_offset(B);
  // window index of the sample delayed by m: kFbRing + 32 t - m
  auto X = [&](int u) {                              // u = kFbRing - m (wave-uniform part)
    return win[(u & 31) * kWinRow + (u >> 5) + t];
  };
  double re = 0., im = 0.;
  int n = 1;
  if (B == 0) {
    // the reference's doubled ring buffer makes band 0's delay-1456 tap read the
    // NEWEST sample (fb_buf[offset + 1456] aliases fb_buf[offset]); reproduced
    const double x1 = X(kFbRing - (D + 1)), x2 = X(kFbRing - 0);
    re += (x1 + x2) * hr[1];
    im += (x1 - x2) * hi[1];
    n = 2;
  }
#pragma unroll 8
  for (; n < H; ++n) {
    const double x1 = X(kFbRing - (D + n)), x2 = X(kFbRing - (D + N - n));
    re += (x1 + x2) * hr[n];                         // even symmetry
    im += (x1 - x2) * hi[n];                         // odd symmetry
  }
  const double xm = X(kFbRing - (D + H));
  re_out = re + xm * hr[H];
  im_out = im + xm * hi[H];
}

template <int B>
struct FirAll {
  __device__ __forceinline__ static void run(const double* win, int t, const FbTables* fb, double (&re)[kFbBands],
                                             double (&im)[kFbBands]) {
    fir_band<B>(win, t, fb, re[B], im[B]);
    FirAll<B + 1>::run(win, t, fb, re, im);
  }
};
template <>
struct FirAll<kFbBands> {
  __device__ __forceinline__ static void run(const double*, int, const FbTables*, double (&)[kFbBands],
                                             double (&)[kFbBands]) {}
};

__global__ __launch_bounds__(64) void fb_bank_kernel(FbFrontArgs a, unsigned n_signals) {
  __shared__ BankLds sh;
  const int lane = threadIdx.x;
  const unsigned g = blockIdx.x;
  const int sig = g & 1;
  const int chan = (g >> 1) % a.channels;
  const unsigned pair = g / (2 * a.channels);
  const unsigned n_blocks = a.n_blocks ? a.n_blocks[pair] : a.n_blocks_uniform;
  if (n_blocks <= a.block0) return;
  const unsigned nb_mine = min(a.blocks_per_launch, n_blocks - a.block0);
  const BandTables* __restrict__ bt = a.bands;
  const FbTables* __restrict__ fb = a.fb;
  const size_t row_len = a.hp_row_stride;
  const size_t row_valid = (size_t)kFbRing + (size_t)a.blocks_per_launch * kFbFrame;
  const double* __restrict__ row = a.hp_scratch + (size_t)g * row_len;
  FbSignalState* __restrict__ st = a.fbstate + g;

  // recurrent state -> LDS / registers
  if (lane < kFbBands) {
    sh.cu[lane] = st->cu[lane];
#pragma unroll
    for (int i = 0; i < 10; ++i) sh.hist[lane][i] = st->e0_hist[lane][i];
  }
  double exc = lane < kFbBands ? st->excitation[lane] : 0.;
  // (1-A)^(t+1): decay of the slope-filter state that enters a tile
  double decay = 1. - kSlopeA;
  {
    double p = 1. - kSlopeA, acc = 1.;
    int e = lane + 1;
    while (e) {
      if (e & 1) acc *= p;
      p *= p;
      e >>= 1;
    }
    decay = acc;
  }
  wave_lds_fence();

  for (unsigned b0 = 0; b0 < nb_mine; b0 += kTileBlocks) {
    const unsigned nvb = min((unsigned)kTileBlocks, nb_mine - b0);   // valid blocks in this tile
    const int nvs = 6 * nvb;                                         // valid sub-samples
    // ---- window of the filtered signal: samples [192 b0 - 1456, 192 b0 + 59*32] ---------
    {
      const double* src = row + (size_t)b0 * kFbFrame;               // row index 0 = sample -1456 of the launch
      const int avail = (int)min((size_t)kWin, row_valid - (size_t)b0 * kFbFrame);
      for (int wdx = lane; wdx < kWin; wdx += 64)
        sh.win[(wdx & 31) * kWinRow + (wdx >> 5)] = wdx < avail ? src[wdx] : 0.;
    }
    wave_lds_fence();
    // ---- 40 complex FIR filters at sample 32 t (lane t) ----------------------------------
    double re[kFbBands], im[kFbBands];
    const int t = lane < kTileSub ? lane : kTileSub - 1;
    FirAll<0>::run(sh.win, t, fb, re, im);

    // ---- spreading (fbearmodel.c:327-354); bands in DESCENDING order so that the
    // accumulation can run in place: band b only receives from bands below it -------------
#pragma unroll
    for (int b = kFbBands - 1; b >= 0; --b) {
      const double level = 10. * log10(re[b] * re[b] + im[b] * im[b]);
      const double slope = fmax(4., 24. + 230. / bt->fc[b] - 0.2 * level);
      const double dist_s = exp(slope * kLnDist);                    // pow(DIST, s)
      // cu[t] = cu[t-1] + A (dist_s[t] - cu[t-1]): inclusive scan over the time lanes
      double v = kSlopeA * dist_s, m = 1. - kSlopeA;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d, 64);
        if (lane >= d) v += m * o;
        m *= m;
      }
      const double cu = v + decay * sh.cu[b];
      const double carry = __shfl(cu, nvs - 1, 64);
      wave_lds_fence();
      if (lane == 0) sh.cu[b] = carry;
      if (b < kFbBands - 1) {
        double d1 = re[b], d2 = im[b];
#pragma unroll
        for (int j = b + 1; j < kFbBands; ++j) {
          d1 *= cu;
          d2 *= cu;
          re[j] += d1;
          im[j] += d2;
        }
      }
    }
#pragma unroll
    for (int b = kFbBands - 1; b > 0; --b) {
      re[b - 1] += kCL * re[b];
      im[b - 1] += kCL * im[b];
    }
    // ---- rectification + backward masking (fbearmodel.c:357-382) ---------------------------
#pragma unroll
    for (int b = 0; b < kFbBands; ++b) {
      const double e0 = re[b] * re[b] + im[b] * im[b];
      wave_lds_fence();
      if (lane < 10) sh.e0row[lane] = sh.hist[b][lane];
      if (lane < kTileSub) sh.e0row[10 + lane] = e0;
      wave_lds_fence();
      if (lane < kTileBlocks) {
        // newest sample of block `lane` is sub-sample 6 lane + 5 -> e0row[6 lane + 15]
        const double* p = sh.e0row + 6 * lane + 15;
        double e1 = 0.;
#pragma unroll
        for (int i = 0; i < 5; ++i) e1 += (p[-i] + p[-(10 - i)]) * fb->back_mask[i];
        e1 += p[-5] * fb->back_mask[5];
        sh.e1[b][lane] = e1;
      }
      // history for the next tile: the 10 newest VALID sub-samples, oldest first
      if (lane < 10) sh.hist[b][lane] = sh.e0row[nvs + lane];
    }
    wave_lds_fence();
    // ---- internal noise + forward masking (fbearmodel.c:385-394), lanes = bands -------------
    if (lane < kFbBands) {
      const double noise = bt->internal_noise[lane], ac = bt->ear_tc[lane];
      for (unsigned bl = 0; bl < nvb; ++bl) {
        const double unsm = sh.e1[lane][bl] + noise;
        exc = ac * exc + (1. - ac) * unsm;
        double* rec = a.records + ((size_t)(pair * a.blocks_per_launch + b0 + bl) * a.channels + chan) * kFbRecDoubles;
        rec[(sig ? kFbRecUnsmTest : kFbRecUnsmRef) + lane] = unsm;
        rec[(sig ? kFbRecExcTest : kFbRecExcRef) + lane] = exc;
      }
    }
    wave_lds_fence();
  }
  if (lane < kFbBands) {
    st->cu[lane] = sh.cu[lane];
#pragma unroll
    for (int i = 0; i < 10; ++i) st->e0_hist[lane][i] = sh.hist[lane][i];
    st->excitation[lane] = exc;
  }
}

hipError_t launch_fb_frontend(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const unsigned n_signals = n_pairs * a.channels * 2;
  if (n_signals == 0 || a.blocks_per_launch == 0) return hipSuccess;
  hipLaunchKernelGGL(fb_hp_kernel, dim3((n_signals + 63) / 64), dim3(64), 0, stream, a, n_signals);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fb_bank_kernel, dim3(n_signals), dim3(64), 0, stream, a, n_signals);
  return hipGetLastError();
}

}  // namespace peaq
