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
      const double* __restrict__ prev_row = a.hp_prev ? a.hp_prev + (size_t)gg * row_len : my_row;
      for (int i = 0; i < kFbRing; ++i) my_row[i] = prev_row[tail + i];
    }
  }
  HpWalk w{st->hp[0], st->hp[1], st->hp[2], st->hp[3], st->hp[4], st->hp[5]};

  // samples are fetched 16 ahead of their use: the walk itself is a chain of dependent FP64
  // operations, the loads (one cache line per lane) must never be waited for
  auto fetch = [&](unsigned bl, int k, bool mine) -> float {
    const long long s = (long long)(a.block0 + bl - a.block_origin) * kFbFrame + off + k;
    return (mine && s < (long long)n_sig) ? x[s * a.channels] : 0.f;   // zero padding: gstpeaq.c:733-738
  };
  float xq[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) xq[j] = fetch(0, j, 0 < nb_mine);
  for (unsigned bl = 0; bl < nb_max; ++bl) {
    const bool mine = bl < nb_mine;
    float sum = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f, h3 = 0.f, h4 = 0.f;   // |x| of the last five samples
    int above = 0;
    for (int k0 = 0; k0 < kFbFrame; k0 += 8) {
      double y[8];
      float xin[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xin[j] = xq[j];
        xq[j] = xq[j + 8];
      }
      {                                              // refill: samples k0+16 .. k0+23 (may be the next block)
        const int kn = k0 + 16;
        const unsigned bln = kn < kFbFrame ? bl : bl + 1;
        const int kk = kn < kFbFrame ? kn : kn - kFbFrame;
#pragma unroll
        for (int j = 0; j < 8; ++j) xq[8 + j] = fetch(bln, kk + j, bln < nb_mine);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        const float xv = xin[j];
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
      // the workgroup is ONE wave: LDS is in program order, no barrier needed -- and
      // __syncthreads() would drain the prefetched loads and the stores (vmcnt(0)) every 8 samples
      wave_lds_fence();
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int src = r * 8 + (lane >> 3);
        const unsigned gs = g0 + src;
        const unsigned pair_s = gs / (2 * a.channels);
        const unsigned nbs = gs < n_signals ? (a.n_blocks ? a.n_blocks[pair_s] : a.n_blocks_uniform) : 0;
        if (gs < n_signals && a.block0 + bl < nbs)
          rows[(size_t)gs * row_len + kFbRing + (size_t)bl * kFbFrame + k0 + (lane & 7)] = tile[src][lane & 7];
      }
      wave_lds_fence();
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
// kernel 2: filter bank, spreading, masking.  One workgroup of four waves per
// signal; a tile is 60 sub-samples (10 blocks).  Phase 1 (the bulk of the work):
// every wave evaluates 10 of the 40 complex FIR filters for all 60 time points
// (lane = time point) from the shared window in LDS; the bands are dealt out so
// that the four waves carry the same number of taps.  Phases 2..5 turn the 40 x 60
// filter outputs into excitation patterns.
// ---------------------------------------------------------------------------
constexpr int kTileSub = 60;                        // sub-samples per tile (10 blocks)
constexpr int kTileBlocks = 10;
constexpr int kWin = (kTileSub - 1) * 32 + kFbRing + 1;   // 3345 filtered samples
constexpr int kWinCols = (kWin + 31) / 32;          // 105
constexpr int kWinRow = kWinCols + 1;               // row stride in doubles (106)
constexpr int kACols = 64;                          // A[band][time] row stride

// window sample with uniform index part u (the lane adds its time point): row u mod 32,
// column u div 32 -- lanes (time points 32 samples apart) then sit in consecutive columns
__host__ __device__ constexpr int win_off(int u) { return (u & 31) * kWinRow + (u >> 5); }

struct BankLds {
  union {
    double win[32 * kWinRow];                       // phase 1
    struct {
      double re[kFbBands][kACols];                  // phases 2..4: A[band][time]
      double im[kFbBands][kACols];
    } a;
  };
  double e1[kFbBands][kTileBlocks];
  double hist[kFbBands][10];       // the 10 newest E0 values of the previous tile, oldest first
  double cu[kFbBands];
};

// One band's complex FIR at this lane's time point (fbearmodel.c:404-434).
// win_t = window + t.  Taps are consumed in groups of 8; all LDS offsets inside a 32-tap
// macro step are compile-time constants.  Software pipeline: while group G is evaluated
// (16 LDS reads, 16 add/sub, 16 fma) the 8 coefficient pairs of group G+1 are requested
// through the SCALAR cache (the address is wave-uniform; v_fma_f64 takes the coefficient
// straight from an SGPR pair).  Broadcast loads through the vector memory pipe cost a full
// 64-lane transaction each and saturated the texture addresser (measured: 1.4e9 VMEM
// instructions per launch = the whole kernel time).
// A plain ds_read_b64 moves 256 B/clk/CU, the merged ds_read2_b64 the compiler likes to form
// only 128 (MI355X_MICROARCH.md, LDS table): read the window through volatile accesses so that
// every sample is its own ds_read_b64 with an immediate offset.
__device__ __forceinline__ double lds_rd(const double* p) {
  return *(const volatile __attribute__((address_space(3))) double*)p;
}

typedef double v2d __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) v2d* gcoef_t;

template <int B>
__device__ __forceinline__ void fir_band(const double* __restrict__ win_t, const double2* __restrict__ coef,
                                         double& re_out, double& im_out) {
  constexpr int N = kLen[B];
  constexpr int D = 1 + (kLen[0] - N) / 2;           // (31) in BS.1387
  constexpr int H = N / 2;
  constexpr int U1 = kFbRing - D;                    // x1(n): u = U1 - n   (delay D + n)
  constexpr int U2 = kFbRing - D - N;                // x2(n): u = U2 + n   (delay D + N - n)
  constexpr int FULL = (H - 1) / 32;
  constexpr int REM = (H - 1) - 32 * FULL;
  constexpr int REMG = (REM + 7) / 8;                // groups in the remainder
  gcoef_t hc = (gcoef_t)(coef + coef_offset(B));
  double re = 0., im = 0.;
  // GROUP_FENCE makes the next loads depend on the accumulators: without it the compiler
  // hoists the loads of a whole filter to the top and spills them.
#define GROUP_FENCE(ptr) asm volatile("" : "+s"(ptr), "+v"(re), "+v"(im))
  v2d cc[8], cn[8];
  {
    gcoef_t c0 = hc + 1;
    GROUP_FENCE(c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) cc[j] = (FULL > 0 || j < REM) ? c0[j] : v2d{0., 0.};
  }
#pragma unroll 1
  for (int q = 0; q < FULL; ++q) {
    // n = 1 + 32 q + r: moving 32 taps on shifts the column by one, the row pattern repeats
    const double* p1 = win_t - q;
    const double* p2 = win_t + q;
    gcoef_t c = hc + 1 + 32 * q;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      GROUP_FENCE(c);
      // coefficients of the NEXT group (the first group of the remainder after the last macro step)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rn = 8 * (g + 1) + j;
        cn[j] = (g < 3 || q + 1 < FULL || j < REM) ? c[rn] : v2d{0., 0.};   // c[32..39]: next macro step / remainder
      }
      double a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * g + j;
        a[j] = lds_rd(p1 + win_off(U1 - 1 - r));
        // the reference's doubled ring buffer makes band 0's delay-1456 tap (n = 1) read the
        // NEWEST sample (fb_buf[offset + 1456] aliases fb_buf[offset]); reproduced
        b[j] = (B == 0 && r == 0) ? (q == 0 ? lds_rd(win_t + win_off(kFbRing)) : lds_rd(p2 + win_off(U2 + 1 + r)))
                                  : lds_rd(p2 + win_off(U2 + 1 + r));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        re = fma(a[j] + b[j], cc[j].x, re);          // even symmetry
        im = fma(a[j] - b[j], cc[j].y, im);          // odd symmetry
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) cc[j] = cn[j];
    }
  }
  if (REM > 0) {
    const double* p1 = win_t - FULL;
    const double* p2 = win_t + FULL;
    gcoef_t c = hc + 1 + 32 * FULL;
#pragma unroll
    for (int g = 0; g < REMG; ++g) {
      GROUP_FENCE(c);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int rn = 8 * (g + 1) + j;
        if (g + 1 < REMG) cn[j] = rn < REM ? c[rn] : v2d{0., 0.};
      }
      double a[8], b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * g + j;
        if (r < REM) {
          a[j] = lds_rd(p1 + win_off(U1 - 1 - r));
          b[j] = (B == 0 && FULL == 0 && r == 0) ? lds_rd(win_t + win_off(kFbRing)) : lds_rd(p2 + win_off(U2 + 1 + r));
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = 8 * g + j;
        if (r < REM) {
          re = fma(a[j] + b[j], cc[j].x, re);
          im = fma(a[j] - b[j], cc[j].y, im);
        }
      }
      if (g + 1 < REMG) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cc[j] = cn[j];
      }
    }
  }
  gcoef_t hcv = hc + H;
  GROUP_FENCE(hcv);
#undef GROUP_FENCE
  const double xm = win_t[win_off(U1 - H)];          // centre tap, once
  const v2d ch = *hcv;
  re_out = fma(xm, ch.x, re);
  im_out = fma(xm, ch.y, im);
}

// the ten bands of wave w: { w, 7-w, 8+w, 15-w, 16+w, 23-w, 24+w, 31-w, 32+w, 39-w } --
// a longest-first deal of the filter lengths: 2728 tap pairs per wave
template <int W>
__device__ __forceinline__ void fir_wave(const double* win_t, const double2* coef, double (&re)[10], double (&im)[10]) {
  // scheduling fences: the ten filters are independent, without them the scheduler overlaps
  // their load phases and runs out of registers
  fir_band<W>(win_t, coef, re[0], im[0]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<7 - W>(win_t, coef, re[1], im[1]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<8 + W>(win_t, coef, re[2], im[2]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<15 - W>(win_t, coef, re[3], im[3]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<16 + W>(win_t, coef, re[4], im[4]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<23 - W>(win_t, coef, re[5], im[5]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<24 + W>(win_t, coef, re[6], im[6]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<31 - W>(win_t, coef, re[7], im[7]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<32 + W>(win_t, coef, re[8], im[8]);
  __builtin_amdgcn_sched_barrier(0);
  fir_band<39 - W>(win_t, coef, re[9], im[9]);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ int wave_band(int w, int i) { return (i & 1) ? 8 * (i >> 1) + 7 - w : 8 * (i >> 1) + w; }

__global__ __launch_bounds__(256, 3) void fb_bank_kernel(FbFrontArgs a, unsigned n_signals) {
  __shared__ BankLds sh;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: say so
  const unsigned g = blockIdx.x;
  const int sig = g & 1;
  const int chan = (g >> 1) % a.channels;
  const unsigned pair = g / (2 * a.channels);
  const unsigned n_blocks = a.n_blocks ? a.n_blocks[pair] : a.n_blocks_uniform;
  if (n_blocks <= a.block0) return;
  const unsigned nb_mine = min(a.blocks_per_launch, n_blocks - a.block0);
  const BandTables* __restrict__ bt = a.bands;
  const FbTables* __restrict__ fb = a.fb;
  const double2* __restrict__ coef = reinterpret_cast<const double2*>(fb->h_ri);
  const size_t row_len = a.hp_row_stride;
  const size_t row_valid = (size_t)kFbRing + (size_t)a.blocks_per_launch * kFbFrame;
  const double* __restrict__ row = a.hp_scratch + (size_t)g * row_len;
  FbSignalState* __restrict__ st = a.fbstate + g;

  // recurrent state -> LDS / registers
  if (tid < kFbBands) {
    sh.cu[tid] = st->cu[tid];
#pragma unroll
    for (int i = 0; i < 10; ++i) sh.hist[tid][i] = st->e0_hist[tid][i];
  }
  double exc = tid < kFbBands ? st->excitation[tid] : 0.;
  // (1-A)^(t+1): decay of the slope-filter state that enters a tile
  double decay;
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
  const int t = lane < kTileSub ? lane : kTileSub - 1;

  for (unsigned b0 = 0; b0 < nb_mine; b0 += kTileBlocks) {
    const unsigned nvb = min((unsigned)kTileBlocks, nb_mine - b0);   // valid blocks in this tile
    const int nvs = 6 * nvb;                                         // valid sub-samples
    __syncthreads();                                                 // previous tile is done with the LDS
    // ---- phase 0: window of the filtered signal, samples [192 b0 - 1456, 192 b0 + 59*32] ------
    {
      const double* src = row + (size_t)b0 * kFbFrame;               // row index 0 = sample -1456 of the launch
      const int avail = (int)min((size_t)kWin, row_valid - (size_t)b0 * kFbFrame);
      for (int wdx = tid; wdx < kWin; wdx += 256) sh.win[win_off(wdx)] = wdx < avail ? src[wdx] : 0.;
    }
    __syncthreads();
    // ---- phase 1: the complex FIR filters (fbearmodel.c:399-435) ---------------------------------
    double re[10], im[10];
    {
      const double* win_t = sh.win + t;
      switch (wv) {
        case 0: fir_wave<0>(win_t, coef, re, im); break;
        case 1: fir_wave<1>(win_t, coef, re, im); break;
        case 2: fir_wave<2>(win_t, coef, re, im); break;
        default: fir_wave<3>(win_t, coef, re, im); break;
      }
    }
    __syncthreads();                                                 // everybody is done with the window
    // ---- phase 2a: A = filter outputs (overlays the window) ----------------------------------------
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int b = wave_band(wv, i);
      sh.a.re[b][lane] = re[i];
      sh.a.im[b][lane] = im[i];
    }
    __syncthreads();
    // ---- phase 2b: level-dependent upward spreading (fbearmodel.c:327-349).  The slope
    // filter runs along time = along the lanes (inclusive scan with the carried-in state);
    // every source band adds its geometric tail into the bands above it with LDS atomics
    // (one column per lane: no contention inside an instruction) -----------------------------------
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int b = wave_band(wv, i);
      const double level = 10. * log10(re[i] * re[i] + im[i] * im[i]);
      const double slope = fmax(4., 24. + 230. / bt->fc[b] - 0.2 * level);
      const double dist_s = exp(slope * kLnDist);                    // pow(DIST, s)
      double v = kSlopeA * dist_s, m = 1. - kSlopeA;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(v, d, 64);
        if (lane >= d) v += m * o;
        m *= m;
      }
      const double cu = v + decay * sh.cu[b];
      const double carry = __shfl(cu, nvs - 1, 64);
      if (lane == 0) sh.cu[b] = carry;                               // only this wave touches cu[b]
      double d1 = re[i], d2 = im[i];
      for (int j = b + 1; j < kFbBands; ++j) {
        d1 *= cu;
        d2 *= cu;
        atomicAdd(&sh.a.re[j][lane], d1);
        atomicAdd(&sh.a.im[j][lane], d2);
      }
    }
    __syncthreads();
    // ---- phase 3: downward spreading (fbearmodel.c:351-354): wave 0 the real, wave 1 the
    // imaginary parts; a column per lane ----------------------------------------------------------
    if (wv < 2) {
      double (*A)[kACols] = wv == 0 ? sh.a.re : sh.a.im;
      double acc = A[kFbBands - 1][lane];
#pragma unroll 13
      for (int b = kFbBands - 1; b > 0; --b) {
        acc = A[b - 1][lane] + kCL * acc;
        A[b - 1][lane] = acc;
      }
    }
    __syncthreads();
    // ---- phase 4: rectification + backward masking at block rate (fbearmodel.c:357-382);
    // one (band, block) per thread --------------------------------------------------------------------
    for (int item = tid; item < kFbBands * kTileBlocks; item += 256) {
      const int b = item / kTileBlocks, blk = item - b * kTileBlocks;
      // E0 of sub-sample s (negative: previous tile)
      auto e0 = [&](int s) {
        if (s < 0) return sh.hist[b][10 + s];
        const double x = sh.a.re[b][s], y = sh.a.im[b][s];
        return x * x + y * y;
      };
      const int s_new = 6 * blk + 5;                 // newest sub-sample of the block
      double e1 = 0.;
#pragma unroll
      for (int i = 0; i < 5; ++i) e1 += (e0(s_new - i) + e0(s_new - 10 + i)) * fb->back_mask[i];
      e1 += e0(s_new - 5) * fb->back_mask[5];
      sh.e1[b][blk] = e1;
    }
    __syncthreads();
    // history for the next tile: the 10 newest VALID sub-samples, oldest first
    // (400 (band, slot) items on 256 threads: two per thread)
    double hnew[2] = {0., 0.};
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int item = tid + 256 * rep;
      if (item < kFbBands * 10) {
        const int b = item / 10, k = item - b * 10;
        const int s = nvs - 10 + k;
        if (s < 0) {
          hnew[rep] = sh.hist[b][10 + s];
        } else {
          const double x = sh.a.re[b][s], y = sh.a.im[b][s];
          hnew[rep] = x * x + y * y;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int item = tid + 256 * rep;
      if (item < kFbBands * 10) sh.hist[item / 10][item % 10] = hnew[rep];
    }
    // ---- phase 5: internal noise + forward masking (fbearmodel.c:385-394), thread = band ------------
    if (tid < kFbBands) {
      const double noise = bt->internal_noise[tid], ac = bt->ear_tc[tid];
      for (unsigned bl = 0; bl < nvb; ++bl) {
        const double unsm = sh.e1[tid][bl] + noise;
        exc = ac * exc + (1. - ac) * unsm;
        double* rec = a.records + ((size_t)(pair * a.blocks_per_launch + b0 + bl) * a.channels + chan) * kFbRecDoubles;
        rec[(sig ? kFbRecUnsmTest : kFbRecUnsmRef) + tid] = unsm;
        rec[(sig ? kFbRecExcTest : kFbRecExcRef) + tid] = exc;
      }
    }
  }
  __syncthreads();
  if (tid < kFbBands) {
    st->cu[tid] = sh.cu[tid];
#pragma unroll
    for (int i = 0; i < 10; ++i) st->e0_hist[tid][i] = sh.hist[tid][i];
    st->excitation[tid] = exc;
  }
}

hipError_t launch_fb_hp(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const unsigned n_signals = n_pairs * a.channels * 2;
  if (n_signals == 0 || a.blocks_per_launch == 0) return hipSuccess;
  hipLaunchKernelGGL(fb_hp_kernel, dim3((n_signals + 63) / 64), dim3(64), 0, stream, a, n_signals);
  return hipGetLastError();
}

hipError_t launch_fb_bank(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const unsigned n_signals = n_pairs * a.channels * 2;
  if (n_signals == 0 || a.blocks_per_launch == 0) return hipSuccess;
  hipLaunchKernelGGL(fb_bank_kernel, dim3(n_signals), dim3(256), 0, stream, a, n_signals);
  return hipGetLastError();
}

hipError_t launch_fb_frontend(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const hipError_t e = launch_fb_hp(a, n_pairs, stream);
  return e != hipSuccess ? e : launch_fb_bank(a, n_pairs, stream);
}

}  // namespace peaq
