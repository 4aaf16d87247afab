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
//                   stores are 128-byte runs.
//  fb_bank_kernel   one WORKGROUP of four waves per (pair, channel, signal), walking the
//                   chunk in tiles of 60 sub-samples (= 10 blocks of 192 samples).  The
//                   window of the filtered signal sits in LDS; the 40 complex FIR filters
//                   (:399-435) run as two folded GEMMs on the matrix cores (fir_mfma
//                   below).  Then with lanes = TIME points (every 32nd sample, :314):
//                   level dependent spreading (slope filter as a wave scan, :327-354),
//                   rectification (:357-360); the 11-tap backward-masking FIR at block
//                   rate (:364-382), internal noise and forward masking (:385-394)
//                   with threads = bands.
#include <hip/hip_runtime.h>

#include "peaq_device.h"
#include "peaq_kernels.h"
#include "peaq_wave.h"
#include <type_traits>
#ifdef PEAQ_DEV_PROBES                               // VARIANT builds only (csrc/Makefile): never in the product library
#include "dev_probes.inc"
#endif

namespace peaq {

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
    // (in - 2 x1: doubling is exact, so the fused form rounds once to the same value as the two operations)
    const double ya = __builtin_fma(-2., x1, in) + x2 + 1.99517 * y1a - 0.995174 * y2a;
    const double yb = __builtin_fma(-2., y1a, ya) + y2a + 1.99799 * y1b - 0.997998 * y2b;
    x2 = x1;
    x1 = in;
    y2a = y1a;
    y1a = ya;
    y2b = y1b;
    y1b = yb;
    return yb;
  }
};

// Workgroups of kHpWaves INDEPENDENT waves.  Beside the FP64 engine's bank kernel -- whose two workgroups fill a CU's
// registers and LDS -- a workgroup of this kernel takes the place of one of them for as long as the walk lasts: in
// workgroups of one wave the dispatcher spreads a 4096-pair batch's 256 waves over 256 CUs, in workgroups of four
// (one wave per SIMD) over 64.  Measured, 4096 pairs, advanced pass of the FP64 engine: round 4 (the walk 28 ms
// alone) 1 wave 413-416 ms, 2: 409, 4: 404, 8: 418; round 5 (17 ms, two of the three walks beside the FFT path's
// head) 1: 355, 2: 353, 4: 350, 8: 353 (profiles/r05_ab_adv.txt).
#ifndef PEAQ_HP_WAVES
#define PEAQ_HP_WAVES 4
#endif
constexpr int kHpWaves = PEAQ_HP_WAVES;
// The boundary detector of gstpeaq.c:1083-1096 (FLOAT running sum over five |x|, tested from i = 5 on) in the same
// arithmetic with less of it: |x| as a double is the absolute value of the sample's double, the history is kept
// converted (zeros at a block's start: i < 5 adds |x| itself), and the threshold 200 / 32768 is a float -- so
// "sum >= threshold at some i >= 5" is a float maximum compared once per block.
struct BoundaryDetector {
  double sumd = 0., g0 = 0., g1 = 0., g2 = 0., g3 = 0., g4 = 0.;
  float smax = 0.f;
  __device__ __forceinline__ void step(double xd, bool is_sample_4) {
    const double ax = fabs(xd);
    sumd = (double)(float)(sumd + (ax - g0));
    smax = fmaxf(smax, (float)sumd);                 // (a NaN sum is never "reached", like the comparison)
    if (is_sample_4) smax = 0.f;
    g0 = g1; g1 = g2; g2 = g3; g3 = g4; g4 = ax;
  }
  __device__ __forceinline__ int above() const { return smax >= (float)(200. / 32768); }
};

__global__ __launch_bounds__(64 * kHpWaves) void fb_hp_kernel(FbFrontArgs a, unsigned n_signals) {
  __shared__ double tiles[kHpWaves][64][17];        // per wave: [signal in wave][16 samples], padded
  __shared__ float inbufs[kHpWaves][64 * 18];
  const int lane = threadIdx.x & 63;
  double (*tile)[17] = tiles[threadIdx.x >> 6];
  float* inbuf = inbufs[threadIdx.x >> 6];
  const unsigned g0 = blockIdx.x * (64 * kHpWaves) + (threadIdx.x & ~63u);
  if (g0 >= n_signals) return;                       // (the waves of a workgroup are independent: no barrier below)
  // A wave with fewer than 64 signals left: the spare lanes walk the LAST signal too -- the same loads, the same
  // arithmetic, the same stores of the same values to the same addresses in the same instructions -- so that every
  // wave is a full one and its blocks can take the straight-line path below (a single pair is four signals).
  const unsigned gg = min(g0 + lane, n_signals - 1);
  const int sig = gg & 1;
  const int chan = (gg >> 1) % a.channels;
  const unsigned pair = gg / (2 * a.channels);
  const unsigned n_sig = sig ? (a.n_test ? a.n_test[pair] : a.n_uniform_test)
                             : (a.n_ref ? a.n_ref[pair] : a.n_uniform_ref);
  const unsigned n_blocks = a.n_blocks ? a.n_blocks[pair] : a.n_blocks_uniform;
  const long long off = sig ? a.off_test : a.off_ref;
  const size_t row_len = a.hp_row_stride;
  double* __restrict__ rows = a.hp_scratch;
  // where this signal's blocks of the launch start, how many there are, and where its state lives
  unsigned blk0 = a.block0, origin = a.block_origin, prev_blocks = a.prev_blocks, nb_mine = 0, state_idx = gg;
  bool first = a.first_launch;
  if (a.windows) {
    const FbPairWindow w = a.windows[pair];
    blk0 = origin = w.block0;
    prev_blocks = w.prev_blocks;
    first = w.block0 == 0;
    state_idx = (w.slot * a.channels + chan) * 2 + sig;
    nb_mine = w.n_blocks;
  } else if (n_blocks > a.block0) {
    nb_mine = min(a.blocks_per_launch, n_blocks - a.block0);
  }
  FbSignalState* __restrict__ st = a.fbstate + state_idx;

  unsigned nb_max = nb_mine;                         // wave-uniform loop bound
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) nb_max = max(nb_max, (unsigned)__shfl_xor((int)nb_max, d, 64));

  // history: the newest 1456 filtered samples of the previous launch sit at the row's tail and move to its head.
  // Their peak is taken here, where it is exact: with launches of a block or two (sessions, the broker) the head
  // spans up to eight earlier launches, and "this launch and the previous one" would miss a burst that lies further
  // back.  Row by row with the whole wave (coalesced; all of a row's loads in front of its stores): a lane moving its
  // own row sample by sample took longer than the walk itself in a broker tick (64 cache lines per instruction,
  // 1456 round trips).  With launches of fewer than eight blocks source and destination overlap: the loads of a row
  // all lie at or above everything written before them, so the order is safe -- but only ONCE per row, which is why
  // the spare lanes' copies of the last row are left out.
  double peak_head = 0.;
  {
    const int n_rows = (int)min(64u, n_signals - g0);
    constexpr int kPer = (kFbRing + 63) / 64;
    for (int r = 0; r < n_rows; ++r) {
      if (__builtin_amdgcn_readlane((int)nb_mine, r) == 0) continue;
      const size_t at = (size_t)(unsigned)__builtin_amdgcn_readlane((int)state_idx, r) * row_len;
      double* dst = rows + at;                       // (not restrict: a short launch's source overlaps it)
      if (__builtin_amdgcn_readlane((int)first, r)) {
#pragma unroll
        for (int q = 0; q < kPer; ++q)
          if (lane + 64 * q < kFbRing) dst[lane + 64 * q] = 0.;
      } else {
        const double* src = (a.hp_prev ? a.hp_prev : rows) + at + (size_t)(unsigned)__builtin_amdgcn_readlane((int)prev_blocks, r) * kFbFrame;
        double v[kPer], pm = 0.;
#pragma unroll
        for (int q = 0; q < kPer; ++q) v[q] = lane + 64 * q < kFbRing ? src[lane + 64 * q] : 0.;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
          pm = fmax(pm, fabs(v[q]));
          if (lane + 64 * q < kFbRing) dst[lane + 64 * q] = v[q];
        }
        pm = wave_max(pm);
        if (lane == r || (r == n_rows - 1 && lane >= n_rows)) peak_head = pm;
      }
    }
  }
  HpWalk w{st->hp[0], st->hp[1], st->hp[2], st->hp[3], st->hp[4], st->hp[5]};
  double peak = 0.;                                  // largest |filtered sample| of this signal's blocks

  // The input: CHUNKS of 16 samples, one chunk ahead of their use (the walk is a chain of dependent FP64 operations:
  // the loads must never be waited for).  A chunk of a (pair, signal) row is 64 C contiguous bytes (C channels
  // interleaved); the wave fetches the chunks of all its rows as 16-byte pieces -- four per lane, eight lanes' pieces
  // per cache line -- parks them in LDS and every lane picks its channel's 16 samples there.  (One dword per lane
  // and sample, as before: 64 different cache lines per load instruction, a 64-bit address and a bounds check per
  // sample -- a quarter of the kernel's instructions.)
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int C = a.channels;
  const int kInStride = C == 2 ? 34 : 18;            // floats per row in LDS: 16 C samples + padding (8-byte rows; the
                                                     // channels' reads fall on different banks, two-way at worst)
  const int ld_row = lane / C, ld_piece0 = 4 * (lane % C);            // the row this lane fetches for, its first piece
  const unsigned ld_gg = min(g0 + (unsigned)((ld_row >> 1) * C) * 2u + (unsigned)(ld_row & 1), n_signals - 1);   // that row's channel-0 signal
  const unsigned ld_pair = ld_gg / (2 * C);          // (spare rows: the last pair's)
  const float* __restrict__ ld_base = ((ld_row & 1) ? a.test : a.ref) + (size_t)ld_pair * a.pair_stride * C;
  const long long ld_len = (long long)a.pair_stride * C;              // floats in a row
  // sample index (in its row) of sample 0 of the launch's block bl: the same for every pair (broker launches:
  // blk0 == origin; batch launches: both uniform)
  const long long s_first = (long long)(blk0 - origin) * kFbFrame;
  const long long ld_s0 = s_first + ((ld_row & 1) ? a.off_test : a.off_ref);
  const long long my_s0 = s_first + off;
  const unsigned n_chunks = nb_max * (kFbFrame / 16);
  auto load_chunk = [&](unsigned c, f4u (&v)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long long f = (ld_s0 + 16ll * c) * C + 4 * (ld_piece0 + j);    // first float of the piece in its row
      f4u t = {0.f, 0.f, 0.f, 0.f};
      if (c < n_chunks && f >= 0) {
        if (f + 4 <= ld_len) {
          t = *reinterpret_cast<const f4u*>(ld_base + f);
        } else {                                                        // the row ends inside the piece
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (f + e < ld_len) t[e] = ld_base[f + e];
        }
      }
      v[j] = t;
    }
  };
  f4u nxt[4];
  load_chunk(0, nxt);
  const int in_at = (2 * (lane / (2 * C)) + sig) * kInStride + chan;       // this lane's first sample of a chunk in LDS
  auto chunk_to_lds = [&]() {                        // the chunk requested a chunk ago: into LDS, to be read by channel
    float* dst = inbuf + ld_row * kInStride + 4 * ld_piece0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      reinterpret_cast<float2*>(dst + 4 * j)[0] = make_float2(nxt[j][0], nxt[j][1]);
      reinterpret_cast<float2*>(dst + 4 * j)[1] = make_float2(nxt[j][2], nxt[j][3]);
    }
    // the workgroup's waves are independent: LDS is in program order, no barrier needed -- and __syncthreads() would
    // drain the prefetched loads and the stores (vmcnt(0))
    wave_lds_fence();
  };

  // Where the transposed tiles go: store instruction r carries sixteen samples of the signals 8 r + lane / 8 (their
  // rows fetched once -- as shuffles in the loop they were LDS round trips in front of every store)
  unsigned row_s[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) row_s[r] = (unsigned)__shfl((int)state_idx, r * 8 + (lane >> 3), 64);
  double* __restrict__ const out0 = rows + kFbRing + 2 * (lane & 7);      // + row * row_len + the sample's index
  typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));

  // STRAIGHT-LINE blocks: the block inside every lane's signal and block count, and the sixteen-byte pieces of its
  // chunks and of the chunk after it inside their rows.  Such a block has no bounds checks, no zero padding,
  // unconditional loads and stores -- and that is what makes it fast: a wave's vector memory operations return
  // through ONE in-order counter, and with loads or stores under lane masks the compiler has to wait for the counter
  // to reach zero wherever it needs a load's data, i.e. for the acknowledgement of every store of the chunk before
  // (measured: the walk took 410 cycles per sample, 145 of them instructions).
  unsigned f_lo, f_hi;
  {
    auto ceil_div = [](long long x, long long d) { return x <= 0 ? 0ll : (x + d - 1) / d; };
    auto floor_div = [](long long x, long long d) { return x <= 0 ? 0ll : x / d; };
    // chunks [c_lo, c_hi) of the launch lie inside this lane's signal; chunks [l_lo, l_hi) inside its loader row
    const long long c_lo = ceil_div(-my_s0, 16), c_hi = min((long long)nb_mine * (kFbFrame / 16), floor_div((long long)n_sig - my_s0, 16));
    const long long l_lo = ceil_div(-ld_s0, 16), l_hi = floor_div((long long)a.pair_stride - ld_s0, 16);
    // block bl: chunks 12 bl .. 12 bl + 11 computed, 12 bl + 1 .. 12 bl + 12 loaded
    const long long lo = max(ceil_div(c_lo, 12), ceil_div(l_lo - 1, 12));
    const long long hi = min(floor_div(c_hi, 12), l_hi >= 13 ? (l_hi - 13) / 12 + 1 : 0ll);
    unsigned ulo = (unsigned)min(lo, 1ll << 28), uhi = (unsigned)min(hi, 1ll << 28);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      ulo = max(ulo, (unsigned)__shfl_xor((int)ulo, d, 64));
      uhi = min(uhi, (unsigned)__shfl_xor((int)uhi, d, 64));
    }
    f_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)ulo);
    f_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)uhi);
  }
  const float* __restrict__ ld_ptr = ld_base + ld_s0 * C + 4 * ld_piece0;   // (never dereferenced outside the row)

  for (unsigned bl = 0; bl < nb_max; ++bl) {
    const bool mine = bl < nb_mine;
    BoundaryDetector det;
    if (bl >= f_lo && bl < f_hi) {
      auto fast_chunk = [&](const int kc, const bool block_head) __attribute__((always_inline)) {
        const unsigned c = bl * (kFbFrame / 16) + kc;
        float xc[16];
        chunk_to_lds();
#pragma unroll
        for (int k = 0; k < 16; ++k) xc[k] = inbuf[in_at + k * C];
        wave_lds_fence();
        {
          const float* __restrict__ src = ld_ptr + (size_t)(16u * (c + 1)) * C;
#pragma unroll
          for (int j = 0; j < 4; ++j) nxt[j] = *reinterpret_cast<const f4u*>(src + 4 * j);
        }
        double y[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          y[k] = w.step((double)xc[k] * a.level_factor);
          peak = fmax(peak, fabs(y[k]));
        }
        // the detector only until every lane's block has reached the threshold (a block's flag never falls back):
        // with anything but near-silence that is its first chunk, and a quarter of the walk's instructions is gone
        if (block_head || !__all(det.above())) {
#pragma unroll
          for (int k = 0; k < 16; ++k) det.step((double)xc[k], block_head && k == 4);
        }
        // 64 signals x 16 samples transposed through LDS: a store instruction writes eight 128-byte runs
#pragma unroll
        for (int k = 0; k < 16; ++k) tile[lane][k] = y[k];
        wave_lds_fence();
        double2 v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double* t = &tile[r * 8 + (lane >> 3)][2 * (lane & 7)];
          v[r] = make_double2(t[0], t[1]);
        }
        const size_t at = (size_t)bl * kFbFrame + 16 * kc;
#pragma unroll
        // (streaming stores: 21 GB per launch of a 4096-pair batch pass through L2 once -- + 0.5 % on the pass)
        for (int r = 0; r < 8; ++r)
          __builtin_nontemporal_store(d2u{v[r].x, v[r].y}, reinterpret_cast<d2u*>(out0 + ((size_t)row_s[r] * row_len + at)));
        wave_lds_fence();
      };
      // the first chunk on its own: the loop's entry then has the same operations in flight as its back edge (four
      // loads, then eight stores), and the compiler waits for exactly the loads
      fast_chunk(0, true);
#pragma unroll 1
      for (int kc = 1; kc < kFbFrame / 16; ++kc) fast_chunk(kc, false);
    } else {
      // ---- any other block (a signal's first and last, ragged batches): sample by sample out of LDS.  Rolled, so
      // that this path costs the kernel neither registers nor instruction cache; the next sample is requested while
      // the current one is worked on ----
#pragma unroll 1
      for (int kc = 0; kc < kFbFrame / 16; ++kc) {
        const unsigned c = bl * (kFbFrame / 16) + kc;
        chunk_to_lds();
        float xn = inbuf[in_at];
        load_chunk(c + 1, nxt);
        const long long s_c = my_s0 + 16ll * c;      // zero padding (gstpeaq.c:733-738) and idle lanes
#pragma unroll 1
        for (int k = 0; k < 16; ++k) {
          const float xr = xn;
          xn = inbuf[in_at + min(k + 1, 15) * C];
          const double xd = (double)((mine && s_c + k >= 0 && s_c + k < (long long)n_sig) ? xr : 0.f);
          const double y = w.step(xd * a.level_factor);
          peak = fmax(peak, fabs(y));
          det.step(xd, kc == 0 && k == 4);
          tile[lane][k] = y;
        }
        wave_lds_fence();
        const size_t at = (size_t)bl * kFbFrame + 16 * kc;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const double* t = &tile[r * 8 + (lane >> 3)][2 * (lane & 7)];
          const d2u v = {t[0], t[1]};
          if (bl < (unsigned)__shfl((int)nb_mine, r * 8 + (lane >> 3), 64)) *reinterpret_cast<d2u*>(out0 + ((size_t)row_s[r] * row_len + at)) = v;
        }
        wave_lds_fence();
      }
    }
    if (mine && sig == 0)
      a.records[((size_t)(pair * a.blocks_per_launch + bl) * a.channels + chan) * kFbRecDoubles + kFbRecFlags] =
          (double)det.above();
    // the wave walks on (with zero input) until its longest signal is done: the state goes home as it
    // is after this signal's own last block
    if (bl + 1 == nb_mine) {
      st->hp[0] = w.x1; st->hp[1] = w.x2; st->hp[2] = w.y1a; st->hp[3] = w.y2a; st->hp[4] = w.y1b;
      st->hp[5] = w.y2b;
      const int slot = a.launch_idx % 3;
      st->peak_slot[slot][0] = peak;
      st->peak_slot[slot][1] = peak_head;              // the window's head: the last 1456 filtered samples before this launch
      st->peak_last = peak;
    }
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
#ifndef PEAQ_FB_OLD_NEXT_MAP
#define PEAQ_FB_OLD_NEXT_MAP 0                       // (development: 1 = round 5's sample-major hand-over of the next window)
#endif
constexpr int kTileSub = 60;                        // sub-samples per tile (10 blocks)
constexpr int kTileBlocks = 10;
constexpr int kWin = (kTileSub - 1) * 32 + kFbRing + 1;   // 3345 filtered samples
constexpr int kWinCols = (kWin + 31) / 32;          // 105
// row stride in doubles.  The four delay groups of a wave (lanes 16 g .. 16 g + 15) read 16 consecutive
// doubles each from four consecutive rows in one ds_read_b64: with a stride of 16 mod 32 doubles two
// neighbouring rows cover all 64 banks between them, so both halves of the wave are conflict free
// (the old stride 106 made them overlap on 12 banks: 30 % of the LDS cycles were conflicts)
constexpr int kWinRow = 112;
static_assert(kWinRow > kWinCols && kWinRow % 32 == 16, "window rows: long enough, and two rows apart = 32 banks");
// A[band][time] row stride: an odd number of doubles, so that the (band, block) items of phase 4 -- lanes on
// different rows at the same column -- do not all fall on one pair of banks.
constexpr int kACols = 65;

// window sample with uniform index part u (the lane adds its time point): row u mod 32,
// column u div 32 -- lanes (time points 32 samples apart) then sit in consecutive columns
__host__ __device__ constexpr int win_off(int u) { return (u & 31) * kWinRow + (u >> 5); }

// The window of the filtered signal in LDS, B operand of the GEMM.  WT = the type the FIR phase computes in:
// double (v_mfma_f64_16x16x4_f64), float (v_mfma_f32_16x16x4_f32) -- 32 rows of kWinRow, see win_off -- or
// _Float16 (v_mfma_f32_16x16x32_f16 on split operands, fir_mfma_h3): two linear arrays (high and low part) with
// 16 bytes of padding after every 32 samples, so that the sixteen time points of a B operand (32 samples = 80
// bytes apart) start on sixteen different groups of four banks.
template <typename WT>
struct Window {
  WT v[32 * kWinRow];
  __device__ __forceinline__ void put(int u, double x, double /*scale*/) { v[win_off(u)] = (WT)x; }
  __device__ __forceinline__ double get(int u, double /*unscale*/) const { return (double)v[win_off(u)]; }
  // the next tile's window starts 60 columns (1920 samples) further on
  __device__ __forceinline__ void shift(int tid) {
    for (int e = tid; e < 32 * (kWinCols - kTileSub); e += 256) {
      const int r = e / (kWinCols - kTileSub), c = e - r * (kWinCols - kTileSub);
      v[r * kWinRow + c] = v[r * kWinRow + kTileSub + c];
    }
  }
  // the same with all of a thread's reads in front of its writes (one LDS round trip instead of six: as a loop of
  // read - write the compiler has to assume that a write changes what the next read returns)
  __device__ __forceinline__ void shift_batched(int tid) {
    constexpr int kC = kWinCols - kTileSub, kN = 32 * kC, kPer = (kN + 255) / 256;
    static_assert(256 / kC == 5 && kN > 256 * (kPer - 1), "element e + 256 sits five rows and 256 - 5 kC columns on");
    int r = tid / kC, c = tid - r * kC;              // element tid + 256 q: row r, column c of the kept part
    WT tmp[kPer];
    int off[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
      off[q] = r * kWinRow + c;
      tmp[q] = v[(q < kPer - 1 || tid + 256 * q < kN ? off[q] : 0) + kTileSub];
      c += 256 - 5 * kC;
      r += 5;
      if (c >= kC) {
        c -= kC;
        r += 1;
      }
    }
#pragma unroll
    for (int q = 0; q < kPer; ++q)
      if (q < kPer - 1 || tid + 256 * q < kN) v[off[q]] = tmp[q];
  }
};
constexpr int kWinHBlocks = 109;                    // blocks of 32 samples: window index up to 1455 + 32 * 63 + 7
constexpr int kWinHBytes = kWinHBlocks * 80;
__host__ __device__ constexpr int winh_off(int u) { return 2 * u + 16 * (u >> 5); }   // byte offset of sample u
template <>
struct Window<_Float16> {
  alignas(16) unsigned char hi[kWinHBytes];
  alignas(16) unsigned char lo[kWinHBytes];
  __device__ __forceinline__ void put(int u, double x, double scale) {
    // (full scale sits at 2^10..2^11; a launch whose filtered signal peaks above 2^15 at that scale runs at a
    // smaller power of two instead -- fb_bank_body, `xs` -- so the clamp below only ever meets NaN/inf input)
    const float s = fminf(fmaxf((float)(x * scale), -65504.f), 65504.f);
    const _Float16 h = (_Float16)s;
    *reinterpret_cast<_Float16*>(hi + winh_off(u)) = h;
    *reinterpret_cast<_Float16*>(lo + winh_off(u)) = (_Float16)(s - (float)h);
  }
  __device__ __forceinline__ double get(int u, double unscale) const {
    return ((double)*reinterpret_cast<const _Float16*>(hi + winh_off(u)) +
            (double)*reinterpret_cast<const _Float16*>(lo + winh_off(u))) * unscale;
  }
  __device__ __forceinline__ void shift(int tid) {   // 60 blocks = 4800 bytes; source and destination do not overlap
    constexpr int kMove = (kWinHBytes - kTileSub * 80) / 16;
    static_assert(kMove <= 256 && kWinHBytes - kTileSub * 80 <= kTileSub * 80, "one 16-byte piece per thread, no overlap");
    if (tid < kMove) {
      reinterpret_cast<uint4*>(hi)[tid] = reinterpret_cast<const uint4*>(hi + kTileSub * 80)[tid];
      reinterpret_cast<uint4*>(lo)[tid] = reinterpret_cast<const uint4*>(lo + kTileSub * 80)[tid];
    }
  }
};

template <typename WT>
struct BankLds {
  // (the window is not the first member: bs_pair below reads up to 44 entries in front of a staging row -- values it
  // then discards -- and those addresses must stay inside the allocation whichever wave's rows they are)
  double hist[kFbBands][10];       // the 10 newest E0 values of the previous tile, oldest first
  double cu[kFbBands];
  double c0[kFbBands];             // ln DIST (24 + 230 / fc): constant of the slope exponent, per band
  double vst[kBsChains][2];        // FP64 engine: the running sums V of the block-sum form after the previous tile
  Window<WT> win;                                   // phase 1: the filtered signal
  struct {
    double re[kFbBands][kACols];                    // A[band][time]: GEMM result, then phases 2..4 in place
    double im[kFbBands][kACols];
  } a;
  double e1[kFbBands][kTileBlocks];
  double ex[kFbBands][kTileBlocks];                 // excitation per (band, block) on its way to the records
};

// A plain ds_read_b64 moves 256 B/clk/CU, the merged ds_read2_b64 the compiler likes to form
// only 128 (MI355X_MICROARCH.md, LDS table): volatile accesses keep every read its own ds_read_b64.
__device__ __forceinline__ double lds_rd(const double* p) {
  return *(const volatile __attribute__((address_space(3))) double*)p;
}
__device__ __forceinline__ float lds_rd(const float* p) {
  return *(const volatile __attribute__((address_space(3))) float*)p;
}

// ---------------------------------------------------------------------------
// Phase 1, the 40 complex FIR filters (fbearmodel.c:399-435), on the matrix cores.
// For the 60 time points of a tile (one every 32 samples) the bank is a pair of GEMMs over the
// delays d = 2..729 (peaq_device.h, kMf*):
//   re[b][t] = sum_d Hre[b][d] (x[32t - d] + x[32t - (1458 - d)])
//   im[b][t] = sum_d Him[b][d] (x[32t - d] - x[32t - (1458 - d)])
// v_mfma_f64_16x16x4_f64: 16 bands x 16 time points x 4 delays per instruction, the B operands
// built on the fly from two LDS reads of the window.  A wave keeps eight accumulator tiles
// (64 time points x {re, im}) so that every coefficient fetch (two coalesced 512-byte reads
// through L2 per K step) feeds eight MFMAs.  The 261 K steps of the three row tiles are cut
// into four runs, one per wave; a run that ends inside a row tile leaves a partial sum, so all
// runs are added into A with LDS atomics (A is zeroed in phase 0).  FP64 MFMA has the rate of
// the FP64 vector pipe, but it issues once per 64 cycles instead of once per 4: the pipe is
// actually kept full and the VALU stays free for the operand construction.
// ---------------------------------------------------------------------------
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

// the two matrix instructions behind one interface; ROW(lane >> 4, i) = row of accumulator element i
struct MfmaF64 {
  typedef double T;
  typedef v4d Acc;
  static __device__ __forceinline__ Acc mma(double a, double b, Acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int kk, int i) { return kk + 4 * i; }
};
struct MfmaF32 {
  typedef float T;
  typedef v4f Acc;
  static __device__ __forceinline__ Acc mma(float a, float b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int kk, int i) { return 4 * kk + i; }
};

// split-FP16 variant (fir_mfma_h3): accumulators and result layout of the FP32 instruction
struct MfmaH3 {
  typedef _Float16 T;
  typedef v4f Acc;
  static __device__ __forceinline__ int row(int kk, int i) { return 4 * kk + i; }
};

// FP32 variant (PEAQ_FIR_F32, selectable): the mixed-precision ledger (tools/precision_ledger.py,
// profiles/r02_precision_ledger.json) prices it at max |dODG| = 5e-8 over 39 advanced cases -- the FIR outputs
// only enter the model as |A|^2 in 40 bands after spreading and masking.  The engine's default is the FP64 block-sum
// form (bs_pair, the arithmetic the stage tests hold to 1e-9 of the oracle); peaq_ctx_set_fir_mode() / PEAQ_AMD_FIR
// select this one or the split-FP16 form below (fir_mfma_h3).
typedef float v2f __attribute__((ext_vector_type(2)));

// Adds the accumulator tiles of one run of K steps into A (LDS atomics; A was zeroed in phase 0).
// D layout: column = lane & 15 (time), row = M::row(lane >> 4, i)  ->  band 16 r + row
template <typename M>
__device__ __forceinline__ void fir_store(BankLds<typename M::T>& sh, int r, int j, int kk,
                                          const typename M::Acc (&accr)[4], const typename M::Acc (&acci)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = 16 * r + M::row(kk, i);
    if (b < kFbBands) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        atomicAdd(&sh.a.re[b][16 * nt + j], (double)accr[nt][i]);
        atomicAdd(&sh.a.im[b][16 * nt + j], (double)acci[nt][i]);
      }
    }
  }
}

// which row tile a K step belongs to, where the tile starts and how many steps it has
struct FirSeg {
  int r, s0, n, d0;
  __device__ __forceinline__ FirSeg(int g, int g_end) {
    r = g >= kMfBase[2] ? 2 : g >= kMfBase[1] ? 1 : 0;
    const int base = r == 2 ? kMfBase[2] : r == 1 ? kMfBase[1] : 0;
    const int steps = r == 2 ? kMfSteps[2] : r == 1 ? kMfSteps[1] : kMfSteps[0];
    d0 = r == 2 ? kMfD0[2] : r == 1 ? kMfD0[1] : kMfD0[0];
    s0 = g - base;
    n = min(steps - s0, g_end - g);                  // K steps of this segment
  }
};

// The direct (folded) form for the FP64 engine's short filters: ONE row tile, bands 24 .. 39 (peaq_device.h kMfd*),
// 30 K steps of four delays.  The work is dealt to the four waves by OUTPUT -- wave = (real or imaginary part) x (time
// points 0 .. 31 or 32 .. 63) -- so that every element of A is one wave's own sequential sum over all K steps, written
// with a plain store: the result does not depend on which wave gets where first, and two runs on the same input agree
// bit for bit (the reference's sums are sequential, fbearmodel.c:399-435).  Until round 5 the K steps were cut into
// four runs whose partial sums met in A through LDS atomics, in whatever order the waves arrived.  Per K step a wave
// now needs one coefficient (its part's), four window reads and two additions for its two matrix instructions;
// the coefficients are requested eight steps ahead.
template <typename M>
__device__ __forceinline__ void fir_mfma_tail(BankLds<typename M::T>& sh, const typename M::T* __restrict__ mf_re,
                                              const typename M::T* __restrict__ mf_im, int wv, int lane) {
  typedef typename M::T T;
  typedef typename M::Acc Acc;
  const int j = lane & 15, kk = lane >> 4;
  const bool imag = wv & 1;                            // wave-uniform
  const int q0 = 2 * (wv >> 1);                        // the wave's two time tiles: q0, q0 + 1
  const Acc zero = {0, 0, 0, 0};
  Acc a0 = zero, a1 = zero;
  // (the address is hidden from the compiler: the coefficients are the same in every tile, and hoisted out of the
  // tile loop their registers -- with the other phases' -- end up in scratch)
  const T* cp = (imag ? mf_im : mf_re) + lane;
  asm volatile("" : "+v"(cp));
  constexpr int kAhead = 8;
  T h[kAhead];
#pragma unroll
  for (int q = 0; q < kAhead; ++q) h[q] = cp[64 * q];
  // window operands three K steps ahead of their use (a step is two matrix instructions = 128 cycles of the pipe, an
  // LDS round trip under load takes longer): step s reads x[-d - 4 s] and its mirror x[-(1458 - d - 4 s)], d = this
  // lane's delay in step 0, at the wave's two time tiles
  constexpr int kDepth = 3;
  T bx[kDepth][2], by[kDepth][2];
  // Eight steps are one window column (32 samples): the addresses of steps k, k + 8, k + 16, k + 24 differ by one
  // double each, so eight offsets per operand serve all thirty steps through the reads' immediate offsets (no address
  // arithmetic inside the loop; recomputed per tile rather than kept over it -- the kernel has no registers to spare).
  // o1[k]: x[-d - 4 k] three columns on (the delays grow, the columns fall: steps k + 8 m read at + (3 - m) doubles),
  // o2[k]: the mirror x[-(1458 - d - 4 k)] (columns rise: + m doubles).
  const T* o1[8];
  const T* o2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    o1[k] = sh.win.v + win_off(kFbRing - (kMfdD0 + 4 * k) - kk) + j + 16 * q0 - 3;
    o2[k] = sh.win.v + win_off(kMfdD0 + 4 * k - 2 + kk) + j + 16 * q0;
  }
  auto fetch = [&](int s, T (&x)[2], T (&y)[2]) {
    const int k = s & 7, m = s >> 3;
    x[0] = lds_rd(o1[k] + (3 - m));
    x[1] = lds_rd(o1[k] + (3 - m) + 16);
    y[0] = lds_rd(o2[k] + m);
    y[1] = lds_rd(o2[k] + m + 16);
  };
#pragma unroll
  for (int s = 0; s < kDepth; ++s) fetch(s, bx[s], by[s]);
#pragma unroll
  for (int s = 0; s < kMfdSteps; ++s) {
    const int sl = s % kDepth;
    const T o0 = imag ? bx[sl][0] - by[sl][0] : bx[sl][0] + by[sl][0];
    const T o1 = imag ? bx[sl][1] - by[sl][1] : bx[sl][1] + by[sl][1];
    if (s + kDepth < kMfdSteps) fetch(s + kDepth, bx[sl], by[sl]);
    const T c = h[s % kAhead];
    if (s + kAhead < kMfdSteps) h[s % kAhead] = cp[64 * (s + kAhead)];
    __builtin_amdgcn_sched_barrier(0);                 // keep the reads up here (the scheduler sinks them to their use)
    a0 = M::mma(c, o0, a0);
    a1 = M::mma(c, o1, a1);
  }
  double (*A)[kACols] = imag ? sh.a.im : sh.a.re;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = kMfdBand0 + M::row(kk, i);
    A[b][16 * q0 + j] = (double)a0[i];
    A[b][16 * (q0 + 1) + j] = (double)a1[i];
  }
}

// ---------------------------------------------------------------------------
// The block-sum form of the long filters, bands 0 .. 23 (peaq_device.h kBs*; fbearmodel.c:399-435 are the sums it
// evaluates).  A wave takes three PAIRS of neighbouring bands and, pair by pair, all on its own (no workgroup barrier):
//   tile     D[16 rows][64 columns] = coef[16][32] x (32 samples x 64 window columns from bs_col_head on): per band
//            the three exponentials' ENTER rows (re, im) and the filter's own coefficients on the block its window
//            ENDS in -- 32 matrix instructions, written to the wave's staging rows by output;
//   sums     a lane per (chain, segment of kBsSeg = 9 consecutive outputs), eight lanes per chain, the pair's six
//            chains at once: d(t) = enter(t) - rot^J enter(t - J) from the staging rows (what leaves: from the history,
//            FbSignalState::bs_hist, for the segments in front of output J -- the segments are shifted so that output
//            J starts one), V(t) = rot V(t-1) + d(t) once with nothing carried in, a weighted DPP scan over the chain's
//            eight segments, V(-1)'s share, then the recurrence again; the sums are written over the enter values.
//            (First version: lanes = outputs and a wave-wide prefix scan per chain in the frame of output -1 -- 64 vector
//            instructions per chain; bs_chain is still that, for V(-1) on a launch's first tile.)
//   left     lanes = outputs again: the filter's own coefficients on the block the window STARTS in, at most 16 taps on
//            the vector ALU (FbTables::bs_left_g; two rows of sixteen would fill a matrix tile to a quarter), and
//            y(t) = V_0 + V_1 + V_2 + the two edge blocks from the rows.
// The running sums V are carried from tile to tile in LDS (vst); the first tile of a launch computes them from the
// history (V(-1) = sum_s rot^s enter(-1 - s): one more scan) -- so the only state is the history, which is exact:
// rounding errors of the sums cannot travel further than a launch, and nothing is left of a sample 1456 samples
// after it, as in the reference's delay line.
// What a lane reads of its neighbour chain in the scan is multiplied by zero, not selected away: every value a lane can
// reach has to be FINITE -- outputs that do not exist included (zeroed slots in bs_pair, A zeroed once per workgroup,
// a window that ends with the signal's own blocks: fb_bank_body, row_valid).
// ---------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) double kdouble;   // read through the scalar cache when the address is uniform
typedef const __attribute__((address_space(4))) int kint;
constexpr int kLogTabAt = 512;                       // FP64 engine: the logarithm table's place in e1 / ex (doubles from e1[0][0]), behind the
                                                     // left-edge coefficients / the slopes of bands 32 .. 39 (512 doubles each in their phase)
constexpr int kStRow = 72;                           // staging row stride in doubles: eight front slots + 64 outputs (= 8 mod 32: the rows of
                                                     // the four chains of a half wave start on four different bank groups)
constexpr int kStOrg = 8;                            // index of output 0 in a row (entries in front: outputs < 0 of the shifted rows)
constexpr int kStWave = 1280;                        // doubles per wave (16 rows + the last row's reads beyond output 63)
static_assert(16 * kStRow + 16 <= kStWave, "rows + overhang");
static_assert(4 * kStWave <= 2 * kFbBands * kACols, "the staging rows of the four waves live where A will be");

// the A operands of a pair's tile (eight K steps): requested a pair ahead of their use
__device__ __forceinline__ void bs_coef_load(const FbTables* __restrict__ fb, int p, int lane, double (&a)[8]) {
  const double* coef = fb->bs_coef[p < kBsPairs ? p : kBsPairs - 1][0] + lane;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) a[ks] = coef[ks * 64];
}

__device__ __forceinline__ void bs_tile(const double* __restrict__ win, const double (&a)[8], int col0, int lane, v4d (&acc)[4]) {
  const int j = lane & 15, kk = lane >> 4;
  // B operand of K step ks, column tile nt: samples 4 ks + kk of the blocks col0 + 16 nt + j (rows of the window
  // array are samples-in-block, its columns blocks: consecutive lanes read consecutive doubles)
  const double* p = win + kk * kWinRow + col0 + j;
  const v4d zero = {0, 0, 0, 0};
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) acc[nt] = zero;
  double b[4], nb[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) b[nt] = lds_rd(p + 16 * nt);
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    if (ks < 7) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) nb[nt] = lds_rd(p + 4 * (ks + 1) * kWinRow + 16 * nt);
    }
    __builtin_amdgcn_sched_barrier(0);               // the next step's reads stay ahead of this step's matrix instructions
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[nt], acc[nt], 0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) b[nt] = nb[nt];
  }
}

// The running sums of one exponential over the wave's outputs: lane t gets r^(t+1) v + sum_{s <= t} r^(t - s) u_s
// from (pr, pi) = r^(lane + 1) alone: every input is turned back into the frame of output -1 (times conj r^(s+1)),
// the frame's plain prefix sum is taken (DPP row shifts and row broadcasts) and
// turned forward again.  |r| = 1: the turns cost no accuracy.
__device__ __forceinline__ void bs_chain(double& re, double& im, double vr, double vi, double pr, double pi, int lane,
                                         double& z15, double& z31) {
  double ar = fma(pr, re, pi * im), ai = fma(pr, im, -pi * re);       // conj(p) u
#define PEAQ_BS_LEVEL(SH)                    \
  ar += dpp_d0<kDppRowShr + SH>(ar);         \
  ai += dpp_d0<kDppRowShr + SH>(ai);
  PEAQ_BS_LEVEL(1)
  PEAQ_BS_LEVEL(2)
  PEAQ_BS_LEVEL(4)
  PEAQ_BS_LEVEL(8)
#undef PEAQ_BS_LEVEL
  // what came before a row: the rows in front of it (two DPP broadcasts, peaq_wave.h) and v
  ar += row_carry_15(ar, z15);
  ai += row_carry_15(ai, z15);
  ar += row_carry_31(ar, z31) + vr;
  ai += row_carry_31(ai, z31) + vi;
  re = fma(pr, ar, -pi * ai);                                         // p (...)
  im = fma(pr, ai, pi * ar);
}

// eight taps (samples 8 G .. 8 G + 7 of the block) of a window's first block.  The coefficients (re, im per tap) sit
// in the wave's corner of LDS (bs_pair); every lane reads the same 16 bytes -- a broadcast, no bank conflict -- and
// its own sample of the block: two LDS reads and two multiply-adds per tap.  (From the scalar cache every tap
// waited for its own load; as DPP row broadcasts of a register each product took a move and a multiply-add.)
template <int G>
__device__ __forceinline__ void bs_left_group(const double* __restrict__ xl, const double* __restrict__ cl, double& sr,
                                              double& si) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  double x[8];
  v2d c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    x[k] = lds_rd(xl + (8 * G + k) * kWinRow);
    c[k] = *(const __attribute__((address_space(3))) v2d*)(cl + 2 * (8 * G + k));
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    sr = fma(c[k].x, x[k], sr);
    si = fma(c[k].y, x[k], si);
  }
}

#ifdef PEAQ_FB_PROFILE
__device__ unsigned long long g_fb_prof[4 * 16 + 4];
#endif
// -DPEAQ_FB_SUBPROF (with -DPEAQ_FB_PROFILE): the steps of bs_pair are timed into slots 5 .. 11 in place of the tile's later phases
#ifdef PEAQ_FB_SUBPROF
#define BS_MARK(i)                                                                       \
  do {                                                                                   \
    const unsigned long long now_ = __builtin_readcyclecounter();                        \
    if (lane == 0 && (blockIdx.x & 15) == 5) atomicAdd(&g_fb_prof[(threadIdx.x >> 6) * 16 + (i)], now_ - bs_t_); \
    bs_t_ = __builtin_readcyclecounter();                                                \
  } while (0)
#else
#define BS_MARK(i) do { } while (0)
#endif
// one pair of bands (2 p, 2 p + 1): their filter outputs at the tile's outputs lane = 0 .. 59 -> (yr, yi)[band in pair];
// a holds the pair's A operands on entry and the next pair's on return
__device__ __forceinline__ void bs_pair(const double* __restrict__ win, double* __restrict__ stg, double (*vst)[2],
                                        double* __restrict__ lcoef, const FbTables* __restrict__ fb, double* __restrict__ hist,
                                        int p, bool first_tile, int nvs, int lane, double (&a)[8], double (&yr)[2],
                                        double (&yi)[2]) {
  typedef double v2d __attribute__((ext_vector_type(2)));
  const int j = lane & 15, kk = lane >> 4;
  // the tables' uniform entries travel through the scalar cache (constant address space)
  kint* t_head = (kint*)(const void*)fb->bs_col_head;
  kint* t_off = (kint*)(const void*)fb->bs_off_enter;
  kint* t_left = (kint*)(const void*)fb->bs_col_left;
  kint* t_lg = (kint*)(const void*)fb->bs_left_g;
  kint* t_whole = (kint*)(const void*)fb->bs_whole;
  kint* t_segs = (kint*)(const void*)fb->bs_seg_s;
#ifdef PEAQ_FB_SUBPROF
  unsigned long long bs_t_ = __builtin_readcyclecounter();
#endif
  const int col_head = t_head[p];
  const int J0 = t_whole[2 * p], J1 = t_whole[2 * p + 1];
  // the lane's part in the running sums: chain c of the pair (3 * band + exponential; lanes 48 .. 63 repeat chain 5),
  // segment g = outputs t0 .. t0 + 8
  const int c = min(lane >> 3, 5), g = lane & 7;
  const bool second = c >= 3;
  const int Jl = second ? J1 : J0;
  const int t0 = kBsSeg * g - (second ? t_segs[2 * p + 1] : t_segs[2 * p]);
  const bool from_hist = t0 < Jl;                      // (output J starts a segment: all of this one leaves from the history)
  // what leaves the sums at the lane's outputs if that is history: requested first, used after the matrix work (the
  // other lanes read valid memory too and replace it below)
  double lr[kBsSeg], li[kBsSeg];
  {
    const double* hp = hist + ((size_t)(2 * p + (second ? 1 : 0)) * 6 + 2 * (c - (second ? 3 : 0))) * kBsHist + kBsHistOrg +
                       (from_hist ? t0 : 0);
#pragma unroll
    for (int k = 0; k < kBsSeg; ++k) {
      lr[k] = hp[k];
      li[k] = hp[kBsHist + k];
    }
  }
  double cl[2];                                        // the first blocks' coefficients (bs_left): 64 doubles per band
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) cl[sub] = (&fb->bs_left[2 * p + sub][0][0])[lane];
  // the chain's constants (FbTables::bs_seg): scan weights, rot^(9 g - s), rot, rot^J
  v2d w1, w2, w4, kp, rot, rj;
  {
    const v2d* sg = reinterpret_cast<const v2d*>(&fb->bs_seg[p][0][lane][0]);
    w1 = sg[0];
    w2 = sg[64];
    w4 = sg[128];
    kp = sg[192];
    rot = sg[256];
    rj = sg[320];
  }
  // staging index of accumulator element i (row kk + 4 i of the tile, bs_row) of column tile 0: a row's entries sit at
  // their OUTPUT's index (the rows of the second band and of the end blocks start some columns later)
  int ih[4];
  {
    const int off0 = t_off[2 * p], off1 = t_off[2 * p + 1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = kk + 4 * i;
      const bool sb = i == 3 ? (row & 1) : (i == 0 ? row >= 3 : (i == 1 ? row < 6 : row >= 9));   // the row's band in the pair
      ih[i] = row * kStRow + kStOrg + j - (sb ? off1 : off0) - (i == 3 ? 1 : 0);
    }
  }
  {
    v4d acc[4];
    bs_tile(win, a, col_head, lane, acc);
    bs_coef_load(fb, p + 4, lane, a);                  // the wave's next pair (the last one's request is not used)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int i = 0; i < 4; ++i) stg[ih[i] + 16 * nt] = acc[nt][i];
  }
  // the eight slots in front of output 0 of the twelve enter rows read as zero (outputs in front of the first segment;
  // the shifted rows of the tile have just put their columns -3 .. -1 there)
  // -- of all sixteen rows: the last enter row's outputs 64 .. 71 are the first end-block row's front slots, and what
  // a lane computes from them is multiplied by zero in its neighbour chain's scan: it has to be finite
  stg[(lane >> 3) * kStRow + (lane & 7)] = 0.;
  stg[(8 + (lane >> 3)) * kStRow + (lane & 7)] = 0.;
  // ... and so have outputs 61 .. 63 (nobody's, but inside the last segment), which a row shifted by two or three
  // columns does not get from the tile: they would be whatever phase of the previous tile used this part of A last
  if (lane < 48) stg[(lane / 3) * kStRow + kStOrg + 61 + lane % 3] = 0.;
  lcoef[lane] = cl[0];                                 // the first blocks' coefficients where every lane can read them
  lcoef[64 + lane] = cl[1];
  wave_lds_fence();
  BS_MARK(5);   // loads issued, tile, staging writes
  if (first_tile) {
    // a launch's first tile: V(-1) of the six chains from the history, lanes = history entries (one prefix scan each)
    double z15 = 0., z31 = 0.;
#pragma unroll 1
    for (int c2 = 0; c2 < 6; ++c2) {
      const int sub = c2 >= 3 ? 1 : 0, J = sub ? J1 : J0;
      const double* hp = hist + ((size_t)(2 * p + sub) * 6 + 2 * (c2 - 3 * sub)) * kBsHist + kBsHistOrg + min(lane, kBsHist - kBsHistOrg - 1);
      double ar = hp[0], ai = hp[kBsHist];
      const double2 pw = *reinterpret_cast<const double2*>(fb->bs_pow[6 * p + c2][lane]);
      bs_chain(ar, ai, 0., 0., pw.x, pw.y, lane, z15, z31);
      const double v_r = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ar), J - 1),
                                          __builtin_amdgcn_readlane(__double2loint(ar), J - 1));
      const double v_i = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(ai), J - 1),
                                          __builtin_amdgcn_readlane(__double2loint(ai), J - 1));
      if (lane == 0) {
        vst[6 * p + c2][0] = v_r;
        vst[6 * p + c2][1] = v_i;
      }
    }
    wave_lds_fence();
  }
  // ---- the running sums, a lane per (chain, segment) -----------------------------------------------------------------
  double* erow = stg + c * kStRow + kStOrg + t0;       // the chain's enter values (re; im six rows on) at the lane's outputs
  if (!from_hist) {                                    // what leaves is this tile's: the entries J outputs back
#pragma unroll
    for (int k = 0; k < kBsSeg; ++k) {
      lr[k] = erow[k - Jl];
      li[k] = erow[6 * kStRow + k - Jl];
    }
  }
  double dr[kBsSeg], di[kBsSeg];                       // enter(t) - rot^J enter(t - J)
#pragma unroll
  for (int k = 0; k < kBsSeg; ++k) {
    dr[k] = fma(-rj.x, lr[k], fma(rj.y, li[k], erow[k]));
    di[k] = fma(-rj.x, li[k], fma(-rj.y, lr[k], erow[6 * kStRow + k]));
  }
  BS_MARK(6);   // first-tile branch, d
  // the history after this tile: lane i < J <- enter(nvs - J + i) -- before the sums take the rows' place.  (Two copies
  // of the loop: a tile shorter than J -- the last of a launch at most -- keeps part of the old history and has to load
  // it; in one copy, the wait for that load would stand in every tile's path, and it waits for ALL loads in flight,
  // the next pair's coefficients included.)
  if (nvs >= max(J0, J1)) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int J = sub ? J1 : J0;
      const double* src = stg + kStOrg + nvs - J + lane;
      double hn[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) hn[r] = src[bs_row(sub, r) * kStRow];
      double* hrow = hist + (size_t)(2 * p + sub) * 6 * kBsHist + kBsHistOrg;
      if (lane < J) {
#pragma unroll
        for (int r = 0; r < 6; ++r) hrow[r * kBsHist + lane] = hn[r];
      }
    }
  } else {
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      const int J = sub ? J1 : J0;
      const int src = nvs - J + lane;
      double* hrow = hist + (size_t)(2 * p + sub) * 6 * kBsHist + kBsHistOrg;
#pragma unroll 1
      for (int r = 0; r < 6; ++r) {
        const double fresh = stg[bs_row(sub, 0) * kStRow + (r >> 1) * kStRow + (r & 1) * 6 * kStRow + kStOrg + max(src, 0)];
        const double old = hrow[r * kBsHist + min(lane + nvs, kBsHist - kBsHistOrg - 1)];
        const double v = src >= 0 ? fresh : old;
        if (lane < J) hrow[r * kBsHist + lane] = v;
      }
    }
  }
  BS_MARK(7);   // history update
  // the segment's sum with nothing carried in (Horner in rot), ...
  double tr = dr[0], ti = di[0];
#pragma unroll
  for (int k = 1; k < kBsSeg; ++k) {
    const double nr = fma(rot.x, tr, fma(-rot.y, ti, dr[k]));
    ti = fma(rot.x, ti, fma(rot.y, tr, di[k]));
    tr = nr;
  }
  // ... the scan over the chain's eight segments (weights rot^9, rot^18, rot^36; zero where the source would be another
  // chain's lane) ...
#define PEAQ_BS_SCAN(SH, W)                                          \
  {                                                                  \
    const double sr_ = dpp_d0<kDppRowShr + SH>(tr), si_ = dpp_d0<kDppRowShr + SH>(ti); \
    tr = fma(W.x, sr_, fma(-W.y, si_, tr));                          \
    ti = fma(W.x, si_, fma(W.y, sr_, ti));                           \
  }
  PEAQ_BS_SCAN(1, w1)
  PEAQ_BS_SCAN(2, w2)
  PEAQ_BS_SCAN(4, w4)
#undef PEAQ_BS_SCAN
  // ... what the lane starts from: the sums at the end of the segment in front of it, plus what V(-1) has become there
  double vr, vi;
  {
    const double pr_ = dpp_d0<kDppRowShr + 1>(tr), pi_ = dpp_d0<kDppRowShr + 1>(ti);
    const double m = g > 0 ? 1. : 0.;
    const v2d v = *(const __attribute__((address_space(3))) v2d*)&vst[6 * p + c][0];
    vr = fma(m, pr_, fma(kp.x, v.x, -kp.y * v.y));
    vi = fma(m, pi_, fma(kp.x, v.y, kp.y * v.x));
  }
  BS_MARK(8);   // Horner, scan, carry-in
  // ... and the sums themselves, written over the enter values
#pragma unroll
  for (int k = 0; k < kBsSeg; ++k) {
    const double nr = fma(rot.x, vr, fma(-rot.y, vi, dr[k]));
    vi = fma(rot.x, vi, fma(rot.y, vr, di[k]));
    vr = nr;
    erow[k] = vr;
    erow[6 * kStRow + k] = vi;
  }
  wave_lds_fence();                                    // (other lanes read the sums from here on)
  if (lane < 6) {                                      // carried to the next tile: the sums at the last valid output
    vst[6 * p + lane][0] = stg[lane * kStRow + kStOrg + nvs - 1];
    vst[6 * p + lane][1] = stg[(6 + lane) * kStRow + kStOrg + nvs - 1];
  }
  BS_MARK(9);   // final pass, V writes, vst
  // ---- lanes = outputs again: the block a window starts in (the filter's own coefficients from q0 on), the block it
  // ends in (rows 12 .. 15 of the tile) and the three sums ----------------------------------------------------------------
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int b = 2 * p + sub;
    double sr = 0., si = 0.;
    {
      const double* xl = win + t_left[b] + lane;
      const int ga = t_lg[2 * b], gb = t_lg[2 * b + 1];   // the groups of eight taps that count (at most two, FbTables::bs_left_g)
      const double* lc = lcoef + 64 * sub;
      if (ga <= 0 && 0 < gb) bs_left_group<0>(xl, lc, sr, si);
      if (ga <= 1 && 1 < gb) bs_left_group<1>(xl, lc, sr, si);
      if (ga <= 2 && 2 < gb) bs_left_group<2>(xl, lc, sr, si);
      if (3 < gb) bs_left_group<3>(xl, lc, sr, si);
    }
    const double* col = stg + kStOrg + lane;
    sr += col[(12 + sub) * kStRow];
    si += col[(14 + sub) * kStRow];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      sr += col[(3 * sub + e) * kStRow];
      si += col[(6 + 3 * sub + e) * kStRow];
    }
    yr[sub] = sr;
    yi[sub] = si;
  }
  BS_MARK(10);  // left edges, assembly
  wave_lds_fence();                                    // the next pair's tile overwrites the staging rows
}

// The FP32 instruction's loop, trimmed to what the matrix pipe needs per K step: four ds_read2_b32 (two
// time tiles each), eight packed FP32 adds for the folded operands, two address updates -- the LDS offsets
// of eight consecutive K steps (32 delays = one window column) are kept per lane and move by one column
// per round -- and the two coefficient loads, requested eight steps ahead into the register just used.
__device__ __forceinline__ void fir_mfma_f32(BankLds<float>& sh, const float* __restrict__ mf_re,
                                             const float* __restrict__ mf_im, int wv, int lane) {
  typedef MfmaF32 M;
  typedef v4f Acc;
  const int j = lane & 15, kk = lane >> 4;
  int g = wv == 0 ? 0 : 1 + 65 * wv;
  const int g_end = 66 + 65 * wv;
  while (g < g_end) {
    const FirSeg sg(g, g_end);
    const int n = sg.n;
    const Acc zero = {0, 0, 0, 0};
    Acc ar0 = zero, ar1 = zero, ar2 = zero, ar3 = zero, ai0 = zero, ai1 = zero, ai2 = zero, ai3 = zero;
    const int d = sg.d0 + 4 * sg.s0 + kk;
    int o1[8], o2[8];                                // float offsets of x[-d - 4 k], x[-(1458 - d - 4 k)] at t = j
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o1[k] = win_off(kFbRing - d - 4 * k) + j;
      o2[k] = win_off(d - 2 + 4 * k) + j;
    }
    // coefficient of K step s (clamped: the requests run up to eight steps past the end of the run)
    auto coef = [&](const float* __restrict__ tab, int s) {
      return tab[(size_t)min(g + s, kMfTotalSteps - 1) * 64 + lane];
    };
    float hr[8], hi[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      hr[k] = coef(mf_re, k);
      hi[k] = coef(mf_im, k);
    }
    v2f xa, xb, ya, yb;                              // the operands of the step about to run
    auto fetch = [&](int k) {
      const float* p1 = sh.win.v + o1[k];
      const float* p2 = sh.win.v + o2[k];
      xa = v2f{p1[0], p1[16]};
      xb = v2f{p1[32], p1[48]};
      ya = v2f{p2[0], p2[16]};
      yb = v2f{p2[32], p2[48]};
      o1[k] -= 1;                                    // eight steps on: 32 delays = one column
      o2[k] += 1;
    };
    fetch(0);
    auto mma8 = [&](float cr, float ci, v2f sa, v2f da, v2f sb, v2f db) {
      ar0 = M::mma(cr, sa.x, ar0);
      ai0 = M::mma(ci, da.x, ai0);
      ar1 = M::mma(cr, sa.y, ar1);
      ai1 = M::mma(ci, da.y, ai1);
      ar2 = M::mma(cr, sb.x, ar2);
      ai2 = M::mma(ci, db.x, ai2);
      ar3 = M::mma(cr, sb.y, ar3);
      ai3 = M::mma(ci, db.y, ai3);
    };
    int s = 0;
    for (; s + 8 <= n; s += 8) {                     // full rounds: no branches inside
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const v2f sa = xa + ya, da = xa - ya, sb = xb + yb, db = xb - yb;
        const float cr = hr[k], ci = hi[k];
        fetch((k + 1) & 7);
        hr[k] = coef(mf_re, s + k + 8);
        hi[k] = coef(mf_im, s + k + 8);
        __builtin_amdgcn_sched_barrier(0);           // reads and requests stay ahead of the matrix instructions
        mma8(cr, ci, sa, da, sb, db);
      }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {                    // the last n mod 8 steps (wave-uniform branches)
      if (s + k < n) {
        const v2f sa = xa + ya, da = xa - ya, sb = xb + yb, db = xb - yb;
        fetch(k + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma8(hr[k], hi[k], sa, da, sb, db);
      }
    }
    const Acc accr[4] = {ar0, ar1, ar2, ar3}, acci[4] = {ai0, ai1, ai2, ai3};
    fir_store<M>(sh, sg.r, j, kk, accr, acci);
    g += n;
  }
}

// The bank on v_mfma_f32_16x16x32_f16 (peaq_device.h kHf*): per block of 32 delays and time tile twelve
// instructions -- {re, im} x {X1, X2} x {hi hi, hi lo, lo hi} -- at 17 cycles each against eight FP32 ones at 32
// for FOUR delays, and no vector arithmetic on the operands at all: a B operand is one 16-byte LDS read of
// the split window (lane = time point + 16 x delay group: eight consecutive samples), an A operand one
// 16-byte read of the coefficient table through L2, reused for the four time tiles.  The 34 blocks are cut
// into four runs (9 + 9 + 8 + 8), one per wave; a run's tiles are scaled back (power-of-two factors of
// signal and band) and added into A with LDS atomics like the FP32 form's.
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void fir_mfma_h3(BankLds<_Float16>& sh, const FbTables* __restrict__ fb, double xunscale, int wv,
                                            int lane) {
  typedef v4f Acc;
  const int j = lane & 15, kg = lane >> 4;
  int g = wv < 2 ? 9 * wv : 18 + 8 * (wv - 2);
  const int g_end = g + (wv < 2 ? 9 : 8);
  static_assert(9 + 9 + 8 + 8 == kHfTotalBlocks, "split of the blocks over the four waves");
  typedef const __attribute__((address_space(3))) v8h* lds_v8h;
  const lds_v8h whi = (lds_v8h)sh.win.hi, wlo = (lds_v8h)sh.win.lo;
  while (g < g_end) {
    const int r = g >= kHfBase[2] ? 2 : g >= kHfBase[1] ? 1 : 0;
    const int base = r == 2 ? kHfBase[2] : r == 1 ? kHfBase[1] : 0;
    const int blocks = r == 2 ? kHfBlocks[2] : r == 1 ? kHfBlocks[1] : kHfBlocks[0];
    const int d1 = r == 2 ? kHfD1[2] : r == 1 ? kHfD1[1] : kHfD1[0];
    const int d2 = r == 2 ? kHfD2[2] : r == 1 ? kHfD2[1] : kHfD2[0];
    const int s0 = g - base;
    const int n = min(blocks - s0, g_end - g);
    const Acc zero = {0, 0, 0, 0};
    Acc ar[4] = {zero, zero, zero, zero}, ai[4] = {zero, zero, zero, zero};
    // window positions of this lane's eight samples at time tile 0 (16-byte units: the arrays are read as v8h);
    // a block further on X1 moves one 80-byte block down, X2 one up; a time tile is 16 blocks (1280 bytes) up
    int p1 = winh_off(kFbRing - (d1 + 32 * s0 + 8 * kg + 7) + 32 * j) / 16;
    int p2 = winh_off(d2 + 32 * s0 + 8 * kg - 2 + 32 * j) / 16;
    // (Requesting the coefficients a block ahead and the window operands a time tile ahead was measured: 5.40
    // against 5.76 M frame-pairs/s -- the second wave of the SIMD already covers those latencies, the extra
    // live registers only cost.)
    const v8h* __restrict__ tab = reinterpret_cast<const v8h*>(&fb->hf[g][0][lane][0]);   // [block][operand][lane]
    for (int s = 0; s < n; ++s) {
      v8h h[HF_OPERANDS];
#pragma unroll
      for (int o = 0; o < HF_OPERANDS; ++o) h[o] = tab[(s * HF_OPERANDS + o) * 64];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const v8h x1h = whi[p1 + 80 * nt], x1l = wlo[p1 + 80 * nt], x2h = whi[p2 + 80 * nt], x2l = wlo[p2 + 80 * nt];
        Acc cr = ar[nt], ci = ai[nt];
        cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_RE_HI_1], x1h, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_IM_HI_1], x1h, ci, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_RE_HI_2], x2h, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_NIM_HI_2], x2h, ci, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_RE_HI_1], x1l, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_IM_HI_1], x1l, ci, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_RE_HI_2], x2l, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_NIM_HI_2], x2l, ci, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_RE_LO_1], x1h, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_IM_LO_1], x1h, ci, 0, 0, 0);
        cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_RE_LO_2], x2h, cr, 0, 0, 0);
        ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(h[HF_NIM_LO_2], x2h, ci, 0, 0, 0);
        ar[nt] = cr;
        ai[nt] = ci;
      }
      p1 -= 5;                                       // 80 bytes
      p2 += 5;
    }
    // D layout: column = lane & 15 (time), row = 4 (lane >> 4) + i  ->  band 16 r + row
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = 16 * r + 4 * kg + i;
      if (b < kFbBands) {
        const double us = fb->hf_unscale[b] * xunscale;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          atomicAdd(&sh.a.re[b][16 * nt + j], (double)ar[nt][i] * us);
          atomicAdd(&sh.a.im[b][16 * nt + j], (double)ai[nt][i] * us);
        }
      }
    }
    g += n;
  }
}

// the ten bands a wave carries through phases 2..: { w, 7-w, 8+w, 15-w, 16+w, 23-w, 24+w, 31-w, 32+w, 39-w }
__host__ __device__ constexpr int wave_band(int w, int i) { return (i & 1) ? 8 * (i >> 1) + 7 - w : 8 * (i >> 1) + w; }

// upward spreading of wave W's ten source bands (ascending: wave_band(W, 0) < ... < wave_band(W, 9)):
// target band j receives sum_{sources b < j} A[b] cu_b^(j-b); everything about the band indices
// is a compile-time constant, the code is a straight line of multiplies, adds and 78 atomics
template <int W, typename WT>
__device__ __forceinline__ void spread_up(BankLds<WT>& sh, const double (&re)[10], const double (&im)[10],
                                          const double (&cu)[10], int lane) {
  double tr[10], ti[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    tr[i] = re[i];
    ti[i] = im[i];
  }
#pragma unroll
  for (int j = 1; j < kFbBands; ++j) {
    double sr = 0., si = 0.;
    bool any = false;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      if (wave_band(W, i) < j) {                     // compile time
        tr[i] *= cu[i];
        ti[i] *= cu[i];
        sr += tr[i];
        si += ti[i];
        any = true;
      }
    }
    if (any) {
      atomicAdd(&sh.a.re[j][lane], sr);
      atomicAdd(&sh.a.im[j][lane], si);
    }
  }
}

// The FP64 engine's upward spreading, dealt to the waves by TARGET: wave W owns the bands j = W, W + 4, W + 8, ...
// (j >= 1) and sums, per time point = lane, what every lower band sends there -- all sources' outputs and slopes come
// from LDS (A, and the slope exchange cux), the ten sums stay in registers until every wave has read its sources.
// One wave, one sum in one fixed order per element: the result does not depend on the order in which the waves run, and
// two RUNS agree bit for bit (that, not more: against the reference's loop, fbearmodel.c:340-348, the order differs --
// a band's own output joins its sum when its turn as a source comes, after the lower bands' terms, and a source's
// powers are formed as c^2, c^4, c^2 c and steps of c^4 rather than by repeated multiplication -- so the last bits
// differ from the oracle's as any other reordering's would; the tests hold the blocks to 1e-9).  spread_up above has
// the four waves add their partial sums into A with LDS atomics in whatever order they arrive (kept for the
// reduced-precision engines).  A source's terms at the owned targets are
// four bands apart: its tail starts at cu^r (r = 1 .. 4 bands up to the first owned target) and moves on by cu^4.
template <int W, typename CUX>
__device__ __forceinline__ void spread_up_owned(const double (*are)[kACols], const double (*aim)[kACols], CUX cux, int lane,
                                                double (&acr)[10], double (&aci)[10]) {
#pragma unroll
  for (int n = 0; n < 10; ++n) acr[n] = aci[n] = 0.;
  // The sources in batches of kBatch, a batch's reads issued while the batch before it is worked on (read - compute
  // per source, every source waited for its own LDS round trip: 39 of them per tile).  A source that is one of the
  // wave's own targets also starts that target's sum with its own output: what leaves this function is the new A.
  constexpr int kBatch = 4, kSrc = kFbBands;         // (band 39 sends nothing, but its output starts wave 3's last sum)
  constexpr int kBatches = (kSrc + kBatch - 1) / kBatch;
  double c1[2][kBatch], sr[2][kBatch], si[2][kBatch];
  auto request = [&](int g, int buf) {
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int b = g * kBatch + k < kSrc ? g * kBatch + k : kSrc - 1;
      c1[buf][k] = b < kFbBands - 1 ? cux(b, lane) : 0.;
      sr[buf][k] = lds_rd(&are[b][lane]);
      si[buf][k] = lds_rd(&aim[b][lane]);
    }
  };
  request(0, 0);
#pragma unroll
  for (int g = 0; g < kBatches; ++g) {
    const int buf = g & 1;
    if (g + 1 < kBatches) request(g + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);               // (the scheduler would sink the next batch's reads to their use)
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      const int b = g * kBatch + k;
      if (b >= kSrc) continue;
      if (b >= 1 && b % 4 == W) {                    // own target: its sum starts from its own output
        const int n = (b - (W == 0 ? 4 : W)) / 4;
        acr[n] += sr[buf][k];
        aci[n] += si[buf][k];
      }
      const int j0 = b + 1 + ((W - (b + 1)) % 4 + 4) % 4;        // the first owned target above b (compile time)
      if (j0 < kFbBands) {
        const int r = j0 - b;
        const double c = c1[buf][k], c2 = c * c, c4 = c2 * c2;
        // (the POWER walks on, not the two products as in the reference's loop, fbearmodel.c:343-348: one multiply and
        // two multiply-adds per (source, target) pair instead of two multiplies and two additions; the same sums in
        // another association -- last bits, like the rest of this function's order)
        double pw = r == 1 ? c : r == 2 ? c2 : r == 3 ? c2 * c : c4;
        const double ar = sr[buf][k], ai = si[buf][k];
#pragma unroll
        for (int j = j0; j < kFbBands; j += 4) {
          const int n = (j - (W == 0 ? 4 : W)) / 4;               // slot of target j among the wave's targets
          acr[n] = fma(pw, ar, acr[n]);
          aci[n] = fma(pw, ai, aci[n]);
          if (j + 4 < kFbBands) pw *= c4;
        }
      }
    }
  }
}

// Phase timing for tools/fb_profile.py (development builds with -DPEAQ_FB_PROFILE only): cycles between
// consecutive marks, summed per wave role over one sampled workgroup in 16.
#ifdef PEAQ_FB_PROFILE
#ifdef PEAQ_FB_SUBPROF
#define FB_SLOT_OK(i) ((i) < 5 || (i) > 10)
#else
#define FB_SLOT_OK(i) true
#endif
#define FB_MARK(i)                                                                       \
  do {                                                                                   \
    const unsigned long long now_ = __builtin_readcyclecounter();                        \
    if (FB_SLOT_OK(i) && lane == 0 && (blockIdx.x & 15) == 5) atomicAdd(&g_fb_prof[wv * 16 + (i)], now_ - prof_t_); \
    prof_t_ = __builtin_readcyclecounter();                                              \
  } while (0)
#else
#define FB_MARK(i) do { } while (0)
#endif

// The same in packed FP32 for the reduced-precision engine (split-FP16 FIR): (re, im) of a source travel as one
// register pair, a step is one v_pk_mul_f32 and one v_pk_add_f32 at two cycles each instead of four FP64
// instructions at four; the per-wave sums leave as FP64 atomics (ds_add_f32 is 22 x slower, see BankLds).
template <int W, typename WT>
__device__ __forceinline__ void spread_up_f32(BankLds<WT>& sh, const double (&re)[10], const double (&im)[10],
                                              const float (&cu)[10], int lane) {
  v2f t[10], c[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    t[i] = v2f{(float)re[i], (float)im[i]};
    c[i] = v2f{cu[i], cu[i]};
  }
#pragma unroll
  for (int j = 1; j < kFbBands; ++j) {
    v2f sum = {0.f, 0.f};
    bool any = false;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      if (wave_band(W, i) < j) {                     // compile time
        t[i] *= c[i];
        sum += t[i];
        any = true;
      }
    }
    if (any) {
      atomicAdd(&sh.a.re[j][lane], (double)sum.x);
      atomicAdd(&sh.a.im[j][lane], (double)sum.y);
    }
  }
}

template <typename M>
__device__ __forceinline__ void fb_bank_body(const FbFrontArgs& a, unsigned n_signals) {
  typedef typename M::T WT;
  __shared__ BankLds<WT> sh;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: say so
  const unsigned g = blockIdx.x;
  const int sig = g & 1;
  const int chan = (g >> 1) % a.channels;
  const unsigned pair = g / (2 * a.channels);
  unsigned nb_mine, state_idx = g;
  if (a.windows) {                                   // broker launch: this session's own window and state
    const FbPairWindow w = a.windows[pair];
    nb_mine = w.n_blocks;
    state_idx = (w.slot * a.channels + chan) * 2 + sig;
  } else {
    const unsigned n_blocks = a.n_blocks ? a.n_blocks[pair] : a.n_blocks_uniform;
    nb_mine = n_blocks > a.block0 ? min(a.blocks_per_launch, n_blocks - a.block0) : 0;
  }
  if (nb_mine == 0) return;
#ifdef PEAQ_FB_STAGGER
  // the two workgroups of a CU run the same code from the same start: without an offset they sit in the
  // FIR phase (matrix pipe) together and in the vector phases together.  The second workgroup of every CU
  // in the first dispatch round starts half a tile late; equal run times then keep the pairs staggered.
  if (blockIdx.x >= 256 && blockIdx.x < 512)
    for (int i = 0; i < PEAQ_FB_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
  const BandTables* __restrict__ bt = a.bands;
  const FbTables* __restrict__ fb = a.fb;
  const size_t row_len = a.hp_row_stride;
  // what the high-pass walk has written of this signal's row in this launch: its OWN blocks.  (Not the launch's:
  // behind a shorter session's or pair's blocks the row holds whatever the memory held before -- outputs that do
  // not exist are computed from it, and the FP64 engine's running sums need those to be finite, bs_pair.)
  const size_t row_valid = (size_t)kFbRing + (size_t)nb_mine * kFbFrame;
  const double* __restrict__ row = a.hp_scratch + (size_t)state_idx * row_len;
  FbSignalState* __restrict__ st = a.fbstate + state_idx;
  // Scale of the split-FP16 operands: the launch's power of two (full scale at 2^10..2^11, 30 dB of headroom)
  // unless this signal's window -- this launch and the tail of the previous one, peaks recorded by fb_hp_kernel --
  // reaches beyond 2^15 there: then the power of two that puts ITS peak into [2^14, 2^15).  Samples far beyond
  // full scale (float WAV files) are therefore scaled, not saturated; ordinary signals keep the one scale that
  // makes their quantisation independent of where launches cut the stream.
  int xk = 0;                                        // the scale is 2^-xk times the launch's (kept as the exponent: one
  if constexpr (sizeof(WT) == 2) {                   // scalar register over the tile loop instead of two doubles)
    const int slot = a.launch_idx % 3;
    const double pk = fmax(st->peak_slot[slot][0], st->peak_slot[slot][1]) * a.hf_xscale;
    if (pk >= 32768. && pk < __builtin_inf()) xk = __builtin_amdgcn_frexp_exp(pk) - 15;   // pk in [2^(14+xk), 2^(15+xk))
    xk = __builtin_amdgcn_readfirstlane(xk);
  }
#define xs __builtin_amdgcn_ldexp(a.hf_xscale, -xk)
#define xus __builtin_amdgcn_ldexp(a.hf_xunscale, xk)

  // recurrent state -> LDS / registers
  if (tid < kFbBands) {
    sh.cu[tid] = st->cu[tid];
#pragma unroll
    for (int i = 0; i < 10; ++i) sh.hist[tid][i] = st->e0_hist[tid][i];
  }
  double exc = tid < kFbBands ? st->excitation[tid] : 0.;
  // Slope filter (fbearmodel.c:335-339): cu_t = m cu_(t-1) + g dist_t with (m, g) = (1 - A, A) as the
  // pseudo code of BS.1387 has it (shipped), or (A, 1 - A) with SWAP_SLOPE_FILTER_COEFFICIENTS.
  // m^(t+1): decay of the state that enters a tile; m^((t & 15) + 1) for the row carries of the scan
  const double sm = a.cfg.swap_slope ? kSlopeA : 1. - kSlopeA, sg = a.cfg.swap_slope ? 1. - kSlopeA : kSlopeA;
  auto pow_m = [sm](int e) {
    double p = sm, acc = 1.;
    while (e) {
      if (e & 1) acc *= p;
      p *= p;
      e >>= 1;
    }
    return acc;
  };
  const double decay = pow_m(lane + 1), decay_row = pow_m((lane & 15) + 1);
  const double kM1 = sm, kM2 = kM1 * kM1, kM4 = kM2 * kM2, kM8 = kM4 * kM4, kM16 = kM8 * kM8;

  constexpr double kC1 = -2. * kLnDist / 2.302585092994046;         // -0.2 * 10 / ln 10 * ln DIST
  // ln DIST (24 + 230 / fc) of the wave's ten bands: in registers for the reduced-precision engine (its kernel has
  // them to spare and is 4 % slower reading them from LDS), in LDS for the FP64 engines, whose FIR loops leave
  // none (the FP64 kernel spilled 20 registers with them, 10 -- outside the tile loop -- without: 2 % on its time)
  constexpr bool kC0InRegs = sizeof(WT) == 2;
  double c0r[kC0InRegs ? 10 : 1];
  if constexpr (kC0InRegs) {
#pragma unroll
    for (int i = 0; i < 10; ++i) c0r[i] = kLnDist * (24. + 230. / bt->fc[wave_band(wv, i)]);
  } else if (tid < kFbBands) {
    sh.c0[tid] = kLnDist * (24. + 230. / bt->fc[tid]);
  }

  constexpr int kKeep = kWin - kTileSub * 32;                       // 1425 samples shared by consecutive tiles
  constexpr int kPre = (kWin - kKeep + 255) / 256;                  // new samples per thread (8)

#ifdef PEAQ_FB_PROFILE
  unsigned long long prof_t_ = __builtin_readcyclecounter();
#endif
  // per-thread and uniform constants of the tile loop, fetched once (inside, every tile waited for them again)
  const double fm_noise = tid < kFbBands ? bt->internal_noise[tid] : 0., fm_ac = tid < kFbBands ? bt->ear_tc[tid] : 0.;
  const double alias_re = ((kdouble*)(const void*)fb->h_re)[1], alias_im = ((kdouble*)(const void*)fb->h_im)[1];   // (scalar registers)
  const int tid_k = tid, lane_k = lane, wv_k = wv;
  if constexpr (sizeof(WT) == 8) {
    // FP64 engine: the staging rows of the block-sum form live in A, and some of their entries are read before anything
    // of this kernel has been written there (outputs 61 .. 63 of rows the tile shifts by two columns); what is computed
    // from them is thrown away or multiplied by zero (bs_pair) -- so it must be finite, whatever the workgroup before
    // this one left in LDS.  Once per workgroup; the first tile's barrier comes before any use.
    double2* az = reinterpret_cast<double2*>(&sh.a.re[0][0]);
    for (int i = tid; i < 2 * kFbBands * kACols / 2; i += 256) az[i] = make_double2(0., 0.);
  }
  double ltab_in[2] = {0., 0.};                      // FP64 engine: this thread's entries of the logarithm table, on their way to LDS
  auto request_ltab = [&]() {
    if constexpr (sizeof(WT) == 8) {
      const double* lt = &fb->log_tab[0][0];
      asm volatile("" : "+v"(lt));                   // (not hoisted out of the tile loop: four more registers over all phases)
      ltab_in[0] = lt[tid_k];
      ltab_in[1] = lt[256 + (tid_k & 3)];
    }
  };
  request_ltab();
  for (unsigned b0 = 0; b0 < nb_mine; b0 += kTileBlocks) {
    // The thread's indices are re-derived (as far as the compiler can tell) in every tile: otherwise it computes the
    // LDS addresses of ALL phases once in front of the loop -- some two hundred registers, most of which the FP64
    // engine's phases then push into scratch memory, whose reloads cost every phase a multiple of its time.
    int tid_v = tid_k, lane_v = lane_k, wv_s = wv_k;
    asm volatile("" : "+v"(tid_v), "+v"(lane_v), "+s"(wv_s));
    const int tid = tid_v, lane = lane_v, wv = wv_s;
    double pre[kPre];                                // the next tile's new samples on their way from HBM to the window
#pragma unroll
    for (int q = 0; q < kPre; ++q) pre[q] = 0.;
    const unsigned nvb = min((unsigned)kTileBlocks, nb_mine - b0);   // valid blocks in this tile
#ifdef PEAQ_FB_PROFILE
    if (lane == 0 && (blockIdx.x & 15) == 5) atomicAdd(&g_fb_prof[64 + wv], 1ull);
#endif
    const int nvs = 6 * nvb;                                         // valid sub-samples
    // (no barrier here: after the last barrier of the previous tile nobody reads the window or A
    // any more; wave 0 may still be in its phase 5, which only touches e1 and global memory)
    // ---- phase 0: window of the filtered signal, samples [192 b0 - 1456, 192 b0 + 59*32].  Only the
    // first tile reads all of it; later tiles keep the 1425 samples they share with their
    // predecessor (moved inside LDS) and take the 1920 new ones from registers, where they were
    // requested a whole tile ago (see below phase 2a) -------------------------------------------------
    if (b0 == 0) {
      if constexpr (sizeof(WT) == 2)                 // beyond the window proper: read by the unused time points 60..63 only
        for (int wdx = kWin + tid; wdx < 32 * kWinHBlocks; wdx += 256) sh.win.put(wdx, 0., 0.);
      if constexpr (sizeof(WT) == 8)                 // block-sum form: its tiles span 64 columns (outputs 60..63 are never
        for (int e = tid; e < 32 * 8; e += 256) {    // used) and whole blocks (the coefficients beyond a window's end are 0)
          const int r = e >> 3, c = kWinRow - 8 + (e & 7);
          if (32 * c + r >= kWin) sh.win.v[r * kWinRow + c] = 0.;
        }
      const int avail = (int)min((size_t)kWin, row_valid);
      for (int wdx = tid; wdx < kWin; wdx += 256) sh.win.put(wdx, wdx < avail ? row[wdx] : 0., xs);
    }                                                // (later tiles: the previous tile has already put its successor's samples)
    double bs_a[sizeof(WT) == 8 ? 8 : 1];              // FP64 engine: A operands of the wave's first pair, requested
    if constexpr (sizeof(WT) == 8) bs_coef_load(fb, wv, lane, bs_a);   // in front of the barrier
    if constexpr (sizeof(WT) != 8) {
      double2* az = reinterpret_cast<double2*>(&sh.a.re[0][0]);
      for (int i = tid; i < kFbBands * kACols; i += 256) az[i] = make_double2(0., 0.);
    }
    // FP64 engine: the logarithm table of the slope exponents (log_tab, peaq_wave.h; 260 doubles) into the part of
    // e1 / ex that neither the left-edge coefficients of phase 1 nor the slope exchange of phase 2 use -- every tile,
    // because phase 5 writes its excitations there; requested in front of the barrier, stored behind it (the last
    // tile's record writers may still be reading ex in front of it)
    __syncthreads();
    if constexpr (sizeof(WT) == 8) {
      double* ltab_w = &sh.e1[0][0] + kLogTabAt;
      ltab_w[tid] = ltab_in[0];
      if (tid < 4) ltab_w[256 + tid] = ltab_in[1];
    }
    FB_MARK(0);
    // ---- phase 1: the complex FIR filters (fbearmodel.c:399-435) as a GEMM on the matrix cores ---
    if constexpr (sizeof(WT) == 8) {
      // FP64 engine: bands 0 .. 23 in the block-sum form (three pairs per wave; the staging rows occupy what will
      // be A, the results wait in registers), then bands 24 .. 39 as one direct tile
      double* stg = (wv < 2 ? &sh.a.re[0][0] : &sh.a.im[0][0]) + (wv & 1) * kStWave;   // waves 0, 1 in the re half of A, 2, 3 in the im half
      double yr[3][2], yi[3][2];
#pragma unroll 1
      for (int q = 0; q < 3; ++q) {
        double pr[2], pi[2];
        // (e1, ex are idle until phase 4: 128 doubles of them per wave hold a pair's left-edge coefficients)
        bs_pair(reinterpret_cast<const double*>(sh.win.v), stg, sh.vst, &sh.e1[0][0] + 128 * wv, fb, &st->bs_hist[0][0][0],
                wv + 4 * q, b0 == 0, nvs, lane, bs_a, pr, pi);
        // (the loop is not unrolled -- unrolled, the scheduler lifts the next pair's loads over this one's work and
        // spills 200 registers -- so y goes to its registers through a uniform branch)
#define PEAQ_Y_KEEP(k)                                 \
  yr[k][0] = pr[0], yr[k][1] = pr[1], yi[k][0] = pi[0], yi[k][1] = pi[1];
        if (q == 0) {
          PEAQ_Y_KEEP(0)
        } else if (q == 1) {
          PEAQ_Y_KEEP(1)
        } else {
          PEAQ_Y_KEEP(2)
        }
#undef PEAQ_Y_KEEP
      }
      // (rows 24 .. 39 of A, which the direct tile writes, lie inside the staging rows of waves 1 and 3: the barrier
      // below stands between their last use and those stores)
      FB_MARK(13);
      __syncthreads();                                               // every wave is done with its staging rows
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          // (outputs 60..63 do not exist: what the scan left there is whatever the staging rows held)
          sh.a.re[2 * (wv + 4 * q) + sub][lane] = lane < kTileSub ? yr[q][sub] : 0.;
          sh.a.im[2 * (wv + 4 * q) + sub][lane] = lane < kTileSub ? yi[q][sub] : 0.;
        }
      FB_MARK(14);
      fir_mfma_tail<M>(sh, reinterpret_cast<const WT*>(fb->mfd_re), reinterpret_cast<const WT*>(fb->mfd_im), wv, lane);
    } else if constexpr (sizeof(WT) == 4)
      fir_mfma_f32(sh, fb->mf_re_f, fb->mf_im_f, wv, lane);
    else
      fir_mfma_h3(sh, fb, xus, wv, lane);
    FB_MARK(1);
    __syncthreads();                                                 // A is complete
    FB_MARK(2);
#ifdef PEAQ_DEV_DUMP_FIR                             // development (dev_probes.inc): the raw filter outputs instead of the patterns
    PEAQ_DEV_DUMP_FIR_SAVE
#endif
    // ---- phase 2a: every wave picks up its ten bands at its time point ---------------------------
    double re[10], im[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int b = wave_band(wv, i);
      re[i] = sh.a.re[b][lane];
      im[i] = sh.a.im[b][lane];
    }
    if (wv == 0) {
      // band 0 (= re[0] of wave 0): its tap at delay 1456 reads the NEWEST sample in the reference
      // (the doubled ring buffer makes fb_buf[offset + 1456] alias fb_buf[offset], fbearmodel.c:413-414)
      const int tt = lane < kTileSub ? lane : kTileSub - 1;
      const double delta = sh.win.get(kFbRing + 32 * tt, xus) - sh.win.get(32 * tt, xus);
      re[0] = fma(alias_re, delta, re[0]);
      im[0] = fma(-alias_im, delta, im[0]);
      sh.a.re[0][lane] = re[0];                      // band 0 is nobody's spreading target
      sh.a.im[0][lane] = im[0];
    }
    // The FP64 engine deals the upward spreading to the waves by target band (spread_up_owned): its waves meet after
    // the slopes, and its window moves on after the spreading -- until then the window's first 60 columns, dead since
    // the filters, carry the slopes from the wave that computed them to the waves that need them.
    constexpr bool kOwned = sizeof(WT) == 8;
    if constexpr (!kOwned) __syncthreads();                          // ... before phase 2b adds into A
    FB_MARK(3);
    // ---- the next tile's window: nobody reads this tile's any more.  Its last 45 columns become
    // the next tile's first 45 (a tile advances by 60 columns = 1920 samples); the new samples are
    // requested further down (request_next) and land in registers while the remaining phases run ----------
    if constexpr (!kOwned)
      if (b0 + kTileBlocks < nb_mine) sh.win.shift(tid);
    // Which of the next window's new samples a thread brings: sample kKeep + n of thread t's q-th request.  Sample-major
    // (n = t + 256 q) the 64 lanes of a store walk down a column of the window, rows 112 doubles apart: two bank pairs
    // for 64 lanes, sixteen-way conflicts on all eight stores of every tile.  FP64 engine: a lane takes column
    // (lane & 15) + 16 (q & 3) and row (lane >> 4) + 4 wave + 16 (q >> 2) of the 64 x 32 new samples instead -- sixteen
    // consecutive columns in four consecutive rows per store (two-way), sixteen cache lines per load.
    auto next_wdx = [&](int q) {
      if constexpr (sizeof(WT) == 8 && !PEAQ_FB_OLD_NEXT_MAP)
        return kKeep + 32 * ((lane & 15) + 16 * (q & 3)) + (lane >> 4) + 4 * wv + 16 * (q >> 2);
      else
        return kKeep + tid + 256 * q;
    };
    auto request_next = [&]() {
      if (b0 + kTileBlocks < nb_mine) {
        const size_t first = (size_t)(b0 + kTileBlocks) * kFbFrame;  // row index of the next window's u = 0
        const double* src = row + first;
        const int avail = (int)min((size_t)kWin, row_valid - first);
#pragma unroll
        for (int q = 0; q < kPre; ++q) {
          const int wdx = next_wdx(q);
          pre[q] = wdx < avail ? src[wdx] : 0.;
        }
      }
    };
    // (the reduced-precision engine has the registers to keep them through the spreading; in the FP64 engine's
    // kernel they would be written to scratch memory as they arrive -- the wave waits for them HERE, reads them back in
    // the next tile's phase 0 -- so it asks for them after the spreading, with three phases left to cover the latency)
    if constexpr (sizeof(WT) != 8) request_next();
    FB_MARK(4);
    // ---- phase 2b: level-dependent upward spreading (fbearmodel.c:327-349).  The slope
    // filter runs along time = along the lanes (inclusive scan with the carried-in state);
    // every source band adds its geometric tail into the bands above it.  A wave first sums
    // the tails of ITS ten sources per target band in registers and then issues ONE LDS atomic
    // per target and part (312 per tile instead of 1560; one column per lane: no contention
    // inside an instruction) ------------------------------------------------------------------------
    // (the reduced-precision engine spreads in FP32: it keeps the ten slopes as floats)
    typename std::conditional<sizeof(WT) == 2, float, double>::type cuv[10];
    // FP64 engines: the ten bands' exp(min(., . + C1 ln |A|^2)) five at a time in lockstep (log_nonneg_n)
    double dist_s5[sizeof(WT) == 2 ? 1 : 5];
    double z15 = 0., z31 = 0.;                       // the scans' row carries (dpp_rows_keep)
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int b = wave_band(wv, i);
      if constexpr (sizeof(WT) != 2) {
        if (i % 5 == 0) {
#pragma unroll
          for (int k = 0; k < 5; ++k) dist_s5[k] = re[i + k] * re[i + k] + im[i + k] * im[i + k];
          if constexpr (sizeof(WT) == 8) {
            // logarithms from the table in LDS (17 instead of 31 instructions each; five independent evaluations)
            const double* ltab = &sh.e1[0][0] + kLogTabAt;
#pragma unroll
            for (int k = 0; k < 5; ++k) dist_s5[k] = log_tab_nonneg(dist_s5[k], ltab);
          } else {
            log_nonneg_n<5>(dist_s5);
          }
#pragma unroll
          for (int k = 0; k < 5; ++k) dist_s5[k] = fmin(4. * kLnDist, sh.c0[wave_band(wv, i + k)] + kC1 * dist_s5[k]);
          exp_fast_n<5>(dist_s5);
        }
      }
      // pow(DIST, s), s = max(4, 24 + 230/fc - 0.2 L), L = 10 log10 |A|^2 (fbearmodel.c:329-333), as
      // exp(min(4 ln DIST, ln DIST (24 + 230/fc) - 2 ln DIST / ln 10 * ln |A|^2))  (ln DIST < 0)
      double cu;
      if constexpr (sizeof(WT) == 2) {
        // reduced-precision engine: the hardware's FP32 log2 / exp2 (1 ulp) on |A|^2 taken apart into exponent and
        // mantissa in FP64, so that the tiny energies of silent bands do not underflow
        const double p = re[i] * re[i] + im[i] * im[i];
        const float l2 = (float)__builtin_amdgcn_frexp_exp(p) + __builtin_amdgcn_logf((float)__builtin_amdgcn_frexp_mant(p));
        const float ex = fminf((float)(4. * kLnDist / kLn2), fmaf((float)kC1, l2, (float)(c0r[i] * (1. / kLn2))));
        const float dist_s = __builtin_amdgcn_exp2f(p == 0. ? -__builtin_inff() : ex);
        // The slope filter itself runs in FP64 like the reference's: it is the one recurrence ALONG the stream in
        // this phase, and in FP32 its rounding depended on where a launch (hence a tile) happened to start -- a
        // session and the batch path then disagreed in the eighth digit.  Everything per time point stays FP32.
        const double v = wave_prefix_geometric(sg * (double)dist_s, kM1, kM2, kM4, kM8, kM16, decay_row, lane, z15, z31);
        cu = fma(decay, sh.cu[b], v);
      } else {
        const double dist_s = dist_s5[i % 5];
        const double v = wave_prefix_geometric(sg * dist_s, kM1, kM2, kM4, kM8, kM16, decay_row, lane, z15, z31);
        cu = v + decay * sh.cu[b];
      }
      // the state after the tile's last valid time point goes on (a lane read through the scalar unit: nvs is uniform)
      const double carry = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(cu), nvs - 1),
                                            __builtin_amdgcn_readlane(__double2loint(cu), nvs - 1));
      if (lane == 0) sh.cu[b] = carry;                               // only this wave touches cu[b]
      cuv[i] = (decltype(cuv[0] + 0))cu;
    }
    FB_MARK(5);
    if constexpr (sizeof(WT) == 2) {
      switch (wv) {
        case 0: spread_up_f32<0, WT>(sh, re, im, cuv, lane); break;
        case 1: spread_up_f32<1, WT>(sh, re, im, cuv, lane); break;
        case 2: spread_up_f32<2, WT>(sh, re, im, cuv, lane); break;
        default: spread_up_f32<3, WT>(sh, re, im, cuv, lane); break;
      }
    } else if constexpr (!kOwned) {
      switch (wv) {
        case 0: spread_up<0, WT>(sh, re, im, cuv, lane); break;
        case 1: spread_up<1, WT>(sh, re, im, cuv, lane); break;
        case 2: spread_up<2, WT>(sh, re, im, cuv, lane); break;
        default: spread_up<3, WT>(sh, re, im, cuv, lane); break;
      }
    } else {
      // the slopes of this wave's ten bands to where every wave finds them: band b < 32 in row b of the window array,
      // columns 0 .. 59 = time points (dead since the filters; columns 60 .. are the next tile's), bands 32 .. 39 in
      // e1 / ex (idle until phase 4)
      double* wcol = reinterpret_cast<double*>(sh.win.v);
      double* xtra = &sh.e1[0][0];
      static_assert(sizeof(sh.e1) + sizeof(sh.ex) >= (kLogTabAt + 2 * kLogTabEntries + 2) * sizeof(double) && kLogTabAt >= 8 * 64 &&
                        kTileSub <= 60, "room for the slopes of bands 32 .. 39 and the logarithm table behind them");
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int b = wave_band(wv, i);                                // (wave-uniform)
        if (b < 32) {
          if (lane < kTileSub) wcol[b * kWinRow + lane] = (double)cuv[i];
        } else {
          xtra[(b - 32) * 64 + lane] = (double)cuv[i];
        }
      }
      __syncthreads();                               // slopes and A are in LDS; nobody reads this tile's window any more
      FB_MARK(4);
      auto cux = [&](int b, int l) { return b < 32 ? lds_rd(wcol + b * kWinRow + l) : lds_rd(xtra + (b - 32) * 64 + l); };
      double acr[10], aci[10];
      switch (wv) {
        case 0: spread_up_owned<0>(sh.a.re, sh.a.im, cux, lane, acr, aci); break;
        case 1: spread_up_owned<1>(sh.a.re, sh.a.im, cux, lane, acr, aci); break;
        case 2: spread_up_owned<2>(sh.a.re, sh.a.im, cux, lane, acr, aci); break;
        default: spread_up_owned<3>(sh.a.re, sh.a.im, cux, lane, acr, aci); break;
      }
      __syncthreads();                               // every wave has read its sources: the targets may change
      if (lane < kTileSub) {                         // (time points 60 .. 63 do not exist: their sums came from window data)
#pragma unroll
        for (int n = 0; n < 10; ++n) {
          const int j = (wv == 0 ? 4 : wv) + 4 * n;
          if (j < kFbBands) {
            sh.a.re[j][lane] = acr[n];
            sh.a.im[j][lane] = aci[n];
          }
        }
      }
      // the next tile's window moves into place (over the slopes' exchange area, dead since the barrier above)
      if (b0 + kTileBlocks < nb_mine) sh.win.shift_batched(tid);
    }
    FB_MARK(6);
    __syncthreads();
    FB_MARK(7);
    // ---- phase 3: downward spreading (fbearmodel.c:351-354): wave 0 the real, wave 1 the
    // imaginary parts; a column per lane ----------------------------------------------------------
    // (the column is read into registers first: as a loop of read - multiply-add - write the 39 steps each waited
    // for their own LDS round trip, 4.7 k cycles for 39 multiply-adds while the other two waves stood at the barrier)
    if (wv < 2) {
      double (*A)[kACols] = wv == 0 ? sh.a.re : sh.a.im;
      double col[kFbBands];
#pragma unroll
      for (int b = 0; b < kFbBands; ++b) col[b] = A[b][lane];
#pragma unroll
      for (int b = kFbBands - 1; b > 0; --b) {
        col[b - 1] = col[b - 1] + kCL * col[b];
        A[b - 1][lane] = col[b - 1];
      }
    }
    __syncthreads();
    if constexpr (sizeof(WT) == 8) {
      request_next();
      // ... and the next tile's copy of the logarithm table (see phase 0): asked for HERE and waited for in front of
      // the records' stores below.  Asked for behind phase 5 and first touched at the next tile's top, the wait there
      // was `vmcnt(0)` -- the compiler cannot count the stores of the records' loop -- i.e. a wait for the
      // acknowledgement of this tile's stores from HBM: 2.2 k cycles at the top of every tile.
      if (b0 + kTileBlocks < nb_mine) request_ltab();
    }
    FB_MARK(8);
    // ---- phase 4: rectification + backward masking at block rate (fbearmodel.c:357-382), wave-local: a
    // wave takes the ten bands it carried through phase 2, first E0 = re^2 + im^2 for all time points (lane =
    // time, written over re), then one (band, block) per lane: eleven reads of E0, no barrier in between ------
    {
      double x[10], y[10];                           // (all reads first: written as read - square - write per band,
#pragma unroll                                       // every band waited for its own LDS round trip)
      for (int i = 0; i < 10; ++i) {
        const int b = wave_band(wv, i);
        x[i] = sh.a.re[b][lane];
        y[i] = sh.a.im[b][lane];
      }
#pragma unroll
      for (int i = 0; i < 10; ++i) sh.a.re[wave_band(wv, i)][lane] = x[i] * x[i] + y[i] * y[i];
    }
    wave_lds_fence();
    double hnew[2] = {0., 0.};
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int item = lane + 64 * rep;              // 100 (band, block) / (band, history slot) items per wave
      if (item < 100) {
        const int i = item / 10, k = item - 10 * i;
        const int b = wave_band(wv, i);
        // E0 of sub-sample s (negative: previous tile): one read through a selected address
        auto e0 = [&](int s) { return *(s < 0 ? &sh.hist[b][10 + s] : &sh.a.re[b][s]); };
        const int s_new = 6 * k + 5;                 // newest sub-sample of block k
        double tap[11];
#pragma unroll
        for (int t = 0; t < 11; ++t) tap[t] = e0(s_new - t);
        double e1 = 0.;
#pragma unroll
        for (int t = 0; t < 5; ++t) e1 += (tap[t] + tap[10 - t]) * kBackMask[t];
        e1 += tap[5] * kBackMask[5];
        sh.e1[b][k] = e1;
        // history for the next tile: the 10 newest VALID sub-samples, oldest first
        hnew[rep] = e0(nvs - 10 + k);
      }
    }
    wave_lds_fence();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int item = lane + 64 * rep;
      if (item < 100) sh.hist[wave_band(wv, item / 10)][item % 10] = hnew[rep];
    }
    __syncthreads();
    FB_MARK(9);
    FB_MARK(10);
    // ---- phase 5: internal noise + forward masking (fbearmodel.c:385-394).  The recurrence along
    // the blocks is walked by one thread per band into LDS; then all threads write the records --------
    // (measured in round 5: walked by the wave that carried the band through phase 4 -- ten lanes of every wave, no
    // barrier in front -- the phase takes 1.8 k cycles instead of 1.0 k and the pass 1 % longer)
    if (tid < kFbBands) {
      const double noise = fm_noise, ac = fm_ac;
      if (nvb == kTileBlocks) {
        // a full tile: the ten values first, then the recurrence in registers, then the results -- as a loop of
        // read - two multiply-adds - write every block waited for its own LDS round trip (1.6 k cycles for 20
        // multiply-adds, with three waves at the barrier)
        double u[kTileBlocks];
#pragma unroll
        for (int bl = 0; bl < kTileBlocks; ++bl) u[bl] = sh.e1[tid][bl];
#pragma unroll
        for (int bl = 0; bl < kTileBlocks; ++bl) {
          const double unsm = u[bl] + noise;
          exc = ac * exc + (1. - ac) * unsm;
          sh.e1[tid][bl] = unsm;
          sh.ex[tid][bl] = exc;
        }
      } else {
        for (unsigned bl = 0; bl < nvb; ++bl) {
          const double unsm = sh.e1[tid][bl] + noise;
          exc = ac * exc + (1. - ac) * unsm;
          sh.e1[tid][bl] = unsm;
          sh.ex[tid][bl] = exc;
        }
      }
    }
    __syncthreads();
    FB_MARK(11);
    if constexpr (sizeof(WT) == 8)                   // (the table's entries have arrived: nothing the next tile's top
      asm volatile("" : "+v"(ltab_in[0]), "+v"(ltab_in[1]));   // needs is in flight behind the stores below)
    else if (b0 + kTileBlocks < nb_mine)
      request_ltab();
    for (int item = tid; item < kFbBands * (int)nvb; item += 256) {
      const int bl = item / kFbBands, b = item - bl * kFbBands;      // 40 consecutive doubles per block
      double* rec = a.records + ((size_t)(pair * a.blocks_per_launch + b0 + bl) * a.channels + chan) * kFbRecDoubles;
      rec[(sig ? kFbRecUnsmTest : kFbRecUnsmRef) + b] = sh.e1[b][bl];
      rec[(sig ? kFbRecExcTest : kFbRecExcRef) + b] = sh.ex[b][bl];
    }
#ifdef PEAQ_DEV_DUMP_FIR
    PEAQ_DEV_DUMP_FIR_WRITE
#endif
    FB_MARK(15);
    // the next tile's new samples have arrived: into the window (its old columns 45.. are dead since the shift; the
    // barrier of the next tile's phase 0 stands between these writes and the filters).  (Measured: in front of the
    // records' stores this phase takes 0.5 k cycles longer, and touching the lines early from phase 4 does not help.)
    if (b0 + kTileBlocks < nb_mine) {
#pragma unroll
      for (int q = 0; q < kPre; ++q) {
        const int wdx = next_wdx(q);
        if (wdx < kWin) sh.win.put(wdx, pre[q], xs);
      }
    }
    FB_MARK(12);
  }
  __syncthreads();
  if (tid < kFbBands) {
    st->cu[tid] = sh.cu[tid];
#pragma unroll
    for (int i = 0; i < 10; ++i) st->e0_hist[tid][i] = sh.hist[tid][i];
    st->excitation[tid] = exc;
  }
#undef xs
#undef xus
}

// The kernels proper.  Two workgroups per CU either way (LDS).  The reduced-precision kernel is held to 200 registers
// (amdgpu_num_vgpr counts architectural registers and the compiler doubles it on this unified register file): that was
// to leave a SIMD with two of its waves room for a wave of round 4's fb_hp_kernel, whose 28 ms the bank launch waited
// for (with 212 registers the bank kernel was as fast and the advanced pass 55 ms longer).  Round 5's walk is faster
// and larger than that room, and the pass does not notice (tests/test_kernel_budgets.py); the limit stays because the
// kernel is no faster without it.
template <typename M>
__global__ __launch_bounds__(256, 2) void fb_bank_kernel(FbFrontArgs a, unsigned n_signals) {
  fb_bank_body<M>(a, n_signals);
}
__global__ __launch_bounds__(256, 2) __attribute__((amdgpu_num_vgpr(100))) void fb_bank_kernel_h3(FbFrontArgs a,
                                                                                                unsigned n_signals) {
  fb_bank_body<MfmaH3>(a, n_signals);
}

hipError_t launch_fb_hp(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const unsigned n_signals = n_pairs * a.channels * 2;
  if (n_signals == 0 || a.blocks_per_launch == 0) return hipSuccess;
  hipLaunchKernelGGL(fb_hp_kernel, dim3((n_signals + 64 * kHpWaves - 1) / (64 * kHpWaves)), dim3(64 * kHpWaves), 0, stream, a,
                     n_signals);
  return hipGetLastError();
}

hipError_t launch_fb_bank(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const unsigned n_signals = n_pairs * a.channels * 2;
  if (n_signals == 0 || a.blocks_per_launch == 0) return hipSuccess;
  if (a.fir_fp64 == 1)
    hipLaunchKernelGGL(fb_bank_kernel<MfmaF64>, dim3(n_signals), dim3(256), 0, stream, a, n_signals);
  else if (a.fir_fp64 == 2)
    hipLaunchKernelGGL(fb_bank_kernel_h3, dim3(n_signals), dim3(256), 0, stream, a, n_signals);
  else
    hipLaunchKernelGGL(fb_bank_kernel<MfmaF32>, dim3(n_signals), dim3(256), 0, stream, a, n_signals);
  return hipGetLastError();
}

hipError_t launch_fb_frontend(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream) {
  const hipError_t e = launch_fb_hp(a, n_pairs, stream);
  return e != hipSuccess ? e : launch_fb_bank(a, n_pairs, stream);
}

}  // namespace peaq

#ifdef PEAQ_FB_PROFILE
// development builds only: reads and clears the bank kernel's phase counters (tools/fb_profile.py)
extern "C" int peaq_debug_fb_profile(unsigned long long* out68) {
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  if (hipMemcpyFromSymbol(out68, HIP_SYMBOL(peaq::g_fb_prof), sizeof(peaq::g_fb_prof)) != hipSuccess) return 1;
  static const unsigned long long zero[4 * 16 + 4] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(peaq::g_fb_prof), zero, sizeof(zero)) != hipSuccess;
}
#endif
