// peaq_backend.hip -- the stateful half of the PEAQ path: everything that
// carries state from one frame to the next.  One workgroup per pair, one
// wavefront per channel, frames processed in order; a lane owns two critical
// bands (one for the 40-band filter bank) and keeps their recurrent state in
// registers; lane i additionally owns MOV accumulator i of its channel (its
// twelve fields and the per-band constants live in LDS).
//
// Reference functions restated here (file:line under /root/reference/src):
//   time smearing                    fftearmodel.c:496-504
//   level + pattern adaptation       leveladapter.c:243-340
//   modulation patterns              modpatt.c:223-251
//   loudness gate                    earmodel.c:891-907, gstpeaq.c:841-845
//   modulation difference MOVs       movs.c:205-254
//   noise loudness MOVs              movs.c:354-371, 551-577, 679-743
//   NMR / relative disturbed frames  movs.c:1002-1022
//   detection probability MOVs       movs.c:1224-1276
//   accumulation + tentative logic   movaccum.c:317-425, gstpeaq.c:858-920, 965-1010
//   read-out, DI, ODG                movaccum.c:438-481, nn.c:187-216,304-335,372-375
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdlib>

#include "peaq_device.h"
#include "peaq_kernels.h"
#include "peaq_wave.h"
#ifdef PEAQ_DEV_PROBES                               // VARIANT builds only (csrc/Makefile): never in the product library
#define PEAQ_DEV_TU_BACKEND
#include "dev_probes.inc"
#endif
#ifndef PEAQ_DEV_SPIN_INSTEAD_OF_BACKEND
#define PEAQ_DEV_SPIN_INSTEAD_OF_BACKEND(a, block, stream)
#endif

namespace peaq {

constexpr int kLdsBands = 128;      // 2 bands x 64 lanes: every lane may store, only valid bands are read
// pattern-adaptation ratios in LDS: kPaPad zeros, 128 band slots (zero beyond the last band), spare
constexpr int kPaPad = 4, kPaStride = kPaPad + kLdsBands + 4;

// ---------------------------------------------------------------------------
// MOV accumulator owned by one lane (movaccum.c)
// ---------------------------------------------------------------------------
// The twelve fields live in LDS (field k of the accumulator at f[k * kAccLdsStride]); they are
// touched once per frame, registers are better spent on the band state.
constexpr int kAccLdsStride = 16;
struct LaneAcc {
  double* f;
  int mode, status;
  enum { NUM, DEN, NUM2, P0, P1, P2, MX, FILT, S_NUM, S_DEN, S_NUM2, S_MX };
  __device__ __forceinline__ double& at(int k) const { return f[k * kAccLdsStride]; }

  __device__ __forceinline__ void load(double* lds, const double* g, int mode_, int status_) {
    f = lds;
#pragma unroll
    for (int k = 0; k < kAccFields; ++k) at(k) = g[k];
    mode = mode_;
    status = status_;
  }
  __device__ __forceinline__ void store(double* g) const {
#pragma unroll
    for (int k = 0; k < kAccFields; ++k) g[k] = at(k);
  }
  // movaccum.c:317-362
  __device__ __forceinline__ void set_tentative(bool tentative) {
    if (tentative) {
      if (status == kNormal) {
        at(S_NUM) = at(NUM);
        at(S_DEN) = at(DEN);
        at(S_NUM2) = at(NUM2);
        at(S_MX) = at(MX);         // FILTERED_MAX: only `max`, not the filter state (:343-346)
        status = kTentative;
      }
    } else {
      status = kNormal;
    }
  }
  // movaccum.c:368-425
  __device__ __forceinline__ void add(double val, double w) {
    if (status == kInit) return;
    switch (mode) {
      case kRms:
        w *= w;
        at(NUM) += w * val * val;
        at(DEN) += w;
        break;
      case kRmsAsym:
        at(NUM) += val * val;
        at(NUM2) += w * w;
        at(DEN) += 1.;
        break;
      case kAvg:
      case kAvgLog:
      case kAdb:
        at(NUM) += w * val;
        at(DEN) += w;
        break;
      case kAvgWindow: {
        const double sq = sqrt(val);
        const double p0 = at(P0), p1 = at(P1), p2 = at(P2);
        if (!isnan(p0)) {
          double ws = ((sq + p0) + p1) + p2;
          ws /= 4.;
          ws *= ws;
          ws *= ws;
          at(NUM) += ws;
          at(DEN) += 1.;
        }
        at(P0) = p1;
        at(P1) = p2;
        at(P2) = sq;
        break;
      }
      case kFilteredMax: {
        const double filt = 0.9 * at(FILT) + 0.1 * val;
        at(FILT) = filt;
        if (filt > at(MX)) at(MX) = filt;
        break;
      }
    }
  }
};

__device__ __forceinline__ int acc_mode(bool advanced, int i) {
  // gstpeaq.c:528-557
  if (advanced) return i == 0 ? kRms : i == 1 ? kRmsAsym : kAvg;
  switch (i) {
    case 2: return kAvgLog;
    case 3: return kAvgWindow;
    case 4: return kAdb;
    case 8: return kRms;
    case 9: return kFilteredMax;
    default: return kAvg;
  }
}

// value of one channel's accumulator (movaccum.c:448-477, per-channel term)
__device__ __forceinline__ double acc_channel_value(int mode, bool tentative, const double* f) {
  const double num = tentative ? f[8] : f[0], den = tentative ? f[9] : f[1];
  const double num2 = tentative ? f[10] : f[2], mx = tentative ? f[11] : f[6];
  switch (mode) {
    case kAvg: return num / den;
    case kAvgLog: return 10. * log10(num / den);
    case kAvgWindow:
    case kRms: return sqrt(num / den);
    case kRmsAsym: return sqrt(num / den) + 0.5 * sqrt(num2 / den);
    case kFilteredMax: return mx;
    case kAdb: return den > 0 ? (num == 0. ? -0.5 : log10(num / den)) : 0.;
  }
  return 0.;
}

// ---------------------------------------------------------------------------
// shared per-frame building blocks; SLOTS bands per lane, band b = SLOTS*lane + s
// ---------------------------------------------------------------------------
template <int NB, int SLOTS>
struct BandLane {
  int lane;
  __device__ __forceinline__ int band(int s) const { return SLOTS * lane + s; }
  __device__ __forceinline__ bool valid(int s) const { return band(s) < NB; }
};

// Per-band constants as the building blocks see them: straight from the tables in global memory
// (filter-bank back end), or from a copy in LDS (FFT back end: nine tables x two bands would
// otherwise sit in 36 registers for the whole frame loop -- or be spilled).
struct GlobalTabs {
  const BandTables* __restrict__ p;
  __device__ __forceinline__ double adapt_tc(int b) const { return p->adapt_tc[b]; }
  __device__ __forceinline__ double ear_tc(int b) const { return p->ear_tc[b]; }
  __device__ __forceinline__ double threshold(int b) const { return p->threshold[b]; }
  __device__ __forceinline__ double loud_factor(int b) const { return p->loud_factor[b]; }
  __device__ __forceinline__ double exc_threshold(int b) const { return p->exc_threshold[b]; }
  __device__ __forceinline__ double internal_noise(int b) const { return p->internal_noise[b]; }
  __device__ __forceinline__ double noise_pow03(int b) const { return p->noise_pow03[b]; }
  __device__ __forceinline__ double mask_diff(int b) const { return p->mask_diff[b]; }
  __device__ __forceinline__ double ln_internal_noise(int b) const { return p->ln_internal_noise[b]; }
  __device__ __forceinline__ double inv_window_count(int b) const { return p->inv_window_count[b]; }
  __device__ __forceinline__ double deriv_factor() const { return p->deriv_factor; }
  // transcendentals of the MOV layer: the logarithm from the table in LDS (see LdsTabs)
  const double* ltab;
  const double* etab;               // ... and the exponential from its own (exp_tab, peaq_wave.h)
#if defined(PEAQ_LEDGER_FP32_BACKEND) || defined(PEAQ_NO_LOGTAB_BE)
  __device__ __forceinline__ double log(double x) const { return be_log(x); }
  __device__ __forceinline__ double exp(double x) const { return be_exp(x); }
  __device__ __forceinline__ double pow(double x, double y) const { return be_pow(x, y); }
#else
  __device__ __forceinline__ double log(double x) const { return log_tab(x, ltab); }
  __device__ __forceinline__ double exp(double x) const { return exp_tab(x, etab); }
  __device__ __forceinline__ double pow(double x, double y) const { return exp_tab(y * log_tab(x, ltab), etab); }
#endif
};
enum { T_ADAPT, T_EAR, T_THR, T_LOUDF, T_EXCTHR, T_INOISE, T_NPOW03, T_MASK, T_ISN, T_ISN03, T_LNINOISE, T_RCNT, T_COUNT };
struct LdsTabs {
  const double* t;                  // [T_COUNT][kBandStride] in LDS
  int off;                          // 0, but opaque to the compiler (re-read per frame, not hoisted)
  double deriv;
  __device__ __forceinline__ double at(int tab, int b) const { return t[off + tab * kBandStride + b]; }
  __device__ __forceinline__ double adapt_tc(int b) const { return at(T_ADAPT, b); }
  __device__ __forceinline__ double ear_tc(int b) const { return at(T_EAR, b); }
  __device__ __forceinline__ double threshold(int b) const { return at(T_THR, b); }
  __device__ __forceinline__ double loud_factor(int b) const { return at(T_LOUDF, b); }
  __device__ __forceinline__ double exc_threshold(int b) const { return at(T_EXCTHR, b); }
  __device__ __forceinline__ double internal_noise(int b) const { return at(T_INOISE, b); }
  __device__ __forceinline__ double noise_pow03(int b) const { return at(T_NPOW03, b); }
  __device__ __forceinline__ double mask_diff(int b) const { return at(T_MASK, b); }
  __device__ __forceinline__ double ln_internal_noise(int b) const { return at(T_LNINOISE, b); }
  __device__ __forceinline__ double inv_window_count(int b) const { return at(T_RCNT, b); }
  __device__ __forceinline__ double deriv_factor() const { return deriv; }
  // transcendentals of the MOV layer: the logarithm from the 129-entry table in LDS (log_tab, peaq_wave.h) --
  // ten logarithms per band and frame are a tenth of this kernel's vector instructions otherwise
  const double* ltab;               // [kLogTabEntries][2] in LDS
  // (the exponential stays the polynomial here: this kernel runs beside the front end, whose LDS array is the busier
  // unit -- with exp_tab the basic step measured 0 ... - 1 %, the filter-bank back end above + 0.8 % on its pass)
#if defined(PEAQ_LEDGER_FP32_BACKEND) || defined(PEAQ_NO_LOGTAB_BE)
  __device__ __forceinline__ double log(double x) const { return be_log(x); }
  __device__ __forceinline__ double exp(double x) const { return be_exp(x); }
  __device__ __forceinline__ double pow(double x, double y) const { return be_pow(x, y); }
#else
  __device__ __forceinline__ double log(double x) const { return log_tab(x, ltab); }
  __device__ __forceinline__ double exp(double x) const { return be_exp(x); }
  __device__ __forceinline__ double pow(double x, double y) const { return be_exp(y * log_tab(x, ltab)); }
#endif
};

// leveladapter.c:243-340.  e_ref/e_test: excitation patterns of this frame.
// st: [6][SLOTS] state (filt_ref, filt_test, num, den, pattcorr_ref, pattcorr_test)
template <int NB, int SLOTS, class TAB>
__device__ __forceinline__ void level_adapt(const BandLane<NB, SLOTS>& bl, const TAB& bt,
                                            const double (&e_ref)[SLOTS], const double (&e_test)[SLOTS],
                                            double (&st)[6][SLOTS], double* pa_lds /* [2][kPaStride], zero padded */,
                                            double (&ad_ref)[SLOTS], double (&ad_test)[SLOTS]) {
  double num = 0., den = 0.;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (bl.valid(s)) {
      const double a = bt.adapt_tc(bl.band(s));
      st[0][s] = a * st[0][s] + (1 - a) * e_ref[s];          // (42)/(43) in BS.1387
      st[1][s] = a * st[1][s] + (1 - a) * e_test[s];
      num += sqrt_pos(st[0][s] * st[1][s]);                  // (45)
      den += st[1][s];
    }
  }
  wave_sum2(num, den, num, den);
  // lev = (num / den)^2 and its reciprocal, once per wave; the quotients of the per-band loop below
  // are products with reciprocals that exist anyway (1 ulp from the reference's divisions)
  const double n2 = num * num, d2 = den * den;
  const double lev = div_fast(n2, d2), inv_lev = div_fast(d2, n2);
  double lc_ref[SLOTS], lc_test[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (lev > 1) {                                           // (46)/(47)
      lc_ref[s] = e_ref[s] * inv_lev;
      lc_test[s] = e_test[s];
    } else {
      lc_ref[s] = e_ref[s];
      lc_test[s] = e_test[s] * lev;
    }
    double pr = 0., pt = 0.;
    if (bl.valid(s)) {
      const double a = bt.adapt_tc(bl.band(s));
      st[2][s] = a * st[2][s] + lc_test[s] * lc_ref[s];      // (48): no (1-a) gain, leveladapter.c:293-298
      st[3][s] = a * st[3][s] + lc_ref[s] * lc_ref[s];
      if (st[2][s] >= st[3][s]) {                            // (49)
        pr = 1.;
        pt = div_fast(st[3][s], st[2][s]);
      } else {
        pr = div_fast(st[2][s], st[3][s]);
        pt = 1.;
      }
    }
    pa_lds[kPaPad + bl.band(s)] = pr;                        // 0 for the slots beyond the last band
    pa_lds[kPaStride + kPaPad + bl.band(s)] = pt;
  }
  wave_lds_fence();
  constexpr int M1 = NB / 36, M2 = NB / 25;                  // leveladapter.c:315-316
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    ad_ref[s] = 0.;
    ad_test[s] = 0.;
    if (bl.valid(s)) {
      const int k = bl.band(s);
      // (50)/(51): window [k - m1, k + m2] summed in ascending order like the reference.  The
      // array is zero outside [0, NB), so the full window [k - M1, k + M2] gives the same sums
      // bit for bit (x + 0 = x) with compile-time offsets from one address.
      double rr = 0., rt = 0.;
#pragma unroll
      for (int j = -M1; j <= M2; ++j) {
        rr += pa_lds[kPaPad + k + j];
        rt += pa_lds[kPaStride + kPaPad + k + j];
      }
      const double rcnt = bt.inv_window_count(k);            // 1 / (m1 + m2 + 1)
      rr *= rcnt;
      rt *= rcnt;
      const double a = bt.adapt_tc(k);
      st[4][s] = a * st[4][s] + (1 - a) * rr;
      st[5][s] = a * st[5][s] + (1 - a) * rt;
      ad_ref[s] = lc_ref[s] * st[4][s];                      // (52)/(53)
      ad_test[s] = lc_test[s] * st[5][s];
    }
  }
  wave_lds_fence();
}

// modpatt.c:223-251; st: prev_loud, filt_loud, filt_dloud
template <int NB, int SLOTS, class TAB>
__device__ __forceinline__ void modulation(const BandLane<NB, SLOTS>& bl, const TAB& bt,
                                           const double (&loud)[SLOTS], double (&st)[3][SLOTS],
                                           double (&mod)[SLOTS]) {
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    mod[s] = 0.;
    if (bl.valid(s)) {
      const double a = bt.adapt_tc(bl.band(s));
      const double dl = bt.deriv_factor() * fabs(loud[s] - st[0][s]);
      st[2][s] = a * st[2][s] + (1 - a) * dl;
      st[1][s] = a * st[1][s] + (1. - a) * loud[s];
      mod[s] = div_fast(st[2][s], 1. + st[1][s] * (1. / 0.3));
      st[0][s] = loud[s];
    }
  }
}

// earmodel.c:891-907
template <int NB, int SLOTS, class TAB>
__device__ __forceinline__ double total_loudness_part(const BandLane<NB, SLOTS>& bl, const TAB& bt,
                                                      const double (&exc)[SLOTS]) {
  double t = 0.;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (bl.valid(s)) {
      const int b = bl.band(s);
      const double thr = bt.threshold(b);
      const double l = bt.loud_factor(b) * (bt.pow(1. - thr + div_fast(thr * exc[s], bt.exc_threshold(b)), 0.23) - 1.);
      t += fmax(l, 0.);
    }
  }
  return t;
}

// movs.c:709-743.  `lead`: the factor (ethres / stest)^0.23 per slot -- computed here (KEEP: and handed back) or taken
// from a call with the same thres_fac, s0 and mod_test (USE): RmsNoiseLoudAsym's missing-components term and AvgLinDist
// (movs.c:551-577, 679-706) share it, a logarithm and an exponential per band and block.
enum LeadMode { LEAD_OWN, LEAD_KEEP, LEAD_USE };
// (the lane's part of the sum over the bands: callers with several sums in a frame take them through ONE reduction,
// wave_sum2 / wave_sum4, and finish with noise_loudness_total)
template <int NB, int SLOTS, class TAB, LeadMode LM = LEAD_OWN>
__device__ __forceinline__ double noise_loudness_part(const BandLane<NB, SLOTS>& bl, const TAB& bt,
                                                      double alpha, double thres_fac, double s0,
                                                      const double (&mod_ref)[SLOTS], const double (&mod_test)[SLOTS],
                                                      const double (&e_ref)[SLOTS], const double (&e_test)[SLOTS],
                                                      double* lead = nullptr) {
  double nl = 0.;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (bl.valid(s)) {
      const double sref = thres_fac * mod_ref[s] + s0;
      const double stest = thres_fac * mod_test[s] + s0;
      const double ethres = bt.internal_noise(bl.band(s));
      const double beta = bt.exp(div_fast(-alpha * (e_test[s] - e_ref[s]), e_ref[s]));
      // (ethres / stest)^0.23 from the logarithms: ln ethres is a table entry
      double ld;
      if (LM == LEAD_USE)
        ld = lead[s];
      else
        ld = bt.exp(0.23 * (bt.ln_internal_noise(bl.band(s)) - bt.log(stest)));
      if (LM == LEAD_KEEP) lead[s] = ld;
      nl += ld *
            (bt.pow(1. + div_fast(fmax(stest * e_test[s] - sref * e_ref[s], 0.), ethres + sref * e_ref[s] * beta), 0.23) -
             1.);
    }
  }
  return nl;
}
template <int NB>
__device__ __forceinline__ double noise_loudness_total(double sum, double nl_min) {
  const double nl = sum * (24. / NB);
  return nl < nl_min ? 0. : nl;
}

// movs.c:224-251: returns d1 (un-normalised), d2, weight
template <int NB, int SLOTS, class TAB>
__device__ __forceinline__ void mod_difference(const BandLane<NB, SLOTS>& bl, const TAB& bt,
                                               double lev_wt, const double (&mr)[SLOTS], const double (&mt)[SLOTS],
                                               const double (&loud_ref)[SLOTS], double& d1, double& d2, double& wt) {
  double a1 = 0., a2 = 0., aw = 0.;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    if (bl.valid(s)) {
      const double diff = fabs(mr[s] - mt[s]);
      a1 += div_fast(diff, 1. + mr[s]);
      a2 += div_fast((mt[s] >= mr[s] ? 1. : .1) * diff, 0.01 + mr[s]);
      aw += div_fast(loud_ref[s], loud_ref[s] + lev_wt * bt.noise_pow03(bl.band(s)));
    }
  }
  double unused;
  wave_sum4(a1, a2, aw, 0., d1, d2, wt, unused);
}

// ---------------------------------------------------------------------------
// FFT-model back end.  ADV = false: basic version (109 bands, 11 MOVs).
// ADV = true: the FFT part of the advanced version (55 bands; SegmentalNMR, EHS).
// ---------------------------------------------------------------------------
enum { MB_BW_REF, MB_BW_TEST, MB_NMR, MB_WINMOD, MB_ADB, MB_EHS, MB_AVGMOD1, MB_AVGMOD2, MB_NOISELOUD, MB_MFPD,
       MB_RELDIST };                                         // gstpeaq.c:95-108
enum { MA_RMSMOD, MA_NLASYM, MA_SEGNMR, MA_EHS, MA_LINDIST };   // gstpeaq.c:86-93

struct BackendShared {
  double pa[2][2][kPaStride];       // [wave][ref/test][pad + band] pattern-adaptation ratios
  double pc[2][kLdsBands];          // exponents xb of the detection probabilities 1 - 0.5^xb, per channel
  double qc[2][kLdsBands];
  double acc[2][kAccFields][kAccLdsStride];   // [channel][field][accumulator]
  double energy[2];                 // totalsnr: signal and noise energy so far (lane 0 of channel 0)
  int gate[2];
};

// 3 waves per SIMD requested: the back end runs BESIDE the front end of the next chunk (which
// holds 3 x 168 VGPRs per SIMD); with the default budget (256) it could never be co-scheduled
// DBG = true (basic version only, peaq_debug_backend): the per-frame patterns are also written to
// a.debug for the stage-level parity tests; the arithmetic is the same instantiation otherwise.
template <int NB, bool ADV, bool DBG = false>
__global__ __launch_bounds__(128, 3) void backend_kernel(BackendArgs a) {
  __shared__ BackendShared sh;
  __shared__ double sh_tab[T_COUNT * kBandStride];
  __shared__ __attribute__((aligned(16))) double sh_ltab[2 * kLogTabEntries + 2];
  constexpr int SLOTS = 2;
  const int lane = threadIdx.x & 63;
  const int chan = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keep it scalar
  const int channels = a.channels;                  // == blockDim.x / 64
  const unsigned pair = blockIdx.x;
  const BandLane<NB, SLOTS> bl{lane};
  {
    const BandTables* __restrict__ g = a.bands;
    const double* const src[T_COUNT] = {g->adapt_tc, g->ear_tc, g->threshold, g->loud_factor, g->exc_threshold,
                                        g->internal_noise, g->noise_pow03, g->mask_diff, g->inv_spread_norm,
                                        g->inv_spread_norm_pow03, g->ln_internal_noise, g->inv_window_count};
#pragma unroll
    for (int t = 0; t < T_COUNT; ++t)
      for (int i = threadIdx.x; i < kBandStride; i += blockDim.x) sh_tab[t * kBandStride + i] = src[t][i];
    for (int i = threadIdx.x; i < 2 * kLogTabEntries; i += blockDim.x) sh_ltab[i] = a.common->log_tab[i >> 1][i & 1];
  }
  LdsTabs bt{sh_tab, 0, a.bands->deriv_factor, sh_ltab};
  PairState* __restrict__ ps = a.state + (a.pair_slot ? a.pair_slot[pair] : pair);
  ChannelState* __restrict__ cs = &ps->ch[chan];

  unsigned f_begin, f_end;
  if (a.pair_frame0) {                               // broker launch: this pair's own window
    f_begin = a.pair_frame0[pair];
    f_end = f_begin + a.pair_nframes[pair];
  } else {
    const unsigned n_frames = a.n_frames ? a.n_frames[pair] : a.n_frames_uniform;
    f_begin = a.frame0;
    f_end = a.frame0 + a.frames_per_launch;
    if (f_end > n_frames) f_end = n_frames;
  }
  if (f_begin >= f_end) return;
  const bool clk_wave = a.clk && blockIdx.x == 0 && chan == 0;   // wave-uniform
  unsigned long long clk_s0 = 0, clk_w0 = 0;
  if (clk_wave) {
    clk_w0 = wall_clock64();
    clk_s0 = __builtin_readcyclecounter();
  }
  if (lane < kPaPad) sh.pa[chan][0][lane] = sh.pa[chan][1][lane] = 0.;
  __syncthreads();                                   // the table copy is complete

  // ---- recurrent state -> registers -----------------------------------------------
  double sm[2][SLOTS];                               // smeared excitation filters (ref, test)
  double la[6][SLOTS];
  double mdr[3][SLOTS], mdt[3][SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int b = bl.band(s);
    sm[0][s] = cs->vec[kSmearRef][b < kBandStride ? b : 0];
    sm[1][s] = cs->vec[kSmearTest][b < kBandStride ? b : 0];
#pragma unroll
    for (int v = 0; v < 6; ++v) la[v][s] = cs->vec[kLaFiltRef + v][b < kBandStride ? b : 0];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      mdr[v][s] = cs->vec[kModPrevRef + v][b < kBandStride ? b : 0];
      mdt[v][s] = cs->vec[kModPrevTest + v][b < kBandStride ? b : 0];
    }
  }
  LaneAcc acc;
  {
    const int i = lane < kMaxAcc ? lane : 0;
    acc.load(&sh.acc[chan][0][lane < kAccLdsStride ? lane : kAccLdsStride - 1], cs->acc[i], acc_mode(ADV, i),
             ps->status[i]);   // lanes beyond the 11 accumulators work on dummy slots
  }
  unsigned loud_reached = ps->loudness_reached;
  if (chan == 0 && lane == 0) {
    sh.energy[0] = ps->sig_energy;
    sh.energy[1] = ps->noise_energy;
  }

  for (unsigned frame = f_begin; frame < f_end; ++frame) {
    asm volatile("" : "+v"(bt.off));                 // the tables are re-read from LDS where they are used
    const double* __restrict__ rec0 =
        a.records + ((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels) * kRecDoubles;
    const double* __restrict__ rec = rec0 + (size_t)chan * kRecDoubles;

    // ---- frame flags over all channels (gstpeaq.c:858-862, movs.c:1374-1381) -----
    int fl_ref = (int)rec0[kRecFlagsRef], fl_test = (int)rec0[kRecFlagsTest];
    if (channels == 2) {
      fl_ref |= (int)rec0[kRecDoubles + kRecFlagsRef];
      fl_test |= (int)rec0[kRecDoubles + kRecFlagsTest];
    }
    const bool above = fl_ref & 1;
    const bool ehs_valid = ((fl_ref | fl_test) & 2) != 0;
    if (!ADV || lane == MA_SEGNMR || lane == MA_EHS) acc.set_tentative(!above);

    // ---- this frame's patterns -------------------------------------------------------
    double ur[SLOTS], ut[SLOTS], lr[SLOTS], lt[SLOTS], nz[SLOTS];
    {
      const int b0 = bl.band(0);
      const int bb = b0 < kBandStride ? b0 : 0;
      const double2 v0 = *reinterpret_cast<const double2*>(rec + kRecRootRef + bb);
      const double2 v1 = *reinterpret_cast<const double2*>(rec + kRecRootTest + bb);
      const double2 v4 = *reinterpret_cast<const double2*>(rec + kRecNoise + bb);
      // unsmeared excitation (fftearmodel.c:593-597) and its 0.3rd power (modpatt.c:235) from the roots
      const double n0 = bt.at(T_ISN, bb), n1 = bt.at(T_ISN, bb + 1);
      const double m0 = bt.at(T_ISN03, bb), m1 = bt.at(T_ISN03, bb + 1);
      excitation_from_root(v0.x, n0, m0, ur[0], lr[0]);
      excitation_from_root(v0.y, n1, m1, ur[1], lr[1]);
      excitation_from_root(v1.x, n0, m0, ut[0], lt[0]);
      excitation_from_root(v1.y, n1, m1, ut[1], lt[1]);
      nz[0] = v4.x; nz[1] = v4.y;
    }
    double nl_part = 0.;                               // basic version: the lane's part of the noise loudness, summed with the NMR's
    bool nl_open = false;
    // time smearing, fftearmodel.c:496-504
    double er[SLOTS], et[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int b = bl.band(s) < kBandStride ? bl.band(s) : 0;
      const double ac = bt.ear_tc(b);
      sm[0][s] = ac * sm[0][s] + (1. - ac) * ur[s];
      er[s] = sm[0][s] > ur[s] ? sm[0][s] : ur[s];
      if (!ADV) {
        sm[1][s] = ac * sm[1][s] + (1. - ac) * ut[s];
        et[s] = sm[1][s] > ut[s] ? sm[1][s] : ut[s];
      } else {
        et[s] = 0.;
      }
    }

    // lane i owns accumulator i: every MOV value of the frame is routed to its owner as soon as
    // it exists (two selects) instead of being kept in a per-lane table
    double my_v = 0., my_w = 1.;
    bool my_hit = false;
    auto route = [&](int idx, double v, double w) {
      if (lane == idx) {
        my_v = v;
        my_w = w;
        my_hit = true;
      }
    };

    if (!ADV) {
      // ---- pattern processing (gstpeaq.c:834-845) ------------------------------------
      double ad_ref[SLOTS], ad_test[SLOTS], mr[SLOTS], mt[SLOTS];
      level_adapt<NB, SLOTS>(bl, bt, er, et, la, &sh.pa[chan][0][0], ad_ref, ad_test);
      modulation<NB, SLOTS>(bl, bt, lr, mdr, mr);
      modulation<NB, SLOTS>(bl, bt, lt, mdt, mt);
      if (DBG) {
        double* __restrict__ d =
            a.debug + ((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels + chan) * kDbgDoubles;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (bl.valid(s)) {
            const int b = bl.band(s);
            d[kDbgExcRef + b] = er[s];
            d[kDbgExcTest + b] = et[s];
            d[kDbgAdaptRef + b] = ad_ref[s];
            d[kDbgAdaptTest + b] = ad_test[s];
            d[kDbgModRef + b] = mr[s];
            d[kDbgModTest + b] = mt[s];
            d[kDbgAvgLoudRef + b] = mdr[1][s];
            d[kDbgAvgLoudTest + b] = mdt[1][s];
          }
        }
      }
      if (loud_reached == UINT_MAX) {                // wave-uniform
        double n_ref, n_test;
        wave_sum2(total_loudness_part<NB, SLOTS>(bl, bt, er), total_loudness_part<NB, SLOTS>(bl, bt, et), n_ref, n_test);
        n_ref *= 24. / NB;
        n_test *= 24. / NB;
        if (lane == 0) sh.gate[chan] = (n_ref > 0.1 && n_test > 0.1);
        if (DBG && lane == 0) {
          double* __restrict__ d =
              a.debug + ((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels + chan) * kDbgDoubles;
          d[kDbgLoudnessRef] = n_ref;
          d[kDbgLoudnessTest] = n_test;
        }
      }
      // ---- detection probability, per channel part (movs.c:1239-1262) -----------------
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        double pc = 0., qc = 0.;
        if (bl.valid(s)) {
          const double er_db = (10. * kInvLn10) * bt.log(er[s]);      // 10 log10: excitations are > 0
          const double et_db = (10. * kInvLn10) * bt.log(et[s]);
          const double l = 0.3 * fmax(er_db, et_db) + 0.7 * et_db;
          const double l2 = l * l;
          // (6.39468 / l)^1.71332 = exp(1.71332 (ln 6.39468 - ln l)); one reciprocal of s for both quotients
          const double sd = l > 0. ? 5.95072 * bt.exp(1.71332 * (1.8554663946857675 - bt.log(l))) + 9.01033e-11 * l2 * l2 +
                                         5.05622e-6 * l2 * l - 0.00102438 * l * l + 0.0550197 * l - 0.198719
                                   : 1e30;
          const double inv_sd = div_fast(1., sd);
          const double e = er_db - et_db;
          const double x = e * inv_sd, x2 = x * x;
          const double xb = er_db > et_db ? x2 * x2 : x2 * x2 * x2;   // (e/s)^b, b = 4 or 6
          // The channel's detection probability is pc = 1 - 0.5^xb (movs.c:1253); what the frame needs of it is
          // prod_b (1 - max_c pc) = 0.5^(sum_b max_c xb) (pc grows with xb, so the maxima agree): the EXPONENTS are
          // exchanged and summed, and the one exponential of the frame is taken after the reduction -- an exponential
          // per band, channel and frame less, and the product's own reduction rides in the free slot of the sums'.
          pc = xb;
          qc = fabs(a.cfg.floor_steps ? floor(e) : trunc(e)) * inv_sd;        // movs.c:1256-1260
        }
        sh.pc[chan][bl.band(s)] = pc;
        sh.qc[chan][bl.band(s)] = qc;
      }
      __syncthreads();
      if (loud_reached == UINT_MAX) {
        const int g = sh.gate[0] | (channels == 2 ? sh.gate[1] : 0);
        if (g) loud_reached = frame;
      }
      double* __restrict__ dmov =                    // debug instantiation: this (frame, channel)'s MOV values
          DBG ? a.debug + ((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels + chan) * kDbgDoubles + kDbgMov
              : nullptr;
      // ---- modulation difference (gstpeaq.c:871-877) --------------------------------
      if (DBG || frame >= 24) {
        double d1, d2, wt;
        mod_difference<NB, SLOTS>(bl, bt, 100., mr, mt, mdr[1], d1, d2, wt);
        d1 *= 100. / NB;
        d2 *= 100. / NB;
        if (frame >= 24) {
          route(MB_AVGMOD1, d1, wt);
          route(MB_AVGMOD2, d2, wt);
          route(MB_WINMOD, d1, 1.);
        }
        if (DBG && lane == 0) {
          dmov[0] = d1;
          dmov[1] = d2;
          dmov[2] = wt;
        }
      }
      // ---- noise loudness (gstpeaq.c:880-886; unsigned compare with UINT_MAX sentinel)
      // (its sum over the bands goes through the reduction of the noise-to-mask ratio below)
      nl_open = DBG || (frame >= 24 && frame - 3 >= loud_reached);
      if (nl_open) nl_part = noise_loudness_part<NB, SLOTS>(bl, bt, 1.5, 0.15, 0.5, mr, mt, ad_ref, ad_test);
      // ---- bandwidth (movs.c:797-807) ------------------------------------------------------
      {
        const double bw_ref = rec[kRecBwRef];
        if (bw_ref > 346.) {
          route(MB_BW_REF, bw_ref, 1.);
          route(MB_BW_TEST, rec[kRecBwTest], 1.);
        }
      }
    }
    // ---- noise-to-mask ratio (movs.c:1002-1022), detection probability's binaural part (movs.c:1263-1275): the
    // lanes' parts first, then ONE reduction for the three sums of this place (with the noise loudness's from above) ----
    {
      double nsum = 0., nmax = 0.;
#pragma unroll
      for (int s = 0; s < SLOTS; ++s) {
        if (bl.valid(s)) {
          const double r = div_fast(nz[s] * bt.mask_diff(bl.band(s)), er[s]);   // noise / (excitation / mask)
          nsum += r;
          if (r > nmax) nmax = r;
        }
      }
      double xsum = 0., qsum = 0.;                    // sum of the bands' exponents (pc above), of the steps
      if (!ADV && chan == 0) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (bl.valid(s)) {
            const int b = bl.band(s);
            double x = fmax(sh.pc[0][b], 0.), q = sh.qc[0][b];      // (fmax: a NaN exponent counts as 0, like `pc > p`)
            if (channels == 2) {
              if (sh.pc[1][b] > x) x = sh.pc[1][b];
              if (sh.qc[1][b] > q) q = sh.qc[1][b];
            }
            xsum += x;
            qsum += q;
          }
        }
      }
      double nl_sum;
      if (!ADV)
        wave_sum4(nsum, nl_part, qsum, xsum, nsum, nl_sum, qsum, xsum);
      else
        nsum = wave_sum(nsum);
      nsum /= NB;
      // RelDistFrames asks whether ANY band's ratio is above 1.5 dB: a vote, not a maximum (the debug build reports the value)
      const bool disturbed = __any(nmax > 1.41253754462275);
      if (DBG) nmax = wave_max(nmax);
      if (!ADV && nl_open) {
        const double nl = noise_loudness_total<NB>(nl_sum, 0.);
        if (frame >= 24 && frame - 3 >= loud_reached) route(MB_NOISELOUD, nl, 1.);
        if (DBG && lane == 0)
          a.debug[((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels + chan) * kDbgDoubles + kDbgMov + 3] = nl;
      }
      if (DBG && lane == 0) {
        double* __restrict__ d =
            a.debug + ((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels + chan) * kDbgDoubles + kDbgMov;
        d[4] = nsum;
        d[5] = nmax;
      }
      if (!ADV) {
        route(MB_NMR, nsum, 1.);                                    // MODE_AVG_LOG
        route(MB_RELDIST, disturbed ? 1. : 0., 1.);
      } else {
        const double seg = (10. * kInvLn10) * log_pos(nsum);        // 10 log10, MODE_AVG; nsum > 0 (floored bands)
        route(MA_SEGNMR, seg, 1.);
        if (DBG && lane == 0)
          a.debug[((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels + chan) * kDbgDoubles + kDbgMov + 3] = seg;
      }
      if (!ADV && chan == 0) {
        const double p_bin = 1. - bt.exp(-kLn2 * xsum);             // 1 - prod_b 0.5^xb (movs.c:1263-1270)
        if (DBG && lane == 0) {
          double* __restrict__ d =
              a.debug + ((size_t)(pair * a.frames_per_launch + (frame - f_begin)) * channels) * kDbgDoubles + kDbgMov;
          d[6] = p_bin;
          d[7] = qsum;
        }
        if (p_bin > 0.5) route(MB_ADB, qsum, 1.);
        route(MB_MFPD, p_bin, 1.);
      }
    }
    // ---- error harmonic structure (movs.c:1374-1381,1442) ------------------------------
    if (ehs_valid) {
      route(ADV ? MA_EHS : MB_EHS, 1000. * rec[kRecEhs], 1.);
    }
    // ---- totalsnr (gstpeaq.c:913-918) --------------------------------------------------------
    if (chan == 0 && lane == 0) {
      sh.energy[0] += rec0[kRecSigE] + (channels == 2 ? rec0[kRecDoubles + kRecSigE] : 0.);
      sh.energy[1] += rec0[kRecNoiseE] + (channels == 2 ? rec0[kRecDoubles + kRecNoiseE] : 0.);
    }
    // ---- accumulate: lane i owns accumulator i --------------------------------------------------
    if (my_hit) acc.add(my_v, my_w);
    if (!ADV) __syncthreads();                       // sh.pc/qc/gate are rewritten next frame
  }

  // ---- registers -> recurrent state -------------------------------------------------------
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int b = bl.band(s);
    if (b < kBandStride) {
      cs->vec[kSmearRef][b] = sm[0][s];
      cs->vec[kSmearTest][b] = sm[1][s];
      if (!ADV) {                                    // advanced: these belong to the filter-bank back end
#pragma unroll
        for (int v = 0; v < 6; ++v) cs->vec[kLaFiltRef + v][b] = la[v][s];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
          cs->vec[kModPrevRef + v][b] = mdr[v][s];
          cs->vec[kModPrevTest + v][b] = mdt[v][s];
        }
      }
    }
  }
  if (lane < kMaxAcc) {
    if (!ADV || lane == MA_SEGNMR || lane == MA_EHS) {
      acc.store(cs->acc[lane]);
      if (chan == 0) ps->status[lane] = acc.status;
    }
  }
  if (chan == 0 && lane == 0) {
    ps->frame_counter = f_end;
    if (!ADV) ps->loudness_reached = loud_reached;
    ps->sig_energy = sh.energy[0];
    ps->noise_energy = sh.energy[1];
  }
  if (clk_wave && lane == 0) {                       // launches of one batch follow each other on one stream: one writer
    a.clk[0] += __builtin_readcyclecounter() - clk_s0;
    a.clk[1] += wall_clock64() - clk_w0;
  }
}

hipError_t launch_backend(const BackendArgs& a, unsigned n_pairs, hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  const dim3 block(64 * a.channels);
  PEAQ_DEV_SPIN_INSTEAD_OF_BACKEND(a, block, stream)
  if (!a.advanced && a.debug)
    hipLaunchKernelGGL((backend_kernel<109, false, true>), dim3(n_pairs), block, 0, stream, a);
  else if (a.debug)
    hipLaunchKernelGGL((backend_kernel<55, true, true>), dim3(n_pairs), block, 0, stream, a);
  else if (!a.advanced)
    hipLaunchKernelGGL((backend_kernel<109, false>), dim3(n_pairs), block, 0, stream, a);
  else
    hipLaunchKernelGGL((backend_kernel<55, true>), dim3(n_pairs), block, 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Filter-bank back end of the advanced version (gstpeaq.c:965-1010): 40 bands,
// one band per lane, blocks of 192 samples in order.
// ---------------------------------------------------------------------------
struct FbBackendShared {
  double pa[2][2][kPaStride];
  double acc[2][kAccFields][kAccLdsStride];
  int gate[2];
};

// DBG = true (peaq_debug_backend_advanced): the block's MOV values are computed for every block and written to
// a.debug; the arithmetic is the same instantiation otherwise.
// (held to 128 registers -- 44 B of them spilled around the block loop, none inside: beside the FP64 bank this kernel
// runs in the place of ONE of a CU's two bank workgroups, 272 registers per SIMD lane, and two of its waves fit there
// instead of one: + 0.5 % on the advanced pass)
template <bool DBG>
__global__ __launch_bounds__(128, 4) void fb_backend_kernel(FbBackendArgs a) {
  __shared__ FbBackendShared sh;
  __shared__ __attribute__((aligned(16))) double sh_ltab[2 * kLogTabEntries + 2];
  __shared__ double sh_etab[kExpTabEntries];
  constexpr int NB = kFbBands, SLOTS = 1;
  const int lane = threadIdx.x & 63;
  const int chan = threadIdx.x >> 6;
  const int channels = a.channels;
  const unsigned pair = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * kLogTabEntries; i += blockDim.x) sh_ltab[i] = a.common->log_tab[i >> 1][i & 1];
  if (threadIdx.x < kExpTabEntries) sh_etab[threadIdx.x] = a.common->exp_tab[threadIdx.x];
  __syncthreads();
  const GlobalTabs bt{a.bands, sh_ltab, sh_etab};
  const BandLane<NB, SLOTS> bl{lane};
  unsigned b_begin, b_end, slot = pair;
  if (a.windows) {                                   // broker launch: this session's own window and state
    const FbPairWindow w = a.windows[pair];
    b_begin = w.block0;
    b_end = w.block0 + w.n_blocks;
    slot = w.slot;
  } else {
    const unsigned n_blocks = a.n_blocks ? a.n_blocks[pair] : a.n_blocks_uniform;
    b_begin = a.block0;
    b_end = a.block0 + a.blocks_per_launch;
    if (b_end > n_blocks) b_end = n_blocks;
  }
  if (b_begin >= b_end) return;
  PairState* __restrict__ ps = a.state + slot;
  ChannelState* __restrict__ cs = &ps->ch[chan];

  const int bb = lane < kBandStride ? lane : 0;
  double la[6][SLOTS], mdr[3][SLOTS], mdt[3][SLOTS];
#pragma unroll
  for (int v = 0; v < 6; ++v) la[v][0] = cs->vec[kLaFiltRef + v][bb];
#pragma unroll
  for (int v = 0; v < 3; ++v) {
    mdr[v][0] = cs->vec[kModPrevRef + v][bb];
    mdt[v][0] = cs->vec[kModPrevTest + v][bb];
  }
  LaneAcc acc;
  {
    const int i = lane < kMaxAcc ? lane : 0;
    acc.load(&sh.acc[chan][0][lane < kAccLdsStride ? lane : kAccLdsStride - 1], cs->acc[i], acc_mode(true, i),
             ps->status[i]);
  }
  if (lane < kPaPad) sh.pa[chan][0][lane] = sh.pa[chan][1][lane] = 0.;
  wave_lds_fence();
  const bool owns = lane == MA_RMSMOD || lane == MA_NLASYM || lane == MA_LINDIST;
  unsigned loud_reached = ps->loudness_reached;

  // the block's values are requested one block ahead: the walk is a chain of dependent transcendental
  // arithmetic, a record load per block would add its full memory latency 320 times per launch
  const int lb = lane < NB ? lane : 0;
  struct BlockIn {
    double ur, ut, er, et, f0, f1;
  };
  auto fetch = [&](unsigned blk) {
    const double* __restrict__ rec0 =
        a.records + ((size_t)(pair * a.blocks_per_launch + (blk - b_begin)) * channels) * kFbRecDoubles;
    const double* __restrict__ rec = rec0 + (size_t)chan * kFbRecDoubles;
    BlockIn in;
    in.ur = rec[kFbRecUnsmRef + lb];
    in.ut = rec[kFbRecUnsmTest + lb];
    in.er = rec[kFbRecExcRef + lb];
    in.et = rec[kFbRecExcTest + lb];
    in.f0 = rec0[kFbRecFlags];
    in.f1 = channels == 2 ? rec0[kFbRecDoubles + kFbRecFlags] : 0.;
    return in;
  };
  BlockIn nxt = fetch(b_begin);
  for (unsigned blk = b_begin; blk < b_end; ++blk) {
    const BlockIn cur = nxt;
    if (blk + 1 < b_end) nxt = fetch(blk + 1);
    // boundary detector on the 192-sample block, any reference channel (gstpeaq.c:971-979)
    const bool above = cur.f0 != 0. || cur.f1 != 0.;
    if (owns) acc.set_tentative(!above);

    double ur[SLOTS], ut[SLOTS], er[SLOTS], et[SLOTS], lr[SLOTS], lt[SLOTS];
    ur[0] = cur.ur;
    ut[0] = cur.ut;
    er[0] = cur.er;
    et[0] = cur.et;
    lr[0] = bt.pow(ur[0], 0.3);                      // modpatt.c:235
    lt[0] = bt.pow(ut[0], 0.3);
    double ad_ref[SLOTS], ad_test[SLOTS], mr[SLOTS], mt[SLOTS];
    level_adapt<NB, SLOTS>(bl, bt, er, et, la, &sh.pa[chan][0][0], ad_ref, ad_test);
    modulation<NB, SLOTS>(bl, bt, lr, mdr, mr);
    modulation<NB, SLOTS>(bl, bt, lt, mdt, mt);
    double* __restrict__ dbg =
        DBG ? a.debug + ((size_t)(pair * a.blocks_per_launch + (blk - b_begin)) * channels + chan) * kDbgFbDoubles : nullptr;
    if (loud_reached == UINT_MAX) {                  // workgroup-uniform
      double n_ref, n_test;
      wave_sum2(total_loudness_part<NB, SLOTS>(bl, bt, er), total_loudness_part<NB, SLOTS>(bl, bt, et), n_ref, n_test);
      n_ref *= 24. / NB;
      n_test *= 24. / NB;
      if (lane == 0) sh.gate[chan] = (n_ref > 0.1 && n_test > 0.1);
      if (DBG && lane == 0) {
        dbg[5] = n_ref;
        dbg[6] = n_test;
      }
      __syncthreads();
      const int g = sh.gate[0] | (channels == 2 ? sh.gate[1] : 0);
      __syncthreads();
      if (g) loud_reached = blk;
    }
    double v0 = 0., w0 = 1.;
    bool hit = false;
    if (DBG || blk >= 125) {                         // gstpeaq.c:988-993
      double d1, d2, wt;
      mod_difference<NB, SLOTS>(bl, bt, 1., mr, mt, mdr[1], d1, d2, wt);
      d1 *= 100. / sqrt((double)NB);                 // MODE_RMS variant, movs.c:243-244
      if (blk >= 125 && lane == MA_RMSMOD) {
        v0 = d1;
        w0 = wt;
        hit = true;
      }
      if (DBG && lane == 0) {
        dbg[0] = d1;
        dbg[1] = wt;
      }
    }
    if (DBG || (blk >= 125 && blk - 13 >= loud_reached)) {    // gstpeaq.c:996-1007
      // movs.c:551-577; SWAP_MOD_PATTS_FOR_NOISE_LOUDNESS_MOVS (shipped: 1) exchanges the modulation
      // patterns of the missing-components term ...
      const bool swap = a.cfg.swap_mod_patts != 0;   // workgroup-uniform
      const double nl_p = noise_loudness_part<NB, SLOTS>(bl, bt, 2.5, 0.3, 1., mr, mt, ad_ref, ad_test);
      double lead[SLOTS] = {};                       // (ethres / stest)^0.23: the same stest in both calls below
      const double mc_p = noise_loudness_part<NB, SLOTS, GlobalTabs, LEAD_KEEP>(bl, bt, 1.5, 0.15, 1., swap ? mt : mr,
                                                                                swap ? mr : mt, ad_test, ad_ref, lead);
      // ... and (movs.c:679-706) takes the reference modulation twice; unadapted FB excitation
      const double ld_p = noise_loudness_part<NB, SLOTS, GlobalTabs, LEAD_USE>(bl, bt, 1.5, 0.15, 1., mr, swap ? mr : mt,
                                                                               ad_ref, er, lead);
      double nl, mc, ld, none;                       // the three sums over the bands in one reduction
      wave_sum4(nl_p, mc_p, ld_p, 0., nl, mc, ld, none);
      nl = noise_loudness_total<NB>(nl, 0.1);
      mc = noise_loudness_total<NB>(mc, 0.);
      ld = noise_loudness_total<NB>(ld, 0.);
      const bool open = blk >= 125 && blk - 13 >= loud_reached;
      if (open && lane == MA_NLASYM) {
        v0 = nl;
        w0 = mc;
        hit = true;
      }
      if (open && lane == MA_LINDIST) {
        v0 = ld;
        w0 = 1.;
        hit = true;
      }
      if (DBG && lane == 0) {
        dbg[2] = nl;
        dbg[3] = mc;
        dbg[4] = ld;
      }
    }
    if (hit) acc.add(v0, w0);
  }

  if (lane < kBandStride) {
#pragma unroll
    for (int v = 0; v < 6; ++v) cs->vec[kLaFiltRef + v][lane] = la[v][0];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      cs->vec[kModPrevRef + v][lane] = mdr[v][0];
      cs->vec[kModPrevTest + v][lane] = mdt[v][0];
    }
  }
  if (owns) {
    acc.store(cs->acc[lane]);
    if (chan == 0) ps->status[lane] = acc.status;
  }
  if (chan == 0 && lane == 0) {
    ps->fb_counter = b_end;
    ps->loudness_reached = loud_reached;
  }
}

hipError_t launch_fb_backend(const FbBackendArgs& a, unsigned n_pairs, hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  if (a.debug)
    hipLaunchKernelGGL(fb_backend_kernel<true>, dim3(n_pairs), dim3(64 * a.channels), 0, stream, a);
  else
    hipLaunchKernelGGL(fb_backend_kernel<false>, dim3(n_pairs), dim3(64 * a.channels), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// state initialisation (gstpeaq.c:357-361, movaccum.c:276-299, *_state_alloc: zeros)
// ---------------------------------------------------------------------------
__global__ void state_init_kernel(PairState* st, unsigned n_pairs) {
  const unsigned pair = blockIdx.x;
  if (pair >= n_pairs) return;
  PairState* ps = st + pair;
  double* raw = reinterpret_cast<double*>(ps);
  for (unsigned i = threadIdx.x; i < sizeof(PairState) / sizeof(double); i += blockDim.x) raw[i] = 0.;
  __syncthreads();
  if (threadIdx.x == 0) {
    ps->loudness_reached = UINT_MAX;
    for (int i = 0; i < kMaxAcc; ++i) ps->status[i] = kInit;
    for (int c = 0; c < 2; ++c)
      for (int i = 0; i < kMaxAcc; ++i) {
        // AVG_WINDOW history starts as NaN sentinels (movaccum.c:293)
        ps->ch[c].acc[i][3] = ps->ch[c].acc[i][4] = ps->ch[c].acc[i][5] = __builtin_nan("");
      }
  }
}

hipError_t launch_state_init(PairState* state, int /*advanced*/, unsigned n_pairs, hipStream_t stream) {
  if (n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(state_init_kernel, dim3(n_pairs), dim3(256), 0, stream, state, n_pairs);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// read-out: one thread per pair (movaccum.c:438-481, gstpeaq.c:1013-1078, nn.c)
// ---------------------------------------------------------------------------
__constant__ double nb_amin[11] = {393.916656, 361.965332, -24.045116, 1.110661, -0.206623, 0.074318,
                                   1.113683, 0.950345, 0.029985, 0.000101, 0.};
__constant__ double nb_amax[11] = {921, 881.131226, 16.212030, 107.137772, 2.886017, 13.933351,
                                   63.257874, 1145.018555, 14.819740, 1., 1.};
__constant__ double nb_wx[11][3] = {{-0.502657, 0.436333, 1.219602},  {4.307481, 3.246017, 1.123743},
                                    {4.984241, -2.211189, -0.192096}, {0.051056, -1.762424, 4.331315},
                                    {2.321580, 1.789971, -0.754560},  {-5.303901, -3.452257, -10.814982},
                                    {2.730991, -6.111805, 1.519223},  {0.624950, -1.331523, -5.955151},
                                    {3.102889, 0.871260, -5.922878},  {-1.051468, -0.939882, -0.142913},
                                    {-1.804679, -0.503610, -0.620456}};
__constant__ double nb_wxb[3] = {-2.518254, 0.654841, -2.207228};
__constant__ double nb_wy[3] = {-3.817048, 4.107138, 4.629582};
__constant__ double na_amin[5] = {13.298751, 0.041073, -25.018791, 0.061560, 0.02452};
__constant__ double na_amax[5] = {2166.5, 13.24326, 13.46708, 10.226771, 14.224874};
__constant__ double na_wx[5][5] = {{21.211773, -39.013052, -1.382553, -14.545348, -0.320899},
                                   {-8.981803, 19.956049, 0.935389, -1.686586, -3.238586},
                                   {1.633830, -2.877505, -7.442935, 5.606502, -1.783120},
                                   {6.103821, 19.587435, -0.240284, 1.088213, -0.511314},
                                   {11.556344, 3.892028, 9.720441, -3.287205, -11.031250}};
__constant__ double na_wxb[5] = {1.330890, 2.686103, 2.096598, -1.327851, 3.087055};
__constant__ double na_wy[5] = {-4.696996, -3.289959, 7.004782, 6.651897, 4.009144};

__global__ void finalize_kernel(const PairState* __restrict__ st, int advanced, int channels, unsigned n_pairs,
                                ResultRecord* __restrict__ out, int clamp_movs) {
  const unsigned pair = blockIdx.x * blockDim.x + threadIdx.x;
  if (pair >= n_pairs) return;
  const PairState* ps = st + pair;
  ResultRecord r;
  const int n_movs = advanced ? 5 : 11;
  for (int i = 0; i < 11; ++i) r.movs[i] = 0.;
  for (int i = 0; i < n_movs; ++i) {
    const int mode = acc_mode(advanced, i);
    // basic: ADB and MFPD have ONE channel (gstpeaq.c:580-584)
    const int nch = (!advanced && (i == MB_ADB || i == MB_MFPD)) ? 1 : channels;
    const bool tent = ps->status[i] == kTentative;
    double v = 0.;
    for (int c = 0; c < nch; ++c) v += acc_channel_value(mode, tent, ps->ch[c].acc[i]);
    r.movs[i] = v / nch;
  }
  double di;
  if (!advanced) {
    double x[3] = {nb_wxb[0], nb_wxb[1], nb_wxb[2]};
    for (int i = 0; i < 11; ++i) {
      double m = (r.movs[i] - nb_amin[i]) / (nb_amax[i] - nb_amin[i]);
      if (clamp_movs) m = m < 0. ? 0. : m > 1. ? 1. : m;                 // CLAMP_MOVS, nn.c:202-207
      for (int j = 0; j < 3; ++j) x[j] += nb_wx[i][j] * m;
    }
    di = -0.307594;
    for (int j = 0; j < 3; ++j) di += nb_wy[j] / (1 + exp(-x[j]));
  } else {
    double x[5];
    for (int j = 0; j < 5; ++j) x[j] = na_wxb[j];
    for (int i = 0; i < 5; ++i) {
      double m = (r.movs[i] - na_amin[i]) / (na_amax[i] - na_amin[i]);
      if (clamp_movs) m = m < 0. ? 0. : m > 1. ? 1. : m;                 // nn.c:320-325
      for (int j = 0; j < 5; ++j) x[j] += na_wx[i][j] * m;
    }
    di = -1.360308;
    for (int j = 0; j < 5; ++j) di += na_wy[j] / (1 + exp(-x[j]));
  }
  r.di = di;
  r.odg = -3.98 + (0.22 - -3.98) / (1 + exp(-di));          // nn.c:92-93,372-375
  r.totalsnr = 10 * log10(ps->sig_energy / ps->noise_energy);
  r.frames = (double)ps->frame_counter;
  r.fb_blocks = (double)ps->fb_counter;
  out[pair] = r;
}

hipError_t launch_finalize(const PairState* state, int advanced, int channels, unsigned n_pairs, ResultRecord* out,
                           hipStream_t stream, const Settings& cfg) {
  if (n_pairs == 0) return hipSuccess;
  hipLaunchKernelGGL(finalize_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, stream, state, advanced, channels,
                     n_pairs, out, cfg.clamp_movs);
  return hipGetLastError();
}

}  // namespace peaq
