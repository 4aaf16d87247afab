// peaq_device.h -- data layouts shared by the host code and the HIP kernels.
//
// Everything here lives in HBM.  Terminology follows the reference's domain:
// pair = one (reference, test) signal pair; frame = one 2048-sample FFT frame
// (hop 1024) of a pair; block = one 192-sample filter-bank block (advanced);
// band = critical band (109 basic / 55 advanced FFT model / 40 filter bank);
// MOV = model output variable.
#pragma once

#include <stdint.h>

namespace peaq {

constexpr int kFrame      = 2048;   // fftearmodel.c:51
constexpr int kHop        = 1024;   // fftearmodel.c:226
constexpr int kBins       = 1025;
constexpr int kBandStride = 112;    // band vectors are padded to 112 doubles
constexpr int kFbFrame    = 192;    // fbearmodel.c:48
constexpr int kFbBands    = 40;
constexpr int kFbRing     = 1456;   // fbearmodel.c:52
constexpr int kFbTaps     = 11;     // backward-masking FIR history (fbearmodel.c:262)
// its taps 0..5 (the filter is symmetric), cos^2(pi (i - 5) / 12) 0.9761 / 6 (fbearmodel.c:182-185) as the host's libm
// rounds them -- compile-time constants in the kernel (read from FbTables every tile's 12 requests stood in the way of
// the phase's LDS reads); peaq_tables.cpp checks them against its own evaluation
constexpr double kBackMask[6] = {0x1.6518acf1cfa3cp-7, 0x1.4d2ceb622adf4p-5, 0x1.4d2ceb622adf3p-4,
                                 0x1.f3c36113404ebp-4, 0x1.36db60930de4ep-3, 0x1.4d2ceb622adf1p-3};
// Filter bank on the matrix cores.  All 40 filters are centred on the same delay
// (D + N/2 = 729 samples) and h(N - n) = conj(h(n)), so the bank is two GEMMs over the delays
// d = 1..729 only: the real parts against x[t-d] + x[t-(1458-d)], the imaginary parts against
// x[t-d] - x[t-(1458-d)].  3 row tiles of 16 bands (longest first); first delay, K steps (4 delays
// each) and offset (in K steps) of every tile in FbTables::mf_re / mf_im.
constexpr int kFbCentre = 729;
constexpr int kMfTiles = 3;
constexpr int kMfD0[kMfTiles]    = {2, 470, 677};
constexpr int kMfSteps[kMfTiles] = {182, 65, 14};
constexpr int kMfBase[kMfTiles]  = {0, 182, 247};
constexpr int kMfTotalSteps = 261;
// The same bank on the FP16 matrix instruction (v_mfma_f32_16x16x32_f16: 32 delays per instruction at
// 15 x the MAC rate of the FP32 one) with both operands split into a high and a low FP16 part, three
// products per term (hi hi, hi lo, lo hi: 22 bits), and the fold undone -- re = Hre X1 + Hre X2,
// im = Him X1 - Him X2 with X1[d][t] = x[32 t - d], X2[d][t] = x[32 t - (1458 - d)] -- so that the B
// operands are plain 16-byte reads of the window.  A row tile's delays are cut into blocks of 32 that start
// at kHfD1 (X1, = 1 mod 8) and kHfD2 (X2, = 2 mod 8): that puts both operands' 8-sample groups on 16-byte
// boundaries of the window; delays outside a band's filter carry zero coefficients.
constexpr int kHfBlocks[kMfTiles] = {23, 9, 2};
constexpr int kHfBase[kMfTiles]   = {0, 23, 32};
constexpr int kHfD1[kMfTiles]     = {1, 465, 673};
constexpr int kHfD2[kMfTiles]     = {2, 466, 674};
constexpr int kHfTotalBlocks = 34;
enum { HF_RE_HI_1, HF_RE_LO_1, HF_IM_HI_1, HF_IM_LO_1, HF_RE_HI_2, HF_RE_LO_2, HF_NIM_HI_2, HF_NIM_LO_2, HF_OPERANDS };
// The FP64 engine's bank (peaq_fb.hip, "block-sum form"): the Hann window of a filter is three rectangular
// windows, 4/N cos^2(pi m/N) e^(-j w m) = sum_i g_i e^(-j w_i m) with w_i = w, w + 2 pi/N, w - 2 pi/N, so the
// part of a filter's window that consists of WHOLE 32-sample blocks of the signal is, per exponential, a
// running sum along the outputs (one every 32 samples): V(t) = e^(j 32 w_i) V(t-1) + enter(t) - leave(t), where
// enter is the dot product of the block that became whole with a fixed row of 32 coefficients and leave that of
// the block that stopped being whole -- the same block's enter value of J outputs ago (J = the window's whole
// blocks) times the constant e^(j 32 w_i J), so it is taken from a history of the last J enter values
// (FbSignalState::bs_hist) instead of being computed again.  Only the two blocks the window's ends fall into
// need the filter's own coefficients.  Per band and output that is 8 rows of 32 multiply-adds on the matrix
// cores (6 enter, 2 for the block the window ends in) and one row on the vector ALU (the block it starts in),
// whatever the filter's length, against N for the direct form: the bands 0 .. kBsBands-1 (N >= 262) run this
// way, two bands per 16-row tile of the matrix instruction, the rest as one direct (folded) GEMM tile.
constexpr int kBsBands  = 24;
constexpr int kBsPairs  = kBsBands / 2;
constexpr int kBsChains = kBsBands * 3;
constexpr int kBsHistOrg = 8;     // a history row starts with eight zeros (outputs in front of a window's first whole block)
constexpr int kBsHist   = kBsHistOrg + 48;   // ... followed by up to 45 enter values: the longest run of whole blocks
// Rows of a pair's 16-row tile: the enter rows first (re of the six chains -- chain = 3 * band-in-pair + exponential --,
// then im), then the rows of the block the windows end in (re of both bands, im of both bands).  ty as in the
// tables' construction: 2 * exponential + {re, im} for ty < 6, 6 / 7 = re / im of the end block.
constexpr int bs_row(int sub, int ty) { return ty < 6 ? (ty & 1) * 6 + 3 * sub + (ty >> 1) : 12 + 2 * (ty - 6) + sub; }
// The running sums are evaluated with a lane per (chain, segment of kBsSeg consecutive outputs): eight segments per
// chain, shifted so that output J = bs_whole[band] starts a segment (the segments in front of it take the values that
// leave from the history, the others from this tile) -- s = bs_seg_s[band] outputs of the first segment lie in front
// of output 0 and read zeros.
constexpr int kBsSeg = 9;
// the direct tile: bands 24 .. 39, delays kMfdD0 .. 729 in kMfdSteps K steps of four
constexpr int kMfdBand0 = 24;
constexpr int kMfdD0    = 611;
constexpr int kMfdSteps = 30;

// ---- constant tables (built on the host in FP64, peaq_tables.cpp) ----------
struct CommonTables {
  // Hann window, fftearmodel.c:167-172, in the form the front end evaluates it: sample k = 2 (lane + 64 r) + j sits at the
  // angle theta(2 lane + j) + r * 2 pi 128 / 2047, so w = A - A cos(theta_lane) C_r + A sin(theta_lane) S_r
  // with A = sqrt(8/3) / 2 and 16 compile-time (C_r, S_r): 2 KB of table per wave instead of 16 KB
  double hann_lane[64][4];      // { A cos th(2l), A sin th(2l), A cos th(2l+1), A sin th(2l+1) }
  double ear_w2[kBins];         // fftearmodel.c:253-256
  double tw_re[kFrame];         // exp(-2 pi i k / 2048), k = 0..2047 (filter-bank tables, tools)
  double tw_im[kFrame];
  // The front end's twiddle factors, lane-major: every load is one coalesced 16-byte access per lane
  // (gathers from tw_re / tw_im cost the vector-memory pipe up to 64 cycles per instruction, and that
  // pipe is what bounds the kernel).  Entry e of lane l is W_2048^(kTwLaneStride[e] * (l & kTwLaneMask[e])):
  //   0 W_2048^l   1 W_1024^l   2 W_256^(l&15)   3 W_512^l   4 W_64^(l&7)   5 W_16^(l&3)   6 W_64^(l&15)   7 W_256^l
  // higher powers and the r-dependent parts are products with compile-time constants.
  double tw_lane[8][64][2];
  double ear_w2_pair[8][64][2]; // ear_w2 of bin l + 64 q and of its mirror bin (spec_bin(8 + q, l)), q = 0..7
  double ehs_window[256];       // movs.c:1366-1367
  double ehs_window_centred[256];   // movs.c:1363-1364 (CENTER_EHS_CORRELATION_WINDOW)
  // Logarithm by table (log_tab, peaq_wave.h): x = m 2^e with m in [0.5, 1); bin i = round(128 (2 m - 1)),
  // i = 0..128, holds the mantissas with 2 m in [C - 1/256, C + 1/256) around the centre C = 1 + i / 128:
  //   [i][0] = 2 / C              r = m [i][0] - 1 = 2 m / C - 1,  |r| <= 2^-8
  //   [i][1] = ln C - ln 2        (i >= kLogTabFold: ln x = e ln 2 + [i][1] + log1p(r))
  //            ln C               (i <  kLogTabFold: the same with e - 1 -- the lower bins are read as
  //                                [1, sqrt 2) of the binade BELOW)
  // Arguments around 1 fall into bin 0 (from above) or bin 128 (from below), whose entries are exactly
  // {2, 0} and {1, 0}: there r = x - 1 exactly and the power of two counts 0, so ln 1 = 0 and the
  // relative accuracy holds however close to 1 the argument is.
  double log_tab[130][2];
  // Exponential by table (exp_tab, peaq_wave.h): e^x = 2^k 2^(j/64) e^r with x = (64 k + j) ln 2 / 64 + r,
  // |r| <= ln 2 / 128: [j] = 2^(j/64), rounded to nearest
  double exp_tab[64];
};
constexpr int kLogTabEntries = 129;
constexpr int kExpTabEntries = 64;
constexpr int kLogTabFold = 54;     // first bin whose centre lies above sqrt 2

struct BandTables {             // earmodel.c:279-323 + fftearmodel.c:693-788
  int    bands;
  int    step;                  // hop in samples
  double delta_z;
  double dz02;                  // 0.2 * delta_z
  double aLe;                   // (1/a_L)^0.4
  double deriv_factor;          // 48000 / step (modpatt.c:231)
  double fc[kBandStride];
  double internal_noise[kBandStride];
  double noise_pow03[kBandStride];      // internal_noise^0.3 (movs.c:243)
  double ear_tc[kBandStride];
  double adapt_tc[kBandStride];
  double exc_threshold[kBandStride];
  double threshold[kBandStride];
  double loud_factor[kBandStride];
  double ln_internal_noise[kBandStride];   // ln(internal_noise): x^0.23 with that base as exp(0.23 (ln a - ln b))
  double inv_window_count[kBandStride];    // 1 / (bands in the averaging window of leveladapter.c:315-328)
  // FFT model only
  int    lo[kBandStride], hi[kBandStride];
  double wlo[kBandStride], whi[kBandStride];
  double ln_aUC[kBandStride];
  double gIL[kBandStride];
  double inv_spread_norm[kBandStride];
  double inv_spread_norm_pow03[kBandStride];   // inv_spread_norm^0.3 (for E^0.3)
  double mask_diff[kBandStride];
};

struct FbTables {               // fbearmodel.c:57-61,182-225
  int    flen[kFbBands];
  int    delay[kFbBands];       // D = 1 + (1456 - N)/2
  int    coef_off[kFbBands];    // offset of band's coefficients in h_re/h_im
  double back_mask[6];
  double h_re[12000];           // sum(N/2+1) = 10954 coefficients
  double h_im[12000];
  // A operands of the two GEMMs (see kMf* above): step s of tile r, lane = band_in_tile + 16 (d - d0 - 4 s);
  // the centre tap carries half its weight (its two "mirror" samples coincide)
  double mf_re[kMfTotalSteps * 64];
  double mf_im[kMfTotalSteps * 64];
  float  mf_re_f[kMfTotalSteps * 64];   // the same, rounded, for the FP32 matrix instruction
  float  mf_im_f[kMfTotalSteps * 64];
  // FP16 x 3 form: A operands [block][operand][lane][8] as raw FP16 bits -- lane = band_in_tile + 16 kg holds the
  // delays kHfD1 + 32 s + 8 kg + (7 - e) (X1: the window is read in ascending sample order) or
  // kHfD2 + 32 s + 8 kg + e (X2), e = 0..7, scaled by 2^hf_exp[band] -- and the factor that undoes that scale
  unsigned short hf[kHfTotalBlocks][HF_OPERANDS][64][8];
  double hf_unscale[kMfTiles * 16];     // 2^-hf_exp per band row (0 for the rows beyond band 39)
  // ---- block-sum form (kBs* above).  Window columns are blocks of 32 samples (column c = samples 32 c .. 32 c + 31
  // of the kernel's window, whose sample 727 + 32 t is the centre of every filter at output t).  Band b's window at
  // t = 0 is [728 - N/2, 726 + N/2]: cL = first column it touches, cR = last; columns cL+1 .. cR-1 are whole.
  int    bs_col_head[kBsPairs];         // first column of the pair's tile at t = 0: min(cR) - 1
  int    bs_off_enter[kBsBands];        // column of band b's enter rows at output t: bs_col_head + t + this (edge rows: + 1)
  int    bs_col_left[kBsBands];         // cL: column (at t = 0) of the block the window starts in
  int    bs_left_q0[kBsBands];          // first sample of that block inside the window
  // When fewer than half of that block's samples lie OUTSIDE the window (q0 < 16), the block is counted among the
  // whole ones (bs_whole = J + 1) and what it has too much -- samples 0 .. q0 - 1 times the periodic continuation
  // of the window, which is what the three exponentials add up to there -- is taken off instead: bs_left then holds
  // minus those coefficients.  Either way at most 16 taps; bs_left_g = the groups of eight taps that are not all zero.
  int    bs_left_g[kBsBands][2];        // [first group, end)
  int    bs_whole[kBsBands];            // J = cR - 1 - cL whole columns
  // A operands: [pair][K step][lane = row + 16 (k mod 4)], row = 8 (band in pair) + type,
  // type 0..5 = re, im of the three exponentials' enter rows, 6, 7 = re, im of the block the window ends in
  double bs_coef[kBsPairs][8][64];
  double bs_left[kBsBands][32][2];      // the filter's own coefficients (re, im) on the block the window starts in
  double bs_rot[kBsChains][2][2];       // chain = 3 band + i: rot = e^(j 32 w_i) and rot^J (re, im)
  double bs_pow[kBsChains][64][2];      // rot^(l + 1), l = 0..63
  int    bs_seg_s[kBsBands];            // (kBsSeg - J % kBsSeg) % kBsSeg
  // per pair and lane (chain c = min(lane >> 3, 5) of the pair, segment g = lane & 7), complex:
  //   [0..2] rot^(kBsSeg 2^l), l = 0..2, the weights of the scan over a chain's segments (0 where g < 2^l)
  //   [3] rot^(kBsSeg g - s)   [4] rot   [5] rot^J
  double bs_seg[kBsPairs][6][64][2];
  double mfd_re[kMfdSteps * 64];        // the direct tile's A operands, lane = band - 24 + 16 (d - kMfdD0 - 4 s)
  double mfd_im[kMfdSteps * 64];
  double log_tab[130][2];               // CommonTables::log_tab again: the bank kernel's slope exponents take their logarithms
                                        // from a copy in LDS (peaq_fb.hip), and its arguments carry no CommonTables
};

// ---- per-frame record: front end -> back end --------------------------------
// The reference's compile-time readings of BS.1387 (settings.h:47-97) as run-time switches; the defaults
// are the values the reference ships with (include/peaq_amd.h, peaq_settings).
struct Settings {
  int swap_mod_patts = 1;        // SWAP_MOD_PATTS_FOR_NOISE_LOUDNESS_MOVS   (movs.c:566-575, 693-703)
  int centre_ehs_window = 0;     // CENTER_EHS_CORRELATION_WINDOW            (movs.c:1362-1368)
  int ehs_dc_before_window = 1;  // EHS_SUBTRACT_DC_BEFORE_WINDOW            (movs.c:1409-1433)
  int floor_steps = 0;           // USE_FLOOR_FOR_STEPS_ABOVE_THRESHOLD      (movs.c:1256-1260)
  int clamp_movs = 0;            // CLAMP_MOVS                               (nn.c:202-207, 320-325)
  int swap_slope = 0;            // SWAP_SLOPE_FILTER_COEFFICIENTS           (fbearmodel.c:335-339)
};

// One record per (pair, frame, channel), kRecDoubles doubles: what the stateless front end hands to the
// back end.  The excitation travels as ONE number per band and signal, root = E2^(1/4) of the spread band
// energy E2 (fftearmodel.c:556-597): the unsmeared excitation E = E2^2.5 / norm and E^0.3 (modpatt.c:235)
// are both a few multiplications away (excitation_from_root), so neither needs its own vector.
constexpr int kRecRootRef   = 0 * kBandStride;
constexpr int kRecRootTest  = 1 * kBandStride;
constexpr int kRecNoise     = 2 * kBandStride;   // noise in bands (movs.c:992-1000)
constexpr int kRecScalars   = 3 * kBandStride;   // 336
constexpr int kRecBwRef     = kRecScalars + 0;
constexpr int kRecBwTest    = kRecScalars + 1;
constexpr int kRecEhs       = kRecScalars + 2;   // EHS of this frame (not yet x1000)
constexpr int kRecFlagsRef  = kRecScalars + 3;   // bit0 above-threshold (ref), bit1 energy(ref)
constexpr int kRecFlagsTest = kRecScalars + 4;   // bit1 energy(test)
constexpr int kRecSigE      = kRecScalars + 5;   // sum ref^2 over the hop (totalsnr)
constexpr int kRecNoiseE    = kRecScalars + 6;   // sum (ref-test)^2
constexpr int kRecDoubles   = 352;

// E = E2^2.5 / norm = root^10 inv_norm (fftearmodel.c:593-597) and E^0.3 = root^3 inv_norm^0.3
#if defined(__HIPCC__)
__host__ __device__
#endif
inline void excitation_from_root(double root, double inv_norm, double inv_norm_pow03, double& unsm, double& loud) {
  const double r1 = root * root, e2 = r1 * r1;
  unsm = e2 * e2 * r1 * inv_norm;
  loud = r1 * root * inv_norm_pow03;
}

// The layout peaq_debug_frontend / peaq_debug_backend show to the parity tests (include/peaq_amd.h,
// PEAQ_DEBUG_RECORD_DOUBLES): the record with both derived vectors written out.
constexpr int kPubUnsmRef   = 0 * kBandStride;   // unsmeared excitation, reference
constexpr int kPubUnsmTest  = 1 * kBandStride;
constexpr int kPubLoudRef   = 2 * kBandStride;   // unsmeared^0.3 (modpatt.c:235)
constexpr int kPubLoudTest  = 3 * kBandStride;
constexpr int kPubNoise     = 4 * kBandStride;
constexpr int kPubScalars   = 5 * kBandStride;   // 560: the scalars in the order of kRecScalars
constexpr int kPubDoubles   = 576;               // == PEAQ_DEBUG_RECORD_DOUBLES

// stage-level dump of the back end (peaq_debug_backend): per (frame, channel) the patterns the
// reference's pattern layer hands to the MOV layer
constexpr int kDbgExcRef       = 0 * kBandStride;   // excitation after time smearing (fftearmodel.c:496-504)
constexpr int kDbgExcTest      = 1 * kBandStride;
constexpr int kDbgAdaptRef     = 2 * kBandStride;   // spectrally adapted patterns (leveladapter.c:331-336)
constexpr int kDbgAdaptTest    = 3 * kBandStride;
constexpr int kDbgModRef       = 4 * kBandStride;   // modulation (modpatt.c:245-247)
constexpr int kDbgModTest      = 5 * kBandStride;
constexpr int kDbgAvgLoudRef   = 6 * kBandStride;   // average loudness (modpatt.c:240-243)
constexpr int kDbgAvgLoudTest  = 7 * kBandStride;
constexpr int kDbgLoudnessRef  = 8 * kBandStride;   // total loudness while the gate is closed (earmodel.c:891-907)
constexpr int kDbgLoudnessTest = 8 * kBandStride + 1;
// the MOV layer's per-frame values BEFORE accumulation, computed for every frame in the debug instantiation
// (the accumulators only see them when the gates of gstpeaq.c:871,880-881 are open):
constexpr int kDbgMov          = 8 * kBandStride + 8;   // (advanced, 55 bands: 3 = SegmentalNMR's 10 log10 of the mean, movs.c:1010-1020)
                                                        // + 0 ModDiff1, 1 ModDiff2, 2 TempWt (movs.c:205-254), 3 noise
                                                        // loudness (:354-371), 4 mean and 5 maximum of the band NMRs
                                                        // (:971-1023); channel 0 only: 6 detection probability, 7 steps
                                                        // above threshold (:1224-1276)
constexpr int kDbgDoubles      = 8 * kBandStride + 16;  // == PEAQ_DEBUG_BACKEND_DOUBLES

// stage-level dump of the filter-bank back end (peaq_debug_backend_advanced): per (block, channel) the MOV layer's
// values BEFORE accumulation, computed for every block in the debug instantiation (the accumulators only see them
// when the gates of gstpeaq.c:988,996-997 are open):
//   0 RmsModDiff (movs.c:205-254 with the RMS normalisation of :243-244)   1 its weight (TempWt, level weight 1)
//   2 noise loudness of RmsNoiseLoudAsym (movs.c:551-577)   3 its missing-components term (the accumulator's weight)
//   4 AvgLinDist (movs.c:679-706)   5, 6 total loudness of ref / test while the gate is closed (earmodel.c:891-907)
constexpr int kDbgFbDoubles = 8;                      // == PEAQ_DEBUG_ADV_BLOCK_DOUBLES

// filter-bank record per (pair, block, channel): unsmeared/excitation of both
// signals + above-threshold flag
constexpr int kFbRecUnsmRef  = 0;
constexpr int kFbRecUnsmTest = 40;
constexpr int kFbRecExcRef   = 80;
constexpr int kFbRecExcTest  = 120;
constexpr int kFbRecFlags    = 160;
constexpr int kFbRecDoubles  = 168;

// ---- recurrent state of one pair --------------------------------------------
constexpr int kAccFields = 12;  // num, den, num2, past0..2, max, filt, saved num, den, num2, max
constexpr int kMaxAcc    = 11;

enum AccMode { kAvg = 0, kAvgLog, kRms, kRmsAsym, kAvgWindow, kFilteredMax, kAdb };
enum AccStatus { kInit = 0, kNormal, kTentative };

enum StateVec {                 // band vectors per channel
  kSmearRef = 0, kSmearTest,    // fftearmodel.c:498-500 filtered excitation (FFT model)
  kLaFiltRef, kLaFiltTest, kLaNum, kLaDen, kLaPcRef, kLaPcTest,     // leveladapter.c:57-70
  kModPrevRef, kModLoudRef, kModDLoudRef,                           // modpatt.c:57-66
  kModPrevTest, kModLoudTest, kModDLoudTest,
  kStateVecs
};

struct ChannelState {
  double vec[kStateVecs][kBandStride];
  double acc[kMaxAcc][kAccFields];
};

struct PairState {
  uint32_t frame_counter;       // gstpeaq.c:124
  uint32_t fb_counter;          // gstpeaq.c:125
  uint32_t loudness_reached;    // gstpeaq.c:126, starts at UINT_MAX
  uint32_t pad0;
  int32_t  status[kMaxAcc];     // movaccum.c:95-108
  int32_t  pad1;
  double   sig_energy;          // gstpeaq.c:137-138
  double   noise_energy;
  ChannelState ch[2];
};

// filter-bank ear-model state of one (pair, channel, signal)  (fbearmodel.c:93-107)
struct FbSignalState {
  double hp[6];                 // hpfilter1_x1,x2,y1,y2, hpfilter2_y1,y2
  double cu[kFbBands];
  double e0_hist[kFbBands][kFbTaps];
  double excitation[kFbBands];
  double ring[kFbRing];         // the last 1456 filtered samples, newest first
  // Largest |filtered sample| for the split-FP16 operands' scale (peaq_fb.hip): slot launch_idx % 3 holds
  // { this launch's blocks, the window's head = the 1456 filtered samples in front of them } -- three slots because
  // the high-pass walk of launch i + 1 runs beside the bank kernel of launch i; peak_last = the last launch's own peak.
  double peak_slot[3][2];
  double peak_last;
  // FP64 engine, block-sum form of the long filters (kBs*): per band and exponential the last J = bs_whole[band]
  // enter values (re, im), oldest first, behind kBsHistOrg zeros that are never written -- what the direct form's
  // delay line is to the samples
  double bs_hist[kBsBands][6][kBsHist];
};

}  // namespace peaq
