// peaq_kernels.h -- launch interfaces of the HIP kernels (host side sees plain
// functions; the kernels themselves live in the .hip files).
#pragma once
#include <hip/hip_runtime.h>

#include "peaq_device.h"

namespace peaq {

// ---- FFT ear-model front end -------------------------------------------------
struct FrontendArgs {
  const float* ref;             // [pair][pair_stride][channels] interleaved F32
  const float* test;
  size_t pair_stride;           // samples (per channel) between consecutive pairs
  const uint32_t* n_ref;        // per-pair lengths in samples per channel (device), or nullptr
  const uint32_t* n_test;
  uint32_t n_uniform_ref;       // lengths used when n_ref/n_test are null
  uint32_t n_uniform_test;
  const uint32_t* n_frames;     // per-pair total frame count (device), or nullptr
  uint32_t n_frames_uniform;
  // frame f starts at buffer sample (f - frame_origin) * 1024 + off_{ref,test}
  // (batch: all zero; sessions: the staging buffer holds a window of the stream)
  unsigned frame_origin;
  long long off_ref, off_test;
  int channels;
  unsigned frame0;              // first frame of this launch
  unsigned frames_per_launch;
  unsigned fpl_magic;           // ceil(2^32 / frames_per_launch): x / frames_per_launch = mulhi(x, magic), exact while
                                // x (magic d - 2^32) < 2^32 -- launch_frontend fills it in and refuses launches beyond that
  double level_factor;          // fftearmodel.c:312-313
  const CommonTables* common;
  const BandTables* bands;      // FFT model, 109 or 55 bands
  double* records;              // [pair][frame - frame0][channel][kRecDoubles]
  // Broker launches (many live sessions in one grid): every "pair" of the launch is a session
  // at its own position in its stream.  When set, pair p processes frames
  // [pair_frame0[p], pair_frame0[p] + pair_nframes[p]) and its buffer starts at that frame.
  const uint32_t* pair_frame0;
  const uint32_t* pair_nframes;
  // -DPEAQ_FE_PROFILE builds only (tools/fe_profile.py): [2 waves][16 phases] cycle sums + [32] wave count
  unsigned long long* prof;
  Settings cfg;                 // centre_ehs_window, ehs_dc_before_window
};
hipError_t launch_frontend(int bands, const FrontendArgs& a, unsigned n_pairs, hipStream_t stream);
// Most frames per launch for which the kernel's reciprocal multiplication is exact whatever the divisor:
// the error term magic d - 2^32 is below d and the dividend below n_pairs d, so n_pairs d^2 <= 2^32 is enough.
inline unsigned max_frames_per_launch(unsigned n_pairs) {
  unsigned long long d = 65535;
  while (d > 1 && (unsigned long long)(n_pairs ? n_pairs : 1) * d * d > (1ull << 32)) d >>= 1;
  return (unsigned)d;
}

// ---- pattern back end (time smearing .. MOV accumulation) -------------------
struct BackendArgs {
  const double* records;        // as written by the front end
  unsigned frame0, frames_per_launch;
  const uint32_t* n_frames;
  uint32_t n_frames_uniform;
  int channels;
  int advanced;                 // 0: basic (109 bands, 11 MOVs); 1: FFT part of advanced (55 bands)
  const CommonTables* common;   // log_tab
  const BandTables* bands;
  PairState* state;             // [pair]
  // broker launches: per-pair frame window and state slot (see FrontendArgs)
  const uint32_t* pair_frame0;
  const uint32_t* pair_nframes;
  const uint32_t* pair_slot;    // state index of pair p (nullptr: p)
  double* debug;                // basic version only: [pair][frame - frame0][channel][kDbgDoubles], or nullptr
  Settings cfg;                 // floor_steps
  // batch launches only (nullptr otherwise): workgroup 0 adds the shader-clock ticks and the constant-rate ticks of
  // its own lifetime to clk[0], clk[1] -- the clock the device held while the step ran (peaq_batch_last_clock)
  unsigned long long* clk;
};
hipError_t launch_backend(const BackendArgs& a, unsigned n_pairs, hipStream_t stream);

// ---- read-out: accumulators -> MOVs -> DI -> ODG --------------------------------
struct ResultRecord {           // mirrors peaq_result in include/peaq_amd.h
  double movs[11];
  double di, odg, totalsnr, frames, fb_blocks;
};
hipError_t launch_finalize(const PairState* state, int advanced, int channels, unsigned n_pairs,
                           ResultRecord* out, hipStream_t stream, const Settings& cfg = Settings());   // clamp_movs
hipError_t launch_state_init(PairState* state, int advanced, unsigned n_pairs, hipStream_t stream);

// ---- advanced mode: filter-bank ear model ------------------------------------------
// Broker launches: "pair" p of the launch is a live session at its own position in its stream.
struct FbPairWindow {
  uint32_t block0;              // index of the session's first block of this launch (0 = its very first block)
  uint32_t n_blocks;            // blocks of this session in this launch
  uint32_t prev_blocks;         // n_blocks of the session's previous launch (where its history tail sits)
  uint32_t slot;                // index of the session's filter state, rows and PairState
};

struct FbFrontArgs {
  const float* ref;
  const float* test;
  size_t pair_stride;
  const uint32_t* n_ref;        // per-pair lengths (device) or nullptr
  const uint32_t* n_test;
  uint32_t n_uniform_ref, n_uniform_test;
  const uint32_t* n_blocks;     // per-pair total block count (device) or nullptr
  uint32_t n_blocks_uniform;
  // block b starts at buffer sample (b - block_origin) * 192 + off_{ref,test}
  unsigned block_origin;
  long long off_ref, off_test;
  int channels;
  unsigned block0, blocks_per_launch;
  unsigned launch_idx;          // running index of this launch (selects the peak slot, FbSignalState::peak_slot)
  unsigned prev_blocks;         // blocks_per_launch of the previous launch on these rows
  int first_launch;             // no history yet: the filter-bank delay line starts at zero
  double level_factor;          // fbearmodel.c:252-253
  const BandTables* bands;      // 40 bands
  const FbTables* fb;
  FbSignalState* fbstate;       // [pair][channel][2]
  double* hp_scratch;           // rows [signal][hp_row_stride]: 1456 history + filtered samples of the launch
  const double* hp_prev;        // rows of the previous launch (history source); nullptr: hp_scratch itself
  size_t hp_row_stride;
  double* records;              // [pair][block - block0][channel][kFbRecDoubles]
  const FbPairWindow* windows;  // broker launches (see above); nullptr: the uniform fields apply, slot = pair
  Settings cfg;                 // swap_slope
  int fir_fp64;                 // arithmetic of the FIR bank (peaq_fb.hip): 0 = v_mfma_f32, 1 = v_mfma_f64 (exact to the oracle's
                                // 1e-9), 2 = v_mfma_f32_16x16x32_f16 on operands split in two FP16 parts (three products per term)
  double hf_xscale, hf_xunscale;   // mode 2: power of two that puts the filtered signal's full scale at 2^10..2^11, and its inverse
};
hipError_t launch_fb_frontend(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream);   // high-pass + filter bank
hipError_t launch_fb_hp(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream);
hipError_t launch_fb_bank(const FbFrontArgs& a, unsigned n_pairs, hipStream_t stream);

struct FbBackendArgs {
  const double* records;
  unsigned block0, blocks_per_launch;
  const uint32_t* n_blocks;
  uint32_t n_blocks_uniform;
  int channels;
  const CommonTables* common;   // log_tab
  const BandTables* bands;
  PairState* state;
  const FbPairWindow* windows;  // broker launches
  Settings cfg;                 // swap_mod_patts
  // stage tests only (peaq_debug_backend_advanced): [pair][block - block0][channel][kDbgFbDoubles] -- the block's MOV
  // values before accumulation, computed for EVERY block in the debug instantiation -- or nullptr
  double* debug;
};
hipError_t launch_fb_backend(const FbBackendArgs& a, unsigned n_pairs, hipStream_t stream);

// ---- synthetic workload --------------------------------------------------------------
hipError_t launch_synth(uint32_t seed0, unsigned n_pairs, int channels, uint32_t n_samples, size_t pair_stride,
                        float* ref, float* test, hipStream_t stream);

}  // namespace peaq
