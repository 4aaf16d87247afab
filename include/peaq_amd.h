/* peaq_amd.h -- C ABI of the MI355X-native PEAQ engine (libpeaq_amd.so).
 *
 * This is the drop-in boundary of SURVEY.md 8(b): everything the `peaq`
 * GStreamer element of HSU-ANT/gstpeaq calls below its adapters -- ear models,
 * level/pattern adaptation, modulation patterns, MOV calculators, MOV
 * accumulators and the neural network (reference src/{fftearmodel,fbearmodel,
 * earmodel,leveladapter,modpatt,movs,movaccum,nn}.c) -- is replaced by batched
 * HIP kernels for gfx950 behind the entry points declared here.  Plain C,
 * plain pointers and sizes; no GLib, GStreamer or torch types.
 *
 * Two ways in:
 *   session API  one handle per element instance / per (ref,test) stream; the
 *                element's pad_chain / change_state / get_property call it
 *                (gstpeaq_amd/gst/gstpeaq_amd.c is that element).
 *   batch API    N whole (ref,test) pairs resident in device memory, the shape
 *                of BASELINE.json configs 2-4.
 *
 * All functions return PEAQ_OK (0) or a negative error; peaq_last_error()
 * gives the message of the calling thread's last failure.  The reference's DSP
 * calls cannot fail (SURVEY.md 8(b) "error convention"); device/allocation
 * errors are new and map to GST_FLOW_ERROR in the element.
 */
#ifndef PEAQ_AMD_H
#define PEAQ_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PEAQ_OK            0
#define PEAQ_ERR_ARG      -1   /* bad argument */
#define PEAQ_ERR_DEVICE   -2   /* HIP runtime error (no GPU, launch failure, ...) */
#define PEAQ_ERR_NOMEM    -3
#define PEAQ_ERR_STATE    -4   /* call not valid in the handle's current state */

#define PEAQ_MOVS_BASIC     11  /* order = enum _MovBasic, gstpeaq.c:95-108 */
#define PEAQ_MOVS_ADVANCED   5  /* order = enum _MovAdvanced, gstpeaq.c:86-93 */

/* One result record per pair (batch API) -- 16 doubles.
 * movs[] holds 11 (basic) or 5 (advanced) values, the rest is 0. */
typedef struct {
  double movs[PEAQ_MOVS_BASIC];
  double di;          /* peaq_calculate_di_basic/_advanced, nn.c:187,304 */
  double odg;         /* peaq_calculate_odg, nn.c:372 */
  double totalsnr;    /* element property "totalsnr", gstpeaq.c:493-497 */
  double frames;      /* FFT frame-pairs processed (frame_counter, gstpeaq.c:920) */
  double fb_blocks;   /* filter-bank blocks processed (advanced; gstpeaq.c:1009) */
} peaq_result;

const char *peaq_last_error (void);
/* Number of frames the element processes for signals of n_ref / n_test samples per
 * channel: full frames while both adapters hold one (do_processing,
 * gstpeaq.c:596-611) plus ONE zero-padded frame if anything is left on either
 * side (do_flush, :716-745).  filter_bank = 0: FFT frames (2048 / hop 1024);
 * 1: filter-bank blocks (192 / 192).  Pure host arithmetic, no GPU needed. */
uint32_t peaq_frame_count (uint64_t n_ref, uint64_t n_test, int filter_bank);
/* "x.y.z gfx950" */
const char *peaq_version (void);

/* ---- device context -------------------------------------------------------
 * Owns the constant tables in HBM (Hann window, ear weights, band tables,
 * twiddles, filter-bank impulse responses -- what the reference builds in
 * fftearmodel.c:160-173,240-257,693-788, fbearmodel.c:188-225,
 * earmodel.c:279-323) for one GPU.  Thread-safe; create one per process/GPU. */
typedef struct peaq_ctx peaq_ctx;
int peaq_ctx_create (int device_ordinal, peaq_ctx **out);
void peaq_ctx_destroy (peaq_ctx *ctx);
int peaq_ctx_device (const peaq_ctx *ctx);
/* ---- the reference's readings of BS.1387 as run-time switches ---------------
 * The reference is compiled with ONE value for each of six interpretation
 * switches (src/settings.h:47-97); a context carries them as data.  The
 * defaults (peaq_settings_default, or passing NULL) are the values the reference
 * ships with, and every golden of the default build pins exactly those.
 * A change applies to batch calls made, and to sessions and brokers CREATED,
 * afterwards.  Each field replaces the settings.h macro of the same name. */
typedef struct peaq_settings {
  int swap_mod_patts_for_noise_loudness_movs;   /* settings.h:47  default 1  (movs.c:566-575, 693-703) */
  int center_ehs_correlation_window;            /* settings.h:56  default 0  (movs.c:1362-1368) */
  int ehs_subtract_dc_before_window;            /* settings.h:66  default 1  (movs.c:1409-1433) */
  int use_floor_for_steps_above_threshold;      /* settings.h:76  default 0  (movs.c:1256-1260) */
  int clamp_movs;                               /* settings.h:86  default 0  (nn.c:202-207, 320-325) */
  int swap_slope_filter_coefficients;           /* settings.h:97  default 0  (fbearmodel.c:335-339) */
} peaq_settings;
void peaq_settings_default (peaq_settings *s);
int peaq_ctx_set_settings (peaq_ctx *ctx, const peaq_settings *s);
int peaq_ctx_get_settings (const peaq_ctx *ctx, peaq_settings *s);

/* Advanced version only: the arithmetic of the filter-bank ear model's front half
 * (fbearmodel.c:327-435: the 40 complex FIR filters, the level-dependent slopes, the upward spreading).
 *   PEAQ_FIR_F64 (default): the reference's double arithmetic -- v_mfma_f64_16x16x4_f64 and FP64 vector
 *       instructions throughout; follows the oracle to 1e-9 per block (tests/gpu_common.py, column "default").
 *       Bands 0..23 are evaluated in the block-sum form of DESIGN.md section 10 (a third of the reference's
 *       multiply-adds), bands 24..39 as plain sums.
 *   PEAQ_FIR_F16X3 (opt-in, 1.5x the throughput of the default on MI355X): the FIR bank on
 *       v_mfma_f32_16x16x32_f16 with signal and coefficients split into a high and a low FP16 part and three
 *       products per term (about 22 bits), FP32 accumulation; slopes and upward spreading in FP32, the slope
 *       filter (the one recurrence along the stream) FP64.  Tolerances this mode is held to by the parity suite
 *       (tests/gpu_common.py, column "f16x3"): MOVs 2e-6 relative, DI and ODG 1e-6 against the real reference's
 *       goldens; per-block excitation 1e-4; a stream cut into launches in different ways (session, broker,
 *       batch) agrees with itself to 1e-9.  Measured max |dODG| against the FP64 engine: 5e-7 over 39 advanced
 *       cases (profiles/r02_precision_ledger.json), 6e-6 over 4096 ten-second pairs (bench.py,
 *       advanced.reduced_precision_f16x3).  No range limit: a signal whose filtered peak leaves the FP16
 *       headroom (30 dB above full scale) runs at its own power-of-two scale.
 *   PEAQ_FIR_F32 (opt-in): the FIR bank on v_mfma_f32_16x16x4_f32 (5e-8), everything after it FP64.
 * peaq_ctx_set_fir_fp64(ctx, 0) selects PEAQ_FIR_F16X3, (ctx, 1) PEAQ_FIR_F64; the environment variables
 * PEAQ_AMD_FIR=f64|f16x3|f32 and PEAQ_AMD_FIR_FP64=1|0 set the mode at context creation.  peaq_version() names
 * the default.  Until release 0.2.0 the default was PEAQ_FIR_F16X3.  The basic version and everything
 * downstream of the spreading are FP64 in every mode.  Applies to the launches that follow.
 * Non-finite input samples (NaN / Inf in a float stream): the reference's DC-rejection filters are recursive
 * (fbearmodel.c:289-303), so one such sample makes every later filter-bank output of that signal NaN there, and it
 * does here; what differs is only how far BACK it reaches inside the launch that contains it -- PEAQ_FIR_F64
 * evaluates the long filters through running sums that are re-anchored once per launch (up to 840 blocks), so the
 * blocks of that launch in front of the sample may read NaN as well.  Finite input is assumed, as by the reference. */
#define PEAQ_FIR_F32   0
#define PEAQ_FIR_F64   1
#define PEAQ_FIR_F16X3 2
int peaq_ctx_set_fir_mode (peaq_ctx *ctx, int mode);
int peaq_ctx_get_fir_mode (const peaq_ctx *ctx);
int peaq_ctx_set_fir_fp64 (peaq_ctx *ctx, int enable);
int peaq_ctx_get_fir_fp64 (const peaq_ctx *ctx);

/* ---- session API ------------------------------------------------------------
 * Replaces, per element instance: g_object_new(PEAQ_TYPE_FFTEARMODEL /
 * _FILTERBANKEARMODEL) gstpeaq.c:364-365, peaq_movaccum_new x11 :376,
 * "number-of-bands"/"playback-level" :481-526, peaq_movaccum_set_mode :528-557,
 * _set_channels :580-584, alloc_per_channel_data :442-473.
 * `channels` is what set_caps learns (1 or 2); changing caps or the `advanced`
 * property means destroying and re-creating the session, as the reference
 * re-allocates all per-channel state there (gstpeaq.c:519,559,575,586). */
typedef struct peaq_session peaq_session;
int peaq_session_create (peaq_ctx *ctx, int advanced, int channels,
                         double playback_level_db, peaq_session **out);
void peaq_session_destroy (peaq_session *s);

/* pad_chain (gstpeaq.c:614-661): append interleaved F32 samples
 * (n = samples per channel) to the ref (pad 0) or test (pad 1) adapter and
 * process every frame that both adapters now hold (do_processing :596-611:
 * FFT frames 2048/1024; advanced additionally filter-bank blocks 192/192).
 * The data is copied before the call returns (SURVEY.md 8(b) ownership).
 * Processing is asynchronous on the session's HIP stream. */
int peaq_session_push (peaq_session *s, int pad, const float *interleaved, size_t n);

/* change_state PAUSED->READY (gstpeaq.c:764-776): one zero-padded frame from
 * whatever is left in the adapters (do_flush :716-745); leftovers may differ
 * between ref and test. */
int peaq_session_flush (peaq_session *s);

/* get_property "di"/"odg" (gstpeaq.c:484-492 -> calculate_di_* :1013-1063,
 * calculate_odg :1066-1078): callable at any time, idempotent.  movs receives
 * 11 or 5 values (may be NULL).  NaN-on-empty-accumulator behaviour of the
 * reference is preserved. */
int peaq_session_results (peaq_session *s, peaq_result *out);

/* set_property "playback_level" (gstpeaq.c:508-515 -> fftearmodel.c:305-314,
 * fbearmodel.c:249-254): applies to every frame processed from now on, state is
 * kept -- exactly what the reference's ear models do when the property changes
 * mid-stream. */
int peaq_session_set_level (peaq_session *s, double playback_level_db);

/* explicit reset (the reference never resets, gstpeaq.c:357-361; the element
 * does not call this) */
int peaq_session_reset (peaq_session *s);

/* ---- broker: many live sessions, one launch per tick -------------------------
 * Replaces, for a process hosting MANY `peaq` elements (BASELINE.json
 * configs[5]), the per-element device work of pad_chain -> do_processing
 * (gstpeaq.c:596-640): every element owns a session SLOT of one shared broker;
 * push only queues on the host, and each tick runs the frames that became
 * ready in all sessions as ONE batched front-end + back-end launch.  The
 * result of a session is identical to that of a peaq_session fed the same
 * stream (basic and advanced version).  All entry points are thread-safe. */
typedef struct peaq_broker peaq_broker;
typedef struct peaq_broker_stats_t {
  uint64_t ticks;        /* ticks executed (including idle ones)             */
  uint64_t launches;     /* ticks that launched device work                  */
  uint64_t frames;       /* FFT frames (per session, not per channel) run    */
  uint32_t max_active;   /* most sessions served by a single launch          */
  uint32_t worker_failed;/* the tick thread stopped on a device error        */
  /* Timing, in microseconds, over the ticks that launched device work (max, and the 99th percentile from a
   * histogram with 9 % resolution).  A session's LATENCY is the time from "a whole frame (block) of both pads
   * sits in its FIFO" (or its flush was requested) to "the device work of the tick that took it is complete",
   * i.e. wait for the next tick + that tick's host part + its device part: what the reference does inside
   * pad_chain (gstpeaq.c:614-661) the broker does at most one tick period later. */
  double   tick_host_us_max, tick_host_us_p99, tick_host_us_mean;        /* scan + staging copies + enqueue           */
  double   tick_device_us_max, tick_device_us_p99, tick_device_us_mean;  /* copies and kernels of one tick on the GPU */
  double   latency_us_max, latency_us_p99, latency_us_mean;
  uint64_t latency_samples;                         /* (session, tick) pairs behind the latency figures */
} peaq_broker_stats_t;
int  peaq_broker_create  (peaq_ctx *ctx, int advanced, int channels, double playback_level_db,
                          int max_sessions, peaq_broker **out);
/* The same over SEVERAL GPUs of a node (BASELINE.json configs[4] on an 8-GPU box): one device broker -- its own
 * context, slots, launch stream and tick thread -- per entry of `devices` (an ordinal may appear more than once:
 * two brokers on that GPU); a session is opened on the device that holds the fewest, stays there, and every call
 * below is forwarded by its id.  Sessions never exchange anything (gstpeaq.c:110-139: all state is per element), so
 * the devices' ticks are independent.  `settings` NULL = the reference's shipped values, `fir_mode` < 0 = the
 * engine's default (PEAQ_FIR_*).  max_sessions is rounded up to a multiple of the device count.  peaq_broker_stats
 * adds the counts up over the devices and reports the worst device's times. */
int  peaq_broker_create_multi (const int *devices, int n_devices, int advanced, int channels,
                               double playback_level_db, int max_sessions, const peaq_settings *settings,
                               int fir_mode, peaq_broker **out);
int  peaq_broker_devices (const peaq_broker *b);                       /* 1 for peaq_broker_create's */
/* Test hook: puts one device's share of a multi-device broker (shard 0 .. devices - 1 in the order of `devices`; 0 of a
 * plain broker) into the state a device error during one of its ticks leaves it in -- stopped for good, `message` kept
 * as its error.  What must hold then (tests/test_gpu_broker.py): peaq_broker_tick still ticks every other device and
 * returns this device's error; sessions on the other devices run to their results; every call on a session of the
 * failed device reports that device's own message. */
int  peaq_debug_broker_fail_shard (peaq_broker *b, int shard, const char *message);
void peaq_broker_destroy (peaq_broker *b);
int  peaq_broker_open    (peaq_broker *b, int *session_id);            /* gst_peaq_init / READY->PAUSED      */
int  peaq_broker_close   (peaq_broker *b, int session_id);             /* finalize                           */
int  peaq_broker_push    (peaq_broker *b, int session_id, int pad, const float *interleaved, size_t n_samples);
int  peaq_broker_flush   (peaq_broker *b, int session_id);             /* EOS: do_flush on the next tick     */
int  peaq_broker_tick    (peaq_broker *b, unsigned *n_active);         /* one batched launch, any thread     */
int  peaq_broker_results (peaq_broker *b, int session_id, peaq_result *out);  /* drains this session first   */
int  peaq_broker_start   (peaq_broker *b, unsigned period_us);         /* own tick thread (0 = 2000 us)      */
int  peaq_broker_stop    (peaq_broker *b);
int  peaq_broker_stats   (peaq_broker *b, peaq_broker_stats_t *out);
size_t peaq_broker_stats_size (void);   /* sizeof (peaq_broker_stats_t) in THIS library: a caller built against another
                                         * header compares before it hands over a buffer */

/* ---- batch API ------------------------------------------------------------
 * n_pairs whole pairs, inputs already in device memory as interleaved F32
 * [pair][sample][channel] with a fixed stride of `pair_stride` samples between
 * pairs.  n_ref / n_test give each pair's length in samples per channel (host
 * arrays of n_pairs entries; NULL = every pair has n_uniform samples).  Each
 * pair is framed and flushed exactly as one element run would be (full frames
 * + one zero-padded frame).  d_results: device array of n_pairs peaq_result.
 * `stream` is a hipStream_t (NULL = default stream); the call enqueues work
 * and returns, results are valid after the stream is synchronised. */
int peaq_batch_run (peaq_ctx *ctx, int advanced, int channels, double playback_level_db,
                    int n_pairs, const float *d_ref, const float *d_test, size_t pair_stride,
                    const uint32_t *n_ref, const uint32_t *n_test, uint32_t n_uniform,
                    peaq_result *d_results, void *stream);

/* Scratch the batch path needs for a given shape, in bytes (it is allocated
 * lazily inside the context and reused across calls). */
size_t peaq_batch_workspace_bytes (int advanced, int channels, int n_pairs, uint32_t n_max);

/* Timing of the last peaq_batch_run on this context, measured with HIP events
 * on `stream`: total milliseconds, and milliseconds / launch count of the
 * dominant kernel (the FFT ear-model front end).  Valid after the stream has
 * been synchronised.  Used by bench.py for the roofline line. */
typedef struct {
  float total_ms;
  float frontend_ms;      /* sum over launches of the front-end kernel */
  int   frontend_launches;
  float backend_ms;
  int   backend_launches;
  float fb_ms;            /* advanced: filter-bank kernels */
  int   fb_launches;
} peaq_batch_timing;
int peaq_batch_last_timing (peaq_ctx *ctx, peaq_batch_timing *out);
/* The shader clock (MHz) the device held while the last batch ran: one workgroup of every back-end launch reads the
 * shader-clock counter and the constant-rate counter around its own lifetime -- beside the front end of the next chunk,
 * i.e. under the step's own load; 0 before the first batch.  With peaq_calibrate() what lets a throughput number be
 * read as "this library at this clock". */
int peaq_batch_last_clock (peaq_ctx *ctx, double *shader_clock_mhz);

/* One whole pair from HOST memory (interleaved F32, n samples per channel each): upload + the batch path
 * with one pair + result.  For callers that hold both signals completely (gstpeaq_amd/cli/peaq.c); same
 * framing, flush and results as a session fed with the same samples (gstpeaq.c:596-611, 716-745). */
int peaq_run_pair (peaq_ctx *ctx, int advanced, int channels, double playback_level_db,
                   const float *ref, size_t n_ref, const float *test, size_t n_test, peaq_result *out);

/* ---- device calibration (measurement support, bench.py) -----------------------
 * Runs a fixed FP64 multiply-add kernel (ONE wave per SIMD, sixteen independent chains; `iterations` x 512
 * multiply-adds per wave, <= 0: about 70 ms) on the context's device -- alone: it waits for everything this PROCESS has
 * in flight on the device first (hipDeviceSynchronize) and runs on a stream of its own; work of other processes on the
 * same device is the caller's to exclude -- and reports the shader clock the device held under that load and the FP64
 * rate it gave.  MI355X clocks to its power budget, and this path's kernels are FP64-dense: the same library differs
 * by several per cent from box to box.  bench.py calls this before and after its timed region so that a throughput
 * line carries the clock it was measured at.  The kernel starts from the idle device's clock: its FIRST half is
 * reported as the ramp (ramp_*), its SECOND half as the steady state (the unprefixed fields). */
typedef struct {
  double elapsed_ms;          /* HIP events around the whole kernel (both halves) */
  double shader_clock_mhz;    /* steady state: s_memtime ticks / constant-rate wall-clock ticks, mean over all waves */
  double fp64_tflops;         /* steady state: 2 x the half's multiply-adds / mean duration of a wave's second half */
  double cycles_per_fma;      /* steady state: shader cycles per v_fma_f64 of a wave that has its SIMD to itself (4 = the pipe) */
  double max_clock_mhz;       /* hipDeviceProp_t::clockRate */
  int    compute_units;
  double ramp_clock_mhz;      /* the same over the first half: from the idle clock upwards, waves still being dispatched */
  double ramp_cycles_per_fma;
  double event_fp64_tflops;   /* 2 x all multiply-adds / elapsed_ms: includes launch, ramp and tail */
  /* where the kernel's waves (one per SIMD of the device as the runtime reports it) actually ran: distinct SIMDs
   * (HW_ID / XCC_ID of every wave), the most waves any one of them got (2 = that SIMD ran them one after the other:
   * the kernel then lasts twice a wave's lifetime and event_fp64_tflops halves), and the time between the first and
   * the last wave's start */
  int    simds_used;
  int    max_waves_on_a_simd;
  double dispatch_spread_ms;
} peaq_calibration;
int peaq_calibrate (peaq_ctx *ctx, int iterations, peaq_calibration *out);

/* ---- synthetic workload (include/peaq_synth.h on the device) -------------
 * Fills d_ref/d_test [n_pairs][pair_stride][channels] with the seeded pairs
 * seed0 .. seed0+n_pairs-1, n_samples each.  Benchmark / test utility. */
int peaq_synth_fill (peaq_ctx *ctx, uint32_t seed0, int n_pairs, int channels,
                     uint32_t n_samples, size_t pair_stride,
                     float *d_ref, float *d_test, void *stream);

/* ---- stage-level access for parity tests ----------------------------------
 * Runs only the stateless front end (window, FFT, power spectra, band
 * grouping, spreading, per-frame MOV ingredients) on ONE pair resident in
 * device memory and copies the per-frame records to host memory:
 *   out[frame][channel][PEAQ_DEBUG_RECORD_DOUBLES]
 * see DESIGN.md "frame record" for the layout. */
#define PEAQ_DEBUG_RECORD_DOUBLES 576
int peaq_debug_frontend (peaq_ctx *ctx, int bands, int channels, double playback_level_db,
                         const float *d_ref, const float *d_test, uint32_t n_ref, uint32_t n_test,
                         int n_frames, double *host_out);

/* The same for the filter-bank ear model (advanced): per-block records
 *   out[block][channel][PEAQ_DEBUG_FB_RECORD_DOUBLES] =
 *   { unsmeared ref[40], unsmeared test[40], excitation ref[40], excitation test[40],
 *     above-threshold flag, pad[7] } */
#define PEAQ_DEBUG_FB_RECORD_DOUBLES 168
int peaq_debug_filterbank (peaq_ctx *ctx, int channels, double playback_level_db,
                           const float *d_ref, const float *d_test, uint32_t n_ref, uint32_t n_test,
                           int n_blocks, int blocks_per_launch, double *host_out);

/* The stateful back end of the basic version on its own: feeds n_frames front-end
 * records of ONE pair (host memory, [frame][channel][PEAQ_DEBUG_RECORD_DOUBLES],
 * e.g. from peaq_debug_frontend or hand-built) through time smearing, level and
 * pattern adaptation, modulation processing and the MOV layer starting from a
 * fresh state, and returns per (frame, channel) PEAQ_DEBUG_BACKEND_DOUBLES doubles:
 *   8 band vectors of 112 -- excitation ref/test (fftearmodel.c:496-504), spectrally
 *   adapted ref/test (leveladapter.c:243-340), modulation ref/test and average
 *   loudness ref/test (modpatt.c:223-251) -- then total loudness ref, test
 *   (earmodel.c:891-907; only written while the loudness gate is still closed), six unused, and then the MOV
 *   layer's values of this frame BEFORE accumulation, computed for every frame whether or not the gates of
 *   gstpeaq.c:871,880-881 let the accumulators see them: ModDiff1, ModDiff2, TempWt (movs.c:205-254), noise
 *   loudness (:354-371), mean and maximum of the band noise-to-mask ratios (:971-1023) and, with channel 0
 *   only, detection probability and steps above threshold of the frame (:1224-1276).
 * `result` (may be NULL) receives the MOVs/DI/ODG after the last frame.
 * Pins the HIP pattern layer against the reference's own known-answer vectors
 * (testpeaq.c:433-599,748-810). */
#define PEAQ_DEBUG_BACKEND_DOUBLES 912
int peaq_debug_backend (peaq_ctx *ctx, int channels, int n_frames, const double *host_records,
                        double *host_out, peaq_result *result);

/* Host only, no device: self-check of the constant tables of the FP64 filter-bank engine.  Its long filters run
 * in a "block-sum" form (three rectangular windows per Hann window, running sums over 32-sample blocks: DESIGN.md 3);
 * the function evaluates that form and the direct tile of the short filters FROM THE TABLES on a pseudo-random
 * window and returns the largest deviation from the plain sums of fbearmodel.c:399-435, relative to each band's
 * largest output (about 4e-15).  tests/test_capi_host.py runs it on the CPU. */
double peaq_debug_fb_tables_selfcheck (void);

/* The advanced version's MOV layer on its own (fresh state): the filter-bank back end over
 *   fb_records  [n_blocks][channels][PEAQ_DEBUG_FB_RECORD_DOUBLES]   (as peaq_debug_filterbank returns them)
 * and the 55-band FFT back end over
 *   fft_records [n_frames][channels][PEAQ_DEBUG_RECORD_DOUBLES]      (as peaq_debug_frontend returns them for 55 bands),
 * every block's / frame's MOV values BEFORE accumulation -- also those the gates of gstpeaq.c:988,996-997 keep from
 * the accumulators:
 *   out_blocks [n_blocks][channels][PEAQ_DEBUG_ADV_BLOCK_DOUBLES] = { RmsModDiff (movs.c:205-254, RMS normalisation
 *              :243-244), its weight, the noise loudness of RmsNoiseLoudAsym and its missing-components term
 *              (movs.c:551-577), AvgLinDist (movs.c:679-706), total loudness of ref and test while the loudness
 *              gate is closed (earmodel.c:891-907; 0 afterwards), pad }
 *   out_frames [n_frames][channels][2] = { SegmentalNMR's 10 log10 of the mean band NMR (movs.c:1010-1020), that mean }
 * and the pair's result after the last block and frame. */
#define PEAQ_DEBUG_ADV_BLOCK_DOUBLES 8
int peaq_debug_backend_advanced (peaq_ctx *ctx, int channels, int n_blocks, const double *fb_records,
                                 int n_frames, const double *fft_records, double *out_blocks,
                                 double *out_frames, peaq_result *result);

#ifdef __cplusplus
}
#endif
#endif /* PEAQ_AMD_H */
