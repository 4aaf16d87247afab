// peaq_capi.hip -- the C ABI of include/peaq_amd.h: device context, batch
// driver, streaming sessions.  Host code only; the kernels are in
// peaq_frontend.hip / peaq_backend.hip / peaq_fb.hip / peaq_synth.hip.
//
// Framing follows the reference element: FFT frames of 2048 samples every 1024
// (do_processing, gstpeaq.c:596-611), filter-bank blocks of 192 every 192, and
// at the end ONE zero-padded frame/block built from whatever is left on either
// side (do_flush, gstpeaq.c:716-745).  Frame f of a pair reads samples
// [1024 f, 1024 f + 2048) of each signal, zero beyond the signal's length.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/peaq_amd.h"
#include "peaq_device.h"
#include "peaq_kernels.h"
#include "peaq_tables.h"
#ifdef PEAQ_DEV_PROBES                               // VARIANT builds only (csrc/Makefile): never in the product library
#define PEAQ_DEV_TU_CAPI
#include "dev_probes.inc"
#endif
#ifndef PEAQ_DEV_SERIAL_KERNELS
#define PEAQ_DEV_SERIAL_KERNELS false
#endif
#ifndef PEAQ_DEV_SKIP_BACKEND
#define PEAQ_DEV_SKIP_BACKEND false
#endif
#ifndef PEAQ_DEV_BE_STREAM_PRIO
#define PEAQ_DEV_BE_STREAM_PRIO(prio, lo, hi)
#endif

using namespace peaq;

static_assert(sizeof(ResultRecord) == sizeof(peaq_result), "result layouts must match");
static_assert(kPubDoubles == PEAQ_DEBUG_RECORD_DOUBLES, "record layouts must match");
static_assert(kDbgDoubles == PEAQ_DEBUG_BACKEND_DOUBLES, "debug layouts must match");

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? PEAQ_ERR_NOMEM : PEAQ_ERR_DEVICE,                           \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                     \
  } while (0)

extern "C" const char* peaq_last_error(void) { return g_err.c_str(); }
extern "C" const char* peaq_version(void) { return "0.2.0 gfx950 (advanced version: FP64 filter bank by default; f16x3 and f32 opt-in)"; }
// host only, no device: the filter-bank tables of the FP64 engine against the reference's plain sums (peaq_tables.cpp)
extern "C" double peaq_debug_fb_tables_selfcheck(void) { return peaq::fb_tables_selfcheck(); }

// ---------------------------------------------------------------------------
// framing arithmetic
// ---------------------------------------------------------------------------
// number of frames the element processes for signals of n_ref / n_test samples:
// full frames while BOTH adapters hold `frame` samples, then one flush frame if
// anything is left on either side.
static uint32_t count_frames(uint64_t n_ref, uint64_t n_test, uint32_t frame, uint32_t hop) {
  const uint64_t n = std::min(n_ref, n_test);
  const uint64_t full = n >= frame ? (n - frame) / hop + 1 : 0;
  const bool left = n_ref > full * hop || n_test > full * hop;
  return static_cast<uint32_t>(full + (left ? 1 : 0));
}

extern "C" uint32_t peaq_frame_count(uint64_t n_ref, uint64_t n_test, int filter_bank) {
  return filter_bank ? count_frames(n_ref, n_test, kFbFrame, kFbFrame) : count_frames(n_ref, n_test, kFrame, kHop);
}

// ---------------------------------------------------------------------------
// growable device buffer
// ---------------------------------------------------------------------------
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) cap = bytes;
    // PEAQ_AMD_POISON=1 (tests/test_gpu_poison.py): every workspace starts as NaNs (all bits set) instead of whatever
    // the allocator hands out -- fresh memory is zero, recycled memory is not; nothing may depend on either.  State
    // that has to start from zero is set to zero explicitly where it is created.
    static const bool poison = [] { const char* v = std::getenv("PEAQ_AMD_POISON"); return v && *v && *v != '0'; }();
    if (e == hipSuccess && poison) {
      e = hipMemset(p, 0xFF, bytes);
      if (e == hipSuccess) e = hipDeviceSynchronize();   // (the owners' own streams do not wait for the null stream)
    }
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <typename T>
  T* as() const { return static_cast<T*>(p); }
};

// a DevBuf that frees itself on every way out of a function (the debug entry points)
struct TmpBuf : DevBuf {
  ~TmpBuf() { release(); }
};

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
struct TimedSpan {
  hipEvent_t a, b;
  int kind;   // 0 front end, 1 back end, 2 filter bank
};

struct peaq_ctx {
  int device = 0;
  CommonTables* d_common = nullptr;
  BandTables* d_bands109 = nullptr;
  BandTables* d_bands55 = nullptr;
  BandTables* d_bands40 = nullptr;
  FbTables* d_fb = nullptr;
  std::mutex mu;            // serialises batch calls / workspace use
  // batch workspace
  DevBuf records, records2, fb_records, fb_records2, state, fbstate, hp_scratch, hp_scratch2, counts;
  hipStream_t aux = nullptr;   // the back end runs here, overlapped with the next chunk's front end
  hipStream_t aux2 = nullptr;  // advanced: the filter-bank path runs here, beside the FFT path
  hipStream_t aux3 = nullptr, aux4 = nullptr;   // ... its high-pass stage and its back end (3-stage pipeline)
  hipEvent_t batch_begin = nullptr, batch_end = nullptr;
  bool batch_pending = false;
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
  unsigned long long* d_prof = nullptr;   // -DPEAQ_FE_PROFILE builds only
  int fir_fp64 = 1;                       // advanced version: arithmetic of the FIR bank (PEAQ_FIR_*; default the reference's FP64)
  Settings settings;                      // the reference's settings.h switches (peaq_ctx_set_settings)

  hipEvent_t next_event() {
    if (events_used == event_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      event_pool.push_back(e);
    }
    return event_pool[events_used++];
  }
};

extern "C" int peaq_ctx_create(int device, peaq_ctx** out) {
  if (!out) return fail(PEAQ_ERR_ARG, "peaq_ctx_create: out is NULL");
  *out = nullptr;
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev == 0)
    return fail(PEAQ_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
  if (device < 0 || device >= n_dev) return fail(PEAQ_ERR_ARG, "peaq_ctx_create: bad device ordinal");
  HIP_TRY(hipSetDevice(device));
  peaq_ctx* c = new (std::nothrow) peaq_ctx;
  if (!c) return fail(PEAQ_ERR_NOMEM, "out of host memory");
  c->device = device;
  {
    const char* e = std::getenv("PEAQ_AMD_FIR_FP64");    // "0": the reduced-precision engine, anything else: FP64 (the default)
    if (e && *e) c->fir_fp64 = *e != '0' ? 1 : 2;
    if (const char* m = std::getenv("PEAQ_AMD_FIR")) {     // "f16x3" | "f32" | "f64"
      const std::string mode(m);
      if (mode == "f64") c->fir_fp64 = 1;
      else if (mode == "f32") c->fir_fp64 = 0;
      else if (mode == "f16x3") c->fir_fp64 = 2;
      else {
        delete c;
        return fail(PEAQ_ERR_ARG, "PEAQ_AMD_FIR: expected f16x3, f32 or f64");
      }
    }
  }
  if (const char* e = std::getenv("PEAQ_AMD_SETTINGS")) {
    // "CLAMP_MOVS=1,center_ehs_correlation_window=1": the settings.h macro names, any case -- lets the CLI and
    // the element (which have no such property, like the reference's) run the other readings of BS.1387
    struct { const char* name; int* field; } tab[] = {
        {"swap_mod_patts_for_noise_loudness_movs", &c->settings.swap_mod_patts},
        {"center_ehs_correlation_window", &c->settings.centre_ehs_window},
        {"ehs_subtract_dc_before_window", &c->settings.ehs_dc_before_window},
        {"use_floor_for_steps_above_threshold", &c->settings.floor_steps},
        {"clamp_movs", &c->settings.clamp_movs},
        {"swap_slope_filter_coefficients", &c->settings.swap_slope}};
    std::string spec(e);
    size_t pos = 0;
    while (pos < spec.size()) {
      const size_t end = std::min(spec.find(',', pos), spec.size());
      std::string item = spec.substr(pos, end - pos);
      pos = end + 1;
      auto trim = [](std::string t) {
        const size_t a = t.find_first_not_of(" \t"), b = t.find_last_not_of(" \t");
        return a == std::string::npos ? std::string() : t.substr(a, b - a + 1);
      };
      item = trim(item);
      if (item.empty()) continue;
      const size_t eq = item.find('=');
      std::string key = trim(item.substr(0, eq));
      const std::string val = eq == std::string::npos ? std::string() : trim(item.substr(eq + 1));
      for (char& ch : key) ch = (char)std::tolower((unsigned char)ch);
      bool known = false;
      for (auto& t : tab)
        if (key == t.name) {
          // a bare NAME is refused rather than read as 0 (the opposite of what its author meant)
          if (val != "0" && val != "1") {
            delete c;
            return fail(PEAQ_ERR_ARG, "PEAQ_AMD_SETTINGS: '" + item + "': write NAME=0 or NAME=1");
          }
          *t.field = val == "1";
          known = true;
        }
      if (!known) {
        delete c;
        return fail(PEAQ_ERR_ARG, "PEAQ_AMD_SETTINGS: unknown switch '" + item + "' (settings.h macro names, NAME=0|1)");
      }
    }
  }
  const int rc = [&]() -> int {
  {
    std::vector<CommonTables> h(1);
    build_common_tables(h[0]);
    HIP_TRY(hipMalloc(&c->d_common, sizeof(CommonTables)));
    HIP_TRY(hipMemcpy(c->d_common, h.data(), sizeof(CommonTables), hipMemcpyHostToDevice));
  }
  {
    BandTables t;
    build_fft_band_tables(109, t);
    HIP_TRY(hipMalloc(&c->d_bands109, sizeof t));
    HIP_TRY(hipMemcpy(c->d_bands109, &t, sizeof t, hipMemcpyHostToDevice));
    build_fft_band_tables(55, t);
    HIP_TRY(hipMalloc(&c->d_bands55, sizeof t));
    HIP_TRY(hipMemcpy(c->d_bands55, &t, sizeof t, hipMemcpyHostToDevice));
    std::vector<FbTables> fb(1);
    build_fb_band_tables(t, fb[0]);
    HIP_TRY(hipMalloc(&c->d_bands40, sizeof t));
    HIP_TRY(hipMemcpy(c->d_bands40, &t, sizeof t, hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&c->d_fb, sizeof(FbTables)));
    HIP_TRY(hipMemcpy(c->d_fb, fb.data(), sizeof(FbTables), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipEventCreate(&c->batch_begin));
  HIP_TRY(hipEventCreate(&c->batch_end));
  {
    // the back end is the latency-bound consumer of the pipeline: give its stream priority
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    int prio = hi;
    PEAQ_DEV_BE_STREAM_PRIO(prio, lo, hi)
    HIP_TRY(hipStreamCreateWithPriority(&c->aux, hipStreamNonBlocking, prio));
  }
  HIP_TRY(hipStreamCreateWithFlags(&c->aux2, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&c->aux3, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&c->aux4, hipStreamNonBlocking));
#ifdef PEAQ_FE_PROFILE
  HIP_TRY(hipMalloc(&c->d_prof, 64 * sizeof(unsigned long long)));
  HIP_TRY(hipMemset(c->d_prof, 0, 64 * sizeof(unsigned long long)));
#endif
    return PEAQ_OK;
  }();
  if (rc != PEAQ_OK) {           // nothing allocated so far is leaked (destroy copes with a half-built context)
    const std::string msg = g_err;
    peaq_ctx_destroy(c);
    return fail(rc, msg);
  }
  *out = c;
  return PEAQ_OK;
}

extern "C" void peaq_ctx_destroy(peaq_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  (void)hipFree(c->d_common);
  (void)hipFree(c->d_bands109);
  (void)hipFree(c->d_bands55);
  (void)hipFree(c->d_bands40);
  (void)hipFree(c->d_fb);
  (void)hipFree(c->d_prof);
  c->records.release();
  c->records2.release();
  if (c->aux) (void)hipStreamDestroy(c->aux);
  if (c->aux2) (void)hipStreamDestroy(c->aux2);
  if (c->aux3) (void)hipStreamDestroy(c->aux3);
  if (c->aux4) (void)hipStreamDestroy(c->aux4);
  c->fb_records2.release();
  c->hp_scratch2.release();
  c->fb_records.release();
  c->state.release();
  c->fbstate.release();
  c->hp_scratch.release();
  c->counts.release();
  for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
  if (c->batch_begin) (void)hipEventDestroy(c->batch_begin);
  if (c->batch_end) (void)hipEventDestroy(c->batch_end);
  delete c;
}

extern "C" int peaq_ctx_device(const peaq_ctx* c) { return c ? c->device : -1; }

extern "C" int peaq_ctx_set_fir_fp64(peaq_ctx* c, int enable) {
  if (!c) return fail(PEAQ_ERR_ARG, "peaq_ctx_set_fir_fp64: ctx is NULL");
  std::lock_guard<std::mutex> lock(c->mu);
  c->fir_fp64 = enable ? 1 : 2;                      // off = the reduced-precision engine (PEAQ_FIR_F16X3)
  return PEAQ_OK;
}
extern "C" int peaq_ctx_get_fir_fp64(const peaq_ctx* c) { return c ? c->fir_fp64 == 1 : -1; }

extern "C" int peaq_ctx_set_fir_mode(peaq_ctx* c, int mode) {
  if (!c) return fail(PEAQ_ERR_ARG, "peaq_ctx_set_fir_mode: ctx is NULL");
  if (mode < 0 || mode > 2) return fail(PEAQ_ERR_ARG, "peaq_ctx_set_fir_mode: mode must be PEAQ_FIR_F32, _F64 or _F16X3");
  std::lock_guard<std::mutex> lock(c->mu);
  c->fir_fp64 = mode;
  return PEAQ_OK;
}
extern "C" int peaq_ctx_get_fir_mode(const peaq_ctx* c) { return c ? c->fir_fp64 : -1; }

extern "C" void peaq_settings_default(peaq_settings* s) {
  if (!s) return;
  const Settings d;
  s->swap_mod_patts_for_noise_loudness_movs = d.swap_mod_patts;
  s->center_ehs_correlation_window = d.centre_ehs_window;
  s->ehs_subtract_dc_before_window = d.ehs_dc_before_window;
  s->use_floor_for_steps_above_threshold = d.floor_steps;
  s->clamp_movs = d.clamp_movs;
  s->swap_slope_filter_coefficients = d.swap_slope;
}

extern "C" int peaq_ctx_set_settings(peaq_ctx* c, const peaq_settings* s) {
  if (!c) return fail(PEAQ_ERR_ARG, "peaq_ctx_set_settings: ctx is NULL");
  peaq_settings d;
  peaq_settings_default(&d);
  if (!s) s = &d;                                    // NULL: back to the reference's shipped values
  std::lock_guard<std::mutex> lock(c->mu);
  c->settings.swap_mod_patts = s->swap_mod_patts_for_noise_loudness_movs != 0;
  c->settings.centre_ehs_window = s->center_ehs_correlation_window != 0;
  c->settings.ehs_dc_before_window = s->ehs_subtract_dc_before_window != 0;
  c->settings.floor_steps = s->use_floor_for_steps_above_threshold != 0;
  c->settings.clamp_movs = s->clamp_movs != 0;
  c->settings.swap_slope = s->swap_slope_filter_coefficients != 0;
  return PEAQ_OK;
}

extern "C" int peaq_ctx_get_settings(const peaq_ctx* c, peaq_settings* s) {
  if (!c || !s) return fail(PEAQ_ERR_ARG, "peaq_ctx_get_settings: NULL argument");
  s->swap_mod_patts_for_noise_loudness_movs = c->settings.swap_mod_patts;
  s->center_ehs_correlation_window = c->settings.centre_ehs_window;
  s->ehs_subtract_dc_before_window = c->settings.ehs_dc_before_window;
  s->use_floor_for_steps_above_threshold = c->settings.floor_steps;
  s->clamp_movs = c->settings.clamp_movs;
  s->swap_slope_filter_coefficients = c->settings.swap_slope;
  return PEAQ_OK;
}

#ifdef PEAQ_FE_PROFILE
// development builds only: reads and clears the front end's phase counters (tools/fe_profile.py)
extern "C" int peaq_debug_frontend_profile(peaq_ctx* c, unsigned long long* out64) {
  if (!c || !out64) return fail(PEAQ_ERR_ARG, "peaq_debug_frontend_profile: NULL argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out64, c->d_prof, 64 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(c->d_prof, 0, 64 * sizeof(unsigned long long)));
  return PEAQ_OK;
}
#endif

// ---------------------------------------------------------------------------
// batch
// ---------------------------------------------------------------------------
static const size_t kRecordBudget = (size_t)1536 << 20;   // HBM for per-frame records of one chunk
// Filter-bank blocks per launch (a multiple of the tile of 10) and the HBM one buffer of high-passed rows
// may take.  Long launches pay: per block the bank kernel costs 0.150 ms in launches of 320 blocks, 0.142 at
// 840 (fewer drained-CU tails, fewer pipeline hand-overs) -- 4096 stereo pairs x 10 s: 4.57 -> 4.74 M
// frame-pairs/s for 2 x 21 GB of rows + 18 GB of block records, small change on a 288 GB device.  Round 4, FP64
// engine, same job: 420 / 630 / 840 / 1250 blocks per launch = 5.14 / 5.19 / 5.20 / 5.26 M -- but 1250 means
// 2 x 32 GB of rows + 27 GB of records per context that has run such a batch, and a process with three contexts
// (the parity suite has) no longer leaves room for another process on the device: 840 stays.
#ifndef PEAQ_FB_CHUNK
#define PEAQ_FB_CHUNK 840
#endif
#ifndef PEAQ_FB_ROWGB
#define PEAQ_FB_ROWGB 24
#endif
static const unsigned kFbBlocksPerChunk = PEAQ_FB_CHUNK;
static const size_t kFbRowBudget = (size_t)PEAQ_FB_ROWGB << 30;

// blocks per launch of the filter-bank path: kFbBlocksPerChunk unless the batch is so large that the
// rows of that many blocks would not fit the budget; always a multiple of the tile (10 blocks)
static unsigned fb_blocks_per_chunk(int n_pairs, int channels, uint32_t max_blocks) {
  const size_t n_signals = (size_t)n_pairs * channels * 2;
  const size_t per_signal = kFbRowBudget / std::max<size_t>(n_signals, 1) / sizeof(double);
  size_t bc = per_signal > (size_t)kFbRing ? (per_signal - kFbRing) / kFbFrame : 0;
  bc = std::min<size_t>(bc, kFbBlocksPerChunk) / 10 * 10;
  bc = std::max<size_t>(bc, 10);
  return static_cast<unsigned>(std::min<size_t>(bc, (max_blocks + 9) / 10 * 10));
}

static unsigned frames_per_chunk(int n_pairs, int channels, uint32_t max_frames) {
  const size_t per_frame = (size_t)n_pairs * channels * kRecDoubles * sizeof(double);
  size_t fc = kRecordBudget / std::max<size_t>(per_frame, 1);
  fc = std::max<size_t>(fc, 4);
  fc = std::min<size_t>(fc, 64);
  // few pairs: take long chunks so that the launch count stays small
  if ((size_t)max_frames * per_frame <= ((size_t)256 << 20)) fc = max_frames;
  // the front end takes a work item apart with a 32-bit reciprocal of the frames per launch, exact up to
  // max_frames_per_launch (launch_frontend refuses more): a 30-minute mono file is two chunks, not one
  fc = std::min<size_t>(fc, max_frames_per_launch((unsigned)n_pairs));
  return static_cast<unsigned>(std::min<size_t>(fc, std::max<uint32_t>(max_frames, 1)));
}

extern "C" size_t peaq_batch_workspace_bytes(int advanced, int channels, int n_pairs, uint32_t n_max) {
  const uint32_t frames = count_frames(n_max, n_max, kFrame, kHop);
  const unsigned fc = frames_per_chunk(n_pairs, channels, frames);
  size_t b = 2 * (size_t)n_pairs * fc * channels * kRecDoubles * sizeof(double) + (size_t)n_pairs * sizeof(PairState) +
             (size_t)n_pairs * 4 * sizeof(uint32_t);
  if (advanced) {
    const unsigned bc = fb_blocks_per_chunk(n_pairs, channels, count_frames(n_max, n_max, kFbFrame, kFbFrame));
    const size_t nbuf = count_frames(n_max, n_max, kFbFrame, kFbFrame) > bc ? 2 : 1;   // pipelined: double buffers
    b += nbuf * (size_t)n_pairs * bc * channels * kFbRecDoubles * sizeof(double);
    b += (size_t)n_pairs * channels * 2 *
         (sizeof(FbSignalState) + nbuf * ((size_t)bc * kFbFrame + kFbRing) * sizeof(double));
  }
  return b;
}

// split-FP16 FIR (FbFrontArgs.fir_fp64 == 2): the power of two that puts the filtered signal's full scale --
// |x| = 1 times the playback-level factor -- between 2^10 and 2^11 of FP16's 65504 (30 dB of headroom for
// samples beyond full scale and for the high-pass filter's overshoot)
static void set_fir_scale(FbFrontArgs& ff) {
  const int e = 10 - std::ilogb(ff.level_factor);
  ff.hf_xscale = std::ldexp(1., e);
  ff.hf_xunscale = std::ldexp(1., -e);
}

static int run_filterbank_path(peaq_ctx* c, int channels, double level_db, int n_pairs, const float* d_ref,
                               const float* d_test, size_t pair_stride, const uint32_t* d_nref,
                               const uint32_t* d_ntest, uint32_t n_uniform, const uint32_t* d_nblocks,
                               uint32_t max_blocks, hipStream_t stream) {
    // ---- filter-bank path: blocks of 192 samples (gstpeaq.c:648-652) ------------------
    const unsigned n_signals = (unsigned)n_pairs * channels * 2;
    const unsigned bc = fb_blocks_per_chunk(n_pairs, channels, max_blocks);
    const size_t row_stride = (size_t)kFbRing + (size_t)bc * kFbFrame;
    const bool piped = !PEAQ_DEV_SERIAL_KERNELS && max_blocks > bc;   // more than one chunk: 3-stage pipeline, double buffers
    HIP_TRY(c->hp_scratch.reserve((size_t)n_signals * row_stride * sizeof(double)));
    HIP_TRY(c->fb_records.reserve((size_t)n_pairs * bc * channels * kFbRecDoubles * sizeof(double)));
    if (piped) {
      HIP_TRY(c->hp_scratch2.reserve((size_t)n_signals * row_stride * sizeof(double)));
      HIP_TRY(c->fb_records2.reserve((size_t)n_pairs * bc * channels * kFbRecDoubles * sizeof(double)));
    }
    HIP_TRY(c->fbstate.reserve((size_t)n_signals * sizeof(FbSignalState)));
    HIP_TRY(hipMemsetAsync(c->fbstate.p, 0, (size_t)n_signals * sizeof(FbSignalState), stream));
    FbFrontArgs ff{};
    ff.cfg = c->settings;
    ff.fir_fp64 = c->fir_fp64;
    ff.ref = d_ref;
    ff.test = d_test;
    ff.pair_stride = pair_stride;
    ff.n_ref = d_nref;
    ff.n_test = d_ntest;
    ff.n_uniform_ref = ff.n_uniform_test = n_uniform;
    ff.n_blocks = d_nblocks;
    ff.n_blocks_uniform = max_blocks;
    ff.block_origin = 0;
    ff.channels = channels;
    ff.level_factor = fb_level_factor(level_db);
    set_fir_scale(ff);
    ff.bands = c->d_bands40;
    ff.fb = c->d_fb;
    ff.fbstate = c->fbstate.as<FbSignalState>();
    ff.hp_row_stride = row_stride;
    FbBackendArgs fbk{};
    fbk.cfg = c->settings;
    fbk.n_blocks = d_nblocks;
    fbk.n_blocks_uniform = max_blocks;
    fbk.channels = channels;
    fbk.bands = c->d_bands40;
  fbk.common = c->d_common;
    fbk.state = c->state.as<PairState>();
    // Three stages per chunk of blocks, each on its own stream: the high-pass filter (a few hundred
    // waves, latency bound), the filter bank (the bulk), the back end (one workgroup per pair).
    // Stage s of chunk i runs beside stage s+1 of chunk i-1; rows and records are double buffered.
    hipStream_t s_hp = piped ? c->aux3 : stream, s_bank = stream, s_be = piped ? c->aux4 : stream;
    if (piped) {
      hipEvent_t ready = c->next_event();
      if (!ready) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
      HIP_TRY(hipEventRecord(ready, stream));          // state initialised, filter state cleared
      HIP_TRY(hipStreamWaitEvent(s_hp, ready, 0));
      HIP_TRY(hipStreamWaitEvent(s_be, ready, 0));
    }
    double* rows[2] = {c->hp_scratch.as<double>(), piped ? c->hp_scratch2.as<double>() : c->hp_scratch.as<double>()};
    double* recs[2] = {c->fb_records.as<double>(), piped ? c->fb_records2.as<double>() : c->fb_records.as<double>()};
    hipEvent_t bank_done[2] = {nullptr, nullptr}, be_done[2] = {nullptr, nullptr};
    unsigned prev = 0, chunk = 0;
    for (uint32_t b0 = 0; b0 < max_blocks; b0 += bc, ++chunk) {
      const unsigned nb = std::min<uint32_t>(bc, max_blocks - b0);
      const int b = chunk & 1;
      ff.block0 = b0;
      ff.blocks_per_launch = nb;
      ff.launch_idx = chunk;                             // (sessions, broker, stage entry points: one stream, always slot 0)
      ff.prev_blocks = prev;
      ff.first_launch = b0 == 0;
      ff.hp_scratch = rows[b];
      ff.hp_prev = chunk ? rows[b ^ 1] : nullptr;
      ff.records = recs[b];
      fbk.records = recs[b];
      fbk.block0 = b0;
      fbk.blocks_per_launch = nb;
      hipEvent_t e0 = c->next_event(), e1 = c->next_event(), e_hp = c->next_event(), e_be = c->next_event();
      if (!e0 || !e1 || !e_hp || !e_be) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
      if (piped) {
        if (bank_done[b]) HIP_TRY(hipStreamWaitEvent(s_hp, bank_done[b], 0));   // rows[b] no longer read
        if (be_done[b]) HIP_TRY(hipStreamWaitEvent(s_hp, be_done[b], 0));       // recs[b] no longer read
      }
      HIP_TRY(launch_fb_hp(ff, n_pairs, s_hp));
      if (piped) {
        HIP_TRY(hipEventRecord(e_hp, s_hp));
        HIP_TRY(hipStreamWaitEvent(s_bank, e_hp, 0));
      }
      HIP_TRY(hipEventRecord(e0, s_bank));
      HIP_TRY(launch_fb_bank(ff, n_pairs, s_bank));
      HIP_TRY(hipEventRecord(e1, s_bank));
      bank_done[b] = e1;
      if (piped) HIP_TRY(hipStreamWaitEvent(s_be, e1, 0));
      HIP_TRY(launch_fb_backend(fbk, n_pairs, s_be));
      if (piped) {
        HIP_TRY(hipEventRecord(e_be, s_be));
        be_done[b] = e_be;
      }
      c->spans.push_back({e0, e1, 2});
      prev = nb;
    }
    if (piped)
      for (int i = 0; i < 2; ++i)
        if (be_done[i]) HIP_TRY(hipStreamWaitEvent(stream, be_done[i], 0));    // join: `stream` ends the path
  return PEAQ_OK;
}

static int batch_run_locked(peaq_ctx* c, int advanced, int channels, double level_db, int n_pairs, const float* d_ref,
                            const float* d_test, size_t pair_stride, const uint32_t* n_ref, const uint32_t* n_test,
                            uint32_t n_uniform, peaq_result* d_results, hipStream_t stream);

extern "C" int peaq_batch_run(peaq_ctx* c, int advanced, int channels, double level_db, int n_pairs,
                              const float* d_ref, const float* d_test, size_t pair_stride, const uint32_t* n_ref,
                              const uint32_t* n_test, uint32_t n_uniform, peaq_result* d_results, void* stream_) {
  if (!c) return fail(PEAQ_ERR_ARG, "peaq_batch_run: ctx is NULL");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_batch_run: channels must be 1 or 2");
  if (n_pairs < 0) return fail(PEAQ_ERR_ARG, "peaq_batch_run: n_pairs < 0");
  if (n_pairs == 0) return PEAQ_OK;
  if (!d_ref || !d_test || !d_results) return fail(PEAQ_ERR_ARG, "peaq_batch_run: NULL buffer");
  if ((n_ref == nullptr) != (n_test == nullptr))
    return fail(PEAQ_ERR_ARG, "peaq_batch_run: give both n_ref and n_test or neither");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  if (c->batch_pending) {            // the workspace is still owned by the previous call
    HIP_TRY(hipEventSynchronize(c->batch_end));
    c->batch_pending = false;
  }
  c->spans.clear();
  c->events_used = 0;
  const int rc = batch_run_locked(c, advanced, channels, level_db, n_pairs, d_ref, d_test, pair_stride, n_ref, n_test,
                                  n_uniform, d_results, stream);
  if (rc != PEAQ_OK) {
    // part of the pipeline may already run on the context's own streams: nothing may touch the
    // workspace (or free it) before that work has drained
    const std::string msg = g_err;
    (void)hipDeviceSynchronize();
    c->batch_pending = false;
    return fail(rc, msg);
  }
  return PEAQ_OK;
}

// One whole (ref, test) pair from host memory: the batch path with n_pairs = 1 -- for a caller that holds both
// files (the CLI).  The same frames and blocks as a session fed with the same samples (count_frames), but every
// kernel sees the whole stream: one front-end launch, the filter-bank path pipelined over its three streams.
extern "C" int peaq_run_pair(peaq_ctx* c, int advanced, int channels, double level_db, const float* ref, size_t n_ref,
                             const float* test, size_t n_test, peaq_result* out) {
  if (!c || !out) return fail(PEAQ_ERR_ARG, "peaq_run_pair: NULL argument");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_run_pair: channels must be 1 or 2");
  if ((n_ref && !ref) || (n_test && !test)) return fail(PEAQ_ERR_ARG, "peaq_run_pair: NULL samples");
  if (n_ref > 0xFFFFFFFFu || n_test > 0xFFFFFFFFu) return fail(PEAQ_ERR_ARG, "peaq_run_pair: more than 2^32 samples");
  HIP_TRY(hipSetDevice(c->device));
  size_t stride = std::max<size_t>(std::max(n_ref, n_test), 2);
  stride += stride & 1;                              // 8-byte rows: the frame loads are dword pairs
  TmpBuf d_ref, d_test, d_res;
  const size_t bytes = stride * channels * sizeof(float);
  HIP_TRY(d_ref.reserve(bytes));
  HIP_TRY(d_test.reserve(bytes));
  HIP_TRY(d_res.reserve(sizeof(peaq_result)));
  HIP_TRY(hipMemset(d_ref.p, 0, bytes));
  HIP_TRY(hipMemset(d_test.p, 0, bytes));
  if (n_ref) HIP_TRY(hipMemcpy(d_ref.p, ref, n_ref * channels * sizeof(float), hipMemcpyHostToDevice));
  if (n_test) HIP_TRY(hipMemcpy(d_test.p, test, n_test * channels * sizeof(float), hipMemcpyHostToDevice));
  const uint32_t h_n[2] = {(uint32_t)n_ref, (uint32_t)n_test};    // peaq_batch_run takes the lengths as HOST arrays
  const int rc = peaq_batch_run(c, advanced, channels, level_db, 1, d_ref.as<float>(), d_test.as<float>(), stride,
                                h_n, h_n + 1, 0, d_res.as<peaq_result>(), nullptr);
  if (rc != PEAQ_OK) return rc;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, d_res.p, sizeof(peaq_result), hipMemcpyDeviceToHost));
  return PEAQ_OK;
}

static int batch_run_locked(peaq_ctx* c, int advanced, int channels, double level_db, int n_pairs, const float* d_ref,
                            const float* d_test, size_t pair_stride, const uint32_t* n_ref, const uint32_t* n_test,
                            uint32_t n_uniform, peaq_result* d_results, hipStream_t stream) {

  // ---- frame counts -------------------------------------------------------------------
  uint32_t max_frames = 0, max_blocks = 0;
  const uint32_t* d_nref = nullptr;
  const uint32_t* d_ntest = nullptr;
  const uint32_t* d_nframes = nullptr;
  const uint32_t* d_nblocks = nullptr;
  if (n_ref) {
    std::vector<uint32_t> h(4 * (size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
      if (n_ref[p] > pair_stride || n_test[p] > pair_stride)
        return fail(PEAQ_ERR_ARG, "peaq_batch_run: a pair is longer than pair_stride");
      h[p] = n_ref[p];
      h[n_pairs + p] = n_test[p];
      h[2 * (size_t)n_pairs + p] = count_frames(n_ref[p], n_test[p], kFrame, kHop);
      h[3 * (size_t)n_pairs + p] = count_frames(n_ref[p], n_test[p], kFbFrame, kFbFrame);
      max_frames = std::max(max_frames, h[2 * (size_t)n_pairs + p]);
      max_blocks = std::max(max_blocks, h[3 * (size_t)n_pairs + p]);
    }
    HIP_TRY(c->counts.reserve(h.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(c->counts.p, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));      // h goes out of scope
    d_nref = c->counts.as<uint32_t>();
    d_ntest = d_nref + n_pairs;
    d_nframes = d_nref + 2 * (size_t)n_pairs;
    d_nblocks = d_nref + 3 * (size_t)n_pairs;
  } else {
    if (n_uniform > pair_stride) return fail(PEAQ_ERR_ARG, "peaq_batch_run: n_uniform > pair_stride");
    max_frames = count_frames(n_uniform, n_uniform, kFrame, kHop);
    max_blocks = count_frames(n_uniform, n_uniform, kFbFrame, kFbFrame);
  }

  const unsigned fc = frames_per_chunk(n_pairs, channels, max_frames);
  const size_t rec_bytes = (size_t)n_pairs * fc * channels * kRecDoubles * sizeof(double);
  HIP_TRY(c->records.reserve(rec_bytes));
  HIP_TRY(c->records2.reserve(rec_bytes));
  HIP_TRY(c->state.reserve((size_t)n_pairs * sizeof(PairState)));

  HIP_TRY(hipEventRecord(c->batch_begin, stream));
  HIP_TRY(launch_state_init(c->state.as<PairState>(), advanced, n_pairs, stream));
  hipEvent_t fb_done = nullptr;
  if (advanced && max_blocks > 0) {
    // the filter-bank path (its own ear model, accumulators 0, 1, 4) is independent of the FFT
    // path (accumulators 2, 3): it runs on a third stream from here on and joins at the end
    hipEvent_t forked = c->next_event();
    if (!forked) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(forked, stream));
    HIP_TRY(hipStreamWaitEvent(c->aux2, forked, 0));
    hipStream_t s_fb = PEAQ_DEV_SERIAL_KERNELS ? stream : c->aux2;
    const int rc = run_filterbank_path(c, channels, level_db, n_pairs, d_ref, d_test, pair_stride, d_nref, d_ntest,
                                       n_uniform, d_nblocks, max_blocks, s_fb);
    if (rc != PEAQ_OK) return rc;
    fb_done = c->next_event();
    if (!fb_done) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(fb_done, s_fb));
  }

  FrontendArgs fa{};
  fa.cfg = c->settings;
  fa.ref = d_ref;
  fa.test = d_test;
  fa.pair_stride = pair_stride;
  fa.n_ref = d_nref;
  fa.n_test = d_ntest;
  fa.n_uniform_ref = n_uniform;
  fa.n_uniform_test = n_uniform;
  fa.n_frames = d_nframes;
  fa.n_frames_uniform = max_frames;
  fa.frame_origin = 0;
  fa.off_ref = 0;
  fa.off_test = 0;
  fa.channels = channels;
  fa.level_factor = fft_level_factor(level_db);
  fa.common = c->d_common;
  fa.bands = advanced ? c->d_bands55 : c->d_bands109;    // gstpeaq.c:521-526
  fa.prof = c->d_prof;
  BackendArgs ba{};
  ba.cfg = c->settings;
  ba.n_frames = d_nframes;
  ba.n_frames_uniform = max_frames;
  ba.channels = channels;
  ba.advanced = advanced ? 1 : 0;
  ba.bands = fa.bands;
  ba.common = c->d_common;
  ba.state = c->state.as<PairState>();

  // Software pipeline over chunks of frames: the front end of chunk i+1 (throughput bound,
  // millions of workgroups) runs on the caller's stream while the back end of chunk i
  // (one workgroup per pair, latency bound) runs on the context's second stream; the
  // per-frame records are double buffered.
  hipEvent_t back_done[2] = {nullptr, nullptr};
  unsigned chunk = 0;
  for (uint32_t f0 = 0; f0 < max_frames; f0 += fc, ++chunk) {
    const unsigned nf = std::min<uint32_t>(fc, max_frames - f0);
    double* recs = (chunk & 1) ? c->records2.as<double>() : c->records.as<double>();
    fa.frame0 = f0;
    fa.frames_per_launch = nf;
    fa.records = recs;
    ba.frame0 = f0;
    ba.frames_per_launch = nf;
    ba.records = recs;
    hipEvent_t e0 = c->next_event(), e1 = c->next_event(), e2 = c->next_event(), e3 = c->next_event();
    if (!e0 || !e1 || !e2 || !e3) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
    if (back_done[chunk & 1]) HIP_TRY(hipStreamWaitEvent(stream, back_done[chunk & 1], 0));   // buffer free again
    HIP_TRY(hipEventRecord(e0, stream));
    HIP_TRY(launch_frontend(advanced ? 55 : 109, fa, n_pairs, stream));
    HIP_TRY(hipEventRecord(e1, stream));
    HIP_TRY(hipStreamWaitEvent(c->aux, e1, 0));
    HIP_TRY(hipEventRecord(e2, c->aux));
    if (!PEAQ_DEV_SKIP_BACKEND) HIP_TRY(launch_backend(ba, n_pairs, c->aux));
    HIP_TRY(hipEventRecord(e3, c->aux));
    back_done[chunk & 1] = e3;
    c->spans.push_back({e0, e1, 0});
    c->spans.push_back({e2, e3, 1});
  }
  for (int i = 0; i < 2; ++i)
    if (back_done[i]) HIP_TRY(hipStreamWaitEvent(stream, back_done[i], 0));
  if (fb_done) HIP_TRY(hipStreamWaitEvent(stream, fb_done, 0));
  HIP_TRY(launch_finalize(c->state.as<PairState>(), advanced, channels, n_pairs,
                          reinterpret_cast<ResultRecord*>(d_results), stream, c->settings));
  HIP_TRY(hipEventRecord(c->batch_end, stream));
  c->batch_pending = true;
  return PEAQ_OK;
}

extern "C" int peaq_batch_last_timing(peaq_ctx* c, peaq_batch_timing* out) {
  if (!c || !out) return fail(PEAQ_ERR_ARG, "peaq_batch_last_timing: NULL argument");
  std::lock_guard<std::mutex> lock(c->mu);
  std::memset(out, 0, sizeof *out);
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->batch_end));
  c->batch_pending = false;
  HIP_TRY(hipEventElapsedTime(&out->total_ms, c->batch_begin, c->batch_end));
  for (const TimedSpan& s : c->spans) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
    if (s.kind == 0) {
      out->frontend_ms += ms;
      out->frontend_launches++;
    } else if (s.kind == 1) {
      out->backend_ms += ms;
      out->backend_launches++;
    } else {
      out->fb_ms += ms;
      out->fb_launches++;
    }
  }
  return PEAQ_OK;
}

// ---------------------------------------------------------------------------
// synthetic workload
// ---------------------------------------------------------------------------
extern "C" int peaq_synth_fill(peaq_ctx* c, uint32_t seed0, int n_pairs, int channels, uint32_t n_samples,
                               size_t pair_stride, float* d_ref, float* d_test, void* stream) {
  if (!c || !d_ref || !d_test) return fail(PEAQ_ERR_ARG, "peaq_synth_fill: NULL argument");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_synth_fill: channels must be 1 or 2");
  if (n_samples > pair_stride) return fail(PEAQ_ERR_ARG, "peaq_synth_fill: n_samples > pair_stride");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(launch_synth(seed0, n_pairs, channels, n_samples, pair_stride, d_ref, d_test,
                       static_cast<hipStream_t>(stream)));
  return PEAQ_OK;
}

// ---------------------------------------------------------------------------
// stage-level access for parity tests
// ---------------------------------------------------------------------------
extern "C" int peaq_debug_frontend(peaq_ctx* c, int bands, int channels, double level_db, const float* d_ref,
                                   const float* d_test, uint32_t n_ref, uint32_t n_test, int n_frames,
                                   double* host_out) {
  if (!c || !d_ref || !d_test || !host_out) return fail(PEAQ_ERR_ARG, "peaq_debug_frontend: NULL argument");
  if (bands != 109 && bands != 55) return fail(PEAQ_ERR_ARG, "peaq_debug_frontend: bands must be 109 or 55");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_debug_frontend: channels must be 1 or 2");
  const uint32_t total = count_frames(n_ref, n_test, kFrame, kHop);
  if (n_frames < 0 || (uint32_t)n_frames > total) return fail(PEAQ_ERR_ARG, "peaq_debug_frontend: too many frames");
  if (n_frames == 0) return PEAQ_OK;
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = (size_t)n_frames * channels * kRecDoubles * sizeof(double);
  TmpBuf rec_buf, n_buf;
  HIP_TRY(rec_buf.reserve(bytes));
  double* d_rec = rec_buf.as<double>();
  HIP_TRY(hipMemset(d_rec, 0, bytes));
  uint32_t h_n[2] = {n_ref, n_test};
  HIP_TRY(n_buf.reserve(sizeof h_n));
  uint32_t* d_n = n_buf.as<uint32_t>();
  HIP_TRY(hipMemcpy(d_n, h_n, sizeof h_n, hipMemcpyHostToDevice));
  FrontendArgs fa{};
  fa.cfg = c->settings;
  fa.ref = d_ref;
  fa.test = d_test;
  fa.pair_stride = std::max(n_ref, n_test);
  fa.n_ref = d_n;
  fa.n_test = d_n + 1;
  fa.n_frames = nullptr;
  fa.n_frames_uniform = total;
  fa.channels = channels;
  fa.frame0 = 0;
  fa.frames_per_launch = n_frames;
  fa.level_factor = fft_level_factor(level_db);
  fa.common = c->d_common;
  fa.bands = bands == 109 ? c->d_bands109 : c->d_bands55;
  fa.records = d_rec;
  std::vector<double> h_rec((size_t)n_frames * channels * kRecDoubles);
  // one pair: launches of at most max_frames_per_launch(1) frames, the records of a launch follow the previous one's
  hipError_t e = hipSuccess;
  for (uint32_t f0 = 0; f0 < (uint32_t)n_frames && e == hipSuccess; f0 += max_frames_per_launch(1)) {
    fa.frame0 = f0;
    fa.frames_per_launch = std::min<uint32_t>(max_frames_per_launch(1), (uint32_t)n_frames - f0);
    fa.records = d_rec + (size_t)f0 * channels * kRecDoubles;
    e = launch_frontend(bands, fa, 1, nullptr);
  }
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(h_rec.data(), d_rec, bytes, hipMemcpyDeviceToHost);
  if (e != hipSuccess) return fail(PEAQ_ERR_DEVICE, std::string("peaq_debug_frontend: ") + hipGetErrorString(e));
  // the test-facing layout spells the two derived vectors out (the back end's own arithmetic, on the host)
  BandTables t;
  build_fft_band_tables(bands, t);
  for (size_t r = 0; r < (size_t)n_frames * channels; ++r) {
    const double* in = h_rec.data() + r * kRecDoubles;
    double* out = host_out + r * kPubDoubles;
    for (int b = 0; b < kBandStride; ++b) {
      excitation_from_root(in[kRecRootRef + b], t.inv_spread_norm[b], t.inv_spread_norm_pow03[b], out[kPubUnsmRef + b],
                           out[kPubLoudRef + b]);
      excitation_from_root(in[kRecRootTest + b], t.inv_spread_norm[b], t.inv_spread_norm_pow03[b],
                           out[kPubUnsmTest + b], out[kPubLoudTest + b]);
      out[kPubNoise + b] = in[kRecNoise + b];
    }
    for (int i = 0; i < kPubDoubles - kPubScalars; ++i) out[kPubScalars + i] = in[kRecScalars + i];
  }
  return PEAQ_OK;
}

extern "C" int peaq_debug_filterbank(peaq_ctx* c, int channels, double level_db, const float* d_ref,
                                     const float* d_test, uint32_t n_ref, uint32_t n_test, int n_blocks,
                                     int blocks_per_launch, double* host_out) {
  if (!c || !d_ref || !d_test || !host_out) return fail(PEAQ_ERR_ARG, "peaq_debug_filterbank: NULL argument");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_debug_filterbank: channels must be 1 or 2");
  const uint32_t total = count_frames(n_ref, n_test, kFbFrame, kFbFrame);
  if (n_blocks < 0 || (uint32_t)n_blocks > total || blocks_per_launch < 1)
    return fail(PEAQ_ERR_ARG, "peaq_debug_filterbank: bad block counts");
  if (n_blocks == 0) return PEAQ_OK;
  HIP_TRY(hipSetDevice(c->device));
  const unsigned n_signals = 2 * channels;
  const size_t row_stride = (size_t)kFbRing + (size_t)blocks_per_launch * kFbFrame;
  TmpBuf rows, recs, st;
  HIP_TRY(rows.reserve(n_signals * row_stride * sizeof(double)));
  HIP_TRY(recs.reserve((size_t)blocks_per_launch * channels * kFbRecDoubles * sizeof(double)));
  HIP_TRY(st.reserve(n_signals * sizeof(FbSignalState)));
  HIP_TRY(hipMemset(st.p, 0, n_signals * sizeof(FbSignalState)));
  FbFrontArgs ff{};
  ff.cfg = c->settings;
  ff.fir_fp64 = c->fir_fp64;
  ff.ref = d_ref;
  ff.test = d_test;
  ff.pair_stride = std::max(n_ref, n_test);
  ff.n_uniform_ref = n_ref;
  ff.n_uniform_test = n_test;
  ff.n_blocks_uniform = n_blocks;
  ff.channels = channels;
  ff.level_factor = fb_level_factor(level_db);
  set_fir_scale(ff);
  ff.bands = c->d_bands40;
  ff.fb = c->d_fb;
  ff.fbstate = st.as<FbSignalState>();
  ff.hp_scratch = rows.as<double>();
  ff.hp_row_stride = row_stride;
  ff.records = recs.as<double>();
  hipError_t e = hipSuccess;
  unsigned prev = 0;
  for (int b0 = 0; b0 < n_blocks && e == hipSuccess; b0 += blocks_per_launch) {
    const unsigned nb = std::min(blocks_per_launch, n_blocks - b0);
    ff.block0 = b0;
    ff.blocks_per_launch = nb;
    ff.prev_blocks = prev;
    ff.first_launch = b0 == 0;
    e = hipMemset(recs.p, 0, recs.cap);
    if (e == hipSuccess) e = launch_fb_frontend(ff, 1, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess)
      e = hipMemcpy(host_out + (size_t)b0 * channels * kFbRecDoubles, recs.p,
                    (size_t)nb * channels * kFbRecDoubles * sizeof(double), hipMemcpyDeviceToHost);
    prev = nb;
  }
  if (e != hipSuccess) return fail(PEAQ_ERR_DEVICE, std::string("peaq_debug_filterbank: ") + hipGetErrorString(e));
  return PEAQ_OK;
}

extern "C" int peaq_debug_backend(peaq_ctx* c, int channels, int n_frames, const double* host_records,
                                  double* host_out, peaq_result* result) {
  if (!c || !host_records || !host_out) return fail(PEAQ_ERR_ARG, "peaq_debug_backend: NULL argument");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_debug_backend: channels must be 1 or 2");
  if (n_frames < 1) return fail(PEAQ_ERR_ARG, "peaq_debug_backend: n_frames < 1");
  HIP_TRY(hipSetDevice(c->device));
  const size_t rec_bytes = (size_t)n_frames * channels * kRecDoubles * sizeof(double);
  const size_t dbg_bytes = (size_t)n_frames * channels * kDbgDoubles * sizeof(double);
  TmpBuf recs, dbg, st, res;
  HIP_TRY(recs.reserve(rec_bytes));
  HIP_TRY(dbg.reserve(dbg_bytes));
  HIP_TRY(st.reserve(sizeof(PairState)));
  HIP_TRY(res.reserve(sizeof(ResultRecord)));
  {
    // test-facing layout -> the record the kernels exchange: root = (E norm)^(1/10); the E^0.3 vector of the
    // input is implied by E (the back end derives both from the root)
    BandTables t;
    build_fft_band_tables(109, t);
    std::vector<double> h_rec((size_t)n_frames * channels * kRecDoubles, 0.);
    for (size_t r = 0; r < (size_t)n_frames * channels; ++r) {
      const double* in = host_records + r * kPubDoubles;
      double* out = h_rec.data() + r * kRecDoubles;
      for (int b = 0; b < 109; ++b) {
        out[kRecRootRef + b] = std::pow(in[kPubUnsmRef + b] / t.inv_spread_norm[b], 0.1);
        out[kRecRootTest + b] = std::pow(in[kPubUnsmTest + b] / t.inv_spread_norm[b], 0.1);
      }
      for (int b = 0; b < kBandStride; ++b) out[kRecNoise + b] = in[kPubNoise + b];
      for (int i = 0; i < kRecDoubles - kRecScalars; ++i) out[kRecScalars + i] = in[kPubScalars + i];
    }
    HIP_TRY(hipMemcpy(recs.p, h_rec.data(), rec_bytes, hipMemcpyHostToDevice));
  }
  HIP_TRY(hipMemset(dbg.p, 0, dbg_bytes));
  HIP_TRY(launch_state_init(st.as<PairState>(), 0, 1, nullptr));
  BackendArgs ba{};
  ba.cfg = c->settings;
  ba.records = recs.as<double>();
  ba.frame0 = 0;
  ba.frames_per_launch = n_frames;
  ba.n_frames_uniform = n_frames;
  ba.channels = channels;
  ba.advanced = 0;
  ba.bands = c->d_bands109;
  ba.common = c->d_common;
  ba.state = st.as<PairState>();
  ba.debug = dbg.as<double>();
  HIP_TRY(launch_backend(ba, 1, nullptr));
  HIP_TRY(launch_finalize(st.as<PairState>(), 0, channels, 1, res.as<ResultRecord>(), nullptr, c->settings));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(host_out, dbg.p, dbg_bytes, hipMemcpyDeviceToHost));
  if (result) HIP_TRY(hipMemcpy(result, res.p, sizeof(peaq_result), hipMemcpyDeviceToHost));
  return PEAQ_OK;
}

// ---------------------------------------------------------------------------
// sessions: one per `peaq` element instance
// ---------------------------------------------------------------------------
namespace {

constexpr unsigned kSessionMaxFrames = 64;   // FFT frames per launch of a session
constexpr unsigned kSessionMaxBlocks = 120;  // filter-bank blocks per launch of a session

// host-side stand-in for a GstAdapter: the not yet consumed tail of one pad's stream.
// Consumed samples are skipped with a read offset and the storage is compacted only once more
// than half of it is dead, so a pad that runs far ahead of the other one (a whole file pushed on
// `ref` before `test` starts) costs O(n) in total, like gst_adapter_flush, not O(n^2).
struct PadFifo {
  std::vector<float> buf;   // interleaved; live data starts at buf[head]
  size_t head = 0;
  uint64_t base = 0;        // stream sample index (per channel) of buf[head]
  uint64_t total = 0;       // samples (per channel) pushed so far
  const float* at(uint64_t pos, int channels) const { return buf.data() + head + (size_t)(pos - base) * channels; }
  size_t live_floats() const { return buf.size() - head; }
  void append(const float* data, size_t n_floats) { buf.insert(buf.end(), data, data + n_floats); }
  void drop_until(uint64_t keep_from, int channels) {
    if (keep_from <= base) return;
    const size_t drop = std::min((size_t)(keep_from - base) * channels, live_floats());
    head += drop;
    base = keep_from;
    if (head == buf.size()) {
      buf.clear();
      head = 0;
    } else if (head >= 65536 && head > buf.size() / 2) {
      buf.erase(buf.begin(), buf.begin() + head);
      head = 0;
    }
  }
};

}  // namespace

struct peaq_session {
  peaq_ctx* ctx = nullptr;
  int advanced = 0, channels = 1;
  double level_db = 92.;
  Settings cfg;                     // the context's settings when the session was created
  std::mutex mu;
  PadFifo pad[2];
  uint64_t fft_pos[2] = {0, 0};   // stream sample where the next FFT frame starts, per pad
  uint64_t fb_pos[2] = {0, 0};    // ... where the next filter-bank block starts (advanced)
  uint32_t frames_done = 0;
  uint32_t blocks_done = 0;
  uint32_t fb_prev_blocks = 0;
  bool fb_first = true;
  DevBuf fb_records, fbstate, hp_rows;
  hipStream_t stream = nullptr;
  hipEvent_t staged = nullptr;    // the pinned staging buffers may be rewritten after this
  bool staged_pending = false;
  float* h_stage[2] = {nullptr, nullptr};   // pinned
  DevBuf d_sig[2], records, state, result;
  size_t stage_samples = 0;
};

static int session_alloc(peaq_session* s) {
  s->stage_samples = (size_t)(kSessionMaxFrames - 1) * kHop + kFrame;   // >= kSessionMaxBlocks * 192
  const size_t bytes = s->stage_samples * s->channels * sizeof(float);
  for (int p = 0; p < 2; ++p) {
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&s->h_stage[p]), bytes, hipHostMallocDefault));
    HIP_TRY(s->d_sig[p].reserve(bytes));
  }
  HIP_TRY(s->records.reserve((size_t)kSessionMaxFrames * s->channels * kRecDoubles * sizeof(double)));
  HIP_TRY(s->state.reserve(sizeof(PairState)));
  HIP_TRY(s->result.reserve(sizeof(ResultRecord)));
  HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&s->staged, hipEventDisableTiming));
  HIP_TRY(launch_state_init(s->state.as<PairState>(), s->advanced, 1, s->stream));
  if (s->advanced) {
    const unsigned n_signals = 2 * s->channels;
    HIP_TRY(s->fb_records.reserve((size_t)kSessionMaxBlocks * s->channels * kFbRecDoubles * sizeof(double)));
    HIP_TRY(s->fbstate.reserve(n_signals * sizeof(FbSignalState)));
    HIP_TRY(hipMemsetAsync(s->fbstate.p, 0, n_signals * sizeof(FbSignalState), s->stream));
    HIP_TRY(s->hp_rows.reserve((size_t)n_signals * (kFbRing + (size_t)kSessionMaxBlocks * kFbFrame) * sizeof(double)));
  }
  return PEAQ_OK;
}

extern "C" int peaq_session_create(peaq_ctx* c, int advanced, int channels, double level_db, peaq_session** out) {
  if (!c || !out) return fail(PEAQ_ERR_ARG, "peaq_session_create: NULL argument");
  *out = nullptr;
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_session_create: channels must be 1 or 2");
  if (!(level_db >= 0. && level_db <= 130.))
    return fail(PEAQ_ERR_ARG, "peaq_session_create: playback level outside 0..130 dB (gstpeaq.c:275-281)");
  HIP_TRY(hipSetDevice(c->device));
  peaq_session* s = new (std::nothrow) peaq_session;
  if (!s) return fail(PEAQ_ERR_NOMEM, "out of host memory");
  s->ctx = c;
  s->cfg = c->settings;
  s->advanced = advanced ? 1 : 0;
  s->channels = channels;
  s->level_db = level_db;
  const int rc = session_alloc(s);
  if (rc != PEAQ_OK) {
    peaq_session_destroy(s);
    return rc;
  }
  *out = s;
  return PEAQ_OK;
}

extern "C" void peaq_session_destroy(peaq_session* s) {
  if (!s) return;
  (void)hipSetDevice(s->ctx->device);
  if (s->stream) (void)hipStreamSynchronize(s->stream);
  for (int p = 0; p < 2; ++p) {
    if (s->h_stage[p]) (void)hipHostFree(s->h_stage[p]);
    s->d_sig[p].release();
  }
  s->records.release();
  s->state.release();
  s->result.release();
  s->fb_records.release();
  s->fbstate.release();
  s->hp_rows.release();
  if (s->staged) (void)hipEventDestroy(s->staged);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

// run `nf` FFT frames whose first sample is fft_pos[] on each pad; the two
// signals contribute n_valid[] samples (shorter than a whole frame only for
// the flush frame).
static int session_stage(peaq_session* s, const uint64_t pos[2], const uint64_t n_valid[2]) {
  if (s->staged_pending) {
    HIP_TRY(hipEventSynchronize(s->staged));
    s->staged_pending = false;
  }
  for (int p = 0; p < 2; ++p) {
    const PadFifo& f = s->pad[p];
    const size_t cnt = (size_t)n_valid[p] * s->channels;
    if (cnt) {
      std::memcpy(s->h_stage[p], f.at(pos[p], s->channels), cnt * sizeof(float));
      HIP_TRY(hipMemcpyAsync(s->d_sig[p].p, s->h_stage[p], cnt * sizeof(float), hipMemcpyHostToDevice, s->stream));
    }
  }
  HIP_TRY(hipEventRecord(s->staged, s->stream));
  s->staged_pending = true;
  return PEAQ_OK;
}

static int session_run_frames(peaq_session* s, unsigned nf, const uint64_t n_valid[2]) {
  peaq_ctx* c = s->ctx;
  {
    const int rc = session_stage(s, s->fft_pos, n_valid);
    if (rc != PEAQ_OK) return rc;
  }
  FrontendArgs fa{};
  fa.cfg = s->cfg;
  fa.ref = s->d_sig[0].as<float>();
  fa.test = s->d_sig[1].as<float>();
  fa.pair_stride = s->stage_samples;
  fa.n_uniform_ref = static_cast<uint32_t>(n_valid[0]);
  fa.n_uniform_test = static_cast<uint32_t>(n_valid[1]);
  fa.n_frames_uniform = s->frames_done + nf;
  fa.frame_origin = s->frames_done;
  fa.channels = s->channels;
  fa.frame0 = s->frames_done;
  fa.frames_per_launch = nf;
  fa.level_factor = fft_level_factor(s->level_db);
  fa.common = c->d_common;
  fa.bands = s->advanced ? c->d_bands55 : c->d_bands109;
  fa.records = s->records.as<double>();
  HIP_TRY(launch_frontend(s->advanced ? 55 : 109, fa, 1, s->stream));
  BackendArgs ba{};
  ba.cfg = s->cfg;
  ba.records = fa.records;
  ba.frame0 = s->frames_done;
  ba.frames_per_launch = nf;
  ba.n_frames_uniform = s->frames_done + nf;
  ba.channels = s->channels;
  ba.advanced = s->advanced;
  ba.bands = fa.bands;
  ba.common = c->d_common;
  ba.state = s->state.as<PairState>();
  HIP_TRY(launch_backend(ba, 1, s->stream));
  s->frames_done += nf;
  return PEAQ_OK;
}

// run `nb` filter-bank blocks starting at fb_pos[] (advanced mode)
static int session_run_blocks(peaq_session* s, unsigned nb, const uint64_t n_valid[2]) {
  peaq_ctx* c = s->ctx;
  {
    const int rc = session_stage(s, s->fb_pos, n_valid);
    if (rc != PEAQ_OK) return rc;
  }
  FbFrontArgs ff{};
  ff.cfg = s->cfg;
  ff.fir_fp64 = c->fir_fp64;
  ff.ref = s->d_sig[0].as<float>();
  ff.test = s->d_sig[1].as<float>();
  ff.pair_stride = s->stage_samples;
  ff.n_uniform_ref = static_cast<uint32_t>(n_valid[0]);
  ff.n_uniform_test = static_cast<uint32_t>(n_valid[1]);
  ff.n_blocks_uniform = s->blocks_done + nb;
  ff.block_origin = s->blocks_done;
  ff.channels = s->channels;
  ff.block0 = s->blocks_done;
  ff.blocks_per_launch = nb;
  ff.prev_blocks = s->fb_prev_blocks;
  ff.first_launch = s->fb_first;
  ff.level_factor = fb_level_factor(s->level_db);
  set_fir_scale(ff);
  ff.bands = c->d_bands40;
  ff.fb = c->d_fb;
  ff.fbstate = s->fbstate.as<FbSignalState>();
  ff.hp_scratch = s->hp_rows.as<double>();
  ff.hp_row_stride = kFbRing + (size_t)kSessionMaxBlocks * kFbFrame;
  ff.records = s->fb_records.as<double>();
  HIP_TRY(launch_fb_frontend(ff, 1, s->stream));
  FbBackendArgs fbk{};
  fbk.cfg = s->cfg;
  fbk.records = ff.records;
  fbk.block0 = s->blocks_done;
  fbk.blocks_per_launch = nb;
  fbk.n_blocks_uniform = s->blocks_done + nb;
  fbk.channels = s->channels;
  fbk.bands = c->d_bands40;
  fbk.common = c->d_common;
  fbk.state = s->state.as<PairState>();
  HIP_TRY(launch_fb_backend(fbk, 1, s->stream));
  s->blocks_done += nb;
  s->fb_prev_blocks = nb;
  s->fb_first = false;
  return PEAQ_OK;
}

static void session_trim(peaq_session* s) {
  for (int p = 0; p < 2; ++p) {
    PadFifo& f = s->pad[p];
    const uint64_t keep_from = s->advanced ? std::min(s->fft_pos[p], s->fb_pos[p]) : s->fft_pos[p];
    f.drop_until(keep_from, s->channels);
  }
}

// do_processing (gstpeaq.c:596-611)
static int session_drain(peaq_session* s) {
  for (;;) {
    const uint64_t av = std::min(s->pad[0].total - s->fft_pos[0], s->pad[1].total - s->fft_pos[1]);
    if (av < (uint64_t)kFrame) break;
    const uint64_t ready = (av - kFrame) / kHop + 1;
    const unsigned nf = static_cast<unsigned>(std::min<uint64_t>(ready, kSessionMaxFrames));
    const uint64_t need = (uint64_t)(nf - 1) * kHop + kFrame;
    const uint64_t nv[2] = {need, need};
    const int rc = session_run_frames(s, nf, nv);
    if (rc != PEAQ_OK) return rc;
    s->fft_pos[0] += (uint64_t)nf * kHop;
    s->fft_pos[1] += (uint64_t)nf * kHop;
  }
  if (s->advanced) {
    for (;;) {
      const uint64_t av = std::min(s->pad[0].total - s->fb_pos[0], s->pad[1].total - s->fb_pos[1]);
      if (av < (uint64_t)kFbFrame) break;
      const unsigned nb = static_cast<unsigned>(std::min<uint64_t>(av / kFbFrame, kSessionMaxBlocks));
      const uint64_t nv[2] = {(uint64_t)nb * kFbFrame, (uint64_t)nb * kFbFrame};
      const int rc = session_run_blocks(s, nb, nv);
      if (rc != PEAQ_OK) return rc;
      s->fb_pos[0] += nv[0];
      s->fb_pos[1] += nv[1];
    }
  }
  session_trim(s);
  return PEAQ_OK;
}

extern "C" int peaq_session_push(peaq_session* s, int pad, const float* data, size_t n) {
  if (!s) return fail(PEAQ_ERR_ARG, "peaq_session_push: session is NULL");
  if (pad != 0 && pad != 1) return fail(PEAQ_ERR_ARG, "peaq_session_push: pad must be 0 (ref) or 1 (test)");
  if (n == 0) return PEAQ_OK;
  if (!data) return fail(PEAQ_ERR_ARG, "peaq_session_push: data is NULL");
  std::lock_guard<std::mutex> lock(s->mu);          // GST_OBJECT_LOCK in pad_chain (gstpeaq.c:619)
  HIP_TRY(hipSetDevice(s->ctx->device));
  PadFifo& f = s->pad[pad];
  try {
    f.append(data, n * s->channels);
  } catch (const std::bad_alloc&) {
    return fail(PEAQ_ERR_NOMEM, "out of host memory");
  }
  f.total += n;
  return session_drain(s);
}

// do_flush (gstpeaq.c:716-745)
extern "C" int peaq_session_flush(peaq_session* s) {
  if (!s) return fail(PEAQ_ERR_ARG, "peaq_session_flush: session is NULL");
  std::lock_guard<std::mutex> lock(s->mu);
  HIP_TRY(hipSetDevice(s->ctx->device));
  const uint64_t left_r = s->pad[0].total - s->fft_pos[0], left_t = s->pad[1].total - s->fft_pos[1];
  if (left_r || left_t) {
    const uint64_t nv[2] = {std::min<uint64_t>(left_r, kFrame), std::min<uint64_t>(left_t, kFrame)};
    const int rc = session_run_frames(s, 1, nv);
    if (rc != PEAQ_OK) return rc;
    s->fft_pos[0] += nv[0];
    s->fft_pos[1] += nv[1];
  }
  if (s->advanced) {                                 // gstpeaq.c:769-771
    const uint64_t lr = s->pad[0].total - s->fb_pos[0], lt = s->pad[1].total - s->fb_pos[1];
    if (lr || lt) {
      const uint64_t nv[2] = {std::min<uint64_t>(lr, kFbFrame), std::min<uint64_t>(lt, kFbFrame)};
      const int rc = session_run_blocks(s, 1, nv);
      if (rc != PEAQ_OK) return rc;
      s->fb_pos[0] += nv[0];
      s->fb_pos[1] += nv[1];
    }
  }
  session_trim(s);
  return PEAQ_OK;
}

extern "C" int peaq_session_results(peaq_session* s, peaq_result* out) {
  if (!s || !out) return fail(PEAQ_ERR_ARG, "peaq_session_results: NULL argument");
  std::lock_guard<std::mutex> lock(s->mu);
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(launch_finalize(s->state.as<PairState>(), s->advanced, s->channels, 1, s->result.as<ResultRecord>(),
                          s->stream, s->cfg));
  HIP_TRY(hipMemcpyAsync(out, s->result.p, sizeof(peaq_result), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return PEAQ_OK;
}

extern "C" int peaq_session_set_level(peaq_session* s, double level_db) {
  if (!s) return fail(PEAQ_ERR_ARG, "peaq_session_set_level: session is NULL");
  if (!(level_db >= 0. && level_db <= 130.))
    return fail(PEAQ_ERR_ARG, "peaq_session_set_level: playback level outside 0..130 dB (gstpeaq.c:275-281)");
  std::lock_guard<std::mutex> lock(s->mu);
  s->level_db = level_db;            // the level factors are per-launch kernel arguments
  return PEAQ_OK;
}

extern "C" int peaq_session_reset(peaq_session* s) {
  if (!s) return fail(PEAQ_ERR_ARG, "peaq_session_reset: session is NULL");
  std::lock_guard<std::mutex> lock(s->mu);
  HIP_TRY(hipSetDevice(s->ctx->device));
  HIP_TRY(hipStreamSynchronize(s->stream));
  for (int p = 0; p < 2; ++p) {
    s->pad[p] = PadFifo();
    s->fft_pos[p] = 0;
    s->fb_pos[p] = 0;
  }
  s->frames_done = 0;
  s->blocks_done = 0;
  s->fb_prev_blocks = 0;
  s->fb_first = true;
  if (s->advanced)
    HIP_TRY(hipMemsetAsync(s->fbstate.p, 0, 2 * s->channels * sizeof(FbSignalState), s->stream));
  HIP_TRY(launch_state_init(s->state.as<PairState>(), s->advanced, 1, s->stream));
  return PEAQ_OK;
}

// ---------------------------------------------------------------------------
// broker: many live sessions, one launch per tick
// ---------------------------------------------------------------------------
// A process that hosts many `peaq` elements (BASELINE.json configs[5]: 1024
// concurrent live pipelines) would otherwise issue one 1-workgroup launch pair
// per element and buffer.  The broker keeps the FIFOs of all its sessions on
// the host and, on every tick, gathers whatever frames (and, in the advanced
// version, filter-bank blocks) became ready in ANY session into ONE launch per
// kernel: workgroup (pair p, frame fl) of the front-end grid is frame
// pair_frame0[p] + fl of session pair_slot[p]; the filter-bank kernels get a
// FbPairWindow per session.  The recurrent state of a session stays in its slot
// in HBM between ticks, so the result of a session is the same whether its
// frames were run alone, in a batch, or interleaved with other sessions'.
// The framing per session is that of do_processing / do_flush
// (gstpeaq.c:596-611, 716-745, 769-771).
namespace {
constexpr unsigned kBrokerMaxFrames = 8;     // FFT frames one session contributes to one tick
constexpr unsigned kBrokerMaxBlocks = 48;    // filter-bank blocks one session contributes to one tick
constexpr size_t kBrokerStageSamples = (size_t)(kBrokerMaxFrames - 1) * kHop + kFrame;
constexpr size_t kBrokerFbStageSamples = (size_t)kBrokerMaxBlocks * kFbFrame;
constexpr size_t kBrokerRowStride = (size_t)kFbRing + kBrokerFbStageSamples;
// back-pressure: a push returns only once its session has no more than this many samples that are
// READY to be framed (present on both pads) and not yet launched -- four ticks' worth.  Samples one
// pad holds ahead of the other are never counted: like the reference's adapters they may pile up
// without bound while the other pad is silent (gstpeaq.c:626-636).
constexpr uint64_t kBrokerBacklog = (uint64_t)4 * kBrokerMaxFrames * kHop;

struct BrokerSlot {
  std::mutex mu;
  bool open = false;
  bool flush_requested = false, fft_flushed = false, fb_flushed = false;
  PadFifo pad[2];
  uint64_t fft_pos[2] = {0, 0};
  uint64_t fb_pos[2] = {0, 0};
  uint32_t frames_done = 0, blocks_done = 0, fb_prev_blocks = 0;
  double t_ready_us = -1.;          // when the oldest not yet launched frame / block became whole (-1: none waiting)
};

// microseconds on a monotonic clock (latency bookkeeping of the broker)
static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// running maximum + histogram with eight buckets per octave (1 us .. 2^26 us)
struct UsHistogram {
  uint64_t n = 0, bucket[8 * 27] = {};
  double max = 0., sum = 0.;
  void add(double us) {
    ++n;
    sum += us;
    max = std::max(max, us);
    const int i = us <= 1. ? 0 : std::min<int>(8 * 27 - 1, (int)(8. * std::log2(us)));
    ++bucket[i];
  }
  double quantile(double q) const {                  // upper edge of the bucket that holds it
    if (!n) return 0.;
    uint64_t need = (uint64_t)std::ceil(q * (double)n), seen = 0;
    for (int i = 0; i < 8 * 27; ++i) {
      seen += bucket[i];
      if (seen >= need) return std::min(max, std::exp2((i + 1) / 8.));
    }
    return max;
  }
};

// staging of one kind of unit (FFT frames or filter-bank blocks): pinned host + device buffers
struct BrokerStage {
  float* h[2] = {nullptr, nullptr};         // [launch pair][stage samples][channels]
  DevBuf d[2];
  size_t samples = 0;
  int alloc(size_t n_sessions, size_t stage_samples, int channels) {
    samples = stage_samples;
    const size_t bytes = n_sessions * stage_samples * channels * sizeof(float);
    for (int p = 0; p < 2; ++p) {
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&h[p]), bytes, hipHostMallocDefault));
      HIP_TRY(d[p].reserve(bytes));
    }
    return PEAQ_OK;
  }
  void release() {
    for (int p = 0; p < 2; ++p) {
      if (h[p]) (void)hipHostFree(h[p]);
      h[p] = nullptr;
      d[p].release();
    }
  }
};

// what one session contributes to a tick: decided under the slot lock in the tick's serial scan,
// copied into the staging buffers afterwards (by the staging threads, several sessions at a time)
struct BrokerJob {
  int sid = 0;
  bool fft = false, fb = false;
  unsigned fft_idx = 0, fb_idx = 0;
  uint64_t fft_from[2] = {0, 0}, fft_n[2] = {0, 0};
  uint64_t fb_from[2] = {0, 0}, fb_n[2] = {0, 0};
};

// A few helper threads for the tick's one heavy host-side step, the copy of every session's new samples
// into the pinned staging buffers (147 KB per stereo session and tick: with 1024 sessions one thread
// spends 12 ms per tick on it).  run(n, fn) calls fn(i) for i in [0, n) on the helpers and the caller.
class StagePool {
 public:
  explicit StagePool(unsigned helpers) {
    for (unsigned i = 0; i < helpers; ++i) threads_.emplace_back([this] { loop(); });
  }
  ~StagePool() {
    {
      std::lock_guard<std::mutex> l(mu_);
      stop_ = true;
    }
    cv_start_.notify_all();
    for (auto& t : threads_) t.join();
  }
  template <typename F>
  void run(size_t n, F&& fn) {
    if (threads_.empty() || n < 32) {
      for (size_t i = 0; i < n; ++i) fn(i);
      return;
    }
    std::function<void(size_t)> f = fn;
    {
      std::lock_guard<std::mutex> l(mu_);
      fn_ = &f;
      n_ = n;
      next_.store(0);
      busy_ = (unsigned)threads_.size();
      ++epoch_;
    }
    cv_start_.notify_all();
    work();
    std::unique_lock<std::mutex> l(mu_);
    cv_done_.wait(l, [this] { return busy_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      const size_t i0 = next_.fetch_add(8);
      if (i0 >= n_) return;
      for (size_t i = i0; i < std::min(n_, i0 + 8); ++i) (*fn_)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(mu_);
        cv_start_.wait(l, [&] { return stop_ || epoch_ != seen; });
        if (stop_) return;
        seen = epoch_;
      }
      work();
      {
        std::lock_guard<std::mutex> l(mu_);
        if (--busy_ == 0) cv_done_.notify_one();
      }
    }
  }
  std::vector<std::thread> threads_;
  std::mutex mu_;
  std::condition_variable cv_start_, cv_done_;
  const std::function<void(size_t)>* fn_ = nullptr;
  size_t n_ = 0;
  std::atomic<size_t> next_{0};
  unsigned busy_ = 0;
  uint64_t epoch_ = 0;
  bool stop_ = false;
};
}  // namespace

struct peaq_broker {
  peaq_ctx* ctx = nullptr;
  int advanced = 0, channels = 1;
  double level_db = 92.;
  Settings cfg;                     // the context's settings when the broker was created
  int max_sessions = 0;
  std::vector<BrokerSlot*> slots;
  std::mutex tick_mu;               // one tick at a time; guards everything below
  hipStream_t stream = nullptr;
  hipEvent_t staged = nullptr;
  bool staged_pending = false;
  BrokerStage fft, fbs;
  uint32_t* h_meta = nullptr;       // pinned: 5 rows of max_sessions (n_ref, n_test, frame0, nframes, slot) + 2 rows (fb n_ref, n_test)
  FbPairWindow* h_win = nullptr;    // pinned: one per launch pair of the filter-bank launch
  DevBuf d_meta, d_win, records, fb_records, state, fbstate, hp_rows, result;
  std::thread worker;
  std::atomic<bool> running{false};
  unsigned period_us = 0;
  std::atomic<bool> failed{false};  // a tick hit a device error: every later call reports it
  std::mutex err_mu;                // guards worker_error
  std::string worker_error;
  std::mutex cv_mu;                 // pushers blocked by the back-pressure wait here for the next tick
  std::condition_variable tick_cv;
  uint64_t n_ticks = 0, n_launches = 0, n_frames = 0;
  uint32_t max_active = 0;
  // timing (peaq_broker_stats_t): the device part of a tick is known when the next tick (or a read-out) has
  // waited for it, so the waits of its sessions are parked until then
  hipEvent_t t_begin = nullptr, t_end = nullptr;
  UsHistogram h_host, h_device, h_latency;
  std::vector<float> parked_wait_us;
  double parked_host_us = 0.;
  std::vector<BrokerJob> jobs;      // this tick's staging work
  std::unique_ptr<StagePool> stagers;
  // peaq_broker_create_multi: this broker owns no device; it deals its sessions out to one broker per device
  // (session id = shard + n_shards * the shard's own id) and forwards every call
  std::vector<peaq_broker*> shards;
  std::vector<peaq_ctx*> shard_ctx;
  std::vector<int> shard_open;      // open sessions per shard (guarded by tick_mu)
};

static inline bool broker_is_multi(const peaq_broker* b) { return b && !b->shards.empty(); }
// -> the device broker that owns session `sid` of b, and the session's id there
static inline peaq_broker* broker_shard_of(peaq_broker* b, int sid, int* local) {
  const int n = (int)b->shards.size();
  if (sid < 0 || sid >= b->max_sessions) return nullptr;
  *local = sid / n;
  return b->shards[sid % n];
}

// copies nv[p] samples per pad from the slot's FIFOs at pos[] into staging entry `idx`
static void broker_stage_copy(const peaq_broker* b, const BrokerSlot& sl, const BrokerStage& st, unsigned idx,
                              const uint64_t pos[2], const uint64_t nv[2]) {
  const size_t stride = st.samples * b->channels;
  for (int p = 0; p < 2; ++p) {
    const PadFifo& f = sl.pad[p];
    if (nv[p])
      std::memcpy(st.h[p] + idx * stride, f.at(pos[p], b->channels), (size_t)nv[p] * b->channels * sizeof(float));
  }
}

// the previous tick's device work is complete: its device time is known now, and with it the latency of the
// sessions it served (wait for the tick + the tick's host part + its device part)
static void broker_settle_timing(peaq_broker* b) {
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, b->t_begin, b->t_end) != hipSuccess) ms = 0.f;
  const double dev_us = 1e3 * ms;
  b->h_device.add(dev_us);
  b->h_host.add(b->parked_host_us);
  for (float w : b->parked_wait_us) b->h_latency.add((double)w + b->parked_host_us + dev_us);
  b->parked_wait_us.clear();
}

static int broker_tick_locked(peaq_broker* b, unsigned* n_active_out) {
  peaq_ctx* c = b->ctx;
  if (n_active_out) *n_active_out = 0;
  HIP_TRY(hipSetDevice(c->device));
  if (b->staged_pending) {
    HIP_TRY(hipEventSynchronize(b->staged));
    b->staged_pending = false;
    broker_settle_timing(b);
  }
  const double t_tick = now_us();
  const size_t S = (size_t)b->max_sessions;
  uint32_t* m_nref = b->h_meta;
  uint32_t* m_ntest = b->h_meta + S;
  uint32_t* m_f0 = b->h_meta + 2 * S;
  uint32_t* m_nf = b->h_meta + 3 * S;
  uint32_t* m_slot = b->h_meta + 4 * S;
  uint32_t* m_fb_nref = b->h_meta + 5 * S;
  uint32_t* m_fb_ntest = b->h_meta + 6 * S;
  unsigned active = 0, max_nf = 0, fb_active = 0, max_nb = 0;
  uint64_t frames = 0;
  b->jobs.clear();
  for (int sid = 0; sid < b->max_sessions; ++sid) {
    BrokerSlot& sl = *b->slots[sid];
    std::lock_guard<std::mutex> lock(sl.mu);
    if (!sl.open) continue;
    BrokerJob job;
    job.sid = sid;
    // ---- FFT frames: do_processing, else the one zero-padded frame of do_flush -------------------
    {
      const uint64_t left[2] = {sl.pad[0].total - sl.fft_pos[0], sl.pad[1].total - sl.fft_pos[1]};
      const uint64_t av = std::min(left[0], left[1]);
      unsigned nf = 0;
      uint64_t nv[2] = {0, 0}, adv[2] = {0, 0};
      if (av >= (uint64_t)kFrame) {
        nf = static_cast<unsigned>(std::min<uint64_t>((av - kFrame) / kHop + 1, kBrokerMaxFrames));
        nv[0] = nv[1] = (uint64_t)(nf - 1) * kHop + kFrame;
        adv[0] = adv[1] = (uint64_t)nf * kHop;
      } else if (sl.flush_requested && !sl.fft_flushed) {
        sl.fft_flushed = true;
        if (left[0] || left[1]) {
          nf = 1;
          nv[0] = adv[0] = std::min<uint64_t>(left[0], kFrame);
          nv[1] = adv[1] = std::min<uint64_t>(left[1], kFrame);
        }
      }
      if (nf) {
        job.fft = true;
        job.fft_idx = active;
        for (int p = 0; p < 2; ++p) {
          job.fft_from[p] = sl.fft_pos[p];
          job.fft_n[p] = nv[p];
        }
        m_nref[active] = static_cast<uint32_t>(nv[0]);
        m_ntest[active] = static_cast<uint32_t>(nv[1]);
        m_f0[active] = sl.frames_done;
        m_nf[active] = nf;
        m_slot[active] = static_cast<uint32_t>(sid);
        sl.fft_pos[0] += adv[0];
        sl.fft_pos[1] += adv[1];
        sl.frames_done += nf;
        frames += nf;
        max_nf = std::max(max_nf, nf);
        ++active;
      }
    }
    // ---- filter-bank blocks (advanced): whole blocks, else the zero-padded block of the flush ----
    if (b->advanced) {
      const uint64_t left[2] = {sl.pad[0].total - sl.fb_pos[0], sl.pad[1].total - sl.fb_pos[1]};
      const uint64_t av = std::min(left[0], left[1]);
      unsigned nb = 0;
      uint64_t nv[2] = {0, 0};
      if (av >= (uint64_t)kFbFrame) {
        nb = static_cast<unsigned>(std::min<uint64_t>(av / kFbFrame, kBrokerMaxBlocks));
        nv[0] = nv[1] = (uint64_t)nb * kFbFrame;
      } else if (sl.flush_requested && !sl.fb_flushed) {
        sl.fb_flushed = true;
        if (left[0] || left[1]) {
          nb = 1;
          nv[0] = std::min<uint64_t>(left[0], kFbFrame);
          nv[1] = std::min<uint64_t>(left[1], kFbFrame);
        }
      }
      if (nb) {
        job.fb = true;
        job.fb_idx = fb_active;
        for (int p = 0; p < 2; ++p) {
          job.fb_from[p] = sl.fb_pos[p];
          job.fb_n[p] = nv[p];
        }
        m_fb_nref[fb_active] = static_cast<uint32_t>(nv[0]);
        m_fb_ntest[fb_active] = static_cast<uint32_t>(nv[1]);
        b->h_win[fb_active] = FbPairWindow{sl.blocks_done, nb, sl.fb_prev_blocks, static_cast<uint32_t>(sid)};
        sl.fb_pos[0] += nv[0];
        sl.fb_pos[1] += nv[1];
        sl.blocks_done += nb;
        sl.fb_prev_blocks = nb;
        max_nb = std::max(max_nb, nb);
        ++fb_active;
      }
    }
    if (sl.flush_requested && sl.fft_flushed && (!b->advanced || sl.fb_flushed))
      sl.flush_requested = sl.fft_flushed = sl.fb_flushed = false;
    if (job.fft || job.fb) {
      b->jobs.push_back(job);
      if (sl.t_ready_us >= 0.) b->parked_wait_us.push_back((float)(t_tick - sl.t_ready_us));
      // more whole units left behind (the per-tick cap)?  They have been waiting since now at the latest.
      const uint64_t av = std::min(sl.pad[0].total - sl.fft_pos[0], sl.pad[1].total - sl.fft_pos[1]);
      const uint64_t avb = b->advanced ? std::min(sl.pad[0].total - sl.fb_pos[0], sl.pad[1].total - sl.fb_pos[1]) : 0;
      sl.t_ready_us = (av >= (uint64_t)kFrame || avb >= (uint64_t)kFbFrame || sl.flush_requested) ? t_tick : -1.;
    }
  }
  ++b->n_ticks;
  if (!active && !fb_active) return PEAQ_OK;
  // ---- the sessions' new samples into the pinned staging buffers; then drop what both consumers are done
  // with.  (Only ticks consume, and ticks are serialised: the positions recorded above stay valid; a
  // concurrent push may reallocate a FIFO, hence the slot lock around each copy.) ----------------------
  b->stagers->run(b->jobs.size(), [b](size_t i) {
    const BrokerJob& j = b->jobs[i];
    BrokerSlot& sl = *b->slots[j.sid];
    std::lock_guard<std::mutex> lock(sl.mu);
    if (j.fft) broker_stage_copy(b, sl, b->fft, j.fft_idx, j.fft_from, j.fft_n);
    if (j.fb) broker_stage_copy(b, sl, b->fbs, j.fb_idx, j.fb_from, j.fb_n);
    for (int p = 0; p < 2; ++p) {
      const uint64_t keep_from = b->advanced ? std::min(sl.fft_pos[p], sl.fb_pos[p]) : sl.fft_pos[p];
      sl.pad[p].drop_until(keep_from, b->channels);
    }
  });
  HIP_TRY(hipEventRecord(b->t_begin, b->stream));
  HIP_TRY(hipMemcpyAsync(b->d_meta.p, b->h_meta, 7 * S * sizeof(uint32_t), hipMemcpyHostToDevice, b->stream));
  const uint32_t* d_meta = b->d_meta.as<uint32_t>();
  if (active) {
    const size_t stride = b->fft.samples * b->channels;
    for (int p = 0; p < 2; ++p)
      HIP_TRY(hipMemcpyAsync(b->fft.d[p].p, b->fft.h[p], active * stride * sizeof(float), hipMemcpyHostToDevice,
                             b->stream));
    FrontendArgs fa{};
    fa.cfg = b->cfg;
    fa.ref = b->fft.d[0].as<float>();
    fa.test = b->fft.d[1].as<float>();
    fa.pair_stride = b->fft.samples;
    fa.n_ref = d_meta;
    fa.n_test = d_meta + S;
    fa.pair_frame0 = d_meta + 2 * S;
    fa.pair_nframes = d_meta + 3 * S;
    fa.channels = b->channels;
    fa.frames_per_launch = max_nf;
    fa.level_factor = fft_level_factor(b->level_db);
    fa.common = c->d_common;
    fa.bands = b->advanced ? c->d_bands55 : c->d_bands109;
    fa.records = b->records.as<double>();
    HIP_TRY(launch_frontend(b->advanced ? 55 : 109, fa, active, b->stream));
    BackendArgs ba{};
    ba.cfg = b->cfg;
    ba.records = fa.records;
    ba.frames_per_launch = max_nf;
    ba.channels = b->channels;
    ba.advanced = b->advanced;
    ba.bands = fa.bands;
    ba.common = c->d_common;
    ba.state = b->state.as<PairState>();
    ba.pair_frame0 = fa.pair_frame0;
    ba.pair_nframes = fa.pair_nframes;
    ba.pair_slot = d_meta + 4 * S;
    HIP_TRY(launch_backend(ba, active, b->stream));
  }
  if (fb_active) {
    const size_t stride = b->fbs.samples * b->channels;
    for (int p = 0; p < 2; ++p)
      HIP_TRY(hipMemcpyAsync(b->fbs.d[p].p, b->fbs.h[p], fb_active * stride * sizeof(float), hipMemcpyHostToDevice,
                             b->stream));
    HIP_TRY(hipMemcpyAsync(b->d_win.p, b->h_win, fb_active * sizeof(FbPairWindow), hipMemcpyHostToDevice, b->stream));
    FbFrontArgs ff{};
    ff.cfg = b->cfg;
    ff.fir_fp64 = c->fir_fp64;
    ff.ref = b->fbs.d[0].as<float>();
    ff.test = b->fbs.d[1].as<float>();
    ff.pair_stride = b->fbs.samples;
    ff.n_ref = d_meta + 5 * S;
    ff.n_test = d_meta + 6 * S;
    ff.channels = b->channels;
    ff.blocks_per_launch = max_nb;
    ff.level_factor = fb_level_factor(b->level_db);
    set_fir_scale(ff);
    ff.bands = c->d_bands40;
    ff.fb = c->d_fb;
    ff.fbstate = b->fbstate.as<FbSignalState>();
    ff.hp_scratch = b->hp_rows.as<double>();
    ff.hp_row_stride = kBrokerRowStride;
    ff.records = b->fb_records.as<double>();
    ff.windows = b->d_win.as<FbPairWindow>();
    HIP_TRY(launch_fb_frontend(ff, fb_active, b->stream));
    FbBackendArgs fbk{};
    fbk.cfg = b->cfg;
    fbk.records = ff.records;
    fbk.blocks_per_launch = max_nb;
    fbk.channels = b->channels;
    fbk.bands = c->d_bands40;
  fbk.common = c->d_common;
    fbk.state = b->state.as<PairState>();
    fbk.windows = ff.windows;
    HIP_TRY(launch_fb_backend(fbk, fb_active, b->stream));
  }
  HIP_TRY(hipEventRecord(b->t_end, b->stream));
  HIP_TRY(hipEventRecord(b->staged, b->stream));
  b->staged_pending = true;
  b->parked_host_us = now_us() - t_tick;
  ++b->n_launches;
  b->n_frames += frames;
  b->max_active = std::max(b->max_active, std::max(active, fb_active));
  if (n_active_out) *n_active_out = std::max(active, fb_active);
  return PEAQ_OK;
}

// one tick under tick_mu; a failure is remembered and stops the broker for good
static int broker_tick_checked(peaq_broker* b, unsigned* n_active) {
  if (b->failed.load()) {
    std::lock_guard<std::mutex> e(b->err_mu);
    return fail(PEAQ_ERR_DEVICE, "broker stopped after a device error: " + b->worker_error);
  }
  const int rc = broker_tick_locked(b, n_active);
  if (rc != PEAQ_OK) {
    std::lock_guard<std::mutex> e(b->err_mu);
    b->worker_error = g_err;
    b->failed.store(true);
  }
  return rc;
}

static int broker_failed(peaq_broker* b, const char* who) {
  std::lock_guard<std::mutex> e(b->err_mu);
  return fail(PEAQ_ERR_DEVICE, std::string(who) + ": broker stopped after a device error: " + b->worker_error);
}

extern "C" int peaq_broker_create(peaq_ctx* c, int advanced, int channels, double level_db, int max_sessions,
                                  peaq_broker** out) {
  if (!c || !out) return fail(PEAQ_ERR_ARG, "peaq_broker_create: NULL argument");
  *out = nullptr;
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_broker_create: channels must be 1 or 2");
  if (!(level_db >= 0. && level_db <= 130.))
    return fail(PEAQ_ERR_ARG, "peaq_broker_create: playback level outside 0..130 dB (gstpeaq.c:275-281)");
  if (max_sessions < 1 || max_sessions > 65536) return fail(PEAQ_ERR_ARG, "peaq_broker_create: max_sessions 1..65536");
  HIP_TRY(hipSetDevice(c->device));
  peaq_broker* b = new (std::nothrow) peaq_broker;
  if (!b) return fail(PEAQ_ERR_NOMEM, "out of host memory");
  b->ctx = c;
  b->cfg = c->settings;
  b->advanced = advanced ? 1 : 0;
  b->channels = channels;
  b->level_db = level_db;
  b->max_sessions = max_sessions;
  {
    // staging helpers beside the ticking thread: PEAQ_AMD_BROKER_STAGERS (default 3, 0 = none); idle
    // unless a tick has at least 32 sessions' samples to copy
    const char* e = std::getenv("PEAQ_AMD_BROKER_STAGERS");
    const long want = e && *e ? std::strtol(e, nullptr, 10) : 3;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    b->stagers.reset(new StagePool(max_sessions >= 32 ? (unsigned)std::min<long>(std::max<long>(want, 0), hw - 1) : 0));
  }
  b->slots.reserve(max_sessions);
  for (int i = 0; i < max_sessions; ++i) b->slots.push_back(new BrokerSlot);
  const size_t S = (size_t)max_sessions;
  int rc = [&]() -> int {
    int r = b->fft.alloc(S, kBrokerStageSamples, channels);
    if (r != PEAQ_OK) return r;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_meta), 7 * S * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(b->d_meta.reserve(7 * S * sizeof(uint32_t)));
    HIP_TRY(b->records.reserve(S * kBrokerMaxFrames * channels * kRecDoubles * sizeof(double)));
    HIP_TRY(b->state.reserve(S * sizeof(PairState)));
    HIP_TRY(b->result.reserve(sizeof(ResultRecord)));
    HIP_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&b->staged, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&b->t_begin));
    HIP_TRY(hipEventCreate(&b->t_end));
    HIP_TRY(launch_state_init(b->state.as<PairState>(), b->advanced, max_sessions, b->stream));
    if (b->advanced) {
      r = b->fbs.alloc(S, kBrokerFbStageSamples, channels);
      if (r != PEAQ_OK) return r;
      const size_t n_signals = S * channels * 2;
      HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&b->h_win), S * sizeof(FbPairWindow), hipHostMallocDefault));
      HIP_TRY(b->d_win.reserve(S * sizeof(FbPairWindow)));
      HIP_TRY(b->fb_records.reserve(S * kBrokerMaxBlocks * channels * kFbRecDoubles * sizeof(double)));
      HIP_TRY(b->fbstate.reserve(n_signals * sizeof(FbSignalState)));
      HIP_TRY(hipMemsetAsync(b->fbstate.p, 0, n_signals * sizeof(FbSignalState), b->stream));
      HIP_TRY(b->hp_rows.reserve(n_signals * kBrokerRowStride * sizeof(double)));
    }
    return PEAQ_OK;
  }();
  if (rc != PEAQ_OK) {
    peaq_broker_destroy(b);
    return rc;
  }
  *out = b;
  return PEAQ_OK;
}

extern "C" int peaq_broker_create_multi(const int* devices, int n_devices, int advanced, int channels, double level_db,
                                        int max_sessions, const peaq_settings* settings, int fir_mode, peaq_broker** out) {
  if (!devices || !out) return fail(PEAQ_ERR_ARG, "peaq_broker_create_multi: NULL argument");
  *out = nullptr;
  if (n_devices < 1 || n_devices > 64) return fail(PEAQ_ERR_ARG, "peaq_broker_create_multi: 1..64 devices");
  if (max_sessions < n_devices) return fail(PEAQ_ERR_ARG, "peaq_broker_create_multi: fewer sessions than devices");
  peaq_broker* b = new (std::nothrow) peaq_broker;
  if (!b) return fail(PEAQ_ERR_NOMEM, "out of host memory");
  b->advanced = advanced ? 1 : 0;
  b->channels = channels;
  b->level_db = level_db;
  const int per_shard = (max_sessions + n_devices - 1) / n_devices;
  b->max_sessions = per_shard * n_devices;
  for (int i = 0; i < n_devices; ++i) {
    peaq_ctx* c = nullptr;
    int rc = peaq_ctx_create(devices[i], &c);
    if (rc == PEAQ_OK && settings) rc = peaq_ctx_set_settings(c, settings);
    if (rc == PEAQ_OK && fir_mode >= 0) rc = peaq_ctx_set_fir_mode(c, fir_mode);
    peaq_broker* sh = nullptr;
    if (rc == PEAQ_OK) rc = peaq_broker_create(c, advanced, channels, level_db, per_shard, &sh);
    if (rc != PEAQ_OK) {
      const std::string msg = g_err;
      if (c) peaq_ctx_destroy(c);
      peaq_broker_destroy(b);
      return fail(rc, "peaq_broker_create_multi: device " + std::to_string(devices[i]) + ": " + msg);
    }
    b->shards.push_back(sh);
    b->shard_ctx.push_back(c);
    b->shard_open.push_back(0);
  }
  *out = b;
  return PEAQ_OK;
}

extern "C" int peaq_broker_devices(const peaq_broker* b) { return !b ? 0 : broker_is_multi(b) ? (int)b->shards.size() : 1; }
extern "C" size_t peaq_broker_stats_size(void) { return sizeof(peaq_broker_stats_t); }

extern "C" int peaq_broker_stop(peaq_broker* b) {
  if (!b) return fail(PEAQ_ERR_ARG, "peaq_broker_stop: broker is NULL");
  if (broker_is_multi(b)) {
    int rc = PEAQ_OK;
    for (peaq_broker* sh : b->shards) {
      const int r = peaq_broker_stop(sh);
      if (r != PEAQ_OK) rc = r;
    }
    return rc;
  }
  if (b->running.exchange(false) && b->worker.joinable()) b->worker.join();
  b->tick_cv.notify_all();          // blocked pushers go on ticking inline
  return PEAQ_OK;
}

extern "C" void peaq_broker_destroy(peaq_broker* b) {
  if (!b) return;
  if (broker_is_multi(b) || !b->ctx) {               // (no context: a multi broker whose creation failed half way)
    for (peaq_broker* sh : b->shards) peaq_broker_destroy(sh);
    for (peaq_ctx* c : b->shard_ctx) peaq_ctx_destroy(c);
    delete b;
    return;
  }
  (void)peaq_broker_stop(b);
  (void)hipSetDevice(b->ctx->device);
  if (b->stream) (void)hipStreamSynchronize(b->stream);
  b->fft.release();
  b->fbs.release();
  if (b->h_meta) (void)hipHostFree(b->h_meta);
  if (b->h_win) (void)hipHostFree(b->h_win);
  b->d_meta.release();
  b->d_win.release();
  b->records.release();
  b->fb_records.release();
  b->state.release();
  b->fbstate.release();
  b->hp_rows.release();
  b->result.release();
  if (b->staged) (void)hipEventDestroy(b->staged);
  if (b->t_begin) (void)hipEventDestroy(b->t_begin);
  if (b->t_end) (void)hipEventDestroy(b->t_end);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  for (BrokerSlot* s : b->slots) delete s;
  delete b;
}

extern "C" int peaq_broker_open(peaq_broker* b, int* session_id) {
  if (!b || !session_id) return fail(PEAQ_ERR_ARG, "peaq_broker_open: NULL argument");
  if (broker_is_multi(b)) {                            // the device with the fewest open sessions takes it
    std::lock_guard<std::mutex> place(b->tick_mu);
    const int n = (int)b->shards.size();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return b->shard_open[x] < b->shard_open[y]; });
    for (int i : order) {
      int local = -1;
      if (peaq_broker_open(b->shards[i], &local) == PEAQ_OK) {
        ++b->shard_open[i];
        *session_id = i + n * local;
        return PEAQ_OK;
      }
    }
    return fail(PEAQ_ERR_STATE, "peaq_broker_open: all session slots are in use");
  }
  std::lock_guard<std::mutex> tick(b->tick_mu);
  for (int sid = 0; sid < b->max_sessions; ++sid) {
    BrokerSlot& sl = *b->slots[sid];
    std::lock_guard<std::mutex> lock(sl.mu);
    if (sl.open) continue;
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(launch_state_init(b->state.as<PairState>() + sid, b->advanced, 1, b->stream));
    if (b->advanced)
      HIP_TRY(hipMemsetAsync(b->fbstate.as<FbSignalState>() + (size_t)sid * b->channels * 2, 0,
                             (size_t)b->channels * 2 * sizeof(FbSignalState), b->stream));
    sl.open = true;
    sl.flush_requested = sl.fft_flushed = sl.fb_flushed = false;
    sl.pad[0] = PadFifo();
    sl.pad[1] = PadFifo();
    sl.fft_pos[0] = sl.fft_pos[1] = sl.fb_pos[0] = sl.fb_pos[1] = 0;
    sl.frames_done = sl.blocks_done = sl.fb_prev_blocks = 0;
    sl.t_ready_us = -1.;
    *session_id = sid;
    return PEAQ_OK;
  }
  return fail(PEAQ_ERR_STATE, "peaq_broker_open: all session slots are in use");
}

static BrokerSlot* broker_slot(peaq_broker* b, int sid) {
  if (!b || sid < 0 || sid >= b->max_sessions) return nullptr;
  return b->slots[sid];
}

extern "C" int peaq_broker_close(peaq_broker* b, int session_id) {
  if (broker_is_multi(b)) {
    int local;
    peaq_broker* sh = broker_shard_of(b, session_id, &local);
    if (!sh) return fail(PEAQ_ERR_ARG, "peaq_broker_close: bad session id");
    const int rc = peaq_broker_close(sh, local);
    if (rc == PEAQ_OK) {
      std::lock_guard<std::mutex> place(b->tick_mu);
      --b->shard_open[session_id % (int)b->shards.size()];
    }
    return rc;
  }
  BrokerSlot* sl = broker_slot(b, session_id);
  if (!sl) return fail(PEAQ_ERR_ARG, "peaq_broker_close: bad broker or session id");
  std::lock_guard<std::mutex> tick(b->tick_mu);
  std::lock_guard<std::mutex> lock(sl->mu);
  if (!sl->open) return fail(PEAQ_ERR_STATE, "peaq_broker_close: session is not open");
  sl->open = false;
  sl->pad[0] = PadFifo();
  sl->pad[1] = PadFifo();
  return PEAQ_OK;
}

// pad_chain (gstpeaq.c:613-640): only queues; the device work happens on the next tick
extern "C" int peaq_broker_push(peaq_broker* b, int session_id, int pad, const float* data, size_t n) {
  if (broker_is_multi(b)) {
    int local;
    peaq_broker* sh = broker_shard_of(b, session_id, &local);
    return sh ? peaq_broker_push(sh, local, pad, data, n) : fail(PEAQ_ERR_ARG, "peaq_broker_push: bad session id");
  }
  BrokerSlot* sl = broker_slot(b, session_id);
  if (!sl) return fail(PEAQ_ERR_ARG, "peaq_broker_push: bad broker or session id");
  if (pad != 0 && pad != 1) return fail(PEAQ_ERR_ARG, "peaq_broker_push: pad must be 0 (ref) or 1 (test)");
  if (b->failed.load()) return broker_failed(b, "peaq_broker_push");
  if (n == 0) return PEAQ_OK;
  if (!data) return fail(PEAQ_ERR_ARG, "peaq_broker_push: data is NULL");
  auto backlog = [&]() {
    std::lock_guard<std::mutex> lock(sl->mu);
    uint64_t r = std::min(sl->pad[0].total - sl->fft_pos[0], sl->pad[1].total - sl->fft_pos[1]);
    if (b->advanced) r = std::max(r, std::min(sl->pad[0].total - sl->fb_pos[0], sl->pad[1].total - sl->fb_pos[1]));
    return r;
  };
  {
    std::lock_guard<std::mutex> lock(sl->mu);
    if (!sl->open) return fail(PEAQ_ERR_STATE, "peaq_broker_push: session is not open");
    PadFifo& f = sl->pad[pad];
    try {
      f.append(data, n * b->channels);
    } catch (const std::bad_alloc&) {
      return fail(PEAQ_ERR_NOMEM, "out of host memory");
    }
    f.total += n;
    if (sl->t_ready_us < 0.) {
      const uint64_t av = std::min(sl->pad[0].total - sl->fft_pos[0], sl->pad[1].total - sl->fft_pos[1]);
      const uint64_t avb = b->advanced ? std::min(sl->pad[0].total - sl->fb_pos[0], sl->pad[1].total - sl->fb_pos[1]) : 0;
      if (av >= (uint64_t)kFrame || avb >= (uint64_t)kFbFrame) sl->t_ready_us = now_us();
    }
  }
  // back-pressure (the reference processes inside pad_chain, so its caller can never run ahead):
  // wait for the tick thread, or tick right here when there is none
  while (backlog() > kBrokerBacklog) {
    if (b->failed.load()) return broker_failed(b, "peaq_broker_push");
    if (b->running.load()) {
      std::unique_lock<std::mutex> w(b->cv_mu);
      b->tick_cv.wait_for(w, std::chrono::milliseconds(20));
    } else {
      std::lock_guard<std::mutex> tick(b->tick_mu);
      const int rc = broker_tick_checked(b, nullptr);
      if (rc != PEAQ_OK) return rc;
    }
  }
  return PEAQ_OK;
}

extern "C" int peaq_broker_flush(peaq_broker* b, int session_id) {
  if (broker_is_multi(b)) {
    int local;
    peaq_broker* sh = broker_shard_of(b, session_id, &local);
    return sh ? peaq_broker_flush(sh, local) : fail(PEAQ_ERR_ARG, "peaq_broker_flush: bad session id");
  }
  BrokerSlot* sl = broker_slot(b, session_id);
  if (!sl) return fail(PEAQ_ERR_ARG, "peaq_broker_flush: bad broker or session id");
  if (b->failed.load()) return broker_failed(b, "peaq_broker_flush");
  std::lock_guard<std::mutex> lock(sl->mu);
  if (!sl->open) return fail(PEAQ_ERR_STATE, "peaq_broker_flush: session is not open");
  sl->flush_requested = true;
  sl->fft_flushed = sl->fb_flushed = false;
  if (sl->t_ready_us < 0.) sl->t_ready_us = now_us();
  return PEAQ_OK;
}

extern "C" int peaq_broker_tick(peaq_broker* b, unsigned* n_active) {
  if (!b) return fail(PEAQ_ERR_ARG, "peaq_broker_tick: broker is NULL");
  if (broker_is_multi(b)) {                            // every device's launch of this tick (they run side by side)
    unsigned total = 0;
    for (peaq_broker* sh : b->shards) {
      unsigned n = 0;
      const int rc = peaq_broker_tick(sh, &n);
      if (rc != PEAQ_OK) return rc;
      total += n;
    }
    if (n_active) *n_active = total;
    return PEAQ_OK;
  }
  std::lock_guard<std::mutex> tick(b->tick_mu);
  return broker_tick_checked(b, n_active);
}

// true while the session has whole frames / blocks (or a requested flush) not yet launched
static bool broker_slot_busy(const peaq_broker* b, BrokerSlot* sl) {
  std::lock_guard<std::mutex> lock(sl->mu);
  const uint64_t av = std::min(sl->pad[0].total - sl->fft_pos[0], sl->pad[1].total - sl->fft_pos[1]);
  if (av >= (uint64_t)kFrame || sl->flush_requested) return true;
  if (b->advanced) {
    const uint64_t avb = std::min(sl->pad[0].total - sl->fb_pos[0], sl->pad[1].total - sl->fb_pos[1]);
    if (avb >= (uint64_t)kFbFrame) return true;
  }
  return false;
}

extern "C" int peaq_broker_results(peaq_broker* b, int session_id, peaq_result* out) {
  if (broker_is_multi(b)) {
    int local;
    peaq_broker* sh = broker_shard_of(b, session_id, &local);
    return sh ? peaq_broker_results(sh, local, out) : fail(PEAQ_ERR_ARG, "peaq_broker_results: bad session id");
  }
  BrokerSlot* sl = broker_slot(b, session_id);
  if (!sl || !out) return fail(PEAQ_ERR_ARG, "peaq_broker_results: bad broker, session id or out");
  std::lock_guard<std::mutex> tick(b->tick_mu);
  {
    std::lock_guard<std::mutex> lock(sl->mu);
    if (!sl->open) return fail(PEAQ_ERR_STATE, "peaq_broker_results: session is not open");
  }
  if (b->failed.load()) return broker_failed(b, "peaq_broker_results");
  while (broker_slot_busy(b, sl)) {
    const int rc = broker_tick_checked(b, nullptr);
    if (rc != PEAQ_OK) return rc;
  }
  HIP_TRY(hipSetDevice(b->ctx->device));
  HIP_TRY(launch_finalize(b->state.as<PairState>() + session_id, b->advanced, b->channels, 1,
                          b->result.as<ResultRecord>(), b->stream, b->cfg));
  HIP_TRY(hipMemcpyAsync(out, b->result.p, sizeof(peaq_result), hipMemcpyDeviceToHost, b->stream));
  HIP_TRY(hipStreamSynchronize(b->stream));
  return PEAQ_OK;
}

extern "C" int peaq_broker_start(peaq_broker* b, unsigned period_us) {
  if (!b) return fail(PEAQ_ERR_ARG, "peaq_broker_start: broker is NULL");
  if (broker_is_multi(b)) {                            // one tick thread per device
    for (peaq_broker* sh : b->shards) {
      const int rc = peaq_broker_start(sh, period_us);
      if (rc != PEAQ_OK) {
        (void)peaq_broker_stop(b);
        return rc;
      }
    }
    return PEAQ_OK;
  }
  if (b->running.exchange(true)) return fail(PEAQ_ERR_STATE, "peaq_broker_start: already running");
  b->period_us = period_us ? period_us : 2000;
  b->worker = std::thread([b]() {
    // fixed cadence: whatever arrived during one period shares one launch
    auto next = std::chrono::steady_clock::now();
    while (b->running.load()) {
      int rc;
      {
        std::lock_guard<std::mutex> tick(b->tick_mu);
        rc = broker_tick_checked(b, nullptr);
      }
      b->tick_cv.notify_all();
      if (rc != PEAQ_OK) break;
      next += std::chrono::microseconds(b->period_us);
      const auto now = std::chrono::steady_clock::now();
      if (next < now) next = now;
      std::this_thread::sleep_until(next);
    }
  });
  return PEAQ_OK;
}

extern "C" int peaq_broker_stats(peaq_broker* b, peaq_broker_stats_t* out) {
  if (!b || !out) return fail(PEAQ_ERR_ARG, "peaq_broker_stats: NULL argument");
  if (broker_is_multi(b)) {
    // counts add up over the devices (max_active: the most sessions the node served in one tick period, each
    // device's own maximum); times: the worst device's maximum and 99th percentile, means weighted by their samples
    peaq_broker_stats_t acc{};
    double host_w = 0., dev_w = 0.;
    for (peaq_broker* sh : b->shards) {
      peaq_broker_stats_t s{};
      const int rc = peaq_broker_stats(sh, &s);
      if (rc != PEAQ_OK) return rc;
      acc.ticks += s.ticks;
      acc.launches += s.launches;
      acc.frames += s.frames;
      acc.max_active += s.max_active;
      acc.worker_failed |= s.worker_failed;
      acc.tick_host_us_max = std::max(acc.tick_host_us_max, s.tick_host_us_max);
      acc.tick_host_us_p99 = std::max(acc.tick_host_us_p99, s.tick_host_us_p99);
      acc.tick_device_us_max = std::max(acc.tick_device_us_max, s.tick_device_us_max);
      acc.tick_device_us_p99 = std::max(acc.tick_device_us_p99, s.tick_device_us_p99);
      acc.latency_us_max = std::max(acc.latency_us_max, s.latency_us_max);
      acc.latency_us_p99 = std::max(acc.latency_us_p99, s.latency_us_p99);
      acc.tick_host_us_mean += s.tick_host_us_mean * (double)s.launches;
      acc.tick_device_us_mean += s.tick_device_us_mean * (double)s.launches;
      host_w += (double)s.launches;
      dev_w += (double)s.launches;
      acc.latency_us_mean += s.latency_us_mean * (double)s.latency_samples;
      acc.latency_samples += s.latency_samples;
    }
    if (host_w > 0.) acc.tick_host_us_mean /= host_w;
    if (dev_w > 0.) acc.tick_device_us_mean /= dev_w;
    if (acc.latency_samples) acc.latency_us_mean /= (double)acc.latency_samples;
    *out = acc;
    return PEAQ_OK;
  }
  std::lock_guard<std::mutex> tick(b->tick_mu);
  out->ticks = b->n_ticks;
  out->launches = b->n_launches;
  out->frames = b->n_frames;
  out->max_active = b->max_active;
  out->worker_failed = b->failed.load() ? 1 : 0;
  if (b->staged_pending && !b->failed.load()) {        // settle the last tick's timing
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipEventSynchronize(b->staged));
    b->staged_pending = false;
    broker_settle_timing(b);
  }
  out->tick_host_us_max = b->h_host.max;
  out->tick_host_us_p99 = b->h_host.quantile(0.99);
  out->tick_host_us_mean = b->h_host.n ? b->h_host.sum / (double)b->h_host.n : 0.;
  out->tick_device_us_mean = b->h_device.n ? b->h_device.sum / (double)b->h_device.n : 0.;
  out->tick_device_us_max = b->h_device.max;
  out->tick_device_us_p99 = b->h_device.quantile(0.99);
  out->latency_us_max = b->h_latency.max;
  out->latency_us_p99 = b->h_latency.quantile(0.99);
  out->latency_us_mean = b->h_latency.n ? b->h_latency.sum / (double)b->h_latency.n : 0.;
  out->latency_samples = b->h_latency.n;
  return PEAQ_OK;
}
