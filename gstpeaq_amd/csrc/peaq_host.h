// peaq_host.h -- what the host-side translation units behind include/peaq_amd.h share: the error convention, the
// framing arithmetic, device buffers, the context, the host-side stand-in for a GstAdapter, and the pieces of the
// batch driver that sessions, brokers and the stage-level entry points call.  Host code only; the kernels are in
// peaq_frontend.hip / peaq_backend.hip / peaq_fb.hip / peaq_synth.hip.
//   peaq_ctx.hip      errors, version, framing, context, settings, calibration
//   peaq_batch.hip    batch driver (peaq_batch_run, peaq_run_pair), timing, synthetic workload
//   peaq_debug.hip    stage-level entry points for the parity tests
//   peaq_session.hip  streaming sessions (one per `peaq` element)
//   peaq_broker.hip   live-pipeline broker (many sessions, one launch per tick), one or several devices
#pragma once
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

// ---------------------------------------------------------------------------
// errors: every entry point returns PEAQ_OK or a negative code; the message of the calling thread's last failure
// is what peaq_last_error() hands out
// ---------------------------------------------------------------------------
std::string& peaq_err_string();
inline int fail(int code, const std::string& msg) {
  peaq_err_string() = msg;
  return code;
}
#define HIP_TRY(expr)                                                                                     \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess)                                                                                 \
      return fail(e_ == hipErrorOutOfMemory ? PEAQ_ERR_NOMEM : PEAQ_ERR_DEVICE,                           \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                     \
  } while (0)

// ---------------------------------------------------------------------------
// framing arithmetic
// ---------------------------------------------------------------------------
// number of frames the element processes for signals of n_ref / n_test samples:
// full frames while BOTH adapters hold `frame` samples, then one flush frame if
// anything is left on either side.
inline uint32_t count_frames(uint64_t n_ref, uint64_t n_test, uint32_t frame, uint32_t hop) {
  const uint64_t n = std::min(n_ref, n_test);
  const uint64_t full = n >= frame ? (n - frame) / hop + 1 : 0;
  const bool left = n_ref > full * hop || n_test > full * hop;
  return static_cast<uint32_t>(full + (left ? 1 : 0));
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
  peaq::CommonTables* d_common = nullptr;
  peaq::BandTables* d_bands109 = nullptr;
  peaq::BandTables* d_bands55 = nullptr;
  peaq::BandTables* d_bands40 = nullptr;
  peaq::FbTables* d_fb = nullptr;
  std::mutex mu;            // serialises batch calls / workspace use
  // batch workspace
  DevBuf records, records2, fb_records, fb_records2, state, fbstate, hp_scratch, hp_scratch2, counts, clk;
  hipStream_t aux = nullptr;   // the back end runs here, overlapped with the next chunk's front end
  hipStream_t aux2 = nullptr;  // advanced: the filter-bank path runs here, beside the FFT path
  hipStream_t aux3 = nullptr, aux4 = nullptr;   // ... its high-pass stage and its back end (3-stage pipeline)
  hipEvent_t batch_begin = nullptr, batch_end = nullptr;
  hipEvent_t fb_last_bank_begin = nullptr, fb_last_bank_end = nullptr;   // of the running batch's filter-bank path (pool events)
  bool batch_pending = false;
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> event_pool;
  size_t events_used = 0;
  unsigned long long* d_prof = nullptr;   // -DPEAQ_FE_PROFILE builds only
  int fir_fp64 = 1;                       // advanced version: arithmetic of the FIR bank (PEAQ_FIR_*; default the reference's FP64)
  peaq::Settings settings;                      // the reference's settings.h switches (peaq_ctx_set_settings)

  hipEvent_t next_event() {
    if (events_used == event_pool.size()) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) return nullptr;
      event_pool.push_back(e);
    }
    return event_pool[events_used++];
  }
};

// ---- batch driver pieces used elsewhere (peaq_batch.hip) --------------------------------------------------
unsigned fb_blocks_per_chunk(int n_pairs, int channels, uint32_t max_blocks);
unsigned frames_per_chunk(int n_pairs, int channels, uint32_t max_frames);
// split-FP16 FIR (FbFrontArgs.fir_fp64 == 2): the power of two that puts the filtered signal's full scale --
// |x| = 1 times the playback-level factor -- between 2^10 and 2^11 of FP16's 65504 (30 dB of headroom for
// samples beyond full scale and for the high-pass filter's overshoot)
inline void set_fir_scale(peaq::FbFrontArgs& ff) {
  const int e = 10 - std::ilogb(ff.level_factor);
  ff.hf_xscale = std::ldexp(1., e);
  ff.hf_xunscale = std::ldexp(1., -e);
}

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
