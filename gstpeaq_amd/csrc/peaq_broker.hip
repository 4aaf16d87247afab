// peaq_broker.hip -- the live-pipeline broker: many sessions, one launch per kernel and tick; one or several devices.
#include "peaq_host.h"

using namespace peaq;

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
    b->worker_error = peaq_err_string();
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
      const std::string msg = peaq_err_string();
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

// Test hook (include/peaq_amd.h): shard `shard` of a multi-device broker (0 of a plain one) is put into the state a
// device error during one of its ticks leaves it in -- failed for good, with `message` as the error it keeps.
extern "C" int peaq_debug_broker_fail_shard(peaq_broker* b, int shard, const char* message) {
  if (!b) return fail(PEAQ_ERR_ARG, "peaq_debug_broker_fail_shard: broker is NULL");
  peaq_broker* sh = b;
  if (broker_is_multi(b)) {
    if (shard < 0 || shard >= (int)b->shards.size()) return fail(PEAQ_ERR_ARG, "peaq_debug_broker_fail_shard: no such shard");
    sh = b->shards[shard];
  } else if (shard != 0) {
    return fail(PEAQ_ERR_ARG, "peaq_debug_broker_fail_shard: no such shard");
  }
  std::lock_guard<std::mutex> tick(sh->tick_mu);
  std::lock_guard<std::mutex> e(sh->err_mu);
  sh->worker_error = message ? message : "injected fault";
  sh->failed.store(true);
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
    int err_rc = PEAQ_OK;
    std::string err_msg;
    for (int i : order) {
      int local = -1;
      const int rc = peaq_broker_open(b->shards[i], &local);
      if (rc == PEAQ_OK) {
        ++b->shard_open[i];
        *session_id = i + n * local;
        return PEAQ_OK;
      }
      if (rc != PEAQ_ERR_STATE && err_rc == PEAQ_OK) {   // a device's own failure, not "its slots are in use": keep it
        err_rc = rc;
        err_msg = "device " + std::to_string(b->shards[i]->ctx->device) + ": " + peaq_err_string();
      }
    }
    if (err_rc != PEAQ_OK) return fail(err_rc, "peaq_broker_open: " + err_msg);
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
    // EVERY device is ticked, whatever another one reports: a device that failed must not starve the sessions that
    // live on the others; the first error (with its message) is what the call returns afterwards
    unsigned total = 0;
    int first_rc = PEAQ_OK;
    std::string first_msg;
    for (peaq_broker* sh : b->shards) {
      unsigned n = 0;
      const int rc = peaq_broker_tick(sh, &n);
      if (rc != PEAQ_OK && first_rc == PEAQ_OK) {
        first_rc = rc;
        first_msg = peaq_err_string();
      }
      total += n;
    }
    if (n_active) *n_active = total;
    return first_rc == PEAQ_OK ? PEAQ_OK : fail(first_rc, first_msg);
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
