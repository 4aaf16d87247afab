// peaq_session.hip -- streaming sessions: one per `peaq` element instance / (ref, test) stream.
#include "peaq_host.h"

using namespace peaq;

// ---------------------------------------------------------------------------
// sessions: one per `peaq` element instance
// ---------------------------------------------------------------------------
namespace {


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
