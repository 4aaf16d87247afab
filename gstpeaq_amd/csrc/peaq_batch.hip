// peaq_batch.hip -- the batch driver: N whole (ref, test) pairs resident in device memory (BASELINE.json configs
// 2-4), one pair from host memory, the timing of the last batch, the synthetic workload.
//
// Framing follows the reference element: FFT frames of 2048 samples every 1024
// (do_processing, gstpeaq.c:596-611), filter-bank blocks of 192 every 192, and
// at the end ONE zero-padded frame/block built from whatever is left on either
// side (do_flush, gstpeaq.c:716-745).  Frame f of a pair reads samples
// [1024 f, 1024 f + 2048) of each signal, zero beyond the signal's length.
#include "peaq_host.h"

using namespace peaq;

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
unsigned fb_blocks_per_chunk(int n_pairs, int channels, uint32_t max_blocks) {
  const size_t n_signals = (size_t)n_pairs * channels * 2;
  const size_t per_signal = kFbRowBudget / std::max<size_t>(n_signals, 1) / sizeof(double);
  size_t bc = per_signal > (size_t)kFbRing ? (per_signal - kFbRing) / kFbFrame : 0;
  bc = std::min<size_t>(bc, kFbBlocksPerChunk) / 10 * 10;
  bc = std::max<size_t>(bc, 10);
  return static_cast<unsigned>(std::min<size_t>(bc, (max_blocks + 9) / 10 * 10));
}

unsigned frames_per_chunk(int n_pairs, int channels, uint32_t max_frames) {
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

static int run_filterbank_path(peaq_ctx* c, int channels, double level_db, int n_pairs, const float* d_ref,
                               const float* d_test, size_t pair_stride, const uint32_t* d_nref,
                               const uint32_t* d_ntest, uint32_t n_uniform, const uint32_t* d_nblocks,
                               uint32_t max_blocks, hipStream_t stream, hipEvent_t bank_gate) {
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
      if (chunk == 0 && bank_gate) HIP_TRY(hipStreamWaitEvent(s_bank, bank_gate, 0));   // (batch_run_locked: the FFT path's head)
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
      c->fb_last_bank_begin = e0;
      c->fb_last_bank_end = e1;
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
    const std::string msg = peaq_err_string();
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
  hipEvent_t fb_done = nullptr, forked = nullptr;
  c->fb_last_bank_begin = c->fb_last_bank_end = nullptr;
  const bool with_fb = advanced && max_blocks > 0;
  if (with_fb) {
    // the filter-bank path (its own ear model, accumulators 0, 1, 4) is independent of the FFT
    // path (accumulators 2, 3): it runs on a third stream from here on and joins at the end
    forked = c->next_event();
    if (!forked) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(forked, stream));
    HIP_TRY(hipStreamWaitEvent(c->aux2, forked, 0));
  }
  // (its launches are issued further down, between the FFT path's head and the rest of its chunks)
  auto issue_filterbank_path = [&](hipEvent_t bank_gate) -> int {
    hipStream_t s_fb = PEAQ_DEV_SERIAL_KERNELS ? stream : c->aux2;
    const int rc = run_filterbank_path(c, channels, level_db, n_pairs, d_ref, d_test, pair_stride, d_nref, d_ntest,
                                       n_uniform, d_nblocks, max_blocks, s_fb, bank_gate);
    if (rc != PEAQ_OK) return rc;
    fb_done = c->next_event();
    if (!fb_done) return fail(PEAQ_ERR_DEVICE, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(fb_done, s_fb));
    return PEAQ_OK;
  };

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
  HIP_TRY(c->clk.reserve(2 * sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(c->clk.p, 0, 2 * sizeof(unsigned long long), stream));   // (the first back-end launch waits for this stream)
  ba.clk = c->clk.as<unsigned long long>();

  // Software pipeline over chunks of frames: the front end of chunk i+1 (throughput bound,
  // millions of workgroups) runs on the caller's stream while the back end of chunk i
  // (one workgroup per pair, latency bound) runs on the context's second stream; the
  // per-frame records are double buffered.
  hipEvent_t back_done[2] = {nullptr, nullptr}, head_done = nullptr;
  unsigned chunk = 0;
  // Advanced version: the last chunks of the FFT path are held back until the filter-bank path's last bank launch
  // is through.  An advanced pass ends with that launch's back end alone on the device (one workgroup per pair walking
  // 840 blocks: 12 ms at a fraction of the machine) while all FFT chunks, ready from the start, have long run beside
  // the FIRST bank launch and slowed it by their own time; two chunks of frontend_kernel<55> are about what the tail
  // has room for (profiles/r05_timeline_adv.txt).  PEAQ_AMD_ADV_DEFER=<chunks>[b] (development): other counts; "b" =
  // wait for the last bank launch's begin instead of its end.
  const unsigned n_chunks = fc ? (max_frames + fc - 1) / fc : 0u;   // (a pair of two empty signals has no frames at all)
  static const int defer_env = [] { const char* e = std::getenv("PEAQ_AMD_ADV_DEFER"); return e && *e ? std::atoi(e) : -1; }();
  static const bool defer_begin = [] { const char* e = std::getenv("PEAQ_AMD_ADV_DEFER"); return e && std::strchr(e, 'b'); }();
  const bool fb_piped = with_fb && !PEAQ_DEV_SERIAL_KERNELS;
  const unsigned defer = !fb_piped ? 0u : std::min<unsigned>(defer_env >= 0 ? (unsigned)defer_env : 2u, n_chunks > 4 ? n_chunks - 4 : 0u);
  // ... and with the FP64 bank the FIRST bank launch waits for the front end of all chunks before those (the head).
  // Two workgroups of fb_bank_kernel<MfmaF64> fill a CU's registers and LDS: whatever runs beside it does so in place
  // of one of them, and then neither kernel has the waves to cover its latencies -- beside the first bank launch a chunk
  // of frontend_kernel<55> took 33 .. 39 ms, alone it takes 6.6 (profiles/r05_timeline_adv.txt: before / after).  The
  // first two walks of the high-pass filter, which the bank has to wait for anyway, run beside the head.  Measured:
  // 5.38 M -> 5.47 M; the reduced-precision engine, whose bank kernel leaves room beside it, loses 0.6 % and keeps
  // the old order.  PEAQ_AMD_ADV_HEAD=<chunks> (development): other counts.
  static const int head_env = [] { const char* e = std::getenv("PEAQ_AMD_ADV_HEAD"); return e && *e ? std::atoi(e) : -1; }();
  const unsigned head = !fb_piped ? 0u : std::min<unsigned>(head_env >= 0 ? (unsigned)head_env : (c->fir_fp64 == PEAQ_FIR_F64 ? n_chunks : 0u), n_chunks - defer);
  auto issue_chunk = [&](uint32_t f0) -> int {
    const unsigned nf = std::min<uint32_t>(fc, max_frames - f0);
    if (defer && chunk == n_chunks - defer)
      HIP_TRY(hipStreamWaitEvent(stream, defer_begin ? c->fb_last_bank_begin : c->fb_last_bank_end, 0));
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
    head_done = e1;
    c->spans.push_back({e0, e1, 0});
    c->spans.push_back({e2, e3, 1});
    return PEAQ_OK;
  };
  uint32_t f0 = 0;
  for (; chunk < head; f0 += fc, ++chunk) {
    const int rc = issue_chunk(f0);
    if (rc != PEAQ_OK) return rc;
  }
  if (with_fb) {
    const int rc = issue_filterbank_path(head ? head_done : nullptr);
    if (rc != PEAQ_OK) return rc;
  }
  for (; f0 < max_frames; f0 += fc, ++chunk) {
    const int rc = issue_chunk(f0);
    if (rc != PEAQ_OK) return rc;
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

extern "C" int peaq_batch_last_clock(peaq_ctx* c, double* shader_clock_mhz) {
  if (!c || !shader_clock_mhz) return fail(PEAQ_ERR_ARG, "peaq_batch_last_clock: NULL argument");
  std::lock_guard<std::mutex> lock(c->mu);
  *shader_clock_mhz = 0.;
  if (!c->clk.p) return PEAQ_OK;                     // no batch yet
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipEventSynchronize(c->batch_end));
  unsigned long long h[2] = {0, 0};
  HIP_TRY(hipMemcpy(h, c->clk.p, sizeof h, hipMemcpyDeviceToHost));
  int wall_khz = 100000;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device);
  if (h[1]) *shader_clock_mhz = (double)h[0] / (double)h[1] * (wall_khz * 1e-3);
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
