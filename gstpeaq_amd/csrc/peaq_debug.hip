// peaq_debug.hip -- stage-level entry points for the parity tests: the front end, the filter bank and the pattern
// back end each on their own (include/peaq_amd.h, "stage-level access").
#include "peaq_host.h"

using namespace peaq;

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

// test-facing layout (kPub*) -> the record the kernels exchange: root = (E norm)^(1/10); the E^0.3 vector of the
// input is implied by E (the back end derives both from the root)
static hipError_t upload_public_records(int bands, int channels, int n_frames, const double* host_records, void* d_recs) {
  BandTables t;
  build_fft_band_tables(bands, t);
  std::vector<double> h_rec((size_t)n_frames * channels * kRecDoubles, 0.);
  for (size_t r = 0; r < (size_t)n_frames * channels; ++r) {
    const double* in = host_records + r * kPubDoubles;
    double* out = h_rec.data() + r * kRecDoubles;
    for (int b = 0; b < bands; ++b) {
      out[kRecRootRef + b] = std::pow(in[kPubUnsmRef + b] / t.inv_spread_norm[b], 0.1);
      out[kRecRootTest + b] = std::pow(in[kPubUnsmTest + b] / t.inv_spread_norm[b], 0.1);
    }
    for (int b = 0; b < kBandStride; ++b) out[kRecNoise + b] = in[kPubNoise + b];
    for (int i = 0; i < kRecDoubles - kRecScalars; ++i) out[kRecScalars + i] = in[kPubScalars + i];
  }
  return hipMemcpy(d_recs, h_rec.data(), h_rec.size() * sizeof(double), hipMemcpyHostToDevice);
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
  HIP_TRY(upload_public_records(109, channels, n_frames, host_records, recs.p));
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

// The advanced version's two back ends on their own (fresh state): the filter-bank back end over n_blocks block
// records (from peaq_debug_filterbank) and the 55-band FFT back end over n_frames front-end records (from
// peaq_debug_frontend with 55 bands), each in its debug instantiation -- the MOV values of EVERY block / frame
// before accumulation -- followed by the read-out of the pair's result.
extern "C" int peaq_debug_backend_advanced(peaq_ctx* c, int channels, int n_blocks, const double* host_fb_records,
                                           int n_frames, const double* host_fft_records, double* out_blocks,
                                           double* out_frames, peaq_result* result) {
  if (!c || !host_fb_records || !host_fft_records || !out_blocks || !out_frames)
    return fail(PEAQ_ERR_ARG, "peaq_debug_backend_advanced: NULL argument");
  if (channels != 1 && channels != 2) return fail(PEAQ_ERR_ARG, "peaq_debug_backend_advanced: channels must be 1 or 2");
  if (n_blocks < 1 || n_frames < 1) return fail(PEAQ_ERR_ARG, "peaq_debug_backend_advanced: n_blocks, n_frames must be >= 1");
  HIP_TRY(hipSetDevice(c->device));
  const size_t fbrec_bytes = (size_t)n_blocks * channels * kFbRecDoubles * sizeof(double);
  const size_t fbdbg_bytes = (size_t)n_blocks * channels * kDbgFbDoubles * sizeof(double);
  const size_t rec_bytes = (size_t)n_frames * channels * kRecDoubles * sizeof(double);
  const size_t dbg_bytes = (size_t)n_frames * channels * kDbgDoubles * sizeof(double);
  TmpBuf fbrecs, fbdbg, recs, dbg, st, res;
  HIP_TRY(fbrecs.reserve(fbrec_bytes));
  HIP_TRY(fbdbg.reserve(fbdbg_bytes));
  HIP_TRY(recs.reserve(rec_bytes));
  HIP_TRY(dbg.reserve(dbg_bytes));
  HIP_TRY(st.reserve(sizeof(PairState)));
  HIP_TRY(res.reserve(sizeof(ResultRecord)));
  HIP_TRY(hipMemcpy(fbrecs.p, host_fb_records, fbrec_bytes, hipMemcpyHostToDevice));
  HIP_TRY(upload_public_records(55, channels, n_frames, host_fft_records, recs.p));
  HIP_TRY(hipMemset(fbdbg.p, 0, fbdbg_bytes));
  HIP_TRY(hipMemset(dbg.p, 0, dbg_bytes));
  HIP_TRY(launch_state_init(st.as<PairState>(), 1, 1, nullptr));
  BackendArgs ba{};
  ba.cfg = c->settings;
  ba.records = recs.as<double>();
  ba.frame0 = 0;
  ba.frames_per_launch = n_frames;
  ba.n_frames_uniform = n_frames;
  ba.channels = channels;
  ba.advanced = 1;
  ba.bands = c->d_bands55;
  ba.common = c->d_common;
  ba.state = st.as<PairState>();
  ba.debug = dbg.as<double>();
  HIP_TRY(launch_backend(ba, 1, nullptr));
  FbBackendArgs fbk{};
  fbk.cfg = c->settings;
  fbk.records = fbrecs.as<double>();
  fbk.block0 = 0;
  fbk.blocks_per_launch = n_blocks;
  fbk.n_blocks_uniform = n_blocks;
  fbk.channels = channels;
  fbk.common = c->d_common;
  fbk.bands = c->d_bands40;
  fbk.state = st.as<PairState>();
  fbk.debug = fbdbg.as<double>();
  HIP_TRY(launch_fb_backend(fbk, 1, nullptr));
  HIP_TRY(launch_finalize(st.as<PairState>(), 1, channels, 1, res.as<ResultRecord>(), nullptr, c->settings));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out_blocks, fbdbg.p, fbdbg_bytes, hipMemcpyDeviceToHost));
  {
    std::vector<double> h((size_t)n_frames * channels * kDbgDoubles);
    HIP_TRY(hipMemcpy(h.data(), dbg.p, dbg_bytes, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < (size_t)n_frames * channels; ++r) {
      out_frames[2 * r] = h[r * kDbgDoubles + kDbgMov + 3];       // SegmentalNMR's value (dB)
      out_frames[2 * r + 1] = h[r * kDbgDoubles + kDbgMov + 4];   // the mean band NMR it is the logarithm of
    }
  }
  if (result) HIP_TRY(hipMemcpy(result, res.p, sizeof(peaq_result), hipMemcpyDeviceToHost));
  return PEAQ_OK;
}
