// peaq_ctx.hip -- errors, version, framing, the device context and its settings (C ABI of include/peaq_amd.h).
#include "peaq_host.h"

using namespace peaq;

static_assert(sizeof(ResultRecord) == sizeof(peaq_result), "result layouts must match");
static_assert(kPubDoubles == PEAQ_DEBUG_RECORD_DOUBLES, "record layouts must match");
static_assert(kDbgDoubles == PEAQ_DEBUG_BACKEND_DOUBLES, "debug layouts must match");
static_assert(kDbgFbDoubles == PEAQ_DEBUG_ADV_BLOCK_DOUBLES, "debug layouts must match");

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
std::string& peaq_err_string() {
  static thread_local std::string err;
  return err;
}

extern "C" const char* peaq_last_error(void) { return peaq_err_string().c_str(); }
extern "C" const char* peaq_version(void) { return "0.2.0 gfx950 (advanced version: FP64 filter bank by default; f16x3 and f32 opt-in)"; }
// host only, no device: the filter-bank tables of the FP64 engine against the reference's plain sums (peaq_tables.cpp)
extern "C" double peaq_debug_fb_tables_selfcheck(void) { return peaq::fb_tables_selfcheck(); }

extern "C" uint32_t peaq_frame_count(uint64_t n_ref, uint64_t n_test, int filter_bank) {
  return filter_bank ? count_frames(n_ref, n_test, kFbFrame, kFbFrame) : count_frames(n_ref, n_test, kFrame, kHop);
}
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
    const std::string msg = peaq_err_string();
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
  c->clk.release();
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
// device calibration: what the GPU clocks at under an FP64 load, and what FP64 rate that gives
// ---------------------------------------------------------------------------
// Two waves per SIMD run eight independent chains of v_fma_f64 each; one lane of every workgroup reads the shader
// clock (s_memtime) and the constant 100 MHz counter around its chain.  The chip clocks to its power budget
// (MI355X_MICROARCH.md, "DVFS give-back"), and PEAQ's FP64-dense kernels sit at that budget: two boxes -- or one box
// at two moments -- differ by several per cent in the same library's frame-pairs/s.  bench.py runs this before and
// after its timed region so that the line says which.
namespace {
__global__ __launch_bounds__(64) void calib_kernel(double* out, unsigned long long* ticks, int iters) {
  double a0 = threadIdx.x * 1e-3, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double b = 1.0000001, c = 1e-9;
  const unsigned long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a0 = fma(a0, b, c); a1 = fma(a1, b, c); a2 = fma(a2, b, c); a3 = fma(a3, b, c);
      a4 = fma(a4, b, c); a5 = fma(a5, b, c); a6 = fma(a6, b, c); a7 = fma(a7, b, c);
    }
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * 64 + threadIdx.x] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (threadIdx.x == 0) {
    ticks[2 * blockIdx.x] = c1 - c0;
    ticks[2 * blockIdx.x + 1] = w1 - w0;
  }
}
}  // namespace

extern "C" int peaq_calibrate(peaq_ctx* c, int iterations, peaq_calibration* out) {
  if (!c || !out) return fail(PEAQ_ERR_ARG, "peaq_calibrate: NULL argument");
  if (iterations <= 0) iterations = 20000;           // x 64 FMAs per wave: about 5 ms
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  int wall_khz = 100000;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device);
  const int waves = prop.multiProcessorCount * 4 * 2;   // two per SIMD
  TmpBuf sink, ticks;
  HIP_TRY(sink.reserve((size_t)waves * 64 * sizeof(double)));
  HIP_TRY(ticks.reserve((size_t)waves * 2 * sizeof(unsigned long long)));
  if (c->batch_pending) HIP_TRY(hipEventSynchronize(c->batch_end));   // a batch still running would share the device with the probe
  hipLaunchKernelGGL(calib_kernel, dim3(waves), dim3(64), 0, 0, sink.as<double>(), ticks.as<unsigned long long>(), 200);   // warm
  struct Events {                                    // destroyed on every way out
    hipEvent_t a = nullptr, b = nullptr;
    ~Events() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
    }
  } ev;
  HIP_TRY(hipEventCreate(&ev.a));
  HIP_TRY(hipEventCreate(&ev.b));
  HIP_TRY(hipEventRecord(ev.a, 0));
  hipLaunchKernelGGL(calib_kernel, dim3(waves), dim3(64), 0, 0, sink.as<double>(), ticks.as<unsigned long long>(), iterations);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ev.b, 0));
  HIP_TRY(hipEventSynchronize(ev.b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, ev.a, ev.b));
  std::vector<unsigned long long> h((size_t)waves * 2);
  HIP_TRY(hipMemcpy(h.data(), ticks.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double shader = 0., wall = 0.;
  for (int i = 0; i < waves; ++i) {
    shader += (double)h[2 * i];
    wall += (double)h[2 * i + 1];
  }
  const double fmas = (double)waves * 64. * 64. * iterations;   // lanes x FMAs per iteration
  out->elapsed_ms = ms;
  out->shader_clock_mhz = wall > 0. ? shader / wall * (wall_khz * 1e-3) : 0.;
  out->fp64_tflops = ms > 0.f ? 2. * fmas / (ms * 1e-3) * 1e-12 : 0.;
  out->cycles_per_fma = shader / ((double)waves * 64. * iterations);   // per wave, two waves sharing a SIMD
  out->compute_units = prop.multiProcessorCount;
  out->max_clock_mhz = prop.clockRate * 1e-3;
  return PEAQ_OK;
}
