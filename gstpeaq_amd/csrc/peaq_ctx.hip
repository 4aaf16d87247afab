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
// ONE wave per SIMD runs sixteen independent chains of v_fma_f64; one lane of every workgroup reads the shader clock
// (s_memtime) and the constant 100 MHz counter at the start, in the MIDDLE and at the end of its chain.  The chip
// clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"), and PEAQ's FP64-dense kernels sit at that budget:
// two boxes -- or one box at two moments -- differ by several per cent in the same library's frame-pairs/s.  bench.py
// runs this before and after its timed region so that the line says which.  The kernel starts from whatever clock the
// idle device held: the FIRST half of the run is reported as the ramp, the SECOND half -- tens of milliseconds in --
// as the steady state.  (Rounds 5 and, at first, 6 ran TWO waves per SIMD with eight chains each and read "8 cycles per
// multiply-add and wave = the pipe's 4" into it; measured over 70 ms: 9.06 in the first half, 4.69 in the second, a
// kernel that lasts twice a wave's lifetime -- a SIMD issues from its OLDEST ready wave first, so with always-ready
// chains the two waves run one after the other, not side by side, and per-wave times say nothing about the device's
// rate.  One wave per SIMD has no partner to be confused with: its 4.0x cycles per multiply-add ARE the pipe's.)
namespace {
__global__ __launch_bounds__(64) void calib_kernel(double* out, unsigned long long* ticks, int iters) {
  double a[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = threadIdx.x * 1e-3 + k;
  const double b = 1.0000001, c = 1e-9;
  unsigned long long w[3], cy[3];
  w[0] = wall_clock64();
  cy[0] = __builtin_readcyclecounter();
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
#pragma unroll 1
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 32; ++j) {
#pragma unroll
        for (int k = 0; k < 16; ++k) a[k] = fma(a[k], b, c);
      }
    }
    cy[half + 1] = __builtin_readcyclecounter();
    w[half + 1] = wall_clock64();
  }
  double sum = 0.;
#pragma unroll
  for (int k = 0; k < 16; ++k) sum += a[k];
  out[blockIdx.x * 64 + threadIdx.x] = sum;
  if (threadIdx.x == 0) {
    ticks[6 * blockIdx.x] = cy[1] - cy[0];
    ticks[6 * blockIdx.x + 1] = w[1] - w[0];
    ticks[6 * blockIdx.x + 2] = cy[2] - cy[1];
    ticks[6 * blockIdx.x + 3] = w[2] - w[1];
    // where the wave ran: HW_ID (wave slot, SIMD, CU, shader array and engine) and the XCC; and when it started
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    ticks[6 * blockIdx.x + 4] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    ticks[6 * blockIdx.x + 5] = w[0];
  }
}
constexpr int kCalibFmasPerIter = 512;               // 16 chains x 32 (the loop's own ~ 40 cycles per trip: 2 % of 2048)
}  // namespace

extern "C" int peaq_calibrate(peaq_ctx* c, int iterations, peaq_calibration* out) {
  if (!c || !out) return fail(PEAQ_ERR_ARG, "peaq_calibrate: NULL argument");
  if (iterations <= 0) iterations = 80000;           // x 512 multiply-adds per wave: about 70 ms, the clock has settled by half-way
  const int half = (iterations + 1) / 2;
  std::lock_guard<std::mutex> lock(c->mu);
  HIP_TRY(hipSetDevice(c->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, c->device));
  int wall_khz = 100000;
  (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, c->device);
  const int waves = prop.multiProcessorCount * 4;      // one per SIMD
  TmpBuf sink, ticks;
  HIP_TRY(sink.reserve((size_t)waves * 64 * sizeof(double)));
  HIP_TRY(ticks.reserve((size_t)waves * 6 * sizeof(unsigned long long)));
  // "The clock under a fixed load" means NOTHING else on the device: this context's batch, and whatever its sessions,
  // brokers or other contexts of the process still have in flight (their streams do not synchronise with ours).
  // Work of OTHER processes on the device cannot be seen from here: the caller's responsibility (include/peaq_amd.h).
  HIP_TRY(hipDeviceSynchronize());
  struct Scope {                                     // destroyed on every way out
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t s = nullptr;
    ~Scope() {
      if (a) (void)hipEventDestroy(a);
      if (b) (void)hipEventDestroy(b);
      if (s) (void)hipStreamDestroy(s);
    }
  } sc;
  HIP_TRY(hipStreamCreateWithFlags(&sc.s, hipStreamNonBlocking));
  HIP_TRY(hipEventCreate(&sc.a));
  HIP_TRY(hipEventCreate(&sc.b));
  HIP_TRY(hipEventRecord(sc.a, sc.s));
  hipLaunchKernelGGL(calib_kernel, dim3(waves), dim3(64), 0, sc.s, sink.as<double>(), ticks.as<unsigned long long>(), half);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(sc.b, sc.s));
  HIP_TRY(hipEventSynchronize(sc.b));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, sc.a, sc.b));
  std::vector<unsigned long long> h((size_t)waves * 6);
  HIP_TRY(hipMemcpy(h.data(), ticks.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double shader[2] = {0., 0.}, wall[2] = {0., 0.};
  // the SIMDs the waves ran on (XCC, shader engine / array, CU, SIMD: HW_ID bits 4-5 SIMD, 8-11 CU, 12 array, 13-15
  // engine) and how many waves the busiest of them got; the spread of the waves' start times
  std::vector<unsigned long long> simd_key((size_t)waves);
  unsigned long long t_first = ~0ull, t_last = 0;
  for (int i = 0; i < waves; ++i) {
    for (int k = 0; k < 2; ++k) {
      shader[k] += (double)h[6 * i + 2 * k];
      wall[k] += (double)h[6 * i + 2 * k + 1];
    }
    const unsigned long long id = h[6 * i + 4];
    simd_key[i] = (id >> 32 << 16) | (id & 0xff30u);
    t_first = std::min(t_first, h[6 * i + 5]);
    t_last = std::max(t_last, h[6 * i + 5]);
  }
  std::sort(simd_key.begin(), simd_key.end());
  int distinct = 0, busiest = 0;
  for (size_t i = 0; i < simd_key.size();) {
    size_t j = i;
    while (j < simd_key.size() && simd_key[j] == simd_key[i]) ++j;
    ++distinct;
    busiest = std::max(busiest, (int)(j - i));
    i = j;
  }
  const double fmas_half = (double)waves * 64. * kCalibFmasPerIter * half;   // lanes x multiply-adds per iteration x iterations
  const double wall_hz = wall_khz * 1e3;
  auto mhz = [&](int k) { return wall[k] > 0. ? shader[k] / wall[k] * (wall_khz * 1e-3) : 0.; };
  auto cpf = [&](int k) { return shader[k] / ((double)waves * kCalibFmasPerIter * half); };   // per wave = per SIMD
  out->elapsed_ms = ms;
  out->shader_clock_mhz = mhz(1);
  // all waves run at once, each alone on its SIMD: the rate is the half's work over the MEAN duration of a wave's second half
  out->fp64_tflops = wall[1] > 0. ? 2. * fmas_half / (wall[1] / waves / wall_hz) * 1e-12 : 0.;
  out->cycles_per_fma = cpf(1);
  out->compute_units = prop.multiProcessorCount;
  out->max_clock_mhz = prop.clockRate * 1e-3;
  out->ramp_clock_mhz = mhz(0);
  out->ramp_cycles_per_fma = cpf(0);
  out->event_fp64_tflops = ms > 0.f ? 4. * fmas_half / (ms * 1e-3) * 1e-12 : 0.;
  out->simds_used = distinct;
  out->max_waves_on_a_simd = busiest;
  out->dispatch_spread_ms = (double)(t_last - t_first) / wall_hz * 1e3;
  return PEAQ_OK;
}
