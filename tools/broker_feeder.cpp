// broker_feeder.cpp -- native feeder for the live-pipeline broker (BASELINE.json
// configs[4]: "1024 concurrent live ref/test pipelines feeding the batched GPU
// path").  Plain host code on the C ABI of include/peaq_amd.h -- what a process
// hosting that many `peaq` elements does through pad_chain (reference
// gstpeaq.c:614-661), minus GStreamer: T feeder threads own N/T sessions each
// and push their streams buffer by buffer, round robin over their sessions, ref
// then test, so that all N sessions are mid-stream at the same time; the broker's
// own tick thread batches whatever became ready.  At the end every session is
// flushed (do_flush, gstpeaq.c:716-745) and its result compared with the batch
// path run on the SAME seeded pairs: bit-equal in the basic version; in the
// advanced one 1e-9 relative in either arithmetic of the filter bank (LDS-atomic
// summation order; a stream is cut into 48 blocks per tick here,
// 840 per launch in the batch path).
//
//   broker_feeder [--sessions N] [--seconds S] [--threads T] [--chunk SAMPLES]
//                 [--channels C] [--advanced] [--period-us P] [--seed0 K] [--ragged]
//                 [--realtime] [--devices 0,0,1,...]
// --devices: peaq_broker_create_multi on the listed GPU ordinals (an ordinal may repeat: two device brokers on one
//   GPU) instead of one broker on device 0; the JSON line then carries "devices" and every session's id says which
//   device broker serves it (id % devices).
// --realtime: every session delivers --chunk samples per pad every chunk / 48000 s (1024: one
// frame-pair per 21.3 ms, the pace of a live pipeline) instead of as fast as the feeders can
// push; the latency figures of peaq_broker_stats_t are what this mode is for.
// Prints one JSON object; exit status 0 = all sessions match the batch results.
//
// Build: make -C tools   (hipcc; links ../gstpeaq_amd/libpeaq_amd.so)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/peaq_amd.h"

#define CHECK_PEAQ(x)                                                               \
  do {                                                                              \
    if ((x) != PEAQ_OK) {                                                           \
      std::fprintf(stderr, "%s: %s\n", #x, peaq_last_error());                      \
      return 2;                                                                     \
    }                                                                               \
  } while (0)
#define CHECK_HIP(x)                                                                \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      return 2;                                                                     \
    }                                                                               \
  } while (0)

static double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static bool same_value(double a, double b, double rtol) {
  if (std::isnan(a) || std::isnan(b)) return std::isnan(a) && std::isnan(b);
  if (rtol == 0.) return std::memcmp(&a, &b, sizeof a) == 0 || a == b;
  return std::fabs(a - b) <= rtol * std::fmax(std::fabs(a), std::fabs(b)) + 1e-12;
}

int main(int argc, char** argv) {
  int sessions = 1024, threads = 16, channels = 2, advanced = 0, ragged = 0, realtime = 0;
  double seconds = 2.0;
  unsigned chunk = 4096, period_us = 1000, seed0 = 1;
  std::vector<int> devices;
  for (int i = 1; i < argc; ++i) {
    auto arg = [&](const char* name) { return !std::strcmp(argv[i], name) && i + 1 < argc; };
    if (arg("--sessions")) sessions = std::atoi(argv[++i]);
    else if (arg("--seconds")) seconds = std::atof(argv[++i]);
    else if (arg("--threads")) threads = std::atoi(argv[++i]);
    else if (arg("--chunk")) chunk = (unsigned)std::atoi(argv[++i]);
    else if (arg("--channels")) channels = std::atoi(argv[++i]);
    else if (arg("--period-us")) period_us = (unsigned)std::atoi(argv[++i]);
    else if (arg("--seed0")) seed0 = (unsigned)std::strtoul(argv[++i], nullptr, 0);
    else if (arg("--devices")) {
      for (const char* p = argv[++i]; *p;) {
        devices.push_back((int)std::strtol(p, const_cast<char**>(&p), 10));
        if (*p == ',') ++p;
      }
    }
    else if (!std::strcmp(argv[i], "--advanced")) advanced = 1;
    else if (!std::strcmp(argv[i], "--ragged")) ragged = 1;
    else if (!std::strcmp(argv[i], "--realtime")) realtime = 1;
    else {
      std::fprintf(stderr, "unknown argument %s (see the header of tools/broker_feeder.cpp)\n", argv[i]);
      return 1;
    }
  }
  if (sessions < 1 || threads < 1 || chunk < 1 || (channels != 1 && channels != 2)) return 1;
  const uint32_t ns = (uint32_t)std::lround(seconds * 48000.);
  const size_t per_pair = (size_t)ns * channels;

  peaq_ctx* ctx = nullptr;
  CHECK_PEAQ(peaq_ctx_create(0, &ctx));

  // ---- the pairs, generated where the batch path consumes them; then the batch results -------------
  float *d_ref = nullptr, *d_test = nullptr;
  peaq_result* d_res = nullptr;
  CHECK_HIP(hipMalloc(&d_ref, per_pair * sessions * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_test, per_pair * sessions * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_res, sizeof(peaq_result) * sessions));
  CHECK_PEAQ(peaq_synth_fill(ctx, seed0, sessions, channels, ns, ns, d_ref, d_test, nullptr));
  CHECK_PEAQ(peaq_batch_run(ctx, advanced, channels, 92., sessions, d_ref, d_test, ns, nullptr, nullptr, ns, d_res,
                            nullptr));
  CHECK_HIP(hipDeviceSynchronize());
  std::vector<peaq_result> batch(sessions), live(sessions);
  CHECK_HIP(hipMemcpy(batch.data(), d_res, sizeof(peaq_result) * sessions, hipMemcpyDeviceToHost));
  std::vector<float> h_ref(per_pair * sessions), h_test(per_pair * sessions);
  CHECK_HIP(hipMemcpy(h_ref.data(), d_ref, h_ref.size() * sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(h_test.data(), d_test, h_test.size() * sizeof(float), hipMemcpyDeviceToHost));
  CHECK_HIP(hipFree(d_ref));
  CHECK_HIP(hipFree(d_test));
  CHECK_HIP(hipFree(d_res));

  // ---- the live sessions ---------------------------------------------------------------------------
  peaq_broker* br = nullptr;
  if (devices.empty())
    CHECK_PEAQ(peaq_broker_create(ctx, advanced, channels, 92., sessions, &br));
  else
    CHECK_PEAQ(peaq_broker_create_multi(devices.data(), (int)devices.size(), advanced, channels, 92., sessions, nullptr,
                                        peaq_ctx_get_fir_mode(ctx), &br));
  std::vector<int> sid(sessions);
  for (int s = 0; s < sessions; ++s) CHECK_PEAQ(peaq_broker_open(br, &sid[s]));
  // how the sessions were dealt out
  const int n_dev = peaq_broker_devices(br);
  std::vector<int> per_dev(n_dev, 0);
  for (int s = 0; s < sessions; ++s) ++per_dev[sid[s] % n_dev];
  int dev_min = sessions, dev_max = 0;
  for (int d = 0; d < n_dev; ++d) {
    dev_min = std::min(dev_min, per_dev[d]);
    dev_max = std::max(dev_max, per_dev[d]);
  }
  CHECK_PEAQ(peaq_broker_start(br, period_us));

  std::atomic<int> feed_errors{0};
  const double t0 = now_s();
  {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) {
      pool.emplace_back([&, t]() {
        std::vector<int> mine;
        for (int s = t; s < sessions; s += threads) mine.push_back(s);
        std::vector<uint32_t> pos(mine.size(), 0);
        bool more = true;
        unsigned round = 0;
        // --realtime: round r of this thread is due at t0 + r chunk / 48000 (+ a per-thread offset, so that the
        // sessions do not all become ready in the same instant: live pipelines are not phase locked)
        const double round_s = (double)chunk / 48000., phase_s = round_s * t / threads;
        while (more) {
          more = false;
          if (realtime) {
            const double due = t0 + phase_s + round * round_s, wait = due - now_s();
            if (wait > 0) std::this_thread::sleep_for(std::chrono::duration<double>(wait));
            ++round;
          }
          for (size_t k = 0; k < mine.size(); ++k) {
            const int s = mine[k];
            if (pos[k] >= ns) continue;
            // --ragged: every session has its own buffer size, so frame boundaries never line up
            const uint32_t want = ragged ? 480u + (uint32_t)((s * 2654435761u) % 5000u) : chunk;
            const uint32_t n = std::min<uint32_t>(want, ns - pos[k]);
            const size_t off = (size_t)s * per_pair + (size_t)pos[k] * channels;
            if (peaq_broker_push(br, sid[s], 0, h_ref.data() + off, n) != PEAQ_OK ||
                peaq_broker_push(br, sid[s], 1, h_test.data() + off, n) != PEAQ_OK) {
              if (!feed_errors.fetch_add(1)) std::fprintf(stderr, "push: %s\n", peaq_last_error());
              return;
            }
            pos[k] += n;
            if (pos[k] < ns) more = true;
          }
        }
        for (int s : mine)
          if (peaq_broker_flush(br, sid[s]) != PEAQ_OK) feed_errors.fetch_add(1);
      });
    }
    for (auto& th : pool) th.join();
  }
  const double t_fed = now_s();
  for (int s = 0; s < sessions; ++s) CHECK_PEAQ(peaq_broker_results(br, sid[s], &live[s]));
  const double t1 = now_s();
  peaq_broker_stats_t st{};
  CHECK_PEAQ(peaq_broker_stats(br, &st));
  CHECK_PEAQ(peaq_broker_stop(br));

  // ---- compare -------------------------------------------------------------------------------------
  // basic: bit for bit; advanced: 1e-9 in either arithmetic of the filter bank (its one recurrence along the stream
  // runs in FP64 in both, so where launches cut a stream only moves FP64 rounding)
  const double rtol = !advanced ? 0. : 1e-9;
  const int n_movs = advanced ? PEAQ_MOVS_ADVANCED : PEAQ_MOVS_BASIC;
  int mismatches = 0, nan_odg = 0;
  double frames = 0., max_dodg = 0.;
  for (int s = 0; s < sessions; ++s) {
    bool ok = same_value(live[s].di, batch[s].di, rtol) && same_value(live[s].odg, batch[s].odg, rtol) &&
              same_value(live[s].totalsnr, batch[s].totalsnr, advanced ? 1e-9 : 1e-12) &&
              live[s].frames == batch[s].frames && live[s].fb_blocks == batch[s].fb_blocks;
    for (int i = 0; i < n_movs; ++i) ok = ok && same_value(live[s].movs[i], batch[s].movs[i], rtol);
    if (!ok) {
      if (mismatches < 5)
        std::fprintf(stderr, "session %d: live odg %.17g di %.17g frames %g | batch odg %.17g di %.17g frames %g\n", s,
                     live[s].odg, live[s].di, live[s].frames, batch[s].odg, batch[s].di, batch[s].frames);
      ++mismatches;
    }
    if (std::isnan(live[s].odg)) ++nan_odg;
    else max_dodg = std::fmax(max_dodg, std::fabs(live[s].odg - batch[s].odg));
    frames += live[s].frames;
  }
  for (int s = 0; s < sessions; ++s) CHECK_PEAQ(peaq_broker_close(br, sid[s]));
  peaq_broker_destroy(br);
  peaq_ctx_destroy(ctx);

  std::printf("{\"devices\": %d, \"sessions_per_device_min\": %d, \"sessions_per_device_max\": %d, ", n_dev, dev_min, dev_max);
  std::printf(
      "\"sessions\": %d, \"advanced\": %d, \"channels\": %d, \"seconds_per_session\": %g, \"feeder_threads\": %d, "
      "\"chunk\": %u, \"ragged\": %d, \"period_us\": %u, \"frame_pairs\": %.0f, \"feed_s\": %.4f, \"total_s\": %.4f, "
      "\"frame_pairs_per_s\": %.1f, \"x_realtime\": %.1f, \"ticks\": %llu, \"launches\": %llu, "
      "\"max_active\": %u, \"worker_failed\": %u, \"feed_errors\": %d, \"mismatches\": %d, \"odg_nan\": %d, "
      "\"max_abs_dodg_vs_batch\": %.3g, \"realtime\": %d, \"tick_host_us_max\": %.1f, \"tick_host_us_p99\": %.1f, \"tick_host_us_mean\": %.1f, "
      "\"tick_device_us_max\": %.1f, \"tick_device_us_p99\": %.1f, \"tick_device_us_mean\": %.1f, \"latency_us_max\": %.1f, \"latency_us_p99\": %.1f, "
      "\"latency_us_mean\": %.1f, \"latency_samples\": %llu}\n",
      sessions, advanced, channels, seconds, threads, chunk, ragged, period_us, frames, t_fed - t0, t1 - t0,
      frames / (t1 - t0), sessions * seconds / (t1 - t0), (unsigned long long)st.ticks,
      (unsigned long long)st.launches, st.max_active, st.worker_failed, feed_errors.load(), mismatches, nan_odg,
      max_dodg, realtime, st.tick_host_us_max, st.tick_host_us_p99, st.tick_host_us_mean, st.tick_device_us_max, st.tick_device_us_p99,
      st.tick_device_us_mean,
      st.latency_us_max, st.latency_us_p99, st.latency_us_mean, (unsigned long long)st.latency_samples);
  return (mismatches || feed_errors.load() || st.worker_failed) ? 3 : 0;
}
