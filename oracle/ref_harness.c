/* ref_harness.c -- drives the REAL reference element (compiled from
 * /root/reference/src where it lies; see oracle/Makefile target `ref`) and
 * prints its results at full precision.  TEST INFRASTRUCTURE, build-container
 * only: used by tools/make_golden.py to generate tests/golden/ref_*.json and by
 * tests that compare the oracle with the reference when oracle/_ref exists.
 *
 * The reference's gstpeaq.c is #included (not copied) so that this driver can
 * read the element's accumulators through the reference's own
 * peaq_movaccum_get_value() instead of parsing the "%f" console output.
 *
 *   ref_harness pair  ADV CH REF.f32 TEST.f32     raw interleaved F32LE files
 *   ref_harness synth ADV CH SEED NSAMPLES        include/peaq_synth.h pair
 *   ref_harness launch ADV "<gst-launch fragment feeding peaq.ref / peaq.test>"
 *   ref_harness time ADV CH SEED0 NPAIRS NSAMPLES [REPEATS START_EPOCH]
 *                                              wall-clock of the element on NPAIRS synth pairs, with their MOVs/DI/ODG;
 *                                              REPEATS > 0: one timed region that starts at START_EPOCH (all-core runs)
 *   (environment REF_PLAYBACK_LEVEL=<dB> sets the element's playback_level in pair / synth / launch)
 *   ref_harness fftear BANDS FILE.f32             per-frame ear-model dumps (mono, hop 1024)
 *   ref_harness fbear FILE.f32                    per-block filter-bank dumps (mono, 192)
 * Output: one JSON object on stdout.
 */
#include REF_GSTPEAQ_C

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
#include "../include/peaq_synth.h"

GST_PLUGIN_STATIC_DECLARE (peaq);

static void
print_arr (const char *name, const double *v, int n, int last)
{
  int i;
  printf ("\"%s\": [", name);
  for (i = 0; i < n; i++) {
    if (isnan (v[i])) printf ("\"nan\"");
    else if (isinf (v[i])) printf (v[i] > 0 ? "\"inf\"" : "\"-inf\"");
    else printf ("%.17g", v[i]);
    if (i + 1 < n) printf (", ");
  }
  printf ("]%s", last ? "" : ", ");
}

static int
run_pipeline (int advanced, const char *desc)
{
  GError *err = NULL;
  GstElement *pipe, *peaq_el;
  GstBus *bus;
  GstMessage *msg;
  GstPeaq *pq;
  double movs[COUNT_MOV_BASIC], di, odg, snr;
  int i, n;

  pipe = gst_parse_launch (desc, &err);
  if (!pipe) {
    fprintf (stderr, "parse error: %s\n", err ? err->message : "?");
    return 2;
  }
  peaq_el = gst_bin_get_by_name (GST_BIN (pipe), "peaq");
  g_object_set (peaq_el, "advanced", advanced, "console-output", FALSE, NULL);
  if (getenv ("REF_PLAYBACK_LEVEL"))            /* the element's playback_level property (gstpeaq.c:273-281) */
    g_object_set (peaq_el, "playback_level", atof (getenv ("REF_PLAYBACK_LEVEL")), NULL);
  gst_element_set_state (pipe, GST_STATE_PLAYING);
  bus = gst_element_get_bus (pipe);
  msg = gst_bus_timed_pop_filtered (bus, GST_CLOCK_TIME_NONE, GST_MESSAGE_EOS | GST_MESSAGE_ERROR);
  if (GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) {
    gst_message_parse_error (msg, &err, NULL);
    fprintf (stderr, "pipeline error: %s\n", err->message);
    return 3;
  }
  gst_message_unref (msg);
  gst_object_unref (bus);
  gst_element_set_state (pipe, GST_STATE_NULL);     /* PAUSED->READY flushes, gstpeaq.c:764-778 */

  pq = GST_PEAQ (peaq_el);
  n = advanced ? COUNT_MOV_ADVANCED : COUNT_MOV_BASIC;
  for (i = 0; i < n; i++)
    movs[i] = peaq_movaccum_get_value (pq->mov_accum[i]);
  g_object_get (peaq_el, "di", &di, "odg", &odg, "totalsnr", &snr, NULL);
  printf ("{\"advanced\": %d, \"channels\": %d, \"frames\": %u, \"fb_frames\": %u, "
          "\"loudness_reached_frame\": %u, ", advanced, pq->channels, pq->frame_counter,
          pq->frame_counter_fb, pq->loudness_reached_frame);
  print_arr ("movs", movs, n, 0);
  print_arr ("di", &di, 1, 0);
  print_arr ("odg", &odg, 1, 0);
  print_arr ("totalsnr", &snr, 1, 1);
  printf ("}\n");
  gst_object_unref (peaq_el);
  gst_object_unref (pipe);
  return 0;
}

static int
run_files (int advanced, int channels, const char *ref, const char *test)
{
  char desc[2048];
  snprintf (desc, sizeof desc,
            "filesrc location=%s ! rawaudioparse format=pcm pcm-format=f32le sample-rate=48000 "
            "num-channels=%d ! peaq.ref "
            "filesrc location=%s ! rawaudioparse format=pcm pcm-format=f32le sample-rate=48000 "
            "num-channels=%d ! peaq.test peaq name=peaq", ref, channels, test, channels);
  return run_pipeline (advanced, desc);
}

static double
now_s (void)
{
  struct timespec t;
  clock_gettime (CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

/* One pair through the real element: builds the pipeline, plays it to EOS, reads MOVs/DI/ODG. */
static int
run_timed_pair (int advanced, const char *desc, double *movs, double *di, double *odg, unsigned *frames)
{
  GError *err = NULL;
  GstElement *pipe, *peaq_el;
  GstBus *bus;
  GstMessage *msg;
  GstPeaq *pq;
  int i, n = advanced ? COUNT_MOV_ADVANCED : COUNT_MOV_BASIC;
  pipe = gst_parse_launch (desc, &err);
  if (!pipe) return 2;
  peaq_el = gst_bin_get_by_name (GST_BIN (pipe), "peaq");
  g_object_set (peaq_el, "advanced", advanced, "console-output", FALSE, NULL);
  gst_element_set_state (pipe, GST_STATE_PLAYING);
  bus = gst_element_get_bus (pipe);
  msg = gst_bus_timed_pop_filtered (bus, GST_CLOCK_TIME_NONE, GST_MESSAGE_EOS | GST_MESSAGE_ERROR);
  if (GST_MESSAGE_TYPE (msg) == GST_MESSAGE_ERROR) return 3;
  gst_message_unref (msg);
  gst_object_unref (bus);
  gst_element_set_state (pipe, GST_STATE_NULL);
  pq = GST_PEAQ (peaq_el);
  for (i = 0; i < COUNT_MOV_BASIC; i++)
    movs[i] = i < n ? peaq_movaccum_get_value (pq->mov_accum[i]) : 0.;
  g_object_get (peaq_el, "di", di, "odg", odg, NULL);
  *frames = pq->frame_counter;
  gst_object_unref (peaq_el);
  gst_object_unref (pipe);
  return 0;
}

static double
epoch_s (void)
{
  struct timespec t;
  clock_gettime (CLOCK_REALTIME, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

/* CPU baseline: the real element, fed from page-cached raw files; generation of
 * the inputs is not timed, pipeline construction/teardown is (a few ms).
 * repeats == 0: NPAIRS distinct pairs, each generated and then timed on its own
 *   (sum of the per-pair times).
 * repeats != 0: NPAIRS distinct pairs are generated first, then -- from the wall
 *   clock instant start_epoch on, so that many such processes (one per core) run
 *   their timed regions together -- the set is processed `repeats` times (repeats < 0:
 *   over and over for -repeats seconds) in one timed region.  The JSON carries the region's begin/end on the CLOCK_REALTIME
 *   axis for the parent to aggregate.
 * Either way the MOVs / DI / ODG of the first pass over each pair are printed. */
static int
time_pairs (int advanced, int channels, uint32_t seed0, int n_pairs, uint32_t ns, int repeats, double start_epoch)
{
  float *r = malloc ((size_t) ns * channels * 4), *t = malloc ((size_t) ns * channels * 4);
  char fr[256], ft[256], desc[2048];
  double total = 0., t_begin = 0., t_end = 0.;
  double *movs = malloc (sizeof (double) * COUNT_MOV_BASIC * n_pairs);
  double *di = malloc (sizeof (double) * n_pairs), *odg = malloc (sizeof (double) * n_pairs);
  unsigned frames = 0;
  int p, rep, rc, n = advanced ? COUNT_MOV_ADVANCED : COUNT_MOV_BASIC;
  for (p = 0; p < n_pairs; p++) {
    FILE *f;
    double t0;
    unsigned fc;
    snprintf (fr, sizeof fr, "/tmp/refh_%d_%d_r.f32", (int) getpid (), repeats ? p : 0);
    snprintf (ft, sizeof ft, "/tmp/refh_%d_%d_t.f32", (int) getpid (), repeats ? p : 0);
    peaq_synth_pair (seed0 + p, channels, ns, r, t);
    f = fopen (fr, "wb"); fwrite (r, 4, (size_t) ns * channels, f); fclose (f);
    f = fopen (ft, "wb"); fwrite (t, 4, (size_t) ns * channels, f); fclose (f);
    if (repeats)
      continue;
    snprintf (desc, sizeof desc,
              "filesrc location=%s ! rawaudioparse format=pcm pcm-format=f32le sample-rate=48000 "
              "num-channels=%d ! peaq.ref "
              "filesrc location=%s ! rawaudioparse format=pcm pcm-format=f32le sample-rate=48000 "
              "num-channels=%d ! peaq.test peaq name=peaq", fr, channels, ft, channels);
    t0 = now_s ();
    rc = run_timed_pair (advanced, desc, movs + p * COUNT_MOV_BASIC, di + p, odg + p, &fc);
    if (rc) return rc;
    total += now_s () - t0;
    frames += fc;
  }
  if (repeats) {
    while (epoch_s () < start_epoch)
      usleep (200);
    t_begin = epoch_s ();
    /* repeats < 0: as many passes as fit into -repeats seconds (bounded run time on any host) */
    for (rep = 0; repeats > 0 ? rep < repeats : (rep == 0 || epoch_s () < t_begin - repeats); rep++)
      for (p = 0; p < n_pairs; p++) {
        double m[COUNT_MOV_BASIC], d, o;
        unsigned fc;
        snprintf (fr, sizeof fr, "/tmp/refh_%d_%d_r.f32", (int) getpid (), p);
        snprintf (ft, sizeof ft, "/tmp/refh_%d_%d_t.f32", (int) getpid (), p);
        snprintf (desc, sizeof desc,
                  "filesrc location=%s ! rawaudioparse format=pcm pcm-format=f32le sample-rate=48000 "
                  "num-channels=%d ! peaq.ref "
                  "filesrc location=%s ! rawaudioparse format=pcm pcm-format=f32le sample-rate=48000 "
                  "num-channels=%d ! peaq.test peaq name=peaq", fr, channels, ft, channels);
        rc = run_timed_pair (advanced, desc, rep ? m : movs + p * COUNT_MOV_BASIC, rep ? &d : di + p,
                             rep ? &o : odg + p, &fc);
        if (rc) return rc;
        frames += fc;
      }
    t_end = epoch_s ();
    total = t_end - t_begin;
  }
  for (p = 0; p < (repeats ? n_pairs : 1); p++) {
    snprintf (fr, sizeof fr, "/tmp/refh_%d_%d_r.f32", (int) getpid (), p);
    snprintf (ft, sizeof ft, "/tmp/refh_%d_%d_t.f32", (int) getpid (), p);
    remove (fr);
    remove (ft);
  }
  printf ("{\"pairs\": %d, \"repeats\": %d, \"frame_pairs\": %u, \"seconds\": %.6f, \"frame_pairs_per_s\": %.1f, "
          "\"t_begin\": %.6f, \"t_end\": %.6f, \"n_movs\": %d, ", n_pairs, repeats, frames, total, frames / total,
          t_begin, t_end, n);
  print_arr ("odg", odg, n_pairs, 0);
  print_arr ("di", di, n_pairs, 0);
  print_arr ("movs", movs, n_pairs * COUNT_MOV_BASIC, 1);
  printf ("}\n");
  return 0;
}

static float *
read_f32 (const char *path, size_t *n)
{
  FILE *f = fopen (path, "rb");
  float *buf;
  long sz;
  if (!f) { perror (path); exit (2); }
  fseek (f, 0, SEEK_END);
  sz = ftell (f);
  fseek (f, 0, SEEK_SET);
  buf = malloc (sz);
  if (fread (buf, 1, sz, f) != (size_t) sz) { perror ("read"); exit (2); }
  fclose (f);
  *n = sz / sizeof (float);
  return buf;
}

static int
dump_fftear (int bands, const char *path)
{
  size_t n, pos;
  float *x = read_f32 (path, &n);
  PeaqEarModel *ear = g_object_new (PEAQ_TYPE_FFTEARMODEL, "number-of-bands", bands, NULL);
  gpointer st = peaq_earmodel_state_alloc (ear);
  int first = 1;
  printf ("{\"bands\": %d, \"frames\": [", bands);
  for (pos = 0; pos + 2048 <= n; pos += 1024) {
    peaq_earmodel_process_block (ear, st, x + pos);
    printf ("%s{", first ? "" : ", ");
    first = 0;
    print_arr ("power", peaq_fftearmodel_get_power_spectrum (st), 1025, 0);
    print_arr ("weighted", peaq_fftearmodel_get_weighted_power_spectrum (st), 1025, 0);
    print_arr ("unsmeared", peaq_earmodel_get_unsmeared_excitation (ear, st), bands, 0);
    print_arr ("excitation", peaq_earmodel_get_excitation (ear, st), bands, 0);
    printf ("\"energy\": %d, \"loudness\": %.17g}", peaq_fftearmodel_is_energy_threshold_reached (st),
            peaq_earmodel_calc_loudness (ear, st));
  }
  printf ("]}\n");
  return 0;
}

static int
dump_fbear (const char *path)
{
  size_t n, pos;
  float *x = read_f32 (path, &n);
  PeaqEarModel *ear = g_object_new (PEAQ_TYPE_FILTERBANKEARMODEL, NULL);
  gpointer st = peaq_earmodel_state_alloc (ear);
  int first = 1;
  printf ("{\"bands\": 40, \"frames\": [");
  for (pos = 0; pos + 192 <= n; pos += 192) {
    peaq_earmodel_process_block (ear, st, x + pos);
    printf ("%s{", first ? "" : ", ");
    first = 0;
    print_arr ("unsmeared", peaq_earmodel_get_unsmeared_excitation (ear, st), 40, 0);
    print_arr ("excitation", peaq_earmodel_get_excitation (ear, st), 40, 0);
    printf ("\"loudness\": %.17g}", peaq_earmodel_calc_loudness (ear, st));
  }
  printf ("]}\n");
  return 0;
}

static int
dump_tables (int bands)
{
  /* band tables through the reference's public getters / struct fields */
  PeaqEarModel *ear = bands == 40 ? g_object_new (PEAQ_TYPE_FILTERBANKEARMODEL, NULL)
    : g_object_new (PEAQ_TYPE_FFTEARMODEL, "number-of-bands", bands, NULL);
  double tc[109];
  int i;
  for (i = 0; i < bands; i++)
    tc[i] = peaq_earmodel_calc_time_constant (ear, i, 0.008, 0.05);
  printf ("{\"bands\": %d, ", bands);
  print_arr ("fc", ear->fc, bands, 0);
  print_arr ("internal_noise", ear->internal_noise, bands, 0);
  print_arr ("ear_tc", ear->ear_time_constants, bands, 0);
  print_arr ("exc_threshold", ear->excitation_threshold, bands, 0);
  print_arr ("threshold", ear->threshold, bands, 0);
  print_arr ("loud_factor", ear->loudness_factor, bands, 0);
  if (bands != 40)
    print_arr ("mask_diff", peaq_fftearmodel_get_masking_difference (PEAQ_FFTEARMODEL (ear)), bands, 0);
  print_arr ("adapt_tc", tc, bands, 1);
  printf ("}\n");
  return 0;
}

int
main (int argc, char **argv)
{
  /* this image keeps GStreamer under /opt/conda and has no system registry */
  setenv ("GST_PLUGIN_SYSTEM_PATH", "/opt/conda/lib/gstreamer-1.0", 0);
  setenv ("GST_PLUGIN_SCANNER", "/opt/conda/libexec/gstreamer-1.0/gst-plugin-scanner", 0);
  setenv ("GST_REGISTRY", "/tmp/peaq_ref_harness_registry.bin", 0);
  gst_init (NULL, NULL);
  GST_PLUGIN_STATIC_REGISTER (peaq);
  if (argc >= 6 && !strcmp (argv[1], "pair"))
    return run_files (atoi (argv[2]), atoi (argv[3]), argv[4], argv[5]);
  if (argc >= 6 && !strcmp (argv[1], "synth")) {
    int adv = atoi (argv[2]), ch = atoi (argv[3]);
    uint32_t seed = (uint32_t) strtoul (argv[4], NULL, 0), ns = (uint32_t) strtoul (argv[5], NULL, 0);
    float *r = malloc ((size_t) ns * ch * 4), *t = malloc ((size_t) ns * ch * 4);
    char fr[256], ft[256];
    FILE *f;
    int rc;
    peaq_synth_pair (seed, ch, ns, r, t);
    snprintf (fr, sizeof fr, "/tmp/refh_%d_r.f32", (int) getpid ());
    snprintf (ft, sizeof ft, "/tmp/refh_%d_t.f32", (int) getpid ());
    f = fopen (fr, "wb"); fwrite (r, 4, (size_t) ns * ch, f); fclose (f);
    f = fopen (ft, "wb"); fwrite (t, 4, (size_t) ns * ch, f); fclose (f);
    rc = run_files (adv, ch, fr, ft);
    remove (fr);
    remove (ft);
    return rc;
  }
  if (argc >= 7 && !strcmp (argv[1], "time"))
    return time_pairs (atoi (argv[2]), atoi (argv[3]), (uint32_t) strtoul (argv[4], NULL, 0), atoi (argv[5]),
                       (uint32_t) strtoul (argv[6], NULL, 0), argc >= 9 ? atoi (argv[7]) : 0,
                       argc >= 9 ? atof (argv[8]) : 0.);
  if (argc >= 4 && !strcmp (argv[1], "launch"))
    return run_pipeline (atoi (argv[2]), argv[3]);
  if (argc >= 4 && !strcmp (argv[1], "fftear"))
    return dump_fftear (atoi (argv[2]), argv[3]);
  if (argc >= 3 && !strcmp (argv[1], "fbear"))
    return dump_fbear (argv[2]);
  if (argc >= 3 && !strcmp (argv[1], "tables"))
    return dump_tables (atoi (argv[2]));
  fprintf (stderr, "usage: see the header of oracle/ref_harness.c\n");
  return 1;
}
