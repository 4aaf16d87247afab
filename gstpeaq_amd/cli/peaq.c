/* peaq.c -- the `peaq REFFILE TESTFILE` command-line tool on the MI355X engine.
 *
 * Same interface as the reference's CLI (src/peaq.c): options --basic (default),
 * --advanced, --version; prints
 *     Objective Difference Grade: %.3f
 *     Distortion Index: %.3f
 * (peaq.c:217-220); exit status 0, 1 on usage errors (:110-133), 2 when the
 * engine cannot be set up (:147-195).  The reference builds a GStreamer pipeline
 * filesrc ! wavparse ! audioconvert ! audioresample ! peaq (:154-209); this tool
 * reads RIFF/WAVE itself (PCM 8/16/24/32 bit and IEEE float 32/64, mono or
 * stereo) and feeds the engine's session API directly, so that it works on a
 * box without gst-plugins-good.  Integer PCM is scaled by 1/2^(bits-1) like
 * audioconvert does (S16 -> x/32768, verified in SURVEY.md 8(c)).  The ear
 * models are defined for 48 kHz only (earmodel.c:43): files at another rate are
 * converted first, as the reference's `audioresample` does -- here with a
 * Kaiser-windowed sinc interpolator whose parameters are the measured ones of
 * that `audioresample` (resample_to_48k below).  Against the real reference
 * chain ODG/DI agree to 6e-5 in seven of eight pinned cases and 2.3e-3 in the
 * eighth (tests/test_cli_resampler.py; stated tolerance 5e-3).  48 kHz files:
 * digit for digit in both versions -- the advanced version's filter bank runs
 * in FP64 like the reference (the engine's default); PEAQ_AMD_FIR=f16x3 selects
 * the reduced-precision bank (held to 1e-6 in ODG/DI, include/peaq_amd.h).
 * --no-resample refuses files at other rates instead.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "peaq_amd.h"

typedef struct
{
  float *samples;               /* interleaved */
  size_t frames;                /* samples per channel */
  int channels, rate;
} wav_t;

static uint32_t
rd32 (const unsigned char *p)
{
  return (uint32_t) p[0] | ((uint32_t) p[1] << 8) | ((uint32_t) p[2] << 16) | ((uint32_t) p[3] << 24);
}

static int
wav_read (const char *path, wav_t * w)
{
  FILE *f = fopen (path, "rb");
  unsigned char hdr[12], ck[8];
  int fmt_tag = 0, bits = 0, have_fmt = 0, block_align = 0;
  memset (w, 0, sizeof *w);
  if (!f) {
    fprintf (stderr, "Error: cannot open %s\n", path);
    return -1;
  }
  if (fread (hdr, 1, 12, f) != 12 || memcmp (hdr, "RIFF", 4) || memcmp (hdr + 8, "WAVE", 4)) {
    fprintf (stderr, "Error: %s is not a RIFF/WAVE file\n", path);
    fclose (f);
    return -1;
  }
  while (fread (ck, 1, 8, f) == 8) {
    uint32_t size = rd32 (ck + 4);
    if (!memcmp (ck, "fmt ", 4)) {
      unsigned char fmt[40];
      uint32_t n = size < sizeof fmt ? size : sizeof fmt;
      if (size < 16 || fread (fmt, 1, n, f) != n)
        break;
      fmt_tag = fmt[0] | (fmt[1] << 8);
      w->channels = fmt[2] | (fmt[3] << 8);
      w->rate = (int) rd32 (fmt + 4);
      block_align = fmt[12] | (fmt[13] << 8);
      bits = fmt[14] | (fmt[15] << 8);
      if (fmt_tag == 0xFFFE && size >= 26)      /* WAVE_FORMAT_EXTENSIBLE: sub-format GUID */
        fmt_tag = fmt[24] | (fmt[25] << 8);
      have_fmt = 1;
      if (size > n)
        fseek (f, (long) (size - n), SEEK_CUR);
      if (size & 1)
        fseek (f, 1, SEEK_CUR);
    } else if (!memcmp (ck, "data", 4) && have_fmt) {
      size_t bytes_per = (size_t) bits / 8, total, i;
      unsigned char *raw;
      if (w->channels < 1 || w->channels > 2 || block_align != (int) bytes_per * w->channels ||
          !((fmt_tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) ||
              (fmt_tag == 3 && (bits == 32 || bits == 64)))) {
        fprintf (stderr, "Error: %s: unsupported WAVE format (tag %d, %d bit, %d channels)\n", path,
            fmt_tag, bits, w->channels);
        fclose (f);
        return -1;
      }
      raw = malloc (size ? size : 1);
      total = fread (raw, 1, size, f) / bytes_per;     /* tolerate a truncated data chunk */
      total -= total % w->channels;
      w->frames = total / w->channels;
      w->samples = malloc ((total ? total : 1) * sizeof (float));
      for (i = 0; i < total; i++) {
        const unsigned char *p = raw + i * bytes_per;
        double v;
        if (fmt_tag == 3) {
          if (bits == 32) {
            float t;
            memcpy (&t, p, 4);
            v = t;
          } else {
            memcpy (&v, p, 8);
          }
        } else if (bits == 8) {
          v = ((int) p[0] - 128) / 128.;
        } else if (bits == 16) {
          v = (int16_t) (p[0] | (p[1] << 8)) / 32768.;
        } else if (bits == 24) {
          int32_t t = (int32_t) ((uint32_t) p[0] << 8 | (uint32_t) p[1] << 16 | (uint32_t) p[2] << 24) >> 8;
          v = t / 8388608.;
        } else {
          v = (int32_t) rd32 (p) / 2147483648.;
        }
        w->samples[i] = (float) v;
      }
      free (raw);
      fclose (f);
      return 0;
    } else {
      fseek (f, (long) (size + (size & 1)), SEEK_CUR);
    }
  }
  fprintf (stderr, "Error: %s: no usable fmt/data chunks\n", path);
  fclose (f);
  return -1;
}

/* ---- sample-rate conversion to 48 kHz (stands in for audioresample, peaq.c:154-209) ----
 * y[m] = sum_n x[n] h(t_m - n), t_m = m rate / 48000 - 1/8, h = Kaiser-windowed sinc.
 * The parameters are those of the reference's chain as it runs here -- `audioresample` of GStreamer 1.14 at its
 * default quality, measured through its impulse response (tools/make_golden.py resampled; the fit leaves 5e-5
 * of the peak): cutoff 0.94 of the input's Nyquist frequency and 64 taps when the rate goes up, 0.921 of the
 * output's and 64 rate / 48000 taps (rounded up to a multiple of 8) when it goes down, Kaiser beta 8.4-8.5 (85 dB),
 * and a delay of one eighth of an INPUT sample; the last output sample is the last one whose position does not
 * pass the last input sample.  Rational ratios (44.1 kHz: 160 / 147) run from a polyphase table. */
static double
bessel_i0 (double x)
{
  double sum = 1., term = 1.;
  int k;
  for (k = 1; k < 60; k++) {
    term *= (x / (2. * k)) * (x / (2. * k));
    sum += term;
    if (term < 1e-18 * sum)
      break;
  }
  return sum;
}

typedef struct
{
  double fc, half, beta, i0b;
} rs_kernel;

static double
rs_tap (const rs_kernel * k, double d)
{                               /* h(d), d in input samples */
  const double u = d / k->half, arg = 2. * M_PI * k->fc * d;
  if (fabs (u) > 1.)
    return 0.;
  return 2. * k->fc * (fabs (arg) < 1e-12 ? 1. : sin (arg) / arg) * bessel_i0 (k->beta * sqrt (1. - u * u)) / k->i0b;
}

static unsigned long
gcd_ul (unsigned long a, unsigned long b)
{
  while (b) {
    const unsigned long t = a % b;
    a = b;
    b = t;
  }
  return a;
}

static int
resample_to_48k (wav_t * w)
{
  const double ratio = 48000. / w->rate;                        /* output samples per input sample */
  const double delay = 0.125;                                   /* input samples */
  rs_kernel k;
  const unsigned long g = gcd_ul (48000ul, (unsigned long) w->rate);
  const unsigned long L = 48000ul / g, M = (unsigned long) w->rate / g;   /* t_m = m M / L - delay */
  const size_t out_frames = w->frames ? (size_t) floor ((double) (w->frames - 1) * ratio) + 1 : 0;
  float *out = malloc ((out_frames ? out_frames : 1) * w->channels * sizeof (float));
  double *table = NULL;
  long K, j;
  size_t m;
  int c;
  if (!out)
    return -1;
  if (ratio >= 1.) {
    k.fc = 0.94 * 0.5;
    k.half = 32.15;
    k.beta = 8.49;
  } else {
    k.fc = 0.921 * 0.5 * ratio;
    k.half = 4. * ceil (64. / ratio / 8.);
    k.beta = 8.41;
  }
  k.i0b = bessel_i0 (k.beta);
  K = (long) ceil (k.half) + 1;                                 /* taps n = floor(t) - K + 1 .. floor(t) + K */
  if (L <= 4096) {                                              /* phase p = (m M) mod L: d = frac(t) + K - 1 - j */
    unsigned long p;
    table = malloc ((size_t) L * 2 * K * sizeof (double));
    if (!table) {
      free (out);
      return -1;
    }
    for (p = 0; p < L; p++) {
      /* t = q + p / L - delay with an integer q: floor and fraction of p / L - delay */
      const double tf = (double) p / (double) L - delay, fl = floor (tf), fr = tf - fl;
      for (j = 0; j < 2 * K; j++)
        table[p * 2 * K + j] = rs_tap (&k, fr + (double) (K - 1 - j));
    }
  }
  for (m = 0; m < out_frames; m++) {
    const unsigned long long mm = (unsigned long long) m * M;
    const unsigned long p = (unsigned long) (mm % L);
    const double tf = (double) p / (double) L - delay, fl = floor (tf);
    const long n_first = (long) (mm / L) + (long) fl - K + 1;    /* floor(t) - K + 1 */
    const double fr = tf - fl;
    double acc[2] = { 0., 0. };
    for (j = 0; j < 2 * K; j++) {
      const long n = n_first + j;
      double h;
      if (n < 0 || n >= (long) w->frames)
        continue;
      h = table ? table[p * 2 * K + j] : rs_tap (&k, fr + (double) (K - 1 - j));
      for (c = 0; c < w->channels; c++)
        acc[c] += h * w->samples[(size_t) n * w->channels + c];
    }
    for (c = 0; c < w->channels; c++)
      out[m * w->channels + c] = (float) acc[c];
  }
  free (table);
  free (w->samples);
  w->samples = out;
  w->frames = out_frames;
  w->rate = 48000;
  return 0;
}

static void
usage (const char *prog)
{
  printf ("Usage:\n  %s [OPTION...] REFFILE TESTFILE\n\n"
      "peaq computes the Objective Difference Grade based on ITU-R BS.1387-1 (but it\n"
      "does not meet its conformance requirements), on an AMD MI355X.\n\n"
      "  --version     print version information\n"
      "  --advanced    use advanced version\n"
      "  --basic       use basic version (default)\n"
      "  --level=DB    playback level in dB SPL of a full-scale sine (default 92)\n"
      "  --no-resample refuse files that are not sampled at 48 kHz instead of converting them\n", prog);
}

int
main (int argc, char **argv)
{
  int advanced = 0, i, nfiles = 0, rc, allow_resample = 1;
  double level = 92.;
  const char *files[2] = { NULL, NULL };
  wav_t ref, test;
  peaq_ctx *ctx = NULL;
  peaq_session *s = NULL;
  peaq_result r;
  size_t pos;

  for (i = 1; i < argc; i++) {
    if (!strcmp (argv[i], "--advanced"))
      advanced = 1;
    else if (!strcmp (argv[i], "--basic"))
      advanced = 0;
    else if (!strncmp (argv[i], "--level=", 8))
      level = atof (argv[i] + 8);
    else if (!strcmp (argv[i], "--no-resample"))
      allow_resample = 0;
    else if (!strcmp (argv[i], "--version")) {
      printf ("peaq (gstpeaq_amd) %s\n", peaq_version ());
      return 0;
    } else if (!strcmp (argv[i], "--help") || !strcmp (argv[i], "-h")) {
      usage (argv[0]);
      return 0;
    } else if (argv[i][0] == '-' && argv[i][1] == '-') {
      fprintf (stderr, "Failed to initialize: Unknown option %s\n", argv[i]);
      return 1;
    } else if (nfiles < 2)
      files[nfiles++] = argv[i];
    else
      nfiles++;
  }
  if (nfiles != 2) {
    usage (argv[0]);
    return 1;
  }
  if (wav_read (files[0], &ref) || wav_read (files[1], &test))
    return 2;
  if (ref.rate != 48000 || test.rate != 48000) {
    if (!allow_resample || ref.rate < 8000 || test.rate < 8000) {
      fprintf (stderr, "Error: both files must be sampled at 48 kHz (got %d and %d Hz)\n", ref.rate, test.rate);
      return 2;
    }
    if ((ref.rate != 48000 && resample_to_48k (&ref)) || (test.rate != 48000 && resample_to_48k (&test))) {
      fprintf (stderr, "Error: out of memory while resampling\n");
      return 2;
    }
  }
  if (ref.channels != test.channels) {
    /* the element negotiates equal channel counts via audioconvert; up-mix the mono side */
    wav_t *m = ref.channels == 1 ? &ref : &test;
    float *st = malloc ((m->frames ? m->frames : 1) * 2 * sizeof (float));
    for (pos = 0; pos < m->frames; pos++)
      st[2 * pos] = st[2 * pos + 1] = m->samples[pos];
    free (m->samples);
    m->samples = st;
    m->channels = 2;
  }
  if (getenv ("PEAQ_AMD_CLI_DUMP")) {
    /* test hook (tests/test_cli_resampler.py, runs without a GPU): what would be handed to the engine, as raw
     * interleaved F32 in <value>.ref.f32 / <value>.test.f32 -- reader, rate conversion and up-mix on their own */
    const wav_t *w[2] = { &ref, &test };
    const char *suffix[2] = { ".ref.f32", ".test.f32" };
    for (i = 0; i < 2; i++) {
      char path[4096];
      FILE *f;
      snprintf (path, sizeof path, "%s%s", getenv ("PEAQ_AMD_CLI_DUMP"), suffix[i]);
      f = fopen (path, "wb");
      if (!f || fwrite (w[i]->samples, sizeof (float), w[i]->frames * w[i]->channels, f) != w[i]->frames * w[i]->channels) {
        fprintf (stderr, "Error: cannot write %s\n", path);
        return 2;
      }
      fclose (f);
    }
    printf ("dumped %zu and %zu frames, %d channels\n", ref.frames, test.frames, ref.channels);
    return 0;
  }
  if (peaq_ctx_create (getenv ("PEAQ_AMD_DEVICE") ? atoi (getenv ("PEAQ_AMD_DEVICE")) : 0, &ctx) != PEAQ_OK) {
    printf ("Error: peaq engine could not be instantiated - %s\n", peaq_last_error ());
    return 2;
  }
  /* The advanced version's filter bank runs in the engine's default arithmetic, the reference's own (all FP64);
   * the faster reduced-precision bank only on request (PEAQ_AMD_FIR=f16x3, read by peaq_ctx_create). */
  if (!getenv ("PEAQ_AMD_CLI_STREAM")) {
    /* both files are in memory: one call, every kernel sees the whole stream (a 5-minute pair of the advanced
     * version: 2 s instead of the 4 s of buffer-by-buffer sessions) */
    if (peaq_run_pair (ctx, advanced, ref.channels, level, ref.samples, ref.frames, test.samples, test.frames, &r) !=
        PEAQ_OK) {
      printf ("Error: %s\n", peaq_last_error ());
      return 2;
    }
  } else {
    /* PEAQ_AMD_CLI_STREAM=1: feed a session like two streaming threads would, alternating buffers of 4096
     * samples (what the `peaq` element does) */
    if (peaq_session_create (ctx, advanced, ref.channels, level, &s) != PEAQ_OK) {
      printf ("Error: peaq engine could not be instantiated - %s\n", peaq_last_error ());
      return 2;
    }
    for (pos = 0; pos < ref.frames || pos < test.frames; pos += 4096) {
      rc = PEAQ_OK;
      if (pos < ref.frames)
        rc = peaq_session_push (s, 0, ref.samples + pos * ref.channels,
            ref.frames - pos < 4096 ? ref.frames - pos : 4096);
      if (rc == PEAQ_OK && pos < test.frames)
        rc = peaq_session_push (s, 1, test.samples + pos * test.channels,
            test.frames - pos < 4096 ? test.frames - pos : 4096);
      if (rc != PEAQ_OK) {
        printf ("Error: %s\n", peaq_last_error ());
        return 2;
      }
    }
    if (peaq_session_flush (s) != PEAQ_OK || peaq_session_results (s, &r) != PEAQ_OK) {
      printf ("Error: %s\n", peaq_last_error ());
      return 2;
    }
    peaq_session_destroy (s);
  }
  printf ("Objective Difference Grade: %.3f\n", r.odg);
  printf ("Distortion Index: %.3f\n", r.di);
  peaq_ctx_destroy (ctx);
  free (ref.samples);
  free (test.samples);
  return 0;
}
