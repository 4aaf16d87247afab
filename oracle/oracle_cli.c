/* oracle_cli.c -- command-line front end of the CPU oracle.  TEST INFRASTRUCTURE.
 *
 *   oracle_cli pair  ADV CH REF.f32 TEST.f32   raw interleaved F32LE files
 *   oracle_cli synth ADV CH SEED NSAMPLES      include/peaq_synth.h pair
 *   oracle_cli time  ADV CH SEED0 NPAIRS NSAMPLES [REPEATS START_EPOCH]   wall-clock of NPAIRS pairs (1 thread)
 * Prints the same JSON shape as oracle/ref_harness.c so the two can be diffed.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <stdint.h>
#include "peaq_oracle.h"
#include "../include/peaq_synth.h"

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
  buf = malloc (sz > 0 ? sz : 4);
  if (fread (buf, 1, sz, f) != (size_t) sz) { perror ("read"); exit (2); }
  fclose (f);
  *n = sz / sizeof (float);
  return buf;
}

static void
report (int adv, int ch, const float *r, size_t nr, const float *t, size_t nt)
{
  double movs[11], di, odg, snr;
  orc_session *s = orc_session_new (adv, ch, 92.);
  int n;
  /* feed in 1024-sample chunks alternating pads, like two streaming threads */
  size_t pr = 0, pt = 0, chunk = 1024;
  while (pr < nr || pt < nt) {
    size_t c = nr - pr < chunk ? nr - pr : chunk;
    if (c) { orc_session_push_ref (s, r + pr * ch, c); pr += c; }
    c = nt - pt < chunk ? nt - pt : chunk;
    if (c) { orc_session_push_test (s, t + pt * ch, c); pt += c; }
  }
  orc_session_flush (s);
  orc_session_results (s, movs, &di, &odg);
  snr = orc_session_totalsnr (s);
  n = orc_session_mov_count (s);
  printf ("{\"advanced\": %d, \"channels\": %d, \"frames\": %u, ", adv, ch, orc_session_frames (s));
  print_arr ("movs", movs, n, 0);
  print_arr ("di", &di, 1, 0);
  print_arr ("odg", &odg, 1, 0);
  print_arr ("totalsnr", &snr, 1, 1);
  printf ("}\n");
  orc_session_free (s);
}

int
main (int argc, char **argv)
{
  if (argc >= 6 && !strcmp (argv[1], "pair")) {
    int adv = atoi (argv[2]), ch = atoi (argv[3]);
    size_t nr, nt;
    float *r = read_f32 (argv[4], &nr), *t = read_f32 (argv[5], &nt);
    report (adv, ch, r, nr / ch, t, nt / ch);
    return 0;
  }
  if (argc >= 6 && !strcmp (argv[1], "synth")) {
    int adv = atoi (argv[2]), ch = atoi (argv[3]);
    uint32_t seed = (uint32_t) strtoul (argv[4], NULL, 0), ns = (uint32_t) strtoul (argv[5], NULL, 0);
    float *r = malloc ((size_t) ns * ch * 4), *t = malloc ((size_t) ns * ch * 4);
    peaq_synth_pair (seed, ch, ns, r, t);
    report (adv, ch, r, ns, t, ns);
    return 0;
  }
  if (argc >= 7 && !strcmp (argv[1], "time")) {
    /* same arguments and JSON shape as `ref_harness time` (see there) */
    int adv = atoi (argv[2]), ch = atoi (argv[3]);
    uint32_t seed0 = (uint32_t) strtoul (argv[4], NULL, 0);
    int np = atoi (argv[5]), p, rep;
    uint32_t ns = (uint32_t) strtoul (argv[6], NULL, 0);
    int repeats = argc >= 9 ? atoi (argv[7]) : 0;
    double start_epoch = argc >= 9 ? atof (argv[8]) : 0.;
    size_t per = (size_t) ns * ch;
    float *r = malloc (per * 4 * (repeats ? np : 1)), *t = malloc (per * 4 * (repeats ? np : 1));
    double *movs = calloc (11 * (size_t) np, sizeof (double)), *di = malloc (sizeof (double) * np),
      *odg = malloc (sizeof (double) * np);
    double total = 0., t_begin = 0., t_end = 0.;
    unsigned frames = 0;
    struct timespec a, b;
    for (p = 0; p < np; p++) {
      orc_session *s;
      float *rp = r + (repeats ? p * per : 0), *tp = t + (repeats ? p * per : 0);
      peaq_synth_pair (seed0 + p, ch, ns, rp, tp);      /* generation is not timed */
      if (repeats)
        continue;
      clock_gettime (CLOCK_MONOTONIC, &a);
      s = orc_session_new (adv, ch, 92.);
      orc_session_push_ref (s, rp, ns);
      orc_session_push_test (s, tp, ns);
      orc_session_flush (s);
      orc_session_results (s, movs + 11 * p, di + p, odg + p);
      clock_gettime (CLOCK_MONOTONIC, &b);
      frames += orc_session_frames (s);
      orc_session_free (s);
      total += (b.tv_sec - a.tv_sec) + 1e-9 * (b.tv_nsec - a.tv_nsec);
    }
    if (repeats) {
      do
        clock_gettime (CLOCK_REALTIME, &a);
      while (a.tv_sec + 1e-9 * a.tv_nsec < start_epoch);
      t_begin = a.tv_sec + 1e-9 * a.tv_nsec;
      for (rep = 0; repeats > 0 ? rep < repeats : rep == 0 || (clock_gettime (CLOCK_REALTIME, &b), b.tv_sec + 1e-9 * b.tv_nsec < t_begin - repeats); rep++)
        for (p = 0; p < np; p++) {
          double m[11], d, o;
          orc_session *s = orc_session_new (adv, ch, 92.);
          orc_session_push_ref (s, r + p * per, ns);
          orc_session_push_test (s, t + p * per, ns);
          orc_session_flush (s);
          orc_session_results (s, rep ? m : movs + 11 * p, rep ? &d : di + p, rep ? &o : odg + p);
          frames += orc_session_frames (s);
          orc_session_free (s);
        }
      clock_gettime (CLOCK_REALTIME, &b);
      t_end = b.tv_sec + 1e-9 * b.tv_nsec;
      total = t_end - t_begin;
    }
    printf ("{\"pairs\": %d, \"repeats\": %d, \"frame_pairs\": %u, \"seconds\": %.6f, \"frame_pairs_per_s\": %.1f, "
            "\"t_begin\": %.6f, \"t_end\": %.6f, \"n_movs\": %d, ", np, repeats, frames, total, frames / total,
            t_begin, t_end, adv ? 5 : 11);
    print_arr ("odg", odg, np, 0);
    print_arr ("di", di, np, 0);
    print_arr ("movs", movs, 11 * np, 1);
    printf ("}\n");
    return 0;
  }
  fprintf (stderr, "usage: see the header of oracle/oracle_cli.c\n");
  return 1;
}
