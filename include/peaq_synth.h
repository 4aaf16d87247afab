/* peaq_synth.h -- deterministic synthetic (reference, test) signal pairs.
 *
 * Header-only, pure 32-bit integer arithmetic, random access in the sample
 * index: the same bits come out of gcc on the host, hipcc on gfx950 and the
 * numpy transcription in tests/synth_np.py.  This is the workload generator
 * SURVEY.md 8(d) asks for ("deterministic from a seed via an integer PRNG
 * defined in the repo"); it is not part of the PEAQ algorithm and has no
 * counterpart in the reference (the reference is fed by GStreamer sources).
 *
 * Signal model (48 kHz, F32 in [-1,1), value = integer / 2^23):
 *   noise  w_c[n]  : 15-bit uniform from a counter hash of (pair seed, channel, n)
 *   tones  t_c[n]  : three parabolic pseudo-sines (integer phase accumulators)
 *   ref_c[n]  = env[n] * ( LP18k(w_c)[n] * gn + t_c[n] )
 *   test_c[n] = Q_bits( gain * env[n] * ( LPfc(w_c)[n] * gn + t_c[n] ) )
 * with fc in {16,14,12,10} kHz, bits in 8..15 and gain = (256+dg)/256 chosen by
 * the seed, env a slow triangular amplitude modulation, and optional leading /
 * trailing digital silence (exercises the accumulators' INIT / TENTATIVE
 * states, reference movaccum.c:317-362).  The reference is band limited to
 * 18 kHz so that the Bandwidth MOVs are defined (the reference C code returns
 * ODG = NaN when no frame has a reference bandwidth above bin 346,
 * movs.c:797 -- SURVEY.md Appendix B.6).
 */
#ifndef PEAQ_SYNTH_H
#define PEAQ_SYNTH_H

#include <stdint.h>

#ifndef PEAQ_SYNTH_FN
#  if defined(__HIPCC__)
#    define PEAQ_SYNTH_FN static inline __host__ __device__
#  else
#    define PEAQ_SYNTH_FN static inline
#  endif
#endif

#define PEAQ_SYNTH_NTAPS 63

/* tools/design_synth_filters.py: Kaiser(7) windowed-sinc, DC gain 2^15.
 * row 0 = reference (18 kHz), rows 1..4 = test (16, 14, 12, 10 kHz). */
static const int16_t peaq_synth_taps[5][PEAQ_SYNTH_NTAPS] = {
  {-1, 4, -5, 0, 11, -23, 22, 0, -39, 72, -65, 0, 101, -176, 151, 0, -220, 372, -313, 0, 440, -738, 620, 0, -891, 1538, -1355, 0, 2385, -5145, 7351, 24576, 7351, -5145, 2385, 0, -1355, 1538, -891, 0, 620, -738, 440, 0, -313, 372, -220, 0, 151, -176, 101, 0, -65, 72, -39, 0, 22, -23, 11, 0, -5, 4, -1},
  {2, 0, -6, 9, 0, -20, 27, 0, -48, 62, 0, -100, 124, 0, -185, 224, 0, -322, 383, 0, -539, 639, 0, -907, 1092, 0, -1660, 2139, 0, -4456, 9002, 21848, 9002, -4456, 0, 2139, -1660, 0, 1092, -907, 0, 639, -539, 0, 383, -322, 0, 224, -185, 0, 124, -100, 0, 62, -48, 0, 27, -20, 0, 9, -6, 0, 2},
  {1, -4, 2, 9, -11, -11, 30, 0, -54, 36, 65, -100, -37, 176, -55, -224, 220, 186, -428, 0, 602, -369, -620, 907, 326, -1538, 496, 2139, -2385, -2573, 10041, 19114, 10041, -2573, -2385, 2139, 496, -1538, 326, 907, -620, -369, 602, 0, -428, 186, 220, -224, -55, 176, -37, -100, 65, 36, -54, 0, 30, -11, -11, 9, 2, -4, 1},
  {-2, 0, 7, 0, -16, 0, 32, 0, -56, 0, 92, 0, -143, 0, 214, 0, -311, 0, 443, 0, -623, 0, 877, 0, -1261, 0, 1917, 0, -3373, 0, 10395, 16384, 10395, 0, -3373, 0, 1917, 0, -1261, 0, 877, 0, -623, 0, 443, 0, -311, 0, 214, 0, -143, 0, 92, 0, -56, 0, 32, 0, -16, 0, 7, 0, -2},
  {1, 4, 2, -9, -11, 11, 30, 0, -54, -36, 65, 100, -37, -176, -55, 224, 220, -186, -428, 0, 602, 369, -620, -907, 326, 1538, 496, -2139, -2385, 2572, 10040, 13654, 10040, 2572, -2385, -2139, 496, 1538, 326, -907, -620, 369, 602, 0, -428, -186, 220, 224, -55, -176, -37, 100, 65, -36, -54, 0, 30, 11, -11, -9, 2, 4, 1},
};

/* murmur3 32-bit finaliser: a bijective mixer, all ops wrap mod 2^32 */
PEAQ_SYNTH_FN uint32_t peaq_synth_mix32 (uint32_t x)
{
  x ^= x >> 16; x *= 0x85ebca6bu;
  x ^= x >> 13; x *= 0xc2b2ae35u;
  x ^= x >> 16;
  return x;
}

/* Per-pair parameters, all derived from the 32-bit pair seed. */
typedef struct {
  uint32_t seed;
  uint32_t chan_key[2];   /* noise stream key per channel */
  uint32_t tone_inc[3];   /* phase increment per sample (2^32 = one period) */
  uint32_t tone_ph0[2][3];/* start phase per channel */
  int32_t  tone_amp[3];   /* Q15 amplitude of each tone */
  int32_t  noise_gain;    /* Q15 */
  uint32_t lfo_inc;       /* AM rate */
  int32_t  lfo_depth;     /* Q15 */
  int32_t  test_filter;   /* row 1..4 of peaq_synth_taps */
  int32_t  test_shift;    /* quantiser: drop this many LSBs (of the 2^-23 grid) */
  int32_t  test_gain;     /* Q8, 248..264 */
  uint32_t lead_silence;  /* samples of digital silence at the start */
  uint32_t tail_silence;  /* ... and before n_samples */
} peaq_synth_params;

PEAQ_SYNTH_FN void peaq_synth_init (peaq_synth_params *p, uint32_t seed)
{
  uint32_t h = peaq_synth_mix32 (seed * 0x9e3779b9u + 0x7f4a7c15u);
  int i, c;
  p->seed = seed;
  for (c = 0; c < 2; c++)
    p->chan_key[c] = peaq_synth_mix32 (h + 0x632be5abu * (uint32_t) (c + 1));
  for (i = 0; i < 3; i++) {
    uint32_t r = peaq_synth_mix32 (h ^ (0x1000193u * (uint32_t) (i + 1)));
    /* 150 Hz .. ~6.3 kHz : inc = f / 48000 * 2^32 ; 1 Hz = 89478.49 */
    uint32_t f_hz = 150u + (r % 6144u);
    p->tone_inc[i] = f_hz * 89478u + (r >> 20);
    p->tone_amp[i] = (int32_t) (1024u + ((r >> 8) % 3072u));  /* -30 .. -18 dBFS */
    for (c = 0; c < 2; c++)
      p->tone_ph0[c][i] = peaq_synth_mix32 (r + 77u * (uint32_t) (c + 1));
  }
  {
    uint32_t r = peaq_synth_mix32 (h ^ 0xdeadbeefu);
    p->noise_gain  = (int32_t) (4096u + (r % 12288u));        /* Q15: -18 .. -6 dB on a -12 dBFS noise */
    p->lfo_inc     = (2u + ((r >> 13) % 14u)) * 89478u;       /* 2 .. 15 Hz */
    p->lfo_depth   = (int32_t) ((r >> 17) % 16384u);          /* Q15: 0 .. 0.5 */
    p->test_filter = 1 + (int32_t) ((r >> 3) % 4u);
  }
  {
    uint32_t r = peaq_synth_mix32 (h ^ 0x0badf00du);
    p->test_shift  = 23 - (8 + (int32_t) (r % 8u));           /* 8 .. 15 bit */
    p->test_gain   = 248 + (int32_t) ((r >> 4) % 17u);
    /* one pair in four starts and/or ends with 0.25 s of digital silence */
    p->lead_silence = ((r >> 10) % 4u == 0u) ? 12000u : 0u;
    p->tail_silence = ((r >> 12) % 4u == 0u) ? 12000u : 0u;
  }
}

PEAQ_SYNTH_FN int32_t peaq_synth_noise (uint32_t key, int64_t n)
{
  if (n < 0)
    return 0;
  return (int32_t) (peaq_synth_mix32 (key + (uint32_t) n * 0x9e3779b1u) >> 17) - 16384;
}

/* parabolic pseudo-sine of a 32-bit phase, output in [-32768, 32767] */
PEAQ_SYNTH_FN int32_t peaq_synth_psin (uint32_t phase)
{
  int32_t x = (int32_t) phase >> 16;            /* -32768 .. 32767 */
  int32_t ax = x < 0 ? -x : x;
  return (x * (32768 - ax)) >> 13;              /* 4 x (1-|x|) in Q15 */
}

/* One sample of (ref, test) for channel c at index n, as integers on the
 * 2^-23 grid.  Cost: 63 hashes + 126 MACs (callers that generate runs of
 * samples should use peaq_synth_block below, which shares the noise). */
PEAQ_SYNTH_FN void peaq_synth_sample_from_noise (const peaq_synth_params *p, int c,
                                                 uint32_t n, uint32_t n_samples,
                                                 const int32_t *w /* w[k] = noise(n-k), k<63 */,
                                                 int32_t *ref_out, int32_t *test_out)
{
  int k, i;
  int32_t acc_r = 0, acc_t = 0;
  int32_t tone = 0, lfo, env, vr, vt;
  if (n < p->lead_silence || n + p->tail_silence >= n_samples) {
    *ref_out = 0;
    *test_out = 0;
    return;
  }
  for (k = 0; k < PEAQ_SYNTH_NTAPS; k++) {
    acc_r += (int32_t) peaq_synth_taps[0][k] * w[k];              /* |acc| < 2^31: sum|h| < 2^16.1, |w| <= 2^14 */
    acc_t += (int32_t) peaq_synth_taps[p->test_filter][k] * w[k];
  }
  /* Q15 taps * 15-bit noise -> >>15 gives 15-bit; apply Q15 gain */
  acc_r = ((acc_r >> 15) * p->noise_gain) >> 15;
  acc_t = ((acc_t >> 15) * p->noise_gain) >> 15;
  for (i = 0; i < 3; i++)
    tone += (peaq_synth_psin (p->tone_ph0[c][i] + n * p->tone_inc[i]) * p->tone_amp[i]) >> 15;
  lfo = peaq_synth_psin (n * p->lfo_inc);                          /* +-2^15 */
  env = 32768 - p->lfo_depth + ((lfo * p->lfo_depth) >> 15);       /* Q15, 1-2d .. 1 */
  vr = ((acc_r + tone) * env) >> 15;                               /* 16-bit scale */
  vt = ((acc_t + tone) * env) >> 15;
  vt = (vt * p->test_gain) >> 8;
  /* to the 2^-23 grid (full scale 2^15 -> 2^23) and quantise the test signal */
  vr *= 256;
  vt *= 256;
  vt = ((vt + (1 << (p->test_shift - 1))) >> p->test_shift) * (1 << p->test_shift);
  *ref_out = vr;
  *test_out = vt;
}

PEAQ_SYNTH_FN float peaq_synth_to_float (int32_t v)
{
  return (float) v * (1.0f / 8388608.0f);   /* exact: |v| < 2^24 */
}

/* Fill interleaved F32 buffers ref/test[n_samples][channels] for one pair
 * (host side convenience; the device generator in csrc/ calls the same
 * per-sample function). */
PEAQ_SYNTH_FN void peaq_synth_pair (uint32_t seed, int channels, uint32_t n_samples,
                                    float *ref, float *test)
{
  peaq_synth_params p;
  int32_t w[PEAQ_SYNTH_NTAPS];
  int c, k;
  uint32_t n;
  peaq_synth_init (&p, seed);
  for (c = 0; c < channels; c++) {
    for (k = 0; k < PEAQ_SYNTH_NTAPS; k++)
      w[k] = 0;
    for (n = 0; n < n_samples; n++) {
      int32_t r, t;
      for (k = PEAQ_SYNTH_NTAPS - 1; k > 0; k--)
        w[k] = w[k - 1];
      w[0] = peaq_synth_noise (p.chan_key[c], (int64_t) n);
      peaq_synth_sample_from_noise (&p, c, n, n_samples, w, &r, &t);
      ref[(size_t) n * channels + c] = peaq_synth_to_float (r);
      test[(size_t) n * channels + c] = peaq_synth_to_float (t);
    }
  }
}

#endif /* PEAQ_SYNTH_H */
