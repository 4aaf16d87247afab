// peaq_tables.cpp -- constant tables of the PEAQ ear models, built once per
// context on the host in FP64 and uploaded to HBM.
//
// What the reference computes at object-construction time:
//   Hann window            fftearmodel.c:160-173
//   outer/middle ear       earmodel.c:702-709, fftearmodel.c:249-256
//   band edges + weights   fftearmodel.c:701-760
//   spreading constants    fftearmodel.c:723-725,764-767,778-781
//   masking offsets        fftearmodel.c:770-772
//   per-band constants     earmodel.c:279-323, 627-635
//   filter-bank responses  fbearmodel.c:57-61,182-225
//   EHS window             movs.c:1360-1368
#include "peaq_tables.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace peaq {

static const double kPi = 3.14159265358979323846;
static const double kFs = 48000.0;

static double ear_weight_db_to_lin(double f_hz) {
  // W(f) of BS.1387 (7) / Kabal (6); earmodel.c:702-709
  const double f = f_hz / 1000.0;
  const double w_db = -0.6 * 3.64 * std::pow(f, -0.8) + 6.5 * std::exp(-0.6 * std::pow(f - 3.3, 2.0)) -
                      1e-3 * std::pow(f, 3.6);
  return std::pow(10.0, w_db / 20.0);
}

static double smoothing_coeff(double fc, int step, double tau_min, double tau_100) {
  // earmodel.c:627-635
  const double tau = tau_min + 100.0 / fc * (tau_100 - tau_min);
  return std::exp(step / (-48000.0 * tau));
}

static void fill_common_bands(BandTables& t, const std::vector<double>& fc, int step, double loudness_scale,
                              double tau_min, double tau_100) {
  t.bands = static_cast<int>(fc.size());
  t.step = step;
  t.deriv_factor = kFs / step;
  for (int i = 0; i < t.bands; ++i) {
    const double f = fc[i];
    t.fc[i] = f;
    t.internal_noise[i] = std::pow(10.0, 0.4 * 0.364 * std::pow(f / 1000.0, -0.8));
    t.noise_pow03[i] = std::pow(t.internal_noise[i], 0.3);
    t.ln_internal_noise[i] = std::log(t.internal_noise[i]);
    {
      // leveladapter.c:315-328: the correction factors are averaged over [k - m1, k + m2]
      const int M1 = t.bands / 36, M2 = t.bands / 25;
      const int m1 = i < M1 ? i : M1, m2 = (t.bands - i - 1) < M2 ? (t.bands - i - 1) : M2;
      t.inv_window_count[i] = 1.0 / (m1 + m2 + 1);
    }
    t.exc_threshold[i] = std::pow(10.0, 0.364 * std::pow(f / 1000.0, -0.8));
    t.threshold[i] =
        std::pow(10.0, 0.1 * (-2.0 - 2.05 * std::atan(f / 4000.0) - 0.75 * std::atan(f / 1600.0 * f / 1600.0)));
    t.loud_factor[i] = loudness_scale * std::pow(t.exc_threshold[i] / (1e4 * t.threshold[i]), 0.23);
    t.ear_tc[i] = smoothing_coeff(f, step, tau_min, tau_100);
    t.adapt_tc[i] = smoothing_coeff(f, step, 0.008, 0.05);  // leveladapter.c:205, modpatt.c:185
  }
  // neutral padding so that idle lanes never produce NaN/Inf
  for (int i = t.bands; i < kBandStride; ++i) {
    t.fc[i] = 1000.0;
    t.internal_noise[i] = 1.0;
    t.noise_pow03[i] = 1.0;
    t.ln_internal_noise[i] = 0.0;
    t.inv_window_count[i] = 1.0;
    t.exc_threshold[i] = 1.0;
    t.threshold[i] = 0.5;
    t.loud_factor[i] = 0.0;
    t.ear_tc[i] = 0.0;
    t.adapt_tc[i] = 0.0;
  }
}

// Reference do_spreading (fftearmodel.c:637-676) evaluated on an all-ones
// pattern; only used to derive the normalisation table.
static void spread_reference_order(const BandTables& t, const std::vector<double>& aUC, const std::vector<double>& pp,
                                   std::vector<double>& e2) {
  const int nb = t.bands;
  std::vector<double> up(nb), en(nb);
  for (int i = 0; i < nb; ++i) {
    const double a = aUC[i] * std::pow(pp[i], 0.2 * t.delta_z);
    const double giu = (1.0 - std::pow(a, nb - i)) / (1.0 - a);
    const double e = pp[i] / (t.gIL[i] + giu - 1.0);
    up[i] = std::pow(a, 0.4);
    en[i] = std::pow(e, 0.4);
  }
  e2.assign(nb, 0.0);
  e2[nb - 1] = en[nb - 1];
  for (int i = nb - 1; i > 0; --i) e2[i - 1] = t.aLe * e2[i] + en[i - 1];
  for (int i = 0; i < nb - 1; ++i) {
    double r = en[i];
    for (int j = i + 1; j < nb; ++j) {
      r *= up[i];
      e2[j] += r;
    }
  }
  for (int i = 0; i < nb; ++i) e2[i] = std::pow(e2[i], 2.5);
}

void fill_log_tab(double (*tab)[2]);

void build_fft_band_tables(int bands, BandTables& t) {
  std::memset(&t, 0, sizeof t);
  const int n = kFrame;
  t.delta_z = 27.0 / (bands - 1);
  t.dz02 = 0.2 * t.delta_z;
  const double a_l = std::pow(10.0, -2.7 * t.delta_z);  // 1/a_L
  t.aLe = std::pow(a_l, 0.4);
  const double z_lo = 7.0 * std::asinh(80.0 / 650.0);
  const double z_hi = 7.0 * std::asinh(18000.0 / 650.0);
  std::vector<double> fc(bands), aUC(bands);
  for (int i = 0; i < bands; ++i) {
    const double zl = z_lo + i * t.delta_z;
    const double zu = std::fmin(z_hi, z_lo + (i + 1) * t.delta_z);
    const double fl = 650.0 * std::sinh(zl / 7.0);
    const double fu = 650.0 * std::sinh(zu / 7.0);
    fc[i] = 650.0 * std::sinh((zl + zu) / 2.0 / 7.0);
    t.lo[i] = static_cast<int>(std::round(fl / kFs * n));
    t.hi[i] = static_cast<int>(std::round(fu / kFs * n));
    // fraction of the edge bins that falls inside the band (Kabal 2.6)
    double edge = (2 * t.lo[i] + 1) / 2.0 * kFs / n;
    if (edge > fu) edge = fu;
    t.wlo[i] = (edge - fl) * n / kFs;
    if (t.lo[i] == t.hi[i]) {
      t.whi[i] = 0.0;
    } else {
      edge = (2 * t.hi[i] - 1) / 2.0 * kFs / n;
      t.whi[i] = (fu - edge) * n / kFs;
    }
    aUC[i] = std::pow(10.0, (-2.4 - 23.0 / fc[i]) * t.delta_z);
    t.ln_aUC[i] = std::log(aUC[i]);
    t.gIL[i] = (1.0 - std::pow(a_l, i + 1)) / (1.0 - a_l);
    t.mask_diff[i] = std::pow(10.0, (i * t.delta_z <= 12.0 ? 3.0 : 0.25 * i * t.delta_z) / 10.0);
  }
  fill_common_bands(t, fc, n / 2, 1.07664, 0.008, 0.030);  // fftearmodel.c:53,224-228
  std::vector<double> ones(bands, 1.0), norm;
  spread_reference_order(t, aUC, ones, norm);
  for (int i = 0; i < bands; ++i) {
    t.inv_spread_norm[i] = 1.0 / norm[i];
    t.inv_spread_norm_pow03[i] = std::pow(t.inv_spread_norm[i], 0.3);
  }
  for (int i = bands; i < kBandStride; ++i) {
    t.lo[i] = t.hi[i] = 0;
    t.wlo[i] = t.whi[i] = 0.0;
    t.ln_aUC[i] = -1.0;
    t.gIL[i] = 1.0;
    t.inv_spread_norm[i] = 0.0;
    t.inv_spread_norm_pow03[i] = 0.0;
    t.mask_diff[i] = 1.0;
  }
}

void build_fb_band_tables(BandTables& t, FbTables& fb) {
  std::memset(&t, 0, sizeof t);
  std::memset(&fb, 0, sizeof fb);
  static const int kLen[kFbBands] = {1456, 1438, 1406, 1362, 1308, 1244, 1176, 1104, 1030, 956, 884, 814, 748, 686,
                                     626,  570,  520,  472,  430,  390,  354,  320,  290,  262, 238, 214, 194, 176,
                                     158,  144,  130,  118,  106,  96,   86,   78,   70,   64,  58,  52};
  std::vector<double> fc(kFbBands);
  int off = 0;
  for (int b = 0; b < kFbBands; ++b) {
    // Kabal (36)-(37) centre frequencies, fbearmodel.c:201-204
    fc[b] = std::sinh(std::asinh(50.0 / 650.0) + b * (std::asinh(18000.0 / 650.0) - std::asinh(50.0 / 650.0)) / 39.0) *
            650.0;
    const int len = kLen[b];
    const double wt = ear_weight_db_to_lin(fc[b]);
    fb.flen[b] = len;
    fb.delay[b] = 1 + (kLen[0] - len) / 2;
    fb.coef_off[b] = off;
    for (int k = 0; k <= len / 2; ++k) {
      const double s = std::sin(kPi * k / len);
      const double win = 4.0 / len * s * s * wt;
      const double ph = 2 * kPi * fc[b] * (k - len / 2.0) / 48000.0;
      fb.h_re[off + k] = win * std::cos(ph);
      fb.h_im[off + k] = win * std::sin(ph);
    }
    off += len / 2 + 1;
  }
  // GEMM form of the bank (peaq_device.h): tap n = 1 .. N/2 of band b sits at delay d = D_b + n and
  // shares its coefficient (conjugated) with the tap at delay 1458 - d.  Band 0's tap at delay 1456
  // reads the NEWEST sample in the reference (doubled ring buffer, fbearmodel.c:413-414); the kernel
  // corrects that one product after the GEMM.
  for (int r = 0; r < kMfTiles; ++r) {
    const int longest = kLen[16 * r];
    if (fb.delay[16 * r] + 1 != kMfD0[r] || (kFbCentre - kMfD0[r]) / 4 + 1 != kMfSteps[r] ||
        fb.delay[16 * r] + longest / 2 != kFbCentre)
      std::abort();                                      // peaq_device.h constants
    for (int s = 0; s < kMfSteps[r]; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int b = 16 * r + (lane & 15), d = kMfD0[r] + 4 * s + (lane >> 4);
        double vr = 0., vi = 0.;
        if (b < kFbBands) {
          const int half = kLen[b] / 2, n = d - fb.delay[b];
          if (n >= 1 && n <= half) {
            vr = fb.h_re[fb.coef_off[b] + n] * (n == half ? 0.5 : 1.);
            vi = n == half ? 0. : fb.h_im[fb.coef_off[b] + n];
          }
        }
        fb.mf_re[(size_t)(kMfBase[r] + s) * 64 + lane] = vr;
        fb.mf_im[(size_t)(kMfBase[r] + s) * 64 + lane] = vi;
        fb.mf_re_f[(size_t)(kMfBase[r] + s) * 64 + lane] = (float)vr;
        fb.mf_im_f[(size_t)(kMfBase[r] + s) * 64 + lane] = (float)vi;
      }
  }
  // the FP16 x 3 form of the same coefficients (peaq_device.h kHf*)
  {
    auto coef = [&](int b, int d, double& vr, double& vi) {          // as above: band b, delay d
      vr = vi = 0.;
      if (b >= kFbBands) return;
      const int half = kLen[b] / 2, n = d - fb.delay[b];
      if (n >= 1 && n <= half) {
        vr = fb.h_re[fb.coef_off[b] + n] * (n == half ? 0.5 : 1.);
        vi = n == half ? 0. : fb.h_im[fb.coef_off[b] + n];
      }
    };
    auto bits = [](_Float16 h) {
      unsigned short u;
      std::memcpy(&u, &h, sizeof u);
      return u;
    };
    for (int r = 0; r < kMfTiles; ++r) {
      if (kHfD1[r] > kMfD0[r] || kHfD1[r] + 32 * kHfBlocks[r] <= kFbCentre || kHfD1[r] % 8 != 1 || kHfD2[r] % 8 != 2 ||
          kHfD2[r] > kMfD0[r] || kHfD2[r] + 32 * kHfBlocks[r] <= kFbCentre)
        std::abort();                                                // the blocks must cover the tile's delays
      for (int i = 0; i < 16; ++i) {
        const int b = 16 * r + i;
        double peak = 0.;
        for (int d = 1; d <= kFbCentre; ++d) {
          double vr, vi;
          coef(b, d, vr, vi);
          peak = std::max(peak, std::max(std::fabs(vr), std::fabs(vi)));
        }
        // largest coefficient of the band in [1024, 2048): far from FP16's 65504, 25 octaves above its smallest normal
        const int e = peak > 0. ? 10 - std::ilogb(peak) : 0;
        fb.hf_unscale[b] = peak > 0. ? std::ldexp(1., -e) : 0.;
        for (int s = 0; s < kHfBlocks[r]; ++s)
          for (int kg = 0; kg < 4; ++kg)
            for (int el = 0; el < 8; ++el) {
              const int lane = i + 16 * kg;
              auto& blk = fb.hf[kHfBase[r] + s];
              for (int which = 0; which < 2; ++which) {
                const int d = which == 0 ? kHfD1[r] + 32 * s + 8 * kg + (7 - el) : kHfD2[r] + 32 * s + 8 * kg + el;
                double vr, vi;
                coef(b, d, vr, vi);
                if (which == 1) vi = -vi;                            // im = Him X1 - Him X2
                const double sr = std::ldexp(vr, e), si = std::ldexp(vi, e);
                const _Float16 rh = (_Float16)sr, ih = (_Float16)si;
                const _Float16 rl = (_Float16)(sr - (double)rh), il = (_Float16)(si - (double)ih);
                blk[which == 0 ? HF_RE_HI_1 : HF_RE_HI_2][lane][el] = bits(rh);
                blk[which == 0 ? HF_RE_LO_1 : HF_RE_LO_2][lane][el] = bits(rl);
                blk[which == 0 ? HF_IM_HI_1 : HF_NIM_HI_2][lane][el] = bits(ih);
                blk[which == 0 ? HF_IM_LO_1 : HF_NIM_LO_2][lane][el] = bits(il);
              }
            }
      }
    }
  }
  // The FP64 engine's tables (peaq_device.h kBs*, kMfd*).  Direct tile first: bands 24 .. 39, as above.
  if (fb.delay[kMfdBand0] + 1 != kMfdD0 || (kFbCentre - kMfdD0) / 4 + 1 != kMfdSteps) std::abort();
  for (int s = 0; s < kMfdSteps; ++s)
    for (int lane = 0; lane < 64; ++lane) {
      const int b = kMfdBand0 + (lane & 15), d = kMfdD0 + 4 * s + (lane >> 4);
      const int half = kLen[b] / 2, n = d - fb.delay[b];
      double vr = 0., vi = 0.;
      if (n >= 1 && n <= half) {
        vr = fb.h_re[fb.coef_off[b] + n] * (n == half ? 0.5 : 1.);
        vi = n == half ? 0. : fb.h_im[fb.coef_off[b] + n];
      }
      fb.mfd_re[(size_t)s * 64 + lane] = vr;
      fb.mfd_im[(size_t)s * 64 + lane] = vi;
    }
  // Block-sum form of bands 0 .. 23.  In window coordinates (sample u of the kernel's window; output t has its
  // filters' centre tap on u = 727 + 32 t, i.e. delay 729) tap m = u - 727 - 32 t of band b carries
  //   h(m) = 4/N cos^2(pi m / N) W e^(-j w m),  |m| < N/2      (fbearmodel.c:214-220 with n = N/2 - m)
  //        = sum_i g_i e^(-j w_i m),  (g_i, w_i) = (2W/N, w), (W/N, w + 2 pi/N), (W/N, w - 2 pi/N).
  {
    const long double pi = 3.14159265358979323846264338327950288L;
    int cL[kBsBands], cR[kBsBands];
    for (int b = 0; b < kBsBands; ++b) {
      const int half = kLen[b] / 2;
      cL[b] = (728 - half) >> 5;
      cR[b] = (726 + half) >> 5;
      fb.bs_left_q0[b] = (728 - half) & 31;
      // (peaq_device.h, bs_left_g: with q0 < 16 the block the window starts in counts as whole)
      fb.bs_whole[b] = cR[b] - 1 - cL[b] + (fb.bs_left_q0[b] < 16 ? 1 : 0);
      if (fb.bs_whole[b] < 1 || fb.bs_whole[b] > kBsHist - kBsHistOrg) std::abort();
    }
    for (int p = 0; p < kBsPairs; ++p) {
      const int b0 = 2 * p, b1 = 2 * p + 1;
      fb.bs_col_head[p] = std::min(cR[b0], cR[b1]) - 1;
      for (int b = b0; b <= b1; ++b) {
        fb.bs_off_enter[b] = cR[b] - 1 - fb.bs_col_head[p];
        // the kernel's tiles span 64 columns for 60 outputs, an edge row sits one column beyond its enter rows
        if (fb.bs_off_enter[b] < 0 || fb.bs_off_enter[b] > 2) std::abort();
      }
    }
    for (int b = 0; b < kBsBands; ++b) {
      const int n_taps = kLen[b], half = n_taps / 2, p = b / 2, sub = b & 1;
      const long double w0 = 2 * pi * (long double)fc[b] / 48000.0L, dw = 2 * pi / n_taps;
      const long double wi[3] = {w0, w0 + dw, w0 - dw};
      const long double wt = ear_weight_db_to_lin(fc[b]);
      const long double gi[3] = {2 * wt / n_taps, wt / n_taps, wt / n_taps};
      // the filter's own coefficient at window offset m (re, im), from the table the direct form uses
      auto own = [&](int m, double& vr, double& vi) {
        const int n = half - std::abs(m);
        vr = vi = 0.;
        if (n < 1) return;                               // beyond the window (h(N/2 - 0) = h(0) = 0 as well)
        vr = fb.h_re[fb.coef_off[b] + n];
        vi = m == 0 ? 0. : (m > 0 ? fb.h_im[fb.coef_off[b] + n] : -fb.h_im[fb.coef_off[b] + n]);
      };
      fb.bs_col_left[b] = cL[b];
      const int q0 = fb.bs_left_q0[b];
      const bool counted_whole = q0 < 16;
      fb.bs_left_g[b][0] = counted_whole ? 0 : q0 >> 3;
      fb.bs_left_g[b][1] = counted_whole ? (q0 + 7) >> 3 : 4;
      for (int q = 0; q < 32; ++q) {
        double row[8];
        for (int i = 0; i < 3; ++i) {                      // a block enters the running sums as column cR - 1
          const long double ph = wi[i] * (long double)(32 * (cR[b] - 1) + q - 727);
          row[2 * i] = (double)(gi[i] * std::cos(ph));
          row[2 * i + 1] = (double)(-gi[i] * std::sin(ph));
        }
        own(32 * cR[b] + q - 727, row[6], row[7]);
        for (int ty = 0; ty < 8; ++ty) fb.bs_coef[p][q / 4][bs_row(sub, ty) + 16 * (q & 3)] = row[ty];
        if (!counted_whole) {
          own(32 * cL[b] + q - 727, fb.bs_left[b][q][0], fb.bs_left[b][q][1]);
        } else {                                          // minus the exponentials' sum where the window has ended
          long double sr_ = 0, si_ = 0;
          for (int i = 0; i < 3; ++i) {
            const long double ph = wi[i] * (long double)(32 * cL[b] + q - 727);
            sr_ += gi[i] * std::cos(ph);
            si_ -= gi[i] * std::sin(ph);
          }
          fb.bs_left[b][q][0] = q < q0 ? (double)-sr_ : 0.;
          fb.bs_left[b][q][1] = q < q0 ? (double)-si_ : 0.;
        }
      }
      for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 2; ++k) {
          const long double ph = 32.0L * wi[i] * (long double)(k == 0 ? 1 : fb.bs_whole[b]);
          fb.bs_rot[3 * b + i][k][0] = (double)std::cos(ph);
          fb.bs_rot[3 * b + i][k][1] = (double)std::sin(ph);
        }
        for (int l = 0; l < 64; ++l) {
          const long double ph = 32.0L * wi[i] * (long double)(l + 1);
          fb.bs_pow[3 * b + i][l][0] = (double)std::cos(ph);
          fb.bs_pow[3 * b + i][l][1] = (double)std::sin(ph);
        }
        // the lane-per-(chain, segment) evaluation of the running sums (peaq_device.h, kBsSeg)
        const int J = fb.bs_whole[b], sg = (kBsSeg - J % kBsSeg) % kBsSeg, c = 3 * sub + i;
        fb.bs_seg_s[b] = sg;
        for (int l = 0; l < 64; ++l) {
          if (std::min(l >> 3, 5) != c) continue;
          const int g = l & 7;
          auto put = [&](int e, long double steps, bool zero) {
            const long double ph = 32.0L * wi[i] * steps;
            fb.bs_seg[p][e][l][0] = zero ? 0. : (double)std::cos(ph);
            fb.bs_seg[p][e][l][1] = zero ? 0. : (double)std::sin(ph);
          };
          for (int lv = 0; lv < 3; ++lv) put(lv, (long double)(kBsSeg << lv), g < (1 << lv));
          put(3, (long double)(kBsSeg * g - sg), false);
          put(4, 1.0L, false);
          put(5, (long double)J, false);
        }
      }
    }
  }
  for (int k = 0; k < 6; ++k) {
    const double c = std::cos(kPi * (k - 5.0) / 12.0);
    fb.back_mask[k] = c * c * 0.9761 / 6.0;     // what this host's libm gives; the kernel's compile-time copies
  }                                              // (kBackMask, peaq_device.h) are compared in fb_tables_selfcheck()
  fill_log_tab(fb.log_tab);
  fill_common_bands(t, fc, kFbFrame, 1.26539, 0.004, 0.020);  // fbearmodel.c:171-177
  t.delta_z = 0.0;
}

void build_common_tables(CommonTables& c) {
  const int n = kFrame;
  for (int l = 0; l < 64; ++l)
    for (int j = 0; j < 2; ++j) {
      const double th = 2 * kPi * (2 * l + j) / (n - 1), amp = std::sqrt(8.0 / 3.0) * 0.5;
      c.hann_lane[l][2 * j] = amp * std::cos(th);
      c.hann_lane[l][2 * j + 1] = amp * std::sin(th);
    }
  for (int k = 0; k < n; ++k) {
    c.tw_re[k] = std::cos(2 * kPi * k / n);
    c.tw_im[k] = -std::sin(2 * kPi * k / n);
  }
  // exact values on the axes keep the transform of symmetric inputs clean
  c.tw_re[0] = 1.0;  c.tw_im[0] = 0.0;
  c.tw_re[n / 4] = 0.0;  c.tw_im[n / 4] = -1.0;
  c.tw_re[n / 2] = -1.0;  c.tw_im[n / 2] = 0.0;
  c.tw_re[3 * n / 4] = 0.0;  c.tw_im[3 * n / 4] = 1.0;
  for (int k = 0; k <= n / 2; ++k) {
    const double w = ear_weight_db_to_lin(static_cast<double>(k) * kFs / n);
    c.ear_w2[k] = w * w;
  }
  {
    const int stride[8] = {1, 2, 8, 4, 32, 128, 32, 8}, mask[8] = {63, 63, 15, 63, 7, 3, 15, 63};
    for (int e = 0; e < 8; ++e)
      for (int l = 0; l < 64; ++l) {
        const int k = stride[e] * (l & mask[e]);
        c.tw_lane[e][l][0] = c.tw_re[k];
        c.tw_lane[e][l][1] = c.tw_im[k];
      }
    for (int q = 0; q < 8; ++q)
      for (int l = 0; l < 64; ++l) {
        const int k = l + 64 * q;
        const int km = (l == 0 && q == 0) ? 512 : 1024 - k;       // spec_bin(8 + q, l) in peaq_frontend.hip
        c.ear_w2_pair[q][l][0] = c.ear_w2[k];
        c.ear_w2_pair[q][l][1] = c.ear_w2[km];
      }
  }
  for (int i = 0; i < 256; ++i) {  // movs.c:1362-1368: the shipped window, and the one centred at lag zero
    c.ehs_window[i] = 0.81649658092773 * (1.0 - std::cos(2 * kPi * i / 255.0)) / 256.0;
    c.ehs_window_centred[i] = 0.81649658092773 * (1.0 + std::cos(2 * kPi * i / 511.0)) / 256.0;
  }
  fill_log_tab(c.log_tab);
  for (int j = 0; j < kExpTabEntries; ++j) c.exp_tab[j] = (double)exp2l(j / 64.0L);
}

void fill_log_tab(double (*tab)[2]) {                 // log_tab (peaq_device.h), 130 entries
  for (int i = 0; i < 130; ++i) {
    const long double centre = 1.0L + i / 128.0L, ln2 = 0.693147180559945309417232121458176568L;
    tab[i][0] = (double)(2.0L / centre);
    // the centre that the ROUNDED reciprocal stands for: r = fma(m, [i][0], -1) is then exact with respect to it,
    // and the only rounding left in ln m = log1p(r) + ln C - ln 2 is that of the entry itself
    const long double c_eff = 2.0L / (long double)tab[i][0];
    // the lower bins count one binade less in e instead of carrying - ln 2
    tab[i][1] = (double)(i < kLogTabFold ? std::log(c_eff) : std::log(c_eff) - ln2);
  }
  tab[128][1] = 0.;                                   // ln 2 - ln 2 (the long-double difference is 0 anyway)
}

// Self-check of the FP64 engine's filter-bank tables on the host (no device involved; tests/test_capi_host.py):
// the block-sum form of bands 0 .. 23 and the direct tile of bands 24 .. 39, evaluated from FbTables on a window of
// pseudo-random samples, against the plain sums of fbearmodel.c:399-435 over the same coefficients
// (FbTables::h_re / h_im).  Returns the largest deviation relative to the largest output of its band.
double fb_tables_selfcheck() {
  std::vector<BandTables> bt(1);
  std::vector<FbTables> fbv(1);
  build_fb_band_tables(bt[0], fbv[0]);
  const FbTables& fb = fbv[0];
  constexpr int kOut = 60, kCols = 112;
  std::vector<double> x(32 * kCols, 0.);                   // window sample u = 32 column + row
  uint64_t st = 0x9E3779B97F4A7C15ull;
  for (int u = 0; u < kFbRing + 32 * kOut; ++u) {
    st = st * 6364136223846793005ull + 1442695040888963407ull;
    x[u] = (double)(int64_t)(st >> 11) / 4503599627370496.0 - 1.0 + 0.5 * std::sin(0.013 * u);
  }
  double worst = 0.;
  // the backward-masking taps the kernel holds as compile-time constants against this host's evaluation of
  // fbearmodel.c:182-185: another libm may differ in the last bit (1.1e-16 .. 2.2e-16 relative), nothing more
  for (int k = 0; k < 6; ++k) worst = std::max(worst, std::fabs(fb.back_mask[k] - kBackMask[k]) / kBackMask[k]);
  if (worst > 2.5e-16) return 1.;
  worst = 0.;
  for (int b = 0; b < kFbBands; ++b) {
    const int half = fb.flen[b] / 2, off = fb.coef_off[b];
    // the reference's sum: taps n = 1 .. N - 1 at delays D + n, i.e. window samples 727 + N/2 - n + 32 t
    std::vector<double> dr(kOut), di(kOut);
    double peak = 0.;
    for (int t = 0; t < kOut; ++t) {
      long double re = 0, im = 0;
      for (int n = 1; n < fb.flen[b]; ++n) {
        const int k = n <= half ? n : fb.flen[b] - n;
        const double hr = fb.h_re[off + k], hi = n <= half ? fb.h_im[off + k] : -fb.h_im[off + k];
        const double v = x[727 + half - n + 32 * t];
        re += (long double)hr * v;
        im += (long double)(n == half ? 0. : hi) * v;
      }
      dr[t] = (double)re;
      di[t] = (double)im;
      peak = std::max(peak, std::hypot(dr[t], di[t]));
    }
    std::vector<double> yr(kOut, 0.), yi(kOut, 0.);
    if (b < kBsBands) {
      const int p = b / 2, sub = b & 1, J = fb.bs_whole[b];
      auto row_dot = [&](int ty, int col) {                // one row of the pair's tile on window column col
        double acc = 0.;
        for (int q = 0; q < 32; ++q)
          acc += fb.bs_coef[p][q / 4][bs_row(sub, ty) + 16 * (q & 3)] * (col >= 0 ? x[32 * col + q] : 0.);
        return acc;
      };
      for (int i = 0; i < 3; ++i) {
        // the running sums the way the kernel evaluates them (peaq_fb.hip, bs_pair): eight lanes per chain, a lane takes
        // kBsSeg consecutive outputs from kBsSeg g - s on; local sums with nothing carried in, a weighted scan over the
        // segments, then the outputs themselves
        const int c = 3 * sub + i, sg = fb.bs_seg_s[b];
        constexpr int kSpan = 8 * kBsSeg;
        std::vector<double> er(kSpan + J + kBsSeg, 0.), ei(kSpan + J + kBsSeg, 0.);   // enter values of the outputs -J .. kSpan - 1
        for (int t = -J; t < kSpan; ++t) {
          const int ch = fb.bs_col_head[p] + t + fb.bs_off_enter[b];
          if (ch >= kCols) continue;                       // (beyond the window: outputs >= 60, never looked at)
          er[t + J] = row_dot(2 * i, ch);
          ei[t + J] = row_dot(2 * i + 1, ch);
        }
        auto enter = [&](int t, double& r, double& m) {    // zeros in front of the history, like the rows' front slots
          r = t >= -J ? er[t + J] : 0.;
          m = t >= -J ? ei[t + J] : 0.;
        };
        const double (&rt)[2][2] = fb.bs_rot[3 * b + i];
        double v_r = 0., v_i = 0.;                         // V(-1) from the history (the kernel: on a launch's first tile)
        for (int t = -J; t < 0; ++t) {
          const double nr = rt[0][0] * v_r - rt[0][1] * v_i + er[t + J], ni = rt[0][0] * v_i + rt[0][1] * v_r + ei[t + J];
          v_r = nr;
          v_i = ni;
        }
        double dr8[8][kBsSeg], di8[8][kBsSeg], tr[8], ti[8];
        for (int g = 0; g < 8; ++g) {
          const int lane = 8 * c + g, t0 = kBsSeg * g - sg;
          const double rjr = fb.bs_seg[p][5][lane][0], rji = fb.bs_seg[p][5][lane][1];
          const double rr = fb.bs_seg[p][4][lane][0], ri = fb.bs_seg[p][4][lane][1];
          if (t0 < J && t0 + kBsSeg > J) return 1.;        // output J must start a segment
          for (int k = 0; k < kBsSeg; ++k) {
            double e_r, e_i, l_r, l_i;
            enter(t0 + k < 0 ? -J - 1 : t0 + k, e_r, e_i);
            enter(t0 + k < 0 ? -J - 1 : t0 + k - J, l_r, l_i);
            dr8[g][k] = e_r - (rjr * l_r - rji * l_i);
            di8[g][k] = e_i - (rjr * l_i + rji * l_r);
          }
          tr[g] = dr8[g][0];
          ti[g] = di8[g][0];
          for (int k = 1; k < kBsSeg; ++k) {
            const double nr = rr * tr[g] - ri * ti[g] + dr8[g][k], ni = rr * ti[g] + ri * tr[g] + di8[g][k];
            tr[g] = nr;
            ti[g] = ni;
          }
        }
        for (int lv = 0; lv < 3; ++lv) {                   // Hillis-Steele over the eight segments
          double nr[8], ni[8];
          for (int g = 0; g < 8; ++g) {
            const int lane = 8 * c + g, src = g - (1 << lv);
            const double wr = fb.bs_seg[p][lv][lane][0], wi_ = fb.bs_seg[p][lv][lane][1];
            const double sr_ = src >= 0 ? tr[src] : 123., si_ = src >= 0 ? ti[src] : -77.;   // (whatever the neighbouring group holds)
            nr[g] = tr[g] + (wr * sr_ - wi_ * si_);
            ni[g] = ti[g] + (wr * si_ + wi_ * sr_);
          }
          for (int g = 0; g < 8; ++g) {
            tr[g] = nr[g];
            ti[g] = ni[g];
          }
        }
        for (int g = 0; g < 8; ++g) {
          const int lane = 8 * c + g, t0 = kBsSeg * g - sg;
          const double kr = fb.bs_seg[p][3][lane][0], ki = fb.bs_seg[p][3][lane][1];
          const double rr = fb.bs_seg[p][4][lane][0], ri = fb.bs_seg[p][4][lane][1];
          double vr = (g > 0 ? tr[g - 1] : 0.) + (kr * v_r - ki * v_i), vi = (g > 0 ? ti[g - 1] : 0.) + (kr * v_i + ki * v_r);
          for (int k = 0; k < kBsSeg; ++k) {
            const double nr = rr * vr - ri * vi + dr8[g][k], ni = rr * vi + ri * vr + di8[g][k];
            vr = nr;
            vi = ni;
            const int t = t0 + k;
            if (t >= 0 && t < kOut) {
              yr[t] += vr;
              yi[t] += vi;
            }
          }
        }
      }
      for (int t = 0; t < kOut; ++t) {
        const int ch = fb.bs_col_head[p] + t + fb.bs_off_enter[b] + 1, cl = fb.bs_col_left[b] + t;
        yr[t] += row_dot(6, ch);
        yi[t] += row_dot(7, ch);
        for (int q = 8 * fb.bs_left_g[b][0]; q < 8 * fb.bs_left_g[b][1]; ++q) {
          yr[t] += fb.bs_left[b][q][0] * x[32 * cl + q];
          yi[t] += fb.bs_left[b][q][1] * x[32 * cl + q];
        }
      }
    } else {
      for (int t = 0; t < kOut; ++t)
        for (int s = 0; s < kMfdSteps; ++s)
          for (int kk = 0; kk < 4; ++kk) {
            const int d = kMfdD0 + 4 * s + kk, lane = (b - kMfdBand0) + 16 * kk;
            const double x1 = x[kFbRing - d + 32 * t], x2 = x[d - 2 + 32 * t];
            yr[t] += fb.mfd_re[(size_t)s * 64 + lane] * (x1 + x2);
            yi[t] += fb.mfd_im[(size_t)s * 64 + lane] * (x1 - x2);
          }
    }
    for (int t = 0; t < kOut; ++t) worst = std::max(worst, std::hypot(yr[t] - dr[t], yi[t] - di[t]) / peak);
  }
  return worst;
}

double fft_level_factor(double level_db) {
  // fftearmodel.c:312-313
  const double gamma = 0.84971762641205;
  const double g = gamma / 4 * (kFrame - 1);
  return std::pow(10.0, level_db / 10.0) / (8.0 / 3.0 * g * g);
}

double fb_level_factor(double level_db) { return std::pow(10.0, level_db / 20.0); }  // fbearmodel.c:252-253

}  // namespace peaq
