// Wide-band PESQ (ITU-T P.862 with the P.862.2 extension, MOS-LQO output) for the validation loop: the contract of the reference's
// `cal_pesq` (tools_for_estimate.py:51-84: ctypes call `pesq(clean float64[n], degraded float64[n], n, n) -> double` into a prebuilt x86
// PESQ.so at 16 kHz).  Restated from the published algorithm (P.862 sections 10.1-10.2), double precision:
//   level alignment (band-passed power 350-3250 Hz to 1e7) -> P.862.2 input high-pass -> 32 ms Hann frames, 50 % overlap -> Bark
//   power densities (49 bands) -> frequency-response and gain compensation -> Zwicker loudness -> symmetric / asymmetric disturbance
//   with the dead zone -> L6 over 320 ms, L2 over time -> raw score -> MOS-LQO mapping.
// Time alignment: the validation loop scores an enhanced signal against the clean signal it was made from - same length, no delay
// beyond a filter's few samples.  This implementation treats the file as ONE utterance with ONE delay: the lag (within +-256 samples)
// that maximises the cross-correlation of the two level-aligned, input-filtered signals; P.862's envelope-based crude alignment,
// per-utterance histogram alignment, utterance splitting and bad-interval re-alignment are not restated.  Pinned against MOS-LQO values
// of the reference's PESQ.so on 34 such pairs (tests/golden/pesq_golden.npz, tests/test_scorers_cpu.py: |delta| <= 0.01).
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>
#include "pesq_tables.h"

namespace sefd_pesq {

constexpr int kFs = 16000, kDown = 64, kSearch = 75, kPadMs = 320, kNb = 49, kNf = 512;
constexpr double kPi = 3.14159265358979323846;
constexpr double kTargetAvgPower = 1e7;

static void fft(std::vector<std::complex<double>>& a, bool inv) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = 2 * kPi / (double)len * (inv ? 1 : -1);
    const std::complex<double> wl(std::cos(ang), std::sin(ang));
    for (size_t i = 0; i < n; i += len) {
      std::complex<double> w(1, 0);
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v; a[i + k + len / 2] = u - v;
        w *= wl;
      }
    }
  }
  if (inv) for (auto& x : a) x /= (double)n;
}

static double interpolate(double freq, const double* curve, int npts) {     // curve: (Hz, dB) pairs, clamped at both ends
  if (freq <= curve[0]) return curve[1];
  if (freq >= curve[2 * (npts - 1)]) return curve[2 * (npts - 1) + 1];
  int i = 1;
  while (curve[2 * i] < freq) ++i;
  const double f0 = curve[2 * (i - 1)], f1 = curve[2 * i], d0 = curve[2 * (i - 1) + 1], d1 = curve[2 * i + 1];
  return d0 + (d1 - d0) * (freq - f0) / (f1 - f0);
}

// A signal as P.862 lays it out: kSearch * kDown zeros | n samples | kPadMs of zeros | kSearch * kDown zeros
struct Signal {
  std::vector<double> data;
  long nsamples;                      // n + 2 * kSearch * kDown
};
static Signal make_signal(const double* x, long n, double scale) {
  Signal s;
  s.nsamples = n + 2L * kSearch * kDown;
  s.data.assign((size_t)s.nsamples + kPadMs * (kFs / 1000), 0.0);
  for (long i = 0; i < n; ++i) s.data[(size_t)kSearch * kDown + i] = x[i] * scale;
  return s;
}
static double pow_of(const std::vector<double>& x, long start, long stop, long divisor) {
  double p = 0;
  for (long i = start; i < stop; ++i) p += x[(size_t)i] * x[(size_t)i];
  return p / (double)divisor;
}
// zero-phase filter given as a (Hz, dB) curve, applied to the whole active part through one FFT
static void apply_filter(std::vector<double>& data, long nsamples, const double* curve, int npts) {
  const long n = nsamples - 2L * kSearch * kDown + kPadMs * (kFs / 1000);
  size_t p2 = 1;
  while ((long)p2 < n) p2 <<= 1;
  std::vector<std::complex<double>> a(p2);
  for (long i = 0; i < n; ++i) a[(size_t)i] = data[(size_t)kSearch * kDown + i];
  fft(a, false);
  const double gain1k = interpolate(1000.0, curve, npts), res = (double)kFs / (double)p2;
  for (size_t i = 0; i <= p2 / 2; ++i) {
    const double f = std::pow(10.0, (interpolate(i * res, curve, npts) - gain1k) / 20.0);
    a[i] *= f;
    if (i > 0 && i < p2 / 2) a[p2 - i] *= f;
  }
  fft(a, true);
  for (long i = 0; i < n; ++i) data[(size_t)kSearch * kDown + i] = a[(size_t)i].real();
}
// false: no power in the 350 - 3250 Hz band (an all-zero or DC-only clip: a model that outputs zeros early in training, a silent reference) -
// the level cannot be fixed (0 * inf = NaN through every later stage); the caller returns the floor of the scale instead
static bool fix_power_level(Signal& s, long max_nsamples) {
  std::vector<double> f = s.data;
  apply_filter(f, s.nsamples, kAlignFilterDb, 26);
  const double p = pow_of(f, kSearch * kDown, s.nsamples - kSearch * kDown + kPadMs * (kFs / 1000),
                          max_nsamples - 2L * kSearch * kDown + kPadMs * (kFs / 1000));
  if (!(p > 0.0) || !std::isfinite(p)) return false;
  const double g = std::sqrt(kTargetAvgPower / p);
  for (long i = 0; i < s.nsamples; ++i) s.data[(size_t)i] *= g;
  return true;
}
// P.862.2 input filter: one biquad (direct form II) over the samples of the file - NOT over the zero padding behind them (the ringing would
// sit in the last transform frames) - behind a 15-sample linear fade-in / fade-out (sample k and sample n-1-k times (k + 1) / 16, k < 15).  Both
// details were read off the reference's PESQ.so in round 4 (its exported IIRFilt interposed: input ratios 1/16 .. 15/16 at both ends, Nx = the
// file length): without them a clip that ends mid-wave splatters into the top Bark bands of its last frame, which moves the reference's
// time-averaged spectrum there by up to 10x and, through the frequency-response compensation, every frame's disturbance (worst 0.066 MOS-LQO).
static void wb_input_filter(Signal& s) {
  const double b0 = kWbHpSos[0], b1 = kWbHpSos[1], b2 = kWbHpSos[2], a1 = kWbHpSos[3], a2 = kWbHpSos[4];
  double z1 = 0, z2 = 0;
  const long n = s.nsamples - 2L * kSearch * kDown;
  for (long k = 0; k < 15 && k < n; ++k) {
    const double w = (double)(k + 1) / 16.0;
    s.data[(size_t)kSearch * kDown + k] *= w;
    if (n - 1 - k != k) s.data[(size_t)kSearch * kDown + n - 1 - k] *= w;
  }
  for (long i = 0; i < n; ++i) {
    double& x = s.data[(size_t)kSearch * kDown + i];
    const double z0 = x - a1 * z1 - a2 * z2;
    x = b0 * z0 + b1 * z1 + b2 * z2;
    z2 = z1; z1 = z0;
  }
}

static void short_term_fft(const Signal& s, const double* win, long start, double* hz) {
  std::vector<std::complex<double>> a(kNf);
  for (int n = 0; n < kNf; ++n) a[n] = s.data[(size_t)start + n] * win[n];
  fft(a, false);
  for (int k = 0; k < kNf / 2; ++k) hz[k] = std::norm(a[k]);
  hz[0] = 0;
}
static void freq_warping(const double* hz, double* pitch) {
  int b = 0;
  for (int band = 0; band < kNb; ++band) {
    double sum = 0;
    for (int i = 0; i < kHzBinsPerBand[band]; ++i) sum += hz[b++];
    pitch[band] = sum * kPowDensCorrection[band] * kSp16k;
  }
}
static double total_audible(const double* pitch, double factor) {
  double r = 0;
  for (int band = 1; band < kNb; ++band)
    if (pitch[band] > factor * kAbsThreshPower[band]) r += pitch[band];
  return r;
}
static void intensity_warping(const double* pitch, double* loud) {
  for (int band = 0; band < kNb; ++band) {
    const double thr = kAbsThreshPower[band], in = pitch[band];
    double h = kCentreOfBandBark[band] < 4 ? 6.0 / (kCentreOfBandBark[band] + 2.0) : 1.0;
    if (h > 2) h = 2;
    h = std::pow(h, 0.15);
    const double zw = 0.23 * h;
    loud[band] = in > thr ? std::pow(thr / 0.5, zw) * (std::pow(0.5 + 0.5 * in / thr, zw) - 1.0) : 0.0;
    loud[band] *= kSl16k;
  }
}
static double pseudo_lp(const double* x, double p) {
  double tw = 0, r = 0;
  for (int band = 1; band < kNb; ++band) {
    const double w = kWidthOfBandBark[band];
    r += std::pow(std::fabs(x[band]) * w, p);
    tw += w;
  }
  return std::pow(r / tw, 1.0 / p) * tw;
}
static double lpq_weight(long start, long stop, double ps, double pt, const std::vector<double>& fd, const std::vector<double>& tw) {
  double rt = 0, tt = 0;
  for (long s0 = start; s0 <= stop; s0 += 10) {
    double rs = 0;
    int cnt = 0;
    for (long f = s0; f < s0 + 20; ++f) {
      if (f <= stop) rs += std::pow(fd[(size_t)f], ps);
      ++cnt;
    }
    rs = std::pow(rs / cnt, 1.0 / ps);
    rt += std::pow(tw[(size_t)(s0 - start)] * rs, pt);
    tt += std::pow(tw[(size_t)(s0 - start)], pt);
  }
  return std::pow(rt / tt, 1.0 / pt);
}

// raw P.862 score of (reference, degraded), both n samples at 16 kHz, zero delay
constexpr double kRawFloor = -0.5;          // lowest raw score of P.862 (MOS-LQO 1.04 through the P.862.2 mapping): silent / DC-only input
static double pesq_raw(const double* refx, const double* degx, long n, double in_scale) {
  Signal ref = make_signal(refx, n, in_scale), deg = make_signal(degx, n, in_scale);
  const long maxn = std::max(ref.nsamples, deg.nsamples);
  const long pad = kPadMs * (kFs / 1000);
  if (!fix_power_level(ref, maxn) || !fix_power_level(deg, maxn)) return kRawFloor;
  wb_input_filter(ref);
  wb_input_filter(deg);

  // one delay for the whole file: arg max of |cross-correlation| (FFT), lags -256 .. 256
  long delay = 0;
  {
    const long na = maxn + pad;
    size_t p2 = 1;
    while ((long)p2 < 2 * na) p2 <<= 1;
    std::vector<std::complex<double>> fr(p2), fd2(p2);
    for (long i = 0; i < na; ++i) { fr[(size_t)i] = ref.data[(size_t)i]; fd2[(size_t)i] = deg.data[(size_t)i]; }
    fft(fr, false); fft(fd2, false);
    for (size_t i = 0; i < p2; ++i) fr[i] = std::conj(fr[i]) * fd2[i];          // r[lag] = sum_t ref[t] deg[t + lag]
    fft(fr, true);
    // |r|: the psychoacoustic model works on power spectra, so an enhanced signal of inverted polarity scores like the upright one (SI-SNR does
    // not see the sign: one of 18 reference-trained models came out negated) - with the signed maximum the search locked half a pitch period off
    double best = std::fabs(fr[0].real());
    for (long lag = -256; lag <= 256; ++lag) {
      const double v = std::fabs(fr[(size_t)((lag + (long)p2) % (long)p2)].real());
      if (v > best * (1.0 + 1e-9)) { best = v; delay = lag; }
    }
  }
  long skip_start = 0, skip_end = 0;
  double sum5;
  do {
    sum5 = 0;
    for (int i = 0; i < 5; ++i) sum5 += std::fabs(ref.data[(size_t)kSearch * kDown + skip_start + i]);
    if (sum5 < 500) ++skip_start;
  } while (sum5 < 500 && skip_start < maxn / 2);
  do {
    sum5 = 0;
    for (int i = 0; i < 5; ++i) sum5 += std::fabs(ref.data[(size_t)(maxn - kSearch * kDown + pad - 1 - skip_end - i)]);
    if (sum5 < 500) ++skip_end;
  } while (sum5 < 500 && skip_end < maxn / 2);
  const long start_frame = skip_start / (kNf / 2);
  const long stop_frame = (maxn - 2L * kSearch * kDown + pad - skip_end) / (kNf / 2) - 1;
  if (stop_frame < start_frame) return 4.5;
  const long nfr = stop_frame + 1;

  double win[kNf];
  for (int i = 0; i < kNf; ++i) win[i] = 0.5 * (1.0 - std::cos(2 * kPi * i / kNf));
  std::vector<double> pr((size_t)nfr * kNb), pd((size_t)nfr * kNb);
  std::vector<char> silent((size_t)nfr);
  double hz[kNf / 2];
  for (long f = 0; f < nfr; ++f) {
    const long st = kSearch * kDown + f * kNf / 2;
    short_term_fft(ref, win, st, hz);
    freq_warping(hz, &pr[(size_t)f * kNb]);
    const long sd = st + delay;
    if (sd > 0 && sd + kNf < maxn + pad) short_term_fft(deg, win, sd, hz);
    else std::fill(hz, hz + kNf / 2, 0.0);
    freq_warping(hz, &pd[(size_t)f * kNb]);
    silent[(size_t)f] = total_audible(&pr[(size_t)f * kNb], 1e2) < 1e7;
  }
  // time-averaged audible power per band (non-silent frames), frequency-response compensation of the reference
  const double total_frames = (double)((maxn - 2L * kSearch * kDown + pad) / (kNf / 2) - 1);
  for (int band = 0; band < kNb; ++band) {
    double ar = 0, ad = 0;
    for (long f = 0; f < nfr; ++f) {
      if (silent[(size_t)f]) continue;
      const double hr = pr[(size_t)f * kNb + band], hd = pd[(size_t)f * kNb + band];
      if (hr > 100 * kAbsThreshPower[band]) ar += hr;
      if (hd > 100 * kAbsThreshPower[band]) ad += hd;
    }
    ar /= total_frames; ad /= total_frames;
    double x = (ad + 1000.0) / (ar + 1000.0);
    x = std::min(100.0, std::max(0.01, x));
    for (long f = 0; f < nfr; ++f) pr[(size_t)f * kNb + band] *= x;
  }
  std::vector<double> fd((size_t)nfr), fda((size_t)nfr), tpr((size_t)nfr), tw((size_t)nfr, 1.0);
  double old_scale = 1;
  for (long f = 0; f < nfr; ++f) {
    double* r = &pr[(size_t)f * kNb];
    double* d = &pd[(size_t)f * kNb];
    const double ta_r = total_audible(r, 1), ta_d = total_audible(d, 1);
    tpr[(size_t)f] = ta_r;
    double scale = (ta_r + 5e3) / (ta_d + 5e3);
    if (f > 0) scale = 0.2 * old_scale + 0.8 * scale;
    old_scale = scale;
    scale = std::min(5.0, std::max(3e-4, scale));
    for (int band = 0; band < kNb; ++band) d[band] *= scale;
    double lr[kNb], ld[kNb], dist[kNb];
    intensity_warping(r, lr);
    intensity_warping(d, ld);
    for (int band = 0; band < kNb; ++band) {
      const double dd = ld[band] - lr[band], m = 0.25 * std::min(ld[band], lr[band]);
      dist[band] = dd > m ? dd - m : (dd < -m ? dd + m : 0.0);
    }
    fd[(size_t)f] = pseudo_lp(dist, 2.0);
    for (int band = 0; band < kNb; ++band) {
      double h = std::pow((d[band] + 50.0) / (r[band] + 50.0), 1.2);
      if (h > 12) h = 12;
      if (h < 3) h = 0;
      dist[band] *= h;
    }
    fda[(size_t)f] = pseudo_lp(dist, 1.0);
  }
  // ---- bad intervals (P.862 psychoacoustic model): runs of frames whose symmetric disturbance exceeds 30 (smeared over +-2 frames, at least
  // 5 frames long) get a local delay from the cross-correlation of the rectified signals over +-4 transform lengths; the degraded frames of the
  // interval are re-analysed at that delay and each frame keeps the SMALLER of its two disturbances
  {
    std::vector<char> bad((size_t)nfr), sm((size_t)nfr, 0);
    for (long f = 0; f < nfr; ++f) bad[(size_t)f] = fd[(size_t)f] > 30.0;
    const long stopf = stop_frame;
    for (long f = 2; f < stopf - 2; ++f) {
      char l = bad[(size_t)f], r = bad[(size_t)f];
      for (int i = -2; i <= 0; ++i) l = std::max(l, bad[(size_t)(f + i)]);
      for (int i = 0; i <= 2; ++i) r = std::max(r, bad[(size_t)(f + i)]);
      sm[(size_t)f] = std::min(l, r);
    }
    struct Iv { long f0, f1, s0, s1, delay; };
    std::vector<Iv> ivs;
    long f = 0;
    while (f <= stopf) {
      while (f <= stopf && !sm[(size_t)f]) ++f;
      if (f <= stopf) {
        const long f0 = f;
        while (f <= stopf && sm[(size_t)f]) ++f;
        if (f <= stopf && f - f0 >= 5) ivs.push_back(Iv{f0, f, 0, 0, 0});
      }
    }
    if (!ivs.empty()) {
      const long nn = maxn - kSearch * kDown + pad;          // one past the last usable sample index
      auto tweaked = [&](long i) {                           // the degraded signal at the file delay, clamped like the standard's copy loop
        long j = i + delay;
        if (j < 0) j = 0;
        if (j >= (long)deg.data.size()) j = (long)deg.data.size() - 1;
        return deg.data[(size_t)j];
      };
      const long sr = 4L * kNf;
      for (Iv& iv : ivs) {
        iv.s0 = iv.f0 * (kNf / 2) + (long)kSearch * kDown;
        iv.s1 = iv.f1 * (kNf / 2) + kNf + (long)kSearch * kDown;
        const long ns = iv.s1 - iv.s0, tot = 2 * sr + ns;
        size_t p2 = 1;
        while ((long)p2 < 2 * tot) p2 <<= 1;
        std::vector<std::complex<double>> a(p2), b2(p2);
        double p1 = 0, pw2 = 0;
        for (long i = 0; i < tot; ++i) {
          double rv = (i >= sr && i < sr + ns) ? ref.data[(size_t)(iv.s0 + i - sr)] : 0.0;
          long j = iv.s0 - sr + i;
          if (j < (long)kSearch * kDown) j = (long)kSearch * kDown;
          if (j >= nn) j = nn - 1;
          const double dv = tweaked(j);
          p1 += rv * rv; pw2 += dv * dv;
          a[(size_t)i] = std::fabs(rv); b2[(size_t)i] = std::fabs(dv);
        }
        const double norm = std::sqrt((p1 / (double)tot) * ((double)tot / (double)p2) * (pw2 / (double)tot) * ((double)tot / (double)p2));
        fft(a, false); fft(b2, false);
        for (size_t i = 0; i < p2; ++i) a[i] = std::conj(a[i]) * b2[i];
        fft(a, true);
        double best = 0; long bd = 0;
        for (long i = -sr; i <= -1; ++i) { const double h = std::fabs(a[(size_t)(i + (long)p2)].real()) / (norm * (double)p2); if (h > best) { best = h; bd = i; } }
        for (long i = 0; i < sr; ++i) { const double h = std::fabs(a[(size_t)i].real()) / (norm * (double)p2); if (h > best) { best = h; bd = i; } }
        iv.delay = best < 0.5 ? 0 : bd;
      }
      // doubly aligned degraded signal inside the intervals, re-analysis of their frames
      Signal dd = deg;
      for (size_t i = 0; i < dd.data.size(); ++i) dd.data[i] = tweaked((long)i);
      for (const Iv& iv : ivs)
        for (long i = iv.s0; i < iv.s1 && i < (long)dd.data.size(); ++i) {
          long j = i + iv.delay;
          if (j < 0) j = 0;
          if (j >= nn) j = nn - 1;
          dd.data[(size_t)i] = tweaked(j);
        }
      double old2 = 1;
      for (const Iv& iv : ivs)
        for (long fr2 = iv.f0; fr2 < iv.f1 && fr2 < nfr; ++fr2) {
          const long st = (long)kSearch * kDown + fr2 * kNf / 2;
          double* r = &pr[(size_t)fr2 * kNb];
          double d2[kNb];
          short_term_fft(dd, win, st, hz);
          freq_warping(hz, d2);
          const double ta_r = total_audible(r, 1), ta_d = total_audible(d2, 1);
          double scale = (ta_r + 5e3) / (ta_d + 5e3);
          if (fr2 > 0) scale = 0.2 * old2 + 0.8 * scale;
          old2 = scale;
          scale = std::min(5.0, std::max(3e-4, scale));
          for (int band = 0; band < kNb; ++band) d2[band] *= scale;
          double lr[kNb], ld[kNb], dist[kNb];
          intensity_warping(r, lr);
          intensity_warping(d2, ld);
          for (int band = 0; band < kNb; ++band) {
            const double q = ld[band] - lr[band], m = 0.25 * std::min(ld[band], lr[band]);
            dist[band] = q > m ? q - m : (q < -m ? q + m : 0.0);
          }
          fd[(size_t)fr2] = std::min(fd[(size_t)fr2], pseudo_lp(dist, 2.0));
          for (int band = 0; band < kNb; ++band) {
            double h = std::pow((d2[band] + 50.0) / (r[band] + 50.0), 1.2);
            if (h > 12) h = 12;
            if (h < 3) h = 0;
            dist[band] *= h;
          }
          fda[(size_t)fr2] = std::min(fda[(size_t)fr2], pseudo_lp(dist, 1.0));
        }
    }
  }
  if (nfr > 1000) {
    const long nn = (maxn - 2L * kSearch * kDown) / (kNf / 2) - 1;
    double twf = std::min(0.5, ((double)nn - 1000.0) / 5500.0);
    for (long f = 0; f < nfr; ++f) tw[(size_t)f] = (1.0 - twf) + twf * (double)f / (double)nn;
  }
  for (long f = 0; f < nfr; ++f) {
    const double h = std::pow((tpr[(size_t)f] + 1e5) / 1e7, 0.04);
    fd[(size_t)f] = std::min(45.0, fd[(size_t)f] / h);
    fda[(size_t)f] = std::min(45.0, fda[(size_t)f] / h);
  }
  const double di = lpq_weight(start_frame, stop_frame, 6, 2, fd, tw), ai = lpq_weight(start_frame, stop_frame, 6, 2, fda, tw);
  return 4.5 - 0.1 * di - 0.0309 * ai;
}

double pesq_wb_mos_lqo(const double* ref, const double* deg, long n, double in_scale) {
  const double raw = pesq_raw(ref, deg, n, in_scale);
  return 0.999 + 4.0 / (1.0 + std::exp(-1.3669 * raw + 3.8224));      // P.862.2 mapping to MOS-LQO
}

}  // namespace sefd_pesq
