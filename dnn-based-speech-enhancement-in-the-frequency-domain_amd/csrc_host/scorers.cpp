// Host-side objective scorers of the validation loop (reference tools_for_estimate.py:51-99), multi-threaded C++.
//   STOI  - short-time objective intelligibility (Taal et al., IEEE TASLP 2011) with the framing conventions of pystoi 0.3.3,
//           which the reference calls as stoi(clean, estimated, cfg.fs, extended=False) (tools_for_estimate.py:91-99).
//           pystoi is not vendored in the reference: parity unpinned (see oracle/stoi.py), the algorithm is the published one.
// Plain C ABI (include/sefd_scorers.h): double-precision arithmetic throughout, no GPU, no torch.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdint>
#include <thread>
#include <vector>

namespace sefd_pesq { double pesq_wb_mos_lqo(const double* ref, const double* deg, long n, double in_scale); }

namespace {

constexpr int kFs = 10000, kFrame = 256, kNfft = 512, kBands = 15, kSeg = 30;
constexpr double kMinFreq = 150.0, kBeta = -15.0, kDynRange = 40.0;
constexpr double kPi = 3.14159265358979323846;
const double kEps = 2.220446049250313e-16;

double bessel_i0(double x) {                       // series of the modified Bessel function (np.kaiser uses i0)
  double s = 1.0, t = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 200; ++k) { t *= q / ((double)k * k); s += t; if (t < 1e-18 * s) break; }
  return s;
}

// Octave-style `resample` low-pass (Kaiser-windowed sinc, 60 dB, 10 % roll-off), unit DC gain
std::vector<double> resample_window(int p, int q) {
  { int a = p, b = q; while (b) { const int t = a % b; a = b; b = t; } p /= a; q /= a; }
  const double fc = 1.0 / (2.0 * std::max(p, q)), roll = fc / 10.0, rej = 60.0;
  const int L = (int)std::ceil((rej - 8.0) / (28.714 * roll));
  const double beta = 0.1102 * (rej - 8.7);
  std::vector<double> h(2 * L + 1);
  double sum = 0;
  for (int i = 0; i <= 2 * L; ++i) {
    const double t = i - L, a = 2.0 * fc * t;
    const double sinc = a == 0.0 ? 1.0 : std::sin(kPi * a) / (kPi * a);
    const double r = 2.0 * i / (2.0 * L) - 1.0;
    h[i] = 2.0 * p * fc * sinc * bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bessel_i0(beta);
    sum += h[i];
  }
  for (double& v : h) v /= sum;
  return h;
}

// scipy.signal.resample_poly(x, up, down, window=h) with zero padding: polyphase evaluation of  upfirdn(h * up, x, up, down)
std::vector<double> resample_poly(const std::vector<double>& x, int up, int down, const std::vector<double>& win) {
  int a = up, b = down;
  while (b) { const int t = a % b; a = b; b = t; }
  up /= a; down /= a;
  if (up == 1 && down == 1) return x;
  const int64_t n_in = (int64_t)x.size();
  int64_t n_out = n_in * up;
  n_out = n_out / down + (n_out % down ? 1 : 0);
  const int half = ((int)win.size() - 1) / 2;
  const int pre = down - half % down;
  const int64_t pre_remove = (half + pre) / down;
  std::vector<double> y((size_t)n_out);
  // full output sample m = sum_k hp[k] * xu[m*down - k], xu[j] = x[j / up] when up | j;  hp = [0]*pre ++ win*up
  for (int64_t o = 0; o < n_out; ++o) {
    const int64_t pos = (o + pre_remove) * down;            // index into the up-sampled stream
    double acc = 0;
    // k = pre + i (i: tap of win), need (pos - k) % up == 0 and 0 <= (pos - k) / up < n_in
    int64_t i0 = pos - pre;                                 // i = i0 - up*j' ...  iterate source samples instead
    int64_t jhi = std::min<int64_t>(n_in - 1, i0 / up);     // largest source index with tap i = i0 - j*up >= 0
    if (i0 < 0) { y[(size_t)o] = 0; continue; }
    int64_t jlo = (i0 - ((int64_t)win.size() - 1) + up - 1) / up;
    if (jlo < 0) jlo = 0;
    for (int64_t j = jlo; j <= jhi; ++j) acc += win[(size_t)(i0 - j * up)] * x[(size_t)j];
    y[(size_t)o] = acc * up;
  }
  return y;
}

void fft(std::vector<std::complex<double>>& a) {            // in-place radix-2, size a power of two
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = -2.0 * kPi / (double)len;
    const std::complex<double> wl(std::cos(ang), std::sin(ang));
    for (size_t i = 0; i < n; i += len) {
      std::complex<double> w(1.0, 0.0);
      for (size_t k = 0; k < len / 2; ++k) {
        const std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
        w *= wl;
      }
    }
  }
}

std::vector<double> hann_inner(int n) {                     // np.hanning(n + 2)[1:-1]
  std::vector<double> w(n);
  for (int i = 0; i < n; ++i) w[i] = 0.5 - 0.5 * std::cos(2.0 * kPi * (i + 1) / (n + 1));
  return w;
}

double stoi_one(const float* clean, const float* est, int n, int fs) {
  std::vector<double> x(clean, clean + n), y(est, est + n);
  if (fs != kFs) {
    const std::vector<double> h = resample_window(kFs, fs);
    x = resample_poly(x, kFs, fs, h);
    y = resample_poly(y, kFs, fs, h);
  }
  const std::vector<double> w = hann_inner(kFrame);
  const int hop = kFrame / 2;
  // ---- remove silent frames (energy of the clean frame more than 40 dB below the loudest), overlap-add what is left
  {
    std::vector<int> starts;
    for (int i = 0; i < (int)x.size() - kFrame; i += hop) starts.push_back(i);
    if (starts.empty()) return 1e-5;
    std::vector<double> e(starts.size());
    double emax = -1e300;
    for (size_t f = 0; f < starts.size(); ++f) {
      double s = 0;
      for (int j = 0; j < kFrame; ++j) { const double v = w[j] * x[starts[f] + j]; s += v * v; }
      e[f] = 20.0 * std::log10(std::sqrt(s) + kEps);
      emax = std::max(emax, e[f]);
    }
    std::vector<int> keep;
    for (size_t f = 0; f < starts.size(); ++f) if (emax - kDynRange - e[f] < 0) keep.push_back(starts[f]);
    std::vector<double> xs((keep.size() - 1) * hop + kFrame, 0.0), ys(xs.size(), 0.0);
    for (size_t f = 0; f < keep.size(); ++f)
      for (int j = 0; j < kFrame; ++j) { xs[f * hop + j] += w[j] * x[keep[f] + j]; ys[f * hop + j] += w[j] * y[keep[f] + j]; }
    x.swap(xs); y.swap(ys);
  }
  // ---- one-third octave band envelopes of the 256 / 128 STFT
  int lo[kBands], hi[kBands];
  for (int b = 0; b < kBands; ++b) {
    const double fl = kMinFreq * std::pow(2.0, (2.0 * b - 1.0) / 6.0), fh = kMinFreq * std::pow(2.0, (2.0 * b + 1.0) / 6.0);
    double bl = 1e300, bh = 1e300;
    for (int k = 0; k <= kNfft / 2; ++k) {
      const double f = (double)kFs * k / kNfft;
      if ((f - fl) * (f - fl) < bl) { bl = (f - fl) * (f - fl); lo[b] = k; }
      if ((f - fh) * (f - fh) < bh) { bh = (f - fh) * (f - fh); hi[b] = k; }
    }
  }
  std::vector<int> starts;
  for (int i = 0; i < (int)x.size() - kFrame; i += hop) starts.push_back(i);
  const int nf = (int)starts.size();
  if (nf < kSeg) return 1e-5;
  std::vector<double> xt((size_t)kBands * nf), yt((size_t)kBands * nf);
  std::vector<std::complex<double>> bx(kNfft), by(kNfft);
  for (int f = 0; f < nf; ++f) {
    for (int j = 0; j < kNfft; ++j) {
      bx[j] = j < kFrame ? std::complex<double>(w[j] * x[starts[f] + j], 0.0) : std::complex<double>(0.0, 0.0);
      by[j] = j < kFrame ? std::complex<double>(w[j] * y[starts[f] + j], 0.0) : std::complex<double>(0.0, 0.0);
    }
    fft(bx); fft(by);
    for (int b = 0; b < kBands; ++b) {
      double sx = 0, sy = 0;
      for (int k = lo[b]; k < hi[b]; ++k) { sx += std::norm(bx[k]); sy += std::norm(by[k]); }
      xt[(size_t)b * nf + f] = std::sqrt(sx);
      yt[(size_t)b * nf + f] = std::sqrt(sy);
    }
  }
  // ---- clipped, normalised correlation over 30-frame segments
  const double clip = 1.0 + std::pow(10.0, -kBeta / 20.0);
  double total = 0;
  const int J = nf - kSeg + 1;
  for (int m = 0; m < J; ++m)
    for (int b = 0; b < kBands; ++b) {
      const double* xs = &xt[(size_t)b * nf + m];
      const double* ys = &yt[(size_t)b * nf + m];
      double nx = 0, ny = 0;
      for (int i = 0; i < kSeg; ++i) { nx += xs[i] * xs[i]; ny += ys[i] * ys[i]; }
      const double c = std::sqrt(nx) / (std::sqrt(ny) + kEps);
      double yp[kSeg], xm = 0, ym = 0;
      for (int i = 0; i < kSeg; ++i) { yp[i] = std::min(ys[i] * c, xs[i] * clip); xm += xs[i]; ym += yp[i]; }
      xm /= kSeg; ym /= kSeg;
      double sxx = 0, syy = 0, sxy = 0;
      for (int i = 0; i < kSeg; ++i) { const double a = xs[i] - xm, q = yp[i] - ym; sxx += a * a; syy += q * q; sxy += a * q; }
      total += sxy / ((std::sqrt(sxx) + kEps) * (std::sqrt(syy) + kEps));
    }
  return total / ((double)J * kBands);
}

template <typename F>
void parallel_for(int n, int nthreads, F f) {
  if (nthreads <= 0) nthreads = (int)std::thread::hardware_concurrency();
  nthreads = std::max(1, std::min(nthreads, n));
  if (nthreads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t) th.emplace_back([=]() { for (int i = t; i < n; i += nthreads) f(i); });
  for (auto& t : th) t.join();
}

}  // namespace

extern "C" {

// STOI of `B` utterance pairs of `n` samples each (row-major [B][n] float32) at sampling rate fs; one score per utterance.
int32_t sefd_stoi_batch(const float* clean, const float* est, int32_t B, int32_t n, int32_t fs, double* out, int32_t nthreads) {
  if (!clean || !est || !out || B < 1 || n < 1 || fs < 1) return -1;
  parallel_for(B, nthreads, [&](int b) { out[b] = stoi_one(clean + (int64_t)b * n, est + (int64_t)b * n, n, fs); });
  return 0;
}

// Wide-band PESQ MOS-LQO (pesq.cpp: P.862 + P.862.2) of B pairs; fs must be 16000 (the reference's PESQ.so is a 16 kHz build).  The score
// does not depend on the input scale (both signals are level-aligned first), so waveforms in [-1, 1] are taken as they are.
int32_t sefd_pesq_batch(const float* clean, const float* deg, int32_t B, int32_t n, int32_t fs, double* out, int32_t nthreads) {
  const double scale = 1.0;
  if (!clean || !deg || !out || B < 1 || n < 512 || fs != 16000) return -1;
  parallel_for(B, nthreads, [&](int b) {
    std::vector<double> r(n), d(n);
    for (int i = 0; i < n; ++i) { r[i] = clean[(int64_t)b * n + i]; d[i] = deg[(int64_t)b * n + i]; }
    out[b] = sefd_pesq::pesq_wb_mos_lqo(r.data(), d.data(), n, scale);
  });
  return 0;
}

}  // extern "C"
