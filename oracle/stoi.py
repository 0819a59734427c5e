"""Oracle (test infrastructure): STOI restated in numpy/scipy from the published algorithm - C. H. Taal, R. C. Hendriks,
R. Heusdens, J. Jensen, "An Algorithm for Intelligibility Prediction of Time-Frequency Weighted Noisy Speech", IEEE TASLP 2011 -
with the framing / resampling conventions of `pystoi` 0.3.3, the package the reference calls
(tools_for_estimate.py:91-99: `stoi(clean, estimated, cfg.fs, extended=False)`; version from SE_tutorials.ipynb cell 3).

PARITY UNPINNED: pystoi is a third-party dependency that is not vendored in /root/reference and not installed here, and the
reference holds no STOI fixture.  This file pins the C++ scorer (csrc_host/scorers.cpp) to a second, independent statement of the
same published algorithm; it cannot pin either to pystoi's bits.
"""
import numpy as np

FS, N_FRAME, NFFT, NUMBAND, MINFREQ, N_SEG, BETA, DYN_RANGE = 10000, 256, 512, 15, 150, 30, -15.0, 40
EPS = np.finfo("float").eps


def resample_window(p, q):
    """Kaiser-windowed sinc of Octave's `resample` (60 dB rejection, 10 % roll-off) for the rational factor p / q."""
    g = np.gcd(p, q)
    p, q = p // g, q // g
    fc = 1.0 / (2 * max(p, q))
    roll = fc / 10
    rej = 60.0
    L = int(np.ceil((rej - 8) / (28.714 * roll)))
    t = np.arange(-L, L + 1)
    h = 2 * p * fc * np.sinc(2 * fc * t) * np.kaiser(2 * L + 1, 0.1102 * (rej - 8.7))
    return h


def resample(x, p, q):
    from scipy.signal import resample_poly
    h = resample_window(p, q)
    return resample_poly(x, p, q, window=h / h.sum())


def thirdoct(fs=FS, nfft=NFFT, nb=NUMBAND, fmin=MINFREQ):
    f = np.linspace(0, fs, nfft + 1)[: nfft // 2 + 1]
    k = np.arange(nb, dtype=float)
    lo, hi = fmin * 2.0 ** ((2 * k - 1) / 6), fmin * 2.0 ** ((2 * k + 1) / 6)
    obm = np.zeros((nb, len(f)))
    for i in range(nb):
        a, b = int(np.argmin((f - lo[i]) ** 2)), int(np.argmin((f - hi[i]) ** 2))
        obm[i, a:b] = 1
    return obm


def _window():
    return np.hanning(N_FRAME + 2)[1:-1]


def remove_silent_frames(x, y):
    w, hop = _window(), N_FRAME // 2
    idx = range(0, len(x) - N_FRAME, hop)
    xf = np.array([w * x[i:i + N_FRAME] for i in idx])
    yf = np.array([w * y[i:i + N_FRAME] for i in idx])
    e = 20 * np.log10(np.linalg.norm(xf, axis=1) + EPS)
    keep = (e.max() - DYN_RANGE - e) < 0
    xf, yf = xf[keep], yf[keep]
    n = (len(xf) - 1) * hop + N_FRAME
    xs, ys = np.zeros(n), np.zeros(n)
    for i in range(len(xf)):
        xs[i * hop:i * hop + N_FRAME] += xf[i]
        ys[i * hop:i * hop + N_FRAME] += yf[i]
    return xs, ys


def _spec(x):
    w, hop = _window(), N_FRAME // 2
    return np.array([np.fft.rfft(w * x[i:i + N_FRAME], n=NFFT) for i in range(0, len(x) - N_FRAME, hop)]).T


def stoi(clean, est, fs):
    x, y = np.asarray(clean, dtype=np.float64), np.asarray(est, dtype=np.float64)
    if fs != FS:
        x, y = resample(x, FS, fs), resample(y, FS, fs)
    x, y = remove_silent_frames(x, y)
    X, Y = _spec(x), _spec(y)
    if X.shape[-1] < N_SEG:
        return 1e-5
    obm = thirdoct()
    xt, yt = np.sqrt(obm @ np.abs(X) ** 2), np.sqrt(obm @ np.abs(Y) ** 2)
    xs = np.array([xt[:, m - N_SEG:m] for m in range(N_SEG, xt.shape[1] + 1)])
    ys = np.array([yt[:, m - N_SEG:m] for m in range(N_SEG, xt.shape[1] + 1)])
    c = np.linalg.norm(xs, axis=2, keepdims=True) / (np.linalg.norm(ys, axis=2, keepdims=True) + EPS)
    yp = np.minimum(ys * c, xs * (1 + 10 ** (-BETA / 20)))
    yp = yp - yp.mean(axis=2, keepdims=True)
    xs = xs - xs.mean(axis=2, keepdims=True)
    yp = yp / (np.linalg.norm(yp, axis=2, keepdims=True) + EPS)
    xs = xs / (np.linalg.norm(xs, axis=2, keepdims=True) + EPS)
    return float((yp * xs).sum() / (xs.shape[0] * xs.shape[1]))
