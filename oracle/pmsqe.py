"""TEST INFRASTRUCTURE (oracle) - PMSQE loss as the reference calls it (tools_for_loss.py:253-269, models.py:313-314).

**Parity unpinned.**  The arithmetic is third-party (`asteroid.losses.SingleSrcPMSQE`, `PITLossWrapper(pit_from='pw_pt')`,
`asteroid_filterbanks.{STFTFB, Encoder, transforms.mag}`); none of it is under /root/reference, no version is pinned there, it is not
installed and there is no network.  This file restates the PUBLISHED algorithm (Martin-Donas et al., "A deep learning loss
function based on the perceptual evaluation of the speech quality", IEEE SPL 2018) with the call chain of the reference:

  waves [N, L] -> view(N, L / 16000, 16000): every second of a clip is one "source"            (tools_for_loss.py:262-263)
  STFT: 512-point, hop 256, no padding, periodic sqrt-Hann analysis window, filters / 16         (Encoder(STFTFB(512, 512, stride 256)))
  spectrum the loss works on: the magnitude sqrt(re^2 + im^2 + 1e-8) that transforms.mag hands over (tools_for_loss.py:267-269; default,
  the reference's literal chain)  [power=True: re^2 + im^2, the paper's own definition - the build's opt-in cfg.pmsqe_power]
  per (estimate second i, clean second j): SLL equalisation -> 49-band Bark spectrum (P.862.2 tables) -> Bark frequency equalisation ->
  gain equalisation -> Zwicker loudness -> symmetric / asymmetric disturbance -> per-frame norms / audible-power weight -> mean over frames
  PIT: minimum over the permutations of the seconds of the mean pair loss, then mean over the batch           (PITLossWrapper 'pw_pt')

torch float64 on the CPU; autograd of this restatement is the gradient oracle of tests/test_gpu_pmsqe.py.  The tables are the data module
pmsqe_tables.py of the package (constants of ITU-T P.862.2), loaded by path - no product code runs here."""
import importlib.util
import itertools
import math
import os

import numpy as np
import torch

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PKG = [p for p in os.listdir(_ROOT) if p.endswith("_amd")][0]
_spec = importlib.util.spec_from_file_location("_pmsqe_tables", os.path.join(_ROOT, _PKG, "pmsqe_tables.py"))
T_ = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(T_)

FS, NFFT, HOP, NB, NBINS = 16000, 512, 256, 49, 257
ALPHA, BETA, EPS = 0.1, 0.309 * 0.1, 1e-8


def constants():
    thr = torch.tensor(T_.ABS_THRESH_POWER, dtype=torch.float64)
    cb = np.array(T_.CENTRE_OF_BAND_BARK)
    h = np.where(cb >= 4, 1.0, 6.0 / (cb + 2.0))
    zp = torch.tensor(0.23 * np.minimum(2.0, h) ** 0.15)                   # P.862 modified Zwicker power
    width = torch.tensor(T_.WIDTH_OF_BAND_BARK, dtype=torch.float64)
    M = torch.zeros(NBINS, NB, dtype=torch.float64)                          # Bark matrix: P.862 frequency warping
    lo = 0
    for k, n in enumerate(T_.HZ_BINS_PER_BAND):
        M[lo:lo + n, k] = T_.POW_DENS_CORRECTION[k]
        lo += n
    mask = torch.zeros(NBINS, dtype=torch.float64)                           # speech band 350 .. 3250 Hz of the SLL mean
    mask[11] = 0.5 * 25.0 / 31.25
    mask[12:104] = 1.0
    mask[104] = 0.5
    mask = mask * (2.0 * (NFFT + 2.0) / NFFT ** 2)                           # sqrt-Hann power correction factor 2.0
    return thr, zp, width, M, mask


def stft_filters():
    n = np.arange(NFFT)
    win = np.hanning(NFFT + 1)[:-1] ** 0.5
    ang = 2 * np.pi * np.outer(np.arange(NBINS), n) / NFFT
    scale = 0.5 * math.sqrt(NFFT * NFFT / HOP)
    return torch.tensor(np.cos(ang) * win / scale), torch.tensor(-np.sin(ang) * win / scale)     # [257, 512] each


def spectra(wav, power=False):
    """wav [N, L] float64 -> [N, S, T, 257] (S seconds, T = 61 frames per second)."""
    N, L = wav.shape
    if L % FS:
        raise ValueError("view(N, -1, fs) needs whole seconds (tools_for_loss.py:262)")
    seg = wav.reshape(N, L // FS, FS)
    fr = seg.unfold(-1, NFFT, HOP)                                            # [N, S, T, 512]
    C, S = stft_filters()
    re, im = fr @ C.T, fr @ S.T
    p = re * re + im * im
    return p if power else torch.sqrt(p + 1e-8)


def single_src_pmsqe(deg, ref):
    """deg, ref: [..., T, 257] spectra -> [...] loss."""
    thr, zp, width, M, mask = constants()
    Tn = deg.shape[-2]

    def sll(x):
        mean_pow = (x * mask).mean(-1, keepdim=True).sum(-2, keepdim=True) / Tn
        return 1e7 * x / mean_pow

    bark = lambda x: T_.SP_16K * (x @ M)
    audible = lambda b, f: torch.where(b > thr * f, b, torch.zeros_like(b)).sum(-1, keepdim=True)
    rb, db = bark(sll(ref)), bark(sll(deg))
    # Bark frequency equalisation of the degraded spectrum
    not_silent = audible(rb, 100.0) >= 1e7
    cond = rb >= thr * 100.0
    z = torch.zeros_like(rb)
    ppb_ref = torch.where(not_silent, torch.where(cond, rb, z), z).sum(-2, keepdim=True)
    ppb_deg = torch.where(not_silent, torch.where(cond, db, z), z).sum(-2, keepdim=True)
    db = torch.clamp((ppb_ref + 1000.0) / (ppb_deg + 1000.0), 0.01, 100.0) * db
    # gain equalisation
    db = torch.clamp((audible(rb, 1.0) + 5e3) / (audible(db, 1.0) + 5e3), 3e-4, 5.0) * db

    def loudness(b):
        a = (thr / 0.5) ** zp
        l = T_.SL_16K * a * ((0.5 + 0.5 * b / thr) ** zp - 1.0)
        return torch.where(b < thr, torch.zeros_like(b), l)

    lr, ld = loudness(rb), loudness(db)
    sym = torch.clamp((ld - lr).abs() - 0.25 * torch.minimum(lr, ld), min=0.0)
    asym = ((db + 50.0) / (rb + 50.0)) ** 1.2
    asym_d = torch.where(asym < 3.0, torch.zeros_like(asym), torch.clamp(asym, max=12.0)) * sym
    d_frame = torch.sqrt(((sym * width) ** 2 + EPS).sum(-1, keepdim=True)) * math.sqrt(float(width.sum()))
    da_frame = (asym_d * width).sum(-1, keepdim=True)
    w = ((audible(rb, 1.0) + 1e5) / 1e7) ** 0.04
    wd, wda = torch.clamp(d_frame / w, max=45.0), torch.clamp(da_frame / w, max=45.0)
    return (ALPHA * wd + BETA * wda).sum((-1, -2)) / Tn


def pairwise(est_wav, clean_wav, power=False):
    """[N, S, S]: loss of estimate second i against clean second j."""
    e, c = spectra(est_wav, power), spectra(clean_wav, power)
    return single_src_pmsqe(e[:, :, None], c[:, None, :])


def pmsqe_loss(clean_wav, est_wav, power=False):
    """get_array_pmsqe_loss(clean_array, est_array) (tools_for_loss.py:258-269) -> scalar."""
    pw = pairwise(est_wav.double(), clean_wav.double(), power)
    S = pw.shape[1]
    perms = list(itertools.permutations(range(S)))
    per = torch.stack([sum(pw[:, i, p[i]] for i in range(S)) / S for p in perms], 1)       # [N, S!]
    return per.min(1).values.mean()
