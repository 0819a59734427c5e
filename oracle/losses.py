"""Oracle (test infrastructure): loss functions restated from tools_for_loss.py.

  l2_norm/sdr/si_snr/si_sdr   tools_for_loss.py:17-94   (quirks of SURVEY Q4 kept verbatim)
  LMS log-mel loss            tools_for_loss.py:120-249 (layout bug Q8 reproduced)
  dispatch                    models.py:303-323
PMSQE (tools_for_loss.py:255-269) is third-party asteroid arithmetic: NOT restated, parity unpinned.
"""
import math
import numpy as np
import torch


def _ip(a, b):
    return torch.sum(a * b, -1, keepdim=True)


def sdr(s1, s2, eps=1e-8):
    sn = _ip(s1, s1)
    d = _ip(s1 - s2, s1 - s2)
    return torch.mean(10 * torch.log10(sn ** 2 / (d ** 2 + eps)))


def si_snr(s1, s2, eps=1e-8):
    s12 = _ip(s1, s2)
    s22 = _ip(s2, s2)
    s_t = s12 / (s22 + eps) * s2
    e = s1 - s_t
    return torch.mean(10 * torch.log10(_ip(s_t, s_t) / (_ip(e, e) + eps) + eps))


def si_sdr(reference, estimation, eps=1e-8):
    re = torch.sum(reference ** 2, -1, keepdim=True)
    a = torch.sum(reference * estimation, -1, keepdim=True) / re + eps
    proj = a * reference
    noise = estimation - proj
    ratio = torch.sum(proj ** 2, -1) / torch.sum(noise ** 2, -1) + eps
    return 10 * torch.log10(torch.mean(ratio) + eps)


def si_sdr_sharded(reference, estimation, all_reduce_sum, world, eps=1e-8):
    """si_sdr over a batch sharded across `world` ranks, as csrc/api.hip `loss_dp_finish_kernel` forms it (there is no distributed code in
    the reference; the arithmetic to reproduce is tools_for_loss.py:91-94 on the WHOLE batch): every rank all-reduces (sum of its ratios,
    its rows), the value is that of the global batch, and the gradient is `world` x the global-batch gradient for this rank's rows (the
    gradient exchange then sums over ranks and the step multiplies by 1 / world).  all_reduce_sum(t): in-place sum of a 1-d tensor over ranks."""
    re = torch.sum(reference ** 2, -1, keepdim=True)
    a = torch.sum(reference * estimation, -1, keepdim=True) / re + eps
    proj = a * reference
    noise = estimation - proj
    ratio = torch.sum(proj ** 2, -1) / torch.sum(noise ** 2, -1) + eps
    S = ratio.sum()
    tot = torch.stack([S.detach().double(), torch.tensor(float(ratio.numel()), dtype=torch.float64)])
    all_reduce_sum(tot)
    m = (S - S.detach()) * (world / float(tot[1])) + float(tot[0] / tot[1])
    return 10 * torch.log10(m + eps)


def main_loss(kind: str, estimated, target):
    """models.py:315-323 (argument order matters)."""
    if kind == "MSE":
        return torch.mean((estimated - target) ** 2)
    if kind == "SDR":
        return -sdr(target, estimated)
    if kind == "SI-SNR":
        return -si_snr(estimated, target)
    if kind == "SI-SDR":
        return -si_sdr(target, estimated)
    raise ValueError(kind)


# ---------------------------------------------------------------- LMS (tools_for_loss.py:120-249)
def _freq_to_mel(f):
    return 1127.01048 * math.log(1 + f / 700.0)


def _mel_to_freq(m):
    return 700 * (math.exp(m / 1127.01048) - 1)


def mel_filter_bank(num_bands: int, n_fft=512, fs=16000) -> np.ndarray:
    """[n_fft/2+1, num_bands] float32 triangles with floor-binned edges (tools_for_loss.py:140-184).

    The float32 intermediate array of the reference (`melRange.astype(np.float32)`) is kept: the
    floor() of the band edges depends on it.
    """
    max_hz = fs / 2
    n_bins = int(n_fft / 2) + 1
    max_mel = _freq_to_mel(max_hz)
    min_mel = _freq_to_mel(0)
    centers = np.arange(num_bands + 2).astype(np.float32) * (max_mel - min_mel) / (num_bands + 1) + min_mel
    for i in range(num_bands + 2):
        centers[i] = _mel_to_freq(centers[i])
        centers[i] = math.floor(n_bins * centers[i] / max_hz)
    fb = np.zeros((num_bands, n_bins))
    for i in range(1, num_bands + 1):
        s, c, e = int(centers[i - 1]), int(centers[i]), int(centers[i + 1])
        for j in range(s, c):
            fb[i - 1, j] = (float(j) - s) / (c - s)
        for j in range(c, e):
            fb[i - 1, j] = 1 - ((float(j) - c) / (e - c))
    return fb.T.astype(np.float32).copy()


def lms_loss(clean_mags: torch.Tensor, est_mags: torch.Tensor, scales=(16, 32, 64), n_fft=512):
    """get_array_lms_loss (tools_for_loss.py:242-249). Inputs [B, 257, T] *contiguous*; Q8 flat re-view kept."""
    total = 0.0
    B = clean_mags.shape[0]
    for b in range(B):
        per = 0.0
        for nb in scales:
            bank = torch.from_numpy(mel_filter_bank(nb, n_fft))
            outs = []
            for x in (clean_mags[b], est_mags[b]):
                p = x.contiguous().view(-1, n_fft // 2 + 1) / n_fft        # Q8: NOT transposed
                outs.append(torch.log(torch.mm(p, bank) + 1e-7))
            t, e = outs
            rm = torch.sqrt(torch.mean((e - t) ** 2, -1) + 1e-7)
            per = per + torch.mean(rm)
        total = total + per / len(scales)
    return total / B
