"""Host-side constants of the conv-STFT front end, kept only so that `state_dict()` carries the same buffers as the
reference (stft.weight, istft.weight, istft.window, istft.enframe - tools_for_model.py:16-33, 46-47, 80-88).
The device kernels use the planner's own copies (csrc/plan.cpp); a CPU test checks both against each other."""
import numpy as np
import torch


def window_fn(win_type, win_len):
    if win_type is None or win_type == 'None':
        return np.ones(win_len)
    if win_type in ('hanning', 'hann'):                 # scipy get_window(..., fftbins=True): periodic Hann
        n = np.arange(win_len, dtype=np.float64)
        return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_len)
    from scipy.signal import get_window                 # tools_for_model.py:20: any window name or (name, parameter) tuple scipy knows
    return np.asarray(get_window(win_type, win_len, fftbins=True), dtype=np.float64)


def stft_kernels(win_len, fft_len, win_type):
    w = window_fn(win_type, win_len)
    n = np.arange(win_len, dtype=np.float64)[None, :]
    k = np.arange(fft_len // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * ((k * n) % fft_len) / fft_len
    K = np.concatenate([np.cos(ang), -np.sin(ang)], 0)                       # [fft_len+2, win_len], unwindowed
    # pinv(K)^T in closed form: K^T K = (N/2) I + E with E[n, m] = [n - m even]  (Sherman-Morrison per parity class)
    half = fft_len / 2.0
    par = (np.arange(win_len) % 2)
    cnt = np.array([(par == 0).sum(), (par == 1).sum()], dtype=np.float64)
    sums = np.stack([K[:, par == 0].sum(1), K[:, par == 1].sum(1)], 1)        # [rows, 2]
    corr = sums[:, par] / (half + cnt[par])[None, :]
    Kinv = (K - corr) / half
    f32 = lambda a: torch.from_numpy(a.astype(np.float32))
    return f32((K * w)[:, None, :]), f32((Kinv * w)[:, None, :]), f32(w[None, :, None])
