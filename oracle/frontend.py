"""Oracle (test infrastructure): STFT / iSTFT front end restated on CPU.

Follows the reference:
  init_kernels      tools_for_model.py:16-33
  ConvSTFT.forward  tools_for_model.py:54-68
  ConviSTFT.forward tools_for_model.py:90-112
  stft / mag_phase  tools_for_model.py:628-684  (FullSubNet: torch.stft hop 300)
"""
import numpy as np
import torch
import torch.nn.functional as F


def periodic_hann(win_len: int) -> np.ndarray:
    # scipy.signal.get_window('hann'|'hanning', N, fftbins=True)  
    n = np.arange(win_len, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * n / win_len)


def window_of(win_len: int, win_type="hanning") -> np.ndarray:
    """init_kernels' window (tools_for_model.py:17-20): np.ones for win_type None / 'None', else scipy's periodic window."""
    if win_type in (None, "None"):
        return np.ones(win_len)
    if win_type in ("hanning", "hann"):
        return periodic_hann(win_len)
    from scipy.signal import get_window            # tools_for_model.py:20: any other scipy window name
    return np.asarray(get_window(win_type, win_len, fftbins=True), dtype=np.float64)


def analysis_kernel(win_len=400, fft_len=512, window=True, win_type="hanning") -> np.ndarray:
    """K[2*(N/2+1), win_len] float64: rows 0..N/2 = w*cos, rows N/2+1.. = -w*sin (tools_for_model.py:22-26,31)."""
    n = np.arange(win_len, dtype=np.float64)[None, :]
    k = np.arange(fft_len // 2 + 1, dtype=np.float64)[:, None]
    ang = 2.0 * np.pi * k * n / fft_len
    K = np.concatenate([np.cos(ang), -np.sin(ang)], 0)
    if window:
        K = K * window_of(win_len, win_type)[None, :]
    return K


def synthesis_kernel(win_len=400, fft_len=512, win_type="hanning") -> np.ndarray:
    """pinv(K_unwindowed).T * w  (tools_for_model.py:28-31)."""
    K = analysis_kernel(win_len, fft_len, window=False)
    return np.linalg.pinv(K).T * window_of(win_len, win_type)[None, :]


def conv_stft(wav: torch.Tensor, win_len=400, hop=100, fft_len=512, win_type="hanning") -> torch.Tensor:
    """[B, L] -> [B, 2*(N/2+1), T] (real rows then imag rows)."""
    K = torch.from_numpy(analysis_kernel(win_len, fft_len, win_type=win_type).astype(np.float32))[:, None, :]
    x = F.pad(wav[:, None, :], [win_len - hop, win_len - hop])
    return F.conv1d(x, K, stride=hop)


def conv_istft(spec: torch.Tensor, win_len=400, hop=100, fft_len=512, win_type="hanning") -> torch.Tensor:
    """[B, 2*(N/2+1), T] -> [B, 1, L]."""
    Kinv = torch.from_numpy(synthesis_kernel(win_len, fft_len, win_type).astype(np.float32))[:, None, :]
    w = torch.from_numpy(window_of(win_len, win_type).astype(np.float32))[None, :, None]
    out = F.conv_transpose1d(spec, Kinv, stride=hop)
    t = w.repeat(1, 1, spec.size(-1)) ** 2
    coff = F.conv_transpose1d(t, torch.eye(win_len)[:, None, :], stride=hop)
    out = out / (coff + 1e-8)
    return out[..., win_len - hop:-(win_len - hop)]


def ola_normaliser(T: int, win_len=400, hop=100) -> np.ndarray:
    """coff[(T-1)*hop + win_len] float32 exactly as the reference builds it (fp32 window squared, summed)."""
    w = periodic_hann(win_len).astype(np.float32) ** 2
    coff = np.zeros((T - 1) * hop + win_len, np.float32)
    for t in range(T):
        coff[t * hop:t * hop + win_len] += w
    return coff


def torch_stft(wav: torch.Tensor, n_fft=512, hop=300, win_len=400) -> torch.Tensor:
    """FullSubNet front end (tools_for_model.py:628-648): centre/reflect, hann_window(400)."""
    return torch.stft(wav, n_fft, hop_length=hop, win_length=win_len,
                      window=torch.hann_window(win_len), return_complex=True)
