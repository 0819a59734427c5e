"""Module-level helpers of the reference's tools_for_model.py that the FullSubNet trainer calls (trainer.py:100-104, 341-345):
`stft`, `mag_phase`, `build_complex_ideal_ratio_mask`, `decompress_cIRM` - same signatures, HIP kernels underneath.
cuda tensors only (no CPU fallback)."""
import ctypes as C

import torch

from . import _lib
from . import config as cfg
from .plan import PHASE_FWD, Plan

_FE_CACHE = {}


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stft(y, n_fft=cfg.fft_len, hop_length=int(cfg.win_len * cfg.ola_ratio), win_length=cfg.win_len):
    """tools_for_model.py:628-648: torch.stft(center=True, reflect, hann_window(win_length)) -> complex64 [B, F, T]."""
    assert y.dim() == 2
    if not y.is_cuda:
        raise RuntimeError("sefd stft runs on the MI355X only (cuda tensors); there is no CPU fallback")
    y = y.detach().float().contiguous()
    B, L = y.shape
    key = (B, L, n_fft, hop_length, win_length, str(y.device))
    fe = _FE_CACHE.get(key)
    if fe is None:
        plan = Plan(B, L, win_len=win_length, win_inc=hop_length, fft_len=n_fft, model="TorchSTFT")
        fe = (plan, plan.alloc_arenas(y.device))
        _FE_CACHE[key] = fe
    plan, ar = fe
    plan.io(ar, "wav", (B, L)).copy_(y)
    plan.run(PHASE_FWD, ar, torch.cuda.current_stream().cuda_stream)
    return torch.view_as_complex(plan.io(ar, "spec", (B, plan.NF, plan.T, 2)).clone())


def istft(features, n_fft=cfg.fft_len, hop_length=int(cfg.win_len * cfg.ola_ratio), win_length=cfg.win_len, length=None, use_mag_phase=False):
    """tools_for_model.py:651-680: wrapper of torch.istft(window=hann_window(win_length), center=True, length=length) -> [B, L].
    `features`: complex [B, F, T], the reference's real-pair [B, F, T, 2] (which torch >= 2 rejects in the reference itself,
    SURVEY Q9 - accepted here through view_as_complex semantics), or (mag, phase) with use_mag_phase."""
    if use_mag_phase:
        assert isinstance(features, (tuple, list))
        mag, phase = features
        features = torch.stack([mag * torch.cos(phase), mag * torch.sin(phase)], dim=-1)
    if torch.is_complex(features):
        features = torch.view_as_real(features)
    if not features.is_cuda:
        raise RuntimeError("sefd istft runs on the MI355X only (cuda tensors); there is no CPU fallback")
    features = features.detach().float().contiguous()
    B, F, T, two = features.shape
    assert two == 2 and F == n_fft // 2 + 1
    L = int(length) if length is not None else hop_length * (T - 1)
    key = ("istft", B, L, T, n_fft, hop_length, win_length, str(features.device))
    fe = _FE_CACHE.get(key)
    if fe is None:
        plan = Plan(B, L, win_len=win_length, win_inc=hop_length, fft_len=n_fft, model="TorchISTFT")
        if plan.T != T:
            raise ValueError(f"istft: {T} frames do not match length {L} at hop {hop_length} (expected {plan.T})")
        fe = (plan, plan.alloc_arenas(features.device))
        _FE_CACHE[key] = fe
    plan, ar = fe
    plan.io(ar, "spec", (B, F, T, 2)).copy_(features)
    plan.run(PHASE_FWD, ar, torch.cuda.current_stream().cuda_stream)
    return plan.io(ar, "wav", (B, L)).clone()


def _targets(noisy, clean, want_mag, want_phase, want_cirm):
    L_ = _lib.lib()
    nr = torch.view_as_real(noisy.contiguous())
    n = noisy.numel()
    mag = torch.empty(noisy.shape, dtype=torch.float32, device=noisy.device) if want_mag else None
    ph = torch.empty(noisy.shape, dtype=torch.float32, device=noisy.device) if want_phase else None
    cirm = torch.empty(noisy.shape + (2,), dtype=torch.float32, device=noisy.device) if want_cirm else None
    cr = torch.view_as_real(clean.contiguous()) if clean is not None else None
    rc = L_.sefd_fsn_targets(_vp(nr), _vp(cr), n, _vp(mag), _vp(ph), _vp(cirm), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        raise RuntimeError(f"sefd_fsn_targets failed ({rc})")
    return mag, ph, cirm


def mag_phase(complex_tensor):
    """tools_for_model.py:683-684."""
    mag, ph, _ = _targets(complex_tensor, None, True, True, False)
    return mag, ph


def build_complex_ideal_ratio_mask(noisy, clean):
    """tools_for_model.py:687-705 (+ compress_cIRM K=10, C=0.1): complex [B,F,T] x2 -> [B,F,T,2]."""
    return _targets(noisy, clean, False, False, True)[2]


def decompress_cIRM(mask, K=10, limit=9.9):
    """tools_for_model.py:720-723 (validation path; element-wise torch is plumbing here, not the training hot path)."""
    mask = limit * (mask >= limit) - limit * (mask <= -limit) + mask * (torch.abs(mask) < limit)
    return -K * torch.log((K - mask) / (K + mask))


def Bar(iterable, *args, **kwargs):
    """tools_for_model.py:1396-1416 wraps every loader in a console progress bar; the training loops only iterate it."""
    return iterable
