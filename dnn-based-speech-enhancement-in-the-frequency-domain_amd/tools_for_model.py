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
