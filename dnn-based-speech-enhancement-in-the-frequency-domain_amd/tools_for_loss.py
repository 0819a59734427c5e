"""Loss surface of the reference's tools_for_loss.py (:17-94) on the fused HIP reductions.

Each loss is ONE pass over est/target (three inner products per utterance), a one-workgroup finalize and - for the
backward - one elementwise kernel `grad = ca[b]*est + cb[b]*target`.  Same call signatures and argument order as the
reference (`sdr(s1, s2)`, `si_snr(s1, s2)`, `si_sdr(reference, estimation)`); cuda fp32 tensors only.
"""
import ctypes as C

import torch

from . import _lib

LOSS_KINDS = {"MSE": 0, "SDR": 1, "SI-SNR": 2, "SI-SDR": 3}


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def loss_forward_raw(kind, est, tgt, stream):
    L_ = _lib.lib()
    B, L = est.shape
    ws = torch.empty(L_.sefd_loss_ws_floats(B), dtype=torch.float32, device=est.device)
    out = torch.empty((), dtype=torch.float32, device=est.device)
    rc = L_.sefd_loss_forward(kind, _vp(est), _vp(tgt), B, L, _vp(ws), _vp(out), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_forward failed ({rc})")
    return ws, out


def loss_backward_raw(kind, est, tgt, ws, grad_scale, grad_out, stream):
    L_ = _lib.lib()
    B, L = est.shape
    rc = L_.sefd_loss_backward(kind, _vp(est), _vp(tgt), B, L, _vp(ws), _vp(grad_scale), _vp(grad_out), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_backward failed ({rc})")


class _Loss(torch.autograd.Function):
    """value = the quantity the fused kernel computes: the *negated* metric for SDR / SI-SNR / SI-SDR, the MSE itself."""

    @staticmethod
    def forward(ctx, kind, est, tgt):
        if not (est.is_cuda and tgt.is_cuda):
            raise RuntimeError("sefd losses run on the MI355X only (cuda tensors); there is no CPU fallback")
        est2 = est.float().contiguous().view(-1, est.shape[-1])
        tgt2 = tgt.float().contiguous().view(-1, tgt.shape[-1])
        stream = torch.cuda.current_stream().cuda_stream
        ws, out = loss_forward_raw(kind, est2, tgt2, stream)
        ctx.kind, ctx.shape = kind, est.shape
        ctx.save_for_backward(est2, tgt2, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        est2, tgt2, ws = ctx.saved_tensors
        grad = torch.empty_like(est2)
        gs = g.float().contiguous().view(1)
        loss_backward_raw(ctx.kind, est2, tgt2, ws, gs, grad, torch.cuda.current_stream().cuda_stream)
        return None, grad.view(ctx.shape), None


def mse(estimated, target):
    return _Loss.apply(0, estimated, target)


def sdr(s1, s2, eps=1e-8):
    """tools_for_loss.py:29-33: s1 = target, s2 = estimate.  Returns +SDR (the model negates it, models.py:319)."""
    return -_Loss.apply(1, s2, s1)


def si_snr(s1, s2, eps=1e-8):
    """tools_for_loss.py:36-44: s1 = estimate, s2 = target."""
    return -_Loss.apply(2, s1, s2)


def si_sdr(reference, estimation, eps=1e-8):
    """tools_for_loss.py:47-94."""
    return -_Loss.apply(3, estimation, reference)
