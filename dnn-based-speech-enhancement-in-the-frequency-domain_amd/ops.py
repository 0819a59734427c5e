"""`sefd::` PyTorch custom ops (torch.library) over the C ABI of libsefd_hip.so - the dispatcher-level surface of the hot path.

The reference has no native op surface (everything below `models.py` is ATen / cuDNN); north_star asks for the HIP path
"exposed as PyTorch-ROCm custom ops".  Each op below is a thin, allocation-explicit wrapper of ONE C-ABI entry point of include/sefd.h
(same argument meaning, cuda tensors only, current HIP stream), registered with a fake (meta) implementation so that it traces, and - for
the losses - with a registered backward.  The nn.Module mirrors (`models.py`, `tools_for_loss.py`) call the same C entry points directly
through ctypes (no dispatcher hop inside the fused train step); `tests/test_gpu_ops.py::test_torch_library_ops` pins op == mirror.

    torch.ops.sefd.loss(kind, est, tgt)                       scalar loss of tools_for_loss.py:17-94 (kind 0 MSE, 1 SDR, 2 SI-SNR, 3 SI-SDR),
                                                              differentiable w.r.t. est (rows of any length) and tgt (rows <= 16 elements)
    torch.ops.sefd.adam_step_(param, grad, m, v, step, ...)   torch.optim.Adam.step on flat fp32 buffers, in place
    torch.ops.sefd.mix_snr(speech, bank, start, snr_db, q)   generate_noisy_data.py:46-67 on the GPU
    torch.ops.sefd.plan_run(handle, phase, arenas)            one phase (0 forward, 1 backward) of a planned model: sefd_plan_run
"""
import ctypes as C
from typing import List, Tuple

import torch

from . import _lib

ROWS_MAX_L = 16


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("sefd ops run on the MI355X only (cuda tensors); there is no CPU fallback")


# ---------------------------------------------------------------------------------------------- losses
@torch.library.custom_op("sefd::loss_forward", mutates_args=())
def loss_forward(kind: int, est: torch.Tensor, tgt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(loss [], workspace): est / tgt fp32 [R, L] contiguous."""
    _need_cuda(est, tgt)
    L_ = _lib.lib()
    R, L = est.shape
    out = torch.empty((), dtype=torch.float32, device=est.device)
    if L <= ROWS_MAX_L:
        ws = torch.empty(L_.sefd_loss_rows_ws_floats(R), dtype=torch.float32, device=est.device)
        rc = L_.sefd_loss_rows_forward(kind, _vp(est), _vp(tgt), R, L, _vp(ws), _vp(out), _stream())
    else:
        ws = torch.empty(L_.sefd_loss_ws_floats(R), dtype=torch.float32, device=est.device)
        rc = L_.sefd_loss_forward(kind, _vp(est), _vp(tgt), R, L, _vp(ws), _vp(out), _stream())
    if rc != 0:
        raise RuntimeError(f"sefd::loss_forward failed ({rc})")
    return out, ws


@loss_forward.register_fake
def _(kind, est, tgt):
    # the size functions are host-only arithmetic: a traced graph sees the shape eager execution allocates
    L_ = _lib.lib()
    R, L = est.shape
    n = L_.sefd_loss_rows_ws_floats(R) if L <= ROWS_MAX_L else L_.sefd_loss_ws_floats(R)
    return est.new_empty(()), est.new_empty((n,))


@torch.library.custom_op("sefd::loss_backward", mutates_args=())
def loss_backward(kind: int, est: torch.Tensor, tgt: torch.Tensor, ws: torch.Tensor, grad_scale: torch.Tensor,
                  want_est: bool, want_tgt: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """(d loss / d est, d loss / d tgt) * grad_scale; a gradient that was not asked for is an empty tensor."""
    _need_cuda(est, tgt, ws, grad_scale)
    L_ = _lib.lib()
    R, L = est.shape
    ge = torch.empty_like(est) if want_est else est.new_empty((0,))
    gt = torch.empty_like(tgt) if want_tgt else tgt.new_empty((0,))
    if L <= ROWS_MAX_L:
        rc = L_.sefd_loss_rows_backward(kind, _vp(est), _vp(tgt), R, L, _vp(ws), _vp(grad_scale), _vp(ge) if want_est else None,
                                        _vp(gt) if want_tgt else None, _stream())
    else:
        if want_tgt:
            raise NotImplementedError("sefd::loss: rows longer than %d elements are differentiable w.r.t. the estimate only" % ROWS_MAX_L)
        rc = L_.sefd_loss_backward(kind, _vp(est), _vp(tgt), R, L, _vp(ws), _vp(grad_scale), _vp(ge), _stream()) if want_est else 0
    if rc != 0:
        raise RuntimeError(f"sefd::loss_backward failed ({rc})")
    return ge, gt


@loss_backward.register_fake
def _(kind, est, tgt, ws, grad_scale, want_est, want_tgt):
    return (torch.empty_like(est) if want_est else est.new_empty((0,))), (torch.empty_like(tgt) if want_tgt else tgt.new_empty((0,)))


@torch.library.custom_op("sefd::loss", mutates_args=())
def loss(kind: int, est: torch.Tensor, tgt: torch.Tensor) -> torch.Tensor:
    """The value the fused kernels compute: MSE itself, the NEGATED metric for SDR / SI-SNR / SI-SDR (what model.loss returns).  Argument
    roles: SDR est = s2, tgt = s1; SI-SNR est = s1, tgt = s2; SI-SDR est = estimation, tgt = reference (tools_for_loss.py:29-94)."""
    e2 = est.float().contiguous().view(-1, est.shape[-1])
    t2 = tgt.float().contiguous().view(-1, tgt.shape[-1])
    return loss_forward(kind, e2, t2)[0].clone()


@loss.register_fake
def _(kind, est, tgt):
    return est.new_empty((), dtype=torch.float32)


def _loss_setup(ctx, inputs, output):
    kind, est, tgt = inputs
    ctx.kind = kind
    ctx.save_for_backward(est, tgt)


def _loss_bwd(ctx, g):
    est, tgt = ctx.saved_tensors
    e2 = est.float().contiguous().view(-1, est.shape[-1])
    t2 = tgt.float().contiguous().view(-1, tgt.shape[-1])
    _, ws = loss_forward(ctx.kind, e2, t2)          # three inner products per row: recomputed rather than carried through the dispatcher
    ge, gt = loss_backward(ctx.kind, e2, t2, ws, g.float().contiguous().view(1), ctx.needs_input_grad[1], ctx.needs_input_grad[2])
    return None, (ge.view(est.shape) if ctx.needs_input_grad[1] else None), (gt.view(tgt.shape) if ctx.needs_input_grad[2] else None)


loss.register_autograd(_loss_bwd, setup_context=_loss_setup)


# ---------------------------------------------------------------------------------------------- Adam
@torch.library.custom_op("sefd::adam_step_", mutates_args=("param", "exp_avg", "exp_avg_sq"))
def adam_step_(param: torch.Tensor, grad: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor, step: int, lr: float,
               beta1: float, beta2: float, eps: float, grad_scale: float) -> None:
    """torch.optim.Adam.step (train_interface.py:59 defaults) on flat fp32 buffers: ONE launch (sefd_adam_step); step counts from 1."""
    _need_cuda(param, grad, exp_avg, exp_avg_sq)
    rc = _lib.lib().sefd_adam_step(_vp(param), _vp(grad), _vp(exp_avg), _vp(exp_avg_sq), param.numel(), step, lr, beta1, beta2, eps, grad_scale, _stream())
    if rc != 0:
        raise RuntimeError(f"sefd::adam_step_ failed ({rc})")


# ---------------------------------------------------------------------------------------------- SNR mixing
@torch.library.custom_op("sefd::mix_snr", mutates_args=())
def mix_snr(speech: torch.Tensor, noise_bank: torch.Tensor, noise_start: torch.Tensor, snr_db: torch.Tensor, quantize: bool) -> torch.Tensor:
    """generate_noisy_data.py:46-67 for a batch (sefd_mix_snr): speech [B, L], flat noise bank, per-utterance noise offsets and SNRs -> noisy
    [B, L]; quantize reproduces the int16 file round trip of the offline script."""
    from .dataloader import mix_snr as _mix
    return _mix(speech, noise_bank, noise_start, snr_db, quantize)


@mix_snr.register_fake
def _(speech, noise_bank, noise_start, snr_db, quantize):
    return torch.empty_like(speech, dtype=torch.float32)


# ---------------------------------------------------------------------------------------------- planned models
@torch.library.custom_op("sefd::plan_run", mutates_args=("arenas",))
def plan_run(handle: int, phase: int, arenas: List[torch.Tensor]) -> None:
    """One whole phase of a planned model (sefd_plan_run): `handle` = the sefd_plan* (Plan.h), `arenas` = the six arena tensors in Arena
    order (workspace, parameters, gradients, BatchNorm buffers, constants, I/O block) - the caller owns all device memory."""
    _need_cuda(*arenas)
    ptrs = (C.c_void_p * len(arenas))(*[C.c_void_p(a.data_ptr()) for a in arenas])
    rc = _lib.lib().sefd_plan_run(C.c_void_p(handle), phase, 0, -1, ptrs, _stream())
    if rc != 0:
        raise RuntimeError(f"sefd::plan_run failed ({rc})")
