"""Loss surface of the reference's tools_for_loss.py (:17-94) on the fused HIP reductions.

Each loss is ONE pass over est/target (three inner products per utterance), a one-workgroup finalize and - for the
backward - one elementwise kernel `grad = ca[b]*est + cb[b]*target`.  Same call signatures and argument order as the
reference (`sdr(s1, s2)`, `si_snr(s1, s2)`, `si_sdr(reference, estimation)`); cuda fp32 tensors only.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from . import config as cfg

LOSS_KINDS = {"MSE": 0, "SDR": 1, "SI-SNR": 2, "SI-SDR": 3}


def _vp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


_DP = None           # the active data-parallel exchange (sefd_amd.ddp.GradientExchange) or None: set_data_parallel


def set_data_parallel(exchange):
    """Tell the losses that the batch is sharded over `exchange`'s ranks.  Only SI-SDR cares: tools_for_loss.py:91-94 averages the per-row
    ratios over the WHOLE batch before the log, so each rank's (sum of ratios, rows) pair is sum-all-reduced between the loss's forward
    and backward kernels (two floats, on the current stream) and the loss / gradient are those of the global batch (the gradient times
    `world`: the exchange sums the ranks' gradients and the step applies 1 / world).  MSE / SDR / SI-SNR are means of per-row terms and
    shard exactly without it.  Returns the previous setting."""
    global _DP
    prev, _DP = _DP, (exchange if (exchange is not None and getattr(exchange, "active", False) and exchange.world > 1) else None)
    return prev


def _dp_finish(kind, rows, n, ws, out, stream):
    if kind != LOSS_KINDS["SI-SDR"] or _DP is None:
        return
    L_ = _lib.lib()
    off = L_.sefd_loss_dp_offset(n, rows)
    _DP.all_reduce_stats(ws[off:off + 2])
    rc = L_.sefd_loss_dp_finish(rows, n, _vp(ws), _DP.world, _vp(out), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_dp_finish failed ({rc})")


def loss_forward_raw(kind, est, tgt, stream):
    L_ = _lib.lib()
    B, L = est.shape
    ws = torch.empty(L_.sefd_loss_ws_floats(B), dtype=torch.float32, device=est.device)
    out = torch.empty((), dtype=torch.float32, device=est.device)
    rc = L_.sefd_loss_forward(kind, _vp(est), _vp(tgt), B, L, _vp(ws), _vp(out), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_forward failed ({rc})")
    _dp_finish(kind, 0, B, ws, out, stream)
    return ws, out


def loss_backward_raw(kind, est, tgt, ws, grad_scale, grad_out, stream):
    L_ = _lib.lib()
    B, L = est.shape
    rc = L_.sefd_loss_backward(kind, _vp(est), _vp(tgt), B, L, _vp(ws), _vp(grad_scale), _vp(grad_out), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_backward failed ({rc})")


ROWS_MAX_L = 16                  # rows this short take the one-thread-per-row kernels (sefd_loss_rows_*): FullSubNet.loss, models.py:674-682


def loss_rows_forward_raw(kind, est, tgt, stream):
    """est, tgt: fp32 [R, L], L <= ROWS_MAX_L."""
    L_ = _lib.lib()
    R, L = est.shape
    ws = torch.empty(L_.sefd_loss_rows_ws_floats(R), dtype=torch.float32, device=est.device)
    out = torch.empty((), dtype=torch.float32, device=est.device)
    rc = L_.sefd_loss_rows_forward(kind, _vp(est), _vp(tgt), R, L, _vp(ws), _vp(out), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_rows_forward failed ({rc})")
    _dp_finish(kind, 1, R, ws, out, stream)
    return ws, out


def loss_rows_backward_raw(kind, est, tgt, ws, grad_scale, grad_est, grad_tgt, stream):
    R, L = est.shape
    rc = _lib.lib().sefd_loss_rows_backward(kind, _vp(est), _vp(tgt), R, L, _vp(ws), _vp(grad_scale), _vp(grad_est), _vp(grad_tgt), C.c_void_p(stream))
    if rc != 0:
        raise RuntimeError(f"sefd_loss_rows_backward failed ({rc})")


class _LossRows(torch.autograd.Function):
    """Same value as _Loss for rows of at most ROWS_MAX_L elements, differentiable with respect to BOTH arguments (the reference's
    fullsubnet_train passes the network output in the `target` slot, trainer.py:107)."""

    @staticmethod
    def forward(ctx, kind, est, tgt):
        est2 = est.float().contiguous().view(-1, est.shape[-1])
        tgt2 = tgt.float().contiguous().view(-1, tgt.shape[-1])
        ws, out = loss_rows_forward_raw(kind, est2, tgt2, torch.cuda.current_stream().cuda_stream)
        ctx.kind, ctx.shape = kind, est.shape
        ctx.save_for_backward(est2, tgt2, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        est2, tgt2, ws = ctx.saved_tensors
        ge = torch.empty_like(est2) if ctx.needs_input_grad[1] else None
        gt = torch.empty_like(tgt2) if ctx.needs_input_grad[2] else None
        loss_rows_backward_raw(ctx.kind, est2, tgt2, ws, g.float().contiguous().view(1), ge, gt, torch.cuda.current_stream().cuda_stream)
        return None, (ge.view(ctx.shape) if ge is not None else None), (gt.view(ctx.shape) if gt is not None else None)


def _loss(kind, est, tgt):
    if not (est.is_cuda and tgt.is_cuda):
        raise RuntimeError("sefd losses run on the MI355X only (cuda tensors); there is no CPU fallback")
    if est.shape[-1] <= ROWS_MAX_L:
        return _LossRows.apply(kind, est, tgt)
    if tgt.requires_grad:
        raise NotImplementedError("the long-row loss kernels differentiate the estimate only (rows longer than %d elements)" % ROWS_MAX_L)
    return _Loss.apply(kind, est, tgt)


class _Loss(torch.autograd.Function):
    """value = the quantity the fused kernel computes: the *negated* metric for SDR / SI-SNR / SI-SDR, the MSE itself."""

    @staticmethod
    def forward(ctx, kind, est, tgt):
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
    """F.mse_loss(estimated, target): symmetric, so the operand that needs a gradient is fed as the differentiated one
    (FullSubNet calls model.loss(cIRM, cRM) with the network output in the `target` slot, trainer.py:107)."""
    if target.requires_grad and not estimated.requires_grad:
        estimated, target = target, estimated
    return _loss(0, estimated, target)


def sdr(s1, s2, eps=1e-8):
    """tools_for_loss.py:29-33: s1 = target, s2 = estimate.  Returns +SDR (the model negates it, models.py:319)."""
    return -_loss(1, s2, s1)


def si_snr(s1, s2, eps=1e-8):
    """tools_for_loss.py:36-44: s1 = estimate, s2 = target."""
    return -_loss(2, s1, s2)


def si_sdr(reference, estimation, eps=1e-8):
    """tools_for_loss.py:47-94."""
    return -_loss(3, estimation, reference)


# ------------------------------------------------------------------------------------------ LMS (tools_for_loss.py:120-249)
MEL_SCALES = [16, 32, 64]        # cfg.perceptual == 'LMS' (tools_for_loss.py:114-115)


_MEL_K = 1127.01048


def _mel_edges(nb, nbins):
    """FFT-bin edges of nb triangular mel bands over [0, fs/2].  Parity detail (reference tools_for_loss.py:140-165, pinned by
    tests/golden/*lms*): the nb + 2 band centres live in a float32 array, are mapped mel -> Hz one element at a time and floored
    to a bin index through that float32 array, so the edges are what float32 rounding makes them, not the exact ones."""
    top = _MEL_K * math.log(1 + (cfg.fs / 2) / 700.0)
    pts = np.arange(nb + 2).astype(np.float32) * top / (nb + 1) + 0.0
    for i in range(nb + 2):
        pts[i] = 700 * (math.exp(pts[i] / _MEL_K) - 1)
        pts[i] = math.floor(nbins * pts[i] / (cfg.fs / 2))
    return pts.astype(np.int64)


def _band_table(nfft):
    """Sparse form of the three mel banks (MEL_SCALES): one row (first bin, taps, offset into the tap array, scale index) per band
    and the tap weights: rising edge (j - s) / (m - s) on [s, m), falling edge 1 - (j - m) / (e - m) on [m, e)."""
    nbins = cfg.win_len if nfft is None else int(nfft / 2) + 1
    rows, taps = [], []
    for si, nb in enumerate(MEL_SCALES):
        ed = _mel_edges(nb, nbins)
        for s_, m_, e_ in zip(ed[:-2], ed[1:-1], ed[2:]):
            up = (np.arange(s_, m_, dtype=np.float64) - s_) / max(m_ - s_, 1)
            down = 1 - (np.arange(m_, e_, dtype=np.float64) - m_) / max(e_ - m_, 1)
            w = np.concatenate([up, down]).astype(np.float32)
            nz = np.nonzero(w)[0]                                    # the first rising tap is 0: bands start at their first non-zero bin
            lo, n = (int(nz[0]), int(nz[-1] - nz[0] + 1)) if len(nz) else (0, 0)
            rows.append([int(s_) + lo if n else 0, n, len(taps), si])
            taps.extend(w[lo:lo + n].tolist())
    return rows, taps


_BANK_CACHE = {}


def _banks(device, nfft):
    """(band rows, tap weights, number of bands, bands per scale) of the three mel banks on `device`."""
    key = (str(device), nfft)
    if key not in _BANK_CACHE:
        bands, weights = _band_table(nfft)
        _BANK_CACHE[key] = (torch.tensor(bands, dtype=torch.int32, device=device).contiguous(),
                            torch.tensor(weights or [0.0], dtype=torch.float32, device=device), len(bands),
                            (C.c_int32 * len(MEL_SCALES))(*MEL_SCALES))
    return _BANK_CACHE[key]


class _LMS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clean_r, clean_i, est_r, est_i):
        if not est_r.is_cuda:
            raise RuntimeError("sefd LMS loss runs on the MI355X only (cuda tensors); there is no CPU fallback")
        L_ = _lib.lib()
        f = lambda t: None if t is None else t.detach().float().contiguous()
        clean_r, clean_i, er, ei = f(clean_r), f(clean_i), f(est_r), f(est_i)
        B, NF, T = er.shape
        nfft = cfg.fft_len
        bands, weights, nbands, sizes = _banks(er.device, nfft)
        ws = torch.empty(B * T, dtype=torch.float32, device=er.device)
        out = torch.empty((), dtype=torch.float32, device=er.device)
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = L_.sefd_lms_forward(_vp(clean_r), _vp(clean_i), _vp(er), _vp(ei), B, NF, T, _vp(bands), _vp(weights), nbands, sizes,
                                 len(MEL_SCALES), nfft, _vp(ws), _vp(out), stream)
        if rc != 0:
            raise RuntimeError(f"sefd_lms_forward failed ({rc})")
        ctx.t = (clean_r, clean_i, er, ei)
        return out

    @staticmethod
    def backward(ctx, g):
        L_ = _lib.lib()
        clean_r, clean_i, er, ei = ctx.t
        B, NF, T = er.shape
        bands, weights, nbands, sizes = _banks(er.device, cfg.fft_len)
        gr = torch.empty_like(er)
        gi = torch.empty_like(er) if ei is not None else None
        gs = g.float().contiguous().view(1)
        rc = L_.sefd_lms_backward(_vp(clean_r), _vp(clean_i), _vp(er), _vp(ei), B, NF, T, _vp(bands), _vp(weights), nbands, sizes,
                                  len(MEL_SCALES), cfg.fft_len, _vp(gs), _vp(gr), _vp(gi), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sefd_lms_backward failed ({rc})")
        return None, None, gr, gi


def get_array_lms_loss(clean_array, est_array):
    """tools_for_loss.py:242-249: magnitudes [B, 257, T] in, scalar out (flat re-view quirk Q8 reproduced)."""
    return _LMS.apply(clean_array, None, est_array, None)


def lms_from_spectra(clean_real, clean_imag, est_real, est_imag):
    """models.py:306-312 fused: mags = sqrt(re^2 + im^2 + 1e-7) on both sides, then get_array_lms_loss."""
    return _LMS.apply(clean_real, clean_imag, est_real, est_imag)


# ------------------------------------------------------------------------------------------ PMSQE (tools_for_loss.py:253-269)
# The arithmetic is third-party in the reference (asteroid SingleSrcPMSQE under PITLossWrapper('pw_pt') over asteroid_filterbanks'
# STFTFB(512, 512, stride 256) magnitudes) and absent from its tree: parity unpinned; csrc/pmsqe.hip follows the published algorithm.
_PMSQE_CACHE = {}


def _pmsqe_tables(device):
    """(float table, int table) of csrc/pmsqe.hip on `device`: P.862.2 wide-band constants (pmsqe_tables.py), the speech-band mask of
    the SLL mean, and the sqrt-Hann DFT tables of the 512 / 256 analysis (filters scaled by 1 / (0.5 sqrt(512 * 512 / 256)) = 1 / 16)."""
    key = str(device)
    if key not in _PMSQE_CACHE:
        from . import pmsqe_tables as pt
        nfft, hop, nbins, nb = 512, 256, 257, 49
        thr = np.array(pt.ABS_THRESH_POWER)
        cb = np.array(pt.CENTRE_OF_BAND_BARK)
        zp = 0.23 * np.minimum(2.0, np.where(cb >= 4, 1.0, 6.0 / (cb + 2.0))) ** 0.15          # P.862 modified Zwicker power
        aterm = pt.SL_16K * (thr / 0.5) ** zp
        mask = np.zeros(nbins)
        mask[11], mask[12:104], mask[104] = 0.5 * 25.0 / 31.25, 1.0, 0.5                          # 350 .. 3250 Hz
        mask *= 2.0 * (nfft + 2.0) / nfft ** 2                                                    # sqrt-Hann power correction 2.0
        n = np.arange(nfft)
        win = np.hanning(nfft + 1)[:-1] ** 0.5
        ang = 2 * np.pi * np.outer(n, np.arange(nbins)) / nfft
        scale = 0.5 * math.sqrt(nfft * nfft / hop)
        Cm, Sm = np.cos(ang) * win[:, None] / scale, -np.sin(ang) * win[:, None] / scale          # [512][257]
        tab = np.zeros(int(_lib.lib().sefd_pmsqe_table_floats()), np.float32)
        for off, v in ((0, thr), (49, zp), (98, pt.WIDTH_OF_BAND_BARK), (147, pt.POW_DENS_CORRECTION), (196, aterm), (245, mask)):
            tab[off:off + len(v)] = v
        o = 512
        for m in (Cm, Sm, Cm.T, Sm.T):
            tab[o:o + m.size] = np.ascontiguousarray(m).ravel()
            o += m.size
        itab = np.full(64 + nbins, -1, np.int32)
        lo = np.concatenate([[0], np.cumsum(pt.HZ_BINS_PER_BAND)])
        itab[:nb + 1] = lo
        itab[50:64] = 0
        for k in range(nb):
            itab[64 + lo[k]:64 + lo[k + 1]] = k
        _PMSQE_CACHE[key] = (torch.from_numpy(tab).to(device), torch.from_numpy(itab).to(device))
    return _PMSQE_CACHE[key]


class _PMSQE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, clean, est, power):
        if not est.is_cuda:
            raise RuntimeError("sefd PMSQE loss runs on the MI355X only (cuda tensors); there is no CPU fallback")
        L_ = _lib.lib()
        clean, est = clean.detach().float().contiguous(), est.detach().float().contiguous()
        if est.dim() != 2 or clean.shape != est.shape:
            raise ValueError("get_array_pmsqe_loss takes two [N, L] wave batches")
        B, L = est.shape
        n = L_.sefd_pmsqe_ws_floats(B, L)
        if n < 0:
            raise ValueError(f"PMSQE needs whole seconds of {cfg.fs} samples (the reference's view(N, -1, fs)), at most 6: got L = {L}")
        tab, itab = _pmsqe_tables(est.device)
        ws = torch.empty(n, dtype=torch.float32, device=est.device)
        out = torch.empty((), dtype=torch.float32, device=est.device)
        rc = L_.sefd_pmsqe_forward(_vp(est), _vp(clean), B, L, int(power), _vp(tab), _vp(itab), _vp(ws), _vp(out),
                                   C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sefd_pmsqe_forward failed ({rc})")
        ctx.t = (B, L, int(power), tab, itab, ws)
        return out

    @staticmethod
    def backward(ctx, g):
        B, L, power, tab, itab, ws = ctx.t
        ge = torch.empty(B, L, dtype=torch.float32, device=ws.device)
        gs = g.float().contiguous().view(1)
        rc = _lib.lib().sefd_pmsqe_backward(B, L, power, _vp(tab), _vp(itab), _vp(ws), _vp(gs), _vp(ge),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sefd_pmsqe_backward failed ({rc})")
        return None, ge, None


def get_array_pmsqe_loss(clean_array, est_array):
    """tools_for_loss.py:258-269: [N, L] (or [N, 1, L]) waves, L whole seconds -> scalar.  `cfg.pmsqe_power` (build-side knob, default
    False): False is the reference's literal call chain - the loss is handed the `transforms.mag` magnitudes (tools_for_loss.py:267-269); True is
    an opt-in variant that works on the power spectrum |X|^2, the quantity the published algorithm and its P.862 constants are defined on
    (+0.29 held-out PESQ on synthetic data; not the reference's behaviour)."""
    if clean_array.dim() == 3:
        clean_array, est_array = clean_array.flatten(1), est_array.flatten(1)
    return _PMSQE.apply(clean_array, est_array, bool(getattr(cfg, "pmsqe_power", False)))
