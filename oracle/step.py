"""Oracle (test infrastructure): one training step = forward -> loss -> backward -> Adam.

Restates trainer.py:23-39 (model_train) / :45-82 (model_perceptual_train) with
torch.optim.Adam defaults (train_interface.py:59): betas (0.9, 0.999), eps 1e-8, no weight decay.
Autograd on CPU provides the backward; Adam is written out explicitly.
"""
import torch

from .dccrn import DCCRNConfig, dccrn_forward, is_trainable
from .frontend import conv_stft
from .losses import main_loss, lms_loss


def adam_update(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor formula (no amsgrad / weight decay). `step` is 1-based."""
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / (bc2 ** 0.5)) + eps
    return p - (lr / bc1) * (m / denom), m, v


def dccrn_loss(cfg: DCCRNConfig, loss_kind, perceptual, outputs, targets):
    """trainer.py:30-31 / :61-69 + models.py:303-323."""
    o_r, o_i, wav = outputs
    main = main_loss(loss_kind, wav, targets)
    if not perceptual:
        return main
    if perceptual == "LMS":
        nfreq = cfg.fft_len // 2 + 1
        cs = conv_stft(targets, cfg.win_len, cfg.win_inc, cfg.fft_len, cfg.win_type)      # self.stft(target): the MODEL's window (models.py:306-308)
        clean_mags = torch.sqrt(cs[:, :nfreq] ** 2 + cs[:, nfreq:] ** 2 + 1e-7)
        est_mags = torch.sqrt(o_r ** 2 + o_i ** 2 + 1e-7)
        return (main + lms_loss(clean_mags, est_mags)) / 2
    raise NotImplementedError("PMSQE: third-party arithmetic, parity unpinned")


def dccrn_train_step(P, cfg: DCCRNConfig, inputs, targets, loss_kind="SI-SNR", perceptual=False,
                     adam_state=None, step=1, lr=1e-3):
    """Returns dict(loss, grads{name}, new_params{name}, new_stats{name}, adam_state, outputs)."""
    Pg = {k: (v.detach().clone().requires_grad_(True) if is_trainable(k) else v) for k, v in P.items()}
    outputs, new_stats = dccrn_forward(Pg, inputs, cfg, targets=targets, train=True)
    loss = dccrn_loss(cfg, loss_kind, perceptual, outputs, targets)
    names = [k for k in Pg if is_trainable(k)]
    grads = torch.autograd.grad(loss, [Pg[k] for k in names], allow_unused=True)
    grads = {k: (g if g is not None else torch.zeros_like(Pg[k])) for k, g in zip(names, grads)}
    if adam_state is None:
        adam_state = {k: (torch.zeros_like(P[k]), torch.zeros_like(P[k])) for k in names}
    new_params, new_state = {}, {}
    for k in names:
        m, v = adam_state[k]
        new_params[k], m, v = adam_update(P[k].detach(), grads[k], m, v, step, lr)
        new_state[k] = (m, v)
    return dict(loss=loss.detach(), grads=grads, new_params=new_params, new_stats=new_stats,
                adam_state=new_state, outputs=tuple(o.detach() for o in outputs))
