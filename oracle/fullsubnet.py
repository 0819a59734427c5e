"""Oracle (test infrastructure): FullSubNet restated on CPU PyTorch.

Reference: models.py:568-682 (FullSubNet), tools_for_model.py:628-723 (stft, mag_phase, cIRM), :726-795 (SequenceModel),
:806-837 (unfold), :997-1011 (offline_laplace_norm); step driver trainer.py:85-118.
Dropout (p = 0.8 between the two LSTM layers, active in train mode) is modelled by an explicit keep-mask
argument so that parity can be checked deterministically: mask=None means no dropout (eval, or dropout patched to 0).
"""
from collections import OrderedDict
from dataclasses import dataclass
import numpy as np
import torch
import torch.nn.functional as F

from .dccrn import lstm_layer
from .frontend import torch_stft

EPSILON = float(np.finfo(np.float32).eps)


@dataclass
class FSNConfig:
    sb_num_neighbors: int = 15
    fb_num_neighbors: int = 0
    num_freqs: int = 257
    look_ahead: int = 2
    fb_hidden: int = 512
    sb_hidden: int = 384
    fb_act: str = "ReLU"
    sb_act: str = None
    sequence_model: str = "LSTM"
    norm_type: str = "offline_laplace_norm"
    n_fft: int = 512
    hop: int = 300
    win_len: int = 400


def fsn_state_shapes(cfg: FSNConfig) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    G = 4 if cfg.sequence_model == "LSTM" else 3                  # gate blocks: nn.LSTM i,f,g,o / nn.GRU r,z,n
    def seq(prefix, I, H, O):
        for l in range(2):
            s[f"{prefix}.sequence_model.weight_ih_l{l}"] = (G * H, I if l == 0 else H)
            s[f"{prefix}.sequence_model.weight_hh_l{l}"] = (G * H, H)
            s[f"{prefix}.sequence_model.bias_ih_l{l}"] = (G * H,)
            s[f"{prefix}.sequence_model.bias_hh_l{l}"] = (G * H,)
        s[f"{prefix}.fc_output_layer.weight"] = (O, H)
        s[f"{prefix}.fc_output_layer.bias"] = (O,)
    seq("fb_model", cfg.num_freqs, cfg.fb_hidden, cfg.num_freqs)
    seq("sb_model", (cfg.sb_num_neighbors * 2 + 1) + (cfg.fb_num_neighbors * 2 + 1), cfg.sb_hidden, 2)
    return s


def build_cirm(noisy: torch.Tensor, clean: torch.Tensor, K=10.0, C=0.1) -> torch.Tensor:
    """tools_for_model.py:687-717: complex tensors [B,F,T] -> compressed cIRM [B,F,T,2]."""
    den = noisy.real ** 2 + noisy.imag ** 2 + EPSILON
    mr = (noisy.real * clean.real + noisy.imag * clean.imag) / den
    mi = (noisy.real * clean.imag - noisy.imag * clean.real) / den
    m = torch.stack((mr, mi), -1)
    m = -100 * (m <= -100) + m * (m > -100)
    return K * (1 - torch.exp(-C * m)) / (1 + torch.exp(-C * m))


def decompress_cirm(mask, K=10.0, limit=9.9):
    """tools_for_model.py:720-723."""
    mask = limit * (mask >= limit) - limit * (mask <= -limit) + mask * (torch.abs(mask) < limit)
    return -K * torch.log((K - mask) / (K + mask))


def unfold(x, n):
    """tools_for_model.py:806-837: [B,C,F,T] -> [B,F,C,2n+1,T] with reflect padding along F."""
    B, Cc, Fq, T = x.shape
    if n < 1:
        return x.permute(0, 2, 1, 3).reshape(B, Fq, Cc, 1, T)
    o = F.pad(x.reshape(B * Cc, 1, Fq, T), [0, 0, n, n], mode="reflect")
    o = F.unfold(o, (2 * n + 1, T))
    return o.reshape(B, Cc, 2 * n + 1, T, Fq).permute(0, 4, 1, 2, 3).contiguous()


def laplace_norm(x):
    return x / (torch.mean(x, dim=(1, 2, 3), keepdim=True) + 1e-5)


def _cumulative_stats(x):
    """[B, C, F, T] -> ([B*C, F, T] view, running sum over (F, frames <= t) [B*C, T], running count [1, T])."""
    B, Cc, Fq, T = x.shape
    v = x.reshape(B * Cc, Fq, T)
    count = torch.arange(Fq, Fq * T + 1, Fq, dtype=x.dtype).reshape(1, T)
    return v, count


def norm(x, kind):
    """BaseModel.norm_wrapper's four choices (tools_for_model.py:997-1118), x [B, C, F, T]."""
    if kind == "offline_laplace_norm":
        return laplace_norm(x)
    if kind == "offline_gaussian_norm":                           # :1047-1061 (torch.std: unbiased)
        mu = torch.mean(x, dim=(1, 2, 3), keepdim=True)
        return (x - mu) / (torch.std(x, dim=(1, 2, 3), keepdim=True) + 1e-5)
    v, count = _cumulative_stats(x)
    csum = torch.cumsum(v.sum(1), -1)
    mean = csum / count
    if kind == "cumulative_laplace_norm":                         # :1014-1044
        return (v / (mean.unsqueeze(1) + EPSILON)).reshape(x.shape)
    if kind == "cumulative_layer_norm":                           # :1064-1104
        cpow = torch.cumsum(v.square().sum(1), -1)
        var = (cpow - 2 * mean * csum) / count + mean.pow(2)
        return ((v - mean.unsqueeze(1)) / torch.sqrt(var + EPSILON).unsqueeze(1)).reshape(x.shape)
    raise NotImplementedError(kind)


def gru_layer(x, w_ih, w_hh, b_ih, b_hh):
    """One nn.GRU layer, zero initial state (tools_for_model.py:748-756); x [T, N, I] -> [T, N, H]."""
    h0 = x.new_zeros(1, x.shape[1], w_hh.shape[1])
    return torch._VF.gru(x, h0, [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, False, False, False)[0]


def sequence_model(P, prefix, x, act, keep_mask=None, p_drop=0.8, kind="LSTM"):
    """SequenceModel.forward (tools_for_model.py:779-795); x [N, F, T] -> [N, O, T].  keep_mask [N, T, H] of {0,1} or None."""
    xt = x.permute(0, 2, 1).permute(1, 0, 2)                      # [T, N, F]
    g = lambda l, k: P[f"{prefix}.sequence_model.{k}_l{l}"]
    layer = lstm_layer if kind == "LSTM" else gru_layer
    h = layer(xt, g(0, "weight_ih"), g(0, "weight_hh"), g(0, "bias_ih"), g(0, "bias_hh"))
    if keep_mask is not None:
        h = h * keep_mask.permute(1, 0, 2) / (1.0 - p_drop)
    h = layer(h, g(1, "weight_ih"), g(1, "weight_hh"), g(1, "bias_ih"), g(1, "bias_hh"))
    o = F.linear(h.permute(1, 0, 2), P[f"{prefix}.fc_output_layer.weight"], P[f"{prefix}.fc_output_layer.bias"])
    if act == "ReLU":
        o = torch.relu(o)
    elif act == "Tanh":
        o = torch.tanh(o)
    elif act == "ReLU6":
        o = F.relu6(o)
    return o.permute(0, 2, 1)


def fsn_forward(P, noisy_mag, cfg: FSNConfig, fb_mask=None, sb_mask=None, taps=None):
    """FullSubNet.forward (models.py:626-672): noisy_mag [B,F,T] -> [B,F,T,2]."""
    x = noisy_mag.unsqueeze(1)
    x = F.pad(x, [0, cfg.look_ahead])
    B, Cc, Fq, T = x.shape
    fb_in = norm(x, cfg.norm_type).reshape(B, Cc * Fq, T)
    fb_out = sequence_model(P, "fb_model", fb_in, cfg.fb_act, fb_mask, kind=cfg.sequence_model).reshape(B, 1, Fq, T)
    if taps is not None:
        taps["fb_out"] = fb_out
    fbu = unfold(fb_out, cfg.fb_num_neighbors).reshape(B, Fq, cfg.fb_num_neighbors * 2 + 1, T)
    nmu = unfold(x, cfg.sb_num_neighbors).reshape(B, Fq, cfg.sb_num_neighbors * 2 + 1, T)
    sb_in = norm(torch.cat([nmu, fbu], 2), cfg.norm_type)
    if taps is not None:
        taps["sb_in"] = sb_in
    sb_in = sb_in.reshape(B * Fq, -1, T)
    sb = sequence_model(P, "sb_model", sb_in, cfg.sb_act, sb_mask, kind=cfg.sequence_model)
    sb = sb.reshape(B, Fq, 2, T).permute(0, 2, 1, 3).contiguous()
    return sb[:, :, :, cfg.look_ahead:].permute(0, 2, 3, 1)


def fsn_targets(inputs, targets, cfg: FSNConfig):
    """trainer.py:100-104: (noisy_mag [B,F,T], cIRM [B,F,T,2])."""
    nc = torch_stft(inputs, cfg.n_fft, cfg.hop, cfg.win_len)
    cc = torch_stft(targets, cfg.n_fft, cfg.hop, cfg.win_len)
    return torch.abs(nc), build_cirm(nc, cc)
