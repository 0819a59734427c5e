"""Oracle (test infrastructure): CRN forward restated functionally on CPU PyTorch (reference models.py:329-565).

Real twin of DCCRN on magnitudes: RealConv2d / RealConvTranspose2d (tools_for_model.py:341-425) with half the channels,
ONE-layer nn.LSTM(rnn_input_size -> rnn_units//2) + Linear `tranform` (sic), plain channel concat for the skips,
mask = tanh(out) * mags re-attached to the noisy phase.  `CRN.forward` always computes stft(targets) (models.py:505).
"""
from collections import OrderedDict
from dataclasses import dataclass
import torch
import torch.nn.functional as F

from .dccrn import batch_norm_eval, batch_norm_train, lstm_layer, prelu
from .frontend import conv_istft, conv_stft


@dataclass
class CRNConfig:
    kernel_num: tuple = (32, 64, 128, 256, 256, 256)     # cfg.dccrn_kernel_num (CRN uses half of each)
    rnn_input_size: int = 512
    rnn_units: int = 256                                 # CRN uses rnn_units // 2 (models.py:358)
    win_len: int = 400
    win_inc: int = 100
    fft_len: int = 512
    masking_mode: str = "E"
    skip_type: bool = True
    kernel_size: int = 5

    @property
    def chans(self):
        return (1,) + tuple(k // 2 for k in self.kernel_num)


def crn_state_shapes(cfg: CRNConfig) -> "OrderedDict[str, tuple]":
    s = OrderedDict()
    nb = cfg.fft_len + 2
    s["stft.weight"] = (nb, 1, cfg.win_len)
    s["istft.weight"] = (nb, 1, cfg.win_len)
    s["istft.window"] = (1, cfg.win_len, 1)
    s["istft.enframe"] = (cfg.win_len, 1, cfg.win_len)
    ch = cfg.chans
    n = len(ch) - 1
    for i in range(n):
        s[f"encoder.{i}.0.conv.weight"] = (ch[i + 1], ch[i], cfg.kernel_size, 2)
        s[f"encoder.{i}.0.conv.bias"] = (ch[i + 1],)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[f"encoder.{i}.1.{leaf}"] = (ch[i + 1],)
        s[f"encoder.{i}.1.num_batches_tracked"] = ()
        s[f"encoder.{i}.2.weight"] = (1,)
    for d, idx in enumerate(range(n, 0, -1)):
        cin = ch[idx] * (2 if cfg.skip_type else 1) if idx >= 1 else ch[idx]
        cout = ch[idx - 1]
        s[f"decoder.{d}.0.conv.weight"] = (cin, cout, cfg.kernel_size, 2)
        s[f"decoder.{d}.0.conv.bias"] = (cout,)
        if idx != 1:
            for leaf in ("weight", "bias", "running_mean", "running_var"):
                s[f"decoder.{d}.1.{leaf}"] = (cout,)
            s[f"decoder.{d}.1.num_batches_tracked"] = ()
            s[f"decoder.{d}.2.weight"] = (1,)
    H = cfg.rnn_units // 2
    s["enhance.weight_ih_l0"] = (4 * H, cfg.rnn_input_size)
    s["enhance.weight_hh_l0"] = (4 * H, H)
    s["enhance.bias_ih_l0"] = (4 * H,)
    s["enhance.bias_hh_l0"] = (4 * H,)
    s["tranform.weight"] = (cfg.rnn_input_size, H)
    s["tranform.bias"] = (cfg.rnn_input_size,)
    return s


def crn_forward(P, inputs, targets, cfg: CRNConfig, train=True, taps=None):
    """Returns ((est_mags | out, target_mags, out_wav), new_running_stats)."""
    nfreq = cfg.fft_len // 2 + 1
    new_stats = {}
    specs = conv_stft(inputs, cfg.win_len, cfg.win_inc, cfg.fft_len)
    real, imag = specs[:, :nfreq], specs[:, nfreq:]
    mags = torch.sqrt(real ** 2 + imag ** 2)                     # ConvSTFT 'real': no eps (tools_for_model.py:66)
    phase = torch.atan2(imag, real)
    out = mags.unsqueeze(1)[:, :, 1:]

    def bn(x, pfx):
        if train:
            y, nrm, nrv = batch_norm_train(x, P[pfx + ".weight"], P[pfx + ".bias"], P[pfx + ".running_mean"], P[pfx + ".running_var"])
            new_stats[pfx + ".running_mean"], new_stats[pfx + ".running_var"] = nrm, nrv
            return y
        return batch_norm_eval(x, P[pfx + ".weight"], P[pfx + ".bias"], P[pfx + ".running_mean"], P[pfx + ".running_var"])

    enc_out = []
    n = len(cfg.kernel_num)
    for i in range(n):
        out = F.conv2d(F.pad(out, [1, 0, 0, 0]), P[f"encoder.{i}.0.conv.weight"], P[f"encoder.{i}.0.conv.bias"], stride=(2, 1), padding=(2, 0))
        if taps is not None:
            taps[f"enc{i}.conv"] = out
        out = prelu(bn(out, f"encoder.{i}.1"), P[f"encoder.{i}.2.weight"])
        enc_out.append(out)
    B, C, D, T = out.shape
    r = out.permute(3, 0, 1, 2).reshape(T, B, C * D)
    r = lstm_layer(r, P["enhance.weight_ih_l0"], P["enhance.weight_hh_l0"], P["enhance.bias_ih_l0"], P["enhance.bias_hh_l0"])
    r = F.linear(r, P["tranform.weight"], P["tranform.bias"])
    if taps is not None:
        taps["lstm"] = r
    out = r.reshape(T, B, C, D).permute(1, 2, 3, 0)
    for d in range(n):
        if cfg.skip_type:
            out = torch.cat([out, enc_out[-1 - d]], 1)
        out = F.conv_transpose2d(out, P[f"decoder.{d}.0.conv.weight"], P[f"decoder.{d}.0.conv.bias"], stride=(2, 1), padding=(2, 0),
                                 output_padding=(1, 0))
        if taps is not None:
            taps[f"dec{d}.conv"] = out
        if d != n - 1:
            out = prelu(bn(out, f"decoder.{d}.1"), P[f"decoder.{d}.2.weight"])
        out = out[..., 1:]
    out = F.pad(out.squeeze(1), [0, 0, 1, 0])
    tspec = conv_stft(targets, cfg.win_len, cfg.win_inc, cfg.fft_len)
    target_mags = torch.sqrt(tspec[:, :nfreq] ** 2 + tspec[:, nfreq:] ** 2)
    if cfg.masking_mode == "Direct(None make)":
        first = out
        est = out
    else:
        est = torch.tanh(out) * mags
        first = est
    o_r, o_i = est * torch.cos(phase), est * torch.sin(phase)
    wav = conv_istft(torch.cat([o_r, o_i], 1), cfg.win_len, cfg.win_inc, cfg.fft_len).squeeze(1)
    wav = torch.clamp(wav, -1, 1)
    return (first, target_mags, wav), new_stats
