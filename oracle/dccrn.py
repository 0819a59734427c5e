"""Oracle (test infrastructure): DCCRN / CRN forward restated functionally on CPU PyTorch.

Reference call sites restated here:
  DCCRN.__init__ / forward     models.py:15-284
  CRN.__init__ / forward       models.py:329-532
  ComplexConv2d                tools_for_model.py:199-269   (causal left pad 1, k(5,2) s(2,1) p(2,0))
  ComplexConvTranspose2d       tools_for_model.py:272-338   (k(5,2) s(2,1) p(2,0) op(1,0), caller drops frame 0)
  NavieComplexLSTM             tools_for_model.py:141-181
  complex_cat                  tools_for_model.py:184-193
Parameters are a plain {state_dict key: tensor} mapping with the reference's key names/shapes
(SURVEY.md Appendix B), so the same dict drives the reference module, this oracle and the HIP module.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
import torch
import torch.nn.functional as F

from .frontend import conv_stft, conv_istft


@dataclass
class DCCRNConfig:
    """Mirror of the config.py knobs that shape DCCRN (config.py:35-68)."""
    kernel_num: tuple = (32, 64, 128, 256, 256, 256)
    rnn_layers: int = 2
    rnn_units: int = 256
    win_len: int = 400
    win_inc: int = 100
    fft_len: int = 512
    masking_mode: str = "E"
    lstm: str = "complex"
    skip_type: bool = True
    kernel_size: int = 5
    win_type: object = "hanning"  # ConvSTFT / ConviSTFT window: 'hanning' (cfg.window) or None = rectangular (tools_for_model.py:17-18)
    use_cbn: bool = False        # DCCRN(use_cbn=True): ComplexBatchNorm (tools_for_model.py:430-607) instead of nn.BatchNorm2d

    @property
    def chans(self):
        return (2,) + tuple(self.kernel_num)

    @property
    def hidden_dim(self):
        return self.fft_len // (2 ** (len(self.kernel_num) + 1))


CBN_PARAMS = ("Wrr", "Wri", "Wii", "Br", "Bi")
CBN_BUFFERS = ("RMr", "RMi", "RVrr", "RVri", "RVii")


def _norm_leaves(cfg, C):
    """(leaf, shape) of the normalisation behind a conv, registration order (BatchNorm2d, or ComplexBatchNorm tools_for_model.py:441-467)."""
    if cfg.use_cbn:
        return [(leaf, (C // 2,)) for leaf in CBN_PARAMS + CBN_BUFFERS]
    return [(leaf, (C,)) for leaf in ("weight", "bias", "running_mean", "running_var")]


def dccrn_state_shapes(cfg: DCCRNConfig) -> "OrderedDict[str, tuple]":
    """state_dict keys and shapes in the reference's registration order."""
    s = OrderedDict()
    nb = cfg.fft_len + 2
    s["stft.weight"] = (nb, 1, cfg.win_len)
    s["istft.weight"] = (nb, 1, cfg.win_len)
    s["istft.window"] = (1, cfg.win_len, 1)
    s["istft.enframe"] = (cfg.win_len, 1, cfg.win_len)
    ch = cfg.chans
    for i in range(len(ch) - 1):
        ci, co = ch[i] // 2, ch[i + 1] // 2
        for part in ("real_conv", "imag_conv"):
            s[f"encoder.{i}.0.{part}.weight"] = (co, ci, cfg.kernel_size, 2)
            s[f"encoder.{i}.0.{part}.bias"] = (co,)
        for leaf, shp in _norm_leaves(cfg, ch[i + 1]):
            s[f"encoder.{i}.1.{leaf}"] = shp
        s[f"encoder.{i}.1.num_batches_tracked"] = ()
        s[f"encoder.{i}.2.weight"] = (1,)
    # registration order in DCCRN.__init__: encoder and decoder ModuleLists exist before `enhance` is assigned
    n = len(ch) - 1
    for d, idx in enumerate(range(n, 0, -1)):
        cin = ch[idx] * (2 if cfg.skip_type else 1)
        cout = ch[idx - 1]
        for part in ("real_conv", "imag_conv"):
            s[f"decoder.{d}.0.{part}.weight"] = (cin // 2, cout // 2, cfg.kernel_size, 2)
            s[f"decoder.{d}.0.{part}.bias"] = (cout // 2,)
        if idx != 1:
            for leaf, shp in _norm_leaves(cfg, cout):
                s[f"decoder.{d}.1.{leaf}"] = shp
            s[f"decoder.{d}.1.num_batches_tracked"] = ()
            s[f"decoder.{d}.2.weight"] = (1,)
    hid = cfg.hidden_dim * ch[-1]
    if cfg.lstm == "complex":
        H = cfg.rnn_units // 2
        for l in range(cfg.rnn_layers):
            I = (hid if l == 0 else cfg.rnn_units) // 2
            for part in ("real_lstm", "imag_lstm"):
                s[f"enhance.{l}.{part}.weight_ih_l0"] = (4 * H, I)
                s[f"enhance.{l}.{part}.weight_hh_l0"] = (4 * H, H)
                s[f"enhance.{l}.{part}.bias_ih_l0"] = (4 * H,)
                s[f"enhance.{l}.{part}.bias_hh_l0"] = (4 * H,)
            if l == cfg.rnn_layers - 1:
                for part in ("r_trans", "i_trans"):
                    s[f"enhance.{l}.{part}.weight"] = (hid // 2, H)
                    s[f"enhance.{l}.{part}.bias"] = (hid // 2,)
    else:
        H = cfg.rnn_units
        for l in range(2):
            I = hid if l == 0 else H
            s[f"enhance.weight_ih_l{l}"] = (4 * H, I)
            s[f"enhance.weight_hh_l{l}"] = (4 * H, H)
            s[f"enhance.bias_ih_l{l}"] = (4 * H,)
            s[f"enhance.bias_hh_l{l}"] = (4 * H,)
        s["tranform.weight"] = (hid, H)
        s["tranform.bias"] = (hid,)
    return s


BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked") + CBN_BUFFERS


def is_trainable(name: str) -> bool:
    return not (name.startswith(("stft.", "istft.")) or name.split(".")[-1] in BUFFER_LEAVES)


# ------------------------------------------------------------------ building blocks
def complex_conv2d(x, wr, br, wi, bi):
    """tools_for_model.py:243-269 with complex_axis=1, causal."""
    x = F.pad(x, [1, 0, 0, 0])
    xr, xi = torch.chunk(x, 2, 1)
    conv = lambda a, w, b: F.conv2d(a, w, b, stride=(2, 1), padding=(2, 0))
    real = conv(xr, wr, br) - conv(xi, wi, bi)
    imag = conv(xr, wi, bi) + conv(xi, wr, br)
    return torch.cat([real, imag], 1)


def complex_deconv2d(x, wr, br, wi, bi):
    """tools_for_model.py:311-338."""
    xr, xi = torch.chunk(x, 2, 1)
    dc = lambda a, w, b: F.conv_transpose2d(a, w, b, stride=(2, 1), padding=(2, 0), output_padding=(1, 0))
    real = dc(xr, wr, br) - dc(xi, wi, bi)
    imag = dc(xr, wi, bi) + dc(xi, wr, br)
    return torch.cat([real, imag], 1)


def complex_cat(a, b):
    """tools_for_model.py:184-193 for two inputs on axis 1."""
    ar, ai = torch.chunk(a, 2, 1)
    br_, bi_ = torch.chunk(b, 2, 1)
    return torch.cat([ar, br_, ai, bi_], 1)


def batch_norm_train(x, w, b, rm, rv, eps=1e-5, momentum=0.1):
    """nn.BatchNorm2d in training mode; returns (y, new_running_mean, new_running_var)."""
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    y = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + eps)
    y = y * w[None, :, None, None] + b[None, :, None, None]
    with torch.no_grad():
        nrm = (1 - momentum) * rm + momentum * mean
        nrv = (1 - momentum) * rv + momentum * var * (n / max(n - 1, 1))
    return y, nrm, nrv


def batch_norm_eval(x, w, b, rm, rv, eps=1e-5):
    y = (x - rm[None, :, None, None]) / torch.sqrt(rv[None, :, None, None] + eps)
    return y * w[None, :, None, None] + b[None, :, None, None]


def complex_batch_norm(x, P, pfx, train, eps=1e-5, momentum=0.1):
    """ComplexBatchNorm.forward (tools_for_model.py:487-601) on NCHW x = [real C/2 | imag C/2]; returns (y, new running statistics or {}).
    Whitening by the inverse square root of the per-channel 2 x 2 covariance (closed form), then y = W U (x - M) + B."""
    xr, xi = torch.chunk(x, 2, dim=1)
    v = lambda t: t.view(1, -1, 1, 1)
    new = {}
    if train:
        Mr, Mi = xr.mean((0, 2, 3), keepdim=True), xi.mean((0, 2, 3), keepdim=True)
    else:
        Mr, Mi = v(P[pfx + ".RMr"]), v(P[pfx + ".RMi"])
    xr, xi = xr - Mr, xi - Mi
    if train:
        Vrr, Vri, Vii = (xr * xr).mean((0, 2, 3), keepdim=True), (xr * xi).mean((0, 2, 3), keepdim=True), (xi * xi).mean((0, 2, 3), keepdim=True)
        with torch.no_grad():                     # Tensor.lerp_(batch value, momentum); covariance biased and WITHOUT eps (lines 541-556)
            for leaf, val in (("RMr", Mr), ("RMi", Mi), ("RVrr", Vrr), ("RVri", Vri), ("RVii", Vii)):
                old = P[pfx + "." + leaf]
                new[pfx + "." + leaf] = old + momentum * (val.detach().reshape(-1) - old)
    else:
        Vrr, Vri, Vii = v(P[pfx + ".RVrr"]), v(P[pfx + ".RVri"]), v(P[pfx + ".RVii"])
    Vrr, Vii = Vrr + eps, Vii + eps
    tau, delta = Vrr + Vii, Vrr * Vii - Vri * Vri
    s = delta.sqrt()
    t = (tau + 2 * s).sqrt()
    rst = (s * t).reciprocal()
    Urr, Uii, Uri = (s + Vii) * rst, (s + Vrr) * rst, -Vri * rst
    Wrr, Wri, Wii = v(P[pfx + ".Wrr"]), v(P[pfx + ".Wri"]), v(P[pfx + ".Wii"])
    Zrr, Zri = Wrr * Urr + Wri * Uri, Wrr * Uri + Wri * Uii
    Zir, Zii = Wri * Urr + Wii * Uri, Wri * Uri + Wii * Uii
    yr = Zrr * xr + Zri * xi + v(P[pfx + ".Br"])
    yi = Zir * xr + Zii * xi + v(P[pfx + ".Bi"])
    return torch.cat([yr, yi], 1), new


def prelu(x, a):
    return torch.where(x > 0, x, a * x)


def lstm_layer_loop(x, w_ih, w_hh, b_ih, b_hh):
    """Single-layer unidirectional nn.LSTM, zero initial state, written out step by step; x [T, B, I] -> [T, B, H].
    Gate order i,f,g,o.  Kept as the readable statement of the cell; `lstm_layer` is what the oracle runs."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    gx = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs = []
    for t in range(T):
        g = gx[t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, 0)


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh):
    """The same cell through ATen's fused CPU LSTM (what nn.LSTM.forward calls, tools_for_model.py:167-170): one call for all T
    steps instead of T Python iterations, differentiable; tests/test_oracle_golden.py pins it against `lstm_layer_loop`."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    z = x.new_zeros(1, B, H)
    out, _, _ = torch._VF.lstm(x, (z, z), [w_ih, w_hh, b_ih, b_hh], True, 1, 0.0, False, False, False)
    return out


def complex_lstm(xr, xi, P, prefix, project):
    """NavieComplexLSTM.forward (tools_for_model.py:162-177)."""
    pr = lambda part: [P[f"{prefix}.{part}.{k}"] for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    r2r = lstm_layer(xr, *pr("real_lstm"))
    r2i = lstm_layer(xr, *pr("imag_lstm"))
    i2r = lstm_layer(xi, *pr("real_lstm"))
    i2i = lstm_layer(xi, *pr("imag_lstm"))
    ro = r2r - i2i
    io = i2r + r2i
    if project:
        ro = F.linear(ro, P[f"{prefix}.r_trans.weight"], P[f"{prefix}.r_trans.bias"])
        io = F.linear(io, P[f"{prefix}.i_trans.weight"], P[f"{prefix}.i_trans.bias"])
    return ro, io


def apply_mask(mode, real, imag, mask_real, mask_imag):
    """models.py:258-276; `real`/`imag` are the noisy spectra [B,257,T], masks already DC-padded."""
    if mode == "E":
        spec_mags = torch.sqrt(real ** 2 + imag ** 2 + 1e-8)
        spec_phase = torch.atan2(imag, real)
        mask_mags = (mask_real ** 2 + mask_imag ** 2) ** 0.5
        real_phase = mask_real / (mask_mags + 1e-8)
        imag_phase = mask_imag / (mask_mags + 1e-8)
        mask_phase = torch.atan2(imag_phase, real_phase)
        mask_mags = torch.tanh(mask_mags)
        est_mags = mask_mags * spec_mags
        est_phase = spec_phase + mask_phase
        return est_mags * torch.cos(est_phase), est_mags * torch.sin(est_phase)
    if mode == "C":
        return real * mask_real - imag * mask_imag, real * mask_imag + imag * mask_real
    if mode == "R":
        return real * mask_real, imag * mask_imag
    raise ValueError(mode)


def dccrn_forward(P, inputs, cfg: DCCRNConfig, targets=None, train=True, taps=None):
    """DCCRN.forward (models.py:176-284).

    Returns ((out_real, out_imag, out_wav) or the Direct 5-tuple, new_running_stats dict).
    `taps`, if a dict, receives intermediate activations (reference NCHW layout) for layer-wise parity tests.
    """
    nfreq = cfg.fft_len // 2 + 1
    new_stats = {}
    specs = conv_stft(inputs, cfg.win_len, cfg.win_inc, cfg.fft_len, cfg.win_type)
    real, imag = specs[:, :nfreq], specs[:, nfreq:]
    out = torch.stack([real, imag], 1)[:, :, 1:]
    if taps is not None:
        taps["spec"] = specs

    def bn(x, pfx):
        if cfg.use_cbn:
            y, nst = complex_batch_norm(x, P, pfx, train)
            new_stats.update(nst)
            return y
        if train:
            y, nrm, nrv = batch_norm_train(x, P[pfx + ".weight"], P[pfx + ".bias"],
                                           P[pfx + ".running_mean"], P[pfx + ".running_var"])
            new_stats[pfx + ".running_mean"], new_stats[pfx + ".running_var"] = nrm, nrv
            return y
        return batch_norm_eval(x, P[pfx + ".weight"], P[pfx + ".bias"], P[pfx + ".running_mean"], P[pfx + ".running_var"])

    enc_out = []
    nlayer = len(cfg.kernel_num)
    for i in range(nlayer):
        p = f"encoder.{i}.0"
        out = complex_conv2d(out, P[p + ".real_conv.weight"], P[p + ".real_conv.bias"],
                             P[p + ".imag_conv.weight"], P[p + ".imag_conv.bias"])
        if taps is not None:
            taps[f"enc{i}.conv"] = out
        out = prelu(bn(out, f"encoder.{i}.1"), P[f"encoder.{i}.2.weight"])
        if taps is not None:
            taps[f"enc{i}.out"] = out
        enc_out.append(out)

    B, C, D, T = out.shape
    out = out.permute(3, 0, 1, 2)
    if cfg.lstm == "complex":
        r = out[:, :, :C // 2].reshape(T, B, C // 2 * D)
        i_ = out[:, :, C // 2:].reshape(T, B, C // 2 * D)
        for l in range(cfg.rnn_layers):
            r, i_ = complex_lstm(r, i_, P, f"enhance.{l}", project=(l == cfg.rnn_layers - 1))
            if taps is not None:
                taps[f"lstm{l}.r"], taps[f"lstm{l}.i"] = r, i_
        r = r.reshape(T, B, C // 2, D)
        i_ = i_.reshape(T, B, C // 2, D)
        out = torch.cat([r, i_], 2)
    else:
        out = out.reshape(T, B, C * D)
        for l in range(2):
            out = lstm_layer(out, P[f"enhance.weight_ih_l{l}"], P[f"enhance.weight_hh_l{l}"],
                             P[f"enhance.bias_ih_l{l}"], P[f"enhance.bias_hh_l{l}"])
        out = F.linear(out, P["tranform.weight"], P["tranform.bias"])
        out = out.reshape(T, B, C, D)
    out = out.permute(1, 2, 3, 0)

    for d in range(nlayer):
        if cfg.skip_type:
            out = complex_cat(out, enc_out[-1 - d])
        p = f"decoder.{d}.0"
        out = complex_deconv2d(out, P[p + ".real_conv.weight"], P[p + ".real_conv.bias"],
                               P[p + ".imag_conv.weight"], P[p + ".imag_conv.bias"])
        if taps is not None:
            taps[f"dec{d}.conv"] = out              # T+1 frames: BN statistics include the frame dropped below
        if d != nlayer - 1:
            out = prelu(bn(out, f"decoder.{d}.1"), P[f"decoder.{d}.2.weight"])
        out = out[..., 1:]
        if taps is not None:
            taps[f"dec{d}.out"] = out

    m_r = F.pad(out[:, 0], [0, 0, 1, 0])
    m_i = F.pad(out[:, 1], [0, 0, 1, 0])
    if cfg.masking_mode == "Direct(None make)":
        tspec = conv_stft(targets, cfg.win_len, cfg.win_inc, cfg.fft_len, cfg.win_type)
        wav = conv_istft(torch.cat([m_r, m_i], 1), cfg.win_len, cfg.win_inc, cfg.fft_len, cfg.win_type).squeeze(1).clamp(-1, 1)
        return (m_r, tspec[:, :nfreq], m_i, tspec[:, nfreq:], wav), new_stats
    o_r, o_i = apply_mask(cfg.masking_mode, real, imag, m_r, m_i)
    wav = conv_istft(torch.cat([o_r, o_i], 1), cfg.win_len, cfg.win_inc, cfg.fft_len, cfg.win_type).squeeze(1)
    wav = torch.clamp(wav, -1, 1)
    return (o_r, o_i, wav), new_stats
