"""Golden-vector generator.  RUN ONLY IN THE BUILD CONTAINER (needs /root/reference).

Imports the real reference (seorim0/DNN-based-Speech-Enhancement-in-the-frequency-domain) through the
import shim of SURVEY.md Appendix C, fills its modules with the formula weights of
oracle/weights.py, runs forward / loss / backward / Adam on closed-form signals and writes small
.npz fixtures next to this file.  Only *data* (inputs are closed-form, outputs are stored) is committed;
no reference source travels.

    python tests/golden/make_golden.py
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from oracle.weights import fill_state_dict_, test_signals  # noqa: E402


def import_reference(perceptual=False):
    class _Stub(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
    al = types.ModuleType("asteroid.losses")
    al.SingleSrcPMSQE = al.PITLossWrapper = _Stub
    a = types.ModuleType("asteroid")
    a.losses = al
    af = types.ModuleType("asteroid_filterbanks")
    af.STFTFB = af.Encoder = _Stub
    af.transforms = types.SimpleNamespace(mag=None)
    sys.modules.update({"asteroid": a, "asteroid.losses": al, "asteroid_filterbanks": af})
    with contextlib.redirect_stdout(io.StringIO()):
        import config as cfg
    cfg.DEVICE = "cpu"
    cfg.window = "hann"
    cfg.perceptual = "LMS"          # so that MEL_SCALES is defined at import of tools_for_loss
    import models
    import tools_for_loss
    import tools_for_model
    cfg.perceptual = False
    return cfg, models, tools_for_model, tools_for_loss


def sample(t: torch.Tensor, stride=97):
    f = t.detach().reshape(-1).double()
    return dict(sum=float(f.sum()), asum=float(f.abs().sum()), n=f.numel(),
                samp=t.detach().reshape(-1)[::stride].float().numpy().copy())


def flat(d, prefix):
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            out.update(flat(v, f"{prefix}/{k}"))
        else:
            out[f"{prefix}/{k}"] = np.asarray(v)
    return out


SMALL_PARAMS = lambda name: (name.startswith("encoder.0.0.") or name.endswith(".2.weight") or ".1.weight" in name
                             or ".1.bias" in name or ".1.W" in name or ".1.B" in name or name.endswith("r_trans.bias") or name.endswith("i_trans.bias")
                             or name.startswith("decoder.5.0.") or name.endswith("bias_hh_l0"))


def dccrn_case(cfg, models, name, kernel_num, rnn_units, mask, loss, perceptual, B, L, store_taps=True, lstm="complex", skip=True, gstride=53, scale=1.0,
               use_cbn=False, win_type="hann"):
    cfg.dccrn_kernel_num = list(kernel_num)
    cfg.masking_mode = mask
    cfg.loss = loss
    cfg.perceptual = perceptual
    cfg.lstm = lstm
    cfg.skip_type = skip
    torch.manual_seed(0)
    m = models.DCCRN(rnn_units=rnn_units, masking_mode=mask, use_cbn=use_cbn, win_type=win_type)
    cfg.skip_type = True            # read at construction AND in forward (models.py:107, 222): restored after the forward below
    fill_state_dict_(m)
    m.train()
    x, y = test_signals(B, L)
    x, y = x * scale, y * scale
    taps = {}
    cfg.skip_type = skip
    hooks = []
    if store_taps:
        def mk(key):
            return lambda mod, inp, out: taps.__setitem__(key, sample(out if torch.is_tensor(out) else out[0]))
        hooks.append(m.stft.register_forward_hook(mk("spec")))
        for i, layer in enumerate(m.encoder):
            hooks.append(layer[0].register_forward_hook(mk(f"enc{i}.conv")))
            hooks.append(layer.register_forward_hook(mk(f"enc{i}.out")))
        for i, layer in enumerate(m.decoder):
            hooks.append(layer[0].register_forward_hook(mk(f"dec{i}.conv")))
        for l, layer in enumerate(m.enhance):
            hooks.append(layer.register_forward_hook(
                lambda mod, inp, out, l=l: (taps.__setitem__(f"lstm{l}.r", sample(out[0])),
                                            taps.__setitem__(f"lstm{l}.i", sample(out[1])))[0]))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    if perceptual:
        o_r, o_i, wav = m(x)
        main = m.loss(wav, y)
        perc = m.loss(wav, y, o_r, o_i, perceptual=True)
        lossv = (main + perc) / 2
    else:
        o_r, o_i, wav = m(x, y)
        lossv = m.loss(wav, y)
        main = lossv
        perc = torch.zeros(())
    for h in hooks:
        h.remove()
    cfg.skip_type = True
    opt.zero_grad()
    lossv.backward()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    sd = m.state_dict()
    rec = dict(
        meta=dict(B=B, L=L, kernel_num=np.array(kernel_num), rnn_units=rnn_units, skip=int(skip), gstride=gstride, scale=float(scale), use_cbn=int(use_cbn), rect_window=int(win_type is None),
                  mask=np.array(mask), loss=np.array(loss), perceptual=np.array(str(perceptual))),
        out_real=o_r.detach().numpy(), out_imag=o_i.detach().numpy(), out_wav=wav.detach().numpy(),
        loss=float(lossv), main_loss=float(main), perc_loss=float(perc),
        taps=taps,
        grad_norm={k: float(v.double().norm()) for k, v in g.items()},
        grad={k: v.numpy() for k, v in g.items() if SMALL_PARAMS(k)},
        grad_samp={k: sample(v, gstride)["samp"] for k, v in g.items() if not SMALL_PARAMS(k)},
        after_adam={k: sd[k].numpy().copy() for k in g if SMALL_PARAMS(k)},
        running={k: v.numpy().copy() for k, v in sd.items() if "running_" in k or ".1.RM" in k or ".1.RV" in k},
    )
    np.savez_compressed(os.path.join(HERE, f"dccrn_{name}.npz"), **flat(rec, "g"))
    print(f"dccrn_{name}: loss {float(lossv):.6f} |wav|max {float(wav.abs().max()):.4f}")


def dccrn_direct_case(cfg, models, name, kernel_num, rnn_units, loss, B, L, win_type="hann"):
    """dccrn_direct_train (trainer.py:121-150): spectral mapping, loss = (loss(real) + loss(imag)) / 2."""
    cfg.dccrn_kernel_num = list(kernel_num)
    cfg.masking_mode = "Direct(None make)"
    cfg.loss = loss
    cfg.perceptual = False
    cfg.lstm = "complex"
    cfg.skip_type = True
    m = models.DCCRN(rnn_units=rnn_units, masking_mode="Direct(None make)", win_type=win_type)
    fill_state_dict_(m)
    m.train()
    x, y = test_signals(B, L)
    o_r, t_r, o_i, t_i, wav = m(x, y)
    lossv = (m.loss(o_r, t_r) + m.loss(o_i, t_i)) / 2
    lossv.backward()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    rec = dict(meta=dict(B=B, L=L), out_real=o_r.detach().numpy(), target_real=t_r.detach().numpy(), out_imag=o_i.detach().numpy(),
               target_imag=t_i.detach().numpy(), out_wav=wav.detach().numpy(), loss=float(lossv),
               grad_norm={k: float(v.double().norm()) for k, v in g.items()},
               grad={k: v.numpy() for k, v in g.items() if SMALL_PARAMS(k)})
    np.savez_compressed(os.path.join(HERE, f"dccrn_{name}.npz"), **flat(rec, "g"))
    print(f"dccrn_{name}: loss {float(lossv):.6f}")


def crn_case(cfg, models, name, kernel_num, rnn_units, rnn_input, mask, loss, B, L):
    cfg.dccrn_kernel_num = list(kernel_num)
    cfg.masking_mode = mask
    cfg.loss = loss
    cfg.perceptual = False
    cfg.skip_type = True
    torch.manual_seed(0)
    m = models.CRN(rnn_units=rnn_units, rnn_input_size=rnn_input, masking_mode=mask)
    fill_state_dict_(m)
    m.train()
    x, y = test_signals(B, L)
    taps = {}
    hooks = [layer[0].register_forward_hook(lambda mod, i, o, k=f"enc{j}.conv": taps.__setitem__(k, sample(o))) for j, layer in enumerate(m.encoder)]
    hooks += [layer[0].register_forward_hook(lambda mod, i, o, k=f"dec{j}.conv": taps.__setitem__(k, sample(o))) for j, layer in enumerate(m.decoder)]
    hooks.append(m.tranform.register_forward_hook(lambda mod, i, o: taps.__setitem__("lstm", sample(o))))
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    est_mags, target_mags, wav = m(x, y)
    for h in hooks:
        h.remove()
    # crn_direct_train (trainer.py:169-170) puts the loss on the magnitudes; the masking trainer on the waveform
    lossv = m.loss(est_mags, target_mags) if mask == "Direct(None make)" else m.loss(wav, y)
    opt.zero_grad()
    lossv.backward()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    sd = m.state_dict()
    small = lambda k: (k.startswith("encoder.0.0.") or k.endswith(".2.weight") or ".1.weight" in k or ".1.bias" in k
                       or k.startswith("decoder.5.0.") or k.startswith("tranform.bias") or k.endswith("bias_hh_l0"))
    rec = dict(
        meta=dict(B=B, L=L, kernel_num=np.array(kernel_num), rnn_units=rnn_units, rnn_input=rnn_input, mask=np.array(mask), loss=np.array(loss)),
        est_mags=est_mags.detach().numpy(), target_mags=target_mags.detach().numpy(), out_wav=wav.detach().numpy(), loss=float(lossv),
        taps=taps, grad_norm={k: float(v.double().norm()) for k, v in g.items()},
        grad={k: v.numpy() for k, v in g.items() if small(k)},
        grad_samp={k: sample(v, 53)["samp"] for k, v in g.items() if not small(k)},
        after_adam={k: sd[k].numpy().copy() for k in g if small(k)},
        running={k: v.numpy().copy() for k, v in sd.items() if "running_" in k})
    np.savez_compressed(os.path.join(HERE, f"crn_{name}.npz"), **flat(rec, "g"))
    print(f"crn_{name}: loss {float(lossv):.6f} |wav|max {float(wav.abs().max()):.4f}")


def fsn_case(cfg, models, tfm, name, B, L, hidden=(512, 384), sequence_model="LSTM", norm_type="offline_laplace_norm", loss="MSE"):
    """FullSubNet train step with the LSTM inter-layer dropout patched to 0 (p = 0.8 makes train mode stochastic).
    loss != 'MSE': FullSubNet.loss (models.py:674-682) reduces over the LAST axis of [B, F, T, 2] - two-element rows -, and the trainer
    passes the network output in the `target` slot (trainer.py:107): the gradient flows through s1 of sdr, s2 of si_snr, `reference` of si_sdr."""
    cfg.loss = loss
    torch.manual_seed(0)
    m = models.FullSubNet(fb_model_hidden_size=hidden[0], sb_model_hidden_size=hidden[1], sequence_model=sequence_model, norm_type=norm_type)
    fill_state_dict_(m)
    m.train()
    m.fb_model.sequence_model.dropout = 0.0
    m.sb_model.sequence_model.dropout = 0.0
    x, y = test_signals(B, L)
    nc, cc = tfm.stft(x), tfm.stft(y)
    noisy_mag, _ = tfm.mag_phase(nc)
    cirm = tfm.build_complex_ideal_ratio_mask(nc, cc)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crm = m(noisy_mag)
    lossv = m.loss(cirm, crm)
    opt.zero_grad()
    lossv.backward()
    g = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
    opt.step()
    sd = m.state_dict()
    small = lambda k: "bias" in k or k.startswith("sb_model.fc_output_layer")
    cfg.loss = "MSE"
    rec = dict(meta=dict(B=B, L=L, fb_hidden=hidden[0], sb_hidden=hidden[1], sequence_model=sequence_model, norm_type=norm_type, loss=np.array(loss)),
               noisy_mag=noisy_mag.numpy()[:, ::4, ::3], cirm=cirm.numpy()[:, ::4, ::3], crm=crm.detach().numpy(), loss=float(lossv),
               grad_norm={k: float(v.double().norm()) for k, v in g.items()},
               grad={k: v.numpy() for k, v in g.items() if small(k)},
               grad_samp={k: sample(v, 211)["samp"] for k, v in g.items() if not small(k)},
               after_adam={k: sd[k].numpy().copy() for k in g if small(k)})
    if loss != "MSE":
        # the per-row losses over two-element rows amplify what the cIRM does at near-silent bins (noise / noise through the compression,
        # anything in +-10): the whole target is stored, so that network + loss parity is checked on the SAME target (the cIRM front end has
        # its own test with its own, well-conditioned, tolerance)
        rec["cirm_full"] = cirm.numpy()
    np.savez_compressed(os.path.join(HERE, f"fsn_{name}.npz"), **flat(rec, "g"))
    print(f"fsn_{name}: loss {float(lossv):.6f}")


def fsn_losses(cfg, models, tfm):
    fsn_case(cfg, models, tfm, "small_sdr", 2, 6000, hidden=(128, 64), loss="SDR")        # config.py:36: the reference's DEFAULT cfg.loss
    fsn_case(cfg, models, tfm, "small_sisnr", 2, 6000, hidden=(128, 64), loss="SI-SNR")
    fsn_case(cfg, models, tfm, "small_sisdr", 2, 6000, hidden=(128, 64), loss="SI-SDR")


def fsn_weight_init(cfg, models):
    """FullSubNet(weight_init=True) (models.py:623-624, tools_for_model.py:1120-1184): a digest of every parameter after construction under seed 0."""
    rec = {}
    for seq in ("LSTM", "GRU"):
        torch.manual_seed(0)
        m = models.FullSubNet(fb_model_hidden_size=64, sb_model_hidden_size=32, sequence_model=seq, weight_init=True)
        for k, p in m.named_parameters():
            v = p.detach().double().reshape(-1)
            rec[f"{seq}/{k}"] = np.array([float(v.sum()), float(v.abs().sum())] + [float(t) for t in v[:6]])
    np.savez_compressed(os.path.join(HERE, "fsn_weight_init.npz"), **flat(rec, "g"))
    print("fsn_weight_init:", len(rec), "tensors")


def fsn_variants(cfg, models, tfm):
    fsn_case(cfg, models, tfm, "small_gru_mse", 2, 6000, hidden=(128, 64), sequence_model="GRU")
    fsn_case(cfg, models, tfm, "small_cumlaplace_mse", 2, 6000, hidden=(128, 64), norm_type="cumulative_laplace_norm")
    fsn_case(cfg, models, tfm, "small_gaussian_mse", 2, 6000, hidden=(128, 64), norm_type="offline_gaussian_norm")
    fsn_case(cfg, models, tfm, "small_cumlayer_gru_mse", 2, 6000, hidden=(128, 64), sequence_model="GRU", norm_type="cumulative_layer_norm")


def frontend_and_losses(cfg, models, tfm, tfl):
    out = {}
    K, _ = tfm.init_kernels(400, 100, 512, "hann")
    Kinv, win = tfm.init_kernels(400, 100, 512, "hann", invers=True)
    out["stft_weight_rows"] = K[::37, 0].numpy()
    out["stft_weight_asum"] = float(K.double().abs().sum())
    out["istft_weight_rows"] = Kinv[::37, 0].numpy()
    out["istft_weight_asum"] = float(Kinv.double().abs().sum())
    out["window"] = win[0, :, 0].numpy()
    x, y = test_signals(2, 4000)
    stft = tfm.ConvSTFT(400, 100, 512, "hann", "complex")
    istft = tfm.ConviSTFT(400, 100, 512, "hann", "complex")
    S = stft(x)
    out["stft_out"] = S.numpy()
    out["istft_consistent"] = istft(S).numpy()
    S2 = S.clone()
    S2[:, 257] = 1.0
    S2[:, 0] *= 0.5
    S2[:, 100:140] *= 1.7
    out["istft_inconsistent"] = istft(S2).numpy()
    mags, phase = tfm.ConvSTFT(400, 100, 512, "hann", "real")(x)
    out["stft_mags"] = mags.numpy()
    # losses on fixed pairs (est = noisy, target = clean)
    out["loss_sdr"] = float(tfl.sdr(y, x))
    out["loss_si_snr"] = float(tfl.si_snr(x, y))
    out["loss_si_sdr"] = float(tfl.si_sdr(y, x))
    out["loss_mse"] = float(torch.nn.functional.mse_loss(x, y))
    # si_sdr docstring known answers (tools_for_loss.py:57-74)
    np.random.seed(0)
    ref = torch.from_numpy(np.random.randn(100))
    out["si_sdr_doc"] = np.array([float(tfl.si_sdr(ref, torch.flip(ref, [0]))),
                                  float(tfl.si_sdr(ref, ref + torch.flip(ref, [0]))),
                                  float(tfl.si_sdr(ref, ref + 0.5)),
                                  float(tfl.si_sdr(ref, ref * 2 + 1))])
    # LMS pieces
    for nb in (16, 32, 64):
        out[f"mel_{nb}"] = tfl.melFilterBank(nb, 512).astype(np.float32)
    cm = torch.sqrt(S[:, :257] ** 2 + S[:, 257:] ** 2 + 1e-7)
    Sy = stft(y)
    em = torch.sqrt(Sy[:, :257] ** 2 + Sy[:, 257:] ** 2 + 1e-7)
    out["lms_loss"] = float(tfl.get_array_lms_loss(cm, em))
    # FullSubNet front end known answers
    n = torch.arange(48000, dtype=torch.float64)
    xx = (0.5 * torch.sin(2 * np.pi * 440 * n / 16000) + 0.1 * torch.sin(2 * np.pi * 3000 * n / 16000 + 0.7)).float()[None]
    yy = (0.5 * torch.sin(2 * np.pi * 440 * n / 16000)).float()[None]
    out["q4"] = np.array([float(tfl.si_snr(xx, yy)), float(tfl.sdr(yy, xx)), float(tfl.si_sdr(yy, xx)),
                          float(torch.nn.functional.mse_loss(xx, yy))])
    Sx = stft(xx)
    out["q15"] = np.array([float(Sx[0, 0, 0]), float(Sx[0, 14, 10]), float(Sx[0, 14, 240]), float(Sx[0, 96, 240]),
                           float(Sx[0, 271, 240]), float(Sx[0, 353, 240]), float(Sx[0, 256, 482]), float(Sx.abs().sum())])
    cx = tfm.stft(xx)
    cy = tfm.stft(yy)
    out["fsn_stft_samp"] = torch.view_as_real(cx)[0, ::8, ::10].numpy()
    out["fsn_cirm_samp"] = tfm.build_complex_ideal_ratio_mask(cx, cy)[0, ::8, ::10].numpy()
    np.savez_compressed(os.path.join(HERE, "frontend_losses.npz"), **out)
    print("frontend_losses: q4", out["q4"], "lms", out["lms_loss"])


def dccrn_eval_case(cfg, models, name, kernel_num, rnn_units, mask, loss, B, L, Bv, Lv, use_cbn=False):
    """Validation path (trainer.py:188-241 `model_validate` minus the PESQ/STOI scorers): one training-mode forward, then
    `model.eval()` + `torch.no_grad()` forward and loss on a different batch - BatchNorm uses the UPDATED running
    statistics, so this pins the eval plans and the running-stat update together."""
    cfg.dccrn_kernel_num = list(kernel_num)
    cfg.masking_mode = mask
    cfg.loss = loss
    cfg.perceptual = False
    cfg.lstm = "complex"
    cfg.skip_type = True
    torch.manual_seed(0)
    m = models.DCCRN(rnn_units=rnn_units, masking_mode=mask, use_cbn=use_cbn)
    fill_state_dict_(m)
    m.train()
    x, y = test_signals(B, L)
    # a training-mode forward updates the running statistics; no optimizer step here: Adam turns the rounding-noise
    # gradients of the conv biases in front of BatchNorm into +-lr steps of random sign, which eval mode (running mean
    # taken with the OLD bias) does not cancel - the reference itself is only reproducible to ~1e-3 after such a step
    with torch.no_grad():
        _, _, wav = m(x, y)
        lossv = m.loss(wav, y)
    m.eval()
    xv, yv = test_signals(Bv, Lv)
    xv, yv = xv.flip(0) * 0.8, yv.flip(0) * 0.8             # not the training batch
    with torch.no_grad():
        o_r, o_i, wv = m(xv, yv)
        vloss = m.loss(wv, yv)
    rec = dict(meta=dict(B=B, L=L, Bv=Bv, Lv=Lv, kernel_num=np.array(kernel_num), rnn_units=rnn_units, mask=np.array(mask), loss=np.array(loss), use_cbn=int(use_cbn)),
               train_loss=float(lossv), val_loss=float(vloss), val_wav=wv.numpy(), val_real=sample(o_r), val_imag=sample(o_i))
    np.savez_compressed(os.path.join(HERE, f"dccrn_{name}.npz"), **flat(rec, "g"))
    print(f"dccrn_{name}: train loss {float(lossv):.6f} val loss {float(vloss):.6f}")


def main():
    cfg, models, tfm, tfl = import_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "wide":          # only the wide-LSTM case (rnn_units 512, DCCRN-large's LSTM width)
        dccrn_case(cfg, models, "wide_C_sdr", (16, 32, 32, 64, 64, 64), 512, "C", "SDR", False, 1, 2000, store_taps=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "crn_direct":    # only CRN spectral mapping
        crn_case(cfg, models, "small_direct_mse", (16, 32, 32, 64, 64, 64), 128, 128, "Direct(None make)", "MSE", 2, 4000)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "real":          # only the cfg.lstm == 'real' case
        dccrn_case(cfg, models, "real_E_sisnr", (16, 32, 32, 64, 64, 64), 256, "E", "SI-SNR", False, 2, 3000, store_taps=False, lstm="real")
        cfg.lstm = "complex"
        return
    if len(sys.argv) > 1 and sys.argv[1] == "large":         # BASELINE configs[4]: DCCRN-large (2x channels, rnn_units 512), short clip
        dccrn_case(cfg, models, "large_C_sisnr", (64, 128, 256, 512, 512, 512), 512, "C", "SI-SNR", False, 2, 1600, store_taps=False, gstride=997, scale=0.125)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "hamming":       # ConvSTFT(win_type='hamming'): any scipy.signal.get_window name (tools_for_model.py:19-20)
        dccrn_case(cfg, models, "hamming_C_sisnr", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR", False, 2, 4000, store_taps=False, win_type="hamming")
        # ... and the two paths where the TARGET spectrum comes from the model's own ConvSTFT (models.py:237, 306-308): spectral mapping and the LMS joint loss
        dccrn_direct_case(cfg, models, "hamming_direct_mse", (16, 32, 32, 64, 64, 64), 128, "MSE", 2, 4000, win_type="hamming")
        dccrn_case(cfg, models, "hamming_E_sisnr_lms", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", "LMS", 2, 4000, store_taps=False, win_type="hamming")
        cfg.perceptual = False
        return
    if len(sys.argv) > 1 and sys.argv[1] == "rectwin":       # ConvSTFT(win_type=None): rectangular window (tools_for_model.py:17-18)
        dccrn_case(cfg, models, "rectwin_C_sisnr", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR", False, 2, 4000, store_taps=False, win_type=None)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cbn":           # DCCRN(use_cbn=True): ComplexBatchNorm (tools_for_model.py:430-607), train step + eval forward
        dccrn_case(cfg, models, "cbn_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", False, 2, 4000, use_cbn=True)
        dccrn_eval_case(cfg, models, "cbn_eval", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", 2, 4000, 3, 2400, use_cbn=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "noskip":        # cfg.skip_type = False (models.py:107-137, 222-223)
        dccrn_case(cfg, models, "noskip_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", False, 2, 3000, store_taps=False, skip=False)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fsn_variants":  # cfg.sequence_model == 'GRU' and the three other norm_type choices
        fsn_variants(cfg, models, tfm)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fsn_weight_init":
        fsn_weight_init(cfg, models)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "fsn_losses":    # FullSubNet.loss with SDR / SI-SNR / SI-SDR
        fsn_losses(cfg, models, tfm)
        return
    if len(sys.argv) > 1 and sys.argv[1] == "eval":          # regenerate only the validation-path case
        dccrn_eval_case(cfg, models, "small_eval", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR", 2, 4000, 3, 5000)
        return
    frontend_and_losses(cfg, models, tfm, tfl)
    small = (16, 32, 32, 64, 64, 64)
    dflt = (32, 64, 128, 256, 256, 256)
    dccrn_case(cfg, models, "small_E_sisnr", small, 128, "E", "SI-SNR", False, 2, 4000)
    dccrn_case(cfg, models, "small_C_sdr", small, 128, "C", "SDR", False, 2, 4000, store_taps=False)
    dccrn_case(cfg, models, "small_R_mse", small, 128, "R", "MSE", False, 2, 4000, store_taps=False)
    dccrn_case(cfg, models, "small_E_sisdr", small, 128, "E", "SI-SDR", False, 2, 4000, store_taps=False)
    dccrn_case(cfg, models, "small_E_sisnr_lms", small, 128, "E", "SI-SNR", "LMS", 2, 4000, store_taps=False)
    dccrn_case(cfg, models, "default_E_sisnr", dflt, 256, "E", "SI-SNR", False, 2, 4000)
    dccrn_case(cfg, models, "default_C_sisnr_full", dflt, 256, "C", "SI-SNR", False, 1, 48000, store_taps=False)
    dccrn_direct_case(cfg, models, "small_direct_mse", small, 128, "MSE", 2, 4000)
    crn_case(cfg, models, "default_E_mse", dflt, 256, 512, "E", "MSE", 2, 4000)
    crn_case(cfg, models, "small_E_sisnr", small, 128, 128, "E", "SI-SNR", 2, 4000)
    crn_case(cfg, models, "small_direct_mse", small, 128, 128, "Direct(None make)", "MSE", 2, 4000)
    fsn_case(cfg, models, tfm, "default_mse", 2, 6000)
    fsn_case(cfg, models, tfm, "small_mse", 2, 6000, hidden=(128, 64))
    fsn_variants(cfg, models, tfm)
    fsn_losses(cfg, models, tfm)
    dccrn_eval_case(cfg, models, "small_eval", small, 128, "C", "SI-SNR", 2, 4000, 3, 5000)
    dccrn_case(cfg, models, "wide_C_sdr", small, 512, "C", "SDR", False, 1, 2000, store_taps=False)
    dccrn_case(cfg, models, "real_E_sisnr", small, 256, "E", "SI-SNR", False, 2, 3000, store_taps=False, lstm="real")
    cfg.lstm = "complex"
    dccrn_case(cfg, models, "large_C_sisnr", (64, 128, 256, 512, 512, 512), 512, "C", "SI-SNR", False, 2, 1600, store_taps=False, gstride=997, scale=0.125)
    dccrn_case(cfg, models, "noskip_E_sisnr", small, 128, "E", "SI-SNR", False, 2, 3000, store_taps=False, skip=False)


if __name__ == "__main__":
    main()
