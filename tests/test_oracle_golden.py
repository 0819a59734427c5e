"""CPU tier 1: the oracle restatement against golden vectors captured from the real reference
(tests/golden/make_golden.py) and the known answers of SURVEY.md Appendix A.  Runs anywhere."""
import numpy as np
import pytest
import torch

from oracle import frontend as fe
from oracle import losses as ol
from oracle.dccrn import DCCRNConfig, dccrn_forward, dccrn_state_shapes, is_trainable
from oracle.step import dccrn_train_step
from oracle.weights import formula_state_dict, test_signals as make_signals
from util import load_golden, rel_err, rel_l2, sub, tap_stats


def oracle_params(cfg):
    shapes = dccrn_state_shapes(cfg)
    P = formula_state_dict(shapes)
    for k, shp in shapes.items():
        if k.endswith("num_batches_tracked"):
            P[k] = torch.zeros((), dtype=torch.long)
    return P


def test_frontend_kernels_and_transforms():
    g = load_golden("frontend_losses")
    K = fe.analysis_kernel().astype(np.float32)
    Kinv = fe.synthesis_kernel().astype(np.float32)
    assert rel_err(K[::37], g["stft_weight_rows"]) < 1e-6
    assert rel_err(Kinv[::37], g["istft_weight_rows"]) < 1e-5
    assert abs(np.abs(K.astype(np.float64)).sum() - g["stft_weight_asum"]) / g["stft_weight_asum"] < 1e-6
    assert rel_err(fe.periodic_hann(400), g["window"]) < 1e-6
    x, y = make_signals(2, 4000)
    S = fe.conv_stft(x)
    assert rel_err(S, g["stft_out"]) < 1e-5
    assert rel_err(fe.conv_istft(S), g["istft_consistent"]) < 1e-5
    S2 = S.clone()
    S2[:, 257] = 1.0
    S2[:, 0] *= 0.5
    S2[:, 100:140] *= 1.7
    assert rel_err(fe.conv_istft(S2), g["istft_inconsistent"]) < 1e-5
    assert rel_err(torch.sqrt(S[:, :257] ** 2 + S[:, 257:] ** 2), g["stft_mags"]) < 1e-5
    # closed form of coff: 1.5 in the kept region
    coff = fe.ola_normaliser(43)
    assert np.allclose(coff[300:-300], 1.5, atol=1e-6)


def test_loss_known_answers():
    g = load_golden("frontend_losses")
    x, y = make_signals(2, 4000)
    assert abs(float(ol.sdr(y, x)) - g["loss_sdr"]) < 1e-4
    assert abs(float(ol.si_snr(x, y)) - g["loss_si_snr"]) < 1e-4
    assert abs(float(ol.si_sdr(y, x)) - g["loss_si_sdr"]) < 1e-4
    assert abs(float(ol.main_loss("MSE", x, y)) - g["loss_mse"]) < 1e-8
    np.random.seed(0)
    ref = torch.from_numpy(np.random.randn(100))
    doc = [ol.si_sdr(ref, torch.flip(ref, [0])), ol.si_sdr(ref, ref + torch.flip(ref, [0])),
           ol.si_sdr(ref, ref + 0.5), ol.si_sdr(ref, ref * 2 + 1)]
    assert np.allclose([float(d) for d in doc], g["si_sdr_doc"], atol=1e-6)
    # values published in the reference docstring (tools_for_loss.py:57-74)
    assert np.allclose([float(d) for d in doc], [-25.127672346460717, 0.481070445785553, 6.3704606032577304,
                                                 6.3704606032577304], atol=1e-4)
    # SURVEY Q4 / Q15 known answers
    n = torch.arange(48000, dtype=torch.float64)
    xx = (0.5 * torch.sin(2 * np.pi * 440 * n / 16000) + 0.1 * torch.sin(2 * np.pi * 3000 * n / 16000 + 0.7)).float()[None]
    yy = (0.5 * torch.sin(2 * np.pi * 440 * n / 16000)).float()[None]
    q4 = [float(ol.si_snr(xx, yy)), float(ol.sdr(yy, xx)), float(ol.si_sdr(yy, xx)), float(ol.main_loss("MSE", xx, yy))]
    assert np.allclose(q4, g["q4"], rtol=1e-5)
    assert np.allclose(q4, [13.979399, 27.958801, 13.979399, 0.005], rtol=1e-5)
    Sx = fe.conv_stft(xx)
    q15 = [float(Sx[0, 0, 0]), float(Sx[0, 14, 10]), float(Sx[0, 14, 240]), float(Sx[0, 96, 240]),
           float(Sx[0, 271, 240]), float(Sx[0, 353, 240]), float(Sx[0, 256, 482]), float(Sx.abs().sum())]
    assert np.allclose(q15, g["q15"], rtol=2e-4, atol=2e-4)


def test_lms_pieces():
    g = load_golden("frontend_losses")
    for nb in (16, 32, 64):
        assert np.array_equal(ol.mel_filter_bank(nb).T, g[f"mel_{nb}"])
    x, y = make_signals(2, 4000)
    S, Sy = fe.conv_stft(x), fe.conv_stft(y)
    cm = torch.sqrt(S[:, :257] ** 2 + S[:, 257:] ** 2 + 1e-7)
    em = torch.sqrt(Sy[:, :257] ** 2 + Sy[:, 257:] ** 2 + 1e-7)
    assert abs(float(ol.lms_loss(cm, em)) - g["lms_loss"]) < 1e-5


CASES = [
    ("small_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", False),
    ("small_C_sdr", (16, 32, 32, 64, 64, 64), 128, "C", "SDR", False),
    ("small_R_mse", (16, 32, 32, 64, 64, 64), 128, "R", "MSE", False),
    ("small_E_sisdr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SDR", False),
    ("small_E_sisnr_lms", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", "LMS"),
    ("default_E_sisnr", (32, 64, 128, 256, 256, 256), 256, "E", "SI-SNR", False),
    ("wide_C_sdr", (16, 32, 32, 64, 64, 64), 512, "C", "SDR", False),
    ("real_E_sisnr", (16, 32, 32, 64, 64, 64), 256, "E", "SI-SNR", False),      # cfg.lstm == 'real'
    ("large_C_sisnr", (64, 128, 256, 512, 512, 512), 512, "C", "SI-SNR", False),   # BASELINE configs[4]: DCCRN-large
    ("noskip_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", False),       # cfg.skip_type = False
    ("cbn_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", False),          # DCCRN(use_cbn=True): ComplexBatchNorm
    ("rectwin_C_sisnr", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR", False),      # win_type=None: rectangular window
    ("hamming_C_sisnr", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR", False),      # win_type='hamming': a scipy.signal.get_window name (tools_for_model.py:19-20)
    ("hamming_E_sisnr_lms", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", "LMS"),   # ... with the LMS joint loss: the clean spectrum comes from the model's own (Hamming) ConvSTFT
]


def case_meta(g):
    """(skip_type, input scale, gradient sample stride) of a golden; older fixtures predate these fields."""
    return (bool(int(g["g/meta/skip"])) if "g/meta/skip" in g else True, float(g["g/meta/scale"]) if "g/meta/scale" in g else 1.0,
            int(g["g/meta/gstride"]) if "g/meta/gstride" in g else 53)


@pytest.mark.parametrize("name,kn,ru,mask,loss,perc", CASES)
def test_dccrn_step_against_reference(name, kn, ru, mask, loss, perc):
    g = load_golden("dccrn_" + name)
    skip, scale, gstride = case_meta(g)
    cfg = DCCRNConfig(kernel_num=kn, rnn_units=ru, masking_mode=mask, lstm="real" if name.startswith("real") else "complex", skip_type=skip,
                      use_cbn=bool(int(g["g/meta/use_cbn"])) if "g/meta/use_cbn" in g else False,
                      win_type="hamming" if name.startswith("hamming") else None if "g/meta/rect_window" in g and int(g["g/meta/rect_window"]) else "hanning")
    P = oracle_params(cfg)
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    x, y = make_signals(B, L)
    x, y = x * scale, y * scale
    r = dccrn_train_step(P, cfg, x, y, loss_kind=loss, perceptual=perc)
    o_r, o_i, wav = r["outputs"]
    assert rel_err(o_r, g["g/out_real"]) < 2e-5
    assert rel_err(o_i, g["g/out_imag"]) < 2e-5
    assert rel_err(wav, g["g/out_wav"]) < 2e-5
    assert abs(float(r["loss"]) - float(g["g/loss"])) < 2e-5 * max(1.0, abs(float(g["g/loss"])))
    gn = sub(g, "g/grad_norm")
    for k, v in gn.items():
        if k.endswith("conv.bias") and not k.startswith("decoder.5."):
            continue                # bias in front of a BatchNorm: analytically zero gradient (rounding noise)
        mine = float(r["grads"][k].double().norm())
        assert abs(mine - float(v)) <= 2e-4 * float(v) + 1e-7, (k, mine, float(v))
    for k, v in sub(g, "g/grad").items():
        scale = max(float(np.abs(v).max()), 1e-30)
        # conv biases in front of a BatchNorm have analytically zero gradient: compare on the layer's weight-grad scale
        floor = 1e-6 * float(gn[k.replace(".bias", ".weight")]) if k.endswith("conv.bias") else 0.0
        err = float(np.abs(r["grads"][k].numpy() - v).max())
        # DCCRN-large at T = 19: one BatchNorm input of encoder.2 sits on the PReLU kink (fp32 round-off decides its slope), which
        # moves ONE element of that layer's beta gradient by 1.2e-3; the tensor as a whole agrees to 2e-4 (L2)
        etol = 2e-3 if name.startswith("large") else 2e-4
        assert err <= etol * scale + floor, (k, err, scale)
        assert k.endswith("conv.bias") or rel_l2(r["grads"][k], v) < 3e-4, k
    for k, v in sub(g, "g/grad_samp").items():
        if k.endswith("conv.bias"):
            continue
        assert rel_err(r["grads"][k].reshape(-1)[::gstride], v) < 3e-4, k
    for k, v in sub(g, "g/running").items():
        assert rel_err(r["new_stats"][k], v) < 1e-5, k
    for k, v in sub(g, "g/after_adam").items():
        # Adam's first step moves every weight by ~lr*sign(g): compare the *update*, on the lr scale
        upd_ref = v - P[k].numpy()
        upd = r["new_params"][k].numpy() - P[k].numpy()
        if k.endswith("conv.bias") and not k.startswith("decoder.5."):
            continue                                    # sign of a rounding-noise gradient: not meaningful
        assert np.abs(upd - upd_ref).max() < 2e-5, k    # lr = 1e-3


def test_dccrn_layer_taps_against_reference():
    g = load_golden("dccrn_small_E_sisnr")
    cfg = DCCRNConfig(kernel_num=(16, 32, 32, 64, 64, 64), rnn_units=128, masking_mode="E")
    P = oracle_params(cfg)
    x, y = make_signals(2, 4000)
    taps = {}
    with torch.no_grad():
        dccrn_forward(P, x, cfg, targets=y, train=True, taps=taps)
    names = sorted({k.split("/")[2] for k in g if k.startswith("g/taps/")})
    assert len(names) >= 20
    for nme in names:
        s, a, samp = tap_stats(taps[nme])
        assert abs(a - float(g[f"g/taps/{nme}/asum"])) <= 1e-4 * float(g[f"g/taps/{nme}/asum"]), nme
        assert rel_err(samp, g[f"g/taps/{nme}/samp"]) < 5e-5, nme


def test_dccrn_full_length_forward():
    g = load_golden("dccrn_default_C_sisnr_full")
    cfg = DCCRNConfig(masking_mode="C")
    P = oracle_params(cfg)
    x, y = make_signals(1, 48000)
    with torch.no_grad():
        (o_r, o_i, wav), _ = dccrn_forward(P, x, cfg, targets=y, train=True)
    assert rel_err(o_r, g["g/out_real"]) < 5e-5
    assert rel_err(wav, g["g/out_wav"]) < 5e-5
    assert abs(float(-ol.si_snr(wav, y)) - float(g["g/loss"])) < 1e-4


@pytest.mark.parametrize("name,win", [("small_direct_mse", "hanning"), ("hamming_direct_mse", "hamming")])
def test_dccrn_direct_mode_against_reference(name, win):
    """Spectral mapping ('Direct(None make)', models.py:232-250) + dccrn_direct_train loss (trainer.py:135-138); the target spectra use the model's window."""
    g = load_golden("dccrn_" + name)
    cfg = DCCRNConfig(kernel_num=(16, 32, 32, 64, 64, 64), rnn_units=128, masking_mode="Direct(None make)", win_type=win)
    P = oracle_params(cfg)
    x, y = make_signals(2, 4000)
    Pg = {k: (v.clone().requires_grad_(True) if is_trainable(k) else v) for k, v in P.items()}
    (o_r, t_r, o_i, t_i, wav), _ = dccrn_forward(Pg, x, cfg, targets=y, train=True)
    lossv = (ol.main_loss("MSE", o_r, t_r) + ol.main_loss("MSE", o_i, t_i)) / 2
    assert rel_err(o_r, g["g/out_real"]) < 2e-5 and rel_err(t_i, g["g/target_imag"]) < 2e-5 and rel_err(wav, g["g/out_wav"]) < 2e-5
    assert abs(float(lossv) - float(g["g/loss"])) < 2e-5 * float(g["g/loss"])
    names = [k for k in Pg if is_trainable(k)]
    grads = dict(zip(names, torch.autograd.grad(lossv, [Pg[k] for k in names])))
    for k, v in sub(g, "g/grad_norm").items():
        if not (k.endswith("conv.bias") and not k.startswith("decoder.5.")):
            assert abs(float(grads[k].double().norm()) - float(v)) <= 3e-4 * float(v) + 1e-7, k


def test_oracle_validation_path_after_one_step():
    """Eval-mode forward (BatchNorm running statistics updated by one training-mode forward) vs the reference (trainer.py:188-241)."""
    from oracle.dccrn import DCCRNConfig, dccrn_forward, dccrn_state_shapes
    from oracle.losses import main_loss
    from oracle.step import dccrn_train_step
    from oracle.weights import formula_state_dict, test_signals
    g = load_golden("dccrn_small_eval")
    cfg = DCCRNConfig(kernel_num=tuple(int(k) for k in g["g/meta/kernel_num"]), rnn_units=int(g["g/meta/rnn_units"]), masking_mode="C")
    P = formula_state_dict(dccrn_state_shapes(cfg))
    x, y = test_signals(int(g["g/meta/B"]), int(g["g/meta/L"]))
    with torch.no_grad():
        (_, _, wav0), new_stats = dccrn_forward(P, x, cfg, targets=y, train=True)      # updates the running statistics only
    assert abs(float(main_loss("SI-SNR", wav0, y)) - float(g["g/train_loss"])) < 1e-4 * abs(float(g["g/train_loss"]))
    P2 = dict(P)
    P2.update(new_stats)
    xv, yv = test_signals(int(g["g/meta/Bv"]), int(g["g/meta/Lv"]))
    xv, yv = xv.flip(0) * 0.8, yv.flip(0) * 0.8
    with torch.no_grad():
        (o_r, o_i, wav), _ = dccrn_forward(P2, xv, cfg, targets=yv, train=False)
        vloss = main_loss("SI-SNR", wav, yv)
    assert rel_err(wav, g["g/val_wav"]) < 1e-4
    assert abs(float(vloss) - float(g["g/val_loss"])) < 1e-4 * abs(float(g["g/val_loss"]))


def test_vectorised_lstm_equals_the_written_out_cell():
    """oracle.dccrn.lstm_layer (ATen's fused CPU LSTM, one call for all T steps) == the step-by-step statement of the cell."""
    from oracle.dccrn import lstm_layer, lstm_layer_loop
    torch.manual_seed(0)
    x = torch.randn(50, 3, 40, requires_grad=True)
    wi, wh = (torch.randn(64, 40) * 0.1).requires_grad_(), (torch.randn(64, 16) * 0.1).requires_grad_()
    bi, bh = torch.randn(64) * 0.1, torch.randn(64) * 0.1
    a, b = lstm_layer(x, wi, wh, bi, bh), lstm_layer_loop(x, wi, wh, bi, bh)
    assert rel_err(a, b) < 1e-6
    for p, q in zip(torch.autograd.grad(a.square().sum(), [x, wi, wh]), torch.autograd.grad(b.square().sum(), [x, wi, wh])):
        assert rel_err(p, q) < 1e-5
