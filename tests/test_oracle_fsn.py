"""CPU tier 1 (FullSubNet): oracle restatement (models.py:568-682, trainer.py:85-118) against goldens captured from the real
reference with the inter-layer LSTM dropout patched to 0, plus the Q7 known answers of the hop-300 torch.stft."""
import numpy as np
import pytest
import torch

from oracle.fullsubnet import FSNConfig, fsn_forward, fsn_state_shapes, fsn_targets
from oracle.losses import main_loss
from oracle.step import adam_update
from oracle.weights import formula_state_dict, test_signals as make_signals
from util import load_golden, rel_err, sub


@pytest.mark.parametrize("name,hid,seq,norm", [("default_mse", (512, 384), "LSTM", "offline_laplace_norm"),
                                               ("small_mse", (128, 64), "LSTM", "offline_laplace_norm"),
                                               ("small_gru_mse", (128, 64), "GRU", "offline_laplace_norm"),
                                               ("small_cumlaplace_mse", (128, 64), "LSTM", "cumulative_laplace_norm"),
                                               ("small_gaussian_mse", (128, 64), "LSTM", "offline_gaussian_norm"),
                                               ("small_cumlayer_gru_mse", (128, 64), "GRU", "cumulative_layer_norm"),
                                               # FullSubNet.loss (models.py:674-682) with cfg.loss = SDR (config.py:36 default) / SI-SNR / SI-SDR
                                               ("small_sdr", (128, 64), "LSTM", "offline_laplace_norm"),
                                               ("small_sisnr", (128, 64), "LSTM", "offline_laplace_norm"),
                                               ("small_sisdr", (128, 64), "LSTM", "offline_laplace_norm")])
def test_fsn_step_against_reference(name, hid, seq, norm):
    g = load_golden("fsn_" + name)
    kind = str(g["g/meta/loss"]) if "g/meta/loss" in g else "MSE"
    cfg = FSNConfig(fb_hidden=hid[0], sb_hidden=hid[1], sequence_model=seq, norm_type=norm)
    P = formula_state_dict(fsn_state_shapes(cfg))
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    x, y = make_signals(B, L)
    noisy_mag, cirm = fsn_targets(x, y, cfg)
    assert rel_err(noisy_mag[:, ::4, ::3], g["g/noisy_mag"]) < 1e-5
    assert rel_err(cirm[:, ::4, ::3], g["g/cirm"]) < 1e-5
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    crm = fsn_forward(Pg, noisy_mag, cfg)
    lossv = main_loss(kind, cirm, crm)          # model.loss(cIRM, cRM): the network output sits in the `target` slot (trainer.py:107)
    names = list(Pg)
    grads = dict(zip(names, torch.autograd.grad(lossv, [Pg[k] for k in names])))
    assert rel_err(crm, g["g/crm"]) < 2e-5
    assert abs(float(lossv) - float(g["g/loss"])) < 2e-5 * max(1.0, abs(float(g["g/loss"])))
    for k, v in sub(g, "g/grad_norm").items():
        assert abs(float(grads[k].double().norm()) - float(v)) <= 3e-4 * float(v) + 1e-9, k
    for k, v in sub(g, "g/grad").items():
        assert rel_err(grads[k], v) < 3e-4, k
    for k, v in sub(g, "g/after_adam").items():
        newp, _, _ = adam_update(P[k], grads[k], torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1)
        mask = np.abs(v - P[k].numpy()) > 0       # every bias moves by ~lr on the first step
        assert np.abs((newp.numpy() - P[k].numpy()) - (v - P[k].numpy()))[mask].max() < 5e-5, k


def test_fsn_frontend_known_answers():
    """torch.stft hop 300 / centre / reflect; cIRM(x, y)[0, 14, 80] = [0.49958369, 0.0]."""
    g = load_golden("frontend_losses")
    n = torch.arange(48000, dtype=torch.float64)
    xx = (0.5 * torch.sin(2 * np.pi * 440 * n / 16000) + 0.1 * torch.sin(2 * np.pi * 3000 * n / 16000 + 0.7)).float()[None]
    yy = (0.5 * torch.sin(2 * np.pi * 440 * n / 16000)).float()[None]
    cfg = FSNConfig()
    from oracle.frontend import torch_stft
    from oracle.fullsubnet import build_cirm
    cx, cy = torch_stft(xx), torch_stft(yy)
    assert cx.shape == (1, 257, 161)
    assert rel_err(torch.view_as_real(cx)[0, ::8, ::10], g["fsn_stft_samp"]) < 1e-5
    cirm = build_cirm(cx, cy)
    assert rel_err(cirm[0, ::8, ::10], g["fsn_cirm_samp"]) < 1e-5
    assert abs(float(cirm[0, 14, 80, 0]) - 0.49958369) < 1e-4 and abs(float(cirm[0, 14, 80, 1])) < 1e-4
