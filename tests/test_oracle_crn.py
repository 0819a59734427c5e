"""CPU tier 1 (CRN): oracle restatement of models.py:329-565 against goldens captured from the real reference."""
import numpy as np
import pytest
import torch

from oracle.crn import CRNConfig, crn_forward, crn_state_shapes
from oracle.dccrn import is_trainable
from oracle.losses import main_loss
from oracle.step import adam_update
from oracle.weights import formula_state_dict, test_signals as make_signals
from util import load_golden, rel_err, sub, tap_stats

CASES = [("default_E_mse", (32, 64, 128, 256, 256, 256), 256, 512, "E", "MSE"),
         ("small_E_sisnr", (16, 32, 32, 64, 64, 64), 128, 128, "E", "SI-SNR")]


@pytest.mark.parametrize("name,kn,ru,ri,mask,loss", CASES)
def test_crn_step_against_reference(name, kn, ru, ri, mask, loss):
    g = load_golden("crn_" + name)
    cfg = CRNConfig(kernel_num=kn, rnn_units=ru, rnn_input_size=ri, masking_mode=mask)
    P = formula_state_dict(crn_state_shapes(cfg))
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    x, y = make_signals(B, L)
    Pg = {k: (v.clone().requires_grad_(True) if is_trainable(k) else v) for k, v in P.items()}
    taps = {}
    (est, tm, wav), stats = crn_forward(Pg, x, y, cfg, train=True, taps=taps)
    lossv = main_loss(loss, wav, y)
    names = [k for k in Pg if is_trainable(k)]
    grads = dict(zip(names, torch.autograd.grad(lossv, [Pg[k] for k in names])))
    assert rel_err(est, g["g/est_mags"]) < 2e-5
    assert rel_err(tm, g["g/target_mags"]) < 2e-5
    assert rel_err(wav, g["g/out_wav"]) < 2e-5
    assert abs(float(lossv) - float(g["g/loss"])) < 2e-5 * max(1.0, abs(float(g["g/loss"])))
    for nme in sorted({k.split("/")[2] for k in g if k.startswith("g/taps/")}):
        s, a, samp = tap_stats(taps[nme])
        assert rel_err(samp, g[f"g/taps/{nme}/samp"]) < 5e-5, nme
    noise = lambda k: k.endswith("conv.bias") and not k.startswith("decoder.5.")
    for k, v in sub(g, "g/grad_norm").items():
        if not noise(k):
            assert abs(float(grads[k].double().norm()) - float(v)) <= 3e-4 * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if not noise(k):
            assert rel_err(grads[k], v) < 3e-4, k
    for k, v in sub(g, "g/running").items():
        assert rel_err(stats[k], v) < 1e-5, k
    for k, v in sub(g, "g/after_adam").items():
        if noise(k):
            continue
        newp, _, _ = adam_update(P[k], grads[k], torch.zeros_like(P[k]), torch.zeros_like(P[k]), 1)
        assert np.abs((newp.numpy() - P[k].numpy()) - (v - P[k].numpy())).max() < 2e-5, k


def test_crn_direct_mode_against_reference():
    """'Direct(None make)' + crn_direct_train's loss (models.py:506-517, trainer.py:169-170)."""
    g = load_golden("crn_small_direct_mse")
    kn = (16, 32, 32, 64, 64, 64)
    cfg = CRNConfig(kernel_num=kn, rnn_units=128, rnn_input_size=128, masking_mode="Direct(None make)")
    P = formula_state_dict(crn_state_shapes(cfg))
    x, y = make_signals(int(g["g/meta/B"]), int(g["g/meta/L"]))
    Pg = {k: (v.clone().requires_grad_(True) if is_trainable(k) else v) for k, v in P.items()}
    (est, tm, wav), stats = crn_forward(Pg, x, y, cfg, train=True)
    lossv = torch.nn.functional.mse_loss(est, tm)
    names = [k for k in Pg if is_trainable(k)]
    grads = dict(zip(names, torch.autograd.grad(lossv, [Pg[k] for k in names])))
    assert rel_err(est, g["g/est_mags"]) < 2e-5
    assert rel_err(tm, g["g/target_mags"]) < 2e-5
    assert rel_err(wav, g["g/out_wav"]) < 2e-2        # phase of numerically-zero noisy bins: see test_plan_hostsim.py
    assert abs(float(lossv) - float(g["g/loss"])) < 2e-5 * max(1.0, abs(float(g["g/loss"])))
    noise = lambda k: k.endswith("conv.bias") and not k.startswith("decoder.5.")
    for k, v in sub(g, "g/grad_norm").items():
        if not noise(k):
            assert abs(float(grads[k].double().norm()) - float(v)) <= 3e-4 * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if not noise(k):
            assert rel_err(grads[k], v) < 3e-4, k
