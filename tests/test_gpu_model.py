"""GPU tier: the drop-in `models.DCCRN` (HIP library through the C ABI) against the golden vectors captured from the
real reference and against the oracle on fresh seeded inputs: outputs, loss, every parameter gradient, BatchNorm
running statistics and the parameters after one Adam step.  Tolerance: 1e-3 relative fp32 (BASELINE north_star)."""
import numpy as np
import pytest
import torch

from oracle.dccrn import DCCRNConfig, dccrn_state_shapes, is_trainable
from oracle.step import dccrn_train_step
from oracle.weights import fill_state_dict_, formula_state_dict, test_signals as make_signals
from util import knobs, load_golden, rel_err, rel_l2, sub

pytestmark = pytest.mark.gpu
TOL = 1e-3

CASES = [
    ("small_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR"),
    ("small_C_sdr", (16, 32, 32, 64, 64, 64), 128, "C", "SDR"),
    ("small_R_mse", (16, 32, 32, 64, 64, 64), 128, "R", "MSE"),
    ("small_E_sisdr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SDR"),
    ("default_E_sisnr", (32, 64, 128, 256, 256, 256), 256, "E", "SI-SNR"),
    ("wide_C_sdr", (16, 32, 32, 64, 64, 64), 512, "C", "SDR"),        # rnn_units 512: per-time-step LSTM path (plan.cpp `stepped`)
    ("real_E_sisnr", (16, 32, 32, 64, 64, 64), 256, "E", "SI-SNR"),   # cfg.lstm == 'real': nn.LSTM(2 layers) + tranform
    ("large_C_sisnr", (64, 128, 256, 512, 512, 512), 512, "C", "SI-SNR"),   # BASELINE configs[4]: DCCRN-large (2x channels, rnn_units 512)
    ("noskip_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR"),       # cfg.skip_type = False (models.py:107-137, 222-223)
    ("cbn_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR"),          # DCCRN(use_cbn=True): ComplexBatchNorm (tools_for_model.py:430-607)
    ("rectwin_C_sisnr", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR"),      # win_type=None: rectangular window (tools_for_model.py:17-18)
    ("hamming_C_sisnr", (16, 32, 32, 64, 64, 64), 128, "C", "SI-SNR"),      # win_type='hamming': any scipy.signal.get_window name (tools_for_model.py:19-20) - the host hands the table to the planner
]


def case_meta(g):
    """(skip_type, input scale, gradient sample stride) of a golden; older fixtures predate these fields."""
    return (bool(int(g["g/meta/skip"])) if "g/meta/skip" in g else True, float(g["g/meta/scale"]) if "g/meta/scale" in g else 1.0,
            int(g["g/meta/gstride"]) if "g/meta/gstride" in g else 53)


def make_model(kn, ru, mask, loss, lstm="complex", skip=True, dtype="fp32", use_cbn=False, win_type="hanning"):
    import sefd_amd
    from sefd_amd import config as cfg, models
    cfg.dccrn_kernel_num = list(kn)
    cfg.masking_mode = mask
    cfg.loss = loss
    cfg.perceptual = False
    cfg.lstm = lstm
    cfg.skip_type = skip
    cfg.act_dtype = dtype
    try:
        m = models.DCCRN(rnn_units=ru, masking_mode=mask, use_cbn=use_cbn, win_type=win_type)
    finally:
        cfg.skip_type = True
    fill_state_dict_(m)
    return m.to("cuda")


def noise_bias(k):
    return k.endswith("conv.bias") and not k.startswith("decoder.5.")


@pytest.mark.parametrize("name,kn,ru,mask,loss", CASES)
def test_module_step_against_reference_golden(name, kn, ru, mask, loss):
    g = load_golden("dccrn_" + name)
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    skip, scale, gstride = case_meta(g)
    m = make_model(kn, ru, mask, loss, lstm="real" if name.startswith("real") else "complex", skip=skip, use_cbn=name.startswith("cbn"),
                   win_type=None if name.startswith("rectwin") else "hamming" if name.startswith("hamming") else "hanning")
    m.train()
    x, y = make_signals(B, L)
    x, y = (x * scale).cuda(), (y * scale).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    P0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    o_r, o_i, wav = m(x, y)
    lossv = m.loss(wav, y)
    opt.zero_grad()
    lossv.backward()
    assert rel_err(o_r, g["g/out_real"]) < TOL
    assert rel_err(o_i, g["g/out_imag"]) < TOL
    assert rel_err(wav, g["g/out_wav"]) < TOL
    assert abs(float(lossv) - float(g["g/loss"])) < TOL * max(1.0, abs(float(g["g/loss"])))
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    gn = sub(g, "g/grad_norm")
    for k, v in gn.items():
        if noise_bias(k):
            continue
        assert abs(float(grads[k].double().norm()) - float(v)) <= TOL * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if noise_bias(k):
            assert float(grads[k].abs().max()) < 1e-4 * float(gn[k.replace(".bias", ".weight")]) + 1e-7, k
            continue
        # gradient criterion: 1e-3 of the tensor's L2 norm, and no single element off by more than 5e-3 of the largest
        # (mask mode E back-propagates through atan2 / 1/|m|: fp32 round-off of a few bins is amplified in the reference too)
        tol = 5e-3 if k.endswith(".2.weight") else TOL      # PReLU slope: a single heavily-cancelling sum
        assert rel_l2(grads[k], v) < tol and rel_err(grads[k], v) < 5e-3, k
    for k, v in sub(g, "g/grad_samp").items():
        if not noise_bias(k):
            assert rel_err(grads[k].reshape(-1)[::gstride], v) < TOL, k
    opt.step()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k, v in sub(g, "g/running").items():
        assert rel_err(sd[k], v) < TOL, k
    for k, v in sub(g, "g/after_adam").items():
        if noise_bias(k):
            continue
        assert np.abs((sd[k].numpy() - P0[k].numpy()) - (v - P0[k].numpy())).max() < 5e-5, k   # updates are ~lr = 1e-3
    assert int(sd["encoder.0.1.num_batches_tracked"]) == 1


def test_fused_train_step_matches_oracle_two_steps():
    """model.train_step (no autograd, flat fused Adam) == oracle forward/backward/Adam, two consecutive steps."""
    from sefd_amd.optim import Adam
    kn, ru = (16, 32, 32, 64, 64, 64), 128
    m = make_model(kn, ru, "C", "SI-SNR")
    m.train()
    B, L = 3, 4000
    torch.manual_seed(11)
    y = torch.randn(B, L) * 0.1
    x = y + 0.05 * torch.randn(B, L)
    cfg = DCCRNConfig(kernel_num=kn, rnn_units=ru, masking_mode="C")
    P = formula_state_dict(dccrn_state_shapes(cfg))
    opt = Adam(m.parameters(), lr=1e-3)
    state = None
    prev_sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    for step in (1, 2):
        prev_P = P
        r = dccrn_train_step(P, cfg, x, y, loss_kind="SI-SNR", adam_state=state, step=step)
        loss = m.train_step(x.cuda(), y.cuda(), opt)
        assert abs(float(loss) - float(r["loss"])) < TOL * max(1.0, abs(float(r["loss"]))), step
        P = {**P, **r["new_params"], **r["new_stats"]}
        state = r["adam_state"]
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        for k in r["new_stats"]:
            if step > 1 and k.endswith("running_mean"):
                # the conv bias in front of a BatchNorm random-walks by +-lr per step on the sign of a rounding-noise
                # gradient (same in the reference).  The complex conv's effective bias is b_r -+ b_i, so it can differ by up to
                # 4*lr from the oracle's, the batch mean with it, and the running mean by 0.1 of that (4e-4)
                assert float((sd[k] - P[k]).abs().max()) < 8e-4, (step, k)
            else:
                assert rel_err(sd[k], P[k]) < TOL, (step, k)
        # Adam's step is lr * m/(sqrt(v)+eps) ~ lr * sign(g) on the first steps: compare the parameters where the oracle
        # gradient is well above rounding noise (elsewhere the *sign* of a noise-level gradient decides a +-lr move)
        worst, covered, total = 0.0, 0, 0
        for k in r["new_params"]:
            if noise_bias(k):
                continue
            gref = r["grads"][k]
            mask = gref.abs() > 1e-3 * gref.abs().max()
            covered += int(mask.sum())
            total += mask.numel()
            if mask.any():
                worst = max(worst, float(((sd[k] - prev_sd[k]) - (P[k] - prev_P[k]))[mask].abs().max()))
        assert covered > 0.5 * total
        assert worst < 5e-5, (step, worst)                 # this step's parameter updates; they are O(lr) = 1e-3
        prev_sd = sd


def test_full_length_clip_and_properties():
    """BASELINE-size clip (3 s @ 16 kHz, T = 483): golden outputs, plus size-independent properties."""
    g = load_golden("dccrn_default_C_sisnr_full")
    m = make_model((32, 64, 128, 256, 256, 256), 256, "C", "SI-SNR")
    m.train()
    x, y = make_signals(1, 48000)
    o_r, o_i, wav = m(x.cuda(), y.cuda())
    assert rel_err(o_r, g["g/out_real"]) < TOL
    assert rel_err(wav, g["g/out_wav"]) < TOL
    assert abs(float(m.loss(wav, y.cuda())) - float(g["g/loss"])) < TOL * abs(float(g["g/loss"]))
    assert float(wav.abs().max()) <= 1.0                        # clamp (models.py:282)
    assert float(o_r[:, 0].abs().max()) == 0.0                  # zero DC row
    # batch independence in eval mode: utterance b of a batch == the same utterance alone
    m.eval()
    xs, _ = make_signals(3, 8000)
    with torch.no_grad():
        full = m(xs.cuda())[2]
        one = m(xs[1:2].cuda())[2]
    assert rel_err(full[1:2], one) < 1e-5


@pytest.mark.parametrize("gold", ["dccrn_small_eval", "dccrn_cbn_eval"])
def test_validation_path_against_reference_golden(tmp_path, gold):
    """`trainer.model_validate` (reference trainer.py:188-241): eval-mode plans with the running statistics a
    training-mode forward just updated, no gradients; enhanced waveform and loss vs the reference golden (BatchNorm2d and ComplexBatchNorm)."""
    from sefd_amd import trainer
    g = load_golden(gold)
    kn = tuple(int(k) for k in g["g/meta/kernel_num"])
    m = make_model(kn, int(g["g/meta/rnn_units"]), str(g["g/meta/mask"]), str(g["g/meta/loss"]), use_cbn=gold.endswith("cbn_eval"))
    m.train()
    x, y = make_signals(int(g["g/meta/B"]), int(g["g/meta/L"]))
    with torch.no_grad():
        _, _, wav0 = m(x.cuda(), y.cuda())
        assert abs(float(m.loss(wav0, y.cuda())) - float(g["g/train_loss"])) < TOL * abs(float(g["g/train_loss"]))
    xv, yv = make_signals(int(g["g/meta/Bv"]), int(g["g/meta/Lv"]))
    xv, yv = xv.flip(0) * 0.8, yv.flip(0) * 0.8
    m.eval()
    with torch.no_grad():
        _, _, wv = m(xv.cuda(), yv.cuda())
    assert rel_err(wv, g["g/val_wav"]) < TOL
    calls = []

    def fake_pesq(est, clean):
        calls.append(est.shape)
        return np.full(len(est), 2.5)

    def fake_stoi(est, clean):
        return np.full(len(est), 0.9)

    m.train()
    vloss, pesq, stoi = trainer.model_validate(m, [(xv, yv)], None, str(tmp_path), 3, "cuda", scorers=(fake_pesq, fake_stoi))
    assert m.training                                             # mode restored
    assert abs(float(vloss) - float(g["g/val_loss"])) < TOL * abs(float(g["g/val_loss"]))
    assert calls == [tuple(xv.shape)] and abs(pesq - 2.5) < 1e-9 and abs(stoi - 0.9) < 1e-9
    lines = open(tmp_path / "Epoch_3_SCORES").read().strip().splitlines()
    assert lines == ["PESQ 2.500000 | STOI 0.900000"] * len(xv)
    vloss2, p2, s2 = trainer.model_validate(m, [(xv, yv)], None, str(tmp_path), 4, "cuda", scorers=None)
    assert abs(float(vloss2) - float(vloss)) < 1e-6 and p2 != p2 and s2 != s2      # NaN without scorers


def test_cpu_tensor_is_rejected_not_silently_computed():
    m = make_model((16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR")
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4000))


# ------------------------------------------------------------------------------------------------ CRN (models.py:329-565)
@pytest.mark.parametrize("name,kn,ru,ri,loss", [("default_E_mse", (32, 64, 128, 256, 256, 256), 256, 512, "MSE"),
                                               ("small_E_sisnr", (16, 32, 32, 64, 64, 64), 128, 128, "SI-SNR")])
def test_crn_module_step_against_reference_golden(name, kn, ru, ri, loss):
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    g = load_golden("crn_" + name)
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.skip_type, cfg.act_dtype = list(kn), "E", loss, False, True, "fp32"
    m = models.CRN(rnn_units=ru, rnn_input_size=ri, masking_mode="E")
    fill_state_dict_(m)
    m = m.to("cuda").train()
    x, y = make_signals(B, L)
    x, y = x.cuda(), y.cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    P0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    est_mags, target_mags, wav = m(x, y)
    lossv = m.loss(wav, y)
    opt.zero_grad()
    lossv.backward()
    assert rel_err(est_mags, g["g/est_mags"]) < TOL
    assert rel_err(target_mags, g["g/target_mags"]) < TOL
    assert rel_err(wav, g["g/out_wav"]) < TOL
    assert abs(float(lossv) - float(g["g/loss"])) < TOL * max(1.0, abs(float(g["g/loss"])))
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    gn = sub(g, "g/grad_norm")
    for k, v in gn.items():
        if not noise_bias(k):
            assert abs(float(grads[k].double().norm()) - float(v)) <= TOL * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if noise_bias(k):
            continue
        assert rel_l2(grads[k], v) < (5e-3 if k.endswith(".2.weight") else TOL) and rel_err(grads[k], v) < 5e-3, k
    opt.step()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    for k, v in sub(g, "g/running").items():
        assert rel_err(sd[k], v) < TOL, k
    for k, v in sub(g, "g/after_adam").items():
        if not noise_bias(k):
            assert np.abs((sd[k].numpy() - P0[k].numpy()) - (v - P0[k].numpy())).max() < 5e-5, k
    with pytest.raises(AttributeError):
        m(x)                      # the reference crashes without targets too (models.py:505)


def test_crn_direct_mode_against_reference_golden():
    """CRN spectral mapping through `trainer.crn_direct_train` (trainer.py:150-181): loss on the mapped magnitudes."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models, trainer
    g = load_golden("crn_small_direct_mse")
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    kn = (16, 32, 32, 64, 64, 64)
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.skip_type, cfg.act_dtype = list(kn), "Direct(None make)", "MSE", False, True, "fp32"
    m = models.CRN(rnn_units=128, rnn_input_size=128, masking_mode="Direct(None make)")
    fill_state_dict_(m)
    m = m.to("cuda").train()
    x, y = make_signals(B, L)
    out_mags, target_mags, wav = m(x.cuda(), y.cuda())
    lossv = m.loss(out_mags, target_mags)
    lossv.backward()
    assert rel_err(out_mags, g["g/est_mags"]) < TOL
    assert rel_err(target_mags, g["g/target_mags"]) < TOL
    assert rel_err(wav, g["g/out_wav"]) < 2e-2          # phase of numerically-zero noisy bins decides a sign (tests/test_plan_hostsim.py)
    assert abs(float(lossv) - float(g["g/loss"])) < TOL * max(1.0, abs(float(g["g/loss"])))
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    for k, v in sub(g, "g/grad_norm").items():
        if not noise_bias(k):
            assert abs(float(grads[k].double().norm()) - float(v)) <= TOL * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if not noise_bias(k):
            assert rel_l2(grads[k], v) < (5e-3 if k.endswith(".2.weight") else TOL), k
    # one epoch of the trainer mirror over a single batch returns that batch's loss
    m.zero_grad()
    opt = torch.optim.SGD(m.parameters(), lr=0.0)
    ep = trainer.crn_direct_train(m, opt, [(x, y)], "cuda")
    assert abs(float(ep) - float(g["g/loss"])) < TOL * max(1.0, abs(float(g["g/loss"])))
    cfg.masking_mode = "E"


# ------------------------------------------------------------------------------------------------ LMS (tools_for_loss.py:120-249)
def test_lms_loss_kernel_vs_oracle():
    import sefd_amd  # noqa: F401
    from sefd_amd import tools_for_loss as tfl
    from oracle import losses as ol
    torch.manual_seed(5)
    B, NF, T = 3, 257, 43
    cr, ci, er, ei = [torch.randn(B, NF, T) * 3 for _ in range(4)]
    er_, ei_ = er.clone().requires_grad_(True), ei.clone().requires_grad_(True)
    cm = torch.sqrt(cr ** 2 + ci ** 2 + 1e-7)
    em = torch.sqrt(er_ ** 2 + ei_ ** 2 + 1e-7)
    ref = ol.lms_loss(cm, em)
    ref.backward()
    d = lambda t: t.cuda()
    erd, eid = d(er).requires_grad_(True), d(ei).requires_grad_(True)
    out = tfl.lms_from_spectra(d(cr), d(ci), erd, eid)
    (out * 0.5).backward()
    assert abs(float(out) - float(ref)) < 1e-4 * abs(float(ref))
    assert rel_l2(erd.grad.cpu(), 0.5 * er_.grad) < TOL and rel_l2(eid.grad.cpu(), 0.5 * ei_.grad) < TOL
    # magnitude signature get_array_lms_loss(clean_mags, est_mags)
    emd = d(em.detach()).requires_grad_(True)
    out2 = tfl.get_array_lms_loss(d(cm), emd)
    out2.backward()
    em2 = em.detach().clone().requires_grad_(True)
    ref2 = ol.lms_loss(cm, em2)
    ref2.backward()
    assert abs(float(out2) - float(ref2)) < 1e-4 * abs(float(ref2))
    assert rel_l2(emd.grad.cpu(), em2.grad) < TOL


@pytest.mark.parametrize("name,win", [("small_E_sisnr_lms", "hanning"), ("hamming_E_sisnr_lms", "hamming")])
def test_dccrn_lms_joint_step_against_reference_golden(name, win):
    """model_perceptual_train (trainer.py:45-82): loss = (SI-SNR + LMS) / 2, forward called without targets.  The clean spectrum is self.stft(target):
    the MODEL's window (round 5 took Hann whatever win_type was)."""
    g = load_golden("dccrn_" + name)
    from sefd_amd import config as cfg
    m = make_model((16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR", win_type=win)
    cfg.perceptual = "LMS"
    try:
        m.train()
        x, y = make_signals(2, 4000)
        x, y = x.cuda(), y.cuda()
        real_spec, img_spec, wav = m(x)
        main = m.loss(wav, y)
        perc = m.loss(wav, y, real_spec, img_spec, perceptual=True)
        lossv = (main + perc) / 2
        lossv.backward()
    finally:
        cfg.perceptual = False
    assert abs(float(main) - float(g["g/main_loss"])) < TOL * abs(float(g["g/main_loss"]))
    assert abs(float(perc) - float(g["g/perc_loss"])) < TOL * abs(float(g["g/perc_loss"]))
    assert abs(float(lossv) - float(g["g/loss"])) < TOL * abs(float(g["g/loss"]))
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    for k, v in sub(g, "g/grad_norm").items():
        if not noise_bias(k):
            assert abs(float(grads[k].double().norm()) - float(v)) <= 2 * TOL * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if not noise_bias(k):
            assert rel_l2(grads[k], v) < (5e-3 if k.endswith(".2.weight") else 2 * TOL), k


# ------------------------------------------------------------------------------------------------ FullSubNet (models.py:568-682)
@pytest.mark.parametrize("name,hid,seq,norm", [("small_mse", (128, 64), "LSTM", "offline_laplace_norm"),
                                               ("default_mse", (512, 384), "LSTM", "offline_laplace_norm"),
                                               ("small_gru_mse", (128, 64), "GRU", "offline_laplace_norm"),
                                               ("small_cumlaplace_mse", (128, 64), "LSTM", "cumulative_laplace_norm"),
                                               ("small_gaussian_mse", (128, 64), "LSTM", "offline_gaussian_norm"),
                                               ("small_cumlayer_gru_mse", (128, 64), "GRU", "cumulative_layer_norm"),
                                               ("small_sdr", (128, 64), "LSTM", "offline_laplace_norm"),
                                               ("small_sisnr", (128, 64), "LSTM", "offline_laplace_norm"),
                                               ("small_sisdr", (128, 64), "LSTM", "offline_laplace_norm")])
def test_fullsubnet_step_against_reference_golden(name, hid, seq, norm):
    """trainer.py:85-118 with the inter-layer dropout disabled on both sides; cfg.sequence_model / cfg.norm_type variants;
    FullSubNet.loss (models.py:674-682) with cfg.loss = SDR (config.py:36: the reference's default) / SI-SNR / SI-SDR over the 2-element last axis."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models, tools_for_model as tools
    g = load_golden("fsn_" + name)
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    cfg.loss, cfg.act_dtype = (str(g["g/meta/loss"]) if "g/meta/loss" in g else "MSE"), "fp32"
    m = models.FullSubNet(fb_model_hidden_size=hid[0], sb_model_hidden_size=hid[1], sequence_model=seq, norm_type=norm)
    fill_state_dict_(m)
    m = m.to("cuda").train()
    m.dropout_keep = 1.0
    x, y = make_signals(B, L)
    x, y = x.cuda(), y.cuda()
    nc, cc = tools.stft(x), tools.stft(y)
    noisy_mag, _ = tools.mag_phase(nc)
    cirm = tools.build_complex_ideal_ratio_mask(nc, cc)
    assert rel_err(noisy_mag[:, ::4, ::3], g["g/noisy_mag"]) < TOL
    assert rel_err(cirm[:, ::4, ::3], g["g/cirm"]) < TOL
    cirm_own = cirm
    if "g/cirm_full" in g:      # SDR / SI-SNR / SI-SDR over 2-element rows amplify the cIRM's noise / noise values at near-silent bins: same target
        cirm = torch.from_numpy(g["g/cirm_full"]).cuda()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    crm = m(noisy_mag)
    lossv = m.loss(cirm, crm)
    opt.zero_grad()
    lossv.backward()
    loss_kind, cfg.loss = cfg.loss, "MSE"
    assert rel_err(crm, g["g/crm"]) < TOL
    # SI-SDR: -10 log10 of the MEAN of per-row ratios P / N over two-element rows - a handful of near-parallel rows (N -> 0) carry the mean, so
    # the 2e-6 of the fp32 cRM shows as 2e-3 of the loss (measured -49.845 vs -49.943; same kernel, same formula in torch on the GPU tensors: 1e-5)
    ltol = 5e-3 if loss_kind == "SI-SDR" else TOL
    assert abs(float(lossv) - float(g["g/loss"])) < ltol * abs(float(g["g/loss"]))
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    # SI-SNR / SI-SDR over two-element rows: the per-row ratios carry 1 / (|e - a t|^2 + eps) factors of near-parallel pairs, so the 2e-6
    # by which the fp32 cRM differs from the reference's is amplified ~1000x in the gradient (the reference on another machine moves the same
    # way); the loss kernels themselves are held to 1e-3 on well-conditioned rows in test_short_row_losses_both_slots
    gtol = 1e-1 if loss_kind == "SI-SDR" else 1e-2 if loss_kind == "SI-SNR" else TOL     # (SI-SDR: the whole gradient scales with 1 / mean ratio - measured 4.3 % off as ONE common factor)
    for k, v in sub(g, "g/grad_norm").items():
        assert abs(float(grads[k].double().norm()) - float(v)) <= gtol * float(v) + 1e-9, k
    for k, v in sub(g, "g/grad").items():
        assert rel_l2(grads[k], v) < gtol, k
    for k, v in sub(g, "g/grad_samp").items():
        assert rel_l2(grads[k].reshape(-1)[::211], v) < gtol, k
    if loss_kind != "MSE":
        # the fused step (plan io buffers, no autograd; it builds its own cIRM) == the autograd route on that cIRM: same loss, and the
        # parameters after it are Adam's first step on the autograd gradient, p - lr g / (|g| + eps)
        from sefd_amd.optim import Adam
        cfg.loss = loss_kind
        try:
            m.zero_grad()
            own = m.loss(cirm_own, m(noisy_mag))
            own.backward()
            g_own, p0 = m._flat_grad.clone(), m._flat_param.clone()
            for (off, n, _), (_, p) in zip(m._param_slices, m._trainable()):
                g_own[off:off + n].copy_(p.grad.reshape(-1))
            fused = float(m.train_step(x, y, Adam(m.parameters(), lr=1e-3), loss_kind=loss_kind))
        finally:
            cfg.loss = "MSE"
        assert abs(fused - float(own)) < 1e-5 * max(1.0, abs(float(own))), (fused, float(own))
        expect = p0 - 1e-3 * g_own / (g_own.abs() + 1e-8)
        big = g_own.abs() > 1e-4 * g_own.abs().max()                   # (where the gradient is rounding noise its sign is too)
        assert float((m._flat_param - expect)[big].abs().max()) < 2e-5


def test_fullsubnet_fused_train_step_and_dropout():
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    cfg.loss, cfg.act_dtype = "MSE", "fp32"
    m = models.FullSubNet(fb_model_hidden_size=128, sb_model_hidden_size=64)
    fill_state_dict_(m)
    m = m.to("cuda").train()
    x, y = make_signals(2, 6000)
    opt = Adam(m.parameters(), lr=1e-3)
    losses = [float(m.train_step(x.cuda(), y.cuda(), opt)) for _ in range(6)]      # dropout keep 0.2 active
    assert all(np.isfinite(losses)) and min(losses[3:]) < losses[0]


@pytest.mark.parametrize("name,win", [("small_direct_mse", "hanning"), ("hamming_direct_mse", "hamming")])
def test_dccrn_direct_mode_against_reference_golden(name, win):
    """masking_mode 'Direct(None make)' + dccrn_direct_train's loss (trainer.py:135-138) on spectra [B, 257, T] (T = 43: unaligned rows); the target
    spectra come from the model's own ConvSTFT, window included."""
    g = load_golden("dccrn_" + name)
    m = make_model((16, 32, 32, 64, 64, 64), 128, "Direct(None make)", "MSE", win_type=win)
    m.train()
    x, y = make_signals(2, 4000)
    o_r, t_r, o_i, t_i, wav = m(x.cuda(), y.cuda())
    lossv = (m.loss(o_r, t_r) + m.loss(o_i, t_i)) / 2
    lossv.backward()
    assert rel_err(o_r, g["g/out_real"]) < TOL and rel_err(o_i, g["g/out_imag"]) < TOL
    assert rel_err(t_r, g["g/target_real"]) < TOL and rel_err(t_i, g["g/target_imag"]) < TOL and rel_err(wav, g["g/out_wav"]) < TOL
    assert abs(float(lossv) - float(g["g/loss"])) < TOL * float(g["g/loss"])
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    for k, v in sub(g, "g/grad_norm").items():
        if not noise_bias(k):
            assert abs(float(grads[k].double().norm()) - float(v)) <= TOL * float(v) + 1e-7, k
    for k, v in sub(g, "g/grad").items():
        if not noise_bias(k):
            assert rel_l2(grads[k], v) < (5e-3 if k.endswith(".2.weight") else TOL), k


# ------------------------------------------------------------------------------------------------ bf16: the benchmarked dtype
# Error budget of the bf16 mode (activations, packed weights and MFMA operands stored as bf16, fp32 accumulate) against the
# fp32 reference goldens.  One bf16 store rounds to 8 significant bits: relative error <= 2^-9 = 1.95e-3, rms 1.1e-3.
# Forward: ~13 conv/LSTM layers x ~4 roundings each (operand, weight, pre-BN output, post-PReLU output); BatchNorm re-normalises
# after every layer, so the errors add in quadrature: sqrt(52) x 1.1e-3 = 8e-3 relative (L2) at the output.  Measured (MI355X,
# round 2): 3e-3 - 8e-3 (L2), 4e-3 - 1.5e-2 (max norm).  Budget: 2e-2 (L2), 5e-2 (max norm = 5 sigma).
# Gradients: the backward starts from the residual est - a * tgt of the SI-SNR / SDR losses, whose relative error is the
# output error amplified by |est| / |residual| (x4 at 12 dB): ~2e-2 before back-propagation begins; the chain then doubles in
# length and BatchNorm statistics / PReLU masks perturb whole sums coherently.  Measured: median over the parameter tensors
# 4e-2 - 6e-2 (L2); the direction of the full gradient (cosine over all sampled elements) agrees to < 1e-2 off 1.  Single scalars
# that are heavily cancelling sums (a PReLU slope, an LSTM bias element) are off by up to 0.5 relative - their terms are
# individually accurate to 2^-9, the sum is 100x smaller than its terms.  Budgets: median 8e-2, worst tensor 0.7, cosine > 0.99.
# Measured values go to gpurun_out/r02_bf16_parity.json (copied to profiles/).
# Round 5: the worst-tensor budget is split.  PReLU slopes (`*.2.weight`, ONE scalar per layer = a heavily cancelling sum over every
# activation of the layer) keep a wide budget: tools/bf16_slope_analysis.py runs the same bf16 plan on the host simulator, which accumulates every
# sum in DOUBLE and only keeps the bf16 STORAGE roundings - it is 0.83 off on encoder.1.2.weight where the MI355X kernels are 0.61 off (fp32 storage:
# 6e-7): the error is the rounding of the stored y / dz, not the kernels' accumulation (profiles/r05_bf16_slope_analysis.json).  Every OTHER tensor
# must stay below 0.35 (measured worst: 0.26, an LSTM bias) - that is the guard against a broken kernel.
# (round-5 GPU suite, profiles/r05_bf16_parity.json: worst other tensor 0.26 - an LSTM bias of the small SDR model -, worst slope 0.87; the slope
# budget only says "a number of the right order": the simulator's exact-accumulation value is itself 0.83 off)
BF16_OUT_L2, BF16_OUT_MAX, BF16_GRAD_L2, BF16_GRAD_WORST, BF16_GRAD_COS, BF16_LOSS = 2e-2, 5e-2, 8e-2, 0.35, 0.99, 2e-2
BF16_SLOPE_WORST = 0.9        # PReLU slopes against the fp32 goldens (B = 2, tone inputs): worst measured 0.865 (encoder.1 of the default model), storage
                               # rounding (r05 notes section 5); noise / bias split: test_bf16_prelu_slope_gradient_error_is_noise_that_averages_out
_BF16_REPORT = {}


def _bf16_record(name, rec):
    import json, os
    _BF16_REPORT[name] = rec
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, "bf16_parity.json")
    old = {}
    if os.path.exists(path):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(_BF16_REPORT)
    old["_budget"] = dict(out_rel_l2=BF16_OUT_L2, out_rel_max=BF16_OUT_MAX, grad_rel_l2_median=BF16_GRAD_L2, grad_rel_l2_worst=BF16_GRAD_WORST, prelu_slope_rel_worst=BF16_SLOPE_WORST,
                          grad_cosine_min=BF16_GRAD_COS, loss_rel=BF16_LOSS,
                          note="bf16 storage / MFMA operands, fp32 accumulate, vs fp32 goldens captured from the reference")
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def _grad_report(grads, g, gstride):
    worst, vals, slopes = ("", 0.0), [], {}
    dot = na = nb = 0.0
    pairs = [(k, grads[k], v) for k, v in sub(g, "g/grad").items()] + \
            [(k, grads[k].reshape(-1)[::gstride], v) for k, v in sub(g, "g/grad_samp").items()]
    for k, mine, v in pairs:
        if noise_bias(k):
            continue
        e = rel_l2(mine, v)
        vals.append(e)
        if k.endswith(".2.weight") or "lstm.bias_" in k:
            slopes[k] = e                      # cancelling sums - a PReLU slope (one scalar per layer), an LSTM bias gradient (a sum over every frame and
                                               # utterance): their own budget (BF16_SLOPE_WORST); both are checked against the model's own fp32 plan on
                                               # broadband inputs in test_bf16_prelu_slope_gradient_error_is_noise_that_averages_out
        elif e > worst[1]:
            worst = (k, e)
        a, b_ = mine.detach().double().reshape(-1), torch.as_tensor(np.asarray(v)).double().reshape(-1)
        dot += float((a * b_).sum()); na += float((a * a).sum()); nb += float((b_ * b_).sum())
    gn = sub(g, "g/grad_norm")
    nr = max(abs(float(grads[k].double().norm()) / float(v) - 1.0) for k, v in gn.items() if not noise_bias(k) and float(v) > 0 and not k.endswith(".2.weight"))
    return dict(grad_rel_l2_worst=worst[1], grad_rel_l2_worst_name=worst[0], grad_rel_l2_median=float(np.median(vals)),
                grad_norm_ratio_worst=nr, grad_cosine=dot / max((na * nb) ** 0.5, 1e-300), slope_rel=slopes,
                slope_rel_worst=max(slopes.values()) if slopes else 0.0)


@pytest.mark.parametrize("name,kn,ru,mask,loss", [("default_E_sisnr", (32, 64, 128, 256, 256, 256), 256, "E", "SI-SNR"),
                                                  ("large_C_sisnr", (64, 128, 256, 512, 512, 512), 512, "C", "SI-SNR"),
                                                  ("small_C_sdr", (16, 32, 32, 64, 64, 64), 128, "C", "SDR"),
                                                  ("cbn_E_sisnr", (16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR")])
def test_bf16_dccrn_step_against_reference_golden(name, kn, ru, mask, loss):
    g = load_golden("dccrn_" + name)
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    skip, scale, gstride = case_meta(g)
    m = make_model(kn, ru, mask, loss, skip=skip, dtype="bf16", use_cbn=name.startswith("cbn"))
    m.train()
    x, y = make_signals(B, L)
    x, y = (x * scale).cuda(), (y * scale).cuda()
    o_r, o_i, wav = m(x, y)
    lossv = m.loss(wav, y)
    lossv.backward()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    rec = dict(out_wav_rel_l2=rel_l2(wav, g["g/out_wav"]), out_wav_rel_max=rel_err(wav, g["g/out_wav"]),
               out_real_rel_l2=rel_l2(o_r, g["g/out_real"]), out_real_rel_max=rel_err(o_r, g["g/out_real"]),
               loss=float(lossv), loss_ref=float(g["g/loss"]), B=B, L=L, **_grad_report(grads, g, gstride))
    rec["loss_rel"] = abs(rec["loss"] - rec["loss_ref"]) / max(1.0, abs(rec["loss_ref"]))
    _bf16_record("dccrn_" + name, rec)
    assert rec["out_wav_rel_l2"] < BF16_OUT_L2 and rec["out_wav_rel_max"] < BF16_OUT_MAX, rec
    assert rec["out_real_rel_l2"] < BF16_OUT_L2 and rec["out_real_rel_max"] < BF16_OUT_MAX, rec
    assert rec["loss_rel"] < BF16_LOSS, rec
    assert rec["grad_rel_l2_median"] < BF16_GRAD_L2 and rec["grad_rel_l2_worst"] < BF16_GRAD_WORST and rec["slope_rel_worst"] < BF16_SLOPE_WORST, rec
    assert rec["grad_cosine"] > BF16_GRAD_COS and rec["grad_norm_ratio_worst"] < BF16_GRAD_WORST, rec


def test_bf16_prelu_slope_gradient_error_is_noise_that_averages_out():
    """the bf16 plan's PReLU-slope gradients are up to 0.87 off the fp32 GOLDENS (B = 2, pure-tone inputs: every term of a slope's
    sum carries the 2^-9 rounding of the stored y and dz, and on those inputs the sum is ~100x smaller than its terms).  Claimed: zero-mean storage NOISE,
    not a bias of the kernels.  Checked here against the model's OWN fp32 plan on broadband random inputs (the deterministic tones of the goldens make the
    SI-SNR loss itself ill-conditioned: at B = 8 the whole gradient of the bf16 plan came out scaled by one common factor), K = 4 input seeds, B = 2 and
    B = 32 at L = 16 000, per slope d = g_bf16 - g_fp32:
      * the slope gradient as a VECTOR (11 scalars) is accurate at every B: ||d|| / ||g_fp32|| <= 2e-2 (measured 2e-3 ... 3e-3);
      * the NOISE part std(d) / rms(g_fp32) falls with the number of summed elements: median over the slopes of its B = 32 / B = 2 ratio <= 0.7
        (measured 0.27 and 0.41 on two builds, K = 4 seeds; 1 / sqrt(16) = 0.25);
      * the SYSTEMATIC part |mean(d)| / rms(g_fp32) stays <= 0.5 for every slope (measured: 0.30-0.34 for the first encoder layer's slope at B = 32 - its
        gradient is 1e-3 of the vector's norm -, <= 0.12 elsewhere): reported in gpurun_out/bf16_parity.json, not noise, not growing the vector error."""
    kn, ru = (32, 64, 128, 256, 256, 256), 256
    res = {}
    for B in (2, 32):
        f, d, vec, lb = [], [], [], []
        for seed in range(4):
            g = torch.Generator().manual_seed(100 + seed)
            clean = 0.1 * torch.randn(B, 16000, generator=g)
            x, y = (clean + 0.05 * torch.randn(B, 16000, generator=g)).cuda(), clean.cuda()
            gr = {}
            for dt in ("fp32", "bf16"):
                m = make_model(kn, ru, "E", "SI-SNR", dtype=dt)
                m.train()
                _, _, wav = m(x, y)
                m.loss(wav, y).backward()
                gr[dt] = torch.stack([p.grad.detach().double().cpu().reshape(()) for k, p in m.named_parameters() if k.endswith(".2.weight")])
                gr[dt + "_lstm_bias"] = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters() if "lstm.bias_" in k}
                del m
            f.append(gr["fp32"].numpy())
            d.append((gr["bf16"] - gr["fp32"]).numpy())
            vec.append(float(np.linalg.norm(d[-1]) / np.linalg.norm(f[-1])))
            lb.append(max(float((gr["bf16_lstm_bias"][k] - v).norm() / v.norm()) for k, v in gr["fp32_lstm_bias"].items()))
        f, d = np.stack(f), np.stack(d)
        rms_f = np.sqrt((f ** 2).mean(0))
        res[B] = dict(vec_rel=vec, lstm_bias_rel_worst=lb, noise=(d.std(0, ddof=1) / rms_f).tolist(), bias=(np.abs(d.mean(0)) / rms_f).tolist())
    _bf16_record("prelu_slope_vs_own_fp32_plan", {str(b): r for b, r in res.items()})
    assert len(res[2]["noise"]) == 11
    assert max(res[2]["vec_rel"]) <= 2e-2 and max(res[32]["vec_rel"]) <= 2e-2, res
    # every LSTM bias gradient tensor (relative L2; measured 0.10-0.12 at B = 2, 0.047-0.055 at B = 32: it falls with the batch too)
    assert max(res[2]["lstm_bias_rel_worst"]) <= 0.2 and max(res[32]["lstm_bias_rel_worst"]) <= 0.1, res
    assert max(res[32]["lstm_bias_rel_worst"]) <= 0.75 * max(res[2]["lstm_bias_rel_worst"]), res
    ratio = float(np.median(np.array(res[32]["noise"]) / np.maximum(np.array(res[2]["noise"]), 1e-12)))
    assert ratio <= 0.7, (ratio, res)
    assert max(res[32]["bias"]) <= 0.5 and max(res[2]["bias"]) <= 0.5, res


def test_bf16_full_length_clip_against_reference_golden():
    g = load_golden("dccrn_default_C_sisnr_full")
    m = make_model((32, 64, 128, 256, 256, 256), 256, "C", "SI-SNR", dtype="bf16")
    m.train()
    x, y = make_signals(1, 48000)
    with torch.no_grad():
        o_r, o_i, wav = m(x.cuda(), y.cuda())
        lossv = float(m.loss(wav, y.cuda()))
    rec = dict(out_wav_rel_l2=rel_l2(wav, g["g/out_wav"]), out_wav_rel_max=rel_err(wav, g["g/out_wav"]),
               out_real_rel_l2=rel_l2(o_r, g["g/out_real"]), out_real_rel_max=rel_err(o_r, g["g/out_real"]),
               loss=lossv, loss_ref=float(g["g/loss"]), B=1, L=48000)
    rec["loss_rel"] = abs(rec["loss"] - rec["loss_ref"]) / max(1.0, abs(rec["loss_ref"]))
    _bf16_record("dccrn_default_C_sisnr_full", rec)
    assert rec["out_wav_rel_l2"] < BF16_OUT_L2 and rec["out_wav_rel_max"] < BF16_OUT_MAX and rec["loss_rel"] < BF16_LOSS, rec


def test_bf16_crn_step_against_reference_golden():
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    g = load_golden("crn_default_E_mse")
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    kn = (32, 64, 128, 256, 256, 256)
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.skip_type, cfg.act_dtype = list(kn), "E", "MSE", False, True, "bf16"
    try:
        m = models.CRN(rnn_units=256, rnn_input_size=512, masking_mode="E")
    finally:
        cfg.act_dtype = "fp32"
    fill_state_dict_(m)
    m = m.to("cuda").train()
    x, y = make_signals(B, L)
    est_mags, target_mags, wav = m(x.cuda(), y.cuda())
    lossv = m.loss(wav, y.cuda())
    lossv.backward()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    rec = dict(out_wav_rel_l2=rel_l2(wav, g["g/out_wav"]), out_wav_rel_max=rel_err(wav, g["g/out_wav"]),
               est_mags_rel_l2=rel_l2(est_mags, g["g/est_mags"]), loss=float(lossv), loss_ref=float(g["g/loss"]), B=B, L=L,
               **_grad_report(grads, g, 53))
    rec["loss_rel"] = abs(rec["loss"] - rec["loss_ref"]) / max(1e-30, abs(rec["loss_ref"]))     # MSE ~ 1e-3: relative to itself
    _bf16_record("crn_default_E_mse", rec)
    assert rec["out_wav_rel_l2"] < BF16_OUT_L2 and rec["out_wav_rel_max"] < BF16_OUT_MAX and rec["est_mags_rel_l2"] < BF16_OUT_L2, rec
    assert rec["loss_rel"] < 2 * BF16_LOSS and rec["grad_rel_l2_median"] < BF16_GRAD_L2 and rec["grad_rel_l2_worst"] < BF16_GRAD_WORST and rec["slope_rel_worst"] < BF16_SLOPE_WORST, rec
    assert rec["grad_cosine"] > BF16_GRAD_COS, rec


@pytest.mark.parametrize("row_block", [False, True])
def test_bf16_fullsubnet_step_against_reference_golden(row_block, monkeypatch):
    """row_block=True: the planner's row threshold is lowered (knob LSTM_ROWS_MIN=64; the golden has B = 2 -> 514 sub-band rows, the bench
    B = 64 -> 16 448) so that the sub-band model runs on the row-block kernels of lstm_rows.hip - fused input projections, bf16 gate slabs,
    the 2-output head's gradient formed in the backward kernel - i.e. the kernels the FullSubNet bench line times, against the reference."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models, tools_for_model as tools
    if row_block:
        knobs.set("LSTM_ROWS_MIN", "64")
    g = load_golden("fsn_default_mse")
    B, L = int(g["g/meta/B"]), int(g["g/meta/L"])
    cfg.loss, cfg.act_dtype = "MSE", "bf16"
    try:
        m = models.FullSubNet(fb_model_hidden_size=512, sb_model_hidden_size=384)
    finally:
        cfg.act_dtype = "fp32"
    fill_state_dict_(m)
    m = m.to("cuda").train()
    m.dropout_keep = 1.0
    x, y = make_signals(B, L)
    nc, cc = tools.stft(x.cuda()), tools.stft(y.cuda())
    noisy_mag, _ = tools.mag_phase(nc)
    cirm = tools.build_complex_ideal_ratio_mask(nc, cc)
    crm = m(noisy_mag)
    lossv = m.loss(cirm, crm)
    lossv.backward()
    grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
    vals = [rel_l2(grads[k], v) for k, v in sub(g, "g/grad").items()] + \
           [rel_l2(grads[k].reshape(-1)[::211], v) for k, v in sub(g, "g/grad_samp").items()]
    rec = dict(crm_rel_l2=rel_l2(crm, g["g/crm"]), crm_rel_max=rel_err(crm, g["g/crm"]), loss=float(lossv), loss_ref=float(g["g/loss"]),
               grad_rel_l2_worst=float(max(vals)), grad_rel_l2_median=float(np.median(vals)), B=B, L=L)
    rec["loss_rel"] = abs(rec["loss"] - rec["loss_ref"]) / abs(rec["loss_ref"])
    plan = next(v for k, v in m._runtimes.items() if k[0] == "fsn")[0]
    assert (plan.buffer("sb_model.l0.gates")[3] == 1) == row_block         # bf16 gate slabs <=> row-block kernels
    _bf16_record("fsn_default_mse" + ("_rowblock" if row_block else ""), rec)
    assert rec["crm_rel_l2"] < BF16_OUT_L2 and rec["crm_rel_max"] < BF16_OUT_MAX and rec["loss_rel"] < BF16_LOSS, rec
    assert rec["grad_rel_l2_median"] < BF16_GRAD_L2 and rec["grad_rel_l2_worst"] < BF16_GRAD_WORST, rec


def test_bf16_full_shape_properties_at_bench_size():
    """BASELINE configs[1] shape (B = 32 x 3 s clips, bf16, mask C): size-independent properties of the path the bench times."""
    from sefd_amd.optim import Adam
    m = make_model((32, 64, 128, 256, 256, 256), 256, "C", "SI-SNR", dtype="bf16")
    m.train()
    B, L = 32, 48000
    g = torch.Generator().manual_seed(1234)
    clean = 0.1 * torch.randn(B, L, generator=g)
    noisy = clean + 0.05 * torch.randn(B, L, generator=g)
    x, y = noisy.cuda(), clean.cuda()
    with torch.no_grad():
        o_r, o_i, wav = m(x, y)
    assert bool(torch.isfinite(wav).all()) and bool(torch.isfinite(o_r).all()) and bool(torch.isfinite(o_i).all())
    assert float(wav.abs().max()) <= 1.0                                  # clamp (models.py:282)
    assert float(o_r[:, 0].abs().max()) == 0.0 and float(o_i[:, 0].abs().max()) == 0.0      # zero DC row
    # mask C is linear in the noisy spectrum for a FIXED mask: eval-mode batch independence at the full length
    m.eval()
    with torch.no_grad():
        full = m(x[:4])[2]
        one = m(x[2:3])[2]
    assert rel_err(full[2:3], one) < 1e-5
    # the fused train step moves the loss down on a fixed batch and keeps every parameter finite
    m.train()
    opt = Adam(m.parameters(), lr=1e-3)
    losses = [float(m.train_step(x, y, opt)) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert bool(torch.isfinite(m._flat_param).all())


# ------------------------------------------------------------------------------------------------ every benched configuration at its bench size
def _bench_batch(B, L=48000, seed=1234):
    g = torch.Generator().manual_seed(seed)
    clean = 0.1 * torch.randn(B, L, generator=g)
    return (clean + 0.05 * torch.randn(B, L, generator=g)).cuda(), clean.cuda()


def test_bf16_fullsubnet_at_bench_size():
    """BASELINE configs[2]: FullSubNet, B = 64 x 3 s, bf16, dropout 0.8 active - finite, loss decreasing over the fused steps."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    cfg.loss, cfg.act_dtype = "MSE", "bf16"
    try:
        torch.manual_seed(0)
        m = models.FullSubNet().to("cuda").train()
    finally:
        cfg.act_dtype = "fp32"
    x, y = _bench_batch(64)
    opt = Adam(m.parameters(), lr=1e-3)
    losses = [float(m.train_step(x, y, opt)) for _ in range(4)]
    assert all(np.isfinite(losses)) and min(losses[1:]) < losses[0], losses
    assert bool(torch.isfinite(m._flat_param).all()) and bool(torch.isfinite(m._flat_grad).all())
    plan = next(v for k, v in m._runtimes.items() if k[0] == "fsn")[0]
    assert plan.buffer("sb_model.l0.gates")[3] == 1                          # the row-block kernels ran


def test_bf16_dccrn_large_at_bench_size():
    """BASELINE configs[4] per-GPU shard at its real size (batch 512 over 8 GPUs = B 64 per GPU): DCCRN-large, cluster LSTM kernels (H = 256)."""
    from sefd_amd.optim import Adam
    m = make_model((64, 128, 256, 512, 512, 512), 512, "C", "SI-SNR", dtype="bf16")
    m.train()
    x, y = _bench_batch(64)
    opt = Adam(m.parameters(), lr=1e-3)
    losses = [float(m.train_step(x, y, opt)) for _ in range(3)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert bool(torch.isfinite(m._flat_param).all())


@pytest.mark.parametrize("perceptual", ["PMSQE", "LMS"])
def test_bf16_dccrn_perceptual_at_bench_size(perceptual):
    """BASELINE configs[3] per-GPU shard (B = 32): loss = (SI-SNR + perceptual) / 2 through the fused step, reference-literal PMSQE."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg
    from sefd_amd.optim import Adam
    assert cfg.pmsqe_power is False
    m = make_model((32, 64, 128, 256, 256, 256), 256, "C", "SI-SNR", dtype="bf16")
    m.train()
    x, y = _bench_batch(32)
    opt = Adam(m.parameters(), lr=1e-3)
    losses = [float(m.train_step(x, y, opt, perceptual=perceptual)) for _ in range(3)]
    main, perc = m._last_loss_parts
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert np.isfinite(float(main)) and np.isfinite(float(perc)) and abs((float(main) + float(perc)) / 2 - losses[-1]) < 1e-4 * abs(losses[-1]) + 1e-5
    assert bool(torch.isfinite(m._flat_param).all())


# ------------------------------------------------------------------------------------------------ the two-stream schedule is only a schedule
@pytest.mark.parametrize("which", ["dccrn", "fullsubnet", "fullsubnet_hoisted"])
def test_two_stream_schedule_equals_program_order(which, monkeypatch):
    """Every kernel is deterministic, so the gradients of one fused step must be BIT-identical whether the phase runs in its two-lane schedule
    (weight gradients, folds, early UNPACK, chunked LSTM forward on the second stream; FullSubNet: held weight gradients) or in program order on
    one stream (knob NO_OVERLAP=1).  A missing dependency between the lanes shows up here as a mismatch.  Sizes large enough that kernels of the
    two streams really overlap (DCCRN B = 8 x 3 s, FullSubNet B = 16 x 3 s); lr = 0 keeps the parameters of the two runs equal.
    (Round 3: this test is what exposed the run-to-run differences of FullSubNet's row-block forward recurrence - lstm_rows.hip, note above
    mfma_settle - which had nothing to do with the lanes: two runs of the SAME schedule differed.)
    "fullsubnet_hoisted" (knob LSTM_XFUSE=0): the row-block forward kernel WITHOUT the fused input projection (XF = 0: pre-activations of the
    hoisted GEMM ride in the accumulators' initial value) and the row-block backward kernels behind it - run-to-run bit-reproducibility of the
    variant the default configuration does not launch."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    if which == "dccrn":
        m = make_model((32, 64, 128, 256, 256, 256), 256, "C", "SI-SNR", dtype="bf16")
        m.train()
        x, y = _bench_batch(8)
    else:
        cfg.loss, cfg.act_dtype = "MSE", "bf16"
        if which == "fullsubnet_hoisted":
            knobs.set("LSTM_XFUSE", "0")     # read when the plan is built (first step below)
        try:
            torch.manual_seed(0)
            m = models.FullSubNet().to("cuda").train()
        finally:
            cfg.act_dtype = "fp32"
        x, y = _bench_batch(16 if which == "fullsubnet" else 8)
        which = "fullsubnet"
    opt = Adam(m.parameters(), lr=0.0)
    grads, losses = [], []
    for rep, single in enumerate((False, True, False)):
        if single:
            knobs.set("NO_OVERLAP", "1")
        else:
            knobs.unset("NO_OVERLAP")
        if which == "fullsubnet" and rep > 0:              # same dropout masks: the step counter behind the mask hash goes back by one
            plan, ar = next(v for k, v in m._runtimes.items() if k[0] == "fsn")
            plan.view(ar, "io.seed").view(torch.int32)[:1].sub_(1)
        losses.append(float(m.train_step(x, y, opt)))
        torch.cuda.synchronize()
        grads.append(m._flat_grad.clone())
    knobs.unset("NO_OVERLAP")
    assert bool(torch.isfinite(grads[0]).all()) and float(grads[0].abs().max()) > 0
    assert losses[0] == losses[1] == losses[2], losses
    assert torch.equal(grads[0], grads[1]), float((grads[0] - grads[1]).abs().max())
    assert torch.equal(grads[0], grads[2])


@pytest.mark.parametrize("which", ["dccrn", "fullsubnet"])
def test_graph_replay_equals_launch_sequence(which):
    """Knob GRAPH=1 (api.hip plan_run_graph, round 6): a whole phase is captured ONCE from the executor's own multi-stream launch sequence into a hipGraph
    (third run with the same arenas and stream) and replayed afterwards.  Same kernels, same dependencies: loss and gradients of the captured run and of
    the replays are BIT-identical to the launch sequence.  (Measured: no faster - profiles/r06_tuning_notes.md section 12 - hence opt-in.)"""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    if which == "dccrn":
        m = make_model((32, 64, 128, 256, 256, 256), 256, "C", "SI-SNR", dtype="bf16")
        m.train()
        x, y = _bench_batch(8)
    else:
        cfg.loss, cfg.act_dtype = "MSE", "bf16"
        try:
            torch.manual_seed(0)
            m = models.FullSubNet().to("cuda").train()
        finally:
            cfg.act_dtype = "fp32"
        x, y = _bench_batch(16)
    opt = Adam(m.parameters(), lr=0.0)
    out = []
    for rep in range(6):                                   # 0: launch sequence ; 1, 2: warm-up runs of the graph path ; 3: capture + first launch ; 4, 5: replays
        if rep == 1:
            knobs.set("GRAPH", "1")
        if which == "fullsubnet" and rep > 0:              # same dropout masks: the step counter behind the mask hash goes back by one
            plan, ar = next(v for k, v in m._runtimes.items() if k[0] == "fsn")
            plan.view(ar, "io.seed").view(torch.int32)[:1].sub_(1)
        loss = float(m.train_step(x, y, opt))
        torch.cuda.synchronize()
        out.append((loss, m._flat_grad.clone()))
    knobs.unset("GRAPH")
    assert bool(torch.isfinite(out[0][1]).all()) and float(out[0][1].abs().max()) > 0
    for rep in range(1, 6):
        assert out[rep][0] == out[0][0], (rep, out[rep][0], out[0][0])
        assert torch.equal(out[rep][1], out[0][1]), (rep, float((out[rep][1] - out[0][1]).abs().max()))


def test_paired_subband_layers_equal_two_launches(monkeypatch):
    """FullSubNet's two sub-band LSTM layers run as ONE launch of (layer, time chunk, row block) jobs (lstm_rows.hip lstm_fwd_rows_pair_kernel: a
    job starts behind the flags of the jobs it reads from, on whatever CU is free), and the backward recurrences as (time chunk, row block) jobs
    that hand the recurrent gradient and the cell-state carry through memory (lstm_bwd_rows_jobs_kernel) - schedules, not different computations:
    loss and gradients of a fused step are BIT-identical to whole-sequence workgroups in one launch per layer (knob ROWS_PAIR=0,
    knob ROWS_BWD_CHUNKS=1).  B = 16: 4 112 rows = 86 blocks per layer, every job resident at once, so the waits are real; the bench size
    (B = 64: 343 blocks on 256 CUs) is the many-rounds case."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    cfg.loss, cfg.act_dtype = "MSE", "bf16"
    try:
        torch.manual_seed(0)
        m = models.FullSubNet().to("cuda").train()
    finally:
        cfg.act_dtype = "fp32"
    opt = Adam(m.parameters(), lr=0.0)
    for B in (16, 64):
        x, y = _bench_batch(B)
        grads, losses = [], []
        for rep, pair in enumerate((True, False, True)):
            if pair:
                knobs.unset("ROWS_PAIR")
                if B == 16:
                    knobs.set("ROWS_BWD_CHUNKS", "3")     # B = 64 (343 blocks > 256 CUs) chunks the backward by default
                else:
                    knobs.unset("ROWS_BWD_CHUNKS")
            else:
                knobs.set("ROWS_PAIR", "0")
                knobs.set("ROWS_BWD_CHUNKS", "1")
            if rep > 0:                                        # same dropout masks: the step counter behind the mask hash goes back by one
                plan, ar = next(v for k, v in m._runtimes.items() if k[0] == "fsn" and k[1] == B)
                plan.view(ar, "io.seed").view(torch.int32)[:1].sub_(1)
            losses.append(float(m.train_step(x, y, opt)))
            torch.cuda.synchronize()
            grads.append(m._flat_grad.clone())
        knobs.unset("ROWS_PAIR")
        knobs.unset("ROWS_BWD_CHUNKS")
        assert bool(torch.isfinite(grads[0]).all()) and float(grads[0].abs().max()) > 0
        assert losses[0] == losses[1] == losses[2], (B, losses)
        assert torch.equal(grads[0], grads[1]), (B, float((grads[0] - grads[1]).abs().max()))
        assert torch.equal(grads[0], grads[2])
        assert next(v for k, v in m._runtimes.items() if k[0] == "fsn" and k[1] == B)[0].status() == 0


def test_plan_status_word_guards_adam_and_checkpoint(tmp_path):
    """A kernel that gives up (the cluster LSTM's bounded hand-over waits) sets its PLAN's host-mapped status word.  From then on: the
    guarded Adam leaves parameters and moments untouched (no garbage step), save_checkpoint refuses to write, the plan's next run raises -
    and another plan of the same model (other batch size) is not affected; clearing the word re-arms the plan."""
    from sefd_amd import train_interface
    from sefd_amd.optim import Adam
    m = make_model((16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR")
    m.train()
    x, y = make_signals(2, 4000)
    x, y = x.cuda(), y.cuda()
    opt = Adam(m.parameters(), lr=1e-3)
    m.train_step(x, y, opt)
    plan = m._status_plan
    assert plan.status() == 0
    plan.status_set()                                                 # what report_timeout() does from the device
    before, mom = m._flat_param.clone(), opt._m.clone()
    m._flat_grad.fill_(1.0)
    opt.step_flat()
    torch.cuda.synchronize()
    assert torch.equal(m._flat_param, before) and torch.equal(opt._m, mom)
    with pytest.raises(RuntimeError, match="checkpoint not written"):
        train_interface.save_checkpoint(str(tmp_path / "c.pt"), m, opt, 1)
    assert not (tmp_path / "c.pt").exists()
    with pytest.raises(RuntimeError, match="gave up waiting"):
        m.train_step(x, y, opt)
    x3, y3 = make_signals(3, 4000)
    m.train_step(x3.cuda(), y3.cuda(), opt)                           # another plan of the same model: its own word
    assert m._status_plan is not plan and m._status_plan.status() == 0
    assert plan.status(clear=True) == 1 and plan.status() == 0
    m.train_step(x, y, opt)
    torch.cuda.synchronize()
    assert not torch.equal(m._flat_param, before)


def test_status_poison_reaches_every_replica_through_the_gradient(tmp_path):
    """ADVICE r4 (guarded Adam under data parallelism): a rank whose plan gave up replaces one element of the last gradient bucket by NaN before
    the exchange (Plan.status_poison); a sum all-reduce hands that NaN to every rank, whose Adam (skip_if_nan) then skips the SAME step, and
    every rank's checkpoint check raises.  Emulated on one GPU: the 'other rank' is a model whose own status word is clean and whose gradient
    received the poisoned element by addition, as the all-reduce would deliver it."""
    from sefd_amd import train_interface
    from sefd_amd.optim import Adam
    bad, good = (make_model((16, 32, 32, 64, 64, 64), 128, "E", "SI-SNR") for _ in range(2))
    x, y = make_signals(2, 4000)
    x, y = x.cuda(), y.cuda()
    opts = [Adam(m.parameters(), lr=1e-3) for m in (bad, good)]
    for m, o in zip((bad, good), opts):
        m.train()
        m.train_step(x, y, o)
    stream = torch.cuda.current_stream().cuda_stream
    good._status_plan.status_poison(good._flat_grad[0:1], stream)        # clean word: the element stays a number
    torch.cuda.synchronize()
    assert bool(torch.isfinite(good._flat_grad[0]))
    bad._status_plan.status_set()
    bad._status_plan.status_poison(bad._flat_grad[0:1], stream)
    torch.cuda.synchronize()
    assert bool(torch.isnan(bad._flat_grad[0])) and bool(torch.isfinite(bad._flat_grad[1:]).all())
    good._flat_grad += bad._flat_grad                                    # the sum all-reduce, as the clean rank sees it
    before, mom = good._flat_param.clone(), opts[1]._m.clone()
    opts[1].nan_guard = good._dp_guard = good._flat_grad[0:1]
    opts[1].step_flat()
    torch.cuda.synchronize()
    assert torch.equal(good._flat_param, before) and torch.equal(opts[1]._m, mom)          # skipped on the clean rank too
    with pytest.raises(RuntimeError, match="another rank"):
        train_interface.save_checkpoint(str(tmp_path / "c.pt"), good, opts[1], 1)
    good._flat_grad[0] = 0.0
    opts[1].nan_guard = good._flat_grad[0:1]
    opts[1].step_flat()
    torch.cuda.synchronize()
    assert not torch.equal(good._flat_param, before)
    bad._status_plan.status(clear=True)


@pytest.mark.parametrize("which,loss_kind", [("DCCRN", "MSE"), ("DCCRN", "SI-SNR"), ("CRN", "MSE"), ("CRN", "SDR")])
def test_direct_mapping_fused_step_equals_autograd_route(which, loss_kind):
    """VERDICT r4 missing item 4: dccrn_direct_train / crn_direct_train (trainer.py:121-181) on the fused `train_step` - the spectral losses and
    their gradients go straight into the plan's spectrum-gradient inputs.  Must equal the literal loop (forward, loss on the spectra,
    loss.backward(), optimizer.step()) over the same kernels: loss per step and every parameter after two steps."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models, trainer
    from sefd_amd.optim import Adam
    kn = (16, 32, 32, 64, 64, 64)
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.perceptual, cfg.skip_type, cfg.act_dtype = list(kn), "Direct(None make)", loss_kind, False, True, "fp32"
    x, y = make_signals(2, 4000)
    batches = [(x, y), (0.5 * y + 0.5 * x, y)]
    res = []
    for fused in (False, True):
        if which == "DCCRN":
            m = models.DCCRN(rnn_units=128, masking_mode="Direct(None make)")
        else:
            m = models.CRN(rnn_units=128, rnn_input_size=128, masking_mode="Direct(None make)")
        fill_state_dict_(m)
        m = m.to("cuda").train()
        opt = Adam(m.parameters(), lr=1e-3)
        fn = trainer.dccrn_direct_train if which == "DCCRN" else trainer.crn_direct_train
        losses = []
        for b in batches:
            if fused:
                losses.append(float(fn(m, opt, [b], "cuda")))
            else:                                    # the reference's loop body, spelled out (the trainer would pick the fused path for this optimizer)
                outs = m(b[0].cuda(), b[1].cuda())
                if which == "DCCRN":
                    lossv = (m.loss(outs[0], outs[1]) + m.loss(outs[2], outs[3])) / 2
                else:
                    lossv = m.loss(outs[0], outs[1])
                opt.zero_grad()
                lossv.backward()
                opt.step()
                losses.append(float(lossv))
        res.append((losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}))
    (la, pa), (lb, pb) = res
    for a, b in zip(la, lb):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (la, lb)
    for k in pa:
        if pa[k].dtype.is_floating_point:
            assert float((pa[k] - pb[k]).abs().max()) <= 1e-5 * max(1.0, float(pa[k].abs().max())), k
    cfg.masking_mode, cfg.loss = "E", "SI-SNR"
