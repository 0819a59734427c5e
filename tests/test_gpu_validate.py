"""GPU tier: validation / inference path (reference trainer.py:188-483), tools.istft, checkpoint + resume (train_interface.py:101-116,
204-228) on the HIP library."""
import os

import numpy as np
import pytest
import torch

from oracle.weights import fill_state_dict_, test_signals as make_signals
from util import rel_err

pytestmark = pytest.mark.gpu
SMALL = (16, 32, 32, 64, 64, 64)


def _cfg(**kw):
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg
    base = dict(dccrn_kernel_num=list(SMALL), masking_mode="E", loss="SI-SNR", perceptual=False, lstm="complex", skip_type=True,
                act_dtype="fp32", model="DCCRN", chkpt_model=None)
    base.update(kw)
    for k, v in base.items():
        setattr(cfg, k, v)
    return cfg


def test_tools_istft_matches_torch_istft():
    """tools_for_model.py:651-680 on the HIP inverse-FFT + overlap-add kernels vs torch.istft (CPU), incl. the real-pair input of
    trainer.py:344-345 that the reference itself cannot run on torch >= 2."""
    _cfg()
    from sefd_amd import tools_for_model as tools
    x, _ = make_signals(3, 6000)
    w = torch.hann_window(400)
    S = torch.stft(x, 512, 300, 400, window=w, return_complex=True)
    S2 = S * (1.0 + 0.5 * torch.sin(torch.arange(S.shape[-1]).float()))[None, None]
    for sp in (S, S2):
        ref = torch.istft(sp, 512, 300, 400, window=w, length=6000)
        out = tools.istft(torch.view_as_real(sp).cuda(), length=6000)
        assert rel_err(out, ref) < 1e-4
        out2 = tools.istft(sp.cuda(), length=6000)
        assert torch.equal(out, out2)
    # round trip through the package's own stft
    y = tools.istft(tools.stft(x.cuda()), length=6000)
    assert rel_err(y, x) < 1e-4


def test_fullsubnet_validate_runs_the_enhancement_path(tmp_path):
    cfg = _cfg(loss="MSE", model="FullSubNet")
    from sefd_amd import models, trainer
    m = models.FullSubNet(fb_model_hidden_size=128, sb_model_hidden_size=64)
    fill_state_dict_(m)
    m = m.to("cuda").train()
    x, y = make_signals(2, 6000)
    seen = {}

    def pesq(est, clean):
        seen["shape"] = est.shape
        seen["est"] = est.copy()
        return np.full(len(est), 1.5)

    vloss, p, s = trainer.fullsubnet_validate(m, [(x, y)], None, str(tmp_path), 1, "cuda", scorers=(pesq, lambda e, c: np.full(len(e), 0.5)))
    assert m.training and seen["shape"] == (2, 6000) and np.isfinite(seen["est"]).all() and np.isfinite(float(vloss))
    assert abs(p - 1.5) < 1e-9 and abs(s - 0.5) < 1e-9
    # the enhanced waveform is what the oracle's restatement of trainer.py:331-345 gives for the same cRM
    from sefd_amd import tools_for_model as tools
    from oracle.fullsubnet import decompress_cirm
    m.eval()
    with torch.no_grad():
        nc = tools.stft(x.cuda())
        crm = m(tools.mag_phase(nc)[0]).cpu()
    w = torch.hann_window(400)
    nc_ref = torch.stft(x, 512, 300, 400, window=w, return_complex=True)
    d = decompress_cirm(crm)
    enh = torch.complex(d[..., 0] * nc_ref.real - d[..., 1] * nc_ref.imag, d[..., 1] * nc_ref.real + d[..., 0] * nc_ref.imag)
    ref = torch.istft(enh, 512, 300, 400, window=w, length=6000)
    assert rel_err(seen["est"], ref) < 1e-3


def test_perceptual_and_direct_validate(tmp_path):
    cfg = _cfg(perceptual="LMS")
    from sefd_amd import models, trainer
    m = models.DCCRN(rnn_units=128, masking_mode="E")
    fill_state_dict_(m)
    m = m.to("cuda").train()
    x, y = make_signals(2, 4000)
    try:
        out = trainer.model_perceptual_validate(m, [(x, y), (x * 0.5, y * 0.5)], None, str(tmp_path), 2, "cuda", scorers=None)
    finally:
        cfg.perceptual = False
    loss, main, perc, p, s = out
    assert abs(float(loss) - 0.5 * (float(main) + float(perc))) < 1e-5 * abs(float(loss)) and p != p and s != s
    # same numbers as the training-path pieces in eval mode
    m.eval()
    with torch.no_grad():
        cfg.perceptual = "LMS"
        try:
            vals = []
            for sc in (1.0, 0.5):
                r, i, w = m(x.cuda() * sc)
                vals.append((float(m.loss(w, y.cuda() * sc)), float(m.loss(w, y.cuda() * sc, r, i, perceptual=True))))
        finally:
            cfg.perceptual = False
    assert abs(float(main) - 0.5 * (vals[0][0] + vals[1][0])) < 1e-5 * abs(float(main))
    assert abs(float(perc) - 0.5 * (vals[0][1] + vals[1][1])) < 1e-5 * abs(float(perc))
    # spectral mapping validate functions
    _cfg(masking_mode="Direct(None make)", loss="MSE")
    md = models.DCCRN(rnn_units=128, masking_mode="Direct(None make)")
    fill_state_dict_(md)
    md = md.to("cuda")
    vl, _, _ = trainer.dccrn_direct_validate(md, [(x, y)], None, str(tmp_path), 1, "cuda", scorers=None)
    md.eval()
    with torch.no_grad():
        o_r, t_r, o_i, t_i, _ = md(x.cuda(), y.cuda())
        ref = (md.loss(o_r, t_r) + md.loss(o_i, t_i)) / 2
    assert abs(float(vl) - float(ref)) < 1e-6 * abs(float(ref))
    mc = models.CRN(rnn_units=128, rnn_input_size=128, masking_mode="Direct(None make)")
    fill_state_dict_(mc)
    mc = mc.to("cuda")
    vl, _, _ = trainer.crn_direct_validate(mc, [(x, y)], None, str(tmp_path), 1, "cuda", scorers=None)
    assert np.isfinite(float(vl))
    _cfg()


def test_checkpoint_round_trip_continues_bit_identically(tmp_path):
    """save {'model','optimizer','epoch'} -> fresh model + Adam -> load (reference order: construct, load_state_dict x2,
    train) -> the next fused step equals the uninterrupted run bit for bit; the optimizer state also loads into torch.optim.Adam."""
    _cfg(masking_mode="C")
    from sefd_amd import models, train_interface as ti
    from sefd_amd.optim import Adam
    x, y = make_signals(3, 4000)
    x, y = x.cuda(), y.cuda()

    def fresh():
        m = models.DCCRN(rnn_units=128, masking_mode="C")
        fill_state_dict_(m)
        return m.to("cuda").train()

    a = fresh()
    oa = Adam(a.parameters(), lr=1e-3)
    for _ in range(2):
        a.train_step(x, y, oa)
    path = str(tmp_path / "chkpt_2.pt")
    ti.save_checkpoint(path, a, oa, 2)
    la = float(a.train_step(x, y, oa))                       # step 3 of the uninterrupted run
    pa = a._flat_param.clone()
    b = fresh()
    ob = Adam(b.parameters(), lr=1e-3)
    assert ti.load_checkpoint(path, b, ob, map_location="cuda") == 3
    lb = float(b.train_step(x, y, ob))
    assert la == lb and torch.equal(pa, b._flat_param)
    assert torch.equal(a.state_dict()["encoder.3.1.running_var"], b.state_dict()["encoder.3.1.running_var"])
    assert int(b.state_dict()["encoder.0.1.num_batches_tracked"]) == 3
    # interchange with torch.optim.Adam (the reference's optimizer class)
    ck = torch.load(path)
    c = fresh()
    oc = torch.optim.Adam(c.parameters(), lr=1e-3)
    c.load_state_dict(ck["model"])
    oc.load_state_dict(ck["optimizer"])
    _, _, wav = c(x, y)
    lc = c.loss(wav, y)
    oc.zero_grad()
    lc.backward()
    oc.step()
    assert abs(float(lc) - la) < 1e-5 * abs(la)
    assert rel_err(c._flat_param, pa) < 1e-5
    _cfg()


def test_train_interface_driver_writes_the_reference_artifacts_and_resumes(tmp_path):
    cfg = _cfg(masking_mode="C", job_dir=str(tmp_path / "models") + "/", logs_dir=str(tmp_path / "logs") + "/", expr_num="T1", batch=2)
    from sefd_amd import models, train_interface as ti
    from sefd_amd.optim import Adam
    x, y = make_signals(4, 4000)
    train = [(x[:2], y[:2]), (x[2:], y[2:])]
    valid = [(x[1:3], y[1:3])]
    m = models.DCCRN(rnn_units=128, masking_mode="C")
    fill_state_dict_(m)
    m = m.to("cuda")
    opt = Adam(m.parameters(), lr=1e-3)
    fake = (lambda e, c: np.full(len(e), 2.0), lambda e, c: np.full(len(e), 0.8))
    m, opt, mse, d = ti.run(train, valid, model=m, optimizer=opt, DEVICE="cuda", scorers=fake, max_epochs=2)
    files = sorted(os.listdir(d))
    assert files == ["Epoch_1_SCORES", "Epoch_2_SCORES", "chkpt_1.pt", "chkpt_2.pt", "chkpt_opt.pt", "log.txt", "mse_vali_total.npy"], files
    log = open(os.path.join(d, "log.txt")).read()
    assert "total params" in log and "Epoch [2] | T " in log and "V PESQ: 2.000000 | STOI: 0.800000" in log
    assert np.load(os.path.join(d, "mse_vali_total.npy")).shape == (2,) and mse[1] != 0
    # resume from epoch 2 for one more epoch (train_interface.py:101-116)
    cfg.chkpt_model, cfg.chkpt = os.path.basename(d), "2"
    cfg.chkpt_path = cfg.job_dir + cfg.chkpt_model + '/chkpt_' + cfg.chkpt + '.pt'
    try:
        m2 = models.DCCRN(rnn_units=128, masking_mode="C").to("cuda")
        o2 = Adam(m2.parameters(), lr=1e-3)
        m2, o2, mse2, d2 = ti.run(train, valid, model=m2, optimizer=o2, DEVICE="cuda", scorers=fake, max_epochs=3)
    finally:
        cfg.chkpt_model = None
        del cfg.chkpt_path
    assert d2 == d and os.path.exists(os.path.join(d, "chkpt_3.pt")) and o2._step == 6 and mse2[0] == mse[0] and mse2[2] != 0
    _cfg(job_dir='./models/', logs_dir='./logs/', expr_num='EXPERIMENT_NUMBER', batch=10)


@pytest.mark.gpu
def test_on_gpu_snr_mixing_against_the_offline_script():
    """generate_noisy_data.py:46-67 on the GPU for a batch (sefd_mix_snr): same samples as the offline numpy path (restated in
    oracle/mixing.py and pinned to the reference function on CPU), including the int16 truncation; one LSB of slack on a handful of
    samples (one-pass variance in the kernel, two-pass in numpy)."""
    import numpy as np
    import sefd_amd  # noqa: F401
    from sefd_amd.dataloader import mix_snr
    from oracle.mixing import generate_noisy_wav
    rng = np.random.default_rng(3)
    B, L = 5, 48000
    speech = (rng.standard_normal((B, L)) * 0.08 + 0.003).astype(np.float32)
    noise = (rng.standard_normal(400000) * 0.2 - 0.01).astype(np.float32)
    start = rng.integers(0, noise.size - L, B)
    snr = np.array([0, 5, 10, -5, 15], np.float32)
    got = mix_snr(torch.from_numpy(speech).cuda(), torch.from_numpy(noise).cuda(), start, snr, quantize=True).cpu().numpy()
    for b in range(B):
        want = generate_noisy_wav(speech[b].astype(np.float64), noise.astype(np.float64), float(snr[b]), int(start[b])).astype(np.float64) / 32768
        d = np.abs(got[b].astype(np.float64) - want) * 32768
        assert d.max() <= 1.0 + 1e-6 and (d > 0.5).mean() < 1e-4, (b, d.max(), (d > 0.5).mean())
    raw = mix_snr(torch.from_numpy(speech).cuda(), torch.from_numpy(noise).cuda(), start, snr, quantize=False).cpu().numpy()
    want = generate_noisy_wav(speech[1].astype(np.float64), noise.astype(np.float64), 5.0, int(start[1]), quantize=False)
    assert np.abs(raw[1] - want).max() < 1e-6
    with pytest.raises(RuntimeError):
        mix_snr(torch.from_numpy(speech), torch.from_numpy(noise), start, snr)
