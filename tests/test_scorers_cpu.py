"""CPU tier: the C++ PESQ scorer (csrc_host/pesq.cpp) against MOS-LQO goldens captured from the reference's PESQ.so
(tests/golden/make_pesq_golden.py), and the C++ STOI scorer (csrc_host/scorers.cpp through include/sefd_scorers.h) against the oracle's independent numpy /
scipy statement of the published algorithm (oracle/stoi.py; parity with pystoi itself is unpinned - not installed, not vendored)."""
import ctypes as C
import os

import numpy as np
import pytest

import sefd_amd  # noqa: F401
from sefd_amd import tools_for_estimate as te
from oracle import stoi as so


@pytest.fixture(scope="module", autouse=True)
def _built():
    te.build()


def speechlike(B, n, seed=0):
    """Noise / tone carriers with syllabic (3-6 Hz) envelopes and a pause: something STOI's 30-frame envelope correlation can see."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    out = np.zeros((B, n), np.float32)
    for b in range(B):
        s = np.zeros(n)
        for f0 in (300, 700, 1500, 2800, 4200):
            carrier = np.sin(2 * np.pi * f0 * t * (1 + 0.02 * b) + rng.uniform(0, 6)) + 0.3 * rng.standard_normal(n)
            s += carrier * np.clip(np.sin(2 * np.pi * (3 + rng.uniform(0, 3)) * t + rng.uniform(0, 6)), 0, None) ** 2
        s[int(0.4 * n):int(0.5 * n)] *= 5e-4
        out[b] = (0.1 * s).astype(np.float32)
    return out


def pesq_pairs():
    """(name, clean, degraded) float32 pairs at 16 kHz, regenerated from seeds: additive white / coloured noise over 40 dB of SNR, low-pass,
    clipping, a spectral-subtraction-like gain pattern (what an enhancement output looks like), leading / trailing silence, 1 s .. 4 s."""
    from scipy import signal
    out = []
    for k, (n, seed) in enumerate([(48000, 11), (48000, 12), (16000, 13), (64000, 14)]):
        c = speechlike(1, n, seed=seed)[0]
        rng = np.random.default_rng(100 + k)
        for lvl in (0.0007, 0.003, 0.012, 0.05):
            out.append((f"white_n{n}_s{seed}_l{lvl}", c, (c + lvl * rng.standard_normal(n)).astype(np.float32)))
        pink = signal.lfilter([0.05], [1, -0.95], rng.standard_normal(n))
        out.append((f"pink_n{n}_s{seed}", c, (c + 0.01 * pink).astype(np.float32)))
        b, a = signal.butter(4, 2500 / 8000)
        out.append((f"lowpass_n{n}_s{seed}", c, signal.lfilter(b, a, c).astype(np.float32)))
        out.append((f"clip_n{n}_s{seed}", c, np.clip(c, -0.3 * np.abs(c).max(), 0.3 * np.abs(c).max()).astype(np.float32)))
        f, t, Z = signal.stft(c + 0.01 * rng.standard_normal(n), nperseg=512, noverlap=384)
        g = np.maximum(1 - (0.012 / (np.abs(Z) + 1e-9)) ** 2, 0.05)          # over-subtracting Wiener-like gain: musical noise + attenuation
        e = signal.istft(Z * g, nperseg=512, noverlap=384)[1][:n]
        out.append((f"enhanced_n{n}_s{seed}", c, np.pad(e, (0, n - len(e))).astype(np.float32)))
    c = speechlike(1, 48000, seed=21)[0]
    c[:6000] = 0
    c[-9000:] = 0
    rng = np.random.default_rng(5)
    out.append(("silence_edges", c, (c + 0.004 * rng.standard_normal(48000)).astype(np.float32)))
    out.append(("scaled_half", c, (0.5 * c + 0.002 * rng.standard_normal(48000)).astype(np.float32)))
    return out


def test_library_exports_the_declared_symbols():
    L = te.lib()
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "sefd_scorers.h")).read()
    for sym in te.EXPORTED:
        assert hasattr(L, sym) and sym in hdr


def test_resampler_matches_scipy_resample_poly():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(4801)
    h = so.resample_window(10000, 16000)
    assert len(h) == 581                                         # p/q = 5/8, 60 dB, 10 % roll-off
    y = so.resample(x, 10000, 16000)
    assert len(y) == 3001 and abs(float(np.abs(y).max())) < 4.0


@pytest.mark.parametrize("n", [48000, 16000 + 123])
def test_stoi_cpp_equals_oracle(n):
    clean = speechlike(4, n, seed=3)
    rng = np.random.default_rng(4)
    est = (clean + 0.05 * rng.standard_normal(clean.shape).astype(np.float32) * np.array([[0.1], [0.5], [2.0], [8.0]], np.float32))
    got = np.array(te.cal_stoi(est, clean))
    ref = np.array([so.stoi(clean[i], est[i], 16000) for i in range(4)])
    assert np.abs(got - ref).max() < 1e-9
    assert np.all(np.diff(got) < 0) and got[0] > 0.95 and got[-1] < 0.6     # monotone in the noise level
    assert np.allclose(te.cal_stoi(clean, clean), 1.0, atol=1e-9)
    # single-threaded == multi-threaded, 1-D input accepted
    assert np.array_equal(np.array(te.cal_stoi(est, clean, nthreads=1)), got)
    assert abs(te.cal_stoi(est[1], clean[1])[0] - got[1]) < 1e-12


def test_too_short_signal_returns_the_floor_value():
    clean = speechlike(1, 3000)
    assert te.cal_stoi(clean * 0.9, clean)[0] == pytest.approx(1e-5)       # fewer than 30 frames (pystoi warns and returns 1e-5)


def test_default_scorers_of_the_validation_loop_score_for_real(tmp_path):
    """trainer._validate with scorers="default" (what every *_validate and train_interface.run use): a finite PESQ and STOI per utterance and a
    written Epoch_N_SCORES file.  (A round-2 bug returned None here because a missing cal_pesq was swallowed; every other test passes fakes.)"""
    import torch
    from sefd_amd import trainer
    sc = trainer._default_scorers()
    assert sc is not None and sc[0] is te.cal_pesq and sc[1] is te.cal_stoi
    clean = torch.from_numpy(speechlike(2, 48000, seed=7))
    noisy = clean + 0.02 * torch.randn(clean.shape, generator=torch.Generator().manual_seed(1))
    model = torch.nn.Identity()

    def batch(inputs, targets):
        return (torch.mean((inputs - targets) ** 2),), inputs
    loss, pesq, stoi = trainer._validate(model, [(noisy, clean)], None, str(tmp_path), 1, "cpu", batch, 1, "default")
    assert np.isfinite(stoi) and 0.5 < stoi <= 1.0
    assert np.isfinite(pesq) and 1.0 < pesq < 4.7
    lines = open(tmp_path / "Epoch_1_SCORES").read().strip().splitlines()
    assert len(lines) == 2 and all(l.startswith("PESQ ") and " | STOI 0." in l for l in lines)


def test_pesq_cpp_matches_reference_binary_goldens():
    """|MOS-LQO - PESQ.so| <= 1e-3 on all 34 pairs (measured worst 4.5e-5 since round 4's input-filter fix; 0.0033 before): noise over 40 dB of
    SNR, coloured noise, low-pass (a few samples of delay), clipping, an over-subtracting enhancement gain, silent edges, 1-4 s."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pesq_golden.npz"))
    pairs = pesq_pairs()
    assert [p[0] for p in pairs] == [str(n) for n in g["names"]]
    worst = 0.0
    for (name, clean, deg), gold in zip(pairs, g["mos_lqo"]):
        got = te.cal_pesq(deg[None], clean[None], nthreads=1)[0]
        worst = max(worst, abs(got - float(gold)))
        assert abs(got - float(gold)) <= 1e-3, (name, got, float(gold))
    assert 1.0 < float(g["mos_lqo"].min()) < 1.3 and float(g["mos_lqo"].max()) > 3.9      # the goldens span the scale
    # batch == per-utterance, scale invariance, identical signals score the ceiling
    c = np.stack([pairs[0][1], pairs[8][1]]); d = np.stack([pairs[1][2], pairs[9][2]])
    both = te.cal_pesq(d, c)
    assert abs(both[0] - te.cal_pesq(d[0], c[0])[0]) < 1e-12 and abs(te.cal_pesq(0.25 * d, 0.25 * c)[1] - both[1]) < 1e-6
    assert te.cal_pesq(c, c)[0] > 4.6
    # polarity: the model works on power spectra and the delay search on |cross-correlation| - a negated estimate (SI-SNR cannot tell) scores the same;
    # the reference binary does (one of the 26 reference-trained held-out models of round 4 came out negated: 2.7862 there, 2.7849 here)
    assert abs(te.cal_pesq(-d, c)[0] - both[0]) < 2e-3 and abs(te.cal_pesq(-d, c)[1] - both[1]) < 2e-3
    with pytest.raises(RuntimeError):
        te.cal_pesq(c[:, :100], c[:, :100])


def test_pesq_of_a_silent_clip_is_the_floor_not_nan():
    """A model that outputs zeros early in training (or a DC-only / silent clip): the level normalisation has no band power to work with;
    the score is the floor of the scale (raw -0.5 -> MOS-LQO 1.04), finite, and the epoch's average stays a number."""
    pairs = pesq_pairs()
    clean = pairs[0][1]
    for bad in (np.zeros_like(clean), np.full_like(clean, 0.25)):
        for deg, ref in ((bad, clean), (clean, bad), (bad, bad)):
            got = te.cal_pesq(deg[None], ref[None], nthreads=1)[0]
            floor = 0.999 + 4.0 / (1.0 + np.exp(1.3669 * 0.5 + 3.8224))
            assert np.isfinite(got) and got >= floor - 1e-9, got             # (a DC step leaves edge transients in the band: finite, scored normally)
    assert abs(te.cal_pesq(np.zeros_like(clean)[None], clean[None], nthreads=1)[0] - floor) < 1e-9


def test_pesq_cpp_on_heldout_model_outputs():
    """Three utterances of the held-out evaluation (clean / noisy / fp32-trained DCCRN output, int16) with the reference binary's scores.  Until
    round 4 the enhanced ones differed by up to 0.067 MOS (utt 12 here was the worst of 224): not an alignment effect - the binary fades the
    first and last 15 samples and filters exactly the file's samples before its wide-band input filter (csrc_host/pesq.cpp wb_input_filter);
    with both restated the three agree to 0.0014 (576 reference-enhanced clips: mean 0.0004, worst 0.0025)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pesq_heldout.npz"))
    clean = g["clean"].astype(np.float32) / 32768.0
    for key, gold, tol in (("noisy", g["mos_noisy"], 1e-3), ("enhanced", g["mos_enhanced"], 5e-3)):
        got = np.array(te.cal_pesq(g[key].astype(np.float32) / 32768.0, clean))
        assert np.all(np.abs(got - gold) <= tol), (key, got, gold)
    assert abs(te.cal_pesq(g["enhanced"][2:3].astype(np.float32) / 32768.0, clean[2:3])[0] - float(g["mos_enhanced"][2])) < 5e-3
