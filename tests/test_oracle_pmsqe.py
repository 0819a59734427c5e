"""CPU tier: oracle/pmsqe.py.  PMSQE is third-party arithmetic in the reference (asteroid, absent, unversioned): PARITY UNPINNED - there is
no golden to pin it to, so the oracle is checked for the properties the published loss has; the HIP path is then held to the oracle
(tests/test_gpu_pmsqe.py)."""
import numpy as np
import torch

from oracle import pmsqe


def speechlike(n_utt, seconds=3, seed=0):
    """Harmonic 'voiced' source with a moving pitch and a syllable envelope + coloured noise, values in (-1, 1)."""
    rng = np.random.default_rng(seed)
    L = seconds * 16000
    t = np.arange(L) / 16000
    clean, noisy = [], []
    for _ in range(n_utt):
        f0 = 120 + 60 * rng.random() + 20 * np.sin(2 * np.pi * (0.5 + rng.random()) * t)
        ph = 2 * np.pi * np.cumsum(f0) / 16000
        s = sum(np.sin(k * ph) / k ** (0.6 + rng.random()) for k in range(1, 25))
        env = np.clip(np.sin(2 * np.pi * (2.5 + 2 * rng.random()) * t + 6 * rng.random()), 0, None) ** 0.7
        s = 0.3 * s * env / np.abs(s * env).max()
        v = np.convolve(rng.standard_normal(L + 31), np.exp(-np.arange(32) / (1 + 8 * rng.random())), mode="valid")
        v *= np.sqrt((s ** 2).mean() / (v ** 2).mean() / 10 ** (rng.uniform(0, 15) / 10))
        clean.append(s)
        noisy.append(np.clip(s + v, -1, 1))
    return torch.tensor(np.stack(clean), dtype=torch.float32), torch.tensor(np.stack(noisy), dtype=torch.float32)


def test_shapes_and_frames():
    c, _ = speechlike(2)
    sp = pmsqe.spectra(c.double())
    assert sp.shape == (2, 3, 61, 257)                     # 3 seconds = 3 sources, (16000 - 512) / 256 + 1 = 61 frames, 257 bins
    thr, zp, width, M, mask = pmsqe.constants()
    assert M.shape == (257, 49) and int((M > 0).sum()) == 256 and float(M[256].sum()) == 0.0   # P.862: 256 bins into 49 bands
    assert abs(float(zp[-1]) - 0.23) < 1e-12 and float(zp[0]) > 0.23


def test_identical_is_near_zero_and_noise_is_monotone():
    c, n = speechlike(2)
    for power in (False, True):
        zero = float(pmsqe.pmsqe_loss(c, c, power))
        half = float(pmsqe.pmsqe_loss(c, c + 0.3 * (n - c), power))
        full = float(pmsqe.pmsqe_loss(c, n, power))
        assert 0 <= zero < 1e-3 < half < full < 5


def test_pit_over_the_seconds():
    """PITLossWrapper('pw_pt'): rotating the seconds of the estimate does not change the loss; the pairwise matrix of the rotated
    estimate is the column-rotated one."""
    c, n = speechlike(2, seed=3)
    rot = n.reshape(2, 3, 16000)[:, [1, 2, 0]].reshape(2, -1)
    assert abs(float(pmsqe.pmsqe_loss(c, n)) - float(pmsqe.pmsqe_loss(c, rot))) < 1e-9
    pw, pwr = pmsqe.pairwise(n.double(), c.double()), pmsqe.pairwise(rot.double(), c.double())
    assert torch.allclose(pwr, pw[:, [1, 2, 0]])
    assert float(pmsqe.pmsqe_loss(c, n)) <= float(pw.diagonal(dim1=1, dim2=2).mean()) + 1e-12


def test_level_invariance_and_gradient():
    """SLL equalisation: the loss does not depend on the playback level of either signal (power mode exactly; the 1e-8 inside the
    magnitude makes it approximate there)."""
    c, n = speechlike(1, seed=5)
    a = float(pmsqe.pmsqe_loss(c, n, True))
    assert abs(float(pmsqe.pmsqe_loss(0.5 * c, 2.0 * n, True)) - a) < 1e-6 * max(a, 1)
    e = n.clone().double().requires_grad_()
    pmsqe.pmsqe_loss(c, e).backward()
    assert torch.isfinite(e.grad).all() and float(e.grad.abs().max()) > 0
    # descent along the gradient lowers the loss
    step = 0.02 / float(e.grad.abs().max())
    assert float(pmsqe.pmsqe_loss(c, (e - step * e.grad).detach())) < float(pmsqe.pmsqe_loss(c, n))
