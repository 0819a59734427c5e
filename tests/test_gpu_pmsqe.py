"""GPU tier: the PMSQE loss of csrc/pmsqe.hip (forward, PIT over the seconds, gradient w.r.t. the estimated wave) through the C ABI against
oracle/pmsqe.py (float64 + autograd).  PMSQE parity is UNPINNED (third-party arithmetic absent from the reference): HIP == oracle only."""
import pytest
import torch

from oracle import pmsqe
from test_oracle_pmsqe import speechlike

pytestmark = pytest.mark.gpu
TOL_LOSS = 1e-3        # fp32 vs float64, relative; the loss has hard thresholds (audibility, silence, clamps) that a rounding can flip
TOL_GRAD = 2e-3        # relative L2 of the whole gradient


def _loss_and_grad(c, e, power):
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, tools_for_loss as tfl
    cfg.pmsqe_power = power
    try:
        ed = e.cuda().requires_grad_()
        loss = tfl.get_array_pmsqe_loss(c.cuda(), ed)
        loss.backward()
        return float(loss.detach()), ed.grad.cpu()
    finally:
        cfg.pmsqe_power = False


@pytest.mark.parametrize("power", [False, True])
@pytest.mark.parametrize("B", [1, 5])
def test_pmsqe_loss_and_gradient_match_oracle(B, power):
    c, n = speechlike(B, seed=10 + B)
    loss, grad = _loss_and_grad(c, n, power)
    eo = n.clone().double().requires_grad_()
    lo = pmsqe.pmsqe_loss(c, eo, power)
    lo.backward()
    lo = lo.detach()
    assert abs(loss - float(lo)) <= TOL_LOSS * abs(float(lo)), (loss, float(lo))
    rel = float((grad.double() - eo.grad).norm() / eo.grad.norm())
    assert rel < TOL_GRAD, rel
    # the last 128 samples of every second are in no frame: zero gradient, as in the reference's unpadded STFT
    assert float(grad.reshape(B, 3, 16000)[:, :, 15872:].abs().max()) == 0.0


def test_pit_picks_the_rotation():
    c, n = speechlike(3, seed=4)
    rot = n.reshape(3, 3, 16000)[:, [2, 0, 1]].reshape(3, -1).contiguous()
    l0, _ = _loss_and_grad(c, n, True)
    l1, g1 = _loss_and_grad(c, rot, True)
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    eo = rot.clone().double().requires_grad_()
    pmsqe.pmsqe_loss(c, eo, True).backward()
    assert float((g1.double() - eo.grad).norm() / eo.grad.norm()) < TOL_GRAD


def test_upstream_gradient_and_errors():
    import sefd_amd  # noqa: F401
    from sefd_amd import tools_for_loss as tfl
    c, n = speechlike(2, seconds=2, seed=8)
    e1 = n.cuda().requires_grad_()
    tfl.get_array_pmsqe_loss(c.cuda(), e1).backward()
    e2 = n.cuda().requires_grad_()
    (tfl.get_array_pmsqe_loss(c.cuda(), e2) * 0.5).backward()
    assert torch.allclose(e2.grad, 0.5 * e1.grad, rtol=1e-6, atol=0)
    with pytest.raises(ValueError):
        tfl.get_array_pmsqe_loss(c[:, :20000].cuda(), n[:, :20000].cuda())        # view(N, -1, fs) fails in the reference too
    with pytest.raises(RuntimeError):
        tfl.get_array_pmsqe_loss(c, n)                                             # no CPU path


def test_dccrn_perceptual_step_with_pmsqe():
    """trainer.py:45-82 with cfg.perceptual = 'PMSQE': loss = (main + PMSQE) / 2; PMSQE of the model's own output equals the oracle's and
    a few Adam steps lower the combined loss."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    old = (cfg.perceptual, cfg.loss, cfg.masking_mode, list(cfg.dccrn_kernel_num), cfg.act_dtype)
    cfg.perceptual, cfg.loss, cfg.masking_mode, cfg.dccrn_kernel_num, cfg.act_dtype = "PMSQE", "SI-SNR", "E", [16, 32, 32, 64, 64, 64], "fp32"
    try:
        torch.manual_seed(0)
        m = models.DCCRN(rnn_units=64, masking_mode="E").to("cuda").train()
        opt = Adam(m.parameters(), lr=1e-3)
        c, n = speechlike(4, seed=21)
        c, n = c.cuda(), n.cuda()
        hist = []
        for it in range(6):
            opt.zero_grad()
            real, imag, out = m(n)
            main = m.loss(out, c)
            perc = m.loss(out, c, real, imag, perceptual=True)
            if it == 0:
                ref = float(pmsqe.pmsqe_loss(c.cpu(), out.detach().cpu()))
                assert abs(float(perc) - ref) <= TOL_LOSS * abs(ref)
            loss = (main + perc) / 2
            loss.backward()
            opt.step()
            hist.append(float(loss))
        assert all(map(lambda v: v == v, hist)) and hist[-1] < hist[0], hist
    finally:
        cfg.perceptual, cfg.loss, cfg.masking_mode, cfg.dccrn_kernel_num, cfg.act_dtype = old


@pytest.mark.parametrize("perceptual", ["PMSQE", "LMS"])
def test_fused_train_step_with_perceptual_equals_autograd_route(perceptual):
    """model.train_step(..., perceptual=...) == the model_perceptual_train loop (trainer.py:45-82) through autograd: same loss, same
    parameters after the Adam step."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    old = (cfg.perceptual, cfg.loss, cfg.masking_mode, list(cfg.dccrn_kernel_num), cfg.act_dtype)
    cfg.perceptual, cfg.loss, cfg.masking_mode, cfg.dccrn_kernel_num, cfg.act_dtype = perceptual, "SI-SNR", "E", [16, 32, 32, 64, 64, 64], "fp32"
    try:
        c, n = speechlike(3, seed=31)
        c, n = c.cuda(), n.cuda()
        res = []
        for fused in (False, True):
            torch.manual_seed(0)
            m = models.DCCRN(rnn_units=64, masking_mode="E").to("cuda").train()
            opt = Adam(m.parameters(), lr=1e-3)
            for _ in range(2):
                if fused:
                    loss = m.train_step(n, c, opt, perceptual=perceptual)
                else:
                    real, imag, out = m(n)
                    loss = (m.loss(out, c) + m.loss(out, c, real, imag, perceptual=True)) / 2
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
            res.append((float(loss.detach()), torch.cat([p.detach().flatten() for p in m.parameters()]).cpu()))
        assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[0][0])
        assert float((res[0][1] - res[1][1]).norm() / res[0][1].norm()) < 1e-5
    finally:
        cfg.perceptual, cfg.loss, cfg.masking_mode, cfg.dccrn_kernel_num, cfg.act_dtype = old
