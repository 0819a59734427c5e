"""CPU tier: the drop-in surface (reference module names), the torch.istft plan on the host simulator, config semantics."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_block_resolves_to_the_package():
    """train_interface.py:3-15 of the reference, executed verbatim against `dropin/` on sys.path (fresh interpreter)."""
    code = r'''
import sys
sys.path.insert(0, %r)
import config as cfg
from models import DCCRN, CRN, FullSubNet  # you can import 'DCCRN' or 'CRN' or 'FullSubNet'
from dataloader import create_dataloader
from trainer import model_train, model_validate, \
    model_perceptual_train, model_perceptual_validate, \
    dccrn_direct_train, dccrn_direct_validate, \
    crn_direct_train, crn_direct_validate, \
    fullsubnet_train, fullsubnet_validate
import tools_for_model as tools
import sefd_amd
assert cfg is sefd_amd.config and DCCRN is sefd_amd.models.DCCRN and tools is sefd_amd.tools_for_model
cfg.loss = 'SI-SNR'
assert sefd_amd.config.loss == 'SI-SNR'
m = DCCRN()
keys = list(m.state_dict().keys())
assert keys[0] == 'stft.weight' and 'enhance.1.r_trans.weight' in keys and 'decoder.5.0.imag_conv.bias' in keys
assert sum(p.numel() for p in m.parameters()) == 3671053
for f in (tools.stft, tools.istft, tools.mag_phase, tools.build_complex_ideal_ratio_mask, tools.decompress_cIRM, tools.Bar):
    assert callable(f)
print("OK")
''' % os.path.join(ROOT, "dropin")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd="/tmp")
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


def test_config_is_plain_constants_with_the_reference_names():
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg
    src = open(cfg.__file__).read()
    assert "globals()" not in src
    for name, val in dict(model='DCCRN', perceptual=False, lstm='complex', masking_mode='E', skip_type=True, max_epochs=100, batch=10,
                          fs=16000, win_len=400, win_inc=100, fft_len=512, window='hanning', rnn_layers=2, rnn_units=256,
                          rnn_input_size=512, sb_num_neighbors=15, look_ahead=2, norm_type="offline_laplace_norm", num_freqs=257).items():
        assert name in src and getattr(cfg, name) == val, name
    assert cfg.dccrn_kernel_num[:3] == [32, 64, 128] or len(cfg.dccrn_kernel_num) == 6      # other tests switch sizes at run time


@pytest.mark.parametrize("L", [6000, 4801])
def test_torch_istft_plan_on_host_simulator(L):
    """tools.istft (tools_for_model.py:651-680) = torch.istft(n_fft 512, hop 300, win 400, hann, center, length): the plan's
    inverse-FFT frames + envelope-normalised overlap-add against torch.istft itself, on consistent and inconsistent spectra."""
    from simutil import PHASE_FWD, Plan, sim_run
    B = 2
    plan = Plan(B, L, win_len=400, win_inc=300, fft_len=512, model="TorchISTFT")
    ar = plan.alloc_arenas("cpu")
    torch.manual_seed(0)
    x = torch.randn(B, L) * 0.3
    w = torch.hann_window(400)
    S = torch.stft(x, 512, 300, 400, window=w, return_complex=True)
    assert S.shape[-1] == plan.T
    for sp in (S, S * (1.0 + 0.5 * torch.randn_like(S.real))):
        plan.io(ar, "spec", (B, 257, plan.T, 2)).copy_(torch.view_as_real(sp))
        sim_run(plan, PHASE_FWD, ar)
        ref = torch.istft(sp, 512, 300, 400, window=w, length=L)
        assert float((plan.io(ar, "wav", (B, L)) - ref).abs().max()) < 1e-5 * max(1.0, float(ref.abs().max()))


def test_adam_state_dict_format_and_pending_load():
    """optim.Adam keeps the reference's resume order working: load_state_dict right after construction (train_interface.py:59,110)."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    from sefd_amd.optim import Adam
    cfg.dccrn_kernel_num = [16, 32, 32, 64, 64, 64]
    m = models.DCCRN(rnn_units=128)
    opt = Adam(m.parameters(), lr=1e-3)
    assert opt._owner() is m
    sd = opt.state_dict()
    assert sd["state"] == {} and sd["param_groups"][0]["lr"] == 1e-3 and len(sd["param_groups"][0]["params"]) == len(list(m.parameters()))
    ref = torch.optim.Adam(m.parameters(), lr=5e-4)
    for p in m.parameters():
        p.grad = torch.ones_like(p) * 0.01
    ref.step()
    opt.load_state_dict(ref.state_dict())                      # model still on the CPU: kept, applied at bind time
    assert opt._pending is not None and opt.state_dict()["state"][0]["exp_avg"].shape == next(m.parameters()).shape
    with pytest.raises(RuntimeError):
        opt.step()                                             # no CPU path
    mh = models.DCCRN(rnn_units=128, win_type="hamming")       # any scipy.signal.get_window name (tools_for_model.py:19-20)
    from scipy.signal import get_window
    assert torch.allclose(mh.istft.window.reshape(-1).double(), torch.from_numpy(get_window("hamming", cfg.win_len, fftbins=True)), atol=1e-7)
    cfg.dccrn_kernel_num = [32, 64, 128, 256, 256, 256]


def test_torch_library_namespace_registers_and_traces():
    """`sefd::` custom ops (sefd_amd/ops.py): registered with the dispatcher, fake implementations give the output shapes without a GPU, and a
    CPU tensor is refused loudly (no CPU fallback)."""
    import pytest
    import torch
    import sefd_amd  # noqa: F401
    from sefd_amd import ops  # noqa: F401
    for name in ("loss", "loss_forward", "loss_backward", "adam_step_", "mix_snr", "plan_run"):
        assert hasattr(torch.ops.sefd, name), name
    e, t = torch.empty(4, 100, device="meta"), torch.empty(4, 100, device="meta")
    assert torch.ops.sefd.loss(2, e, t).shape == ()
    out, ws = torch.ops.sefd.loss_forward(1, e, t)
    assert out.shape == () and ws.numel() > 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        torch.ops.sefd.loss(2, torch.zeros(2, 100), torch.zeros(2, 100))


@pytest.mark.parametrize("seq", ["LSTM", "GRU"])
def test_fullsubnet_weight_init_draws_the_reference_values(seq):
    """FullSubNet(weight_init=True) (reference models.py:623-624 -> BaseModel.weight_init, tools_for_model.py:1120-1184): same module tree, same
    traversal, same torch.nn.init calls -> the parameters the reference holds after construction under the same seed (digest captured by
    tests/golden/make_golden.py fsn_weight_init)."""
    import sefd_amd  # noqa: F401
    from sefd_amd import models
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "fsn_weight_init.npz"))
    torch.manual_seed(0)
    m = models.FullSubNet(fb_model_hidden_size=64, sb_model_hidden_size=32, sequence_model=seq, weight_init=True)
    names = [k for k, _ in m.named_parameters()]
    assert [k[len(f"g/{seq}/"):] for k in g.files if k.startswith(f"g/{seq}/")] == names
    for k, p in m.named_parameters():
        v = p.detach().double().reshape(-1)
        got = np.array([float(v.sum()), float(v.abs().sum())] + [float(t) for t in v[:6]])
        assert np.allclose(got, g[f"g/{seq}/{k}"], rtol=1e-6, atol=1e-7), k


def test_dccrn_complex_batch_norm_state_dict_layout():
    """DCCRN(use_cbn=True): the reference's ComplexBatchNorm keys (tools_for_model.py:441-467) in registration order, Wri ~ U(-0.9, 0.9)."""
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    cfg.dccrn_kernel_num = [16, 32, 32, 64, 64, 64]
    torch.manual_seed(0)
    m = models.DCCRN(rnn_units=128, use_cbn=True)
    keys = [k for k in m.state_dict() if k.startswith("encoder.1.1.")]
    assert keys == ["encoder.1.1." + n for n in ("Wrr", "Wri", "Wii", "Br", "Bi", "RMr", "RMi", "RVrr", "RVri", "RVii", "num_batches_tracked")]
    sd = m.state_dict()
    assert sd["encoder.1.1.Wrr"].shape == (16,) and float(sd["encoder.1.1.Wrr"].min()) == 1.0 and float(sd["encoder.1.1.RVii"].max()) == 1.0
    assert float(sd["encoder.1.1.Wri"].abs().max()) <= 0.9 and float(sd["encoder.1.1.Wri"].abs().max()) > 0.1
    assert "decoder.5.1.Wrr" not in sd and "decoder.4.1.Wrr" in sd            # the mask layer has no normalisation (models.py:140-153)
