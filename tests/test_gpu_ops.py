"""GPU tier: every HIP kernel launch of the DCCRN forward+backward op list is compared, op by op, with the
test-only host simulator started from the *same* pre-op state (so an error is localised to one kernel), and stray
writes outside the op's output region are detected.  Then the losses and Adam against the oracle formulas."""
import os

import numpy as np
import pytest
import torch

from oracle import losses as ol
from oracle.dccrn import DCCRNConfig, dccrn_state_shapes
from oracle.step import adam_update
from oracle.weights import fill_state_dict_, formula_state_dict, test_signals as make_signals
from util import knobs, rel_err

pytestmark = pytest.mark.gpu

KIND = {1: "RUNGEMM", 2: "WGRAD", 3: "PACK", 4: "UNPACK", 5: "BN_FINALIZE", 6: "BN_APPLY", 7: "BN_BWD_REDUCE", 8: "BN_BWD_APPLY",
        9: "LSTM_FWD", 10: "LSTM_BWD", 11: "COMBINE_FWD", 12: "COMBINE_BWD", 13: "MASK_FWD", 14: "MASK_BWD", 15: "OLA_FWD",
        16: "OLA_BWD", 17: "SPECOUT_FWD", 18: "SPECOUT_BWD", 19: "MEMSET", 20: "SPLITSUM", 21: "BN_BWD_FINALIZE", 22: "MAGS", 23: "CELL_FWD", 24: "CELL_BWD", 25: "DROPOUT_FWD", 26: "DROPOUT_BWD", 27: "FSN_IN", 28: "FSN_SCALE", 29: "FSN_SBSUM",
        30: "FSN_SBBUILD", 31: "FSN_OUT", 32: "FSN_OUT_BWD", 33: "FSN_SBBWD_SUM", 34: "FSN_SBBWD_APPLY", 35: "REFLECTPAD"}


def _report_path(name):
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, name)


def _typed(t_u8, dt):
    return t_u8.view(torch.bfloat16 if dt == 1 else torch.float32)


@pytest.mark.parametrize("model,B,L,mode,kn,ru,dtype", [("DCCRN", 3, 4000, "E", (16, 32, 32, 64, 64, 64), 128, "fp32"),
                                                        ("DCCRN", 1, 2400, "C", (32, 64, 128, 256, 256, 256), 256, "fp32"),
                                                        ("DCCRN", 3, 4000, "R", (16, 32, 32, 64, 64, 64), 128, "bf16"),
                                                        ("DCCRN", 1, 2400, "C", (32, 64, 128, 256, 256, 256), 256, "bf16"),
                                                        ("DCCRN", 3, 4001, "R", (16, 32, 32, 64, 64, 64), 128, "bf16"),     # L = 4001 marks the cases that send every N <= 64 conv GEMM through the direct-operand kernel (thin.hip)
                                                        ("DCCRN", 1, 2403, "C", (32, 64, 128, 256, 256, 256), 256, "bf16"),
                                                        ("DCCRN", 1, 2401, "C", (32, 64, 128, 256, 256, 256), 256, "bf16"),  # L odd: marks the case that lowers knob CG256_MINM -> wide-tile kernel on every N % 256 == 0 layer
                                                        ("DCCRN", 2, 1600, "C", (16, 32, 32, 64, 64, 64), 512, "bf16"),     # wide LSTM (H = 256): cluster kernels, one partial row block
                                                        ("DCCRN", 18, 2000, "C", (16, 32, 32, 64, 64, 64), 512, "bf16"),    # the same, two row blocks (18 sequences > 16)
                                                        ("DCCRN", 1, 1600, "C", (16, 32, 32, 64, 64, 64), 1024, "bf16"),    # H = 512: 8 workgroups per cluster
                                                        ("DCCRN", 1, 1200, "C", (16, 32, 32, 64, 64, 64), 512, "fp32"),     # wide LSTM in fp32: per-step path
                                                        ("DCCRN", 2, 3400, "C", (16, 32, 32, 64, 64, 64), 128, "bf16"),     # T = 35: chunked two-lane LSTM forward
                                                        ("DCCRN_CBN", 3, 4000, "E", (16, 32, 32, 64, 64, 64), 128, "fp32"),  # DCCRN(use_cbn=True): the six ComplexBatchNorm ops (cbn.hip)
                                                        ("DCCRN_CBN", 1, 2400, "C", (32, 64, 128, 256, 256, 256), 256, "bf16"),
                                                        ("CRN", 3, 4000, "E", (16, 32, 32, 64, 64, 64), 128, "fp32"),
                                                        ("CRN", 2, 2400, "E", (32, 64, 128, 256, 256, 256), 256, "bf16"),
                                                        ("FullSubNet", 2, 13, "E", (128, 64), 0, "fp32"),
                                                        ("FullSubNet", 2, 9, "E", (64, 32), 0, "bf16"),
                                                        ("FullSubNet", 2, 8, "GRU/cumulative_layer_norm", (64, 32), 0, "bf16"),     # cfg.sequence_model / cfg.norm_type variants
                                                        ("FullSubNet", 2, 8, "GRU/offline_gaussian_norm", (64, 32), 0, "fp32"),
                                                        ("FullSubNet", 2, 8, "LSTM/cumulative_laplace_norm", (64, 32), 0, "fp32"),
                                                        ("FullSubNet", 2, 9, "E", (256, 192), 0, "bf16"),      # cluster LSTM kernels on the time-major slabs
                                                        ("FullSubNet", 1, 10, "E", (512, 384), 0, "bf16"),     # reference sizes; T = 10 marks the case that walks 3 row tiles per workgroup
                                                        ("FullSubNet", 1, 11, "E", (512, 384), 0, "bf16"),     # T = 11 marks the case that runs the sub-band model on the row-block kernels (lstm_rows.hip)
                                                        ("FullSubNet", 1, 11, "E", (256, 256), 0, "bf16")])
def test_every_op_against_host_simulator(model, B, L, mode, kn, ru, dtype):
    """Tolerances: fp32 buffers 1e-3 (observed <= 3e-6); bf16 buffers 1.6e-2 = two bf16 ulps of the largest element
    (simulator and kernel round slightly different fp32 accumulations of the SAME bf16 operands); fp32 state written by a whole
    bf16 recurrence op (LSTM_FWD / LSTM_BWD): 4e-3 = one bf16 ulp of h fed back through the frames."""
    from simutil import PHASE_BWD, PHASE_FWD, Plan, fill_params, sim_run
    if L == 2401:                          # the wide-tile kernel needs M >= 4096 by default: lower the bar so that this small case runs it
        L = 2400
        knobs.set("CG256_MINM", "64")
        knobs.set("WG256_MINM", "64")
    else:
        knobs.unset("CG256_MINM")
        knobs.unset("WG256_MINM")
    knobs.unset("DIRECT_MINM")
    knobs.unset("BN_FUSE")
    if model == "FullSubNet" and L == 11 and kn == (512, 384):   # 257 x 11 rows: lower the bar so that the sub-band weight gradients run the 256 x 256 tile
        knobs.set("WG256_MINM", "64")                            # (upper layer: [h1 | h2 | ones] with the bias from the ones MFMA, kRunOnesMfma)
        knobs.set("WGRANK_MINM", "64")                           # ... and the 2-output head's weight gradient the rank-N streaming kernel (kRunRank)
    direct_all = L in (4001, 2403)
    if L in (4001, 2403):                  # the direct-operand kernel takes GEMMs with M >= 65536 by default
        L -= 1 if L == 4001 else 3
        knobs.set("DIRECT_MINM", "0")
        knobs.set("BN_FUSE", "2")   # ... and every BatchNorm layer's backward sums come from its producers' epilogues (thin + tiled kernels)
    if L == 2401 or (model == "DCCRN" and dtype == "fp32" and L == 2400):
        knobs.set("BN_FUSE", "2")   # the same through the wide-tile kernel / in fp32
    knobs.unset("LSTM_RPW")
    if model == "DCCRN" and L == 4000 and dtype == "bf16" and direct_all:
        knobs.set("LSTM_RPW", "16")  # the 16-sequences-per-workgroup recurrences (the default picks one cell per lane below 4096 sequences); read per launch
    knobs.unset("LSTM_MT")
    if model == "FullSubNet" and L == 10:
        knobs.set("LSTM_MT", "3")
    knobs.unset("LSTM_ROWS_MIN")
    if model == "FullSubNet" and L == 11:
        knobs.set("LSTM_ROWS_MIN", "64")
    if model == "FullSubNet":              # L = STFT frames, kn = (fb_hidden, sb_hidden); dropout keep 0.2 exercises the mask hash
        from oracle.fullsubnet import FSNConfig, fsn_state_shapes
        seq, norm = mode.split("/") if "/" in mode else ("LSTM", "offline_laplace_norm")
        P = formula_state_dict(fsn_state_shapes(FSNConfig(fb_hidden=kn[0], sb_hidden=kn[1], sequence_model=seq)))
        plan = Plan(B, L, act_dtype=dtype, model="FullSubNet", fsn=dict(fb_hidden=kn[0], sb_hidden=kn[1], keep=0.2, sequence_model=seq, norm_type=norm))
    elif model == "CRN":
        from oracle.crn import CRNConfig, crn_state_shapes
        P = formula_state_dict(crn_state_shapes(CRNConfig(kernel_num=kn, rnn_units=ru, rnn_input_size=4 * (kn[-1] // 2))))
    else:
        P = formula_state_dict(dccrn_state_shapes(DCCRNConfig(masking_mode=mode, kernel_num=kn, rnn_units=ru, use_cbn=model == "DCCRN_CBN")))
    if model != "FullSubNet":
        plan = Plan(B, L, masking_mode=mode, kernel_num=kn, rnn_units=ru, act_dtype=dtype, model=model.split("_")[0], use_cbn=model == "DCCRN_CBN")
    knobs.unset("BN_FUSE")
    knobs.unset("CG256_MINM")        # the plan is built: later tests get the default thresholds again
    knobs.unset("WG256_MINM")
    knobs.unset("LSTM_ROWS_MIN")
    dev = plan.alloc_arenas("cuda")
    host = plan.alloc_arenas("cpu")
    fill_params(plan, dev, P)
    torch.manual_seed(1)
    if model == "FullSubNet":
        plan.io(dev, "mag", (B, 257, L)).copy_(torch.rand(B, 257, L) * 3)
        plan.io(dev, "grad_crm", (B, 257, L, 2)).copy_(torch.randn(B, 257, L, 2) * 1e-3)
        plan.set_seed(dev, 77)
    else:
        x, y = make_signals(B, L)
        plan.io(dev, "wav", (B, L)).copy_(x)
        if model == "CRN":
            plan.io(dev, "tgt", (B, L)).copy_(y)
        plan.io(dev, "grad_wav", (B, L)).copy_(torch.randn(B, L) * 1e-3)
        plan.io(dev, "grad_real", (B, plan.NF, plan.T)).copy_(torch.randn(B, plan.NF, plan.T) * 1e-4)
        plan.io(dev, "grad_imag", (B, plan.NF, plan.T)).copy_(torch.randn(B, plan.NF, plan.T) * 1e-4)
    # region table: (arena, byte offset, bytes, dtype, name); GRAD / STATE arenas are single fp32 regions
    regions = []
    for name in plan.buffer_names():
        a, off, nb, dt = plan.buffer(name)
        regions.append((a, off, nb, dt, name))
    regions.append((2, 0, plan.arena_bytes[2], 0, "A_GRAD"))
    regions.append((3, 0, plan.arena_bytes[3], 0, "A_STATE"))
    check = [0, 2, 3, 5]
    # The arenas stay on the device.  `host` mirrors the device state at every op boundary and `prev` is a second host copy
    # of that state; after an op only the regions that the kernel or the simulator changed travel (device -> host), so the
    # per-op cost is one device-side compare + one host-side memcmp instead of ten whole-arena copies.
    for a in range(6):
        host[a].copy_(dev[a])
    prev = {a: host[a].clone() for a in check}
    by_arena = {a: [r for r in regions if r[0] == a and r[2] > 0] for a in check}
    # 8-byte words: every region starts on a 256-byte boundary, so a word never straddles two regions
    w64 = {a: (dev[a].view(torch.uint8).numel() // 8) for a in check}
    lo_idx = {a: torch.tensor([r[1] // 8 for r in by_arena[a]], device="cuda") for a in check}
    hi_idx = {a: torch.tensor([min((r[1] + r[2] + 7) // 8, w64[a]) for r in by_arena[a]], device="cuda") for a in check}
    # host side: which regions did the SIMULATOR change?  One vectorised pass per arena (words -> 256-byte blocks -> prefix sums; regions start on
    # 256-byte boundaries, so a block never straddles two regions) instead of one memcmp per region and op (28 000 torch.equal calls per case: half of the
    # suite's run time through round 5)
    nblk = {a: (w64[a] + 31) // 32 for a in check}
    lo_blk = {a: torch.tensor([r[1] // 256 for r in by_arena[a]]) for a in check}
    hi_blk = {a: torch.tensor([min((r[1] + r[2] + 255) // 256, nblk[a]) for r in by_arena[a]]) for a in check}

    def host_changed(a):
        h64 = host[a].view(torch.uint8)[:w64[a] * 8].view(torch.int64)
        p64 = prev[a].view(torch.uint8)[:w64[a] * 8].view(torch.int64)
        ne = h64 != p64
        if ne.numel() % 32:
            ne = torch.cat([ne, torch.zeros(32 - ne.numel() % 32, dtype=torch.bool)])
        cs = torch.cat([torch.zeros(1, dtype=torch.int64), ne.view(-1, 32).any(1).to(torch.int64).cumsum(0)])
        return ((cs[hi_blk[a]] - cs[lo_blk[a]]) > 0).tolist()
    lines, bad = [], []
    for phase in (PHASE_FWD, PHASE_BWD):
        kinds, tags = plan.op_kinds(phase)
        for i in range(plan.num_ops(phase)):
            dbefore = {a: dev[a].view(torch.uint8)[:w64[a] * 8].view(torch.int64).clone() for a in check}
            sim_run(plan, phase, host, i, i + 1)
            plan.run(phase, dev, 0, i, i + 1)
            worst, nchg, stray, where = 0.0, 0, 0, ""
            for a in check:
                if not by_arena[a]:
                    continue
                d64 = dev[a].view(torch.uint8)[:w64[a] * 8].view(torch.int64)
                cs = torch.cumsum(torch.cat([torch.zeros(1, dtype=torch.int32, device="cuda"), (d64 != dbefore[a]).to(torch.int32)]), 0)
                dflags = ((cs[hi_idx[a]] - cs[lo_idx[a]]) > 0).cpu().tolist()        # the one synchronising read per arena
                hflags = host_changed(a)
                h8, p8, d8 = host[a].view(torch.uint8), prev[a].view(torch.uint8), dev[a].view(torch.uint8)
                for (ra, off, nb, dt, name), dchg, hchg in zip(by_arena[a], dflags, hflags):
                    if not (hchg or dchg):
                        continue
                    g8 = d8[off:off + nb].cpu() if dchg else p8[off:off + nb]
                    if hchg:
                        hv, gv = _typed(h8[off:off + nb], dt).double(), _typed(g8, dt).double()
                        den = float(hv.abs().max())
                        err = float((hv - gv).abs().max()) / (den if den > 0 else 1.0)
                        if not np.isfinite(err):
                            err = float("inf")
                        tol = 1.6e-2 if dt == 1 else 1e-3
                        if dt != 1 and dtype == "bf16" and int(kinds[i]) == 1 and name.endswith(".bnpart"):
                            tol = 4e-3     # fp32 partial sums of the bf16 gradient tile this launch ALSO stores: where kernel and simulator round an
                                           # element of that tile to different bf16 neighbours (allowed above: 2 ulps), a 128-row sum with cancellation
                                           # moves by up to that ulp (seen 1.06e-3 of the region's largest sum)
                        if dt != 1 and dtype == "bf16" and int(kinds[i]) in (9, 10):
                            tol = 4e-3     # fp32 state of a bf16 recurrence (cell state, dh): h_t is rounded to bf16 every frame, and a
                                           # rounding flip (one bf16 ulp = 4e-3 of h) between kernel and simulator feeds back into c
                        nchg += hv.numel()
                        if err / tol > worst:
                            worst, where = err / tol, f"{name} err {err:.2e} tol {tol:.0e}"
                    else:
                        # stray write: the kernel changed bytes of a region the simulator did not touch
                        stray += int((g8 != p8[off:off + nb]).sum())
                    h8[off:off + nb].copy_(g8)                                        # both host copies := device state
                    if dchg:
                        p8[off:off + nb].copy_(g8)
            lines.append(f"phase {phase} op {i:3d} {KIND.get(int(kinds[i]), str(int(kinds[i]))):16s} tag {int(tags[i]):4d} elems {nchg:9d} "
                         f"err/tol {worst:.3e} stray {stray} {where}")
            if not (worst < 1.0) or stray:
                bad.append(lines[-1])
    knobs.unset("LSTM_MT")
    knobs.unset("LSTM_RPW")
    knobs.unset("DIRECT_MINM")
    with open(_report_path(f"ops_report_{model}_B{B}_{mode.replace('/', '-')}_{dtype}_{L}.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    assert not bad, "\n".join(bad[:20])


@pytest.mark.parametrize("kind,name", [(0, "MSE"), (1, "SDR"), (2, "SI-SNR"), (3, "SI-SDR")])
def test_fused_losses_forward_backward(kind, name):
    import ctypes as C
    from sefd_amd import _lib
    L_ = _lib.lib()
    B, L = 5, 4802
    x, y = make_signals(B, L)
    est = (x * 0.9).clone().requires_grad_(True)
    ref = ol.main_loss(name, est, y)
    ref.backward()
    e_d, y_d = est.detach().cuda(), y.cuda()
    ws = torch.zeros(L_.sefd_loss_ws_floats(B), device="cuda")
    out = torch.zeros(1, device="cuda")
    g = torch.zeros(B, L, device="cuda")
    gs = torch.full((1,), 0.5, device="cuda")
    vp = lambda t: C.c_void_p(t.data_ptr())
    assert L_.sefd_loss_forward(kind, vp(e_d), vp(y_d), B, L, vp(ws), vp(out), None) == 0
    assert L_.sefd_loss_backward(kind, vp(e_d), vp(y_d), B, L, vp(ws), vp(gs), vp(g), None) == 0
    torch.cuda.synchronize()
    assert abs(float(out) - float(ref)) < 1e-4 * max(1.0, abs(float(ref))), (float(out), float(ref))
    assert rel_err(g.cpu(), 0.5 * est.grad) < 1e-3          # tolerance: 1e-3 relative fp32 (north_star)


@pytest.mark.parametrize("name", ["MSE", "SDR", "SI-SNR", "SI-SDR"])
@pytest.mark.parametrize("L", [2, 5])
def test_short_row_losses_both_slots(name, L):
    """FullSubNet.loss (models.py:674-682): reductions over a 2-element last axis, network output in EITHER slot (trainer.py:107 puts it
    in `target`); the tools_for_loss mirrors must give the oracle's value and the gradient with respect to both arguments."""
    import sefd_amd  # noqa: F401
    from sefd_amd import tools_for_loss as tfl
    torch.manual_seed(3)
    a = (torch.randn(3, 41, 7, L) * 0.7)
    b = (a + 0.3 * torch.randn(3, 41, 7, L))
    fn = {"MSE": lambda e, t: tfl.mse(e, t), "SDR": lambda e, t: -tfl.sdr(t, e), "SI-SNR": lambda e, t: -tfl.si_snr(e, t), "SI-SDR": lambda e, t: -tfl.si_sdr(t, e)}[name]
    for slot in (0, 1):
        e, t = a.clone(), b.clone()
        (e if slot == 0 else t).requires_grad_(True)
        ref = ol.main_loss(name, e, t)
        ref.backward()
        ed, td = e.detach().cuda(), t.detach().cuda()
        (ed if slot == 0 else td).requires_grad_(True)
        out = fn(ed, td)
        (out * 0.5).backward()
        assert abs(float(out) - float(ref)) < 1e-4 * max(1.0, abs(float(ref))), (name, slot, float(out), float(ref))
        gd, gr = (ed if slot == 0 else td).grad.cpu(), (e if slot == 0 else t).grad
        assert rel_err(gd, 0.5 * gr) < 1e-3, (name, slot)


def test_torch_library_ops():
    """`sefd::` custom ops (sefd_amd/ops.py, torch.library): the dispatcher-level surface of the same C entry points the mirrors call - same
    values, registered backward, fake implementations that trace."""
    import sefd_amd  # noqa: F401
    from sefd_amd import ops, models, config as cfg, tools_for_loss as tfl  # noqa: F401
    from sefd_amd.plan import PHASE_FWD
    x, y = make_signals(3, 4802)
    for kind, name in ((0, "MSE"), (1, "SDR"), (2, "SI-SNR"), (3, "SI-SDR")):
        e = (0.9 * x).cuda().requires_grad_(True)
        out = torch.ops.sefd.loss(kind, e, y.cuda())
        out.backward()
        er = (0.9 * x).clone().requires_grad_(True)
        ref = ol.main_loss(name, er, y)
        ref.backward()
        assert abs(float(out) - float(ref)) < 1e-4 * max(1.0, abs(float(ref))) and rel_err(e.grad.cpu(), er.grad) < 1e-3, name
    a, b = torch.randn(50, 2), torch.randn(50, 2)                        # short rows: gradient to the `tgt` slot as well
    bd = b.cuda().requires_grad_(True)
    torch.ops.sefd.loss(2, a.cuda(), bd).backward()
    br = b.clone().requires_grad_(True)
    ol.main_loss("SI-SNR", a, br).backward()
    assert rel_err(bd.grad.cpu(), br.grad) < 1e-3
    p, g = torch.randn(1000), torch.randn(1000) * 1e-2
    pd, gd, md, vd = p.cuda(), g.cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    torch.ops.sefd.adam_step_(pd, gd, md, vd, 1, 1e-3, 0.9, 0.999, 1e-8, 1.0)
    pr, _, _ = adam_update(p, g, torch.zeros(1000), torch.zeros(1000), 1)
    assert rel_err(pd.cpu(), pr) < 1e-6
    # a planned model through sefd::plan_run == the module's forward
    m = make_dccrn_small()
    xs = x[:2, :4000].cuda().contiguous()
    with torch.no_grad():
        want = m(xs)[2]
        rt = next(v for k, v in m._runtimes.items() if isinstance(k[0], int))
        rt.wav.copy_(xs * 0.5)
        torch.ops.sefd.plan_run(rt.plan.h.value if hasattr(rt.plan.h, "value") else int(rt.plan.h), PHASE_FWD, rt.arenas)
        half = rt.out_wav.clone()
        rt.wav.copy_(xs)
        torch.ops.sefd.plan_run(rt.plan.h.value if hasattr(rt.plan.h, "value") else int(rt.plan.h), PHASE_FWD, rt.arenas)
        assert torch.equal(rt.out_wav, want) and not torch.equal(half, want)
    assert torch.ops.sefd.loss(2, torch.empty(4, 100, device="meta"), torch.empty(4, 100, device="meta")).shape == ()      # fake impl traces


def make_dccrn_small():
    import sefd_amd  # noqa: F401
    from sefd_amd import config as cfg, models
    cfg.dccrn_kernel_num, cfg.masking_mode, cfg.loss, cfg.act_dtype, cfg.perceptual, cfg.lstm, cfg.skip_type = [16, 32, 32, 64, 64, 64], "E", "SI-SNR", "fp32", False, "complex", True
    m = models.DCCRN(rnn_units=128, masking_mode="E")
    fill_state_dict_(m)
    return m.to("cuda").eval()


def test_adam_step_matches_torch_formula():
    import ctypes as C
    from sefd_amd import _lib
    L_ = _lib.lib()
    n = 100003
    torch.manual_seed(0)
    p, g = torch.randn(n), torch.randn(n) * 1e-2
    m, v = torch.zeros(n), torch.zeros(n)
    pd, gd, md, vd = p.cuda(), g.cuda(), m.cuda(), v.cuda()
    vp = lambda t: C.c_void_p(t.data_ptr())
    for step in (1, 2, 3):
        p, m, v = adam_update(p, g, m, v, step)
        assert L_.sefd_adam_step(vp(pd), vp(gd), vp(md), vp(vd), n, step, 1e-3, 0.9, 0.999, 1e-8, 1.0, None) == 0
    torch.cuda.synchronize()
    assert rel_err(pd.cpu(), p) < 1e-6
    # and against torch.optim.Adam itself
    q = torch.nn.Parameter(torch.ones(8))
    opt = torch.optim.Adam([q], lr=1e-3)
    q.grad = torch.arange(8.0) * 0.1 - 0.3
    opt.step()
    qd, gq, mq, vq = torch.ones(8).cuda(), q.grad.cuda(), torch.zeros(8).cuda(), torch.zeros(8).cuda()
    L_.sefd_adam_step(vp(qd), vp(gq), vp(mq), vp(vq), 8, 1, 1e-3, 0.9, 0.999, 1e-8, 1.0, None)
    assert rel_err(qd.cpu(), q.detach()) < 1e-6


def test_syncbn_plans_two_ranks_emulated_on_one_gpu():
    """SyncBN op modes on the HIP kernels: two bn_world=2 plans (the two "ranks", 2 utterances each) are advanced in lock
    step on one GPU, their statistics buffers summed at every sync point (what RCCL does between the ranks), and must
    reproduce the single plan over all 4 utterances: outputs, gradients (summed), BatchNorm running statistics."""
    from simutil import PHASE_BWD, PHASE_FWD, Plan, fill_params, read_params
    from sefd_amd.plan import ARENA_GRAD, ARENA_STATE
    kn, ru, B, L = (16, 32, 32, 64, 64, 64), 128, 4, 3000
    P = formula_state_dict(dccrn_state_shapes(DCCRNConfig(masking_mode="C", kernel_num=kn, rnn_units=ru)))
    # PReLU slopes = 1 (identity): the two runs sum their statistics in different orders, so activations differ by ~1e-7, and ONE element
    # whose pre-activation lies that close to zero then takes different PReLU branches in the backward - which moves a whole layer's sums
    # by (1 - slope) * dz of that element (seen: 3e-3 of a layer's sum(dbn), one flip among 1.1 M elements; expected ~0.5 flips per run).
    # That discontinuity is not what this test is about (the SyncBN plumbing is); the kernels' PReLU branches are pinned by the per-op test.
    for k in P:
        if k.endswith(".2.weight"):
            P[k] = torch.ones_like(P[k])
    x, _ = make_signals(B, L)
    torch.manual_seed(7)
    gw = torch.randn(B, L)
    stream = torch.cuda.current_stream().cuda_stream

    def prep(plan, xs):
        ar = plan.alloc_arenas("cuda")
        fill_params(plan, ar, P)
        plan.io(ar, "wav", xs.shape).copy_(xs.cuda())
        return ar

    def seed_grad(plan, ar, gs):
        plan.io(ar, "grad_wav", gs.shape).copy_(gs.cuda())
        plan.io(ar, "grad_real", (gs.shape[0], plan.NF, plan.T)).zero_()
        plan.io(ar, "grad_imag", (gs.shape[0], plan.NF, plan.T)).zero_()

    full = Plan(B, L, masking_mode="C", kernel_num=kn, rnn_units=ru)
    far = prep(full, x)
    full.run(PHASE_FWD, far, stream)
    seed_grad(full, far, gw)
    full.run(PHASE_BWD, far, stream)
    ranks = [Plan(B // 2, L, masking_mode="C", kernel_num=kn, rnn_units=ru, bn_world=2) for _ in range(2)]
    ars = [prep(p, x[2 * r:2 * r + 2]) for r, p in enumerate(ranks)]
    for ph in (PHASE_FWD, PHASE_BWD):
        if ph == PHASE_BWD:
            for r in range(2):
                seed_grad(ranks[r], ars[r], gw[2 * r:2 * r + 2])
        cur = 0
        for sph, op, a, off, cnt, dtype in ranks[0].sync_points():
            if sph != ph:
                continue
            views = []
            for r in range(2):
                ranks[r].run(ph, ars[r], stream, cur, op + 1)
                nb = cnt * (8 if dtype == torch.float64 else 4)
                views.append(ars[r][a].view(torch.uint8)[off:off + nb].view(dtype))
            tot = views[0] + views[1]
            views[0].copy_(tot)
            views[1].copy_(tot)
            cur = op + 1
        for r in range(2):
            ranks[r].run(ph, ars[r], stream, cur, ranks[r].num_ops(ph))
    torch.cuda.synchronize()
    fw = full.io(far, "out_wav", (B, L))
    for r in range(2):
        assert rel_err(ranks[r].io(ars[r], "out_wav", (2, L)), fw[2 * r:2 * r + 2]) < 1e-4
    fg = read_params(full, far, ARENA_GRAD)
    g0, g1 = read_params(ranks[0], ars[0], ARENA_GRAD), read_params(ranks[1], ars[1], ARENA_GRAD)
    for k in fg:
        if k.endswith("conv.bias") and not k.startswith("decoder.5."):
            continue
        assert rel_err(g0[k] + g1[k], fg[k]) < 1e-3, k
    fs, s0 = read_params(full, far, ARENA_STATE, full.state), read_params(ranks[0], ars[0], ARENA_STATE, ranks[0].state)
    for k in fs:
        assert rel_err(s0[k], fs[k]) < 1e-4, k
