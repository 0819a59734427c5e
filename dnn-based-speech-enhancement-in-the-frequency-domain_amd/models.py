"""Drop-in `models` surface: `DCCRN` with the reference's constructor, forward/loss signatures and state_dict keys
(reference models.py:15-323), computed by the gfx950 HIP library through its C ABI.

The nn.Module tree only *holds* parameters (same names, shapes, init and registration order as the reference, so
checkpoints interchange - SURVEY.md Appendix B).  All arithmetic - STFT, complex convs, BatchNorm+PReLU, complex
LSTM, mask, iSTFT, losses, backward, Adam - runs in `libsefd_hip.so`; there is no PyTorch/CPU fallback.
"""
import ctypes as C
import weakref

import torch
import torch.nn as nn

from . import config as cfg
from . import _lib
from .frontend_consts import stft_kernels
from .plan import (ARENA_COUNT, ARENA_CONST, ARENA_GRAD, ARENA_IO, ARENA_PARAM, ARENA_STATE, ARENA_WS, PHASE_BWD,
                   PHASE_FWD, RUN_WAVE_ONLY, Plan)
from . import tools_for_loss as tfl


# ------------------------------------------------------------------------------------------ parameter holders
class ConvSTFT(nn.Module):
    """Buffer holder for `stft.weight` (tools_for_model.py:36-52)."""

    def __init__(self, win_len, win_inc, fft_len, win_type, feature_type='complex', fix=True):
        super().__init__()
        K, _, _ = stft_kernels(win_len, fft_len, win_type)
        self.register_buffer('weight', K)
        self.feature_type, self.stride, self.win_len, self.dim = feature_type, win_inc, win_len, fft_len


class ConviSTFT(nn.Module):
    """Buffer holder for `istft.weight/window/enframe` (tools_for_model.py:71-88)."""

    def __init__(self, win_len, win_inc, fft_len, win_type, feature_type='complex', fix=True):
        super().__init__()
        _, Kinv, w = stft_kernels(win_len, fft_len, win_type)
        self.register_buffer('weight', Kinv)
        self.register_buffer('window', w)
        self.register_buffer('enframe', torch.eye(win_len)[:, None, :])
        self.feature_type, self.stride, self.win_len, self.dim = feature_type, win_inc, win_len, fft_len


class ComplexConv2d(nn.Module):
    """Parameter holder with the reference's layout and init (tools_for_model.py:199-241)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.real_conv = nn.Conv2d(in_channels // 2, out_channels // 2, kernel_size, stride, padding=[padding[0], 0])
        self.imag_conv = nn.Conv2d(in_channels // 2, out_channels // 2, kernel_size, stride, padding=[padding[0], 0])
        for c in (self.real_conv, self.imag_conv):
            nn.init.normal_(c.weight.data, std=0.05)
            nn.init.constant_(c.bias, 0.)


class ComplexConvTranspose2d(nn.Module):
    """Parameter holder (tools_for_model.py:272-309)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, output_padding):
        super().__init__()
        self.real_conv = nn.ConvTranspose2d(in_channels // 2, out_channels // 2, kernel_size, stride, padding=padding,
                                            output_padding=output_padding)
        self.imag_conv = nn.ConvTranspose2d(in_channels // 2, out_channels // 2, kernel_size, stride, padding=padding,
                                            output_padding=output_padding)
        for c in (self.real_conv, self.imag_conv):
            nn.init.normal_(c.weight.data, std=0.05)
            nn.init.constant_(c.bias, 0.)


class ComplexBatchNorm(nn.Module):
    """Parameter / buffer holder of the reference's ComplexBatchNorm (tools_for_model.py:430-480): Wrr, Wri, Wii, Br, Bi and the running
    RMr, RMi, RVrr, RVri, RVii over num_features // 2 complex channels; `reset_parameters` as there (Wri ~ U(-0.9, 0.9)).  The arithmetic is
    csrc/cbn.hip."""

    def __init__(self, num_features, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features // 2, eps, momentum
        h = self.num_features
        for name in ("Wrr", "Wri", "Wii", "Br", "Bi"):
            setattr(self, name, nn.Parameter(torch.empty(h)))
        self.register_buffer("RMr", torch.zeros(h))
        self.register_buffer("RMi", torch.zeros(h))
        self.register_buffer("RVrr", torch.ones(h))
        self.register_buffer("RVri", torch.zeros(h))
        self.register_buffer("RVii", torch.ones(h))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        with torch.no_grad():
            self.Br.zero_()
            self.Bi.zero_()
            self.Wrr.fill_(1)
            self.Wri.uniform_(-.9, +.9)
            self.Wii.fill_(1)


class NavieComplexLSTM(nn.Module):
    """Parameter holder (tools_for_model.py:141-160)."""

    def __init__(self, input_size, hidden_size, projection_dim=None, bidirectional=False, batch_first=False):
        super().__init__()
        self.input_dim, self.rnn_units = input_size // 2, hidden_size // 2
        self.real_lstm = nn.LSTM(self.input_dim, self.rnn_units, num_layers=1, bidirectional=False, batch_first=False)
        self.imag_lstm = nn.LSTM(self.input_dim, self.rnn_units, num_layers=1, bidirectional=False, batch_first=False)
        if projection_dim is not None:
            self.projection_dim = projection_dim // 2
            self.r_trans = nn.Linear(self.rnn_units, self.projection_dim)
            self.i_trans = nn.Linear(self.rnn_units, self.projection_dim)
        else:
            self.projection_dim = None


# ------------------------------------------------------------------------------------------ device runtime
class _Runtime:
    """Plan + device arenas for one (B, L, training, device) combination."""

    def __init__(self, owner, B, L, training, device):
        self.plan = owner._make_plan(B, L, training)
        p = self.plan
        self.arenas = [None] * ARENA_COUNT
        self.arenas[ARENA_WS] = torch.zeros(max(p.arena_bytes[ARENA_WS], 256), dtype=torch.uint8, device=device)
        self.arenas[ARENA_CONST] = torch.from_numpy(p.const_image()).to(device)
        self.arenas[ARENA_IO] = torch.zeros(p.arena_bytes[ARENA_IO], dtype=torch.uint8, device=device)
        self.arenas[ARENA_PARAM] = owner._flat_param
        self.arenas[ARENA_GRAD] = owner._flat_grad
        self.arenas[ARENA_STATE] = owner._flat_state
        T, NF = p.T, p.NF
        self.wav = p.io(self.arenas, "wav", (B, L))
        self.out_wav = p.io(self.arenas, "out_wav", (B, L))
        self.out_real = p.io(self.arenas, "out_real", (B, NF, T))
        self.out_imag = p.io(self.arenas, "out_imag", (B, NF, T))
        self.g_wav = p.io(self.arenas, "grad_wav", (B, L))
        self.g_real = p.io(self.arenas, "grad_real", (B, NF, T))
        self.g_imag = p.io(self.arenas, "grad_imag", (B, NF, T))
        self.tgt = p.io(self.arenas, "tgt", (B, L)) if "io.tgt" in p.buffer_names() else None

    def run(self, phase):
        self.plan.run(phase, self.arenas, torch.cuda.current_stream().cuda_stream)


class _DCCRNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, rt, inputs, targets, *params):
        rt.wav.copy_(inputs)
        if rt.tgt is not None:
            rt.tgt.copy_(targets)
        rt.run(PHASE_FWD)
        rt.stamp = getattr(rt, "stamp", 0) + 1     # the saved activations live in the runtime's arena, not in ctx
        ctx.owner, ctx.rt, ctx.stamp = owner, rt, rt.stamp
        ctx.n = len(params)
        return rt.out_real.clone(), rt.out_imag.clone(), rt.out_wav.clone()

    @staticmethod
    def backward(ctx, g_real, g_imag, g_wav):
        rt, owner = ctx.rt, ctx.owner
        if ctx.stamp != rt.stamp:
            raise RuntimeError("sefd: a later forward of the same (batch, length, mode) overwrote the activations this backward needs; "
                               "call backward before the next forward of that shape")
        for dst, g in ((rt.g_real, g_real), (rt.g_imag, g_imag), (rt.g_wav, g_wav)):
            if g is None:
                dst.zero_()
            else:
                dst.copy_(g)
        rt.run(PHASE_BWD)
        flat = owner._flat_grad.clone()
        grads = [flat[off:off + n].view(shape) for (off, n, shape) in owner._param_slices]
        return (None, None, None, None) + tuple(grads)


# ------------------------------------------------------------------------------------------ shared machinery
class _SefdModule(nn.Module):
    """Flat parameter storage + per-(B, L) device runtimes shared by the HIP-backed models."""

    def _init_runtime_state(self):
        import weakref
        self._flat_param = self._flat_grad = self._flat_state = self._flat_nbt = None
        self._param_slices = None
        self._runtimes = {}
        ref = weakref.ref(self)
        for p in self.parameters():              # lets sefd_amd.optim.Adam(model.parameters()) find the model (optim.py)
            p._sefd_owner = ref

    def flatten_parameters(self):
        pass

    def _trainable(self):
        return [(n, p) for n, p in self.named_parameters()]

    def _bn_buffers(self):
        return [(n, b) for n, b in self.named_buffers() if n.endswith(("running_mean", "running_var", ".RMr", ".RMi", ".RVrr", ".RVri", ".RVii"))]

    def _flat_ok(self, device):
        fp = self._flat_param
        if fp is None or fp.device != device:
            return False
        ps = self._trainable()
        first, last = ps[0][1], ps[-1][1]
        off_last, n_last, _ = self._param_slices[-1]
        bn = self._bn_buffers()
        return (first.data_ptr() == fp.data_ptr() and last.data_ptr() == fp.data_ptr() + 4 * off_last
                and (not bn or bn[0][1].data_ptr() == self._flat_state.data_ptr()))

    def _flatten(self, device):
        """Re-home every parameter / BatchNorm buffer as a view of one flat fp32 tensor (the C ABI's arenas)."""
        ps = self._trainable()
        total = sum(p.numel() for _, p in ps)
        flat = torch.empty(total, dtype=torch.float32, device=device)
        slices, off = [], 0
        for _, p in ps:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[off:off + n].view(p.shape)
            slices.append((off, n, tuple(p.shape)))
            off += n
        bs = self._bn_buffers()
        st = torch.empty(max(sum(b.numel() for _, b in bs), 1), dtype=torch.float32, device=device)
        off = 0
        mods = dict(self.named_modules())
        for name, b in bs:
            n = b.numel()
            st[off:off + n].copy_(b.reshape(-1).to(device=device, dtype=torch.float32))
            mname, leaf = name.rsplit(".", 1)
            mods[mname]._buffers[leaf] = st[off:off + n].view(b.shape)
            off += n
        nbts = [(n, b) for n, b in self.named_buffers() if n.endswith("num_batches_tracked")]
        nbt = torch.zeros(len(nbts), dtype=torch.long, device=device)
        for i, (name, b) in enumerate(nbts):
            nbt[i] = b.to(device)
            mname, leaf = name.rsplit(".", 1)
            mods[mname]._buffers[leaf] = nbt[i]
        self._flat_param, self._flat_state, self._flat_nbt = flat, st, nbt
        self._flat_grad = torch.zeros_like(flat)
        self._param_slices = slices
        self._runtimes = {}

    def _runtime(self, B, L, device):
        if not self._flat_ok(device):
            self._flatten(device)
        key = (B, L, bool(self.training), self.masking_mode, self.act_dtype, getattr(self, "_bn_world", 1), getattr(self, "_grad_buckets", 1))
        rt = self._runtimes.get(key)
        if rt is None:
            rt = _Runtime(self, B, L, bool(self.training), device)
            rt.plan.owner = weakref.ref(self)            # running it makes it this model's `_status_plan` (guarded Adam, checkpoint check)
            # the plan's parameter table must agree with the module tree (names, order, sizes)
            names = [n for n, _ in self._trainable()]
            assert names == list(rt.plan.params.keys()), "parameter order differs from the plan"
            assert [s[0] for s in self._param_slices] == [v[0] for v in rt.plan.params.values()]
            self._runtimes[key] = rt
        return rt

    def get_params(self, weight_decay=0.0):
        weights, biases = [], []
        for name, param in self.named_parameters():
            (biases if 'bias' in name else weights).append(param)
        return [{'params': weights, 'weight_decay': weight_decay}, {'params': biases, 'weight_decay': 0.0}]

    def _main_loss(self, estimated, target):
        """models.py:315-323 / 558-565."""
        if cfg.loss == 'MSE':
            return tfl.mse(estimated, target)
        elif cfg.loss == 'SDR':
            return -tfl.sdr(target, estimated)
        elif cfg.loss == 'SI-SNR':
            return -(tfl.si_snr(estimated, target))
        elif cfg.loss == 'SI-SDR':
            return -(tfl.si_sdr(target, estimated))
        raise ValueError(cfg.loss)

    # ---- fused training step (what trainer.model_train does per batch, trainer.py:27-39) ---------------------
    def train_step(self, inputs, targets, optimizer, loss_kind=None, exchange=None, perceptual=None):
        """forward -> loss -> backward -> Adam with no autograd bookkeeping over the network; returns the loss as a 0-d device tensor.
        perceptual = 'LMS' | 'PMSQE': the step of model_perceptual_train (trainer.py:45-82), loss = (main + perceptual) / 2 (DCCRN only:
        CRN + perceptual crashes in the reference)."""
        from .optim import Adam
        if perceptual and not hasattr(self, "_stft_ref"):
            raise NotImplementedError("perceptual losses are defined for DCCRN only (models.py:303-314)")
        if perceptual not in (None, False, "LMS", "PMSQE"):
            raise ValueError(f"unknown perceptual loss {perceptual!r}")
        if not isinstance(optimizer, Adam):
            raise TypeError("train_step needs sefd_amd.optim.Adam (flat fused Adam)")
        kind = tfl.LOSS_KINDS[loss_kind or cfg.loss]
        inputs = inputs.float()
        targets = targets.float().contiguous()
        B, L = inputs.shape
        # SyncBN (exchange.sync_bn): a plan whose BatchNorm counts are global and whose phases are run in the op ranges
        # between its sync points, with the small statistics buffers all-reduced in between (parity with the reference's
        # single-process big batch).  Default: per-rank statistics, whole phases (two-lane backward).
        sync = exchange is not None and exchange.active and getattr(exchange, "sync_bn", False)
        self._bn_world = exchange.world if sync else 1
        # data parallel: plans with two gradient buckets (decoder + LSTM complete before the encoder backward, see ddp.py)
        self._grad_buckets = 2 if (exchange is not None and exchange.active and not sync) else 1
        rt = self._runtime(B, L, inputs.device)
        rt.plan.tolerate_fault = exchange is not None and exchange.active       # a faulty rank keeps its collectives matched; all ranks raise at the guard check
        optimizer.bind(self)
        self._flat_nbt += 1
        stream = torch.cuda.current_stream().cuda_stream
        rt.wav.copy_(inputs)
        if rt.tgt is not None:
            rt.tgt.copy_(targets)
        # the loss of this step reads the waveform only: no [B, NF, T] spectrum copies, no pass over their (zero) gradients (SEFD_RUN_WAVE_ONLY)
        wave_only = RUN_WAVE_ONLY if (not perceptual and not sync and rt.plan.model_name == "DCCRN" and not str(rt.plan.masking_mode).startswith("Direct")) else 0
        if sync:
            rt.plan.run_synced(PHASE_FWD, rt.arenas, stream, exchange.all_reduce_stats)
        elif wave_only:
            rt.plan.run_cb(PHASE_FWD, rt.arenas, stream, -1, None, flags=wave_only)
        else:
            rt.run(PHASE_FWD)
        # sharded batch: SI-SDR's mean of ratios inside the log (tools_for_loss.py:91-94) is taken over all ranks (two floats all-reduced
        # between the loss kernels); the other losses shard exactly
        direct = str(rt.plan.masking_mode).startswith("Direct")
        prev_dp = tfl.set_data_parallel(exchange)
        try:
            if direct:
                # spectral mapping (dccrn_direct_train / crn_direct_train, trainer.py:121-181): the loss compares SPECTRA, over the rows [B * F] of
                # length T (the reductions of tools_for_loss.py run over the last axis); the waveform output carries no loss
                if perceptual:
                    raise NotImplementedError("the reference has no perceptual trainer for the direct-mapping models")
                NFr, Tr = rt.out_real.shape[1], rt.out_real.shape[2]
                rows = lambda t: t.view(B * NFr, Tr)
                if rt.plan.model_name == "DCCRN":             # loss = (loss(real) + loss(imag)) / 2 against the target's spectra
                    tgt_r, tgt_i = self._stft_ref(targets)
                    half = torch.full((1,), 0.5, dtype=torch.float32, device=inputs.device)
                    ws_r, loss_r = tfl.loss_forward_raw(kind, rows(rt.out_real), rows(tgt_r), stream)
                    ws_i, loss_i = tfl.loss_forward_raw(kind, rows(rt.out_imag), rows(tgt_i), stream)
                    tfl.loss_backward_raw(kind, rows(rt.out_real), rows(tgt_r), ws_r, half, rows(rt.g_real), stream)
                    tfl.loss_backward_raw(kind, rows(rt.out_imag), rows(tgt_i), ws_i, half, rows(rt.g_imag), stream)
                    loss = (loss_r + loss_i) / 2
                else:                                         # CRN: mapped magnitudes (first output) against the target magnitudes (second output)
                    est_m, tgt_m = rows(rt.out_real), rows(rt.out_imag)
                    ws, loss = tfl.loss_forward_raw(kind, est_m, tgt_m, stream)
                    tfl.loss_backward_raw(kind, est_m, tgt_m, ws, None, rows(rt.g_real), stream)
                    rt.g_imag.zero_()
                rt.g_wav.zero_()
            else:
                ws, loss = tfl.loss_forward_raw(kind, rt.out_wav, targets, stream)
        finally:
            tfl.set_data_parallel(prev_dp)
        if not wave_only and not direct:
            rt.g_real.zero_()
            rt.g_imag.zero_()
        if direct:
            pass
        elif perceptual:
            half = torch.full((1,), 0.5, dtype=torch.float32, device=inputs.device)
            tfl.loss_backward_raw(kind, rt.out_wav, targets, ws, half, rt.g_wav, stream)
            with torch.enable_grad():                    # the loss kernels' own autograd wrappers on leaf copies of the outputs
                if perceptual == "PMSQE":
                    est = rt.out_wav.detach().requires_grad_()
                    perc = tfl.get_array_pmsqe_loss(targets, est)
                    (perc * 0.5).backward()
                    rt.g_wav.add_(est.grad)
                else:
                    clean_real, clean_imag = self._stft_ref(targets)
                    er, ei = rt.out_real.detach().requires_grad_(), rt.out_imag.detach().requires_grad_()
                    perc = tfl.lms_from_spectra(clean_real, clean_imag, er, ei)
                    (perc * 0.5).backward()
                    rt.g_real.copy_(er.grad)
                    rt.g_imag.copy_(ei.grad)
            self._last_loss_parts = (loss, perc.detach())       # (main, perceptual): what model_perceptual_train logs
            loss = (loss + perc.detach()) / 2
        else:
            tfl.loss_backward_raw(kind, rt.out_wav, targets, ws, None, rt.g_wav, stream)
        bucket = rt.plan.grad_bucket() if self._grad_buckets == 2 else None
        if sync:
            rt.plan.run_synced(PHASE_BWD, rt.arenas, stream, exchange.all_reduce_stats)
        elif bucket is not None:
            op, lo = bucket                                  # decoder + LSTM gradients are final at op `op`: their all-reduce starts
            rt.plan.run_cb(PHASE_BWD, rt.arenas, stream, op, lambda: exchange.begin(self._flat_grad[lo:]), flags=wave_only)   # under the encoder backward
        elif wave_only:
            rt.plan.run_cb(PHASE_BWD, rt.arenas, stream, -1, None, flags=wave_only)
        else:
            rt.run(PHASE_BWD)
        if exchange is not None and exchange.active:         # DDP: sum gradients over ranks (RCCL), average inside Adam
            # a rank whose plan gave up poisons an element of the LAST bucket: every rank's Adam then skips this step (optim.step_flat)
            optimizer.nan_guard = self._dp_guard = self._flat_grad[0:1]
            rt.plan.status_poison(optimizer.nan_guard, stream)
            if bucket is not None:
                exchange.begin(self._flat_grad[:bucket[1]])
                exchange.finish(self._flat_grad)
            else:
                exchange.all_reduce(self._flat_grad)
            optimizer.grad_scale = exchange.grad_scale
        optimizer.step_flat()
        if exchange is not None and exchange.active:
            _dp_guard_tick(self)
        return loss


DP_GUARD_EVERY = 16          # data parallel: the all-reduced guard element is looked at on the host every this many steps (one 4-byte read behind a sync)


def _check_dp_guard(model):
    """Data parallel: raises on EVERY rank (the guard element went through the sum all-reduce, so all ranks see the same value at the same step) when
    a rank's plan gave up or the gradient itself diverged at that element; replicas skipped those updates (sefd_adam_step_guarded_dp)."""
    guard = getattr(model, "_dp_guard", None)
    if guard is None or bool(torch.isfinite(guard).all()):
        return
    plan = getattr(model, "_status_plan", None)
    mine = plan is not None and plan.status() != 0
    raise RuntimeError(("this rank's plan gave up" + plan._RC5) if mine else
                       "another rank's plan gave up, or the gradient diverged (non-finite guard element of the all-reduced gradient); every replica skipped "
                       "the update of that step - no rank is left waiting in a collective")


def _dp_guard_tick(model):
    model._dp_steps = getattr(model, "_dp_steps", 0) + 1
    if model._dp_steps % DP_GUARD_EVERY == 0:
        _check_dp_guard(model)


# ------------------------------------------------------------------------------------------ the models
class DCCRN(_SefdModule):
    """Same constructor as the reference (models.py:17-28)."""

    def __init__(self, rnn_layers=cfg.rnn_layers, rnn_units=cfg.rnn_units, win_len=cfg.win_len, win_inc=cfg.win_inc,
                 fft_len=cfg.fft_len, win_type=cfg.window, masking_mode=cfg.masking_mode, use_cbn=False, kernel_size=5):
        super().__init__()
        self.use_cbn = bool(use_cbn)
        norm = ComplexBatchNorm if use_cbn else nn.BatchNorm2d         # models.py:76, 120
        self.win_len, self.win_inc, self.fft_len, self.win_type = win_len, win_inc, fft_len, win_type
        self.rnn_units = rnn_units
        self.input_dim = self.output_dim = win_len
        self.hidden_layers = rnn_layers
        self.kernel_size = kernel_size
        self.kernel_num = [2] + list(cfg.dccrn_kernel_num)
        self.masking_mode = masking_mode
        self.fix = True
        self.act_dtype = cfg.act_dtype
        self._lstm_kind = cfg.lstm
        self._skip = bool(cfg.skip_type)
        self.stft = ConvSTFT(win_len, win_inc, fft_len, win_type, 'complex')
        self.istft = ConviSTFT(win_len, win_inc, fft_len, win_type, 'complex')
        self.encoder = nn.ModuleList()
        self.decoder = nn.ModuleList()
        kn = self.kernel_num
        for idx in range(len(kn) - 1):
            self.encoder.append(nn.Sequential(
                ComplexConv2d(kn[idx], kn[idx + 1], kernel_size=(kernel_size, 2), stride=(2, 1), padding=(2, 1)),
                norm(kn[idx + 1]), nn.PReLU()))
        hidden_dim = fft_len // (2 ** len(kn))
        if cfg.lstm == 'complex':
            rnns = []
            for idx in range(rnn_layers):
                rnns.append(NavieComplexLSTM(
                    input_size=hidden_dim * kn[-1] if idx == 0 else rnn_units, hidden_size=rnn_units,
                    projection_dim=hidden_dim * kn[-1] if idx == rnn_layers - 1 else None))
            self.enhance = nn.Sequential(*rnns)  # registered here: the reference assigns it before the decoder is filled
        else:                                    # models.py:96-105: two real layers over all features + `tranform` (sic)
            self.enhance = nn.LSTM(input_size=hidden_dim * kn[-1], hidden_size=rnn_units, num_layers=2, dropout=0.0,
                                   bidirectional=False, batch_first=False)
            self.tranform = nn.Linear(rnn_units, hidden_dim * kn[-1])
        mult = 2 if cfg.skip_type else 1
        for idx in range(len(kn) - 1, 0, -1):
            mods = [ComplexConvTranspose2d(kn[idx] * mult, kn[idx - 1], kernel_size=(kernel_size, 2), stride=(2, 1),
                                           padding=(2, 0), output_padding=(1, 0))]
            if idx != 1:
                mods += [norm(kn[idx - 1]), nn.PReLU()]
            self.decoder.append(nn.Sequential(*mods))
        # state_dict order of the reference: stft, istft, encoder, decoder, enhance
        enh = self._modules.pop('enhance')
        self._modules['enhance'] = enh
        if 'tranform' in self._modules:
            self._modules['tranform'] = self._modules.pop('tranform')
        self._init_runtime_state()

    def _make_plan(self, B, L, training):
        return Plan(B, L, kernel_num=tuple(self.kernel_num[1:]), rnn_layers=self.hidden_layers, rnn_units=self.rnn_units,
                    win_len=self.win_len, win_inc=self.win_inc, fft_len=self.fft_len, masking_mode=self.masking_mode,
                    lstm=self._lstm_kind, skip_type=self._skip, act_dtype=self.act_dtype, training=training, model="DCCRN",
                    bn_world=getattr(self, "_bn_world", 1), grad_buckets=getattr(self, "_grad_buckets", 1), use_cbn=self.use_cbn, win_type=self.win_type)

    # ---- reference surface ----------------------------------------------------------------------------------
    def forward(self, inputs, targets=0):
        if not inputs.is_cuda:
            raise RuntimeError("sefd DCCRN runs on the MI355X only (inputs must be a cuda tensor); there is no CPU fallback")
        inputs = inputs.float().contiguous()
        B, L = inputs.shape
        rt = self._runtime(B, L, inputs.device)
        if self.training:
            self._flat_nbt += 1
        params = [p for _, p in self._trainable()]
        out_real, out_imag, out_wav = _DCCRNFunction.apply(self, rt, inputs, None, *params)
        if self.masking_mode == 'Direct(None make)':            # spectral mapping (models.py:232-250): spectra of the target too
            target_real, target_imag = self._stft_ref(targets)
            return out_real, target_real, out_imag, target_imag, out_wav
        return out_real, out_imag, out_wav

    def loss(self, estimated, target, real_spec=0, img_spec=0, perceptual=False):
        """models.py:303-323."""
        if perceptual:
            if cfg.perceptual == 'LMS':
                clean_real, clean_imag = self._stft_ref(target)          # self.stft(target), models.py:306-308
                return tfl.lms_from_spectra(clean_real, clean_imag, real_spec, img_spec)
            if cfg.perceptual == 'PMSQE':
                return tfl.get_array_pmsqe_loss(target, estimated)       # models.py:313-314 (third-party arithmetic: parity unpinned)
            raise ValueError(f"unknown cfg.perceptual {cfg.perceptual!r}")
        return self._main_loss(estimated, target)

    def _stft_ref(self, wav):
        """ConvSTFT 'complex' of a waveform batch in the reference layout: ([B,257,T] real, [B,257,T] imag), no grad."""
        wav = wav.detach().float().contiguous()
        B, L = wav.shape
        key = ("stft", B, L, str(wav.device), str(self.win_type))
        fe = self._runtimes.get(key)
        if fe is None:
            # the reference forms target / clean spectra with self.stft, i.e. the MODEL's window (models.py:237, 306-308): round 5 built this plan
            # with the default Hann window whatever win_type was - silently wrong losses for 'hamming' / None (goldens dccrn_hamming_direct_mse, _lms)
            plan = Plan(B, L, win_len=self.win_len, win_inc=self.win_inc, fft_len=self.fft_len, model="STFT", win_type=self.win_type)
            fe = (plan, plan.alloc_arenas(wav.device))
            self._runtimes[key] = fe
        plan, ar = fe
        plan.io(ar, "wav", (B, L)).copy_(wav)
        plan.run(PHASE_FWD, ar, torch.cuda.current_stream().cuda_stream)
        return plan.io(ar, "out_real", (B, plan.NF, plan.T)).clone(), plan.io(ar, "out_imag", (B, plan.NF, plan.T)).clone()


class RealConv2d(nn.Module):
    """Parameter holder (tools_for_model.py:341-386)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding=[padding[0], 0])
        nn.init.normal_(self.conv.weight.data, std=0.05)
        nn.init.constant_(self.conv.bias, 0.)


class RealConvTranspose2d(nn.Module):
    """Parameter holder (tools_for_model.py:389-425)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, output_padding):
        super().__init__()
        self.conv = nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride, padding=padding, output_padding=output_padding)
        nn.init.normal_(self.conv.weight.data, std=0.05)
        nn.init.constant_(self.conv.bias, 0.)


class CRN(_SefdModule):
    """Same constructor as the reference (models.py:330-341).  `rnn_layers` is ignored there too (one LSTM layer, Q13)."""

    def __init__(self, rnn_layers=cfg.rnn_layers, rnn_input_size=cfg.rnn_input_size, rnn_units=cfg.rnn_units, win_len=cfg.win_len,
                 win_inc=cfg.win_inc, fft_len=cfg.fft_len, win_type=cfg.window, masking_mode=cfg.masking_mode, kernel_size=5):
        super().__init__()
        self.win_len, self.win_inc, self.fft_len, self.win_type = win_len, win_inc, fft_len, win_type
        self.rnn_input_size = rnn_input_size
        self.rnn_units = rnn_units // 2
        self.input_dim = self.output_dim = win_len
        self.hidden_layers = rnn_layers
        self.kernel_size = kernel_size
        self.kernel_num = [2] + list(cfg.dccrn_kernel_num)
        self.masking_mode = masking_mode
        self.act_dtype = cfg.act_dtype
        self._skip = bool(cfg.skip_type)
        kn = self.kernel_num
        self.stft = ConvSTFT(win_len, win_inc, fft_len, win_type, 'real')
        self.istft = ConviSTFT(win_len, win_inc, fft_len, win_type, 'complex')
        self.encoder = nn.ModuleList()
        self.decoder = nn.ModuleList()
        for idx in range(len(kn) - 1):
            self.encoder.append(nn.Sequential(
                RealConv2d(kn[idx] // 2, kn[idx + 1] // 2, kernel_size=(kernel_size, 2), stride=(2, 1), padding=(2, 1)),
                nn.BatchNorm2d(kn[idx + 1] // 2), nn.PReLU()))
        hidden_dim = fft_len // (2 ** len(kn))
        if hidden_dim * (kn[-1] // 2) != rnn_input_size:
            raise ValueError("rnn_input_size must equal hidden_dim * kernel_num[-1] // 2 (SURVEY Q13)")
        self.enhance = nn.LSTM(input_size=rnn_input_size, hidden_size=self.rnn_units, dropout=0.0, bidirectional=False, batch_first=False)
        self.tranform = nn.Linear(self.rnn_units, rnn_input_size)
        if not cfg.skip_type:
            # the reference's own forward fails there: its full-width decoder (models.py:432-462) expects kernel_num[idx] input channels but is fed
            # kernel_num[idx] // 2 ("expected input[1, 128, 4, 43] to have 256 channels", checked in the build container) - nothing to mirror
            raise NotImplementedError("CRN with cfg.skip_type = False: the reference's own forward raises a channel mismatch (models.py:432-462, 490-493)")
        for idx in range(len(kn) - 1, 0, -1):
            mods = [RealConvTranspose2d(kn[idx], kn[idx - 1] // 2, kernel_size=(kernel_size, 2), stride=(2, 1), padding=(2, 0),
                                        output_padding=(1, 0))]
            if idx != 1:
                mods += [nn.BatchNorm2d(kn[idx - 1] // 2), nn.PReLU()]
            self.decoder.append(nn.Sequential(*mods))
        # reference registration order: stft, istft, encoder, decoder, enhance, tranform
        for name in ("enhance", "tranform"):
            self._modules[name] = self._modules.pop(name)
        self._init_runtime_state()

    def _make_plan(self, B, L, training):
        # models.py:506-532: 'Direct(None make)' = spectral mapping; any other cfg.masking_mode = tanh magnitude mask
        mode = self.masking_mode if self.masking_mode == 'Direct(None make)' else "E"
        return Plan(B, L, kernel_num=tuple(self.kernel_num[1:]), rnn_layers=1, rnn_units=2 * self.rnn_units, win_len=self.win_len,
                    win_inc=self.win_inc, fft_len=self.fft_len, masking_mode=mode, lstm="real", skip_type=self._skip,
                    act_dtype=self.act_dtype, training=training, model="CRN", bn_world=getattr(self, "_bn_world", 1), win_type=self.win_type)

    def forward(self, inputs, targets=0):
        """models.py:467-532: returns (est_mags, target_mags, out_wav).  est_mags (the network's magnitude output: mask x noisy
        magnitude, or the mapped magnitude in 'Direct(None make)' mode) and out_wav carry gradient; target_mags does not."""
        if not torch.is_tensor(targets):
            raise AttributeError("CRN.forward needs targets (the reference calls self.stft(targets) unconditionally, models.py:505)")
        if not inputs.is_cuda:
            raise RuntimeError("sefd CRN runs on the MI355X only (inputs must be a cuda tensor); there is no CPU fallback")
        inputs = inputs.float().contiguous()
        B, L = inputs.shape
        rt = self._runtime(B, L, inputs.device)
        if self.training:
            self._flat_nbt += 1
        params = [p for _, p in self._trainable()]
        est_mags, target_mags, out_wav = _DCCRNFunction.apply(self, rt, inputs, targets.float().contiguous(), *params)
        return est_mags, target_mags.detach(), out_wav

    def loss(self, estimated, target, out_mags=0, target_mags=0, perceptual=False):
        if perceptual:
            raise NotImplementedError("CRN + perceptual loss is unreachable in the reference (SURVEY Q10)")
        return self._main_loss(estimated, target)


# ------------------------------------------------------------------------------------------ FullSubNet (models.py:568-682)
class SequenceModel(nn.Module):
    """Parameter holder (tools_for_model.py:726-777): 2-layer nn.LSTM or nn.GRU (dropout=0.8) + Linear (+ activation)."""

    def __init__(self, input_size, output_size, hidden_size, num_layers, bidirectional, sequence_model="LSTM", output_activate_function="Tanh"):
        super().__init__()
        if sequence_model not in ("LSTM", "GRU") or bidirectional or num_layers != 2:
            raise NotImplementedError("only the 2-layer unidirectional LSTM / GRU SequenceModel is on the HIP path")
        rnn = nn.LSTM if sequence_model == "LSTM" else nn.GRU
        self.sequence_model = rnn(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers, batch_first=True,
                                  bidirectional=False, dropout=0.8)
        self.fc_output_layer = nn.Linear(hidden_size, output_size)
        self.output_activate_function = output_activate_function


def _weight_init(m):
    """BaseModel.weight_init (tools_for_model.py:1120-1184) for the module types a FullSubNet holds: Linear -> xavier_normal_ weight, normal_ bias;
    LSTM / GRU -> orthogonal_ matrices, normal_ vectors.  `Module.apply` visits children first, as the reference's call does: same draws in the same order."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight.data)
        nn.init.normal_(m.bias.data)
    elif isinstance(m, (nn.LSTM, nn.GRU)):
        for param in m.parameters():
            if len(param.shape) >= 2:
                nn.init.orthogonal_(param.data)
            else:
                nn.init.normal_(param.data)


class _FSNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, rt, noisy_mag, *params):
        plan, ar = rt
        B, F, T = noisy_mag.shape
        plan.io(ar, "mag", (B, F, T)).copy_(noisy_mag)
        plan.view(ar, "io.seed").view(torch.int32)[:1].add_(1)        # fresh dropout masks every forward (device-side counter)
        plan.run(PHASE_FWD, ar, torch.cuda.current_stream().cuda_stream)
        plan.stamp = getattr(plan, "stamp", 0) + 1
        ctx.owner, ctx.rt, ctx.shape, ctx.stamp = owner, rt, (B, F, T), plan.stamp
        return plan.io(ar, "crm", (B, F, T, 2)).clone()

    @staticmethod
    def backward(ctx, g):
        plan, ar = ctx.rt
        if ctx.stamp != plan.stamp:
            raise RuntimeError("sefd: a later forward of the same shape overwrote the activations this backward needs")
        B, F, T = ctx.shape
        plan.io(ar, "grad_crm", (B, F, T, 2)).copy_(g)
        plan.run(PHASE_BWD, ar, torch.cuda.current_stream().cuda_stream)
        flat = ctx.owner._flat_grad.clone()
        return (None, None, None) + tuple(flat[off:off + n].view(shape) for (off, n, shape) in ctx.owner._param_slices)


class FullSubNet(_SefdModule):
    """Same constructor as the reference (models.py:569-581)."""

    def __init__(self, sb_num_neighbors=cfg.sb_num_neighbors, fb_num_neighbors=cfg.fb_num_neighbors, num_freqs=cfg.num_freqs,
                 look_ahead=cfg.look_ahead, sequence_model=cfg.sequence_model, fb_output_activate_function=cfg.fb_output_activate_function,
                 sb_output_activate_function=cfg.sb_output_activate_function, fb_model_hidden_size=cfg.fb_model_hidden_size,
                 sb_model_hidden_size=cfg.sb_model_hidden_size, weight_init=cfg.weight_init, norm_type=cfg.norm_type):
        super().__init__()
        assert sequence_model in ("GRU", "LSTM"), f"{self.__class__.__name__} only support GRU and LSTM."
        from .plan import FSN_NORMS
        if norm_type not in FSN_NORMS:                # norm_wrapper (tools_for_model.py:1106-1118)
            raise NotImplementedError("You must set up a type of Norm. e.g. offline_laplace_norm, cumulative_laplace_norm, forgetting_norm, etc.")
        self.fb_model = SequenceModel(num_freqs, num_freqs, fb_model_hidden_size, 2, False, sequence_model, fb_output_activate_function)
        self.sb_model = SequenceModel((sb_num_neighbors * 2 + 1) + (fb_num_neighbors * 2 + 1), 2, sb_model_hidden_size, 2, False,
                                      sequence_model, sb_output_activate_function)
        self.sb_num_neighbors, self.fb_num_neighbors, self.look_ahead, self.num_freqs = sb_num_neighbors, fb_num_neighbors, look_ahead, num_freqs
        self._fsn = dict(sb_num_neighbors=sb_num_neighbors, fb_num_neighbors=fb_num_neighbors, look_ahead=look_ahead,
                         fb_hidden=fb_model_hidden_size, sb_hidden=sb_model_hidden_size, fb_act=fb_output_activate_function,
                         sb_act=sb_output_activate_function, sequence_model=sequence_model, norm_type=norm_type)
        if weight_init:                              # models.py:623-624: self.apply(self.weight_init)
            self.apply(_weight_init)
        self.dropout_keep = 0.2                      # nn.LSTM(dropout=0.8); tests set 1.0 to compare with the dropout-free goldens
        self.masking_mode, self.act_dtype = "cIRM", cfg.act_dtype
        self._init_runtime_state()

    def _fsn_runtime(self, B, T, device):
        if not self._flat_ok(device):
            self._flatten(device)
        keep = self.dropout_keep if self.training else 1.0
        nb = getattr(self, "_grad_buckets", 1)
        key = ("fsn", B, T, bool(self.training), keep, self.act_dtype, nb)
        rt = self._runtimes.get(key)
        if rt is None:
            plan = Plan(B, T, fft_len=2 * (self.num_freqs - 1), act_dtype=self.act_dtype, training=True, model="FullSubNet",
                        fsn=dict(self._fsn, keep=keep), grad_buckets=nb)
            plan.owner = weakref.ref(self)
            assert [n for n, _ in self._trainable()] == list(plan.params.keys()), "parameter order differs from the plan"
            ar = [None] * ARENA_COUNT
            ar[ARENA_WS] = torch.zeros(max(plan.arena_bytes[ARENA_WS], 256), dtype=torch.uint8, device=device)
            ar[ARENA_CONST] = torch.from_numpy(plan.const_image()).to(device)
            ar[ARENA_IO] = torch.zeros(plan.arena_bytes[ARENA_IO], dtype=torch.uint8, device=device)
            ar[ARENA_PARAM], ar[ARENA_GRAD], ar[ARENA_STATE] = self._flat_param, self._flat_grad, self._flat_state
            rt = (plan, ar)
            self._runtimes[key] = rt
        return rt

    def forward(self, noisy_mag):
        """models.py:626-672: noisy_mag [B, F, T] (or [B, 1, F, T]) -> cRM [B, F, T, 2]."""
        if noisy_mag.dim() == 4:
            assert noisy_mag.shape[1] == 1, f"{self.__class__.__name__} takes the mag feature as inputs."
            noisy_mag = noisy_mag[:, 0]
        if not noisy_mag.is_cuda:
            raise RuntimeError("sefd FullSubNet runs on the MI355X only (cuda tensors); there is no CPU fallback")
        noisy_mag = noisy_mag.detach().float().contiguous()
        B, F, T = noisy_mag.shape
        rt = self._fsn_runtime(B, T, noisy_mag.device)
        return _FSNFunction.apply(self, rt, noisy_mag, *[p for _, p in self._trainable()])

    def loss(self, estimated, target):
        """models.py:674-682: the four losses over the LAST axis (the real / imaginary pair of the [B, F, T, 2] masks)."""
        if cfg.loss == 'MSE':
            return tfl.mse(estimated.reshape(estimated.shape[0], -1), target.reshape(target.shape[0], -1))
        elif cfg.loss == 'SDR':
            return -tfl.sdr(target, estimated)
        elif cfg.loss == 'SI-SNR':
            return -tfl.si_snr(estimated, target)
        elif cfg.loss == 'SI-SDR':
            return -tfl.si_sdr(target, estimated)

    def train_step(self, inputs, targets, optimizer, loss_kind=None, exchange=None):
        """fullsubnet_train body (trainer.py:95-111), fused: stft x2 -> |.|, cIRM -> forward -> MSE -> backward -> Adam."""
        from . import tools_for_model as tools
        from .optim import Adam
        if not isinstance(optimizer, Adam):
            raise TypeError("train_step needs sefd_amd.optim.Adam (flat fused Adam)")
        noisy_complex, clean_complex = tools.stft(inputs), tools.stft(targets)
        noisy_mag, _, cirm = tools._targets(noisy_complex, clean_complex, True, False, True)
        B, F, T = noisy_mag.shape
        # data parallel: two gradient buckets - the full-band model's range is final while the sub-band weight gradients still run (ddp.py)
        self._grad_buckets = 2 if (exchange is not None and exchange.active) else 1
        plan, ar = self._fsn_runtime(B, T, noisy_mag.device)
        plan.tolerate_fault = exchange is not None and exchange.active
        optimizer.bind(self)
        stream = torch.cuda.current_stream().cuda_stream
        plan.io(ar, "mag", (B, F, T)).copy_(noisy_mag)
        plan.view(ar, "io.seed").view(torch.int32)[:1].add_(1)
        plan.run(PHASE_FWD, ar, stream)
        kind = tfl.LOSS_KINDS[loss_kind or cfg.loss]
        prev_dp = tfl.set_data_parallel(exchange)            # SI-SDR: mean of the row ratios over ALL ranks' rows inside the log
        try:
            if kind == 0:
                crm = plan.io(ar, "crm", (B, F * T * 2))
                ws, loss = tfl.loss_forward_raw(0, crm, cirm.view(B, -1), stream)
                tfl.loss_backward_raw(0, crm, cirm.view(B, -1), ws, None, plan.io(ar, "grad_crm", (B, F * T * 2)), stream)
            else:
                # model.loss(cIRM, cRM) (trainer.py:107): in all three the cIRM sits in the kernels' `est` role and the network output in
                # the `tgt` role (sdr(s1 = cRM, s2 = cIRM), si_snr(s1 = cIRM, s2 = cRM), si_sdr(reference = cRM, estimation = cIRM))
                crm = plan.io(ar, "crm", (B * F * T, 2))
                ws, loss = tfl.loss_rows_forward_raw(kind, cirm.view(-1, 2), crm, stream)
                tfl.loss_rows_backward_raw(kind, cirm.view(-1, 2), crm, ws, None, None, plan.io(ar, "grad_crm", (B * F * T, 2)), stream)
        finally:
            tfl.set_data_parallel(prev_dp)                   # an exception in the loss kernels must not leave the exchange set for a later validation loss
        bucket = plan.grad_bucket_range() if self._grad_buckets == 2 else None
        if bucket is not None:
            op, lo, hi = bucket
            plan.run_cb(PHASE_BWD, ar, stream, op, lambda: exchange.begin(self._flat_grad[lo:hi]))
        else:
            plan.run(PHASE_BWD, ar, stream)
        if exchange is not None and exchange.active:
            optimizer.nan_guard = self._dp_guard = self._flat_grad[-1:]          # in the last bucket, see _SefdModule.train_step
            plan.status_poison(optimizer.nan_guard, stream)
            if bucket is not None:
                for a, z in ((0, bucket[1]), (bucket[2], self._flat_grad.numel())):
                    if z > a:
                        exchange.begin(self._flat_grad[a:z])
                exchange.finish(self._flat_grad)
            else:
                exchange.all_reduce(self._flat_grad)
            optimizer.grad_scale = exchange.grad_scale
        optimizer.step_flat()
        if exchange is not None and exchange.active:
            _dp_guard_tick(self)
        return loss
