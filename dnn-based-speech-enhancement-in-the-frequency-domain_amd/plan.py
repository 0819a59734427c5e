"""Python handle on a `sefd_plan` (host-side op list + arena sizes) and its device arenas."""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib

ARENA_WS, ARENA_PARAM, ARENA_GRAD, ARENA_STATE, ARENA_CONST, ARENA_IO, ARENA_COUNT = 0, 1, 2, 3, 4, 5, 6
PHASE_FWD, PHASE_BWD = 0, 1
RUN_WAVE_ONLY = 1
MASK_MODES = {"E": 0, "C": 1, "R": 2, "Direct(None make)": 4}
DTYPES = {"fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}


FSN_NORMS = {"offline_laplace_norm": 0, "cumulative_laplace_norm": 1, "offline_gaussian_norm": 2, "cumulative_layer_norm": 3}   # tools_for_model.py:1106-1118


class Plan:
    def __init__(self, B, L, kernel_num=(32, 64, 128, 256, 256, 256), rnn_layers=2, rnn_units=256, win_len=400,
                 win_inc=100, fft_len=512, masking_mode="E", lstm="complex", skip_type=True, act_dtype="fp32",
                 kernel_size=5, training=True, model="DCCRN", fsn=None, bn_world=1, grad_buckets=1, use_cbn=False, win_type="hanning"):
        self.lib = _lib.lib()
        if masking_mode not in MASK_MODES:
            raise NotImplementedError(f"masking_mode {masking_mode!r} is not on the HIP path yet")
        cfg = _lib.ModelConfig()
        cfg.model = {"DCCRN": 0, "CRN": 1, "STFT": 2, "FullSubNet": 3, "TorchSTFT": 4, "TorchISTFT": 5}[model]
        cfg.B, cfg.L = int(B), int(L)
        if model == "FullSubNet":
            # L = number of STFT frames T; kernel_num carries (sb_neighbors, fb_neighbors, look_ahead, fb_hidden, sb_hidden,
            #                                                   fb_act, sb_act, dropout keep probability in 1/1000)
            f = dict(sb_num_neighbors=15, fb_num_neighbors=0, look_ahead=2, fb_hidden=512, sb_hidden=384, fb_act="ReLU",
                     sb_act=None, keep=0.2, sequence_model="LSTM", norm_type="offline_laplace_norm")
            f.update(fsn or {})
            acts = {None: 0, "None": 0, "ReLU": 1, "Tanh": 2, "ReLU6": 3}
            kernel_num = (f["sb_num_neighbors"], f["fb_num_neighbors"], f["look_ahead"], f["fb_hidden"], f["sb_hidden"],
                          acts[f["fb_act"]], acts[f["sb_act"]], int(round(f["keep"] * 1000)),
                          {"LSTM": 0, "GRU": 1}[f["sequence_model"]], FSN_NORMS[f["norm_type"]])
        cfg.win_len, cfg.hop, cfg.fft_len = win_len, win_inc, fft_len
        cfg.n_layers = len(kernel_num) if model != "FullSubNet" else 0
        for i, k in enumerate(kernel_num):
            cfg.kernel_num[i] = int(k)
        cfg.rnn_layers, cfg.rnn_units = rnn_layers, rnn_units
        cfg.mask_mode = MASK_MODES[masking_mode]
        cfg.lstm_complex = 1 if lstm == "complex" else 0
        cfg.skip = 1 if skip_type else 0
        cfg.act_dtype = DTYPES[act_dtype]
        cfg.kernel_size = kernel_size
        cfg.training = 1 if training else 0
        cfg.bn_world = int(bn_world)
        cfg.grad_buckets = int(grad_buckets)
        cfg.use_cbn = 1 if use_cbn else 0
        if win_type in (None, "None"):
            cfg.window = 1
        elif win_type in ("hanning", "hann"):
            cfg.window = 0
        else:                              # any other name scipy.signal.get_window takes (tools_for_model.py:19-20): the table goes to the planner
            from .frontend_consts import window_fn
            self._window_values = np.ascontiguousarray(window_fn(win_type, win_len), dtype=np.float64)
            cfg.window = 2
            cfg.window_values = self._window_values.ctypes.data_as(C.POINTER(C.c_double))
        self.cfg = cfg
        self.model_name = model
        self.masking_mode = masking_mode
        self.h = self.lib.sefd_plan_create(C.byref(cfg))
        err = self.lib.sefd_plan_error(self.h).decode()
        if err:
            self.lib.sefd_plan_destroy(self.h)
            self.h = None
            raise ValueError("sefd plan: " + err)
        self.B, self.L = int(B), int(L)
        self.T = self.lib.sefd_plan_frames(self.h)
        self.NF = fft_len // 2 + 1
        self.arena_bytes = [self.lib.sefd_plan_arena_bytes(self.h, a) for a in range(ARENA_COUNT)]
        self.params = self._param_table(0)
        self.state = self._param_table(1)
        self.n_param = sum(int(np.prod(s)) if len(s) else 1 for _, s in self.params.values()) if self.params else 0
        self.n_state = sum(int(np.prod(s)) if len(s) else 1 for _, s in self.state.values())

    def sync_points(self):
        """SyncBN (bn_world > 1): [(phase, op, arena, byte offset, element count, torch dtype)] in execution order."""
        out = []
        ph, op, ar, dt = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        off, cnt = C.c_int64(), C.c_int64()
        for i in range(self.lib.sefd_plan_num_syncs(self.h)):
            self.lib.sefd_plan_sync(self.h, i, C.byref(ph), C.byref(op), C.byref(ar), C.byref(off), C.byref(cnt), C.byref(dt))
            out.append((ph.value, op.value, ar.value, off.value, cnt.value, torch.float64 if dt.value else torch.float32))
        return out

    def run_synced(self, phase, arenas, stream, all_reduce):
        """Run a whole phase of a SyncBN plan: op ranges between the sync points, `all_reduce(tensor)` (in-place sum over
        the ranks) on each statistics buffer in between."""
        cur = 0
        for ph, op, ar, off, cnt, dtype in self.sync_points():
            if ph != phase:
                continue
            self.run(phase, arenas, stream, cur, op + 1)
            nbytes = cnt * (8 if dtype == torch.float64 else 4)
            all_reduce(arenas[ar].view(torch.uint8)[off:off + nbytes].view(dtype))
            cur = op + 1
        self.run(phase, arenas, stream, cur, self.num_ops(phase))

    def _param_table(self, kind):
        out = OrderedDict()
        shp = (C.c_int64 * 4)()
        for i in range(self.lib.sefd_plan_num_params(self.h, kind)):
            name = self.lib.sefd_plan_param_name(self.h, kind, i).decode()
            nd = self.lib.sefd_plan_param_shape(self.h, kind, i, shp)
            out[name] = (self.lib.sefd_plan_param_offset(self.h, kind, i), tuple(int(shp[k]) for k in range(nd)))
        return out

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.sefd_plan_destroy(self.h)
            self.h = None

    # ---- introspection
    def num_ops(self, phase):
        return self.lib.sefd_plan_num_ops(self.h, phase)

    def ops_ptr(self, phase):
        return self.lib.sefd_plan_ops(self.h, phase)

    def op_kinds(self, phase):
        n, sz = self.num_ops(phase), self.lib.sefd_op_size()
        raw = np.ctypeslib.as_array((C.c_int32 * (n * sz // 4)).from_address(self.ops_ptr(phase))).reshape(n, sz // 4)
        return raw[:, 0].copy(), raw[:, 1].copy()

    def op_info(self, phase, i):
        """dict(kind, tag, M, N, K, dtype, flops, bytes) of op i (RUNGEMM / WGRAD carry the GEMM geometry)."""
        o = (C.c_int64 * 8)()
        if self.lib.sefd_plan_op_info(self.h, phase, i, o) != 0:
            raise IndexError(i)
        return dict(kind=int(o[0]), tag=int(o[1]), M=int(o[2]), N=int(o[3]), K=int(o[4]), dtype=int(o[5]) & 0xff, flags=int(o[5]) >> 8,
                    flops=int(o[6]), bytes=int(o[7]))

    def buffer(self, name):
        a, off, nb, dt = C.c_int32(), C.c_int64(), C.c_int64(), C.c_int32()
        if self.lib.sefd_plan_buffer(self.h, name.encode(), C.byref(a), C.byref(off), C.byref(nb), C.byref(dt)) != 0:
            raise KeyError(name)
        return a.value, off.value, nb.value, dt.value

    def buffer_names(self):
        return [self.lib.sefd_plan_buffer_name(self.h, i).decode() for i in range(self.lib.sefd_plan_num_buffers(self.h))]

    def const_image(self) -> np.ndarray:
        n = self.arena_bytes[ARENA_CONST]
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(self.lib.sefd_plan_const_data(self.h))).copy()

    # ---- arenas
    def alloc_arenas(self, device):
        """uint8 tensors for every arena (PARAM/GRAD/STATE as float32). CONST is filled from the host image."""
        dev = torch.device(device)
        ar = [None] * ARENA_COUNT
        ar[ARENA_WS] = torch.zeros(max(self.arena_bytes[ARENA_WS], 256), dtype=torch.uint8, device=dev)
        ar[ARENA_PARAM] = torch.zeros(self.arena_bytes[ARENA_PARAM] // 4, dtype=torch.float32, device=dev)
        ar[ARENA_GRAD] = torch.zeros(self.arena_bytes[ARENA_GRAD] // 4, dtype=torch.float32, device=dev)
        ar[ARENA_STATE] = torch.zeros(self.arena_bytes[ARENA_STATE] // 4, dtype=torch.float32, device=dev)
        ar[ARENA_CONST] = torch.from_numpy(self.const_image()).to(dev)
        ar[ARENA_IO] = torch.zeros(self.arena_bytes[ARENA_IO], dtype=torch.uint8, device=dev)
        return ar

    def set_seed(self, arenas, seed):
        """FullSubNet dropout: (seed lo, seed hi) for the counter-based mask hash; advance it every step."""
        self.view(arenas, "io.seed").view(torch.int32)[:2].copy_(torch.tensor([seed & 0x7FFFFFFF, (seed >> 31) & 0x7FFFFFFF], dtype=torch.int32))

    def view(self, arenas, name):
        """Typed 1-D view of a named buffer inside its arena."""
        a, off, nb, dt = self.buffer(name)
        raw = arenas[a].view(torch.uint8)[off:off + nb]
        return raw.view(torch.bfloat16 if dt == 1 else torch.float32)

    def io(self, arenas, name, shape):
        return self.view(arenas, "io." + name).view(*shape)

    def grad_bucket(self):
        """(backward op index, first flat element) of the gradient bucket that is complete before the encoder backward, or None."""
        op, el = C.c_int32(), C.c_int64()
        if self.lib.sefd_plan_grad_bucket(self.h, C.byref(op), C.byref(el)) != 0:
            return None
        return op.value, el.value

    # ---- per-plan status word (a kernel that had to give up - the cluster LSTM's bounded hand-over waits - sets it)
    owner = None                 # weakref to the model this plan belongs to: running the plan makes it the model's `_status_plan`
    # Data parallel (set by train_step under an active exchange): a plan whose status word is set does NOT raise from run / run_cb - the rank keeps
    # taking part in every collective of the step (its poisoned gradient element makes all replicas skip the update) and ALL ranks raise together at the
    # next guard check (models._check_dp_guard).  Raising here, on one rank, mid-epoch, left the others blocked in their next all-reduce.
    tolerate_fault = False

    def _mark(self):
        o = self.owner() if self.owner is not None else None
        if o is not None:
            o._status_plan = self

    def status_word(self):
        """Address of the device copy of the word (sefd_adam_step_guarded's skip_if_set)."""
        return self.lib.sefd_plan_status_word(self.h)

    def status_poison(self, grad_elem, stream):
        """Data parallel: NaN into `grad_elem` (a one-element view of the LAST gradient bucket) if this plan's status word is set - the
        all-reduce then tells every rank (sefd_adam_step_guarded_dp's skip_if_nan)."""
        if self.lib.sefd_plan_status_poison(self.h, C.c_void_p(grad_elem.data_ptr()), C.c_void_p(stream)) != 0:
            raise RuntimeError("sefd_plan_status_poison failed")

    def status(self, clear=False):
        return int(self.lib.sefd_plan_status(self.h, 1 if clear else 0))

    def status_set(self):
        """Test hook: set the word from the host, as a kernel that gives up does from the device."""
        if self.lib.sefd_plan_status_set(self.h) != 0:
            raise RuntimeError("sefd_plan_status_set failed")

    _RC5 = (": a cluster-LSTM launch of this plan gave up waiting for a peer workgroup (GPU shared or preempted?) - that step's results "
            "are invalid; the guarded Adam left the parameters untouched (Plan.status(clear=True) re-arms the plan)")

    def grad_bucket_range(self):
        """(backward op index, lo, hi): flat gradient elements [lo, hi) are final once that op has run (None: single-bucket plan)."""
        op, lo, hi = C.c_int32(), C.c_int64(), C.c_int64()
        if self.lib.sefd_plan_grad_bucket_range(self.h, C.byref(op), C.byref(lo), C.byref(hi)) != 0:
            return None
        return op.value, lo.value, hi.value

    def run_cb(self, phase, arenas, stream, at, fn, flags=0):
        """Whole phase (two-lane schedule); `fn()` runs on the host right after op `at` has been enqueued (fn None: no callback).
        flags: RUN_WAVE_ONLY (include/sefd.h SEFD_RUN_WAVE_ONLY)."""
        self._mark()
        ptrs = (C.c_void_p * ARENA_COUNT)(*[C.c_void_p(a.data_ptr()) for a in arenas])
        cb = C.CFUNCTYPE(None, C.c_void_p)(lambda _ctx: fn()) if fn is not None else None
        rc = self.lib.sefd_plan_run_flags(self.h, phase, ptrs, C.c_void_p(stream), flags, at if fn is not None else -1, cb, None)
        if rc == -5 and self.tolerate_fault:
            # the status word was set before this run: nothing was launched and the callback was not called - but the OTHER ranks call theirs (the
            # first bucket's all-reduce), so this rank must too, or the collectives no longer match
            if fn is not None:
                fn()
            return
        if rc != 0:
            raise RuntimeError(f"sefd_plan_run_cb failed ({rc})" + (self._RC5 if rc == -5 else ""))

    def run_timed(self, phase, arenas, stream=0):
        """Whole phase in the real two-lane schedule with HIP events around every op (measurement): list of per-op milliseconds."""
        n = self.num_ops(phase)
        ms = (C.c_float * n)()
        ptrs = (C.c_void_p * ARENA_COUNT)(*[C.c_void_p(a.data_ptr()) for a in arenas])
        rc = self.lib.sefd_plan_run_timed(self.h, phase, ptrs, C.c_void_p(stream), ms, n)
        if rc != 0:
            raise RuntimeError(f"sefd_plan_run_timed failed ({rc})")
        return list(ms)

    def run(self, phase, arenas, stream=0, first=0, last=-1):
        self._mark()
        ptrs = (C.c_void_p * ARENA_COUNT)(*[C.c_void_p(a.data_ptr()) for a in arenas])
        rc = self.lib.sefd_plan_run(self.h, phase, first, last, ptrs, C.c_void_p(stream))
        if rc != 0 and not (rc == -5 and self.tolerate_fault):
            raise RuntimeError(f"sefd_plan_run failed ({rc})" + (self._RC5 if rc == -5 else ""))
