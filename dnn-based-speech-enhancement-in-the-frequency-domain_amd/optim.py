"""Fused Adam over the model's flat parameter buffer (torch.optim.Adam defaults: train_interface.py:59).

`Adam(model.parameters(), lr)` keeps the reference's construction line working; `step()` updates every parameter
with ONE kernel launch on the flat fp32 arena (`sefd_adam_step`), and exposes `state_dict()/load_state_dict()` in
torch.optim.Adam's format so reference checkpoints ({'model','optimizer','epoch'}) interchange.

Life cycle (the reference's resume order works unchanged, train_interface.py:52-59, 101-116): the optimizer finds the model
that owns its parameters by itself (the sefd models tag their parameters), flattens it if that has not happened yet, and a
`load_state_dict` that arrives before the model is on the GPU is kept and applied at bind time."""
import ctypes as C

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._model = None
        self._m = self._v = None
        self._step = 0
        self._pending = None         # a state_dict loaded before the model could be bound
        self.grad_scale = 1.0        # DDP: 1/world_size applied inside the kernel

    # ---- binding
    def _owner(self):
        for p in self.param_groups[0]["params"]:
            ref = getattr(p, "_sefd_owner", None)
            if ref is not None and ref() is not None:
                return ref()
        return None

    def bind(self, model=None):
        """Attach to the sefd model whose parameters are views of one flat buffer.  Called lazily by `train_step`, `step` and
        `load_state_dict`; flattens the model first if no forward has done so yet."""
        model = model if model is not None else (self._model if self._model is not None else self._owner())
        if model is None:
            raise RuntimeError("sefd_amd.optim.Adam: the parameters do not belong to a sefd_amd model")
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("sefd_amd.optim.Adam: move the model to the GPU first (model.to(DEVICE)); there is no CPU path")
        if not model._flat_ok(dev):
            model._flatten(dev)
        if self._model is model and self._m is not None and self._m.device == model._flat_param.device \
                and self._m.numel() == model._flat_param.numel():
            return
        self._model = model
        self._m = torch.zeros_like(model._flat_param)
        self._v = torch.zeros_like(model._flat_param)
        if self._pending is not None:
            sd, self._pending = self._pending, None
            self._apply_state(sd)

    def step_flat(self, grad=None):
        m = self._model
        g = m._flat_grad if grad is None else grad
        self._step += 1
        grp = self.param_groups[0]
        # guarded by the status word of the plan that produced the gradient: a kernel of that plan that gave up (cluster LSTM hand-over
        # timeout) set it earlier on this stream, and the update is then skipped on the device - no garbage step, no host sync here
        plan = getattr(m, "_status_plan", None)
        guard = plan.status_word() if plan is not None else None
        # data parallel (train_step sets nan_guard to the poisoned element of the last all-reduced bucket): EVERY rank skips a step that any
        # rank's plan gave up on - the replicas stay identical (Plan.status_poison)
        nan_guard, self.nan_guard = getattr(self, "nan_guard", None), None
        rc = _lib.lib().sefd_adam_step_guarded_dp(C.c_void_p(m._flat_param.data_ptr()), C.c_void_p(g.data_ptr()),
                                                  C.c_void_p(self._m.data_ptr()), C.c_void_p(self._v.data_ptr()),
                                                  m._flat_param.numel(), self._step, grp["lr"], grp["betas"][0], grp["betas"][1],
                                                  grp["eps"], self.grad_scale, C.c_void_p(guard) if guard else None,
                                                  C.c_void_p(nan_guard.data_ptr()) if nan_guard is not None else None,
                                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sefd_adam_step_guarded_dp failed ({rc})")

    @torch.no_grad()
    def step(self, closure=None):
        """Generic path (after `loss.backward()`): gathers p.grad into the flat gradient buffer, then the fused kernel.
        A parameter whose `.grad` is None is left untouched (value and moments), as torch.optim.Adam does."""
        self.bind()
        m = self._model
        skipped = []
        for p, (off, n, _) in zip([p for _, p in m._trainable()], m._param_slices):
            if p.grad is not None:
                m._flat_grad[off:off + n].copy_(p.grad.reshape(-1))
            else:
                m._flat_grad[off:off + n].zero_()
                skipped.append((off, n, m._flat_param[off:off + n].clone(), self._m[off:off + n].clone(), self._v[off:off + n].clone()))
        self.step_flat()
        for off, n, pv, mv, vv in skipped:
            m._flat_param[off:off + n].copy_(pv)
            self._m[off:off + n].copy_(mv)
            self._v[off:off + n].copy_(vv)

    # ---- torch.optim.Adam compatible checkpoint format
    def state_dict(self):
        # the param_group carries every key of the installed torch.optim.Adam's defaults (weight_decay 0, amsgrad False, ...) so
        # that the checkpoint also loads into the reference's optimizer class (train_interface.py:59, 110)
        tmpl = dict(torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))]).defaults)
        tmpl.update({k: v for k, v in self.param_groups[0].items() if k != "params"})
        sd = {"state": {}, "param_groups": [{**tmpl, "params": list(range(len(self.param_groups[0]["params"])))}]}
        if self._pending is not None and self._m is None:
            return self._pending
        if self._m is not None and self._step > 0:
            for i, (off, n, shape) in enumerate(self._model._param_slices):
                sd["state"][i] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[off:off + n].view(shape).clone(),
                                  "exp_avg_sq": self._v[off:off + n].view(shape).clone()}
        return sd

    def _apply_state(self, sd):
        if not sd["state"]:              # a checkpoint taken before the first step: back to a fresh optimizer
            self._m.zero_(); self._v.zero_(); self._step = 0
        for i, (off, n, shape) in enumerate(self._model._param_slices):
            st = sd["state"].get(i, sd["state"].get(str(i)))
            if st is None:
                continue
            self._m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            self._step = int(float(st["step"]))
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps"):
            if k in g:
                self.param_groups[0][k] = tuple(g[k]) if k == "betas" else g[k]

    def load_state_dict(self, sd):
        """train_interface.py:110: `optimizer.load_state_dict(checkpoint['optimizer'])` right after construction."""
        model = self._model if self._model is not None else self._owner()
        if model is not None and next(model.parameters()).device.type != "cuda":
            self._pending = sd           # model not on the GPU yet (the one deferrable case): applied by the first bind
            return
        self.bind()                      # anything else that fails here (foreign parameters, allocation, ...) is a failed resume: raise
        self._apply_state(sd)
