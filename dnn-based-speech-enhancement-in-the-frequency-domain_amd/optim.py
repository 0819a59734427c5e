"""Fused Adam over the model's flat parameter buffer (torch.optim.Adam defaults: train_interface.py:59).

`Adam(model.parameters(), lr)` keeps the reference's construction line working; `step()` updates every parameter
with ONE kernel launch on the flat fp32 arena (`sefd_adam_step`), and exposes `state_dict()/load_state_dict()` in
torch.optim.Adam's format so reference checkpoints ({'model','optimizer','epoch'}) interchange."""
import ctypes as C

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._model = None
        self._m = self._v = None
        self._step = 0
        self.grad_scale = 1.0        # DDP: 1/world_size applied inside the kernel

    def bind(self, model):
        """Attach to a sefd model whose parameters are views of one flat buffer (done lazily by train_step)."""
        if self._model is model and self._m is not None and self._m.device == model._flat_param.device \
                and self._m.numel() == model._flat_param.numel():
            return
        self._model = model
        self._m = torch.zeros_like(model._flat_param)
        self._v = torch.zeros_like(model._flat_param)

    def step_flat(self, grad=None):
        m = self._model
        g = m._flat_grad if grad is None else grad
        self._step += 1
        grp = self.param_groups[0]
        rc = _lib.lib().sefd_adam_step(C.c_void_p(m._flat_param.data_ptr()), C.c_void_p(g.data_ptr()),
                                       C.c_void_p(self._m.data_ptr()), C.c_void_p(self._v.data_ptr()),
                                       m._flat_param.numel(), self._step, grp["lr"], grp["betas"][0], grp["betas"][1],
                                       grp["eps"], self.grad_scale, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(f"sefd_adam_step failed ({rc})")

    @torch.no_grad()
    def step(self, closure=None):
        """Generic path (after `loss.backward()`): gathers p.grad into the flat gradient buffer, then the fused kernel."""
        if self._model is None:
            raise RuntimeError("call optimizer.bind(model) once (model.train_step does it) before step()")
        m = self._model
        for p, (off, n, _) in zip([p for _, p in m._trainable()], m._param_slices):
            if p.grad is not None:
                m._flat_grad[off:off + n].copy_(p.grad.reshape(-1))
            else:
                m._flat_grad[off:off + n].zero_()
        self.step_flat()

    # ---- torch.optim.Adam compatible checkpoint format
    def state_dict(self):
        sd = {"state": {}, "param_groups": [{**{k: v for k, v in self.param_groups[0].items() if k != "params"},
                                             "params": list(range(len(self.param_groups[0]["params"])))}]}
        if self._m is not None:
            for i, (off, n, shape) in enumerate(self._model._param_slices):
                sd["state"][i] = {"step": torch.tensor(float(self._step)), "exp_avg": self._m[off:off + n].view(shape).clone(),
                                  "exp_avg_sq": self._v[off:off + n].view(shape).clone()}
        return sd

    def load_state_dict(self, sd):
        if self._model is None:
            raise RuntimeError("bind(model) before load_state_dict")
        for i, (off, n, shape) in enumerate(self._model._param_slices):
            st = sd["state"].get(i)
            if st is None:
                continue
            self._m[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self._v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            self._step = int(float(st["step"]))
        g = sd["param_groups"][0]
        self.param_groups[0]["lr"] = g.get("lr", self.param_groups[0]["lr"])
