"""Fused Adam over the model's flat parameter buffer (one HIP kernel per step).

Semantics are torch.optim.Adam's as the reference configures it (algorithms/algorithms.py:474-478):
``Adam(params, lr, weight_decay)`` -> betas (0.9, 0.999), eps 1e-8, L2 decay added to the gradient,
no amsgrad.  Parameters without a gradient (the dead ``net0``/``net1`` tensors) are left untouched,
exactly as torch does for ``grad is None``."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not hasattr(model, "flat_params"):
            raise TypeError("FusedAdam needs a model with a flat parameter buffer (ST_GCN_model)")
        self.model = model
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__([p for _, p in model._named_live()], defaults)
        self._exp_avg = None
        self._exp_avg_sq = None
        self._steps = 0

    def _state_buffers(self):
        flat = self.model.flat_params
        if self._exp_avg is None or self._exp_avg.device != flat.device:
            old = (self._exp_avg, self._exp_avg_sq)
            self._exp_avg = torch.zeros_like(flat)
            self._exp_avg_sq = torch.zeros_like(flat)
            if old[0] is not None:                      # model moved after optimizer creation (trainer.py:96-98)
                self._exp_avg.copy_(old[0])
                self._exp_avg_sq.copy_(old[1])
        return self._exp_avg, self._exp_avg_sq

    def zero_grad(self, set_to_none: bool = True):
        for p in self.param_groups[0]["params"]:
            p.grad = None

    @torch.no_grad()
    def step(self, closure=None, grad_scale: float = 1.0, from_bucket: bool = False, guard=None):
        """``from_bucket=True``: the gradient is already in ``model.bucket`` (fused path);
        otherwise it is gathered from the parameters' ``.grad`` (autograd path).
        ``guard``: a one-element device tensor (the loss in the bucket); when it is not finite the kernel leaves parameters and
        moments untouched (the f16 range guard of ST_GCN's matrix-core training chain, see ``ST_GCN_model.guard_tensor``)."""
        model = self.model
        flat = model.flat_params
        if not flat.is_cuda:
            raise RuntimeError("FusedAdam runs on the HIP kernel only: move the model to a CUDA (ROCm) device")
        # parameters that never receive a gradient sit at one END of some flat buffers (RGCNU's second head at the tail, STNet's
        # thresholded 1x1 convolution at the head): torch.optim.Adam leaves `grad is None` parameters untouched -- no update, no
        # weight decay -- so the kernel runs over [start, end) only
        start, end = getattr(model, "optimized_range", (0, int(getattr(model, "num_optimized", flat.numel()))))
        n = end - start
        if not from_bucket:
            grads = [p.grad for p in self.param_groups[0]["params"]]
            live, off = [], 0
            for p_, g_ in zip(self.param_groups[0]["params"], grads):
                inside = start <= off < end
                if inside and g_ is None:
                    raise RuntimeError("FusedAdam.step(): a live parameter has no gradient")
                if inside:
                    live.append(g_.reshape(-1))
                off += p_.numel()
            model.bucket[start:end].copy_(torch.cat(live))
        m, v = self._state_buffers()
        g = self.param_groups[0]
        self._steps += 1
        state = getattr(model, "_step_state", None)
        o = 4 * start
        if state is not None:          # device-resident step (hipGraph-capturable, see graphs.py)
            if guard is not None:
                raise RuntimeError("FusedAdam.step(guard=...) with a device-resident step state: the guarded kernel takes the step count "
                                   "from the host; a captured graph runs the whole-step entry, which honours the guard itself")
            _lib.check(_lib.load().rulgnn_adam_step_dev_f32(
                flat.data_ptr() + o, model.bucket.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, n, state.data_ptr(),
                float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(grad_scale), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rulgnn_adam_step_dev_f32")
            return None
        if guard is not None:
            _lib.check(_lib.load().rulgnn_adam_step_guarded_f32(
                flat.data_ptr() + o, model.bucket.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, n, self._steps,
                float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(grad_scale), guard.data_ptr(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rulgnn_adam_step_guarded_f32")
            return None
        _lib.check(_lib.load().rulgnn_adam_step_f32(
            flat.data_ptr() + o, model.bucket.data_ptr() + o, m.data_ptr() + o, v.data_ptr() + o, n, self._steps,
            float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
            float(grad_scale), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rulgnn_adam_step_f32")
        return None

    def state_dict(self):
        m, v = self._state_buffers()
        return {"step": self._steps, "exp_avg": m.clone(), "exp_avg_sq": v.clone(),
                "param_groups": [{k: v_ for k, v_ in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd):
        m, v = self._state_buffers()
        m.copy_(sd["exp_avg"])
        v.copy_(sd["exp_avg_sq"])
        self._steps = int(sd["step"])
        for k, v_ in sd["param_groups"][0].items():
            self.param_groups[0][k] = v_
