"""Base class of the model modules: flat fp32 parameter storage, the data-parallel bucket, workspace caches and the Adam
argument block every family's C-ABI call takes.

A model module keeps the reference's ``nn.Module`` tree (same construction order => same RNG consumption => same initial
weights, same ``state_dict`` keys; the sub-modules only hold parameters) and runs on ONE flat device buffer:

  * ``_flat``       every parameter, in the order of the family's flat layout (include/rulgnn.h); the ``nn.Parameter``s are views
  * ``_grad_flat``  ``[gradient | loss | family extras]`` -- what the kernels write and one all-reduce carries (``bucket``)
  * ``_bufs``       per-batch-size workspaces (activations kept from forward to backward), a small LRU
  * BatchNorm statistics / counters of the families that have them live in ``_bn`` / ``_nbt`` (``_reflatten_buffers`` hook)

``nn.Module._apply`` (``.to()``, ``.float()``...) converts tensors one by one: ``_apply`` below rebuilds the views when that
happened and leaves everything in place when it was a no-op (the per-epoch ``model.to(device)`` of the trainers; captured
hipGraphs and the optimizer state point into the buffers)."""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib, params as PL


def current_stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatModule(nn.Module):
    bucket_tail = 1            # floats behind the gradient in ``_grad_flat``: the loss (+ what a family appends)
    flat_order = None          # parameter names in flat-layout order; None = ``named_parameters()`` order
    workspace_slots = 2        # batch sizes whose workspaces are kept (training batch + evaluation batch)

    # ---- construction ----------------------------------------------------------------------------------
    def _init_flat(self, layout=None, count=None):
        """Call at the end of ``__init__``, once the parameter-holding sub-modules exist.  ``layout`` (name -> (offset, shape), with
        ``count`` floats in all) is the family's own table where the flat order is not the ``named_parameters()`` order."""
        table = dict(self.named_parameters())
        if layout is not None:
            self._layout, self._slices = OrderedDict(layout), []
            for name, (off, shape) in self._layout.items():
                if tuple(table[name].shape) != tuple(shape):
                    raise RuntimeError(f"flat layout of '{name}' is {tuple(shape)}, the parameter is {tuple(table[name].shape)}")
                self._slices.append((off, table[name].numel(), tuple(shape)))
            self._count = int(count)
        else:
            names = list(self.flat_order) if self.flat_order is not None else list(table)
            self._layout, self._slices, off = OrderedDict(), [], 0
            for name in names:
                p = table[name]
                self._layout[name] = (off, tuple(p.shape))
                self._slices.append((off, p.numel(), tuple(p.shape)))
                off += p.numel()
            self._count = off
        self._flat = self._grad_flat = None
        self._bufs, self._pin_bufs, self._step_state = {}, False, None
        self._reflatten()

    def _named(self):
        table = dict(self.named_parameters())
        return [table[name] for name in self._layout]

    def _named_live(self):
        return list(zip(self._layout, self._named()))

    # ---- BatchNorm counters ----------------------------------------------------------------------------
    # num_batches_tracked of the families with BatchNorm: a fused step only counts (``_nbt_pending``); the int64 device tensor ``_nbt``
    # (the modules' buffers are views of it) is brought up to date when somebody looks (state_dict, a move)
    _nbt, _nbt_pending = None, 0

    def _track_batchnorm_counters(self):
        """Call in ``__init__`` before ``_init_flat``."""
        self._nbt_pending = 0
        self.register_state_dict_pre_hook(lambda module, prefix, keep_vars: module._flush_nbt())

    def _flush_nbt(self):
        if self._nbt_pending and self._nbt is not None:
            self._nbt += self._nbt_pending
            self._nbt_pending = 0

    # ---- flat storage ----------------------------------------------------------------------------------
    def _bucket_floats(self) -> int:
        return self._count + self.bucket_tail

    def _reflatten_buffers(self, dev):
        """Hook: move the family's BatchNorm statistics / counters into their flat buffers (``_bn``, ``_nbt``)."""

    def _reset_caches(self):
        self._bufs, self._step_state = {}, None

    def _reflatten(self):
        self._flush_nbt()
        ps = self._named()
        dev = ps[0].device
        flat = torch.empty(self._count, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, (off, n, shape) in zip(ps, self._slices):
                flat[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + n].view(shape)
        self._flat = flat
        self._reflatten_buffers(dev)
        self._grad_flat = torch.zeros(self._bucket_floats(), dtype=torch.float32, device=dev)
        self._reset_caches()
        PL.mark_flat_views(self)

    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        if not PL.flat_views_intact(self):      # a no-op .to(device) (every epoch in the trainers) keeps the buffers
            self._reflatten()                   # a real move converts tensors one by one: rebuild the flat views
        return self

    def _set_buffer(self, dotted, tensor):
        mod = self
        parts = dotted.split(".")
        for a in parts[:-1]:
            mod = getattr(mod, a)
        mod._buffers[parts[-1]] = tensor

    @property
    def flat_params(self):
        return self._flat

    @property
    def bucket(self):
        """[gradient | loss]: what one all-reduce carries in data-parallel training."""
        return self._grad_flat

    @property
    def num_live(self):
        return self._count

    # ---- C-ABI plumbing --------------------------------------------------------------------------------
    def _workspace_entry(self, key, nbytes, unsupported: str, make=None):
        """The cached ``(workspace bytes, prediction buffer, ...)`` of batch size ``key``; allocates (and evicts the oldest entry
        beyond ``workspace_slots`` unless ``_pin_bufs``) on a miss.  ``nbytes`` is a callable: the family's
        ``rulgnn_*_workspace_bytes`` (0 = configuration not covered -> RuntimeError(unsupported))."""
        ent = self._bufs.get(key)
        if ent is None:
            n = nbytes()
            if n == 0:
                raise RuntimeError(unsupported)
            if len(self._bufs) >= self.workspace_slots and not self._pin_bufs:
                self._bufs.pop(next(iter(self._bufs)))
            dev = self._flat.device
            ent = (torch.empty(n, dtype=torch.uint8, device=dev),) + (make(dev) if make is not None else
                                                                       (torch.empty(max(int(key), 1), dtype=torch.float32, device=dev),))
            self._bufs[key] = ent
        return ent

    def _adam_args(self, optimizer, bn=None):
        """``byref(rulgnn_adam_args)`` for a fused step with ``optimizer`` (optim.FusedAdam over this model), advancing its
        step count; None when the call should only produce gradients."""
        if optimizer is None:
            return None
        m, v = optimizer._state_buffers()
        optimizer._steps += 1
        g = optimizer.param_groups[0]
        return C.byref(_lib.AdamArgs(self._flat.data_ptr(), m.data_ptr(), v.data_ptr(), bn.data_ptr() if bn is not None else None,
                                     optimizer._steps, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                     float(g["weight_decay"]), 0.1,
                                     self._step_state.data_ptr() if self._step_state is not None else None))
