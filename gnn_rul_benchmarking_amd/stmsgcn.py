"""Drop-in ``STMSGCN_model`` whose forward/backward run in the gfx950 HIP kernels (csrc/stmsgcn.hip).

Mirrors the reference class (models/STMSGCN/Model.py:63-112): same constructor kwargs
``(num_patch, patch_size, interval, band_width, gcn_dims, gru_hidden_dim)``, same ``forward(x) -> [bs, 1]``,
the same 14 ``state_dict`` keys (``gcn_layers.{i}.linear.*``, ``gru_layer.gru.*_l0``, ``fc.*``) and -- because the
parameter-holding sub-modules are created in the reference's order -- the same initial weights for a torch seed.
None of the sub-modules is ever *called* (the GRU included): every parameter is a view into one flat fp32 buffer
that the kernels read directly (layout in include/rulgnn.h).

There is no CPU path: calling the model with a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream


class GCNLayer(nn.Module):
    """Holder of ``linear`` (Model.py:34-37)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)


class GRULayer(nn.Module):
    """Holder of ``gru`` = nn.GRU(input_dim, hidden_dim, 1, batch_first=True) (Model.py:52-55); only its four
    tensors are used."""

    def __init__(self, input_dim, hidden_dim, num_layers):
        super().__init__()
        self.gru = nn.GRU(input_dim, hidden_dim, num_layers, batch_first=True)


def param_layout(num_patch, gcn_dims, gru_hidden_dim):
    """state_dict name (without the algorithm's ``model.`` prefix) -> (offset, shape) in the flat buffer."""
    dims = [1] + list(gcn_dims)
    H, Cc = gru_hidden_dim, sum(dims)
    out, off = {}, 0

    def put(name, shape):
        nonlocal off
        n = 1
        for s in shape:
            n *= s
        out[name] = (off, tuple(shape))
        off += n

    for l in range(len(gcn_dims)):
        put(f"gcn_layers.{l}.linear.weight", (dims[l + 1], dims[l]))
        put(f"gcn_layers.{l}.linear.bias", (dims[l + 1],))
    put("gru_layer.gru.weight_ih_l0", (3 * H, Cc))
    put("gru_layer.gru.weight_hh_l0", (3 * H, H))
    put("gru_layer.gru.bias_ih_l0", (3 * H,))
    put("gru_layer.gru.bias_hh_l0", (3 * H,))
    put("fc.weight", (1, H * num_patch))
    put("fc.bias", (1,))
    return out, off


class _Function(torch.autograd.Function):
    """model(X) with autograd: forward = rulgnn_stmsgcn_forward_f32, backward = rulgnn_stmsgcn_backward_f32 with the
    incoming d(loss)/d(pred)."""

    @staticmethod
    def forward(ctx, model, x2d, *params):
        pred = model._forward(x2d)
        ctx.model, ctx.x2d = model, x2d
        return pred.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dpred):
        model = ctx.model
        grads = model._backward(ctx.x2d, dpred.contiguous().view(-1).float())
        out = [grads[off:off + n].view(shape).clone() for (off, n, shape) in model._slices]
        return (None, None, *out)


class STMSGCN_model(FlatModule):
    def __init__(self, num_patch, patch_size, interval, band_width, gcn_dims, gru_hidden_dim):
        super().__init__()
        self.num_patch, self.patch_size = int(num_patch), int(patch_size)
        self.interval, self.band_width = int(interval), int(band_width)
        if self.interval < 1 or self.interval >= self.patch_size or (self.patch_size - self.interval) % self.band_width:
            raise RuntimeError(f"shape '[{-1}, {self.band_width}]' is invalid for input of size {self.patch_size - self.interval}")
        dims = [1] + [int(d) for d in gcn_dims]
        self.gcn_dims = dims
        self.gru_hidden_dim = int(gru_hidden_dim)
        # same construction order as the reference => same RNG consumption => same initial weights
        self.gcn_layers = nn.ModuleList([GCNLayer(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.gru_layer = GRULayer(sum(dims), self.gru_hidden_dim, 1)
        self.fc = nn.Linear(self.gru_hidden_dim * self.num_patch, 1)

        self._loss = self._pred_buf = self._ws = None
        self._init_flat(*param_layout(self.num_patch, dims[1:], self.gru_hidden_dim))

    workspace_slots = 4

    def _reset_caches(self):
        super()._reset_caches()
        self._pred_buf = self._ws = None

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        s = _lib.StmsgcnShape()
        s.batch, s.num_patch, s.patch_size = batch, self.num_patch, self.patch_size
        s.interval, s.band_width = self.interval, self.band_width
        s.num_gcn_layers = len(self.gcn_dims) - 1
        if s.num_gcn_layers > _lib.STMSGCN_MAX_LAYERS:
            raise RuntimeError(f"STMSGCN kernels cover at most {_lib.STMSGCN_MAX_LAYERS} GCN layers")
        for i, d in enumerate(self.gcn_dims[1:]):
            s.gcn_dims[i] = d
        s.gru_hidden = self.gru_hidden_dim
        return s

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("STMSGCN_model runs on the HIP kernels only: input must be a CUDA (ROCm) tensor; "
                               "there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        bs = x.size(0)
        if x.numel() != bs * self.num_patch * self.patch_size:
            raise RuntimeError(f"shape '[{bs}, {self.num_patch}, {self.patch_size}]' is invalid for input of size {x.numel()}")
        return x.reshape(bs, self.num_patch * self.patch_size).contiguous().float()

    def _args(self, shp, x2d, y=None, dpred=None, global_batch=None):
        B = x2d.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_stmsgcn_workspace_bytes(C.byref(shp)),
                                    "STMSGCN kernels do not cover this configuration (nodes <= 32, patch_size <= 512, "
                                    "GCN widths <= 64 with sum <= 128, gru_hidden_dim <= 16, num_patch <= 4096)",
                                    make=lambda dev: (torch.empty(B, dtype=torch.float32, device=dev),))
        self._ws, self._pred_buf = ent
        a = _lib.StmsgcnArgs()
        a.x = x2d.data_ptr()
        a.y = y.data_ptr() if y is not None else None
        a.dpred = dpred.data_ptr() if dpred is not None else None
        a.params = self._flat.data_ptr()
        a.grads = self._grad_flat.data_ptr()
        a.pred = self._pred_buf.data_ptr()
        a.loss = self._grad_flat.data_ptr() + 4 * self._count
        a.workspace = self._ws.data_ptr()
        a.workspace_bytes = self._ws.numel()
        a.global_batch = B if global_batch is None else int(global_batch)
        return a

    def _forward(self, x2d):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d)
        _lib.check(_lib.load().rulgnn_stmsgcn_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stmsgcn_forward_f32")
        return self._pred_buf

    def _backward(self, x2d, dpred):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, dpred=dpred)
        _lib.check(_lib.load().rulgnn_stmsgcn_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stmsgcn_backward_f32")
        return self._grad_flat

    def features(self, x):
        """[bs*num_patch, nodes, sum(dims)]: the concatenated GCN features the reference hands to its GRU
        (Model.py:92-103, before the transpose)."""
        x2d = self._check_input(x)
        n = (self.patch_size - self.interval) // self.band_width
        out = torch.empty(x2d.size(0) * self.num_patch, n, sum(self.gcn_dims), dtype=torch.float32, device=x2d.device)
        shp = self._shape(x2d.size(0))
        _lib.check(_lib.load().rulgnn_stmsgcn_features_f32(C.byref(shp), x2d.data_ptr(), self._flat.data_ptr(), out.data_ptr(),
                                                           _stream()), "rulgnn_stmsgcn_features_f32")
        return out

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None):
        """forward + MSE + backward (+ Adam when ``optimizer`` is a FusedAdam over this model) in one C call; fills
        ``self.bucket`` = [grad | loss]; returns (pred [B], loss 0-d tensor) on the device, no host sync."""
        x2d = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x2d.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, y=yv, global_batch=global_batch)
        o = self._adam_args(optimizer)
        _lib.check(_lib.load().rulgnn_stmsgcn_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_stmsgcn_fwdbwd_f32")
        return self._pred_buf, self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, x):
        x2d = self._check_input(x)
        if x2d.size(0) == 0:
            return torch.empty(0, 1, dtype=torch.float32, device=x2d.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._named()):
            return _Function.apply(self, x2d, *self._named())
        return self._forward(x2d).clone().view(-1, 1)
