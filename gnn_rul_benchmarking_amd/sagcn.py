"""Drop-in ``SAGCN_model``.  The whole model runs behind three C entries on one flat parameter buffer
(``rulgnn_sagcn_{forward,backward,fwdbwd}_f32``; ``fused_mse_step`` is forward + MSE + backward + Adam in one call): the 40
hand-crafted statistics of every patch, the cosine adjacency and its normalised aggregation in the gfx950 kernels of csrc/sagcn.hip,
every Linear layer -- over the node axis or the feature axis -- as a matrix-core GEMM on node-major activations.

Mirrors the reference class (models/SAGCN/Model.py:127-156): same constructor kwargs ``(num_patch, patch_size, gcn_hidden_dim,
attention_hidden_dim)``, ``forward(x) -> [bs, 1]``, the same 16 ``state_dict`` keys in the same order and -- sub-modules being created
in the reference's order -- the same initial weights for a torch seed.  The statistics depend on the input alone: forward-only.
There is no CPU path: a non-CUDA input raises.

One behaviour is pinned where the reference leaves it open: the statistic ``median_freq`` indexes the spectrum through an UNSTABLE
``torch.argsort`` of a power spectrum that is mirrored exactly (every value but DC / Nyquist appears twice); here equal powers keep
their bin order (a stable sort), which is what the reference's CPU sort produces for patches of up to 16 points (its PHM2012
Condition_1 row) -- beyond that the reference's own CPU and GPU sorts disagree with each other on the sign of that one feature.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream


class GCNLayer(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)


class GraphProjectionLayer(nn.Module):
    def __init__(self, in_features, out_features, num_nodes):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)
        self.project_matrices = nn.Linear(num_nodes, num_nodes)


class SelfAttentionLayer(nn.Module):
    def __init__(self, num_nodes, attention_hidden_dim):
        super().__init__()
        self.tanh_layer = nn.Linear(num_nodes, attention_hidden_dim)
        self.softmax_layer = nn.Linear(attention_hidden_dim, num_nodes)


class _Function(torch.autograd.Function):
    """model(x) through rulgnn_sagcn_forward_f32 / rulgnn_sagcn_backward_f32."""

    @staticmethod
    def forward(ctx, model, x, *params):
        pred = model._forward(x)
        ctx.model, ctx.x = model, x
        ctx.tape = model._tape.tokens[x.size(0)]
        return pred.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dpred):
        model = ctx.model
        model._tape.check(ctx.x.size(0), ctx.tape, model._bufs, "SAGCN_model")
        grads = model._backward(ctx.x, dpred.reshape(-1).contiguous().float())
        return (None, None, *[grads[off:off + n].view(shape).clone() for off, n, shape in model._slices])


class SAGCN_model(FlatModule):
    def __init__(self, num_patch, patch_size, gcn_hidden_dim, attention_hidden_dim):
        super().__init__()
        self.num_patch, self.patch_size = int(num_patch), int(patch_size)
        self.gcn_hidden_dim, self.attention_hidden_dim = int(gcn_hidden_dim), int(attention_hidden_dim)
        # same construction order as the reference => same RNG consumption => same initial weights; the sub-modules only hold parameters
        self.gcn1 = GCNLayer(40, self.gcn_hidden_dim)
        self.proj1 = GraphProjectionLayer(self.gcn_hidden_dim, self.gcn_hidden_dim, self.num_patch)
        self.proj2 = GraphProjectionLayer(self.gcn_hidden_dim, self.gcn_hidden_dim, self.num_patch)
        self.attn = SelfAttentionLayer(self.num_patch, self.attention_hidden_dim)
        self.fc = nn.Linear(self.gcn_hidden_dim * self.num_patch, 1)
        self._tape = PL.ForwardTape()
        self._init_flat()

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        return _lib.SagcnShape(batch, self.num_patch, self.patch_size, self.gcn_hidden_dim, self.attention_hidden_dim)

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("SAGCN_model runs on the HIP path only: input must be a CUDA (ROCm) tensor; there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        bs = x.size(0)
        if x.numel() != bs * self.num_patch * self.patch_size:
            raise RuntimeError(f"shape '[{bs}, {self.num_patch}, {self.patch_size}]' is invalid for input of size {x.numel()}")
        return x.reshape(bs, self.num_patch * self.patch_size).contiguous().float()

    def _args(self, shp, x, y=None, dpred=None, global_batch=None):
        B = x.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_sagcn_workspace_bytes(C.byref(shp)),
                                    "SAGCN HIP kernels do not cover this configuration (num_patch <= 256, 2 <= patch_size <= 2048, hidden "
                                    "sizes <= 4096, batch * gcn_hidden_dim * num_patch < 2^31)")
        ws, pred = ent
        a = _lib.SagcnArgs()
        a.x = x.data_ptr()
        a.y = y.data_ptr() if y is not None else None
        a.dpred = dpred.data_ptr() if dpred is not None else None
        a.params, a.grads = self._flat.data_ptr(), self._grad_flat.data_ptr()
        a.pred = pred.data_ptr()
        a.loss = self._grad_flat.data_ptr() + 4 * self._count
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.global_batch = B if global_batch is None else int(global_batch)
        return a, pred

    def _forward(self, x):
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred = self._args(shp, x)
        _lib.check(_lib.load().rulgnn_sagcn_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_sagcn_forward_f32")
        return pred[:x.size(0)]

    def _backward(self, x, dpred):
        shp = self._shape(x.size(0))
        a, _ = self._args(shp, x, dpred=dpred)
        _lib.check(_lib.load().rulgnn_sagcn_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_sagcn_backward_f32")
        return self._grad_flat

    def tap(self, batch, which):
        """Workspace taps of the last forward at this batch size (parity tests): 'features' [B, P, 40], 'aggregated' (A_hat X), 'h3',
        'attention' -- the last three returned sample-major [B, P, .] (they are stored node-major)."""
        idx = {"features": 0, "aggregated": 1, "h3": 2, "attention": 3}[which]
        shp = self._shape(batch)
        off = _lib.load().rulgnn_sagcn_tap_offset(C.byref(shp), idx)
        ws = self._bufs[batch][0].view(torch.float32)
        P = self.num_patch
        if idx == 0:
            return ws[off:off + batch * P * 40].view(batch, P, 40).clone()
        w = 40 if idx == 1 else self.gcn_hidden_dim
        return ws[off:off + batch * P * w].view(P, batch, w).permute(1, 0, 2).contiguous()

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None):
        """forward + MSE + backward (+ Adam when ``optimizer`` is a FusedAdam over this model) in one C call; fills ``self.bucket`` =
        [grad | loss]; returns (pred [B], loss 0-d tensor) on the device, no host sync."""
        x = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred = self._args(shp, x, y=yv, global_batch=global_batch)
        o = self._adam_args(optimizer)
        _lib.check(_lib.load().rulgnn_sagcn_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_sagcn_fwdbwd_f32")
        return pred[:x.size(0)], self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, x):
        x2 = self._check_input(x)
        if x2.size(0) == 0:
            raise RuntimeError("SAGCN_model: empty batch")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._named()):
            return _Function.apply(self, x2, *self._named())
        return self._forward(x2).clone().view(-1, 1)
