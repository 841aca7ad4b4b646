"""Drop-in ``RGCNU_model`` (a ``GCNLayer`` user).  The whole model runs behind three C entries on one
flat parameter buffer (``rulgnn_rgcnu_{forward,backward,fwdbwd}_f32``; ``fused_mse_step`` is forward + MSE + backward + Adam in one
call): the learned adjacency, the per-(sample, time-step) graph convolutions, the 1x1 / 'same' convolutions and the two heads in the
gfx950 kernels of csrc/rgcnu.hip, the LSTM over the time steps in the persistent kernels of csrc/bilstm.hip (one direction).

Mirrors the reference class (models/RGCNU/Model.py:96-119): same constructor kwargs
``(num_nodes, time_length, hidden_dim, encoder_hidden_dim, kernel_size, alpha)``, ``forward(X, train=False)`` returning the
prediction ``[bs, 1]`` or ``(prediction, std)``, the same 22 ``state_dict`` keys and -- sub-modules being created in the reference's
order -- the same initial weights for a torch seed.  Reference quirk kept: graph (b, l) is convolved with the adjacency of sample
``(b * time_length + l) % bs`` (``A.repeat(time_length, 1, 1)`` against sample-major node signals, Model.py:104-106).
SCL's ``Dropout(0.5)`` uses the counter-based hash of the other families (mask = f(seed, step, element); torch's Bernoulli stream
cannot be reproduced by any other implementation).  There is no CPU path: a non-CUDA input raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream

SCL_DROPOUT = 0.5          # models/RGCNU/Model.py:31


# ---- parameter holders: created in the reference's order so that a seed gives the reference's initial weights; never called ----
class GCNLayer(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)


class SCL(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.gcn1 = GCNLayer(1, hidden_dim)
        self.gcn2 = GCNLayer(hidden_dim, hidden_dim)
        self.conv1d = nn.Conv1d(hidden_dim, 1, kernel_size=1)
        self.dropout = nn.Dropout(p=SCL_DROPOUT)


class TDL(nn.Module):
    def __init__(self, num_nodes, encoder_hidden_dim):
        super().__init__()
        self.lstm = nn.LSTM(num_nodes, encoder_hidden_dim, batch_first=True)


class FusionModule(nn.Module):
    def __init__(self, num_nodes, encoder_hidden_dim, kernel_size, time_length):
        super().__init__()
        self.cnn1 = nn.Conv1d(num_nodes, encoder_hidden_dim, kernel_size=1)
        self.cnn2 = nn.Conv1d(encoder_hidden_dim, encoder_hidden_dim, kernel_size=kernel_size, padding='same')
        self.fc1 = nn.Linear(encoder_hidden_dim * time_length, 1)
        self.fc2 = nn.Linear(encoder_hidden_dim * time_length, 1)


class adj_construction(nn.Module):
    def __init__(self, num_nodes, time_length, alpha):
        super().__init__()
        self.alpha = alpha
        self.trainable_theta1 = nn.Linear(time_length, num_nodes)
        self.trainable_theta2 = nn.Linear(time_length, num_nodes)


PARAM_ORDER = ["adj.trainable_theta1.weight", "adj.trainable_theta1.bias", "adj.trainable_theta2.weight", "adj.trainable_theta2.bias",
               "scl.gcn1.linear.weight", "scl.gcn1.linear.bias", "scl.gcn2.linear.weight", "scl.gcn2.linear.bias",
               "scl.conv1d.weight", "scl.conv1d.bias",
               "tdl.lstm.weight_ih_l0", "tdl.lstm.weight_hh_l0", "tdl.lstm.bias_ih_l0", "tdl.lstm.bias_hh_l0",
               "fusion.cnn1.weight", "fusion.cnn1.bias", "fusion.cnn2.weight", "fusion.cnn2.bias",
               "fusion.fc1.weight", "fusion.fc1.bias", "fusion.fc2.weight", "fusion.fc2.bias"]


class _Function(torch.autograd.Function):
    """model(x, train=True) through rulgnn_rgcnu_forward_f32 / rulgnn_rgcnu_backward_f32 (flat parameters).  The second head is
    returned detached: its only consumer in the reference is a commented-out loss (algorithms.py:288)."""

    @staticmethod
    def forward(ctx, model, x, training, *params):
        pred, std = model._forward(x, training)
        ctx.model, ctx.x, ctx.step, ctx.training = model, x, model._step, bool(training)
        ctx.tape = model._tape.tokens[x.size(0)]
        ctx.mark_non_differentiable(std)
        return pred.clone().view(-1, 1), std.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dpred, _dstd):
        model = ctx.model
        model._tape.check(ctx.x.size(0), ctx.tape, model._bufs, "RGCNU_model")
        grads = model._backward(ctx.x, dpred.reshape(-1).contiguous().float(), ctx.step, ctx.training)
        model._tape.consume(ctx.x.size(0), ctx.tape)
        outs = [grads[off:off + n].view(shape).clone() for off, n, shape in model._slices]
        return (None, None, None, *outs)


class RGCNU_model(FlatModule):
    dropout_by_sample_offset = True          # dp.py: pass the shard's first global sample index to fused_mse_step
    # graph (b, l) meets the adjacency of sample (b * T + l) % bs: the eval forward depends on WHICH samples share a batch, so a test
    # set must be walked in the reference's batches on every rank, never re-batched per shard (trainer.py / dataloader.data_generator)
    eval_sample_independent = False

    def __init__(self, num_nodes, time_length, hidden_dim, encoder_hidden_dim, kernel_size, alpha):
        super().__init__()
        self.num_nodes, self.time_length = int(num_nodes), int(time_length)
        self.hidden_dim, self.encoder_hidden_dim, self.kernel_size = int(hidden_dim), int(encoder_hidden_dim), int(kernel_size)
        self.alpha = float(alpha)
        self.adj = adj_construction(self.num_nodes, self.time_length, alpha)
        self.scl = SCL(self.hidden_dim)
        self.tdl = TDL(self.num_nodes, self.encoder_hidden_dim)
        self.fusion = FusionModule(self.num_nodes, self.encoder_hidden_dim, self.kernel_size, self.time_length)
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self._step = 0
        if [n for n, _ in self.named_parameters()] != PARAM_ORDER:
            raise RuntimeError("parameter order differs from the flat layout of include/rulgnn.h")
        self._tape = PL.ForwardTape()
        self._init_flat()
        # fusion.fc2 (the `std` head) is not in the loss: its gradient is None in the reference and torch's Adam never touches it
        self.num_optimized = self._layout["fusion.fc2.weight"][0]

    flat_order = PARAM_ORDER
    workspace_slots = 4

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        return _lib.RgcnuShape(batch, self.num_nodes, self.time_length, self.hidden_dim, self.encoder_hidden_dim, self.kernel_size, self.alpha)

    def _check_input(self, x):
        if x.dim() != 3 or x.size(1) != self.num_nodes or x.size(2) != self.time_length:
            raise RuntimeError(f"RGCNU_model expects [bs, {self.num_nodes}, {self.time_length}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("RGCNU_model runs on the HIP path only: input must be a CUDA (ROCm) tensor; there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        return x.contiguous().float()

    def _args(self, shp, x, training, step, y=None, dpred=None, global_batch=None, sample_offset=0):
        B = x.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_rgcnu_workspace_bytes(C.byref(shp)),
                                    "RGCNU HIP kernels do not cover this configuration (num_nodes <= 32, time_length <= 64, hidden widths "
                                    "<= 64, odd kernel_size <= 7)",
                                    make=lambda dev: tuple(torch.empty(max(B, 1), dtype=torch.float32, device=dev) for _ in range(2)))
        ws, pred, std = ent
        a = _lib.RgcnuArgs()
        a.x = x.data_ptr()
        a.y = y.data_ptr() if y is not None else None
        a.dpred = dpred.data_ptr() if dpred is not None else None
        a.params, a.grads = self._flat.data_ptr(), self._grad_flat.data_ptr()
        a.pred, a.std_pred = pred.data_ptr(), std.data_ptr()
        a.loss = self._grad_flat.data_ptr() + 4 * self._count
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.global_batch = B if global_batch is None else int(global_batch)
        a.sample_offset = int(sample_offset)
        a.dropout_p = float(self.scl.dropout.p)
        a.seed, a.step = self._seed, int(step)
        a.training = 1 if training else 0
        return a, pred, std

    def _forward(self, x, training):
        if training:
            self._step += 1
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred, std = self._args(shp, x, training, self._step)
        _lib.check(_lib.load().rulgnn_rgcnu_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_rgcnu_forward_f32")
        return pred[:x.size(0)], std[:x.size(0)]

    def _backward(self, x, dpred, step, training=True):
        """``training`` must be the flag of the forward this backward belongs to: the dropout mask of rg_scl_bwd is applied only then."""
        shp = self._shape(x.size(0))
        a, _, _ = self._args(shp, x, training, step, dpred=dpred)
        _lib.check(_lib.load().rulgnn_rgcnu_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_rgcnu_backward_f32")
        return self._grad_flat

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None, sample_offset=0):
        """forward (train mode) + MSE of the first head + backward (+ Adam when ``optimizer`` is a FusedAdam over this model) in one C
        call; fills ``self.bucket`` = [grad | loss]; returns (pred [B], loss 0-d tensor) on the device, no host sync."""
        x = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x.size(0):
            raise RuntimeError("target size mismatch")
        self._step += 1
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred, _ = self._args(shp, x, True, self._step, y=yv, global_batch=global_batch, sample_offset=sample_offset)
        o = self._adam_args(optimizer)
        _lib.check(_lib.load().rulgnn_rgcnu_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_rgcnu_fwdbwd_f32")
        return pred[:x.size(0)], self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, X, train=False):
        """``train`` selects the return value like the reference (Model.py:115-118); dropout follows ``self.training``."""
        x = self._check_input(X)
        if x.size(0) == 0:
            raise RuntimeError("RGCNU_model: empty batch")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._named()):
            pred, std = _Function.apply(self, x, self.training, *self._named())
        else:
            p, s = self._forward(x, self.training)
            pred, std = p.clone().view(-1, 1), s.clone().view(-1, 1)
        return (pred, std) if train else pred
