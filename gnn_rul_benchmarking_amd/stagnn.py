"""Drop-in ``STAGNN_model`` (the last of the ``GCNLayer`` users).  The whole model runs behind three C entries
on one flat parameter buffer (``rulgnn_stagnn_{forward,backward,fwdbwd}_f32``; ``fused_mse_step`` is train-mode forward + MSE +
backward + Adam in one call) in the gfx950 kernels of csrc/stagnn.hip: covariance adjacency, two GCN + multi-head graph-attention
layers, two temporal convolution blocks with train-mode BatchNorm (batch statistics, differentiated) and two temporal encoders.

Mirrors the reference class (models/STAGNN/Model.py:184-230): same constructor kwargs ``(num_nodes, time_length, hidden_dim,
output_dim, num_heads, threshold)``, ``forward(x) -> [bs, 1]``, the same 90 ``state_dict`` keys in the same order (including the dead
weight-normed ``net0`` / ``net1`` branches and the BatchNorm buffers) and -- sub-modules being created in the reference's order -- the
same initial weights for a torch seed.  ``model.train()`` uses batch statistics and updates the running ones (momentum 0.1, unbiased
variance, ``num_batches_tracked``), ``model.eval()`` uses the running ones.  There is no CPU path: a non-CUDA input raises.
Data parallelism shards the batch with rank-local BatchNorm statistics (what torch's DistributedDataParallel does without SyncBatchNorm).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
from torch.nn.utils import weight_norm

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream


class GCNLayer(nn.Module):
    def __init__(self, in_features, out_features):
        super().__init__()
        self.linear = nn.Linear(in_features, out_features)


class GraphAttentionLayer(nn.Module):
    def __init__(self, in_features, out_features, dropout, alpha=0.1):
        super().__init__()
        self.in_features, self.out_features, self.dropout, self.alpha = in_features, out_features, dropout, alpha
        self.linear = nn.Linear(in_features, out_features)
        self.attention = nn.Linear(2 * out_features, 1)


class GAT(nn.Module):
    def __init__(self, nfeat, nout, dropout, nheads):
        super().__init__()
        self.dropout = dropout
        self.attentions = [GraphAttentionLayer(nfeat, nout, dropout=dropout) for _ in range(nheads)]
        for i, attention in enumerate(self.attentions):
            self.add_module('attention_{}'.format(i), attention)


class Chomp1d(nn.Module):
    def __init__(self, chomp_size):
        super().__init__()
        self.chomp_size = chomp_size


class TemporalConvNet(nn.Module):
    """Holder of the block's tensors (Model.py:85-159).  ``net0`` / ``net1`` exist only so that the state_dict and the RNG consumption
    match the reference: they are never called there either."""

    def __init__(self, num_inputs, num_channels, kernel_size):
        super().__init__()
        c_in, c0, c1 = num_inputs, num_channels[0], num_channels[1]
        p0, p1 = (kernel_size - 1) * 1, (kernel_size - 1) * 2
        self.net0 = nn.Sequential(weight_norm(nn.Conv1d(c_in, c1, kernel_size, stride=1, padding=p0, dilation=1)), nn.ReLU(),
                                  weight_norm(nn.Conv1d(c1, c1, kernel_size, stride=1, padding=p0, dilation=1)), nn.ReLU())
        self.downsample0 = nn.Conv1d(c_in, c1, 1) if c_in != c1 else None
        self.relu = nn.ReLU()
        self.net1 = nn.Sequential(nn.Conv1d(c_in, c1, kernel_size, stride=1, padding=p1, dilation=2), nn.ReLU(),
                                  nn.Conv1d(c1, c1, kernel_size, stride=1, padding=p1, dilation=2), nn.ReLU())
        self.downsample1 = nn.Conv1d(c1, c1, 1) if c0 != c1 else None
        self.conv_block1 = nn.Sequential(nn.Conv1d(c_in, c1, kernel_size=kernel_size, stride=1, bias=False, padding=p0, dilation=1),
                                         Chomp1d(p0), nn.BatchNorm1d(c1), nn.ReLU())
        self.conv_block2 = nn.Sequential(nn.Conv1d(c1, c1, kernel_size=kernel_size, stride=1, bias=False, padding=p1, dilation=2),
                                         Chomp1d(p1), nn.BatchNorm1d(c1), nn.ReLU())


class MultiHeadTemporalEncoder(nn.Module):
    def __init__(self, num_heads, num_features):
        super().__init__()
        self.num_heads = num_heads
        self.linears = nn.ModuleList([nn.Linear(num_features, 1) for _ in range(self.num_heads)])


def live_parameter_names(num_heads):
    """The parameters that receive a gradient in the reference, in named_parameters() order (the flat buffer's layout)."""
    names = []
    for l in (1, 2):
        names += [f"gcn{l}.linear.weight", f"gcn{l}.linear.bias"]
        names += [f"gat{l}.attention_{i}.{k}" for i in range(num_heads) for k in ("linear.weight", "linear.bias", "attention.weight", "attention.bias")]
    for l in (1, 2):
        names += [f"tcn{l}.downsample0.weight", f"tcn{l}.downsample0.bias", f"tcn{l}.conv_block1.0.weight", f"tcn{l}.conv_block1.2.weight",
                  f"tcn{l}.conv_block1.2.bias", f"tcn{l}.conv_block2.0.weight", f"tcn{l}.conv_block2.2.weight", f"tcn{l}.conv_block2.2.bias"]
        names += [f"temporal_encoder{l}.linears.{i}.{k}" for i in range(num_heads) for k in ("weight", "bias")]
    return names + ["fc.weight", "fc.bias"]


_BN_LAYERS = ("tcn1.conv_block1.2", "tcn1.conv_block2.2", "tcn2.conv_block1.2", "tcn2.conv_block2.2")


class _TrainFunction(torch.autograd.Function):
    """model(x) in train mode through rulgnn_stagnn_forward_f32 / rulgnn_stagnn_backward_f32."""

    @staticmethod
    def forward(ctx, model, x, *params):
        pred = model._run_forward(x, training=True)
        model._nbt_pending += 1
        ctx.model, ctx.x = model, x
        ctx.tape = model._tape.tokens[x.size(0)]
        return pred.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dpred):
        model = ctx.model
        model._tape.check(ctx.x.size(0), ctx.tape, model._bufs, "STAGNN_model")
        grads = model._run_backward(ctx.x, dpred.reshape(-1).contiguous().float())
        return (None, None, *[grads[off:off + n].view(shape).clone() for off, n, shape in model._slices])


class STAGNN_model(FlatModule):
    def __init__(self, num_nodes, time_length, hidden_dim, output_dim, num_heads, threshold):
        super().__init__()
        self.num_nodes, self.time_length, self.hidden_dim = int(num_nodes), int(time_length), int(hidden_dim)
        self.output_dim, self.num_heads, self.threshold = int(output_dim), int(num_heads), threshold
        h = self.hidden_dim
        # same construction order as the reference => same RNG consumption => same initial weights; the sub-modules only hold tensors
        self.gcn1 = GCNLayer(self.time_length, h)
        self.gat1 = GAT(h, h, dropout=0, nheads=self.num_heads)
        self.gcn2 = GCNLayer(h, h)
        self.gat2 = GAT(h, h, dropout=0, nheads=self.num_heads)
        self.tcn1 = TemporalConvNet(num_inputs=self.num_nodes, num_channels=[h, h], kernel_size=2)
        self.temporal_encoder1 = MultiHeadTemporalEncoder(self.num_heads, h)
        self.tcn2 = TemporalConvNet(num_inputs=h, num_channels=[self.output_dim, self.output_dim], kernel_size=2)
        self.temporal_encoder2 = MultiHeadTemporalEncoder(self.num_heads, self.output_dim)
        self.fc = nn.Linear(h * self.output_dim, 1)
        table = dict(self.named_parameters())
        for name in live_parameter_names(self.num_heads):
            if name not in table:
                raise RuntimeError(f"STAGNN HIP kernels need the residual 1x1 convolutions (num_nodes != hidden_dim != output_dim): no '{name}'")
        self.flat_order = live_parameter_names(self.num_heads)
        self._bn_channels = (h, h, self.output_dim, self.output_dim)
        self._bn = self._nbt = None
        self._tape = PL.ForwardTape()
        self._track_batchnorm_counters()
        self._init_flat()

    # ---- flat storage ----------------------------------------------------------------------------------
    workspace_slots = 3

    def _reflatten_buffers(self, dev):
        bufs = dict(self.named_buffers())
        bn = torch.empty(2 * sum(self._bn_channels), dtype=torch.float32, device=dev)
        nbt = torch.zeros(4, dtype=torch.int64, device=dev)
        off = 0
        for k, (layer, c) in enumerate(zip(_BN_LAYERS, self._bn_channels)):
            for stat in ("running_mean", "running_var"):
                bn[off:off + c].copy_(bufs[f"{layer}.{stat}"].detach().float())
                self._set_buffer(f"{layer}.{stat}", bn[off:off + c])
                off += c
            nbt[k].copy_(bufs[f"{layer}.num_batches_tracked"])
            self._set_buffer(f"{layer}.num_batches_tracked", nbt[k])
        self._bn, self._nbt = bn, nbt

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        return _lib.StagnnShape(batch, self.num_nodes, self.time_length, self.hidden_dim, self.output_dim, self.num_heads, float(self.threshold))

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("STAGNN_model runs on the HIP path only: input must be a CUDA (ROCm) tensor; there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        if x.dim() != 3 or x.size(1) != self.num_nodes or x.size(2) != self.time_length:
            raise RuntimeError(f"expected input [bs, {self.num_nodes}, {self.time_length}], got {list(x.shape)}")
        return x.contiguous().float()

    def _args(self, shp, x, training, y=None, dpred=None, global_batch=None, update_running_stats=True):
        B = x.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_stagnn_workspace_bytes(C.byref(shp)),
                                    "STAGNN HIP kernels do not cover this configuration (num_nodes <= 32, time_length <= 128, 3 <= hidden_dim "
                                    "<= 64, output_dim <= 16, num_heads <= 4, num_nodes != hidden_dim != output_dim)")
        ws, pred = ent
        a = _lib.StagnnArgs()
        a.x = x.data_ptr()
        a.y = y.data_ptr() if y is not None else None
        a.dpred = dpred.data_ptr() if dpred is not None else None
        a.params, a.grads = self._flat.data_ptr(), self._grad_flat.data_ptr()
        a.pred = pred.data_ptr()
        a.loss = self._grad_flat.data_ptr() + 4 * self._count
        a.bn_state = self._bn.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.global_batch = B if global_batch is None else int(global_batch)
        a.training = 1 if training else 0
        a.update_running_stats = 1 if (training and update_running_stats) else 0
        return a, pred

    def _run_forward(self, x, training):
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred = self._args(shp, x, training)
        _lib.check(_lib.load().rulgnn_stagnn_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stagnn_forward_f32")
        return pred[:x.size(0)]

    def _run_backward(self, x, dpred):
        shp = self._shape(x.size(0))
        a, _ = self._args(shp, x, True, dpred=dpred, update_running_stats=False)
        _lib.check(_lib.load().rulgnn_stagnn_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stagnn_backward_f32")
        return self._grad_flat

    def tap(self, batch, which):
        """Workspace taps of the last forward at this batch size (parity tests)."""
        idx, shape = {"adjacency": (0, (self.num_nodes, self.num_nodes)), "graph": (1, (self.num_nodes, self.hidden_dim)),
                      "tcn1": (2, (self.hidden_dim, self.hidden_dim)), "encoder1": (3, (self.hidden_dim, self.hidden_dim)),
                      "tcn2": (4, (self.output_dim, self.hidden_dim)), "encoder2": (5, (self.output_dim, self.hidden_dim))}[which]
        off = _lib.load().rulgnn_stagnn_tap_offset(C.byref(self._shape(batch)), idx)
        ws = self._bufs[batch][0].view(torch.float32)
        return ws[off:off + batch * shape[0] * shape[1]].view(batch, *shape).clone()

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None, update_running_stats=True):
        """train-mode forward + MSE + backward (+ Adam when ``optimizer`` is a FusedAdam over this model) in one C call; fills
        ``self.bucket`` = [grad | loss]; returns (pred [B], loss 0-d tensor) on the device, no host sync."""
        x = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred = self._args(shp, x, True, y=yv, global_batch=global_batch, update_running_stats=update_running_stats)
        o = self._adam_args(optimizer)
        _lib.check(_lib.load().rulgnn_stagnn_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_stagnn_fwdbwd_f32")
        if update_running_stats:
            self._nbt_pending += 1
        return pred[:x.size(0)], self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, x):
        x = self._check_input(x)
        if x.size(0) == 0:
            raise RuntimeError("STAGNN_model: empty batch")
        if self.training:
            if torch.is_grad_enabled() and any(p.requires_grad for p in self._named()):
                return _TrainFunction.apply(self, x, *self._named())
            pred = self._run_forward(x, training=True)
            self._nbt_pending += 1
            return pred.clone().view(-1, 1)
        return self._run_forward(x, training=False).clone().view(-1, 1)
