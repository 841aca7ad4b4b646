"""Drop-in ``ST_Conv_model`` whose forward/backward run in the gfx950 HIP kernels (csrc/stconv.hip).

Mirrors the reference class (models/ST_Conv/Model.py:173-222): same constructor kwargs ``(num_nodes, time_length,
kernel_size)``, same ``forward(x) -> [bs, 1]``, the same ``state_dict`` keys in the same order -- including the "_2" layers
and the TCN's ``net0``/``net1`` branches that the reference's forward never calls (Model.py:196-206 uses the "_1" modules for
both branches) -- and the same initial weights for a torch seed.  None of the sub-modules is ever *called*: the live
parameters are views into one flat fp32 buffer that the kernels read directly (layout in include/rulgnn.h), the three live
BatchNorms' running statistics views into a second one; like in the reference each of them advances twice per training
forward.

There is no CPU path: calling the model with a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream
from .stgcn import MPNN_mk, TemporalConvNet

LIVE = ("theta1", "theta2", "theta3", "theta4", "gcn_layer_1.theta.0.weight", "gcn_layer_1.theta.0.bias",
        "cnn_layer_1.conv.weight", "cnn_layer_1.conv.bias", "cnn_layer_1.bn.weight", "cnn_layer_1.bn.bias",
        "tcn_layer_1.conv_block1.0.weight", "tcn_layer_1.conv_block1.2.weight", "tcn_layer_1.conv_block1.2.bias",
        "tcn_layer_1.conv_block2.0.weight", "tcn_layer_1.conv_block2.2.weight", "tcn_layer_1.conv_block2.2.bias",
        "fc.weight", "fc.bias")
BN_NAMES = ("tcn_layer_1.conv_block1.2", "tcn_layer_1.conv_block2.2", "cnn_layer_1.bn")


class CNNLayer(nn.Module):
    """Holder of ``conv`` = Conv1d(padding='same') and ``bn`` (Model.py:58-63)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1):
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel_size=kernel_size, padding='same', stride=stride)
        self.bn = nn.BatchNorm1d(out_channels)


class _TrainFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x2d, *params):
        pred = model._run_forward(x2d, training=True)
        model._after_train_forward(x2d.size(0))
        ctx.model, ctx.x2d = model, x2d
        return pred.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dpred):
        model = ctx.model
        grads = model._run_backward(ctx.x2d, dpred.contiguous().view(-1).float())
        out = [grads[off:off + n].view(shape).clone() for (off, n, shape) in model._slices]
        return (None, None, *out)


class ST_Conv_model(FlatModule):
    def __init__(self, num_nodes, time_length, kernel_size):
        super().__init__()
        self.num_nodes, self.time_length, self.kernel_size = int(num_nodes), int(time_length), int(kernel_size)
        # same construction order as the reference (Model.py:176-190) => same RNG consumption => same initial weights
        self.gcn_layer_1 = MPNN_mk(time_length, time_length, k=1)
        self.cnn_layer_1 = CNNLayer(num_nodes, num_nodes, kernel_size)
        self.tcn_layer_1 = TemporalConvNet(num_nodes, [num_nodes, num_nodes], kernel_size)
        self.gcn_layer_2 = MPNN_mk(time_length, time_length, k=1)
        self.cnn_layer_2 = CNNLayer(num_nodes, num_nodes, kernel_size)
        self.tcn_layer_2 = TemporalConvNet(num_nodes, [num_nodes, num_nodes], kernel_size)
        self.theta1 = nn.Parameter(torch.randn(1))
        self.theta2 = nn.Parameter(torch.randn(1))
        self.theta3 = nn.Parameter(torch.randn(1))
        self.theta4 = nn.Parameter(torch.randn(1))
        self.fc = nn.Linear(num_nodes * time_length, 1)

        self._bn = self._bn_batch = self._pred_buf = self._ws = None
        self._track_batchnorm_counters()
        self._init_flat()

    # ---- flat storage ----------------------------------------------------------------------------------
    flat_order = LIVE                                                  # the parameters the forward uses; the rest stay ordinary tensors
    workspace_slots = 4

    def _flush_nbt(self):
        if self._nbt_pending and self._nbt is not None:
            self._nbt += 2 * self._nbt_pending          # every live BatchNorm runs twice per training forward (Model.py:196-206)
            self._nbt_pending = 0

    def _bucket_floats(self):
        return self._count + 1 + 6 * self.num_nodes                   # [gradient | loss | BatchNorm batch moments]

    def _reflatten_buffers(self, dev):
        N = self.num_nodes
        bufs = dict(self.named_buffers())
        bn = torch.empty(6 * N, dtype=torch.float32, device=dev)
        nbt = torch.zeros(3, dtype=torch.int64, device=dev)
        for i, name in enumerate(BN_NAMES):
            for j, leaf in enumerate(("running_mean", "running_var")):
                sl = bn[(2 * i + j) * N:(2 * i + j + 1) * N]
                sl.copy_(bufs[f"{name}.{leaf}"].detach().float())
                self._set_buffer(f"{name}.{leaf}", sl)
            nbt[i].copy_(bufs[name + ".num_batches_tracked"])
            self._set_buffer(name + ".num_batches_tracked", nbt[i])
        self._bn, self._nbt = bn, nbt
        self._bn_batch = torch.zeros(6 * N, dtype=torch.float32, device=dev)

    def _reset_caches(self):
        super()._reset_caches()
        self._pred_buf = self._ws = None

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        return _lib.StconvShape(batch, self.num_nodes, self.time_length, self.kernel_size)

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("ST_Conv_model runs on the HIP kernels only: input must be a CUDA (ROCm) tensor; "
                               "there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        if x.dim() != 3 or x.size(1) != self.num_nodes or x.size(2) != self.time_length:
            raise RuntimeError(f"expected input [bs, {self.num_nodes}, {self.time_length}], got {list(x.shape)}")
        return x.reshape(x.size(0), -1).contiguous().float()

    def _args(self, shp, x2d, training, y=None, dpred=None, global_batch=None, moments_to_bucket=False):
        B = x2d.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_stconv_workspace_bytes(C.byref(shp)),
                                    "ST_Conv kernels do not cover this configuration (kernel_size 6, num_nodes <= 25, time_length <= 64)",
                                    make=lambda dev: (torch.empty(B, dtype=torch.float32, device=dev),))
        self._ws, self._pred_buf = ent
        a = _lib.AstgcnnArgs()
        a.x = x2d.data_ptr()
        a.y = y.data_ptr() if y is not None else None
        a.dpred = dpred.data_ptr() if dpred is not None else None
        a.params = self._flat.data_ptr()
        a.grads = self._grad_flat.data_ptr()
        a.pred = self._pred_buf.data_ptr()
        a.loss = self._grad_flat.data_ptr() + 4 * self._count
        a.bn_stats = self._bn.data_ptr()
        gb = B if global_batch is None else int(global_batch)
        if moments_to_bucket:
            a.bn_batch = self._grad_flat.data_ptr() + 4 * (self._count + 1)
            a.bn_moment_weight = B / float(gb)
        else:
            a.bn_batch = self._bn_batch.data_ptr()
            a.bn_moment_weight = 0.0
        a.workspace = self._ws.data_ptr()
        a.workspace_bytes = self._ws.numel()
        a.global_batch = gb
        a.training = 1 if training else 0
        return a

    def _run_forward(self, x2d, training):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, training)
        _lib.check(_lib.load().rulgnn_stconv_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stconv_forward_f32")
        return self._pred_buf

    def _run_backward(self, x2d, dpred):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, dpred=dpred)
        _lib.check(_lib.load().rulgnn_stconv_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stconv_backward_f32")
        return self._grad_flat

    def _after_train_forward(self, batch, from_bucket_moments=False):
        src = self._grad_flat.data_ptr() + 4 * (self._count + 1) if from_bucket_moments else self._bn_batch.data_ptr()
        shp = self._shape(batch)
        _lib.check(_lib.load().rulgnn_stconv_bn_running_update_f32(C.byref(shp), self._bn.data_ptr(), src, batch * self.time_length,
                                                                   0.1, 1 if from_bucket_moments else 0, _stream()),
                   "rulgnn_stconv_bn_running_update_f32")
        self._nbt_pending += 1

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None, sample_offset=0, update_running_stats=True,
                       moments_to_bucket=False):
        """train forward + MSE + backward (+ Adam and the running statistics with ``optimizer``) in one C call."""
        x2d = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x2d.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, y=yv, global_batch=global_batch, moments_to_bucket=moments_to_bucket)
        o = self._adam_args(optimizer, bn=self._bn)
        _lib.check(_lib.load().rulgnn_stconv_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_stconv_fwdbwd_f32")
        if optimizer is not None:
            self._nbt_pending += 1
        elif update_running_stats:
            self._after_train_forward(x2d.size(0))
        return self._pred_buf, self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, x):
        x2d = self._check_input(x)
        if x2d.size(0) == 0:
            if self.training:
                raise RuntimeError("training forward needs a non-empty batch")
            return torch.empty(0, 1, dtype=torch.float32, device=x2d.device)
        if self.training:
            if torch.is_grad_enabled():
                return _TrainFunction.apply(self, x2d, *[p for _, p in self._named_live()])
            pred = self._run_forward(x2d, training=True)
            self._after_train_forward(x2d.size(0))
            return pred.clone().view(-1, 1)
        return self._run_forward(x2d, training=False).clone().view(-1, 1)
