"""Drop-in ``FC_STGNN_RUL`` whose forward/backward run in the gfx950 HIP kernels (csrc/fcstgnn.hip).

Mirrors the reference class (models/FC_STGNN/Model.py:5-84): same ten constructor kwargs, same ``forward(X) -> [bs, 1]``,
the same 56 ``state_dict`` keys (seven BatchNorms with their buffers, the ``positional_encoding.pe`` table buffer
``[1, 5000, 2*hidden_dim]``) in the same order and -- because the parameter-holding sub-modules are created in the
reference's order -- the same initial weights for a torch seed.  None of the sub-modules is ever *called*: parameters are
views into one flat fp32 buffer that the kernels read directly (layout in include/rulgnn.h), the BatchNorm running
statistics views into a second one.

The positional-encoding dropout (p = 0.1 in train mode, Model.py:25) uses the package's counter-based hash stream
(seed, step, element index) instead of torch's Bernoulli stream; everything else is the reference's arithmetic.

There is no CPU path: calling the model with a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream

PE_DROPOUT = 0.1


class Feature_extractor_1DCNN_RUL(nn.Module):
    """Holder of the two conv blocks (Model_Base.py:12-30)."""

    def __init__(self, input_channels, num_hidden, out_dim, kernel_size=8, stride=1, dropout=0):
        super().__init__()
        self.conv_block1 = nn.Sequential(
            nn.Conv1d(input_channels, num_hidden, kernel_size=kernel_size, stride=stride, bias=False, padding=(kernel_size // 2)),
            nn.BatchNorm1d(num_hidden), nn.ReLU(), nn.Dropout(dropout))
        self.conv_block2 = nn.Sequential(
            nn.Conv1d(num_hidden, out_dim, kernel_size=kernel_size, stride=1, bias=False, padding=1),
            nn.BatchNorm1d(out_dim), nn.ReLU())


class PositionalEncoding(nn.Module):
    """Holder of the ``pe`` buffer (Model_Base.py:111-125; base 100)."""

    def __init__(self, d_model, dropout, max_len=5000):
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * -(math.log(100.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer('pe', pe.unsqueeze(0))


class Dot_Graph_Construction_weights(nn.Module):
    def __init__(self, input_dim):
        super().__init__()
        self.mapping = nn.Linear(input_dim, input_dim)


class MPNN_mk_v2(nn.Module):
    def __init__(self, input_dimension, outpuut_dinmension, k):
        super().__init__()
        self.k = k
        self.theta = nn.ModuleList([nn.Linear(input_dimension, outpuut_dinmension) for _ in range(k)])
        self.bn1 = nn.BatchNorm1d(outpuut_dinmension)


class GraphConvpoolMPNN_block_v6(nn.Module):
    def __init__(self, input_dim, output_dim, num_sensors, time_length, time_window_size, stride, decay, pool_choice):
        super().__init__()
        self.graph_construction = Dot_Graph_Construction_weights(input_dim)
        self.BN = nn.BatchNorm1d(input_dim)
        self.MPNN = MPNN_mk_v2(input_dim, output_dim, k=1)


BN_NAMES = ("nonlin_map.conv_block1.1", "nonlin_map.conv_block2.1", "nonlin_map2.1", "MPNN1.BN", "MPNN1.MPNN.bn1", "MPNN2.BN",
            "MPNN2.MPNN.bn1")


class _TrainFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x2d, *params):
        model._step += 1
        ctx.step = model._step
        pred = model._run_forward(x2d, True, model._step)
        model._after_train_forward()
        ctx.model, ctx.x2d = model, x2d
        return pred.clone().view(-1, 1)

    @staticmethod
    def backward(ctx, dpred):
        model = ctx.model
        grads = model._run_backward(ctx.x2d, dpred.contiguous().view(-1).float(), ctx.step)
        out = [grads[off:off + n].view(shape).clone() for (off, n, shape) in model._slices]
        return (None, None, *out)


class FC_STGNN_RUL(FlatModule):
    def __init__(self, patch_size, num_patch, encoder_time_out, encoder_hidden_dim, encoder_out_dim, encoder_conv_kernel,
                 hidden_dim, num_sequential, num_node, num_windows):
        super().__init__()
        self.cfg = dict(patch_size=int(patch_size), num_patch=int(num_patch), encoder_time_out=int(encoder_time_out),
                        encoder_hidden_dim=int(encoder_hidden_dim), encoder_out_dim=int(encoder_out_dim),
                        encoder_conv_kernel=int(encoder_conv_kernel), hidden_dim=int(hidden_dim),
                        num_sequential=int(num_sequential), num_node=int(num_node), num_windows=int(num_windows))
        self.patch_size, self.num_patch = int(patch_size), int(num_patch)
        # same construction order as the reference (Model.py:20-43) => same RNG consumption => same initial weights
        self.nonlin_map = Feature_extractor_1DCNN_RUL(1, encoder_hidden_dim, encoder_out_dim, kernel_size=encoder_conv_kernel)
        self.nonlin_map2 = nn.Sequential(nn.Linear(encoder_out_dim * encoder_time_out, 2 * hidden_dim), nn.BatchNorm1d(2 * hidden_dim))
        self.positional_encoding = PositionalEncoding(2 * hidden_dim, PE_DROPOUT, max_len=5000)
        self.MPNN1 = GraphConvpoolMPNN_block_v6(2 * hidden_dim, hidden_dim, num_node, num_sequential, 2, 1, 0.7, 'mean')
        self.MPNN2 = GraphConvpoolMPNN_block_v6(2 * hidden_dim, hidden_dim, num_node, num_sequential, 2, 2, 0.7, 'mean')
        self.fc = nn.Sequential(OrderedDict([
            ('fc1', nn.Linear(hidden_dim * num_windows * num_node, 2 * hidden_dim)), ('relu1', nn.ReLU(inplace=True)),
            ('fc2', nn.Linear(2 * hidden_dim, 2 * hidden_dim)), ('relu2', nn.ReLU(inplace=True)),
            ('fc3', nn.Linear(2 * hidden_dim, hidden_dim)), ('relu3', nn.ReLU(inplace=True)),
            ('fc4', nn.Linear(hidden_dim, 1))]))

        # flat layout = named_parameters() order (the order include/rulgnn.h documents)
        self._bn_ch = [dict(self.named_buffers())[n + ".running_mean"].numel() for n in BN_NAMES]
        self._bn = self._bn_batch = self._pred_buf = self._ws = None
        self.side_stream = PL.SideStream()
        self._step = 0
        self._seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.dropout_p = PE_DROPOUT
        # "f32" (default; meets the 1e-4 parity gate) or "bf16": bf16 operands on every product of the window-graph kernels (forward and
        # backward: v_mfma_f32_32x32x16_bf16) and on the row-projection GEMMs; fp32 accumulation / softmax / BatchNorm / weight gradients /
        # optimizer -- BASELINE.json's "FC_STGNN ... bf16" variant, reported separately (rulgnn.h: rulgnn_fcstgnn_args.compute_dtype)
        self.compute_dtype = "f32"
        self._track_batchnorm_counters()
        self._init_flat()
        lib_count = _lib.load().rulgnn_fcstgnn_param_count(C.byref(self._shape(1)))
        if lib_count >= 0 and lib_count != self._count:
            raise RuntimeError(f"flat parameter layout mismatch: module {self._count} vs kernels {lib_count}")

    # ---- flat storage ----------------------------------------------------------------------------------
    workspace_slots = 4

    def _bucket_floats(self):
        return self._count + 1 + 2 * sum(self._bn_ch)                 # [gradient | loss | BatchNorm batch moments]

    def _reflatten_buffers(self, dev):
        bufs = dict(self.named_buffers())
        total = 2 * sum(self._bn_ch)
        bn = torch.empty(total, dtype=torch.float32, device=dev)
        nbt = torch.zeros(len(BN_NAMES), dtype=torch.int64, device=dev)
        o = 0
        for i, (name, c) in enumerate(zip(BN_NAMES, self._bn_ch)):
            bn[o:o + c].copy_(bufs[name + ".running_mean"].detach().float())
            bn[o + c:o + 2 * c].copy_(bufs[name + ".running_var"].detach().float())
            self._set_buffer(name + ".running_mean", bn[o:o + c])
            self._set_buffer(name + ".running_var", bn[o + c:o + 2 * c])
            nbt[i].copy_(bufs[name + ".num_batches_tracked"])
            self._set_buffer(name + ".num_batches_tracked", nbt[i])
            o += 2 * c
        self._bn, self._nbt = bn, nbt
        self._bn_batch = torch.zeros(total, dtype=torch.float32, device=dev)

    def _reset_caches(self):
        super()._reset_caches()
        self._pred_buf = self._ws = None

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        c = self.cfg
        return _lib.FcstgnnShape(batch, c["patch_size"], c["num_patch"], c["encoder_time_out"], c["encoder_hidden_dim"],
                                 c["encoder_out_dim"], c["encoder_conv_kernel"], c["hidden_dim"], c["num_sequential"], c["num_node"],
                                 c["num_windows"])

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("FC_STGNN_RUL runs on the HIP kernels only: input must be a CUDA (ROCm) tensor; "
                               "there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        c = self.cfg
        if x.dim() != 3 or x.size(1) != c["num_node"] or x.size(2) != c["num_patch"] * c["patch_size"]:
            raise RuntimeError(f"shape '[{x.size(0)}, {c['num_node']}, {c['num_patch']}, {c['patch_size']}]' is invalid for input "
                               f"of size {x.numel()}")
        return x.reshape(x.size(0), -1).contiguous().float()

    def _args(self, shp, x2d, training, step, y=None, dpred=None, global_batch=None, sample_offset=0, moments_to_bucket=False):
        B = x2d.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_fcstgnn_workspace_bytes(C.byref(shp)),
                                    "FC_STGNN kernels do not cover this configuration (encoder_time_out must be the second conv's "
                                    "output length, num_windows the windows of the two blocks; num_node <= 20, hidden_dim <= 32, "
                                    "encoder_out_dim <= 64, encoder_hidden_dim <= 16, encoder_conv_kernel <= 4)",
                                    make=lambda dev: (torch.empty(B, dtype=torch.float32, device=dev),))
        self._ws, self._pred_buf = ent
        a = _lib.FcstgnnArgs()
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
        a.sample_offset = int(sample_offset)
        a.dropout_p = float(self.dropout_p)
        a.seed = self._seed
        a.step = int(step)
        a.training = 1 if training else 0
        a.step_state = self._step_state.data_ptr() if self._step_state is not None else None
        a.aux_stream = self.side_stream.pointer(self._flat.device, training)       # the backward's weight / bias gradient GEMMs
        if self.compute_dtype not in ("f32", "bf16"):
            raise RuntimeError(f"compute_dtype must be 'f32' or 'bf16', not {self.compute_dtype!r}")
        a.compute_dtype = _lib.DTYPE_BF16 if self.compute_dtype == "bf16" else _lib.DTYPE_F32
        return a

    def _run_forward(self, x2d, training, step=0):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, training, step)
        _lib.check(_lib.load().rulgnn_fcstgnn_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_fcstgnn_forward_f32")
        return self._pred_buf

    def _run_backward(self, x2d, dpred, step):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, step, dpred=dpred)
        _lib.check(_lib.load().rulgnn_fcstgnn_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_fcstgnn_backward_f32")
        return self._grad_flat

    def _after_train_forward(self, batch=None, from_bucket_moments=False, from_bucket_stats=False):
        """BatchNorm side effects of a training forward.  ``batch``: the (global) batch the statistics were taken over;
        only needed when it differs from the last forward's (data parallel).  ``from_bucket_moments``: the bucket tail holds
        the all-reduced (E[z], E[z^2]) (local BatchNorm); ``from_bucket_stats``: it holds the global (mean, var) (synchronised)."""
        in_bucket = from_bucket_moments or from_bucket_stats
        src = self._grad_flat.data_ptr() + 4 * (self._count + 1) if in_bucket else self._bn_batch.data_ptr()
        shp = self._shape(int(batch) if batch is not None else self._pred_buf.numel())
        _lib.check(_lib.load().rulgnn_fcstgnn_bn_running_update_f32(C.byref(shp), self._bn.data_ptr(), src, 0.1,
                                                                    1 if from_bucket_moments else 0, _stream()),
                   "rulgnn_fcstgnn_bn_running_update_f32")
        self._nbt_pending += 1

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None, sample_offset=0, update_running_stats=True,
                       moments_to_bucket=False):
        """train forward + MSE + backward (+ Adam and the running statistics with ``optimizer``) in one C call."""
        x2d = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x2d.size(0):
            raise RuntimeError("target size mismatch")
        self._step += 1
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, self._step, y=yv, global_batch=global_batch, sample_offset=sample_offset,
                       moments_to_bucket=moments_to_bucket)
        o = self._adam_args(optimizer, bn=self._bn)
        _lib.check(_lib.load().rulgnn_fcstgnn_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_fcstgnn_fwdbwd_f32")
        if optimizer is not None:
            self._nbt_pending += 1
        elif update_running_stats:
            self._after_train_forward(x2d.size(0))
        return self._pred_buf, self._grad_flat[self._count]

    def sync_bn_schedule(self):
        """float64 counts of the all-reduces one synchronised-BatchNorm step issues, in order (dp.py: a rank with an empty shard joins
        them with zeros)."""
        return [128] * 14

    def fused_mse_step_syncbn(self, x, y, global_batch, sample_offset, bn_param_grad_scale, allreduce):
        """``fused_mse_step`` on this rank's shard with every BatchNorm normalising by the GLOBAL batch's statistics (dp.py,
        ``DataParallel(sync_bn=True)``; rulgnn_fcstgnn_fwdbwd_syncbn_f32).  ``allreduce(view)`` is called 14 times with a float64 view of 128 reduction cells
        inside the workspace and must SUM it over the ranks in place, in stream order.  Fills ``self.bucket`` such that a SUM over the
        ranks is the global-batch gradient / loss (the BatchNorm scale / shift gradients are global sums on every rank and enter
        multiplied by ``bn_param_grad_scale``), and ``self._bn_batch`` with the global (mean, biased variance)."""
        x2d = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x2d.size(0):
            raise RuntimeError("target size mismatch")
        self._step += 1
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, self._step, y=yv, global_batch=global_batch, sample_offset=sample_offset)
        ws = self._ws
        cb, user, failure = _lib.allreduce_callback(allreduce, ws)
        rc = _lib.load().rulgnn_fcstgnn_fwdbwd_syncbn_f32(C.byref(shp), C.byref(a), float(bn_param_grad_scale), cb, user, _stream())
        if failure:
            raise failure[0]
        _lib.check(rc, "rulgnn_fcstgnn_fwdbwd_syncbn_f32")
        return self._pred_buf, self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, X):
        x2d = self._check_input(X)
        if x2d.size(0) == 0:
            if self.training:
                raise RuntimeError("training forward needs a non-empty batch")
            return torch.empty(0, 1, dtype=torch.float32, device=x2d.device)
        if self.training:
            if torch.is_grad_enabled():
                return _TrainFunction.apply(self, x2d, *[p for _, p in self._named_live()])
            self._step += 1
            pred = self._run_forward(x2d, True, self._step)
            self._after_train_forward()
            return pred.clone().view(-1, 1)
        return self._run_forward(x2d, False).clone().view(-1, 1)
