"""Drop-in ``ASTGCNN_model`` whose forward/backward run in the gfx950 HIP kernels (csrc/astgcnn.hip).

Mirrors the reference class (models/ASTGCNN/Model.py:233-254): same constructor kwargs
``(num_nodes, time_length, encoder_out_dim, output_dim, K)``, same ``forward(X) -> [bs, 1]``, the same 29
``state_dict`` keys (including the never-called ``tcn.net0`` / ``tcn.net1`` branches, Model.py:86-109) and -- because the
parameter-holding sub-modules are created in the reference's order -- the same initial weights for a torch seed.  None of
the sub-modules is ever *called*: the live parameters are views into one flat fp32 buffer that the kernels read directly
(layout in include/rulgnn.h), the BatchNorm running statistics views into a second one.

There is no CPU path: calling the model with a non-CUDA tensor raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream
from .stgcn import TemporalConvNet

TCN_KERNEL = 6          # Model.py:236


class GatingMechanism(nn.Module):
    """Holder of ``theta`` = Linear(time_length, encoder_out_dim) and ``bias`` (Model.py:169-173)."""

    def __init__(self, num_channels, out_channels):
        super().__init__()
        self.theta = nn.Linear(num_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))


class construct_graph(nn.Module):
    """Holder of ``P`` = Linear(f, f, bias=False) (Model.py:184-187)."""

    def __init__(self, num_features):
        super().__init__()
        self.P = nn.Linear(num_features, num_features, bias=False)


class ChebNet(nn.Module):
    """Holder of ``filters`` [K, in, out], xavier-uniform (Model.py:198-209)."""

    def __init__(self, in_channels, out_channels, K):
        super().__init__()
        self.filters = nn.Parameter(torch.Tensor(K, in_channels, out_channels))
        nn.init.xavier_uniform_(self.filters)


def live_layout(num_nodes, time_length, output_dim, K):
    """state_dict name (without the algorithm's ``model.`` prefix) -> (offset, shape) in the flat buffer."""
    N, T, E, O = num_nodes, time_length, time_length, output_dim
    out, off = {}, 0
    for name, shape in (("tcn.conv_block1.0.weight", (N, N, TCN_KERNEL)), ("tcn.conv_block1.2.weight", (N,)),
                        ("tcn.conv_block1.2.bias", (N,)), ("tcn.conv_block2.0.weight", (N, N, TCN_KERNEL)),
                        ("tcn.conv_block2.2.weight", (N,)), ("tcn.conv_block2.2.bias", (N,)),
                        ("gate.theta.weight", (E, T)), ("gate.theta.bias", (E,)), ("gate.bias", (E,)),
                        ("distance_module.P.weight", (E, E)), ("chebnet.filters", (K, E, O)), ("fc.weight", (1, O)),
                        ("fc.bias", (1,))):
        n = 1
        for s in shape:
            n *= s
        out[name] = (off, shape)
        off += n
    return out, off


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


class ASTGCNN_model(FlatModule):
    def __init__(self, num_nodes, time_length, encoder_out_dim, output_dim, K):
        super().__init__()
        self.num_nodes, self.time_length = int(num_nodes), int(time_length)
        self.encoder_out_dim, self.output_dim, self.K = int(encoder_out_dim), int(output_dim), int(K)
        # same construction order as the reference => same RNG consumption => same initial weights
        self.tcn = TemporalConvNet(self.num_nodes, [self.num_nodes, self.num_nodes], kernel_size=TCN_KERNEL)
        self.gate = GatingMechanism(self.time_length, self.encoder_out_dim)
        self.distance_module = construct_graph(self.encoder_out_dim)
        self.chebnet = ChebNet(self.encoder_out_dim, self.output_dim, self.K)
        self.fc = nn.Linear(self.output_dim, 1)

        self._bn_names = [f"tcn.conv_block{b}.2.running_{k}" for b in (1, 2) for k in ("mean", "var")]
        self._bn = self._bn_batch = self._pred_buf = self._ws = None
        # (off by default: with the step's five parameter-gradient products as one launch pair behind the backward chain, one stream is
        # faster -- 0.119 vs 0.122 ms per step at N-CMAPSS batch 512; ``enabled = True`` runs the pair beside the TCN backward instead)
        self.side_stream = PL.SideStream()
        self.side_stream.enabled = False
        self._track_batchnorm_counters()
        self._init_flat(*live_layout(self.num_nodes, self.time_length, self.output_dim, self.K))

    # ---- flat storage ----------------------------------------------------------------------------------
    workspace_slots = 4

    def _bucket_floats(self):
        return self._count + 1 + 4 * self.num_nodes                   # [gradient | loss | BatchNorm batch moments]

    def _reflatten_buffers(self, dev):
        N = self.num_nodes
        bufs = dict(self.named_buffers())
        bn = torch.empty(4 * N, dtype=torch.float32, device=dev)
        nbt = torch.zeros(2, dtype=torch.int64, device=dev)
        for i, name in enumerate(self._bn_names):
            bn[i * N:(i + 1) * N].copy_(bufs[name].detach().float())
            self._set_buffer(name, bn[i * N:(i + 1) * N])
        for b in (1, 2):
            cname = f"tcn.conv_block{b}.2.num_batches_tracked"
            nbt[b - 1].copy_(bufs[cname])
            self._set_buffer(cname, nbt[b - 1])
        self._bn, self._nbt = bn, nbt
        self._bn_batch = torch.zeros(4 * N, dtype=torch.float32, device=dev)

    def _reset_caches(self):
        super()._reset_caches()
        self._pred_buf = self._ws = None

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        if self.encoder_out_dim != self.time_length:
            raise RuntimeError(f"The size of tensor a ({self.encoder_out_dim}) must match the size of tensor b "
                               f"({self.time_length}) at non-singleton dimension 2")      # what the reference's gate raises
        return _lib.AstgcnnShape(batch, self.num_nodes, self.time_length, self.output_dim, self.K)

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("ASTGCNN_model runs on the HIP kernels only: input must be a CUDA (ROCm) tensor; "
                               "there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        if x.dim() != 3 or x.size(1) != self.num_nodes or x.size(2) != self.time_length:
            raise RuntimeError(f"expected input [bs, {self.num_nodes}, {self.time_length}], got {list(x.shape)}")
        return x.reshape(x.size(0), -1).contiguous().float()

    def _args(self, shp, x2d, training, y=None, dpred=None, global_batch=None, moments_to_bucket=False):
        B = x2d.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_astgcnn_workspace_bytes(C.byref(shp)),
                                    "ASTGCNN kernels do not cover this configuration (num_nodes <= 25, time_length <= 64, "
                                    "output_dim <= 256, K <= 3)", make=lambda dev: (torch.empty(B, dtype=torch.float32, device=dev),))
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
        a.aux_stream = self.side_stream.pointer(self._flat.device, training)
        return a

    def _run_forward(self, x2d, training):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, training)
        _lib.check(_lib.load().rulgnn_astgcnn_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_astgcnn_forward_f32")
        return self._pred_buf

    def _run_backward(self, x2d, dpred):
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, dpred=dpred)
        _lib.check(_lib.load().rulgnn_astgcnn_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_astgcnn_backward_f32")
        return self._grad_flat

    def _after_train_forward(self, batch, from_bucket_moments=False, from_bucket_stats=False):
        """BatchNorm side effects of a training forward (running stats, num_batches_tracked).  ``from_bucket_moments``: the bucket
        tail holds the all-reduced (E[z], E[z^2]) (local BatchNorm); ``from_bucket_stats``: the global (mean, var) (synchronised)."""
        in_bucket = from_bucket_moments or from_bucket_stats
        src = self._grad_flat.data_ptr() + 4 * (self._count + 1) if in_bucket else self._bn_batch.data_ptr()
        shp = self._shape(batch)
        _lib.check(_lib.load().rulgnn_astgcnn_bn_running_update_f32(C.byref(shp), self._bn.data_ptr(), src,
                                                                    batch * self.time_length, 0.1,
                                                                    1 if from_bucket_moments else 0, _stream()),
                   "rulgnn_astgcnn_bn_running_update_f32")
        self._nbt_pending += 1

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None, sample_offset=0, update_running_stats=True,
                       moments_to_bucket=False):
        """train forward + MSE + backward (+ Adam and the running-statistics update when ``optimizer`` is a FusedAdam over
        this model) in one C call; fills ``self.bucket``; returns (pred [B], loss 0-d tensor) on the device."""
        x2d = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x2d.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, y=yv, global_batch=global_batch, moments_to_bucket=moments_to_bucket)
        o = self._adam_args(optimizer, bn=self._bn)
        _lib.check(_lib.load().rulgnn_astgcnn_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_astgcnn_fwdbwd_f32")
        if optimizer is not None:
            self._nbt_pending += 1
        elif update_running_stats:
            self._after_train_forward(x2d.size(0))
        return self._pred_buf, self._grad_flat[self._count]

    def sync_bn_schedule(self):
        """float64 counts of the all-reduces one synchronised-BatchNorm step issues, in order (dp.py: a rank with an empty shard joins
        them with zeros)."""
        return [50] * 4

    def fused_mse_step_syncbn(self, x, y, global_batch, sample_offset, bn_param_grad_scale, allreduce):
        """``fused_mse_step`` on this rank's shard with every BatchNorm normalising by the GLOBAL batch's statistics (dp.py,
        ``DataParallel(sync_bn=True)``; rulgnn_astgcnn_fwdbwd_syncbn_f32).  ``allreduce(view)`` is called 4 times with a float64 view of 50 reduction cells
        inside the workspace and must SUM it over the ranks in place, in stream order.  Fills ``self.bucket`` such that a SUM over the
        ranks is the global-batch gradient / loss (the BatchNorm scale / shift gradients are global sums on every rank and enter
        multiplied by ``bn_param_grad_scale``), and ``self._bn_batch`` with the global (mean, biased variance)."""
        x2d = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x2d.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x2d.size(0))
        a = self._args(shp, x2d, True, y=yv, global_batch=global_batch)
        ws = self._ws
        cb, user, failure = _lib.allreduce_callback(allreduce, ws)
        rc = _lib.load().rulgnn_astgcnn_fwdbwd_syncbn_f32(C.byref(shp), C.byref(a), float(bn_param_grad_scale), cb, user, _stream())
        if failure:
            raise failure[0]
        _lib.check(rc, "rulgnn_astgcnn_fwdbwd_syncbn_f32")
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
            pred = self._run_forward(x2d, training=True)
            self._after_train_forward(x2d.size(0))
            return pred.clone().view(-1, 1)
        return self._run_forward(x2d, training=False).clone().view(-1, 1)
