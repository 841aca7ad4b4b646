"""Drop-in ``STNet_model`` (a ``ChebNet`` user).  The whole model runs behind three C entries on one flat
parameter buffer (``rulgnn_stnet_{forward,backward,fwdbwd}_f32``; ``fused_mse_step`` is forward + MSE + reconstruction loss +
backward + Adam in one call): the STFT front end, the thresholded adjacency and the Chebyshev terms in the gfx950 kernels of
csrc/stnet.hip, every projection as a matrix-core GEMM, the LSTM over the patches in the persistent kernels of csrc/bilstm.hip.

Mirrors the reference class (models/STNet/Model.py:44-169): same constructor kwargs ``(num_patch, patch_size, num_nodes, nperseg,
input_dim, Cheb_layers, lstm_hidden_dim, autoencoder_hidden_dim)``, ``forward(x, train=False)`` returning the prediction ``[bs, 1]``
or ``(prediction, reconstruction_loss)``, the same ``state_dict`` keys in the same order and -- sub-modules being created in the
reference's order -- the same initial weights for a torch seed.  ``cnn`` (the 1x1 convolution behind the 0.7 threshold) never
receives a gradient, exactly as in the reference.  There is no CPU path: a non-CUDA input raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream


class ChebNet(nn.Module):
    """Holder of ``filters`` [K, in_channels, out_channels] with the reference's initialisation (Model.py:7-20)."""

    def __init__(self, in_channels, out_channels, K):
        super().__init__()
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.filters = nn.Parameter(torch.Tensor(K, in_channels, out_channels))
        nn.init.xavier_uniform_(self.filters)


class _Function(torch.autograd.Function):
    """model(x, train=True) through rulgnn_stnet_forward_f32 / rulgnn_stnet_backward_f32.  The reconstruction loss is returned as a
    0-d tensor whose incoming gradient must be 1 (it is a term of the reference's loss): the backward entry folds it in."""

    @staticmethod
    def forward(ctx, model, x, *params):
        pred, recon = model._forward(x)
        ctx.model, ctx.x = model, x
        ctx.tape = model._tape.tokens[x.size(0)]
        ctx.set_materialize_grads(False)
        return pred.clone().view(-1, 1), recon.clone()

    @staticmethod
    def backward(ctx, dpred, drecon):
        # The reconstruction term enters with whatever weight the objective gave it (the reference: 1, algorithms.py:458; a loss on the
        # prediction alone: none) -- handed to the kernels as a device scalar, no host round trip.  A backward of the forward's own
        # workspace: run once per forward (the reconstruction gradients are scaled in place).
        model = ctx.model
        model._tape.check(ctx.x.size(0), ctx.tape, model._bufs, "STNet_model")
        B = ctx.x.size(0)
        dp = dpred.reshape(-1).contiguous().float() if dpred is not None else torch.zeros(B, dtype=torch.float32, device=ctx.x.device)
        w = drecon.reshape(1).contiguous().float() if drecon is not None else torch.zeros(1, dtype=torch.float32, device=ctx.x.device)
        grads = model._backward(ctx.x, dp, recon_weight=w)
        model._tape.consume(ctx.x.size(0), ctx.tape)
        outs = [grads[off:off + n].view(shape).clone() if i >= 2 else None for i, (off, n, shape) in enumerate(model._slices)]
        return (None, None, *outs)


class STNet_model(FlatModule):
    def __init__(self, num_patch, patch_size, num_nodes, nperseg, input_dim, Cheb_layers, lstm_hidden_dim, autoencoder_hidden_dim):
        super().__init__()
        self.num_patch, self.patch_size, self.nperseg = int(num_patch), int(patch_size), int(nperseg)
        self.num_nodes, self.input_dim = int(num_nodes), int(input_dim)
        self.cheb_layers = [int(c) for c in Cheb_layers]
        self.lstm_hidden_dim, self.autoencoder_hidden_dim = int(lstm_hidden_dim), int(autoencoder_hidden_dim)
        dims = [self.input_dim] + self.cheb_layers
        A = self.autoencoder_hidden_dim
        # same construction order as the reference => same RNG consumption => same initial weights; never called
        self.cnn = nn.Conv2d(in_channels=2, out_channels=1, kernel_size=(1, 1))
        self.chebnets = nn.ModuleList([ChebNet(dims[i], dims[i + 1], 3) for i in range(len(dims) - 1)])
        self.encoder = nn.Sequential(nn.Linear(dims[-1] * self.num_nodes, A), nn.ReLU(), nn.Linear(A, A), nn.ReLU(), nn.Linear(A, A), nn.ReLU(),
                                     nn.Linear(A, A))
        self.decoder = nn.Sequential(nn.Linear(A, A), nn.ReLU(), nn.Linear(A, A), nn.ReLU(), nn.Linear(A, A), nn.ReLU(),
                                     nn.Linear(A, dims[-1] * self.num_nodes))
        self.lstm = nn.LSTM(input_size=A, hidden_size=self.lstm_hidden_dim, batch_first=True)
        self.linear = nn.Linear(self.lstm_hidden_dim * self.num_patch, 1)
        self._tape = PL.ForwardTape()
        self._init_flat()
        self.optimized_range = (3, self._count)  # cnn.weight [2] + cnn.bias [1] come first and have no gradient

    bucket_tail = 2            # [gradient | loss | reconstruction]

    @property
    def bucket(self):
        """[gradient | loss]: what one all-reduce carries in data-parallel training (the reconstruction term sits behind it)."""
        return self._grad_flat[:self._count + 1]

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        s = _lib.StnetShape()
        s.batch, s.num_patch, s.patch_size, s.num_nodes, s.nperseg, s.input_dim = batch, self.num_patch, self.patch_size, self.num_nodes, self.nperseg, self.input_dim
        s.num_cheb = len(self.cheb_layers)
        for i, c in enumerate(self.cheb_layers[:4]):
            s.cheb_layers[i] = c
        s.lstm_hidden_dim, s.autoencoder_hidden_dim = self.lstm_hidden_dim, self.autoencoder_hidden_dim
        return s

    def _check_input(self, x):
        if not x.is_cuda:
            raise RuntimeError("STNet_model runs on the HIP path only: input must be a CUDA (ROCm) tensor; there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        bs = x.size(0)
        if x.numel() != bs * self.num_patch * self.patch_size:
            raise RuntimeError(f"shape '[{bs}, {self.num_patch}, {self.patch_size}]' is invalid for input of size {x.numel()}")
        if len(self.cheb_layers) > 4:
            raise RuntimeError("STNet HIP kernels cover up to 4 ChebNet layers")
        return x.reshape(bs, self.num_patch * self.patch_size).contiguous().float()

    def _args(self, shp, x, y=None, dpred=None, global_batch=None):
        B = x.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_stnet_workspace_bytes(C.byref(shp)),
                                    "STNet HIP kernels do not cover this configuration (num_nodes = nperseg / 2 + 1, input_dim = 1 + "
                                    "patch_size / nperseg, even nperseg <= 64, <= 4 ChebNets)")
        ws, pred = ent
        a = _lib.StnetArgs()
        a.x = x.data_ptr()
        a.y = y.data_ptr() if y is not None else None
        a.dpred = dpred.data_ptr() if dpred is not None else None
        a.params, a.grads = self._flat.data_ptr(), self._grad_flat.data_ptr()
        a.pred = pred.data_ptr()
        a.loss = self._grad_flat.data_ptr() + 4 * self._count
        a.recon = self._grad_flat.data_ptr() + 4 * (self._count + 1)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        a.global_batch = B if global_batch is None else int(global_batch)
        return a, pred

    def _forward(self, x):
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred = self._args(shp, x)
        _lib.check(_lib.load().rulgnn_stnet_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stnet_forward_f32")
        return pred[:x.size(0)], self._grad_flat[self._count + 1]

    def _backward(self, x, dpred, recon_weight=None):
        shp = self._shape(x.size(0))
        a, _ = self._args(shp, x, dpred=dpred)
        a.recon_weight = recon_weight.data_ptr() if recon_weight is not None else None
        _lib.check(_lib.load().rulgnn_stnet_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stnet_backward_f32")
        return self._grad_flat

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None):
        """forward + MSE + reconstruction loss + backward (+ Adam when ``optimizer`` is a FusedAdam over this model) in one C call;
        fills ``self.bucket`` = [grad | loss]; returns (pred [B], loss 0-d tensor) on the device, no host sync."""
        x = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x.size(0))
        self._tape.mark(x.size(0))
        a, pred = self._args(shp, x, y=yv, global_batch=global_batch)
        o = self._adam_args(optimizer)
        _lib.check(_lib.load().rulgnn_stnet_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_stnet_fwdbwd_f32")
        return pred[:x.size(0)], self._grad_flat[self._count]

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, x, train=False):
        x2 = self._check_input(x)
        if x2.size(0) == 0:
            raise RuntimeError("STNet_model: empty batch")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._named()):
            pred, recon = _Function.apply(self, x2, *self._named())
        else:
            p, r = self._forward(x2)
            pred, recon = p.clone().view(-1, 1), r.clone()
        return (pred, recon) if train else pred
