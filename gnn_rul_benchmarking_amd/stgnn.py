"""Drop-in ``STGNN_model`` (the first of the ChebNet users).  The whole model runs behind three C
entries on one flat parameter buffer (``rulgnn_stgnn_{forward,backward,fwdbwd}_f32``: ``fused_mse_step`` is forward + MSE +
backward + Adam in one call); the autograd pieces ``_ChebFunction`` / ``_GruFunction`` expose the same kernels op by op.
The graph part -- Gaussian-kernel top-k adjacency of every (sample, patch) graph, the Chebyshev terms, the ChebNet projection
and its filter gradient -- runs in the gfx950 HIP kernels of csrc/stgnn.hip, the one-layer GRU over the (1-5) patches of every
(sample, node) pair in the kernels of csrc/gru.hip (the vendor RNN's backward took 6 ms at batch 4096 for this many short
sequences; the ``nn.GRU`` module only holds the parameters); the final ``Linear`` is a library op on the same stream.

Mirrors the reference class (models/STGNN/Model.py:64-107): same constructor kwargs
``(patch_size, num_patch, num_nodes, hidden_dim, K, top_k)``, same ``forward(x) -> [bs, 1]``, the same 7 ``state_dict`` keys
(``chebnet.filters``, ``gru.*_l0``, ``fc.*``) and -- sub-modules being created in the reference's order -- the same initial
weights for a torch seed.  The input carries no gradient and the adjacency depends on the input alone, so the graph
construction is forward-only.  There is no CPU path for the graph part: a non-CUDA input raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn

from . import _lib, params as PL
from .flat import FlatModule, current_stream as _stream


class _ChebFunction(torch.autograd.Function):
    """ChebNet(compute_adjacency_matrix(x), x) for all graphs of a batch (Model.py:83-91)."""

    @staticmethod
    def forward(ctx, x, filters, shape_tuple, want_adj):
        if not x.is_cuda:
            raise RuntimeError("STGNN's graph kernels run on the HIP path only: tensors must be on a CUDA (ROCm) device")
        lib = _lib.load()
        bs, N, L, f, H, K, top_k = shape_tuple
        shp = _lib.StgnnShape(bs, N, L, f, H, K, top_k)
        nbytes = lib.rulgnn_stgnn_workspace_bytes(C.byref(shp))
        if nbytes == 0:
            raise RuntimeError("STGNN HIP kernels do not cover this configuration (num_nodes <= 32, patch_size <= 128, K <= 4, "
                               "top_k <= num_nodes)")
        x = x.contiguous().float()
        filters = filters.contiguous()
        G = bs * L
        terms = torch.empty(G * N, K * f, dtype=torch.float32, device=x.device)
        adj = torch.empty(G, N, N, dtype=torch.float32, device=x.device) if want_adj else None
        _lib.check(lib.rulgnn_stgnn_terms_f32(C.byref(shp), x.data_ptr(), terms.data_ptr(), adj.data_ptr() if want_adj else None,
                                              _stream()), "rulgnn_stgnn_terms_f32")
        out = torch.empty(G * N, H, dtype=torch.float32, device=x.device)
        _lib.check(lib.rulgnn_stgnn_cheb_forward_f32(C.byref(shp), terms.data_ptr(), filters.data_ptr(), out.data_ptr(), _stream()),
                   "rulgnn_stgnn_cheb_forward_f32")
        ctx.save_for_backward(terms)
        ctx.shape_tuple, ctx.nbytes = shape_tuple, nbytes
        ctx.mark_non_differentiable(*([adj] if want_adj else []))
        return (out, adj) if want_adj else (out, None)

    @staticmethod
    def backward(ctx, dout, _dadj):
        (terms,) = ctx.saved_tensors
        lib = _lib.load()
        bs, N, L, f, H, K, top_k = ctx.shape_tuple
        shp = _lib.StgnnShape(bs, N, L, f, H, K, top_k)
        dout = dout.contiguous().float()
        ws = torch.empty(ctx.nbytes, dtype=torch.uint8, device=dout.device)
        dfilters = torch.empty(K, f, H, dtype=torch.float32, device=dout.device)
        _lib.check(lib.rulgnn_stgnn_cheb_backward_f32(C.byref(shp), terms.data_ptr(), dout.data_ptr(), dfilters.data_ptr(), ws.data_ptr(),
                                                      ws.numel(), _stream()), "rulgnn_stgnn_cheb_backward_f32")
        return None, dfilters, None, None


class _GruFunction(torch.autograd.Function):
    """nn.GRU(I, H, batch_first=True)(x)[0] with h0 = 0 on the HIP kernels of csrc/gru.hip (Model.py:97-98)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        if not x.is_cuda:
            raise RuntimeError("STGNN's GRU runs on the HIP path only: tensors must be on a CUDA (ROCm) device")
        lib = _lib.load()
        x = x.contiguous().float()
        S, L, I = x.shape
        H = w_hh.shape[1]
        shp = _lib.GruShape(S, L, I, H)
        nbytes = lib.rulgnn_gru_workspace_bytes(C.byref(shp))
        if nbytes == 0:
            raise RuntimeError("GRU HIP kernels do not cover this configuration (hidden_dim <= 1024, input_dim <= 4096)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty(S, L, H, dtype=torch.float32, device=x.device)
        a = _lib.GruArgs()
        a.x, a.out = x.data_ptr(), out.data_ptr()
        a.w_ih, a.w_hh, a.b_ih, a.b_hh = w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(lib.rulgnn_gru_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_gru_forward_f32")
        ctx.save_for_backward(x, w_ih, w_hh, b_ih, b_hh)
        ctx.ws, ctx.dims = ws, (S, L, I, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh, b_ih, b_hh = ctx.saved_tensors
        S, L, I, H = ctx.dims
        lib = _lib.load()
        dout = dout.contiguous().float()
        shp = _lib.GruShape(S, L, I, H)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw_ih, dw_hh, db_ih, db_hh = (torch.empty_like(t) for t in (w_ih, w_hh, b_ih, b_hh))
        a = _lib.GruArgs()
        a.x, a.dout, a.dx = x.data_ptr(), dout.data_ptr(), dx.data_ptr() if dx is not None else None
        a.w_ih, a.w_hh, a.b_ih, a.b_hh = w_ih.data_ptr(), w_hh.data_ptr(), b_ih.data_ptr(), b_hh.data_ptr()
        a.dw_ih, a.dw_hh, a.db_ih, a.db_hh = dw_ih.data_ptr(), dw_hh.data_ptr(), db_ih.data_ptr(), db_hh.data_ptr()
        a.workspace, a.workspace_bytes = ctx.ws.data_ptr(), ctx.ws.numel()
        _lib.check(lib.rulgnn_gru_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_gru_backward_f32")
        return dx, dw_ih, dw_hh, db_ih, db_hh


class ChebNet(nn.Module):
    """Holder of ``filters`` [K, in_channels, out_channels] with the reference's initialisation (Model.py:29-41)."""

    def __init__(self, in_channels, out_channels, K):
        super().__init__()
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.filters = nn.Parameter(torch.Tensor(K, in_channels, out_channels))
        nn.init.xavier_uniform_(self.filters)


class _Function(torch.autograd.Function):
    """model(x) through the C entries rulgnn_stgnn_forward_f32 / rulgnn_stgnn_backward_f32 (flat parameters)."""

    @staticmethod
    def forward(ctx, model, x, *params):
        pred = model._forward(x).clone()
        ctx.model, ctx.x = model, x
        return pred.view(-1, 1)

    @staticmethod
    def backward(ctx, dout):
        model = ctx.model
        grads = model._backward(ctx.x, dout.reshape(-1).contiguous().float())
        outs = [grads[off:off + n].view(shape).clone() for off, n, shape in model._slices]
        return (None, None, *outs)


def param_layout(patch_size, num_patch, num_nodes, hidden_dim, K):
    """name -> (offset, shape) in the flat buffer the kernels read (include/rulgnn.h), in state_dict order."""
    H = hidden_dim
    layout, off = {}, 0
    for name, shape in (("chebnet.filters", (K, patch_size, H)), ("gru.weight_ih_l0", (3 * H, H)), ("gru.weight_hh_l0", (3 * H, H)),
                        ("gru.bias_ih_l0", (3 * H,)), ("gru.bias_hh_l0", (3 * H,)), ("fc.weight", (1, H * num_patch * num_nodes)),
                        ("fc.bias", (1,))):
        n = 1
        for d in shape:
            n *= d
        layout[name] = (off, shape)
        off += n
    return layout, off


class STGNN_model(FlatModule):
    def __init__(self, patch_size, num_patch, num_nodes, hidden_dim, K, top_k):
        super().__init__()
        self.num_patch, self.patch_size = int(num_patch), int(patch_size)
        self.num_nodes, self.hidden_dim = int(num_nodes), int(hidden_dim)
        self.top_k = int(top_k)
        # same construction order as the reference => same RNG consumption => same initial weights; the sub-modules only hold
        # the parameters (views into one flat buffer), none of them is ever called
        self.chebnet = ChebNet(self.patch_size, self.hidden_dim, int(K))
        self.gru = nn.GRU(self.hidden_dim, self.hidden_dim, batch_first=True)
        self.fc = nn.Linear(self.hidden_dim * self.num_patch * self.num_nodes, 1)
        self.last_adjacency = None          # filled by forward(x, return_adjacency=True)
        self._init_flat(*param_layout(self.patch_size, self.num_patch, self.num_nodes, self.hidden_dim, int(K)))

    workspace_slots = 4

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, batch):
        return _lib.StgnnShape(batch, self.num_nodes, self.num_patch, self.patch_size, self.hidden_dim, self.chebnet.K, self.top_k)

    def _check_input(self, x):
        if x.dim() != 3 or x.size(1) != self.num_nodes or x.size(2) != self.num_patch * self.patch_size:
            raise RuntimeError(f"STGNN_model expects [bs, {self.num_nodes}, {self.num_patch * self.patch_size}], got {tuple(x.shape)}")
        if not x.is_cuda:
            raise RuntimeError("STGNN_model runs on the HIP path only: input must be a CUDA (ROCm) tensor; there is no CPU fallback")
        if x.device != self._flat.device:
            raise RuntimeError(f"input on {x.device} but model on {self._flat.device}")
        return x.contiguous().float()

    def _args(self, shp, x, y=None, dpred=None, global_batch=None):
        B = x.size(0)
        ent = self._workspace_entry(B, lambda: _lib.load().rulgnn_stgnn_step_workspace_bytes(C.byref(shp)),
                                    "STGNN HIP kernels do not cover this configuration (num_nodes <= 32, patch_size <= 128, K <= 4, "
                                    "top_k <= num_nodes)")
        ws, pred = ent
        a = _lib.StmsgcnArgs()
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
        a, pred = self._args(shp, x)
        _lib.check(_lib.load().rulgnn_stgnn_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stgnn_forward_f32")
        return pred[:x.size(0)]

    def _backward(self, x, dpred):
        shp = self._shape(x.size(0))
        a, _ = self._args(shp, x, dpred=dpred)
        _lib.check(_lib.load().rulgnn_stgnn_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_stgnn_backward_f32")
        return self._grad_flat

    def fused_mse_step(self, x, y, optimizer=None, global_batch=None):
        """forward + MSE + backward (+ Adam when ``optimizer`` is a FusedAdam over this model) in one C call; fills
        ``self.bucket`` = [grad | loss]; returns (pred [B], loss 0-d tensor) on the device, no host sync."""
        x = self._check_input(x)
        yv = y.reshape(-1).contiguous().float()
        if yv.numel() != x.size(0):
            raise RuntimeError("target size mismatch")
        shp = self._shape(x.size(0))
        a, pred = self._args(shp, x, y=yv, global_batch=global_batch)
        o = self._adam_args(optimizer)
        _lib.check(_lib.load().rulgnn_stgnn_fwdbwd_f32(C.byref(shp), C.byref(a), o, _stream()), "rulgnn_stgnn_fwdbwd_f32")
        return pred[:x.size(0)], self._grad_flat[self._count]

    def adjacency(self, x):
        """[bs, num_patch, N, N]: compute_adjacency_matrix of the reference (Model.py:8-25), for inspection / tests."""
        x = self._check_input(x)
        bs = x.size(0)
        shp = self._shape(bs)
        G = bs * self.num_patch
        terms = torch.empty(G * self.num_nodes, self.chebnet.K * self.patch_size, dtype=torch.float32, device=x.device)
        adj = torch.empty(G, self.num_nodes, self.num_nodes, dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().rulgnn_stgnn_terms_f32(C.byref(shp), x.data_ptr(), terms.data_ptr(), adj.data_ptr(), _stream()),
                   "rulgnn_stgnn_terms_f32")
        return adj.view(bs, self.num_patch, self.num_nodes, self.num_nodes)

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, x, return_adjacency=False):
        x = self._check_input(x)
        if x.size(0) == 0:              # like the reference: reshape(0, -1) is ambiguous (Model.py:101)
            raise RuntimeError("cannot reshape tensor of 0 elements into shape [0, -1] because the unspecified dimension size -1 can be "
                               "any value and is ambiguous")
        if return_adjacency:
            self.last_adjacency = self.adjacency(x)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._named()):
            return _Function.apply(self, x, *self._named())
        return self._forward(x).clone().view(-1, 1)
