"""Drop-in ``STGNN_model`` (the first of the ChebNet users): the graph part -- Gaussian-kernel
top-k adjacency of every (sample, patch) graph, the Chebyshev terms, the ChebNet projection and its filter gradient -- runs
in the gfx950 HIP kernels of csrc/stgnn.hip through one autograd function, the one-layer GRU over the (1-5) patches of every
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

from . import _lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


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


class STGNN_model(nn.Module):
    def __init__(self, patch_size, num_patch, num_nodes, hidden_dim, K, top_k):
        super().__init__()
        self.num_patch, self.patch_size = int(num_patch), int(patch_size)
        self.num_nodes, self.hidden_dim = int(num_nodes), int(hidden_dim)
        self.top_k = int(top_k)
        self.chebnet = ChebNet(self.patch_size, self.hidden_dim, int(K))
        self.gru = nn.GRU(self.hidden_dim, self.hidden_dim, batch_first=True)
        self.fc = nn.Linear(self.hidden_dim * self.num_patch * self.num_nodes, 1)
        self.last_adjacency = None          # filled by forward(x, return_adjacency=True)

    def forward(self, x, return_adjacency=False):
        bs, num_node, time_length = x.shape
        L, f, H = self.num_patch, self.patch_size, self.hidden_dim
        if num_node != self.num_nodes or time_length != L * f:
            raise RuntimeError(f"STGNN_model expects [bs, {self.num_nodes}, {L * f}], got {tuple(x.shape)}")
        shape = (bs, num_node, L, f, H, self.chebnet.K, self.top_k)
        cheb, adj = _ChebFunction.apply(x, self.chebnet.filters, shape, bool(return_adjacency))
        if return_adjacency:
            self.last_adjacency = adj.view(bs, L, num_node, num_node)
        # [bs*L*N, H] -> one sequence of L patches per (sample, node)  (Model.py:93-95)
        seq = cheb.view(bs, L, num_node, H).permute(0, 2, 1, 3).reshape(bs * num_node, L, H)
        g = self.gru                       # the nn.GRU module only holds the parameters
        gru_output = _GruFunction.apply(seq, g.weight_ih_l0, g.weight_hh_l0, g.bias_ih_l0, g.bias_hh_l0)
        return self.fc(gru_output.reshape(bs, -1))
