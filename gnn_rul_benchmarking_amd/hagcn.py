"""Drop-in ``HAGCN_model``: the graph stack (cosine adjacency, three GIN + SAGPool levels with top-k node selection and the
KL prior, node means) runs in the gfx950 HIP kernels of csrc/hagcn.hip through one autograd function, and each of the three
bidirectional LSTM layers in front of it in the persistent-recurrence kernels of csrc/bilstm.hip (the reference's axis
convention gives the LSTMs 1-5 sequences of batch*nodes = thousands of steps: a per-step-launch library RNN is slower than the
reference's CPU there).  The ``nn.LSTM`` modules only hold the parameters; dropout, LeakyReLU and the two-layer ``fc`` are
plain torch ops.

Mirrors the reference class (models/HAGCN/Model.py:129-195): same constructor kwargs
``(patch_size, num_patch, encoder_hidden_dim, hidden_dim, output_dim)``, same ``forward(X, train=False)`` returning the
prediction or ``(prediction, total_kl)``, the same 67 ``state_dict`` keys and -- sub-modules being created in the
reference's order -- the same initial weights for a torch seed.  The LSTM stack keeps the reference's axis convention
(sequence axis = batch*nodes, batch axis = num_patch, Model.py:153-157), which couples all samples of a batch: data
parallelism for this model is replicas only.

The graph-stack parameters are views into one flat fp32 buffer that the kernels read directly (layout in
include/rulgnn.h).  There is no CPU path for the graph stack: a non-CUDA input raises.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, params as PL


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class GINLayer(nn.Module):
    """Holder of ``eps`` and ``mlp`` (Model.py:6-14)."""

    def __init__(self, input_dim, hidden_dim):
        super().__init__()
        self.eps = nn.Parameter(torch.Tensor([0]))
        self.mlp = nn.Sequential(nn.Linear(input_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim))


class SAGPool(nn.Module):
    """Holder of ``rank``, ``model`` and ``mlp`` (Model.py:75-87)."""

    def __init__(self, input_dimension, output_dimension, n):
        super().__init__()
        self.rank = nn.Linear(input_dimension, 1)
        self.model = nn.Linear(input_dimension, output_dimension)
        self.n = n
        self.mlp = nn.Sequential(nn.Linear(input_dimension, input_dimension // 2), nn.ReLU(), nn.Linear(input_dimension // 2, 1))


# The side stream that carries the LSTM layers' parameter-gradient GEMMs while a ``deferred_weight_gradients`` block is open (a plain
# module global, not a thread-local: autograd runs the backward on its own thread).  Layer l's dW_ih / dW_hh / db GEMMs (~40 launches of
# 5-16 us per layer) then run UNDER layer l-1's BPTT -- a persistent recurrence on two of the 256 CUs -- instead of in front of it.
_DEFERRED = [None]


class deferred_weight_gradients:
    """``with deferred_weight_gradients(device): loss.backward()`` -- the Bi-LSTM layers' parameter gradients are produced on a side stream;
    leaving the block makes the current stream wait for it, so they are final (in stream order) for whatever follows: ``optimizer.step()``
    in ``HAGCN.update``.  Outside such a block every gradient is produced on the current stream as before."""

    def __init__(self, device):
        self.device = torch.device(device)

    def __enter__(self):
        if self.device.type == "cuda":
            if getattr(deferred_weight_gradients, "_stream", None) is None or deferred_weight_gradients._stream.device != self.device:
                deferred_weight_gradients._stream = torch.cuda.Stream(device=self.device)
            _DEFERRED[0] = deferred_weight_gradients._stream
        return self

    def __exit__(self, *exc):
        side, _DEFERRED[0] = _DEFERRED[0], None
        if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        return False


def _accumulate_grad_steals(leaves):
    """True when autograd's AccumulateGrad will adopt a fresh gradient tensor of every leaf untouched: ``.grad is None``, no tensor or
    post-accumulate hooks, and the backward is not itself recorded (``create_graph``: grad mode is on inside it)."""
    if torch.is_grad_enabled():
        return False
    for p in leaves:
        if p.grad is not None or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
            return False
    return True


class _BiLstmSum(torch.autograd.Function):
    """out = LSTM_forward(x) + LSTM_reverse(x) for one nn.LSTM(bidirectional=True, batch_first=True) (Model.py:58-61)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r):
        if not x.is_cuda:
            raise RuntimeError("the bidirectional LSTM layers run on the HIP kernels only: tensors must be on a CUDA (ROCm) device")
        x = x.contiguous().float()
        Bq, T, I = x.shape
        H = w_hh.shape[1]
        shp = _lib.BilstmShape(T, Bq, I, H)
        nbytes = _lib.load().rulgnn_bilstm_workspace_bytes(C.byref(shp))
        if nbytes == 0:
            raise RuntimeError("bidirectional LSTM kernels do not cover this configuration (hidden <= 128, <= 64 sequences)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        out = torch.empty(Bq, T, H, dtype=torch.float32, device=x.device)
        a = _lib.BilstmArgs()
        a.x, a.out = x.data_ptr(), out.data_ptr()
        for d, (wi, wh, bi, bh) in enumerate(((w_ih, w_hh, b_ih, b_hh), (w_ih_r, w_hh_r, b_ih_r, b_hh_r))):
            a.w_ih[d], a.w_hh[d], a.b_ih[d], a.b_hh[d] = wi.data_ptr(), wh.data_ptr(), bi.data_ptr(), bh.data_ptr()
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(_lib.load().rulgnn_bilstm_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_bilstm_forward_f32")
        ctx.save_for_backward(x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        ctx.ws, ctx.shp = ws, (T, Bq, I, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r = ctx.saved_tensors
        T, Bq, I, H = ctx.shp
        shp = _lib.BilstmShape(T, Bq, I, H)
        dout = dout.contiguous().float()
        grads = [torch.empty_like(t) for t in (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        a = _lib.BilstmArgs()
        a.x, a.dout, a.dx = x.data_ptr(), dout.data_ptr(), dx.data_ptr() if dx is not None else None
        for d, (wi, wh) in enumerate(((w_ih, w_hh), (w_ih_r, w_hh_r))):
            a.w_ih[d], a.w_hh[d] = wi.data_ptr(), wh.data_ptr()
            a.dw_ih[d], a.dw_hh[d], a.db_ih[d], a.db_hh[d] = (grads[4 * d].data_ptr(), grads[4 * d + 1].data_ptr(),
                                                              grads[4 * d + 2].data_ptr(), grads[4 * d + 3].data_ptr())
        a.workspace, a.workspace_bytes = ctx.ws.data_ptr(), ctx.ws.numel()
        side = _DEFERRED[0]
        # The deferred gradients are handed to autograd BEFORE the side stream has written them: sound only when AccumulateGrad
        # takes the tensor as it is (first gradient of the parameter, no hooks, no graph through the backward).  With
        # ``zero_grad(set_to_none=False)``, gradient accumulation or hooks it would read (``grad += new``, ``clone()``) on the main
        # stream ahead of the GEMMs -- so in those cases nothing is deferred.
        leaves = (w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r)
        if side is not None and not _accumulate_grad_steals(leaves):
            side = None
        if side is not None and side.device == x.device:
            a.aux_stream = side.cuda_stream
            for t in (ctx.ws, x, *grads):          # the side stream reads / writes them after this function has returned
                t.record_stream(side)
        _lib.check(_lib.load().rulgnn_bilstm_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_bilstm_backward_f32")
        return (dx, *grads)


def bilstm_sum(lstm: nn.LSTM, x):
    """The two halves of a bidirectional ``nn.LSTM`` output, summed -- on the persistent-recurrence kernels."""
    return _BiLstmSum.apply(x, lstm.weight_ih_l0, lstm.weight_hh_l0, lstm.bias_ih_l0, lstm.bias_hh_l0, lstm.weight_ih_l0_reverse,
                            lstm.weight_hh_l0_reverse, lstm.bias_ih_l0_reverse, lstm.bias_hh_l0_reverse)


class Bi_LSTM_Standard(nn.Module):
    """The reference's LSTM stack (Model.py:26-73); the nn.LSTM modules hold the parameters, csrc/bilstm.hip runs them."""

    def __init__(self, input_dim, num_hidden, time_length):
        super().__init__()
        self.bi_lstm1 = nn.LSTM(input_size=input_dim, hidden_size=num_hidden, num_layers=1, batch_first=True, dropout=0,
                                bidirectional=True)
        self.drop1 = nn.Dropout(p=0.2)
        self.bi_lstm2 = nn.LSTM(input_size=num_hidden, hidden_size=num_hidden * 2, num_layers=1, batch_first=True, dropout=0,
                                bidirectional=True)
        self.drop2 = nn.Dropout(p=0.2)
        self.bi_lstm3 = nn.LSTM(input_size=num_hidden * 2, hidden_size=num_hidden, num_layers=1, batch_first=True,
                                bidirectional=True)
        self.drop3 = nn.Dropout(p=0.2)

    def forward(self, x):
        x = bilstm_sum(self.bi_lstm1, x)
        x = self.drop2(bilstm_sum(self.bi_lstm2, x))
        return F.leaky_relu(self.drop3(bilstm_sum(self.bi_lstm3, x)))


GRAPH_LEAVES = ("eps", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")
POOL_LEAVES = ("rank.weight", "rank.bias", "model.weight", "model.bias", "mlp.0.weight", "mlp.0.bias", "mlp.2.weight", "mlp.2.bias")


def graph_param_names():
    names = []
    for l in (1, 2, 3):
        names += [f"gin{l}.{leaf}" for leaf in GRAPH_LEAVES] + [f"gnn{l}.{leaf}" for leaf in POOL_LEAVES]
    return names


class _GraphFunction(torch.autograd.Function):
    """(nodes [G, N, enc], graph parameters) -> (feats [G, 3h], kl []) on the HIP kernels."""

    @staticmethod
    def forward(ctx, model, nodes, *params):
        feats, kl = model._graph_forward(nodes)
        ctx.model = model
        ctx.shape = tuple(nodes.shape)
        ctx.ws = model._ws                      # the tape lives in the workspace: keep it until backward
        return feats, kl

    @staticmethod
    def backward(ctx, dfeats, dkl):
        model = ctx.model
        dnodes, grads = model._graph_backward(ctx.shape, ctx.ws, dfeats.contiguous().float(), dkl.reshape(1).contiguous().float())
        out = [grads[off:off + n].view(shape) for (off, n, shape) in model._slices]        # views of this backward's own buffer
        return (None, dnodes, *out)


class HAGCN_model(nn.Module):
    # the Bi-LSTM recurs along batch x nodes (Model.py:153-157): a sample's prediction depends on the samples in front of it in the batch
    eval_sample_independent = False

    def __init__(self, patch_size, num_patch, encoder_hidden_dim, hidden_dim, output_dim):
        super().__init__()
        self.patch_size, self.num_patch = int(patch_size), int(num_patch)
        self.enc_dim, self.hidden_dim = int(encoder_hidden_dim), int(hidden_dim)
        # same construction order as the reference (Model.py:135-147) => same RNG consumption => same initial weights
        self.TD = Bi_LSTM_Standard(patch_size, encoder_hidden_dim, None)
        self.gin1 = GINLayer(encoder_hidden_dim, hidden_dim)
        self.gnn1 = SAGPool(hidden_dim, hidden_dim, 10)
        self.gin2 = GINLayer(hidden_dim, hidden_dim)
        self.gnn2 = SAGPool(hidden_dim, hidden_dim, 5)
        self.gin3 = GINLayer(hidden_dim, hidden_dim)
        self.gnn3 = SAGPool(hidden_dim, hidden_dim, 1)
        self.fc = nn.Sequential(nn.Linear(hidden_dim * 3 * num_patch, output_dim), nn.ReLU(inplace=True), nn.Linear(output_dim, 1))

        table = dict(self.named_parameters())
        self._layout, self._slices, off = {}, [], 0
        for name in graph_param_names():
            p = table[name]
            self._layout[name] = (off, tuple(p.shape))
            self._slices.append((off, p.numel(), tuple(p.shape)))
            off += p.numel()
        self._count = off
        self._flat = self._grad_flat = self._ws = None
        self.last_topk = None                   # [G, 16] int32: node indices the last forward kept (10 | 5 | 1)
        self.forced_topk = None                 # optional [G, 16] int32 imposed on the next forwards (parity checks under ties)
        self._reflatten()

    # ---- flat storage of the graph-stack parameters ----------------------------------------------------
    def _graph_params(self):
        table = dict(self.named_parameters())
        return [table[name] for name in self._layout]

    def _reflatten(self):
        ps = self._graph_params()
        dev = ps[0].device
        flat = torch.empty(self._count, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, (off, n, shape) in zip(ps, self._slices):
                flat[off:off + n].copy_(p.detach().reshape(-1).float())
                p.data = flat[off:off + n].view(shape)
        self._flat = flat
        self._grad_flat = torch.zeros(self._count, dtype=torch.float32, device=dev)
        self._ws = None
        PL.mark_flat_views(self)

    def _apply(self, fn, recurse=True):
        super()._apply(fn)
        if not PL.flat_views_intact(self):      # a no-op .to(device) (every epoch in the trainers) keeps the buffers
            self._reflatten()                   # a real move converts tensors one by one: rebuild the flat views
        return self

    @property
    def flat_params(self):
        return self._flat

    # ---- C-ABI calls -----------------------------------------------------------------------------------
    def _shape(self, graphs, num_node):
        return _lib.HagcnShape(graphs, num_node, self.enc_dim, self.hidden_dim)

    def _graph_forward(self, nodes):
        if not nodes.is_cuda:
            raise RuntimeError("HAGCN_model's graph stack runs on the HIP kernels only: tensors must be on a CUDA (ROCm) device; "
                               "there is no CPU fallback")
        nodes = nodes.contiguous().float()
        G, N, _ = nodes.shape
        shp = self._shape(G, N)
        nbytes = _lib.load().rulgnn_hagcn_workspace_bytes(C.byref(shp))
        if nbytes == 0:
            raise RuntimeError("HAGCN graph kernels do not cover this configuration (10 <= num_node <= 20, encoder_hidden_dim <= 64, "
                               "hidden_dim even and <= 64)")
        # a fresh workspace per forward: it carries the tape to this forward's backward
        self._ws = torch.empty(nbytes, dtype=torch.uint8, device=nodes.device)
        feats = torch.empty(G, 3 * self.hidden_dim, dtype=torch.float32, device=nodes.device)
        kl = torch.empty(1, dtype=torch.float32, device=nodes.device)
        topk = torch.zeros(G, _lib.HAGCN_TOPK_SLOTS, dtype=torch.int32, device=nodes.device)
        a = _lib.HagcnArgs()
        a.nodes, a.params, a.feats, a.kl, a.topk = nodes.data_ptr(), self._flat.data_ptr(), feats.data_ptr(), kl.data_ptr(), topk.data_ptr()
        if self.forced_topk is not None:
            f = self.forced_topk.to(device=nodes.device, dtype=torch.int32).contiguous()
            if tuple(f.shape) != (G, _lib.HAGCN_TOPK_SLOTS):
                raise RuntimeError("forced_topk must be [graphs, 16] int32")
            a.forced_topk = f.data_ptr()
            self._forced_keepalive = f
        a.workspace, a.workspace_bytes = self._ws.data_ptr(), self._ws.numel()
        _lib.check(_lib.load().rulgnn_hagcn_graph_forward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_hagcn_graph_forward_f32")
        self.last_topk = topk
        return feats, kl.reshape(())

    def _graph_backward(self, shape, ws, dfeats, dkl):
        G, N, enc = shape
        shp = self._shape(G, N)
        dnodes = torch.empty(G, N, enc, dtype=torch.float32, device=dfeats.device)
        a = _lib.HagcnArgs()
        # a fresh gradient buffer per backward (the kernels write every slot): autograd gets views of it instead of ~30 copies
        grads = torch.empty(self._count, dtype=torch.float32, device=dfeats.device)
        a.params, a.dfeats, a.dkl, a.dnodes, a.grads = (self._flat.data_ptr(), dfeats.data_ptr(), dkl.data_ptr(), dnodes.data_ptr(),
                                                        grads.data_ptr())
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
        _lib.check(_lib.load().rulgnn_hagcn_graph_backward_f32(C.byref(shp), C.byref(a), _stream()), "rulgnn_hagcn_graph_backward_f32")
        return dnodes, grads

    def graph_stack(self, nodes):
        """(feats [G, 3h], kl) of the graph part for given node features [G, N, enc] (what Model.py:164-183 computes)."""
        if torch.is_grad_enabled() and (nodes.requires_grad or any(p.requires_grad for p in self._graph_params())):
            return _GraphFunction.apply(self, nodes, *self._graph_params())
        return self._graph_forward(nodes)

    # ---- nn.Module surface -----------------------------------------------------------------------------
    def forward(self, X, train=False):
        if not X.is_cuda:
            raise RuntimeError("HAGCN_model's graph stack runs on the HIP kernels only: input must be a CUDA (ROCm) tensor; "
                               "there is no CPU fallback")
        bs, num_node, _ = X.size()
        X = torch.reshape(X.float(), [bs, num_node, self.num_patch, self.patch_size])
        X = torch.transpose(X, 1, 2)
        bs, tlen, num_node, dimension = X.size()
        X = torch.transpose(X, 1, 2)
        X = torch.reshape(X, [bs * num_node, tlen, dimension])
        X = torch.transpose(X, 1, 0)                        # LSTM "batch" = patches, "sequence" = batch*nodes (Model.py:153-157)
        TD_output = self.TD(X)
        X = torch.transpose(TD_output, 1, 0)
        X = torch.reshape(X, [bs, num_node, tlen, -1])
        X = torch.transpose(X, 1, 2)
        nodes = torch.reshape(X, [bs * tlen, num_node, -1])
        feats, total_kl_div = self.graph_stack(nodes)
        out = torch.reshape(feats, [bs, -1])
        output = self.fc(out)
        if train:
            return output, total_kl_div
        return output
