"""Torch-CPU restatements of ``FC_STGNN.update``, ``ASTGCNN.update``, ``HAGCN.update`` and ``STMSGCN.update`` -- TEST / MEASUREMENT
INFRASTRUCTURE, NOT PRODUCT CODE (same rule as the numpy oracles: only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import it).

SURVEY.md section 8(d) asks for the CPU baseline to be the path the reference itself takes on a CPU: ATen kernels (``conv1d``,
``native_batch_norm``, ``bmm``, ``cdist``, ``fft``, the fused ``lstm`` / ``gru`` kernels, ``sort``), autograd and ``torch.optim.Adam`` under
``torch.set_num_threads(n)`` -- not an fp64 numpy port.  Like ``oracle/stgcn_torch_cpu.py`` this file states that path in functional
form: a dict of leaf tensors keyed by the reference's ``state_dict`` names, plain ``torch`` / ``torch.nn.functional`` calls in the
reference's order, no ``nn.Module`` tree (the two recurrent families hold ``nn.LSTM`` / ``nn.GRU`` instances only because ATen's fused
recurrent kernels have no functional entry; their weights are the dict's tensors).

  FC_STGNN  <- models/FC_STGNN/Model.py:46-84, Model_Base.py:12-41 (encoder), :44-67 (graph), :72-107 (MPNN), :111-134 (positions),
               :137-170 (windows, decay mask), :175-225 (block); update: algorithms/algorithms.py:51-76
  ASTGCNN   <- models/ASTGCNN/Model.py:72-146 (TCN), :169-181 (gate), :184-195 (graph), :198-230 (ChebNet), :233-254; update :139-163
  HAGCN     <- models/HAGCN/Model.py:6-24 (GIN), :26-73 (Bi-LSTM stack), :75-120 (SAGPool), :122-127, :129-195; update :222-248
  STMSGCN   <- models/STMSGCN/Model.py:7-31 (SED), :34-49 (GCN), :52-60 (GRU), :63-112; update :546-571

Pinned by ``tests/test_torch_cpu_families.py`` against the fixtures the reference itself generated (``tests/golden/*``: eval forwards,
train-mode losses and the loss curves of its own ``update``).
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch
import torch.nn.functional as Fn

from .stgcn_torch_cpu import cpu_model_name  # noqa: F401  (re-exported for bench.py)


def _strip(arrays: dict) -> dict:
    return {(k[6:] if k.startswith("model.") else k): v for k, v in arrays.items()}


class _State:
    """Leaf tensors of the live parameters, BatchNorm running buffers, Adam over the leaves (weight decay as L2 in the gradient)."""

    def __init__(self, arrays: dict, live: list[str], buffers: list[str], lr: float, weight_decay: float, dtype=torch.float32):
        src = _strip(arrays)
        self.p = {k: torch.as_tensor(np.asarray(src[k])).to(dtype).clone().requires_grad_(True) for k in live}
        self.buf = {k: torch.as_tensor(np.asarray(src[k])).to(dtype).clone() for k in buffers}
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr, weight_decay=weight_decay)

    def step(self, loss: torch.Tensor) -> float:
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        return float(loss.item())


def _bn(st: _State, prefix: str, z: torch.Tensor, train: bool) -> torch.Tensor:
    return Fn.batch_norm(z, st.buf[prefix + ".running_mean"], st.buf[prefix + ".running_var"], st.p[prefix + ".weight"],
                         st.p[prefix + ".bias"], train, 0.1, 1e-5)


# ======================================================================================================================
# ASTGCNN
# ======================================================================================================================
class AstgcnnState(_State):
    def __init__(self, arrays, lr=1e-3, weight_decay=1e-4, dtype=torch.float32):
        live = ["tcn.conv_block1.0.weight", "tcn.conv_block1.2.weight", "tcn.conv_block1.2.bias", "tcn.conv_block2.0.weight",
                "tcn.conv_block2.2.weight", "tcn.conv_block2.2.bias", "gate.bias", "gate.theta.weight", "gate.theta.bias",
                "distance_module.P.weight", "chebnet.filters", "fc.weight", "fc.bias"]
        bufs = [f"tcn.conv_block{b}.2.running_{w}" for b in (1, 2) for w in ("mean", "var")]
        super().__init__(arrays, live, bufs, lr, weight_decay, dtype)


def astgcnn_forward(st: AstgcnnState, x: torch.Tensor, train: bool) -> torch.Tensor:
    p = st.p
    k = p["tcn.conv_block1.0.weight"].shape[-1]
    h = x
    for blk, dil in ((1, 1), (2, 2)):
        pad = (k - 1) * dil
        z = Fn.conv1d(h, p[f"tcn.conv_block{blk}.0.weight"], None, 1, pad, dil)[:, :, :-pad].contiguous()      # causal: chomp the right pad
        h = torch.relu(torch.relu(_bn(st, f"tcn.conv_block{blk}.2", z, train)) + h)
    gated = torch.tanh(Fn.linear(x, p["gate.theta.weight"], p["gate.theta.bias"]) + p["gate.bias"]) * h
    proj = Fn.linear(gated, p["distance_module.P.weight"])
    adj = torch.exp(-torch.cdist(proj, proj, p=2))
    filt = p["chebnet.filters"]
    t0, t1 = gated, torch.bmm(adj, gated)
    out = torch.matmul(t0, filt[0])
    if filt.shape[0] > 1:
        out = out + torch.matmul(t1, filt[1])
    for kk in range(2, filt.shape[0]):
        t2 = 2 * torch.bmm(adj, t1) - t0
        out = out + torch.matmul(t2, filt[kk])
        t0, t1 = t1, t2
    return Fn.linear(out.mean(dim=1), p["fc.weight"], p["fc.bias"])


def astgcnn_update(st: AstgcnnState, x, y) -> float:
    pred = astgcnn_forward(st, x, True)
    return st.step(Fn.mse_loss(pred, y.reshape(pred.shape)))


# ======================================================================================================================
# FC_STGNN
# ======================================================================================================================
def fc_live_names():
    out = ["nonlin_map.conv_block1.0.weight", "nonlin_map.conv_block1.1.weight", "nonlin_map.conv_block1.1.bias",
           "nonlin_map.conv_block2.0.weight", "nonlin_map.conv_block2.1.weight", "nonlin_map.conv_block2.1.bias",
           "nonlin_map2.0.weight", "nonlin_map2.0.bias", "nonlin_map2.1.weight", "nonlin_map2.1.bias"]
    for b in ("MPNN1", "MPNN2"):
        out += [f"{b}.graph_construction.mapping.weight", f"{b}.graph_construction.mapping.bias", f"{b}.BN.weight", f"{b}.BN.bias",
                f"{b}.MPNN.theta.0.weight", f"{b}.MPNN.theta.0.bias", f"{b}.MPNN.bn1.weight", f"{b}.MPNN.bn1.bias"]
    for i in (1, 2, 3, 4):
        out += [f"fc.fc{i}.weight", f"fc.fc{i}.bias"]
    return out


def fc_bn_prefixes():
    return ["nonlin_map.conv_block1.1", "nonlin_map.conv_block2.1", "nonlin_map2.1", "MPNN1.BN", "MPNN1.MPNN.bn1", "MPNN2.BN", "MPNN2.MPNN.bn1"]


class FcstgnnState(_State):
    def __init__(self, arrays, cfg: dict, lr=1e-3, weight_decay=1e-4, dtype=torch.float32):
        bufs = [f"{q}.running_{w}" for q in fc_bn_prefixes() for w in ("mean", "var")]
        super().__init__(arrays, fc_live_names(), bufs, lr, weight_decay, dtype)
        self.cfg = dict(cfg)
        d = 2 * int(cfg["hidden_dim"])
        # positional table with base 100 (Model_Base.py:122), rows 0 .. num_patch - 1
        pos = torch.arange(0, 5000).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2) * -(math.log(100.0) / d))
        pe = torch.zeros(5000, d)
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        self.pe = pe.unsqueeze(0).to(dtype)
        # decay mask 0.7^|dt| over (time, sensor) pairs of a 2-step window (Model_Base.py:150-170)
        S = int(cfg["num_node"])
        t = torch.arange(2 * S) // S
        self.mask = (0.7 ** (t[:, None] - t[None, :]).abs().to(dtype)).to(dtype)


def _fc_block(st: FcstgnnState, name: str, feat: torch.Tensor, stride: int, train: bool) -> torch.Tensor:
    p = st.p
    bs, T, S, D = feat.shape
    w = 2
    cols = Fn.unfold(feat.transpose(1, 3), (S, w), stride=stride)                 # [bs, D*S*w, windows]
    nw = cols.shape[-1]
    win = cols.reshape(bs, D, S, w, nw).transpose(1, -1)                          # [bs, windows, S, w, D]
    nodes = win.transpose(2, 3).reshape(bs * nw, w * S, D)                        # node order: time-major
    m = Fn.linear(nodes, p[f"{name}.graph_construction.mapping.weight"], p[f"{name}.graph_construction.mapping.bias"])
    eye = torch.eye(w * S, dtype=feat.dtype).repeat(bs * nw, 1, 1)
    adj = Fn.softmax(Fn.leaky_relu(torch.bmm(m, m.transpose(1, 2)) - eye * 1e8), dim=-1) + eye
    adj = adj * st.mask
    xn = _bn(st, f"{name}.BN", nodes.transpose(-1, -2), train).transpose(-1, -2)
    h = Fn.linear(torch.bmm(adj, xn), p[f"{name}.MPNN.theta.0.weight"], p[f"{name}.MPNN.theta.0.bias"])
    h = Fn.leaky_relu(_bn(st, f"{name}.MPNN.bn1", h.transpose(-1, -2), train).transpose(-1, -2))
    return h.reshape(bs, nw, w, S, -1).mean(2)


def fcstgnn_forward(st: FcstgnnState, x: torch.Tensor, train: bool, dropout: float = 0.1) -> torch.Tensor:
    p, c = st.p, st.cfg
    bs, S, _ = x.shape
    NP, PS = int(c["num_patch"]), int(c["patch_size"])
    k = int(c["encoder_conv_kernel"])
    seq = x.reshape(bs, S, NP, PS).transpose(1, 2).reshape(bs * NP * S, PS, 1).transpose(-1, -2)      # [M, 1, PS]
    z = torch.relu(_bn(st, "nonlin_map.conv_block1.1", Fn.conv1d(seq, p["nonlin_map.conv_block1.0.weight"], None, 1, k // 2), train))
    z = torch.relu(_bn(st, "nonlin_map.conv_block2.1", Fn.conv1d(z, p["nonlin_map.conv_block2.0.weight"], None, 1, 1), train))
    e = _bn(st, "nonlin_map2.1", Fn.linear(z.reshape(bs * NP * S, -1), p["nonlin_map2.0.weight"], p["nonlin_map2.0.bias"]), train)
    e = e.reshape(bs, NP, S, -1).transpose(1, 2).reshape(bs * S, NP, -1)
    e = Fn.dropout(e + st.pe[:, :NP], dropout, train)
    feat = e.reshape(bs, S, NP, -1).transpose(1, 2)                                # [bs, NP, S, 2 hidden]
    f1 = _fc_block(st, "MPNN1", feat, 1, train).reshape(bs, -1)
    f2 = _fc_block(st, "MPNN2", feat, 2, train).reshape(bs, -1)
    h = torch.cat([f1, f2], -1)
    for i in (1, 2, 3):
        h = torch.relu(Fn.linear(h, p[f"fc.fc{i}.weight"], p[f"fc.fc{i}.bias"]))
    return Fn.linear(h, p["fc.fc4.weight"], p["fc.fc4.bias"])


def fcstgnn_update(st: FcstgnnState, x, y, dropout: float = 0.1) -> float:
    pred = fcstgnn_forward(st, x, True, dropout)
    return st.step(Fn.mse_loss(pred, y.reshape(pred.shape)))


# ======================================================================================================================
# STMSGCN
# ======================================================================================================================
class StmsgcnState(_State):
    def __init__(self, arrays, cfg: dict, lr=1e-3, weight_decay=1e-4, dtype=torch.float32):
        self.cfg = dict(cfg)
        dims = [1] + [int(d) for d in cfg["gcn_dims"]]
        self.dims = dims
        live = [f"gcn_layers.{i}.linear.{w}" for i in range(len(dims) - 1) for w in ("weight", "bias")]
        gru = [f"gru_layer.gru.{n}_l0" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        live += gru + ["fc.weight", "fc.bias"]
        super().__init__(arrays, live, [], lr, weight_decay, dtype)
        # ATen's fused GRU through an nn.GRU whose flat weights ARE the leaves above
        self.gru = torch.nn.GRU(sum(dims), int(cfg["gru_hidden_dim"]), 1, batch_first=True).to(dtype)
        for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"):
            delattr(self.gru, n + "_l0")
            setattr(self.gru, n + "_l0", self.p[f"gru_layer.gru.{n}_l0"])
        self.gru._flat_weights = [self.p[g] for g in gru]


def stmsgcn_forward(st: StmsgcnState, x: torch.Tensor) -> torch.Tensor:
    p, c = st.p, st.cfg
    bs = x.shape[0]
    NP, PS, iv, bw = int(c["num_patch"]), int(c["patch_size"]), int(c["interval"]), int(c["band_width"])
    spec = torch.fft.fft(x.reshape(bs * NP, PS), dim=-1)
    sd = spec[:, iv:] - spec[:, :-iv]
    h = (sd.real ** 2 + sd.imag ** 2).view(bs * NP, -1, bw).sum(dim=-1).reshape(bs * NP, -1, 1)
    N = h.shape[1]
    outs = [h]
    for i in range(len(st.dims) - 1):
        a = torch.bmm(h, h.transpose(-1, -2)) + torch.eye(N, dtype=h.dtype)
        d = torch.diag_embed(a.sum(dim=-1) ** -0.5)
        h = Fn.leaky_relu(Fn.linear(torch.bmm(torch.bmm(d, torch.bmm(a, d)), h), p[f"gcn_layers.{i}.linear.weight"], p[f"gcn_layers.{i}.linear.bias"]))
        outs.append(h)
    cat = torch.cat(outs, dim=-1).reshape(bs, NP, N, -1).transpose(1, 2).reshape(bs * N, NP, -1)
    g, _ = st.gru(cat)
    g = g.reshape(bs, N, NP, -1).mean(1)
    return Fn.linear(g.reshape(bs, -1), p["fc.weight"], p["fc.bias"])


def stmsgcn_update(st: StmsgcnState, x, y) -> float:
    pred = stmsgcn_forward(st, x)
    return st.step(Fn.mse_loss(pred, y.reshape(pred.shape)))


# ======================================================================================================================
# HAGCN
# ======================================================================================================================
_LSTM_NAMES = [f"{n}_l0{r}" for r in ("", "_reverse") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]


def hagcn_live_names():
    out = [f"TD.bi_lstm{i}.{n}" for i in (1, 2, 3) for n in _LSTM_NAMES]
    for i in (1, 2, 3):
        out += [f"gin{i}.eps", f"gin{i}.mlp.0.weight", f"gin{i}.mlp.0.bias", f"gin{i}.mlp.2.weight", f"gin{i}.mlp.2.bias"]
        out += [f"gnn{i}.rank.weight", f"gnn{i}.rank.bias", f"gnn{i}.model.weight", f"gnn{i}.model.bias", f"gnn{i}.mlp.0.weight",
                f"gnn{i}.mlp.0.bias", f"gnn{i}.mlp.2.weight", f"gnn{i}.mlp.2.bias"]
    return out + ["fc.0.weight", "fc.0.bias", "fc.2.weight", "fc.2.bias"]


class HagcnState(_State):
    def __init__(self, arrays, cfg: dict, lr=1e-3, weight_decay=1e-4, alpha=100.0, dtype=torch.float32):
        self.cfg, self.alpha = dict(cfg), float(alpha)
        super().__init__(arrays, hagcn_live_names(), [], lr, weight_decay, dtype)
        enc, ps = int(cfg["encoder_hidden_dim"]), int(cfg["patch_size"])
        self.lstm = []
        for i, (inp, hid) in enumerate(((ps, enc), (enc, 2 * enc), (2 * enc, enc)), start=1):
            m = torch.nn.LSTM(input_size=inp, hidden_size=hid, num_layers=1, batch_first=True, bidirectional=True).to(dtype)
            for n in _LSTM_NAMES:
                delattr(m, n)
                setattr(m, n, self.p[f"TD.bi_lstm{i}.{n}"])
            m._flat_weights = [self.p[f"TD.bi_lstm{i}.{n}"] for n in m._flat_weights_names]
            self.lstm.append(m)


def _halves_summed(v):
    a, b = torch.split(v, v.shape[2] // 2, 2)
    return a + b


def _sagpool(st, i, n_keep, X, A):
    p = st.p
    xo = Fn.leaky_relu(Fn.linear(torch.bmm(A, X), p[f"gnn{i}.model.weight"], p[f"gnn{i}.model.bias"]))
    prior = Fn.linear(torch.relu(Fn.linear(X, p[f"gnn{i}.mlp.0.weight"], p[f"gnn{i}.mlp.0.bias"])), p[f"gnn{i}.mlp.2.weight"], p[f"gnn{i}.mlp.2.bias"])
    prior = torch.softmax(prior, dim=1).squeeze()
    score = torch.softmax(Fn.linear(torch.bmm(A, X), p[f"gnn{i}.rank.weight"], p[f"gnn{i}.rank.bias"]), 1).squeeze()
    kl = Fn.kl_div(prior.log(), score, reduction='batchmean')
    _, idx = torch.sort(score, descending=True, dim=1)
    top = idx[:, :n_keep]
    row = torch.arange(X.size(0)).unsqueeze(1)
    a_out = torch.transpose(A[row, top], 1, 2)[row, top]
    return xo[row, top], a_out, kl


def hagcn_nodes(st: HagcnState, x: torch.Tensor, train: bool, dropout: float = 0.2) -> torch.Tensor:
    """The Bi-LSTM stack: [bs, sensors, L] -> node features [bs * num_patch, sensors, encoder_hidden_dim]."""
    c = st.cfg
    bs, S, _ = x.shape
    NP, PS = int(c["num_patch"]), int(c["patch_size"])
    seq = x.reshape(bs, S, NP, PS).reshape(bs * S, NP, PS).transpose(1, 0)         # [num_patch, bs * S, PS]: the LSTM recurs along bs * S
    h, _ = st.lstm[0](seq)
    h, _ = st.lstm[1](_halves_summed(h))
    h = Fn.dropout(_halves_summed(h), dropout, train)
    h, _ = st.lstm[2](h)
    h = Fn.leaky_relu(Fn.dropout(_halves_summed(h), dropout, train))
    return h.transpose(1, 0).reshape(bs, S, NP, -1).transpose(1, 2).reshape(bs * NP, S, -1)


def hagcn_from_nodes(st: HagcnState, nodes: torch.Tensor, bs: int, train: bool):
    """Cosine adjacency, three GIN + SAGPool levels, the head."""
    p = st.p
    gram = torch.matmul(nodes, nodes.transpose(-1, -2))
    nrm = torch.sqrt(torch.sum(nodes ** 2, -1)).unsqueeze(-1)
    A = gram / torch.matmul(nrm, nrm.transpose(-1, -2))
    X, feats, kl_total = nodes, [], 0.0
    for i, n_keep in ((1, 10), (2, 5), (3, 1)):
        g = torch.bmm(A, X) + (1 + p[f"gin{i}.eps"]) * X
        g = Fn.linear(torch.relu(Fn.linear(g, p[f"gin{i}.mlp.0.weight"], p[f"gin{i}.mlp.0.bias"])), p[f"gin{i}.mlp.2.weight"], p[f"gin{i}.mlp.2.bias"])
        X, A, kl = _sagpool(st, i, n_keep, g, A)
        feats.append(X.mean(1))
        kl_total = kl_total + kl
    out = torch.cat(feats, dim=-1).squeeze().reshape(bs, -1)
    out = Fn.linear(torch.relu(Fn.linear(out, p["fc.0.weight"], p["fc.0.bias"])), p["fc.2.weight"], p["fc.2.bias"])
    return (out, kl_total) if train else out


def hagcn_forward(st: HagcnState, x: torch.Tensor, train: bool, dropout: float = 0.2):
    return hagcn_from_nodes(st, hagcn_nodes(st, x, train, dropout), x.shape[0], train)


def hagcn_update(st: HagcnState, x, y, dropout: float = 0.2) -> float:
    pred, kl = hagcn_forward(st, x, True, dropout)
    return st.step(Fn.mse_loss(pred, y.reshape(pred.shape)) + st.alpha * kl)


def hagcn_random_arrays(cfg: dict, seed: int = 0) -> dict:
    """Random parameters with torch's own initialisers (uniform(-1/sqrt(hidden), 1/sqrt(hidden)) for the LSTMs, Kaiming-uniform linears)."""
    g = torch.Generator().manual_seed(seed)
    enc, hid, ps, out, NP = (int(cfg[k]) for k in ("encoder_hidden_dim", "hidden_dim", "patch_size", "output_dim", "num_patch"))
    a = {}

    def uni(shape, bound):
        return ((torch.rand(shape, generator=g) * 2 - 1) * bound).numpy()

    for i, (inp, h) in enumerate(((ps, enc), (enc, 2 * enc), (2 * enc, enc)), start=1):
        b = 1.0 / math.sqrt(h)
        for r in ("", "_reverse"):
            a[f"TD.bi_lstm{i}.weight_ih_l0{r}"] = uni((4 * h, inp), b)
            a[f"TD.bi_lstm{i}.weight_hh_l0{r}"] = uni((4 * h, h), b)
            a[f"TD.bi_lstm{i}.bias_ih_l0{r}"] = uni((4 * h,), b)
            a[f"TD.bi_lstm{i}.bias_hh_l0{r}"] = uni((4 * h,), b)

    def lin(name, o, i_):
        b = 1.0 / math.sqrt(i_)
        a[name + ".weight"], a[name + ".bias"] = uni((o, i_), b), uni((o,), b)

    for i in (1, 2, 3):
        a[f"gin{i}.eps"] = np.zeros(1, np.float32)
        lin(f"gin{i}.mlp.0", hid, enc if i == 1 else hid)
        lin(f"gin{i}.mlp.2", hid, hid)
        lin(f"gnn{i}.rank", 1, hid)
        lin(f"gnn{i}.model", hid, hid)
        lin(f"gnn{i}.mlp.0", hid // 2, hid)
        lin(f"gnn{i}.mlp.2", 1, hid // 2)
    lin("fc.0", out, hid * 3 * NP)
    lin("fc.2", 1, out)
    return a


# ======================================================================================================================
# timing (bench.py: families.*.cpu_baseline)
# ======================================================================================================================
def make_case(family: str, cfg: dict, batch: int, seed: int = 0):
    """(state, x, y, step) of the family's SURVEY section 8(d) workload with random parameters: ``step()`` = one full ``update``."""
    g = torch.Generator().manual_seed(seed)
    if family == "ASTGCNN":
        from oracle import astgcnn_oracle as O
        st = AstgcnnState(O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["output_dim"], cfg["K"], seed=seed))
        x = torch.rand(batch, cfg["num_nodes"], cfg["time_length"], generator=g) * 2 - 1
        step = lambda: astgcnn_update(st, x, y)
    elif family == "FC_STGNN":
        from oracle import fcstgnn_oracle as O
        st = FcstgnnState(O.random_params(O.Config(**cfg), seed=seed), cfg)
        x = torch.rand(batch, cfg["num_node"], cfg["num_patch"] * cfg["patch_size"], generator=g)
        step = lambda: fcstgnn_update(st, x, y)
    elif family == "STMSGCN":
        from oracle import stmsgcn_oracle as O
        c = O.Config(cfg["num_patch"], cfg["patch_size"], cfg["interval"], cfg["band_width"], cfg["gcn_dims"], cfg["gru_hidden_dim"])
        st = StmsgcnState(O.random_params(c, seed=seed), cfg)
        x = torch.rand(batch, 1, cfg["num_patch"] * cfg["patch_size"], generator=g)
        step = lambda: stmsgcn_update(st, x, y)
    elif family == "HAGCN":
        st = HagcnState(hagcn_random_arrays(cfg, seed), cfg)
        x = torch.rand(batch, 14, cfg["num_patch"] * cfg["patch_size"], generator=g)
        step = lambda: hagcn_update(st, x, y)
    else:
        raise ValueError(family)
    y = torch.rand(batch, 1, generator=g)
    return st, x, y, step


def time_update(family: str, cfg: dict, batch: int, threads: int, warmup: int = 5, iters: int = 100, budget_s: float = 8.0) -> dict:
    """``warmup`` untimed + up to ``iters`` timed full updates at ``threads`` ATen threads, cut short when ``budget_s`` runs out (the count
    actually timed is returned)."""
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        _, _, _, step = make_case(family, cfg, batch)
        t0 = time.perf_counter()
        step()
        first = time.perf_counter() - t0
        if first > budget_s / 2:
            return {"threads": threads, "batch": batch, "iterations": 1, "seconds": round(first, 3), "samples_per_s": round(batch / first, 2),
                    "ms_per_iteration": round(first * 1e3, 3), "note": "first iteration exceeded half the time budget: not repeated"}
        for _ in range(warmup - 1):
            step()
            if time.perf_counter() - t0 > budget_s / 3:
                break
        n, t0 = 0, time.perf_counter()
        while n < iters:
            step()
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        el = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    return {"threads": threads, "batch": batch, "iterations": n, "seconds": round(el, 3), "samples_per_s": round(n * batch / el, 2),
            "ms_per_iteration": round(el / n * 1e3, 3)}
