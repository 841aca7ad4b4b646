"""Torch-CPU restatement of ``ST_GCN.update`` -- TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this module (same rule as the numpy oracles).

Why it exists next to ``stgcn_oracle.py``: SURVEY.md section 8(d) asks for the CPU baseline to be the path the
reference itself takes on a CPU -- ATen kernels (``bmm``, ``mkldnn_convolution``, ``native_batch_norm``, autograd,
``torch.optim.Adam``) under ``torch.set_num_threads(n)`` -- not a numpy port.  This file states that path in
functional form (a dict of tensors keyed by the reference's ``state_dict`` names, no ``nn.Module``), so the
same ATen kernels run in the same order as in the reference:

  patch statistics   <- models/ST_GCN/Model.py:7-52
  Pearson adjacency  <- models/ST_GCN/Model.py:53-71
  MPNN (k = 1)       <- models/ST_GCN/Model.py:74-90
  causal TCN block   <- models/ST_GCN/Model.py:92-173  (conv -> chomp -> BatchNorm -> ReLU, twice, with residuals)
  layer loop, head   <- models/ST_GCN/Model.py:176-222
  update             <- algorithms/algorithms.py:481-490 (MSE, backward, Adam with L2-in-gradient weight decay)

Pinned by ``tests/test_torch_cpu_baseline.py`` against ``tests/golden/stgcn_train_curve_14x30_bs32.npz`` (24 steps of the
reference's own ``update``) and against the fp64 numpy oracle.
"""
from __future__ import annotations

import torch
import torch.nn.functional as Fn

NUM_STATS = 10
LEAKY = 0.01


def live_names(num_layers: int) -> list[str]:
    out = []
    for l in range(num_layers):
        p = f"sg_tcn.layers.{l}"
        out += [f"{p}.0.theta.0.weight", f"{p}.0.theta.0.bias"]
        for blk in (1, 2):
            out += [f"{p}.1.conv_block{blk}.0.weight", f"{p}.1.conv_block{blk}.2.weight", f"{p}.1.conv_block{blk}.2.bias"]
    return out + ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]


def stat_names(num_layers: int) -> list[str]:
    return [f"sg_tcn.layers.{l}.1.conv_block{blk}.2.running_{w}" for l in range(num_layers) for blk in (1, 2) for w in ("mean", "var")]


class State:
    """Live parameters (requires_grad leaves), BatchNorm running buffers and the Adam optimizer over the live tensors."""

    def __init__(self, arrays: dict, num_layers: int = 2, lr: float = 1e-3, weight_decay: float = 1e-4, dtype=torch.float32):
        strip = lambda k: k[6:] if k.startswith("model.") else k
        src = {strip(k): v for k, v in arrays.items()}
        self.num_layers = num_layers
        self.p = {k: torch.as_tensor(src[k]).to(dtype).clone().requires_grad_(True) for k in live_names(num_layers)}
        self.buf = {k: torch.as_tensor(src[k]).to(dtype).clone() for k in stat_names(num_layers)}
        self.opt = torch.optim.Adam(list(self.p.values()), lr=lr, weight_decay=weight_decay)


def patch_statistics(x: torch.Tensor) -> torch.Tensor:
    """[B, N, P] windows -> [B, 10, N] statistic rows."""
    P = x.shape[-1]
    mx, mn = x.amax(-1), x.amin(-1)
    mean = x.mean(-1)
    d = x - mean.unsqueeze(-1)
    var = (d * d).sum(-1) / (P - 1)
    sd = var.sqrt()
    zn = d / sd.unsqueeze(-1)
    feats = [mx, mn, mx - mn, var, sd, mean, (x * x).mean(-1).sqrt(), x.abs().mean(-1), (zn ** 3).mean(-1), (zn ** 4).mean(-1) - 3.0]
    return torch.stack(feats, dim=1)


def pearson(feat: torch.Tensor) -> torch.Tensor:
    c = feat - feat.mean(-1, keepdim=True)
    n = c.norm(dim=-1, keepdim=True)
    return torch.bmm(c, c.transpose(1, 2)) / torch.bmm(n, n.transpose(1, 2))


def _tcn_block(h, w, gamma, beta, rmean, rvar, dil, train):
    z = Fn.conv1d(h, w, None, 1, dil, dil)[..., :h.shape[-1]].contiguous()
    return torch.relu(Fn.batch_norm(z, rmean, rvar, gamma, beta, train, 0.1, 1e-5))


def forward(st: State, x: torch.Tensor, num_patch: int, patch_size: int, train: bool, dropout: float = 0.0, keep_masks=None) -> torch.Tensor:
    """``keep_masks`` (train mode, one bool / float tensor [B, 10, num_patch] per layer): the dropout masks IMPOSED instead of drawn from
    torch's generator -- nn.Dropout's arithmetic (kept elements x 1 / (1 - p)) on the masks of the HIP path's counter hash
    (oracle/stgcn_oracle.py: dropout_keep_mask), so that the two paths train the same function with dropout ON (torch's CPU Bernoulli
    stream cannot be reproduced by any GPU path)."""
    B = x.shape[0]
    feat = patch_statistics(x.reshape(B, num_patch, patch_size))
    adj = pearson(feat)
    X = feat
    for l in range(st.num_layers):
        p = f"sg_tcn.layers.{l}"
        H = Fn.leaky_relu(Fn.linear(torch.bmm(adj, X), st.p[f"{p}.0.theta.0.weight"], st.p[f"{p}.0.theta.0.bias"]), LEAKY)
        q = f"{p}.1.conv_block1"
        o0 = torch.relu(_tcn_block(H, st.p[f"{q}.0.weight"], st.p[f"{q}.2.weight"], st.p[f"{q}.2.bias"],
                                   st.buf[f"{q}.2.running_mean"], st.buf[f"{q}.2.running_var"], 1, train) + H)
        q = f"{p}.1.conv_block2"
        o1 = torch.relu(_tcn_block(o0, st.p[f"{q}.0.weight"], st.p[f"{q}.2.weight"], st.p[f"{q}.2.bias"],
                                   st.buf[f"{q}.2.running_mean"], st.buf[f"{q}.2.running_var"], 2, train) + o0)
        if keep_masks is not None and train and dropout > 0.0:
            X = o1 * (keep_masks[l].to(o1.dtype) * (1.0 / (1.0 - dropout))) + X
        else:
            X = Fn.dropout(o1, dropout, train) + X
    pooled = X.amax(dim=1) if not train else X.max(dim=1).values       # max over the ten statistic channels
    y1 = torch.relu(Fn.linear(pooled, st.p["fc1.weight"], st.p["fc1.bias"]))
    return Fn.linear(y1, st.p["fc2.weight"], st.p["fc2.bias"])


def update(st: State, x: torch.Tensor, y: torch.Tensor, num_patch: int, patch_size: int, dropout: float = 0.0, keep_masks=None) -> float:
    """One training step; returns the loss like the reference's ``{'loss': loss.item()}``."""
    pred = forward(st, x, num_patch, patch_size, True, dropout, keep_masks)
    loss = Fn.mse_loss(pred, y.reshape(pred.shape))
    st.opt.zero_grad()
    loss.backward()
    st.opt.step()
    return float(loss.item())


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_update(num_patch: int, patch_size: int, batch: int, threads: int, dropout: float, warmup: int = 20, iters: int = 100,
                budget_s: float = 12.0, eval_forward: bool = False) -> dict:
    """Times ``update`` (or the eval forward) at ``threads`` ATen threads: ``warmup`` untimed + ``iters`` timed iterations,
    cut short only if ``budget_s`` runs out first (the count actually timed is returned)."""
    import time
    import numpy as np
    from oracle import stgcn_oracle as O
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        g = torch.Generator().manual_seed(0)
        st = State(O.random_params(num_patch, 2, seed=0, dtype=np.float32))
        x = torch.rand(batch, num_patch, patch_size, generator=g)
        y = torch.rand(batch, 1, generator=g)

        def one():
            if eval_forward:
                with torch.no_grad():
                    forward(st, x, num_patch, patch_size, False)
            else:
                update(st, x, y, num_patch, patch_size, dropout)
        t0 = time.perf_counter()
        one()
        first = time.perf_counter() - t0
        if first > budget_s / 3:
            # oversubscribed (e.g. 256 ATen threads on a 4096-sample batch): one iteration already exhausts the budget; report
            # it as the single measurement instead of spending another budget on a timed loop
            return {"threads": threads, "batch": batch, "iterations": 1, "seconds": round(first, 3),
                    "samples_per_s": round(batch / first, 1), "ms_per_iteration": round(first * 1e3, 4),
                    "note": "first iteration exceeded the time budget: not repeated"}
        for _ in range(warmup - 1):
            one()
            if time.perf_counter() - t0 > budget_s / 3:
                break
        n, t0 = 0, time.perf_counter()
        while n < iters:
            one()
            n += 1
            if time.perf_counter() - t0 > budget_s:
                break
        el = time.perf_counter() - t0
    finally:
        torch.set_num_threads(prev)
    return {"threads": threads, "batch": batch, "iterations": n, "seconds": round(el, 3), "samples_per_s": round(n * batch / el, 1),
            "ms_per_iteration": round(el / n * 1e3, 4)}
