"""CPU oracle for the ST_GCN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The shipped path (``gnn_rul_benchmarking_amd``) never imports it and
fails loudly when the HIP library is missing.

This is a plain-numpy restatement (no torch) of the reference algorithm, written from the
math of the reference files cited below.  It is generic in dtype: float64 is the checker,
float32 is what ``bench.py`` times as the CPU "port" baseline.

Parity pin: the reference repo holds no tests or golden vectors of its own (SURVEY.md
section 4), so this oracle is pinned against outputs of the reference itself, run in the build
container by ``tests/golden/make_golden.py`` and committed as ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every one of them.

Reference map (paths relative to the reference repo root):
  patch_statistics      <- models/ST_GCN/Model.py:7-52   (segment_and_compute_features, skew, kurtosis)
  pearson_adjacency     <- models/ST_GCN/Model.py:53-71  (pcc_graph_construction)
  _mpnn / layer fwd     <- models/ST_GCN/Model.py:74-90  (MPNN_mk, any order k), :99-173 (TemporalConvNet,
                           only conv_block1/conv_block2 are live), :176-195 (SG_TCN)
  forward               <- models/ST_GCN/Model.py:197-222 (ST_GCN_model.forward)
  mse_train_step        <- algorithms/algorithms.py:481-490 (ST_GCN.update: MSE, backward, Adam)
  adam_update           <- torch.optim.Adam as configured at algorithms/algorithms.py:474-478
  rmse / mae / scores   <- utils.py:136-201
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

NUM_STATS = 10          # statistical features per patch == graph nodes == TCN channels
TCN_KERNEL = 2          # models/ST_GCN/Model.py:183  kernel_size=2
LEAKY_SLOPE = 0.01      # F.leaky_relu default, Model.py:90
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


# ----------------------------------------------------------------------------------------
# parameter bookkeeping
# ----------------------------------------------------------------------------------------
def mpnn_order(prm: dict) -> int:
    """MPNN order k of a parameter dictionary: the number of theta.<kk> Linear layers of layer 0 (Model.py:77)."""
    k = 0
    while f"sg_tcn.layers.0.0.theta.{k}.weight" in prm:
        k += 1
    return k


def live_param_names(num_layers: int, k: int = 1) -> list[str]:
    """Names (reference state_dict keys without the ``model.`` prefix) of the parameters that
    receive gradients, in the flat-buffer order used by the HIP path.  ``net0``/``net1`` are
    constructed by the reference but never called (Model.py:110-131), so they are not live."""
    names = []
    for l in range(num_layers):
        p = f"sg_tcn.layers.{l}"
        for kk in range(k):
            names += [f"{p}.0.theta.{kk}.weight", f"{p}.0.theta.{kk}.bias"]
        names += [f"{p}.1.conv_block1.0.weight", f"{p}.1.conv_block1.2.weight", f"{p}.1.conv_block1.2.bias",
                  f"{p}.1.conv_block2.0.weight", f"{p}.1.conv_block2.2.weight", f"{p}.1.conv_block2.2.bias"]
    names += ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    return names


def bn_buffer_names(num_layers: int) -> list[tuple[str, str]]:
    out = []
    for l in range(num_layers):
        for blk in (1, 2):
            p = f"sg_tcn.layers.{l}.1.conv_block{blk}.2"
            out.append((f"{p}.running_mean", f"{p}.running_var"))
    return out


def random_params(num_patch: int, num_layers: int = 2, seed: int = 0, dtype=np.float32, k: int = 1) -> dict:
    """Random (not reference-initialised) live parameters + BN buffers, for tests/bench."""
    rng = np.random.default_rng(seed)
    N, F = num_patch, NUM_STATS
    s = 1.0 / math.sqrt(N)
    prm = {}
    for l in range(num_layers):
        p = f"sg_tcn.layers.{l}"
        for kk in range(k):
            prm[f"{p}.0.theta.{kk}.weight"] = rng.uniform(-s, s, (N, N))
            prm[f"{p}.0.theta.{kk}.bias"] = rng.uniform(-s, s, (N,))
        for blk in (1, 2):
            q = f"{p}.1.conv_block{blk}"
            prm[f"{q}.0.weight"] = rng.uniform(-0.22, 0.22, (F, F, TCN_KERNEL))
            prm[f"{q}.2.weight"] = rng.uniform(0.5, 1.5, (F,))
            prm[f"{q}.2.bias"] = rng.uniform(-0.3, 0.3, (F,))
            prm[f"{q}.2.running_mean"] = rng.uniform(-0.2, 0.2, (F,))
            prm[f"{q}.2.running_var"] = rng.uniform(0.5, 1.5, (F,))
    prm["fc1.weight"] = rng.uniform(-s, s, (N, N))
    prm["fc1.bias"] = rng.uniform(-s, s, (N,))
    prm["fc2.weight"] = rng.uniform(-s, s, (1, N))
    prm["fc2.bias"] = rng.uniform(-s, s, (1,))
    return {k: np.asarray(v, dtype=dtype) for k, v in prm.items()}


# ----------------------------------------------------------------------------------------
# dropout: counter-based hash RNG shared bit-for-bit with the HIP kernels
# ----------------------------------------------------------------------------------------
def _splitmix64(z: int) -> int:
    z = (z + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def dropout_layer_key(seed: int, step: int, layer: int) -> int:
    """32-bit per-(seed, step, layer) key; computed on the host by the product path too."""
    z = _splitmix64((seed & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(step * 64 + layer + 1))
    return int(z & 0xFFFFFFFF)


def dropout_threshold(p: float) -> int:
    """Element is DROPPED when hash < threshold; threshold = round(p * 2^32) clipped."""
    return int(min(max(round(p * 4294967296.0), 0), 4294967295))


def _lowbias32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h * np.uint32(0x7FEB352D)).astype(np.uint32)
    h ^= h >> np.uint32(15)
    h = (h * np.uint32(0x846CA68B)).astype(np.uint32)
    h ^= h >> np.uint32(16)
    return h


def dropout_keep_mask(batch: int, num_patch: int, key: int, p: float, sample_offset: int = 0) -> np.ndarray:
    """keep[b, c, t] for one layer.  counter = ((b + sample_offset) * 10 + c) * N + t."""
    if p <= 0.0:
        return np.ones((batch, NUM_STATS, num_patch), dtype=bool)
    b = np.arange(batch, dtype=np.uint64)[:, None, None] + np.uint64(sample_offset)
    c = np.arange(NUM_STATS, dtype=np.uint64)[None, :, None]
    t = np.arange(num_patch, dtype=np.uint64)[None, None, :]
    ctr = ((b * np.uint64(NUM_STATS) + c) * np.uint64(num_patch) + t).astype(np.uint32)
    with np.errstate(over="ignore"):
        h = _lowbias32(ctr ^ np.uint32(key))
    return h >= np.uint32(dropout_threshold(p))


# ----------------------------------------------------------------------------------------
# forward pieces
# ----------------------------------------------------------------------------------------
def patch_statistics(patches: np.ndarray) -> np.ndarray:
    """[M, P] -> [M, 10]: max, min, ptp, var(unbiased), std(unbiased), mean, rms, mean|x|,
    skew, excess kurtosis -- Model.py:7-52.  A constant patch gives 0/0 = NaN skew/kurtosis,
    exactly as the reference does (SURVEY.md section 4 hazard 1)."""
    x = patches
    P = x.shape[1]
    mx = x.max(axis=1)
    mn = x.min(axis=1)
    mean = x.mean(axis=1)
    d = x - mean[:, None]
    var = (d * d).sum(axis=1) / (P - 1)
    std = np.sqrt(var)
    rms = np.sqrt((x * x).mean(axis=1))
    mabs = np.abs(x).mean(axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        zn = d / std[:, None]
        skew = (zn ** 3).mean(axis=1)
        kurt = (zn ** 4).mean(axis=1) - 3.0
    return np.stack([mx, mn, mx - mn, var, std, mean, rms, mabs, skew, kurt], axis=-1)


def pearson_adjacency(feat: np.ndarray) -> np.ndarray:
    """[B, 10, N] -> [B, 10, 10] Pearson correlation between the statistic rows -- Model.py:53-71."""
    c = feat - feat.mean(axis=-1, keepdims=True)
    dot = c @ c.transpose(0, 2, 1)
    nrm = np.sqrt((c * c).sum(axis=-1, keepdims=True))
    with np.errstate(divide="ignore", invalid="ignore"):
        return dot / (nrm @ nrm.transpose(0, 2, 1))


def causal_conv(h: np.ndarray, w: np.ndarray, dil: int) -> np.ndarray:
    """Conv1d(k=2, dilation=dil, padding=dil) followed by Chomp1d(dil) -- Model.py:92-98,134-160:
    y[:, co, t] = sum_ci w[co, ci, 0] * h[:, ci, t - dil] + w[co, ci, 1] * h[:, ci, t]."""
    hs = np.zeros_like(h)
    hs[:, :, dil:] = h[:, :, :-dil] if dil < h.shape[2] else 0
    return np.einsum("oi,bit->bot", w[:, :, 0], hs) + np.einsum("oi,bit->bot", w[:, :, 1], h)


def _leaky(x):
    return np.where(x > 0, x, x * LEAKY_SLOPE)


def _relu(x):
    # torch.relu propagates NaN; np.maximum does too
    return np.maximum(x, 0)


@dataclass
class LayerCache:
    X: np.ndarray = None
    AX: np.ndarray = None          # A X (order 1); AXk / Apow: every order of MPNN_mk
    AXk: list = None
    Apow: list = None
    Hpre: np.ndarray = None
    H: np.ndarray = None
    z1: np.ndarray = None
    xhat1: np.ndarray = None
    istd1: np.ndarray = None
    x0: np.ndarray = None
    o0: np.ndarray = None
    z2: np.ndarray = None
    xhat2: np.ndarray = None
    istd2: np.ndarray = None
    x1: np.ndarray = None
    o1: np.ndarray = None
    keep: np.ndarray = None
    bn_mean: list = field(default_factory=list)   # batch mean  (train)
    bn_var: list = field(default_factory=list)    # biased batch var (train)


@dataclass
class FwdCache:
    feat: np.ndarray = None
    adj: np.ndarray = None
    layers: list = field(default_factory=list)
    X_out: np.ndarray = None
    pooled: np.ndarray = None
    argmax: np.ndarray = None
    y1: np.ndarray = None
    pred: np.ndarray = None


def _bn(z, prm, prefix, train, cache_mean, cache_var, stat_reduce=None, stat_count=None):
    g = prm[f"{prefix}.weight"][None, :, None]
    b = prm[f"{prefix}.bias"][None, :, None]
    if train and stat_reduce is not None:
        # synchronised BatchNorm: [sum z | sum z^2] summed over the data-parallel ranks by the caller's
        # ``stat_reduce``; ``stat_count`` = values per channel in the GLOBAL batch
        s = stat_reduce(np.concatenate([z.sum(axis=(0, 2)), (z * z).sum(axis=(0, 2))]))
        mean = s[:z.shape[1]] / stat_count
        var = np.maximum(s[z.shape[1]:] / stat_count - mean * mean, 0)
        cache_mean.append(mean)
        cache_var.append(var)
    elif train:
        mean = z.mean(axis=(0, 2))
        var = z.var(axis=(0, 2))            # biased, used for normalisation
        cache_mean.append(mean)
        cache_var.append(var)
    else:
        mean = prm[f"{prefix}.running_mean"]
        var = prm[f"{prefix}.running_var"]
    istd = 1.0 / np.sqrt(var + z.dtype.type(BN_EPS))
    xhat = (z - mean[None, :, None]) * istd[None, :, None]
    return xhat * g + b, xhat, istd


def forward(prm: dict, x: np.ndarray, num_patch: int, patch_size: int, num_layers: int = 2,
            train: bool = False, dropout: float = 0.0, dropout_keys=None,
            sample_offset: int = 0, stat_reduce=None, stat_count=None) -> FwdCache:
    """Full ST_GCN forward -- Model.py:208-222.  ``x`` is any array with ``x.size ==
    B * num_patch * patch_size`` per the reference's reshape.  ``train`` selects batch-stat
    BatchNorm and (if ``dropout`` > 0) the hash-RNG dropout mask with one key per layer."""
    dt = x.dtype
    prm = {k: np.asarray(v, dtype=dt) for k, v in prm.items()}
    B = x.shape[0]
    N, P, F = num_patch, patch_size, NUM_STATS
    fc = FwdCache()
    feat = patch_statistics(x.reshape(B * N, P)).reshape(B, N, F).transpose(0, 2, 1)   # [B,10,N]
    fc.feat = feat
    fc.adj = pearson_adjacency(feat)
    K = mpnn_order(prm)
    X = feat
    for l in range(num_layers):
        p = f"sg_tcn.layers.{l}"
        lc = LayerCache()
        lc.X = X
        # MPNN_mk, Model.py:81-90: sum over kk of theta_kk(A^(kk+1) X); A_ = bmm(A_, A) is the running power
        lc.Apow, lc.AXk, lc.Hpre = [], [], 0
        A_ = fc.adj
        for kk in range(K):
            if kk > 0:
                A_ = A_ @ fc.adj
            lc.Apow.append(A_)
            lc.AXk.append(A_ @ X)
            lc.Hpre = lc.Hpre + lc.AXk[kk] @ prm[f"{p}.0.theta.{kk}.weight"].T + prm[f"{p}.0.theta.{kk}.bias"]
        lc.AX = lc.AXk[0]
        lc.H = _leaky(lc.Hpre)
        lc.z1 = causal_conv(lc.H, prm[f"{p}.1.conv_block1.0.weight"], 1)
        y, lc.xhat1, lc.istd1 = _bn(lc.z1, prm, f"{p}.1.conv_block1.2", train, lc.bn_mean, lc.bn_var, stat_reduce, stat_count)
        lc.x0 = _relu(y)
        lc.o0 = _relu(lc.x0 + lc.H)
        lc.z2 = causal_conv(lc.o0, prm[f"{p}.1.conv_block2.0.weight"], 2)
        y, lc.xhat2, lc.istd2 = _bn(lc.z2, prm, f"{p}.1.conv_block2.2", train, lc.bn_mean, lc.bn_var, stat_reduce, stat_count)
        lc.x1 = _relu(y)
        lc.o1 = _relu(lc.x1 + lc.o0)
        if train and dropout > 0.0:
            lc.keep = dropout_keep_mask(B, N, dropout_keys[l], dropout, sample_offset)
            dropped = np.where(lc.keep, lc.o1 * dt.type(1.0 / (1.0 - dropout)), dt.type(0))
        else:
            lc.keep = None
            dropped = lc.o1
        X = dropped + X
        fc.layers.append(lc)
    fc.X_out = X
    # AdaptiveMaxPool1d(1) over the 10 channels (after permute), NaN-propagating like torch
    fc.argmax = np.argmax(np.where(np.isnan(X), np.inf, X), axis=1)         # first max / first NaN
    fc.pooled = np.take_along_axis(X, fc.argmax[:, None, :], axis=1)[:, 0, :]     # [B, N]
    fc.y1 = _relu(fc.pooled @ prm["fc1.weight"].T + prm["fc1.bias"])
    fc.pred = fc.y1 @ prm["fc2.weight"].T + prm["fc2.bias"]                 # [B, 1]
    return fc


# ----------------------------------------------------------------------------------------
# backward (manual; checked against torch autograd of the reference via the golden fixtures)
# ----------------------------------------------------------------------------------------
def _bn_backward(dy, xhat, istd, gamma, stat_reduce=None, stat_count=None):
    n = dy.shape[0] * dy.shape[2]
    dgamma = (dy * xhat).sum(axis=(0, 2))
    dbeta = dy.sum(axis=(0, 2))
    if stat_reduce is not None:             # synchronised BatchNorm: [sum dy | sum dy*xhat] over all ranks, global count
        s = stat_reduce(np.concatenate([dbeta, dgamma]))
        dbeta, dgamma, n = s[:dy.shape[1]], s[dy.shape[1]:], stat_count
    dz = (gamma * istd)[None, :, None] * (dy - dbeta[None, :, None] / n - xhat * (dgamma[None, :, None] / n))
    return dz, dgamma, dbeta


def _conv_backward(dz, h, w, dil):
    hs = np.zeros_like(h)
    hs[:, :, dil:] = h[:, :, :-dil] if dil < h.shape[2] else 0
    dw = np.stack([np.einsum("bot,bit->oi", dz, hs), np.einsum("bot,bit->oi", dz, h)], axis=-1)
    dzs = np.zeros_like(dz)                         # dz[t + dil]
    dzs[:, :, :-dil] = dz[:, :, dil:] if dil < h.shape[2] else 0
    dh = np.einsum("oi,bot->bit", w[:, :, 1], dz) + np.einsum("oi,bot->bit", w[:, :, 0], dzs)
    return dh, dw


def backward(prm: dict, fc: FwdCache, dpred: np.ndarray, dropout: float = 0.0, stat_reduce=None, stat_count=None) -> dict:
    """Gradients of sum(pred * dpred) w.r.t. every live parameter (train-mode BN).  With ``stat_reduce`` (synchronised
    BatchNorm) the BatchNorm scale / shift gradients returned are the GLOBAL sums (identical on every rank)."""
    dt = fc.pred.dtype
    prm = {k: np.asarray(v, dtype=dt) for k, v in prm.items()}
    g = {}
    dpred = dpred.reshape(-1, 1).astype(dt)
    g["fc2.weight"] = dpred.T @ fc.y1
    g["fc2.bias"] = dpred.sum(axis=0)
    dy1 = (dpred @ prm["fc2.weight"]) * (fc.y1 > 0)
    g["fc1.weight"] = dy1.T @ fc.pooled
    g["fc1.bias"] = dy1.sum(axis=0)
    dpool = dy1 @ prm["fc1.weight"]                                   # [B, N]
    dX = np.zeros_like(fc.X_out)
    np.put_along_axis(dX, fc.argmax[:, None, :], dpool[:, None, :], axis=1)
    L = len(fc.layers)
    for l in reversed(range(L)):
        p = f"sg_tcn.layers.{l}"
        lc = fc.layers[l]
        d_o1 = dX if lc.keep is None else np.where(lc.keep, dX * dt.type(1.0 / (1.0 - dropout)), dt.type(0))
        gsum = d_o1 * (lc.o1 > 0)                                     # d(x1 + o0)
        dz2, g[f"{p}.1.conv_block2.2.weight"], g[f"{p}.1.conv_block2.2.bias"] = _bn_backward(
            gsum * (lc.x1 > 0), lc.xhat2, lc.istd2, prm[f"{p}.1.conv_block2.2.weight"], stat_reduce, stat_count)
        d_o0, g[f"{p}.1.conv_block2.0.weight"] = _conv_backward(dz2, lc.o0, prm[f"{p}.1.conv_block2.0.weight"], 2)
        d_o0 = d_o0 + gsum
        gsum0 = d_o0 * (lc.o0 > 0)                                    # d(x0 + H)
        dz1, g[f"{p}.1.conv_block1.2.weight"], g[f"{p}.1.conv_block1.2.bias"] = _bn_backward(
            gsum0 * (lc.x0 > 0), lc.xhat1, lc.istd1, prm[f"{p}.1.conv_block1.2.weight"], stat_reduce, stat_count)
        dH, g[f"{p}.1.conv_block1.0.weight"] = _conv_backward(dz1, lc.H, prm[f"{p}.1.conv_block1.0.weight"], 1)
        dH = dH + gsum0
        dHpre = dH * np.where(lc.Hpre > 0, dt.type(1), dt.type(LEAKY_SLOPE))
        for kk in range(len(lc.AXk)):
            g[f"{p}.0.theta.{kk}.weight"] = np.einsum("bcj,bck->jk", dHpre, lc.AXk[kk])
            g[f"{p}.0.theta.{kk}.bias"] = dHpre.sum(axis=(0, 1))
            dX = lc.Apow[kk].transpose(0, 2, 1) @ (dHpre @ prm[f"{p}.0.theta.{kk}.weight"]) + dX
    return g


def mse_loss_and_grad(pred: np.ndarray, y: np.ndarray, global_batch: int | None = None):
    """nn.MSELoss() (mean) -- algorithms.py:44,484.  ``global_batch`` lets a data-parallel
    shard scale its gradient by the GLOBAL batch so that summing shards equals one big batch."""
    B = pred.shape[0] if global_batch is None else global_batch
    diff = pred.reshape(-1) - y.reshape(-1)
    return float((diff * diff).sum() / B), (2.0 / B) * diff


def bn_running_update(prm: dict, fc: FwdCache, num_layers: int) -> dict:
    """nn.BatchNorm1d running-stat update (momentum 0.1, unbiased running var)."""
    out = {}
    for l in range(num_layers):
        lc = fc.layers[l]
        n = lc.z1.shape[0] * lc.z1.shape[2]
        for i, blk in enumerate((1, 2)):
            q = f"sg_tcn.layers.{l}.1.conv_block{blk}.2"
            out[f"{q}.running_mean"] = (1 - BN_MOMENTUM) * prm[f"{q}.running_mean"] + BN_MOMENTUM * lc.bn_mean[i]
            out[f"{q}.running_var"] = (1 - BN_MOMENTUM) * prm[f"{q}.running_var"] + BN_MOMENTUM * lc.bn_var[i] * (n / (n - 1))
    return out


def adam_update(p, grad, m, v, step, lr, weight_decay, beta1=0.9, beta2=0.999, eps=1e-8):
    """One torch.optim.Adam step (L2 weight decay folded into the gradient, no amsgrad)."""
    grad = grad + weight_decay * p
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = np.sqrt(v) / math.sqrt(bc2) + eps
    return p - (lr / bc1) * (m / denom), m, v


def train_step(prm, opt_state, x, y, num_patch, patch_size, num_layers=2, dropout=0.0, seed=0,
               lr=1e-4, weight_decay=1e-4):
    """One ``ST_GCN.update`` (algorithms.py:481-490).  ``opt_state`` = {'step': int, 'm': {}, 'v': {}}.
    Returns (loss, new_prm, new_opt_state, grads, cache)."""
    step = opt_state["step"] + 1
    keys = [dropout_layer_key(seed, step, l) for l in range(num_layers)]
    fc = forward(prm, x, num_patch, patch_size, num_layers, train=True, dropout=dropout, dropout_keys=keys)
    loss, dpred = mse_loss_and_grad(fc.pred, y)
    grads = backward(prm, fc, dpred, dropout)
    new = dict(prm)
    new.update(bn_running_update(prm, fc, num_layers))
    m, v = dict(opt_state["m"]), dict(opt_state["v"])
    for name in live_param_names(num_layers, mpnn_order(prm)):
        g = grads[name].reshape(prm[name].shape)
        m0 = m.get(name, np.zeros_like(prm[name]))
        v0 = v.get(name, np.zeros_like(prm[name]))
        new[name], m[name], v[name] = adam_update(prm[name], g, m0, v0, step, lr, weight_decay)
    return loss, new, {"step": step, "m": m, "v": v}, grads, fc


# ----------------------------------------------------------------------------------------
# metrics -- utils.py:136-201
# ----------------------------------------------------------------------------------------
def rmse_value(pred, real, max_rul):
    pred, real = np.asarray(pred, np.float64), np.asarray(real, np.float64)
    return math.sqrt(float(np.mean((real - pred) ** 2))) * max_rul


def mae_value(pred, real, max_rul):
    pred, real = np.asarray(pred, np.float64), np.asarray(real, np.float64)
    return float(np.mean(np.abs(real - pred))) * max_rul


def score_v1(pred, real, max_rul):
    """utils.py:136-146: asymmetric exponential score (late predictions cost more)."""
    pred, real = np.asarray(pred, np.float64), np.asarray(real, np.float64)
    late = real <= pred
    e = np.where(late, np.exp((pred - real) * max_rul / 10) - 1, np.exp((real - pred) * max_rul / 13) - 1)
    return float(e.sum())


def score_v2(pred, real):
    """utils.py:157-169."""
    pred, real = np.asarray(pred, np.float64), np.asarray(real, np.float64)
    err = (real - pred) / (real + 1e-8) * 100
    e = np.where(err <= 0, np.exp(-math.log(0.5) * (err / 5)), np.exp(math.log(0.5) * (err / 20)))
    return float(e.mean())
