"""Flat-buffer layout of the ST_GCN parameters (mirror of csrc/stgcn_device.hpp / include/rulgnn.h).

The reference model (models/ST_GCN/Model.py:197-207) owns 52 state_dict entries; 20 of them are
the live parameters that receive gradients, 16 are BatchNorm buffers and 16 belong to the dead
``net0``/``net1`` branches (Model.py:110-131, constructed but never called).  The HIP kernels read
the live parameters from ONE contiguous fp32 buffer and the BatchNorm running statistics from
another; this module maps reference key names <-> offsets in those buffers."""
from __future__ import annotations

from collections import OrderedDict

NUM_STATS = 10       # channels / graph nodes
TCN_KERNEL = 2


def layer_stride(num_patch: int, k: int = 1) -> int:
    """``k``: MPNN order -- theta is a ModuleList of k Linear(N, N) (models/ST_GCN/Model.py:74-79)."""
    return k * (num_patch * num_patch + num_patch) + 2 * (NUM_STATS * NUM_STATS * TCN_KERNEL + 2 * NUM_STATS)


def param_count(num_patch: int, num_layers: int, k: int = 1) -> int:
    return num_layers * layer_stride(num_patch, k) + num_patch * num_patch + 2 * num_patch + 1


def live_param_layout(num_patch: int, num_layers: int, k: int = 1) -> "OrderedDict[str, tuple[int, tuple[int, ...]]]":
    """name (reference key without ``model.``) -> (offset in floats, shape), in buffer order = the reference's
    ``named_parameters()`` order without the dead ``net0`` / ``net1`` branches."""
    order = k
    N, F, K = num_patch, NUM_STATS, TCN_KERNEL
    out: "OrderedDict[str, tuple[int, tuple[int, ...]]]" = OrderedDict()
    off = 0

    def add(name, shape):
        nonlocal off
        n = 1
        for s in shape:
            n *= s
        out[name] = (off, tuple(shape))
        off += n

    for l in range(num_layers):
        p = f"sg_tcn.layers.{l}"
        for kk in range(order):
            add(f"{p}.0.theta.{kk}.weight", (N, N))
            add(f"{p}.0.theta.{kk}.bias", (N,))
        for blk in (1, 2):
            add(f"{p}.1.conv_block{blk}.0.weight", (F, F, K))
            add(f"{p}.1.conv_block{blk}.2.weight", (F,))
            add(f"{p}.1.conv_block{blk}.2.bias", (F,))
    add("fc1.weight", (N, N))
    add("fc1.bias", (N,))
    add("fc2.weight", (1, N))
    add("fc2.bias", (1,))
    assert off == param_count(N, num_layers, order)
    return out


def bn_buffer_layout(num_layers: int) -> "OrderedDict[str, tuple[int, tuple[int, ...]]]":
    """running_mean / running_var keys -> offset in the [L][2][2][10] BatchNorm buffer."""
    out: "OrderedDict[str, tuple[int, tuple[int, ...]]]" = OrderedDict()
    for l in range(num_layers):
        for b, blk in enumerate((1, 2)):
            base = ((l * 2 + b) * 2) * NUM_STATS
            q = f"sg_tcn.layers.{l}.1.conv_block{blk}.2"
            out[f"{q}.running_mean"] = (base, (NUM_STATS,))
            out[f"{q}.running_var"] = (base + NUM_STATS, (NUM_STATS,))
    return out


def bn_buffer_count(num_layers: int) -> int:
    return num_layers * 2 * 2 * NUM_STATS


def pack_numpy(state: dict, num_patch: int, num_layers: int, prefix: str = "", k: int = 1):
    """state_dict-like mapping of numpy arrays -> (flat params, flat bn) float32 numpy arrays."""
    import numpy as np

    flat = np.zeros(param_count(num_patch, num_layers, k), np.float32)
    for name, (off, shape) in live_param_layout(num_patch, num_layers, k).items():
        a = np.asarray(state[prefix + name], np.float32)
        assert tuple(a.shape) == shape, (name, a.shape, shape)
        flat[off:off + a.size] = a.reshape(-1)
    bn = np.zeros(bn_buffer_count(num_layers), np.float32)
    for name, (off, shape) in bn_buffer_layout(num_layers).items():
        bn[off:off + NUM_STATS] = np.asarray(state[prefix + name], np.float32)
    return flat, bn


def unpack_numpy(flat, num_patch: int, num_layers: int, k: int = 1) -> dict:
    return {name: flat[off:off + int(__import__("numpy").prod(shape))].reshape(shape)
            for name, (off, shape) in live_param_layout(num_patch, num_layers, k).items()}


# ---- flat-buffer view bookkeeping shared by every model module --------------------------------------------------------
def count_flat_views(module) -> int:
    """How many of the module's parameters / buffers are still views into its flat device buffers
    (``_flat``: parameters, ``_bn``: BatchNorm statistics, ``_nbt``: BatchNorm counters)."""
    owners = {}
    for name in ("_flat", "_bn", "_nbt"):
        t = getattr(module, name, None)
        if t is not None:
            owners[t.untyped_storage().data_ptr()] = t
    n = 0
    for t in list(module.parameters()) + list(module.buffers()):
        o = owners.get(t.untyped_storage().data_ptr())
        if o is not None and t.device == o.device and t.dtype == o.dtype:
            n += 1
    return n


def flat_views_intact(module) -> bool:
    """True when an ``nn.Module._apply`` (``.to()``, ``.cuda()``, ``.float()`` ...) left every view where ``_reflatten`` put it
    -- the per-epoch ``model.to(device)`` of the trainers is such a no-op and must not reallocate the buffers
    (captured hipGraphs and the optimizer state point into them)."""
    want = getattr(module, "_flat_view_count", None)
    return want is not None and getattr(module, "_flat", None) is not None and count_flat_views(module) == want


def mark_flat_views(module) -> None:
    """Call at the end of ``_reflatten``: remembers the view census and tells listeners (graphs.GraphedUpdate) that every
    device pointer of the module changed."""
    module._flat_view_count = count_flat_views(module)
    for hook in list(getattr(module, "_reflatten_listeners", ())):
        hook()


class SideStream:
    """Second HIP stream a model hands to its training calls (``aux_stream`` of the argument structs, include/rulgnn.h): the backward's
    parameter-gradient GEMMs run on it beside the data-gradient chain and are joined before the call returns its last kernel, so the
    caller's stream semantics do not change.  ``enabled = False`` keeps everything on the current stream.  Under hipGraph capture the
    fork / join events become edges of the graph: the replayed step keeps the two branches."""

    def __init__(self):
        self.enabled, self._stream = True, None

    def pointer(self, device, training: bool = True):
        import torch
        if not (self.enabled and training):
            return None
        if self._stream is None or self._stream.device != device:
            self._stream = torch.cuda.Stream(device=device)
        return self._stream.cuda_stream


class ForwardTape:
    """Hazard check of the autograd path of the flat-parameter models.  The activations a backward needs live in ONE workspace per
    batch size (``model._bufs[B]``), written by every forward of that size: a second ``model(x)`` between a forward and its backward
    (two-view losses, an evaluation inside a step) or an eviction of the workspace would silently give wrong gradients.  Every forward
    that writes a workspace calls ``mark(B)``; the autograd Function keeps the token and ``check`` raises when it is stale."""

    def __init__(self):
        self.tokens, self.count = {}, 0

    def mark(self, batch: int) -> int:
        self.count += 1
        self.tokens[int(batch)] = self.count
        return self.count

    def consume(self, batch: int, token: int) -> None:
        """For backwards that rework the saved buffers in place (STNet scales the reconstruction gradients, RGCNU its gate tape): after
        one backward the forward's activations are gone; a second one (``retain_graph=True``, two losses backpropagated separately)
        must raise instead of silently applying the in-place step twice."""
        if self.tokens.get(int(batch)) == token:
            self.tokens[int(batch)] = ("consumed", token)

    def check(self, batch: int, token: int, bufs: dict, name: str) -> None:
        if self.tokens.get(int(batch)) == ("consumed", token):
            raise RuntimeError(f"{name}: backward() ran twice for the same forward; its saved activations were reworked in place by the "
                               "first one. Run the forward again (or sum the losses and call backward() once).")
        if self.tokens.get(int(batch)) != token or int(batch) not in bufs:
            raise RuntimeError(f"{name}: another forward of this batch size ran between this forward and its backward (or its workspace "
                               "was evicted); the saved activations live in one workspace per batch size, not per call, and were "
                               "overwritten. Call backward() before the next model(x), or use Algorithm.update.")
