"""hipGraph capture of ``Algorithm.update`` for launch-bound batch sizes.

At the reference protocol's batch size (100) a training step is a few dozen short kernels and the host spends longer
launching them than the GPU spends running them.  ``GraphedUpdate`` records the device work of one ``update`` (fused
forward/backward kernels, reductions, Adam, BatchNorm bookkeeping) into a hipGraph per input shape -- through
``torch.cuda.CUDAGraph``, which on ROCm is hipStreamBeginCapture / hipGraphLaunch on torch's stream -- and replays it
with ONE launch per step.

A graph bakes its kernel arguments, so nothing that changes from step to step may be an argument:
  * inputs are copied into static device buffers before each replay;
  * the dropout stream position and the Adam step live in the device step state (include/rulgnn.h,
    ``rulgnn_step_state_set``): the first kernel of the step advances them on the device and derives the dropout keys /
    bias corrections from them;
  * workspaces and outputs are the model's per-batch-size buffers, pinned for the lifetime of the graphs.
Numerics are those of the eager path (same kernels, same order): tests/test_graphs_gpu.py checks bit-equality.

Single-process only: with a DataParallel context attached the step contains an RCCL all-reduce and stays eager.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class GraphedUpdate:
    def __init__(self, algorithm, warmup: int = 2, max_graphs: int = 4):
        if getattr(algorithm, "dp", None) is not None:
            raise RuntimeError("hipGraph capture of update() is single-process only (the data-parallel step stays eager)")
        self.algorithm = algorithm
        self.warmup = int(warmup)
        self.max_graphs = int(max_graphs)
        self._seen = {}
        self._graphs = {}
        model, opt = algorithm.model, algorithm.optimizer
        if not model.flat_params.is_cuda:
            raise RuntimeError("enable_graphs() needs the algorithm on a CUDA (ROCm) device: call .to(device) first")
        model._pin_bufs = True
        self._bind_step_state()
        # A REAL move of the model (other device / dtype) rebuilds its flat buffers: every pointer baked into the captured
        # graphs is then stale.  Drop the graphs and re-create the step state; the next updates re-warm and re-capture.
        # (A no-op ``model.to(device)``, which the trainers issue every epoch, keeps the buffers: params.flat_views_intact.)
        listeners = getattr(model, "_reflatten_listeners", None)
        if listeners is None:
            listeners = model._reflatten_listeners = []
        listeners.append(self._invalidate)

    def _bind_step_state(self):
        model, opt = self.algorithm.model, self.algorithm.optimizer
        model._step_state = torch.zeros(_lib.STEP_STATE_BYTES, dtype=torch.uint8, device=model.flat_params.device)
        _lib.check(_lib.load().rulgnn_step_state_set(model._step_state.data_ptr(), int(getattr(model, "_step", 0)), int(opt._steps),
                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rulgnn_step_state_set")

    def _invalidate(self):
        self._graphs.clear()
        self._seen.clear()
        model = self.algorithm.model
        model._pin_bufs = True
        if model.flat_params.is_cuda:
            self._bind_step_state()

    # host-side counters that the eager code advances; a captured step must leave them where they were
    def _counters(self):
        m, o = self.algorithm.model, self.algorithm.optimizer
        return (getattr(m, "_step", None), getattr(m, "_nbt_pending", None), o._steps)

    def _restore(self, c):
        m, o = self.algorithm.model, self.algorithm.optimizer
        if c[0] is not None:
            m._step = c[0]
        if c[1] is not None:
            m._nbt_pending = c[1]
        o._steps = c[2]

    def _advance(self):
        m, o = self.algorithm.model, self.algorithm.optimizer
        if hasattr(m, "_step"):
            m._step += 1
        if hasattr(m, "_nbt_pending"):
            m._nbt_pending += 1
        o._steps += 1

    def update(self, X, y):
        """One training step; returns the loss as a 0-d device tensor."""
        algo = self.algorithm
        key = (tuple(X.shape), tuple(y.shape), X.dtype, y.dtype)
        ent = self._graphs.get(key)
        if ent is None:
            n = self._seen.get(key, 0)
            self._seen[key] = n + 1
            if n < self.warmup or len(self._graphs) >= self.max_graphs:
                return algo._eager_update(X, y)              # also allocates this shape's workspace before any capture
            sx, sy = X.clone(), y.clone()
            before = self._counters()
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                loss = algo._eager_update(sx, sy)
            self._restore(before)                            # capture recorded the work, it did not run it
            ent = self._graphs[key] = (graph, sx, sy, loss)
        graph, sx, sy, loss = ent
        sx.copy_(X)
        sy.copy_(y)
        graph.replay()
        self._advance()
        return loss

    @property
    def num_graphs(self):
        return len(self._graphs)
