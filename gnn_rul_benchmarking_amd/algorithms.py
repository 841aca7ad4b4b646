"""Plugin registry: ``--GNN_method NAME`` -> Algorithm class, as the reference's
algorithms/algorithms.py:29-48 does it (lookup by name in this module's globals,
``NotImplementedError("Algorithm not found: ...")`` otherwise).

The ST_GCN (reference algorithms/algorithms.py:465-490), STMSGCN (:546-571), ASTGCNN (:139-163), FC_STGNN (:51-76), HAGCN (:222-248)
and ST_Conv (:195-220) wrappers are implemented:
the hot paths this package accelerates.  The classes keep the reference contract -- constructor
``(configs, hparams, device)``, attributes ``model`` / ``optimizer`` / ``hparams`` / ``mse``,
``update(X, y, epoch) -> {'loss': float}`` -- so the reference's trainer can drive it unchanged."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import _lib
from .optim import FusedAdam
from .astgcnn import ASTGCNN_model
from .fcstgnn import FC_STGNN_RUL
from .hagcn import HAGCN_model, deferred_weight_gradients
from .rgcnu import RGCNU_model
from .stconv import ST_Conv_model
from .stgcn import ST_GCN_model
from .stgnn import STGNN_model
from .stmsgcn import STMSGCN_model
from .stnet import STNet_model
from .sagcn import SAGCN_model
from .stagnn import STAGNN_model


def get_algorithm_class(algorithm_name):
    """Return the algorithm class with the given name."""
    if algorithm_name not in globals() or algorithm_name.startswith("_") or algorithm_name in _NOT_ALGORITHMS:
        raise NotImplementedError("Algorithm not found: {}".format(algorithm_name))
    return globals()[algorithm_name]


class Algorithm(torch.nn.Module):
    """Base class (reference algorithms/algorithms.py:36-48): subclasses define ``update``."""

    supports_graphs = True       # HAGCN (torch Adam + selection bookkeeping) and the families whose update() has no graph branch opt out

    def __init__(self, configs):
        super(Algorithm, self).__init__()
        self.configs = configs
        self.mse = nn.MSELoss()

    def update(self, *args, **kwargs):
        raise NotImplementedError

    def enable_graphs(self, warmup=2):
        """Replay ``update`` from a hipGraph per input shape (graphs.py): one launch per step instead of dozens --
        for launch-bound batch sizes such as the reference protocol's 100.  Single process only."""
        from .graphs import GraphedUpdate
        if not self.supports_graphs:
            raise RuntimeError(f"{type(self).__name__}.update cannot be captured in a hipGraph "
                               "(its step has host-side control flow); it stays eager")
        self._graphed = GraphedUpdate(self, warmup=warmup)
        return self

    def _finish(self, loss):
        return {'loss': loss.item() if self.sync_loss else loss}


class _FusedAlgorithm(Algorithm):
    """What the ten fused-step wrappers share (the reference repeats this body per class, algorithms.py:51-76, :139-163, ...;
    here it exists once): the model of ``model_class(**configs)``, ``FusedAdam`` over its flat buffer, ``update`` = forward + loss +
    backward + Adam as one C call (``model.fused_mse_step``) -- through ``dp.DataParallel`` when one is attached (the gradient bucket is
    all-reduced over RCCL in between), through a captured hipGraph when ``enable_graphs()`` was called.

    ``sync_loss``: the reference returns ``loss.item()`` (a host sync every step).  That stays the default; ``sync_loss=False``
    returns the 0-d device tensor instead so that a training loop can read it once per epoch -- a VIEW of the loss slot of the model's
    bucket (no extra kernel per step): valid until the next ``update()``, ``clone()`` it to keep it."""

    model_class = None
    needs_train_mode = None      # why update() refuses a model in eval mode (None: train and eval compute the same function)

    def __init__(self, configs, hparams, device):
        super().__init__(configs)
        self.model = self.model_class(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None            # optional dp.DataParallel
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        """``global_batch`` / ``sample_offset`` (data parallel only): size of the whole batch this shard
        belongs to and the shard's first index in it; default = equal shards, rank-ordered."""
        if self.needs_train_mode and not self.model.training:
            raise RuntimeError(f"update() needs algorithm.train() ({self.needs_train_mode})")
        return self._finish(self._one_step(X, y, global_batch, sample_offset))

    def _one_step(self, X, y, global_batch=None, sample_offset=None):
        if self.dp is not None:
            return self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        if self.supports_graphs and getattr(self, "_graphed", None) is not None:
            return self._graphed.update(X, y)
        return self._eager_update(X, y)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return loss

    def _reference_loss(self, X, y):
        return self.mse(self.model(X), y)

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd (slower: the gradient is scattered into ~20 ``.grad`` tensors and
        gathered again); kept for API parity and tested to give the same result as ``update``."""
        loss = self._reference_loss(X, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class ST_GCN(_FusedAlgorithm):
    """ST_GCN training wrapper (reference algorithms.py:465-490): the whole ``update`` body -- forward, MSE, backward, Adam, BatchNorm
    running statistics -- is ONE C call (``ST_GCN_model.fused_train_step``).

    f16 range guard of the matrix-core chain (RULGNN_STEP_MX, include/rulgnn.h): a step whose arithmetic left the f16 range comes back
    as a NaN loss with every piece of state untouched.  The reference never drops a step (algorithms.py:486-490), so:
      * ``sync_loss=True`` (default): the NaN is seen at once, the step is repeated on the fp32 chain and the model stays there (data
        parallel: the NaN is part of the all-reduced bucket, so every rank takes this branch);
      * ``sync_loss=False`` / a captured hipGraph (no per-step read-back): the kernels count rejected steps in a sticky device counter;
        ``check_guard()`` -- called by ``model.eval()``, ``state_dict()`` and available to the caller -- raises if any step was dropped."""

    model_class = ST_GCN_model
    needs_train_mode = "BatchNorm batch statistics, dropout"

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        model = self.model
        if not model.training:
            raise RuntimeError(f"update() needs algorithm.train() ({self.needs_train_mode})")
        loss = self._one_step(X, y, global_batch, sample_offset)
        if not self.sync_loss:
            return {'loss': loss}
        value = loss.item()                                   # the ONE host read-back of the step
        if not math.isfinite(value) and self.dp is not None and getattr(self.dp, "peer", None) is not None:
            self.dp.peer.check()                              # a one-shot BatchNorm collective that gave up on a peer: say so, do not retry
        graphed = self.dp is None and getattr(self, "_graphed", None) is not None
        if not math.isfinite(value) and getattr(model, "guard_tensor", None) is not None:
            if graphed:                                       # a captured graph replays one path: no retry from here
                model.check_guard()
            else:
                model.guard_trips()                           # this trip is being handled: take it off the sticky count
                model.retry_on_fp32_chain(self.optimizer)
                value = self._one_step(X, y, global_batch, sample_offset).item()
        return {'loss': value}

    def check_guard(self):
        self.model.check_guard()

    def _one_step(self, X, y, global_batch=None, sample_offset=None):
        loss = super()._one_step(X, y, global_batch, sample_offset)
        if self.model._last_chain == _lib.STEP_MX:      # (a graph replay does not pass through the model's own bookkeeping)
            self.model._guard_unchecked = True
        return loss

    def _eager_update(self, X, y):
        _, loss = self.model.fused_train_step(X, y, self.optimizer)      # one C call: fwd + MSE + bwd + Adam + BN stats
        return loss


class STMSGCN(_FusedAlgorithm):
    """STMSGCN training wrapper (reference algorithms.py:546-571; SED/GCN/GRU kernels of csrc/stmsgcn.hip).  The model has neither
    BatchNorm nor dropout, so train and eval mode compute the same function."""
    model_class = STMSGCN_model


class ASTGCNN(_FusedAlgorithm):
    """ASTGCNN training wrapper (reference algorithms.py:139-163; csrc/astgcnn.hip): train-mode forward, BatchNorm running statistics
    inside the step."""
    model_class = ASTGCNN_model
    needs_train_mode = "BatchNorm batch statistics"


class ST_Conv(_FusedAlgorithm):
    """ST_Conv training wrapper (reference algorithms.py:195-220; csrc/stconv.hip)."""
    model_class = ST_Conv_model
    needs_train_mode = "BatchNorm batch statistics"


class FC_STGNN(_FusedAlgorithm):
    """FC_STGNN training wrapper (reference algorithms.py:51-76; csrc/fcstgnn.hip)."""
    model_class = FC_STGNN_RUL
    needs_train_mode = "BatchNorm batch statistics, dropout"


class HAGCN(Algorithm):
    """HAGCN training wrapper (reference algorithms.py:222-248): ``loss = mse + alpha * KL`` with the reference's literal
    autograd sequence.  The graph stack is one HIP autograd function (csrc/hagcn.hip); the Bi-LSTM stack and ``fc`` are
    torch modules over this package's recurrent kernels, so the optimizer is ``torch.optim.Adam`` exactly as in the reference.  The LSTM
    recurs along batch*nodes (Model.py:153-157): every sample depends on the whole batch, hence replicas only -- no
    data-parallel sharding for this model."""

    supports_graphs = False
    model_class = HAGCN_model

    def __init__(self, configs, hparams, device):
        super(HAGCN, self).__init__(configs)
        self.model = HAGCN_model(**configs)
        # the reference's torch.optim.Adam (algorithms.py:226-230) in its fused (one multi-tensor kernel) form: same update rule,
        # ~2 launches per step instead of ~14 foreach kernels over the model's ~45 parameter tensors
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"],
                                          fused=True)
        self.hparams = hparams
        self.alpha = hparams["alpha"]
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        raise RuntimeError("HAGCN is not sample-shardable (its LSTM recurs along batch*nodes): run independent replicas")

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        predicted_RUL, KL_Loss = self.model(X, train=True)
        loss_mse = self.mse(predicted_RUL, y)
        loss = loss_mse + self.alpha * KL_Loss
        self.optimizer.zero_grad()
        with deferred_weight_gradients(loss.device):      # LSTM layer l's weight-gradient GEMMs under layer l-1's BPTT (hagcn.py)
            loss.backward()
        self.optimizer.step()
        return self._finish(loss.detach())


class STGNN(_FusedAlgorithm):
    """STGNN training wrapper (reference algorithms.py:383-408; graph / ChebNet kernels of csrc/stgnn.hip, the GRU of csrc/gru.hip).
    Neither BatchNorm nor dropout: samples are independent and data parallelism is the plain ``[gradient | loss]`` bucket."""
    model_class = STGNN_model
    supports_graphs = False


class RGCNU(_FusedAlgorithm):
    """RGCNU training wrapper (reference algorithms.py:250-296): train-mode forward of both heads + MSE of the first head (csrc/rgcnu.hip,
    the one-direction LSTM of csrc/bilstm.hip).  ``lambda`` (the weight of the reference's commented-out uncertainty loss, :266-282) is
    kept as ``lambda_hy`` and, like there, unused.  No BatchNorm: samples are independent except for the adjacency-tiling quirk
    (rgcnu.py), which couples the samples of one call -- data parallelism shards the batch like the other families and therefore pairs
    graphs with the adjacencies of the SHARD (a different, equally arbitrary pairing than the single-process batch; stated, not hidden)."""
    model_class = RGCNU_model
    supports_graphs = False
    needs_train_mode = "SCL's dropout"

    def __init__(self, configs, hparams, device):
        super().__init__(configs, hparams, device)
        self.lambda_hy = hparams.get("lambda", 0.1)

    def _reference_loss(self, X, y):
        predicted_RUL, std_RUL = self.model(X, train=True)       # algorithms.py:284-296
        return self.mse(predicted_RUL, y)


class STNet(_FusedAlgorithm):
    """STNet training wrapper (reference algorithms.py:438-463): MSE + the auto-encoder's reconstruction loss (csrc/stnet.hip, the
    one-direction LSTM of csrc/bilstm.hip, matrix-core GEMMs).  No BatchNorm, no dropout; in data parallel the reconstruction term is
    averaged over the GLOBAL batch's elements."""
    model_class = STNet_model
    supports_graphs = False

    def _reference_loss(self, X, y):
        predicted_RUL, reconstruction_losss = self.model(X, train=True)       # algorithms.py:454-463
        return self.mse(predicted_RUL, y) + reconstruction_losss


class SAGCN(_FusedAlgorithm):
    """SAGCN training wrapper (reference algorithms.py:412-436; csrc/sagcn.hip, matrix-core GEMMs).  No BatchNorm, no dropout."""
    model_class = SAGCN_model
    supports_graphs = False


class STAGNN(_FusedAlgorithm):
    """STAGNN training wrapper (reference algorithms.py:298-323; csrc/stagnn.hip): batch statistics in the four BatchNorm1d layers,
    running statistics updated inside the step.  Data parallelism: rank-local BatchNorm statistics."""
    model_class = STAGNN_model
    supports_graphs = False
    needs_train_mode = "BatchNorm batch statistics"


_NOT_ALGORITHMS = {"Algorithm", "FusedAdam", "RGCNU_model", "STNet_model", "SAGCN_model", "STAGNN_model", "ST_GCN_model", "STMSGCN_model", "ASTGCNN_model", "FC_STGNN_RUL", "HAGCN_model", "ST_Conv_model", "STGNN_model", "get_algorithm_class", "torch", "nn", "annotations", "math", "_lib", "deferred_weight_gradients"}
