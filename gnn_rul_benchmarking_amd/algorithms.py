"""Plugin registry: ``--GNN_method NAME`` -> Algorithm class, as the reference's
algorithms/algorithms.py:29-48 does it (lookup by name in this module's globals,
``NotImplementedError("Algorithm not found: ...")`` otherwise).

The ST_GCN (reference algorithms/algorithms.py:465-490), STMSGCN (:546-571), ASTGCNN (:139-163), FC_STGNN (:51-76), HAGCN (:222-248)
and ST_Conv (:195-220) wrappers are implemented:
the hot paths this package accelerates.  The classes keep the reference contract -- constructor
``(configs, hparams, device)``, attributes ``model`` / ``optimizer`` / ``hparams`` / ``mse``,
``update(X, y, epoch) -> {'loss': float}`` -- so the reference's trainer can drive it unchanged."""
from __future__ import annotations

import torch
import torch.nn as nn

from .optim import FusedAdam
from .astgcnn import ASTGCNN_model
from .fcstgnn import FC_STGNN_RUL
from .hagcn import HAGCN_model
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

    def __init__(self, configs):
        super(Algorithm, self).__init__()
        self.configs = configs
        self.mse = nn.MSELoss()

    supports_graphs = True       # HAGCN (torch Adam + selection bookkeeping) and STGNN opt out

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


class ST_GCN(Algorithm):
    """ST_GCN training wrapper.  ``update`` = forward + MSE + backward + Adam, as the reference
    (algorithms.py:481-490), executed as one fused HIP forward/backward call plus one fused Adam
    kernel; with a ``DataParallel`` context attached (dp.py) the gradient bucket is all-reduced
    over RCCL in between.

    ``sync_loss``: the reference returns ``loss.item()`` (a host sync every step).  That stays the
    default; ``sync_loss=False`` returns the 0-d device tensor instead so a training loop can read
    it once per epoch."""

    def __init__(self, configs, hparams, device):
        super(ST_GCN, self).__init__(configs)
        self.model = ST_GCN_model(**configs)
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
        model = self.model
        if not model.training:
            raise RuntimeError("update() needs algorithm.train() (BatchNorm batch statistics, dropout)")
        graphed = self.dp is None and getattr(self, "_graphed", None) is not None
        loss = self._one_step(X, y, global_batch, sample_offset)
        # f16 range guard of the matrix-core chain (RULGNN_STEP_MX): a NaN loss with every piece of state untouched.  With the
        # reference's per-step loss read-back the step is repeated on the fp32 chain and the model stays there (data parallel: the
        # NaN is part of the all-reduced bucket, so every rank takes this branch).  A captured hipGraph replays one path: no retry.
        if self.sync_loss and not graphed and getattr(model, "guard_tensor", None) is not None and not bool(torch.isfinite(loss)):
            model.retry_on_fp32_chain(self.optimizer)
            loss = self._one_step(X, y, global_batch, sample_offset)
        return self._finish(loss)

    def _one_step(self, X, y, global_batch=None, sample_offset=None):
        if self.dp is not None:
            return self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        if getattr(self, "_graphed", None) is not None:
            return self._graphed.update(X, y)
        return self._eager_update(X, y)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_train_step(X, y, self.optimizer)      # one C call: fwd + MSE + bwd + Adam + BN stats
        return loss

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd (slower: ~20 tiny accumulate ops);
        kept for API parity and tested to give the same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class STMSGCN(Algorithm):
    """STMSGCN training wrapper (reference algorithms.py:546-571): ``update`` = forward + MSE + backward + Adam in one
    C call (SED/GCN/GRU kernels of csrc/stmsgcn.hip + the fused Adam kernel).  The model has neither BatchNorm nor
    dropout, so train and eval mode compute the same function."""

    def __init__(self, configs, hparams, device):
        super(STMSGCN, self).__init__(configs)
        self.model = STMSGCN_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        elif getattr(self, "_graphed", None) is not None:
            loss = self._graphed.update(X, y)
        else:
            loss = self._eager_update(X, y)
        return self._finish(loss)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return loss

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd; same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class ASTGCNN(Algorithm):
    """ASTGCNN training wrapper (reference algorithms.py:139-163): ``update`` = train-mode forward + MSE + backward +
    Adam + BatchNorm running statistics in one C call (csrc/astgcnn.hip + the fused Adam kernel); with a ``DataParallel``
    context attached the gradient bucket is all-reduced in between."""

    def __init__(self, configs, hparams, device):
        super(ASTGCNN, self).__init__(configs)
        self.model = ASTGCNN_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if not self.model.training:
            raise RuntimeError("update() needs algorithm.train() (BatchNorm batch statistics)")
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        elif getattr(self, "_graphed", None) is not None:
            loss = self._graphed.update(X, y)
        else:
            loss = self._eager_update(X, y)
        return self._finish(loss)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return loss

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd; same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class ST_Conv(Algorithm):
    """ST_Conv training wrapper (reference algorithms.py:195-220): ``update`` = train-mode forward + MSE + backward + Adam +
    BatchNorm running statistics in one C call (csrc/stconv.hip + the fused Adam kernel)."""

    def __init__(self, configs, hparams, device):
        super(ST_Conv, self).__init__(configs)
        self.model = ST_Conv_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if not self.model.training:
            raise RuntimeError("update() needs algorithm.train() (BatchNorm batch statistics)")
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        elif getattr(self, "_graphed", None) is not None:
            loss = self._graphed.update(X, y)
        else:
            loss = self._eager_update(X, y)
        return self._finish(loss)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return loss

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd; same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class FC_STGNN(Algorithm):
    """FC_STGNN training wrapper (reference algorithms.py:51-76): ``update`` = train-mode forward + MSE + backward + Adam +
    BatchNorm running statistics in one C call (csrc/fcstgnn.hip + the fused Adam kernel)."""

    def __init__(self, configs, hparams, device):
        super(FC_STGNN, self).__init__(configs)
        self.model = FC_STGNN_RUL(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if not self.model.training:
            raise RuntimeError("update() needs algorithm.train() (BatchNorm batch statistics, dropout)")
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        elif getattr(self, "_graphed", None) is not None:
            loss = self._graphed.update(X, y)
        else:
            loss = self._eager_update(X, y)
        return self._finish(loss)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return loss

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd; same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class HAGCN(Algorithm):
    """HAGCN training wrapper (reference algorithms.py:222-248): ``loss = mse + alpha * KL`` with the reference's literal
    autograd sequence.  The graph stack is one HIP autograd function (csrc/hagcn.hip); the Bi-LSTM stack and ``fc`` are
    torch modules on the vendor libraries, so the optimizer is ``torch.optim.Adam`` exactly as in the reference.  The LSTM
    recurs along batch*nodes (Model.py:153-157): every sample depends on the whole batch, hence replicas only -- no
    data-parallel sharding for this model."""

    supports_graphs = False

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
        loss.backward()
        self.optimizer.step()
        return self._finish(loss.detach())


class STGNN(Algorithm):
    """STGNN training wrapper (reference algorithms.py:383-408): ``update`` = forward + MSE + backward + Adam in one C call
    (graph / ChebNet kernels of csrc/stgnn.hip, the GRU of csrc/gru.hip, the fused Adam kernel).  The model has neither
    BatchNorm nor dropout: samples are independent and data parallelism is the plain ``[gradient | loss]`` bucket."""

    supports_graphs = False      # update() does not consult a captured graph

    def __init__(self, configs, hparams, device):
        super(STGNN, self).__init__(configs)
        self.model = STGNN_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        else:
            loss = self._eager_update(X, y)
        return self._finish(loss)

    def _eager_update(self, X, y):
        _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return loss

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd; same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class RGCNU(Algorithm):
    """RGCNU training wrapper (reference algorithms.py:250-296): ``update`` = train-mode forward of both heads + MSE of the first
    head + backward + Adam in one C call (kernels of csrc/rgcnu.hip, the one-direction LSTM of csrc/bilstm.hip, the fused Adam
    kernel).  ``lambda`` (the weight of the reference's commented-out uncertainty loss, :266-282) is kept as ``lambda_hy`` and,
    like there, unused.  No BatchNorm: samples are independent except for the adjacency-tiling quirk (rgcnu.py), which couples the
    samples of one call -- data parallelism shards the batch like the other families and therefore pairs graphs with the adjacencies
    of the SHARD (a different, equally arbitrary pairing than the single-process batch; stated, not hidden)."""

    supports_graphs = False

    def __init__(self, configs, hparams, device):
        super(RGCNU, self).__init__(configs)
        self.model = RGCNU_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.lambda_hy = hparams.get("lambda", 0.1)
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if not self.model.training:
            raise RuntimeError("update() needs algorithm.train() (SCL's dropout)")
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        else:
            _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return self._finish(loss)

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd (algorithms.py:284-296); same result as ``update``."""
        predicted_RUL, std_RUL = self.model(X, train=True)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class STNet(Algorithm):
    """STNet training wrapper (reference algorithms.py:438-463): ``update`` = forward + MSE + the auto-encoder's reconstruction loss +
    backward + Adam in one C call (csrc/stnet.hip, the one-direction LSTM of csrc/bilstm.hip, matrix-core GEMMs, the fused Adam
    kernel).  No BatchNorm, no dropout: samples are independent and data parallelism is the plain ``[gradient | loss]`` bucket (the
    reconstruction term is averaged over the GLOBAL batch's elements)."""

    supports_graphs = False

    def __init__(self, configs, hparams, device):
        super(STNet, self).__init__(configs)
        self.model = STNet_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        else:
            _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return self._finish(loss)

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd (algorithms.py:454-463); same result as ``update``."""
        predicted_RUL, reconstruction_losss = self.model(X, train=True)
        loss = self.mse(predicted_RUL, y) + reconstruction_losss
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class SAGCN(Algorithm):
    """SAGCN training wrapper (reference algorithms.py:412-436): ``update`` = forward + MSE + backward + Adam in one C call
    (csrc/sagcn.hip, matrix-core GEMMs, the fused Adam kernel).  No BatchNorm, no dropout: samples are independent and data
    parallelism is the plain ``[gradient | loss]`` bucket."""

    supports_graphs = False

    def __init__(self, configs, hparams, device):
        super(SAGCN, self).__init__(configs)
        self.model = SAGCN_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        else:
            _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return self._finish(loss)

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd (algorithms.py:427-436); same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


class STAGNN(Algorithm):
    """STAGNN training wrapper (reference algorithms.py:298-323): ``update`` = train-mode forward (batch statistics in the four
    BatchNorm1d layers, running statistics updated) + MSE + backward + Adam in one C call (csrc/stagnn.hip, the fused Adam kernel).
    Data parallelism: the plain ``[gradient | loss]`` bucket with rank-local BatchNorm statistics."""

    supports_graphs = False

    def __init__(self, configs, hparams, device):
        super(STAGNN, self).__init__(configs)
        self.model = STAGNN_model(**configs)
        self.optimizer = FusedAdam(self.model, lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"])
        self.hparams = hparams
        self.dp = None
        self.sync_loss = True

    def attach_data_parallel(self, dp):
        self.dp = dp
        dp.broadcast_model(self.model)

    def update(self, X, y, epoch=None, global_batch=None, sample_offset=None):
        if not self.model.training:
            raise RuntimeError("update() needs algorithm.train() (BatchNorm batch statistics)")
        if self.dp is not None:
            loss = self.dp.step(self.model, self.optimizer, X, y, global_batch, sample_offset)
        else:
            _, loss = self.model.fused_mse_step(X, y, self.optimizer)
        return self._finish(loss)

    def update_reference_style(self, X, y, epoch=None):
        """The reference's literal sequence through autograd (algorithms.py:314-323); same result as ``update``."""
        predicted_RUL = self.model(X)
        loss = self.mse(predicted_RUL, y)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return {'loss': loss.item()}


_NOT_ALGORITHMS = {"Algorithm", "FusedAdam", "RGCNU_model", "STNet_model", "SAGCN_model", "STAGNN_model", "ST_GCN_model", "STMSGCN_model", "ASTGCNN_model", "FC_STGNN_RUL", "HAGCN_model", "ST_Conv_model", "STGNN_model", "get_algorithm_class", "torch", "nn", "annotations"}
