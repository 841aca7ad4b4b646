"""Dataset loading with the reference's on-disk format and batch semantics (dataloader/dataloader.py:13-94),
re-designed for a 288 GB HBM device: the whole dataset is moved to the GPU ONCE and batches are index
gathers on the device, instead of a pageable host->device copy per batch.

Format: ``<data_path>/train.pt`` and ``test.pt`` = torch.save'd dicts {'samples', 'labels', 'max_ruls'}
(writers: Data_Process/Data_read_*.py).  ``test.pt`` may hold dicts keyed by test-set id.
Batch composition follows torch's own samplers (RandomSampler when the dataset config says shuffle,
drawing from the global torch RNG exactly like DataLoader does), so epochs see the same batches as the
reference's DataLoader for a given seed."""
from __future__ import annotations

import os

import numpy as np
import torch
from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler


def _normalise(X, y):
    """dataloader.py:16-30: to tensors, [n, C, L] with the channel axis second, labels [n, 1]."""
    X, y = np.array(X), np.array(y)
    X, y = torch.from_numpy(X), torch.from_numpy(y)
    if X.dim() < 3:
        X = X.unsqueeze(2)
    if X.shape.index(min(X.shape[1], X.shape[2])) != 1:
        X = X.permute(0, 2, 1)
    if y.dim() == 1:
        y = y.unsqueeze(-1)
    return X.float().contiguous(), y.float().contiguous()


class DeviceBatchLoader:
    """Iterates (X[idx], y[idx]) with idx from torch's samplers; tensors live on ``device``.
    ``rank``/``world_size``: every rank walks the SAME index batches (same seed) and takes its
    contiguous shard of each (dp.shard_bounds), so the global batch equals the single-process one.

    ``shard_samples`` (the TEST sets under data parallelism): the rank keeps only its contiguous shard
    ``[lo, hi)`` of the set (``dp.shard_bounds(n, world_size, rank)``: shard sizes differ by at most one) and walks it in batches of
    ``batch_size`` -- an eval forward is per-sample independent, so the concatenation of the ranks' predictions in rank order is the
    single-process prediction vector.  ``global_n`` / ``shard`` say where the shard sits."""

    def __init__(self, X, y, batch_size, shuffle, drop_last, device, rank=0, world_size=1, shard_samples=False):
        self.global_n, self.shard = X.shape[0], (0, X.shape[0])
        # (a set that is shuffled or drops its ragged tail is not a fixed vector of samples: every rank then walks all of it)
        self.shard_samples = bool(shard_samples) and world_size > 1 and not shuffle and not drop_last
        if self.shard_samples:
            from .dp import shard_bounds
            lo, hi = shard_bounds(X.shape[0], world_size, rank)
            X, y, self.shard = X[lo:hi], y[lo:hi], (lo, hi)
        if shard_samples:
            # a TEST set: either this rank's own shard (above) or -- when the set cannot be sharded -- the whole set in the
            # reference's batches; never a per-batch slice (a slice made rank 0 report metrics of 1 / world of the set)
            rank, world_size = 0, 1
        self.x_data, self.y_data = X.to(device), y.to(device)
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.device, self.rank, self.world_size = device, rank, world_size
        self.n = X.shape[0]
        print('Dataset size ', self.x_data.size())

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else (self.n + self.batch_size - 1) // self.batch_size

    def index_batches(self):
        sampler = RandomSampler(range(self.n)) if self.shuffle else SequentialSampler(range(self.n))
        return BatchSampler(sampler, self.batch_size, self.drop_last)

    def __iter__(self):
        from .dp import shard_bounds
        # torch's DataLoader iterator draws its `_base_seed` from the global RNG before the sampler draws anything
        # (torch/utils/data/dataloader.py, _BaseDataLoaderIter.__init__); drawing it here too keeps the global RNG -- and
        # with it every later shuffle -- in lockstep with the reference's DataLoader for a given seed.
        torch.empty((), dtype=torch.int64).random_()
        for idx in self.index_batches():
            gb = len(idx)
            lo, hi = shard_bounds(gb, self.world_size, self.rank) if self.world_size > 1 else (0, gb)
            ids = torch.as_tensor(idx[lo:hi], device=self.device)
            yield self.x_data[ids], self.y_data[ids], gb, lo


def data_generator(data_path, dataset_configs, hparams, device="cpu", rank=0, world_size=1, shard_test_sets=True):
    """``shard_test_sets=False``: every rank evaluates every test set whole, in the reference's batches -- for a model whose eval
    forward depends on the composition of the batch (``eval_sample_independent = False`` on the model class, e.g. RGCNU)."""
    train = torch.load(os.path.join(data_path, "train.pt"), weights_only=False)
    test = torch.load(os.path.join(data_path, "test.pt"), weights_only=False)
    bs = hparams["batch_size"]
    Xtr, ytr = _normalise(train['samples'], train['labels'])
    train_loader = DeviceBatchLoader(Xtr, ytr, bs, dataset_configs.shuffle, dataset_configs.drop_last, device, rank, world_size)
    tw = (rank, world_size) if shard_test_sets else (0, 1)
    if isinstance(test['samples'], dict):
        test_loader = {}
        for key in test['samples']:
            Xt, yt = _normalise(test['samples'][key], test['labels'][key])
            test_loader[key] = DeviceBatchLoader(Xt, yt, bs, False, dataset_configs.drop_last, device, *tw, shard_samples=True)
    else:
        Xt, yt = _normalise(test['samples'], test['labels'])
        test_loader = DeviceBatchLoader(Xt, yt, bs, False, dataset_configs.drop_last, device, *tw, shard_samples=True)
    return train_loader, test_loader, train['max_ruls']
