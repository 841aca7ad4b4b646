"""Training harness with the reference's loop semantics (trainer.py:25-262), driving the HIP-backed
Algorithm classes of this package, optionally data-parallel (one process per GPU).

Kept from the reference: run loop x epoch loop x batch loop; ``fix_randomness(run_id)`` per run;
the AverageMeter is created once per run, so the logged loss is a running mean over all epochs so
far (trainer.py:101); per-epoch evaluation on the test set(s) in eval mode; "best" rows appended
whenever the TEST RMSE improves, CSV first row inf,inf,inf,inf (trainer.py:219-231,245-256); one
checkpoint per run after the last epoch (trainer.py:125-126); same directory / file names.

New: the dataset lives on the GPU (dataloader.py); under torch.distributed every rank walks the same
batches and trains on its shard (dp.py); every TEST set is sharded over the ranks too (contiguous shards, the four metric sums and
the count travel in one 5-double all-reduce); only rank 0 logs and writes files."""
from __future__ import annotations

import collections
import os

import numpy as np
import pandas as pd
import torch
import torch.nn.functional as F

from .algorithms import get_algorithm_class
from .data_model_configs import get_dataset_class
from .dataloader import data_generator
from .hparams import get_hparams_class
from .metrics import _calc_metrics, device_metric_sums, metrics_from_sums
from .utils import AverageMeter, fix_randomness, save_checkpoint, starting_logs


class GNN_RUL_trainer(object):
    def __init__(self, args):
        self.GNN_method = args.GNN_method
        self.dataset = args.dataset
        self.dataset_id = args.dataset_id
        self.device = torch.device(args.device)
        self.bearing_id = args.bearing_id
        self.run_description = args.run_description
        self.experiment_description = args.experiment_description
        if self.dataset == 'NCMAPSS':
            self.data_path = os.path.join(args.data_path, self.dataset)
        elif self.dataset in ('CMAPSS', 'PHM2012'):
            self.data_path = os.path.join(args.data_path, self.dataset, self.dataset_id)
        elif self.dataset == 'XJTU_SY':
            self.data_path = os.path.join(args.data_path, self.dataset, self.dataset_id, self.bearing_id)
        self.home_path = os.getcwd()
        self.save_dir = args.save_dir
        self.num_runs = args.num_runs
        self.window = getattr(args, "window", None)
        self.num_epochs_override = getattr(args, "num_epochs", None)

        self.rank, self.world_size, self.dp = 0, 1, None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            from .dp import DataParallel
            self.dp = DataParallel()
            self.rank, self.world_size = self.dp.rank, self.dp.world_size
        if self.rank == 0:
            self.create_save_dir()

        self.dataset_configs, self.hparams_class = self.get_configs(args.dataset_id)
        self.train_configs = self.hparams_class.train_params[self.GNN_method]     # KeyError if the method is not listed
        self.model_configs = self.hparams_class.alg_hparams[self.GNN_method]
        if self.num_epochs_override is not None:
            self.train_configs = dict(self.train_configs, num_epochs=int(self.num_epochs_override))
        self.default_hparams = {**self.model_configs, **self.train_configs}

    def get_configs(self, dataset_id):
        dataset_class = get_dataset_class(self.dataset)
        hparams_class = get_hparams_class(self.dataset)
        kw = {"window": self.window} if (self.window and self.dataset in ("CMAPSS", "NCMAPSS")) else {}
        return dataset_class(), hparams_class(dataset_id, **kw)

    def create_save_dir(self):
        if not os.path.exists(self.save_dir):
            os.mkdir(self.save_dir)

    # ---------------------------------------------------------------------------------------------
    def train(self):
        run_name = f"{self.run_description}"
        self.exp_log_dir = os.path.join(self.save_dir, self.experiment_description, run_name)
        if self.rank == 0:
            os.makedirs(self.exp_log_dir, exist_ok=True)
        if self.dp is not None:
            torch.distributed.barrier()

        for run_id in range(self.num_runs):
            fix_randomness(run_id)
            self.logger, self.log_dir = starting_logs(self.dataset, self.GNN_method, self.exp_log_dir, self.dataset_id,
                                                      self.bearing_id, run_id, to_stdout=self.rank == 0) \
                if self.rank == 0 else (_NullLogger(), None)
            # a model whose eval forward depends on which samples share a batch (RGCNU's adjacency pairing, HAGCN's recurrence along
            # batch x nodes) declares ``eval_sample_independent = False``: its test sets stay whole on every rank, in the reference's batches
            algorithm_class = get_algorithm_class(self.GNN_method)
            shard_test = getattr(getattr(algorithm_class, "model_class", None), "eval_sample_independent", True)
            self.train_dl, self.test_dl, self.max_ruls = data_generator(
                self.data_path, self.dataset_configs, self.train_configs, self.device, self.rank, self.world_size,
                shard_test_sets=shard_test)
            if isinstance(self.test_dl, dict):
                self.best_result = {key: [[np.inf], [np.inf], [np.inf], [np.inf]] for key in self.test_dl.keys()}
            else:
                self.best_result = [[np.inf], [np.inf], [np.inf], [np.inf]]

            algorithm = algorithm_class(self.model_configs, self.train_configs, self.device)
            algorithm.to(self.device)
            if self.dp is not None:
                algorithm.attach_data_parallel(self.dp)

            loss_avg_meters = collections.defaultdict(lambda: AverageMeter())
            for epoch in range(1, self.train_configs["num_epochs"] + 1):
                algorithm.train()
                for step, (X, y, global_batch, offset) in enumerate(self.train_dl):
                    X, y = X.float().to(self.device), y.float().to(self.device)
                    losses = algorithm.update(X, y, epoch, global_batch=global_batch, sample_offset=offset)
                    for key, val in losses.items():
                        loss_avg_meters[key].update(val, global_batch)
                self.logger.debug(f'[Epoch : {epoch}/{self.train_configs["num_epochs"]}]')
                for key, val in loss_avg_meters.items():
                    self.logger.debug(f'{key}\t: {float(val.avg):2.4f}')
                self.test_prediction(algorithm)
                self.calc_results_per_run(run_id)          # every rank: the sharded metrics are a collective; rank 0 writes
                self.logger.debug('-------------------------------------')

            self.algorithm = algorithm
            if self.rank == 0:
                save_checkpoint(self.home_path, self.algorithm, self.dataset_configs, self.log_dir, self.default_hparams)

    # ---------------------------------------------------------------------------------------------
    def test_base(self, model, test_dataloader):
        preds, trues, loss_total = [], [], []
        with torch.no_grad():
            for data, labels, _, _ in test_dataloader:
                data = data.float().to(self.device)
                labels = labels.view((-1)).float().to(self.device)
                predictions = model(data).view((-1))
                loss_total.append(F.mse_loss(predictions, labels))
                preds.append(predictions.detach())
                trues.append(labels)
        losses = [float(v) for v in torch.stack(loss_total).cpu()] if loss_total else []
        if not preds:                 # an empty set -- or this rank's empty shard of a set smaller than the world
            if self.device.type == "cuda":
                return torch.empty(0, device=self.device), torch.empty(0, device=self.device), losses
            return np.array([]), np.array([]), losses
        pred_labels, true_labels = torch.cat(preds), torch.cat(trues)
        if pred_labels.is_cuda:
            # the predictions stay on the GPU: the metrics are reduced there (metrics.device_metrics) and they only travel
            # to the host when a best row is saved -- instead of one copy per batch (trainer.py:148-152)
            return pred_labels, true_labels, losses
        return pred_labels.numpy().astype(np.float64), true_labels.numpy().astype(np.float64), losses

    def test_prediction(self, algorithm):
        model = algorithm.model.to(self.device)
        model.eval()
        if isinstance(self.test_dl, dict):
            test_pre, test_real, test_total_loss = {}, {}, {}
            for key, dl in self.test_dl.items():
                pre_i, real_i, loss_i = self.test_base(model, dl)
                test_pre[key], test_real[key], test_total_loss[key] = pre_i, real_i, torch.tensor(loss_i).mean()      # (of this rank's batches)
        else:
            test_pre, test_real, test_total_loss = self.test_base(model, self.test_dl)
            test_total_loss = torch.tensor(test_total_loss).mean()
        self.pred_labels, self.true_labels, self.total_loss = test_pre, test_real, test_total_loss

    def calc_results_per_run(self, run_id):
        names = ('Score_v1', 'Score_v2', 'MAE', 'RMSE')
        save_path = os.path.join(self.exp_log_dir, self.GNN_method + "_run_" + str(run_id))

        def one(pred, real, max_rul, best, stem, label, loader):
            sharded = self.dp is not None and getattr(loader, "shard_samples", False)
            ind = sharded_metrics(pred, real, max_rul, self.dp) if sharded else _calc_metrics(pred, real, max_rul)
            if ind[3] < best[3][-1]:          # the same decision on every rank (they hold the same all-reduced numbers)
                for i in range(4):
                    best[i].append(ind[i])
                if sharded:                   # a best row is saved with its predictions: collect the shards, in rank order = sample order
                    pred, real = gather_shards(pred, loader, self.dp), gather_shards(real, loader, self.dp)
                if self.rank == 0:
                    host = lambda v: v.cpu().numpy().astype(np.float64) if torch.is_tensor(v) else v
                    torch.save({'pre': host(pred), 'real': host(real), 'max_rul': max_rul}, os.path.join(save_path, f"{stem}results.pt"))
            if self.rank == 0:
                pd.DataFrame({n: best[i] for i, n in enumerate(names)}).to_csv(os.path.join(save_path, f"{stem}results.csv"),
                                                                              index=False)
            self.logger.debug(f'Testing{label}, ' + ', '.join(f'{n}: {best[i][-1]}' for i, n in enumerate(names)))

        if isinstance(self.pred_labels, dict):
            for key in self.pred_labels.keys():
                key_save = int(key) if isinstance(key, float) else key
                one(self.pred_labels[key], self.true_labels[key], self.max_ruls[key], self.best_result[key],
                    f"{key_save}_", f" {key_save}", self.test_dl[key])
        else:
            one(self.pred_labels, self.true_labels, self.max_ruls, self.best_result, "", "", self.test_dl)


def sharded_metrics(pred, real, max_rul, dp):
    """(Score_v1, Score_v2, MAE, RMSE) of a test set whose predictions are spread over the ranks (reference formulas utils.py:136-201):
    each rank reduces ITS shard on the device to four fp64 sums (``rulgnn_rul_metric_sums_f32``), one all-reduce carries them and the
    counts, every rank closes with the same divisions.  A rank whose shard is empty (fewer samples than ranks) contributes zeros."""
    if torch.is_tensor(pred) and pred.is_cuda:
        sums = device_metric_sums(pred, real, max_rul)
    else:                                     # host arrays (CPU tests of the collective logic): the vectorised numpy forms, as sums
        p, r = np.asarray(pred, np.float64).reshape(-1), np.asarray(real, np.float64).reshape(-1)
        from .metrics import scoring_function, scoring_function_v2
        n = p.shape[0]
        s = [scoring_function(p, r, max_rul)[0], scoring_function_v2(p, r) * n, float(np.abs(r - p).sum()), float(((r - p) ** 2).sum()),
             float(n)] if n else [0.0] * 5
        sums = torch.tensor(s, dtype=torch.float64)
    torch.distributed.all_reduce(sums, op=torch.distributed.ReduceOp.SUM, group=dp.group)
    return metrics_from_sums(sums, max_rul)


def gather_shards(values, loader, dp):
    """The full vector of a sample-sharded test set on every rank: each rank writes its shard into a zero vector at its offsets and the
    vectors are summed (exact: every element is one value plus zeros; works on every backend, unlike all_gather of ragged CUDA
    tensors under gloo).  Only issued when a best row is saved."""
    v = values if torch.is_tensor(values) else torch.as_tensor(np.asarray(values))
    full = torch.zeros(loader.global_n, dtype=v.dtype, device=v.device)
    lo, hi = loader.shard
    full[lo:hi] = v.reshape(-1)
    torch.distributed.all_reduce(full, op=torch.distributed.ReduceOp.SUM, group=dp.group)
    return full


class _NullLogger:
    def debug(self, *a, **k):
        pass
