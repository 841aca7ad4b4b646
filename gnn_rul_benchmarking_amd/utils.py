"""Harness helpers of the trainer mirror: run seeding, the running loss meter, the per-run log file and the
end-of-run checkpoint.

What is pinned by the reference harness (and by tests/golden/*_reference_run.npz) is BEHAVIOUR, not code:
  * seeding order python -> numpy -> torch (-> torch.cuda), deterministic kernels            (reference utils.py:63-69)
  * the meter is a sample-weighted running mean that the trainer never resets per epoch      (utils.py:44-60, trainer.py:101)
  * log location <exp_log_dir>/<METHOD>_run_<id>/logs_<dd_mm_YYYY_HH_MM_SS>.log and its header (utils.py:91-108)
  * checkpoint.pt = {"configs", "hparams", "model_dict"} with 'model.'-prefixed keys          (utils.py:111-120)
"""
from __future__ import annotations

import logging
import os
import random
import sys
import time
from dataclasses import dataclass

import numpy as np
import torch

_DATASETS_WITH_SUBSETS = frozenset({"CMAPSS", "PHM2012", "XJTU_SY"})
_RULE = "=" * 45


@dataclass
class AverageMeter:
    """Sample-weighted running mean; `val` is the last value fed, `avg` the mean so far."""
    val: float = 0.0
    sum: float = 0.0
    count: float = 0.0

    @property
    def avg(self) -> float:
        return self.sum / self.count if self.count else 0.0

    def reset(self) -> None:
        self.val = self.sum = self.count = 0.0

    def update(self, val, n=1) -> None:
        self.val = val
        self.sum += val * n
        self.count += n


def fix_randomness(seed: int) -> None:
    """Seed every generator a run draws from and force deterministic backend kernels."""
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    torch.backends.cudnn.benchmark = False
    torch.backends.cudnn.deterministic = True


def _open_run_log(path: str, echo: bool) -> logging.Logger:
    """A logger named after its file, writing bare messages to the file (append) and optionally to stdout."""
    log = logging.getLogger(path)
    log.setLevel(logging.DEBUG)
    sinks = [logging.FileHandler(path, mode="a")]
    if echo:
        sinks.insert(0, logging.StreamHandler(sys.stdout))
    plain = logging.Formatter("%(message)s")
    for sink in sinks:
        sink.setFormatter(plain)
        log.addHandler(sink)
    return log


def starting_logs(data_type, GNN_method, exp_log_dir, dataset_id, bearing_id, run_id, to_stdout=True):
    """Create the run directory and its log file, write the header block, return (logger, run directory)."""
    run_dir = os.path.join(exp_log_dir, f"{GNN_method}_run_{run_id}")
    os.makedirs(run_dir, exist_ok=True)
    stamp = time.strftime("%d_%m_%Y_%H_%M_%S")
    log = _open_run_log(os.path.join(run_dir, f"logs_{stamp}.log"), to_stdout)

    header = [_RULE, f"Dataset: {data_type}"]
    if data_type in _DATASETS_WITH_SUBSETS:
        header.append(f"Sub-dataset ID:  {dataset_id}")
        if data_type == "XJTU_SY":
            header.append(f"Bearing ID:  {bearing_id}")
    header += [f"Method:  {GNN_method}", _RULE, f"Run ID: {run_id}", _RULE]
    for line in header:
        log.debug(line)
    return log, run_dir


def save_checkpoint(home_path, algorithm, dataset_configs, log_dir, hparams):
    """Write <home_path>/<log_dir>/checkpoint.pt with the dataset config, the hparams and the algorithm's state on the CPU."""
    weights = {name: tensor.detach().cpu() for name, tensor in algorithm.state_dict().items()}
    torch.save({"configs": vars(dataset_configs), "hparams": dict(hparams), "model_dict": weights},
               os.path.join(home_path, log_dir, "checkpoint.pt"))
