"""Harness helpers with the reference's behaviour (utils.py:44-120): seeding, running-average meter,
per-run logger, end-of-run checkpoint."""
from __future__ import annotations

import logging
import os
import random
import sys
from datetime import datetime

import numpy as np
import torch


class AverageMeter(object):
    """Computes and stores the average and current value (utils.py:44-60)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def fix_randomness(SEED):
    """utils.py:63-69."""
    random.seed(SEED)
    np.random.seed(SEED)
    torch.manual_seed(SEED)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(SEED)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = False


def _logger(logger_name, level=logging.DEBUG, to_stdout=True):
    logger = logging.getLogger(logger_name)
    logger.setLevel(level)
    fmt = logging.Formatter("%(message)s")
    if to_stdout:
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(fmt)
        logger.addHandler(h)
    fh = logging.FileHandler(logger_name, mode='a')
    fh.setFormatter(fmt)
    logger.addHandler(fh)
    return logger


def starting_logs(data_type, GNN_method, exp_log_dir, dataset_id, bearing_id, run_id, to_stdout=True):
    """utils.py:91-108: <exp_log_dir>/<METHOD>_run_<id>/logs_<timestamp>.log with the same header lines."""
    log_dir = os.path.join(exp_log_dir, GNN_method + "_run_" + str(run_id))
    os.makedirs(log_dir, exist_ok=True)
    log_file_name = os.path.join(log_dir, f"logs_{datetime.now().strftime('%d_%m_%Y_%H_%M_%S')}.log")
    logger = _logger(log_file_name, to_stdout=to_stdout)
    logger.debug("=" * 45)
    logger.debug(f'Dataset: {data_type}')
    if data_type in ('CMAPSS', 'PHM2012', 'XJTU_SY'):
        logger.debug(f'Sub-dataset ID:  {dataset_id}')
        if data_type == 'XJTU_SY':
            logger.debug(f'Bearing ID:  {bearing_id}')
    logger.debug(f'Method:  {GNN_method}')
    logger.debug("=" * 45)
    logger.debug(f'Run ID: {run_id}')
    logger.debug("=" * 45)
    return logger, log_dir


def save_checkpoint(home_path, algorithm, dataset_configs, log_dir, hparams):
    """utils.py:111-120: {configs, hparams, model_dict} -> <log_dir>/checkpoint.pt (keys 'model.<...>')."""
    save_dict = {"configs": dataset_configs.__dict__, "hparams": dict(hparams),
                 "model_dict": {k: v.detach().cpu() for k, v in algorithm.state_dict().items()}}
    torch.save(save_dict, os.path.join(home_path, log_dir, "checkpoint.pt"))
