"""RUL metrics with the reference's formulas (utils.py:136-201).  ``_calc_metrics`` returns (Score_v1, Score_v2, MAE, RMSE)
like utils.py:191-201.  Host arrays go through the vectorised numpy forms below (instead of per-sample Python loops); CUDA
tensors go through ``device_metrics`` -- one reduction kernel over the predictions where the eval forward left them
(``rulgnn_rul_metrics_f32``), 32 bytes to the host."""
from __future__ import annotations

import math

import numpy as np


def scoring_function(predicted, real, max_rul):
    """utils.py:136-146: late predictions (pred >= real) cost exp(d/10)-1, early ones exp(d/13)-1."""
    predicted, real = np.asarray(predicted, np.float64), np.asarray(real, np.float64)
    late = real <= predicted
    e = np.where(late, np.exp((predicted - real) * max_rul / 10.0) - 1.0, np.exp((real - predicted) * max_rul / 13.0) - 1.0)
    score = float(e.sum())
    return score, score / predicted.shape[0]


def scoring_function_v2(predicted, real):
    """utils.py:157-169."""
    predicted, real = np.asarray(predicted, np.float64), np.asarray(real, np.float64)
    err = (real - predicted) / (real + 1e-8) * 100.0
    e = np.where(err <= 0, np.exp(-math.log(0.5) * (err / 5.0)), np.exp(math.log(0.5) * (err / 20.0)))
    return float(e.mean())


def rmse_value(predicted, real, max_rul):
    """utils.py:148-151: sqrt(mean squared error) * max_rul."""
    predicted, real = np.asarray(predicted, np.float64), np.asarray(real, np.float64)
    return math.sqrt(float(np.mean((real - predicted) ** 2))) * max_rul


def mae_value(predicted, real, max_rul):
    predicted, real = np.asarray(predicted, np.float64), np.asarray(real, np.float64)
    return float(np.mean(np.abs(real - predicted))) * max_rul


def _calc_metrics(pred_labels, true_labels, max_rul):
    if getattr(pred_labels, "is_cuda", False):          # predictions still on the GPU: reduce them there
        return device_metrics(pred_labels, true_labels, max_rul)
    pred_labels, true_labels = np.array(pred_labels), np.array(true_labels)
    score_v1, _ = scoring_function(pred_labels, true_labels, max_rul)
    return score_v1, scoring_function_v2(pred_labels, true_labels), mae_value(pred_labels, true_labels, max_rul), \
        rmse_value(pred_labels, true_labels, max_rul)


def _device_reduce(pred, real, max_rul, entry):
    import ctypes as C

    import torch

    from . import _lib
    if not (pred.is_cuda and real.is_cuda):
        raise RuntimeError("device_metrics needs CUDA tensors; use _calc_metrics for host arrays")
    lib = _lib.load()
    p = pred.detach().reshape(-1).float().contiguous()
    r = real.detach().reshape(-1).float().contiguous()
    if p.numel() != r.numel() or p.numel() < 1:
        raise RuntimeError(f"device_metrics: {p.numel()} predictions vs {r.numel()} labels")
    n = p.numel()
    ws = torch.empty(int(lib.rulgnn_rul_metrics_workspace_bytes(n)) // 8, dtype=torch.float64, device=p.device)
    out = torch.empty(4, dtype=torch.float64, device=p.device)
    st = C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream)
    _lib.check(getattr(lib, entry)(p.data_ptr(), r.data_ptr(), n, float(max_rul), out.data_ptr(), ws.data_ptr(), ws.numel() * 8, st), entry)
    return out


def device_metrics(pred, real, max_rul):
    """(Score_v1, Score_v2, MAE, RMSE) of two float32 CUDA tensors, computed on the device in fp64.
    Raises if the HIP library is missing: there is no host fallback behind this entry."""
    return tuple(float(v) for v in _device_reduce(pred, real, max_rul, "rulgnn_rul_metrics_f32").cpu())


def device_metric_sums(pred, real, max_rul):
    """This rank's contribution to the metrics of a test set that is SHARDED over the ranks: a float64 device tensor
    ``[sum Score_v1 terms, sum Score_v2 terms, sum |d|, sum d^2, n]`` (``rulgnn_rul_metric_sums_f32``); zeros for an empty shard.
    SUM it over the ranks (one 5-double all-reduce) and close with ``metrics_from_sums``."""
    import torch
    out = torch.zeros(5, dtype=torch.float64, device=pred.device)
    if pred.numel():
        out[:4] = _device_reduce(pred, real, max_rul, "rulgnn_rul_metric_sums_f32")
        out[4] = float(pred.numel())
    return out


def metrics_from_sums(sums, max_rul):
    """(Score_v1, Score_v2, MAE, RMSE) from the (all-reduced) five sums: utils.py:136-169 closes with exactly these divisions."""
    s1, s2, sa, sq, n = (float(v) for v in (sums.cpu() if hasattr(sums, "cpu") else sums))
    if n < 1:
        raise RuntimeError("metrics of an empty test set")
    return s1, s2 / n, sa / n * max_rul, math.sqrt(sq / n) * max_rul
