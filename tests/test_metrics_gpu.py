"""Device-side RUL metrics (rulgnn_rul_metrics_f32) against the reference formulas restated as the
per-sample loops of utils.py:136-169 (fp64), and against the package's vectorised numpy form."""
import math

import numpy as np
import pytest
import torch

from gnn_rul_benchmarking_amd import metrics as M

pytestmark = pytest.mark.gpu


def reference_loops(pred, real, max_rul):
    """utils.py:136-146, 148-155, 157-169, written out sample by sample."""
    s1 = s2 = 0.0
    for p, r in zip(pred, real):
        if r > p:
            s1 += math.exp((r * max_rul - p * max_rul) / 13) - 1
        else:
            s1 += math.exp((p * max_rul - r * max_rul) / 10) - 1
        err = ((r - p) / (r + 1e-8)) * 100
        s2 += math.exp(-math.log(0.5) * (err / 5)) if err <= 0 else math.exp(math.log(0.5) * (err / 20))
    n = len(pred)
    d = np.asarray(real, np.float64) - np.asarray(pred, np.float64)
    return s1, s2 / n, float(np.mean(np.abs(d))) * max_rul, math.sqrt(float(np.mean(d * d))) * max_rul


@pytest.mark.parametrize("n,max_rul", [(1, 125.0), (100, 125.0), (257, 130.0), (10000, 1.0), (300001, 125.0)])
def test_device_metrics_match_reference_formulas(n, max_rul):
    g = torch.Generator().manual_seed(n)
    real = torch.rand(n, generator=g)
    pred = (real + 0.05 * torch.randn(n, generator=g)).clamp_min(0.0)
    if n > 3:
        pred[1] = real[1]                       # exact hit: the "late" branch of both scores (real <= pred, err <= 0)
        real[2] = 0.0                           # zero label: err = -pred / 1e-8 * 100 (huge negative -> Score_v2 term overflows like the reference)
        pred[2] = 0.0
    got = M.device_metrics(pred.cuda(), real.cuda(), max_rul)
    p64, r64 = pred.numpy().astype(np.float64), real.numpy().astype(np.float64)
    want_np = M._calc_metrics(p64, r64, max_rul)
    assert np.allclose(got, want_np, rtol=1e-11, atol=0)
    if n <= 10000:
        want = reference_loops(p64, r64, max_rul)
        assert np.allclose(got, want, rtol=1e-10, atol=0)


def test_device_metrics_are_deterministic_and_dispatch():
    g = torch.Generator().manual_seed(7)
    real, pred = torch.rand(50000, generator=g).cuda(), torch.rand(50000, generator=g).cuda()
    a = M.device_metrics(pred, real, 125.0)
    b = M._calc_metrics(pred, real, 125.0)       # CUDA tensors are routed to the device kernel
    assert a == b
    with pytest.raises(RuntimeError):
        M.device_metrics(pred.cpu(), real.cpu(), 125.0)
    with pytest.raises(RuntimeError):
        M.device_metrics(pred[:10], real[:11], 125.0)


def test_device_metrics_against_the_reference_generated_fixture():
    """tests/golden/metrics_case.npz holds the outputs of the reference's own utils.py:136-201 (tests/golden/make_golden.py imported it):
    the package's numpy forms reproduce them on the fixture's fp64 inputs to 1e-12; the device kernel, which takes the fp32 predictions
    the eval forward wrote, agrees to the rounding of those inputs (Score_v1 is exp(12.5 d): 6e-8 in d is ~1e-6 in a term)."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "metrics_case.npz"))
    pred, real, mr = z["pred"], z["real"], float(z["max_rul"])
    want = (float(z["score_v1"]), float(z["score_v2"]), float(z["mae"]), float(z["rmse"]))
    assert np.allclose(M._calc_metrics(pred, real, mr), want, rtol=1e-12, atol=0)
    p32, r32 = torch.from_numpy(pred.astype(np.float32)).cuda(), torch.from_numpy(real.astype(np.float32)).cuda()
    got = M.device_metrics(p32, r32, mr)
    assert np.allclose(got, want, rtol=2e-5, atol=0)
    # and exactly the reference formulas on the SAME fp32-rounded inputs
    assert np.allclose(got, M._calc_metrics(pred.astype(np.float32).astype(np.float64), real.astype(np.float32).astype(np.float64), mr),
                       rtol=1e-11, atol=0)


@pytest.mark.parametrize("n,cut", [(257, 100), (5000, 1), (64, 64)])
def test_metric_sums_of_two_shards_close_to_the_whole_set_metrics(n, cut):
    """rulgnn_rul_metric_sums_f32: the sums of two contiguous shards (one may be empty), added, and closed with the reference's divisions
    equal the one-pass metrics of the whole set (what the sharded evaluation of trainer.py all-reduces)."""
    g = torch.Generator().manual_seed(n)
    real = torch.rand(n, generator=g).cuda()
    pred = (real + 0.05 * torch.randn(n, generator=g).cuda()).clamp_min(0.0)
    whole = M.device_metrics(pred, real, 125.0)
    sums = M.device_metric_sums(pred[:cut], real[:cut], 125.0) + M.device_metric_sums(pred[cut:], real[cut:], 125.0)
    assert float(sums[4]) == n
    assert np.allclose(M.metrics_from_sums(sums, 125.0), whole, rtol=1e-12, atol=0)
