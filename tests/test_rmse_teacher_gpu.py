"""-m gpu: the RMSE half of BASELINE.json's metric ("training samples/sec + RMSE"), gated.  SURVEY.md section 8(d)'s synthetic task -- a
fixed random teacher ST_GCN labels uniform C-MAPSS-shaped windows -- trained through the HIP path (ST_GCN.update: the matrix-core chain)
and through the torch-CPU restatement of the reference's update (oracle/stgcn_torch_cpu.py, pinned to the reference's own training curve)
from the same initial weights on the same batches; scored with the reference's formula (utils.py:148-151: RMSE x max_rul 125).
North star: RMSE within 1e-3 of the reference.  bench.py reports the same quantity at ~49 k windows in its JSON line."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_teacher_task_rmse_within_1e_3_of_the_torch_cpu_restatement():
    import bench
    r = bench.rmse_teacher_task(torch.device("cuda:0"), epochs=2, n_train=16384, n_test=4096, batch=2048, checkpoints=(4, 16))
    assert r["steps"] == 16 and [m["steps"] for m in r["after_steps"]] == [4, 16] and r["within_1e-3"]
    assert r["abs_diff"] <= 1e-3, r
    assert r["rmse_hip"] < r["rmse_of_predicting_the_mean"], r              # the student did learn something in sixteen steps
    assert abs(r["final_train_loss_hip"] - r["final_train_loss_torch_cpu"]) <= 1e-4 * abs(r["final_train_loss_torch_cpu"]) + 1e-7, r


def test_teacher_task_rmse_with_dropout_on_under_the_same_masks():
    """The same experiment with dropout 0.2 (the reference's protocol trains with dropout): both paths under the SAME masks -- the HIP
    path's counter hash imposed on the torch-CPU restatement (tests/test_torch_cpu_baseline.py pins that form to the fp64 oracle)."""
    import bench
    r = bench.rmse_teacher_task(torch.device("cuda:0"), epochs=2, n_train=16384, n_test=4096, batch=2048, checkpoints=(4, 16), dropout=0.2)
    assert "dropout 0.2" in r["task"] and r["steps"] == 16 and r["within_1e-3"]
    assert r["abs_diff"] <= 1e-3, r
    assert abs(r["final_train_loss_hip"] - r["final_train_loss_torch_cpu"]) <= 1e-4 * abs(r["final_train_loss_torch_cpu"]) + 1e-7, r


def test_bn_free_family_teacher_task_tracks_the_torch_cpu_restatement():
    """STMSGCN (no BatchNorm, no dropout) at the reference's PHM2012 wiring and protocol batch: HIP path vs the torch-CPU restatement of the
    reference's update from the same weights on the same batches; the held-out RMSE stays within 1e-3 relative at every checkpoint."""
    import bench
    r = bench.rmse_teacher_task_stmsgcn(torch.device("cuda:0"), n_train=800, n_test=200, epochs=3, checkpoints=(8, 24))
    assert r["steps"] == 24 and len(r["after_steps"]) == 2
    assert r["within_1e-3_relative"], r
