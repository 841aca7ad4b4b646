"""Pin the numpy oracle (oracle/stgcn_oracle.py) to outputs of the reference itself.

The fixtures were produced by tests/golden/make_golden.py importing the reference's
models/ST_GCN/Model.py and algorithms/algorithms.py in the build container."""
import glob
import os

import numpy as np
import pytest

from oracle import stgcn_oracle as O

from conftest import GOLDEN

FB_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "stgcn_*x*_bs*.npz"))
                  if "train_curve" not in p and "layers" not in p and "order" not in p)
ORDER_CASES = ["stgcn_order2_14x30_bs19", "stgcn_order3_14x30_bs10", "stgcn_order2_40x64_bs5", "stgcn_order3_9x21_layers3_bs7"]


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd:")}
    return z, sd


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def test_cases_present():
    assert len(FB_CASES) >= 6


@pytest.mark.parametrize("name", FB_CASES)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_eval_forward_matches_reference(name, dtype):
    z, sd = load_case(name)
    N, P = int(z["num_patch"]), int(z["patch_size"])
    fc = O.forward(sd, z["x"].astype(dtype), N, P, train=False)
    tol = 2e-6 if dtype is np.float64 else 2e-5
    nan_ref = np.isnan(z["eval_pred"])
    assert np.array_equal(np.isnan(fc.pred), nan_ref)           # NaN in the same places
    ok = ~nan_ref[:, 0]
    assert ok.any()
    assert rel_err(fc.feat[ok], z["feat"][ok]) < tol
    assert rel_err(fc.adj[ok], z["adj"][ok]) < 5 * tol
    for l, lc in enumerate(fc.layers):
        nxt = fc.layers[l + 1].X if l + 1 < len(fc.layers) else fc.X_out
        assert rel_err(nxt[ok], z[f"eval_layer{l}"][ok]) < 5 * tol
    assert rel_err(fc.pred[ok], z["eval_pred"][ok]) < 5 * tol


@pytest.mark.parametrize("name", [n for n in FB_CASES if "nan" not in n])
def test_train_forward_backward_matches_reference_autograd(name):
    z, sd = load_case(name)
    N, P = int(z["num_patch"]), int(z["patch_size"])
    x = z["x"].astype(np.float64)
    fc = O.forward(sd, x, N, P, train=True, dropout=0.0)
    assert rel_err(fc.pred, z["train_pred"]) < 1e-5
    loss, dpred = O.mse_loss_and_grad(fc.pred, z["y"].astype(np.float64))
    assert abs(loss - float(z["train_loss"])) < 1e-5 * abs(float(z["train_loss"]))
    g = O.backward(sd, fc, dpred)
    names = O.live_param_names(2)
    assert sorted(names) == sorted(k[5:] for k in z.files if k.startswith("grad:"))
    for n in names:
        ref = z["grad:" + n]
        got = g[n].reshape(ref.shape)
        assert rel_err(got, ref) < 2e-4, n       # reference grads are fp32 autograd
    new = O.bn_running_update(sd, fc, 2)
    for k, v in new.items():
        assert rel_err(v, z["sd_after:" + k]) < 1e-5, k


@pytest.mark.parametrize("name,L", [("stgcn_layers1_14x30_bs21", 1), ("stgcn_layers3_14x30_bs21", 3)])
def test_other_layer_counts_match_reference(name, L):
    z, sd = load_case(name)
    x = z["x"].astype(np.float64)
    fc = O.forward(sd, x, 14, 30, L, train=False)
    assert rel_err(fc.pred, z["eval_pred"]) < 1e-5
    fc = O.forward(sd, x, 14, 30, L, train=True)
    assert rel_err(fc.pred, z["train_pred"]) < 1e-5
    loss, dpred = O.mse_loss_and_grad(fc.pred, z["y"].astype(np.float64))
    assert abs(loss - float(z["train_loss"])) < 1e-5 * abs(float(z["train_loss"]))
    g = O.backward(sd, fc, dpred)
    for n in O.live_param_names(L):
        assert rel_err(g[n].reshape(z["grad:" + n].shape), z["grad:" + n]) < 2e-4, n


@pytest.mark.parametrize("name", ORDER_CASES)
def test_mpnn_order_above_one_matches_reference(name):
    """MPNN_mk with k = 2, 3 (Model.py:81-90: theta_kk on A^(kk+1) X, summed): fixtures of tests/golden/make_golden_order.py."""
    z, sd = load_case(name)
    N, P, L, K = int(z["num_patch"]), int(z["patch_size"]), int(z["num_layers"]), int(z["k"])
    assert O.mpnn_order(sd) == K
    # the flat order of the live parameters is the reference's named_parameters() order (the dead net0 / net1 branches dropped)
    live = [n for n in z["key_order"].tolist() if ".net0." not in n and ".net1." not in n]
    assert live == O.live_param_names(L, K)
    x = z["x"].astype(np.float64)
    fc = O.forward(sd, x, N, P, L, train=False)
    assert rel_err(fc.pred, z["eval_pred"]) < 2e-5
    fc = O.forward(sd, x, N, P, L, train=True)
    assert rel_err(fc.pred, z["train_pred"]) < 2e-5
    loss, dpred = O.mse_loss_and_grad(fc.pred, z["y"].astype(np.float64))
    assert abs(loss - float(z["train_loss"])) < 2e-5 * abs(float(z["train_loss"]))
    g = O.backward(sd, fc, dpred)
    assert sorted(O.live_param_names(L, K)) == sorted(k[5:] for k in z.files if k.startswith("grad:"))
    for n in O.live_param_names(L, K):
        assert rel_err(g[n].reshape(z["grad:" + n].shape), z["grad:" + n]) < 3e-4, n
    for k, v in O.bn_running_update(sd, fc, L).items():
        assert rel_err(v, z["sd_after:" + k]) < 1e-5, k


def test_mpnn_order_two_training_curve_matches_reference_update():
    z = np.load(os.path.join(GOLDEN, "stgcn_order2_train_curve_14x30_bs16.npz"))
    N, P, steps = int(z["num_patch"]), int(z["patch_size"]), int(z["steps"])
    prm = {k[len("sd0:model."):]: z[k].astype(np.float64) for k in z.files if k.startswith("sd0:model.")}
    assert O.mpnn_order(prm) == 2
    opt = {"step": 0, "m": {}, "v": {}}
    losses = []
    for s in range(steps):
        loss, prm, opt, _, _ = O.train_step(prm, opt, z["xs"][s].astype(np.float64), z["ys"][s].astype(np.float64),
                                            N, P, lr=float(z["lr"]), weight_decay=float(z["wd"]))
        losses.append(loss)
    ref = z["losses"]
    assert np.max(np.abs(np.array(losses) - ref) / ref) < 2e-3
    assert np.max(np.abs(np.array(losses[:3]) - ref[:3]) / ref[:3]) < 5e-5
    for n in O.live_param_names(2, 2):
        assert rel_err(prm[n], z["sdK:model." + n]) < 5e-3, n
    fc = O.forward(prm, z["xs"][0].astype(np.float64), N, P, train=False)
    assert rel_err(fc.pred, z["eval_pred_after"]) < 5e-3


def test_training_curve_matches_reference_update():
    z = np.load(os.path.join(GOLDEN, "stgcn_train_curve_14x30_bs32.npz"))
    N, P, K = int(z["num_patch"]), int(z["patch_size"]), int(z["steps"])
    prm = {k[len("sd0:model."):]: z[k].astype(np.float64) for k in z.files if k.startswith("sd0:model.")}
    opt = {"step": 0, "m": {}, "v": {}}
    losses = []
    for s in range(K):
        loss, prm, opt, _, _ = O.train_step(prm, opt, z["xs"][s].astype(np.float64), z["ys"][s].astype(np.float64),
                                            N, P, lr=float(z["lr"]), weight_decay=float(z["wd"]))
        losses.append(loss)
    ref = z["losses"]
    # fp64 oracle vs fp32 reference over 24 Adam steps
    assert np.max(np.abs(np.array(losses) - ref) / ref) < 2e-3
    assert np.max(np.abs(np.array(losses[:4]) - ref[:4]) / ref[:4]) < 2e-5
    for n in O.live_param_names(2):
        assert rel_err(prm[n], z["sdK:model." + n]) < 5e-3, n
    fc = O.forward(prm, z["xs"][0].astype(np.float64), N, P, train=False)
    assert rel_err(fc.pred, z["eval_pred_after"]) < 5e-3


def test_metrics_match_reference_utils():
    z = np.load(os.path.join(GOLDEN, "metrics_case.npz"))
    mr = float(z["max_rul"])
    assert abs(O.rmse_value(z["pred"], z["real"], mr) - float(z["rmse"])) < 1e-9 * float(z["rmse"])
    assert abs(O.mae_value(z["pred"], z["real"], mr) - float(z["mae"])) < 1e-9 * float(z["mae"])
    assert abs(O.score_v1(z["pred"], z["real"], mr) - float(z["score_v1"])) < 1e-9 * float(z["score_v1"])
    assert abs(O.score_v2(z["pred"], z["real"]) - float(z["score_v2"])) < 1e-9 * float(z["score_v2"])


def test_dropout_mask_statistics_and_determinism():
    key = O.dropout_layer_key(1234, 7, 1)
    m1 = O.dropout_keep_mask(4096, 14, key, 0.2)
    m2 = O.dropout_keep_mask(4096, 14, key, 0.2)
    assert np.array_equal(m1, m2)
    assert abs(m1.mean() - 0.8) < 0.005
    # sample_offset shifts the stream: shard [1024:2048] of a global batch == offset 1024
    m3 = O.dropout_keep_mask(1024, 14, key, 0.2, sample_offset=1024)
    assert np.array_equal(m3, m1[1024:2048])
    assert O.dropout_layer_key(1234, 7, 1) != O.dropout_layer_key(1234, 8, 1)
