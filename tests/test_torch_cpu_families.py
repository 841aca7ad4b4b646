"""Pins oracle/families_torch_cpu.py -- the torch-CPU restatements bench.py times as ``families.*.cpu_baseline`` -- to the fixtures the
reference itself produced (tests/golden/make_golden_*.py imported /root/reference): eval forwards, train-mode losses, and the loss
curves of the reference's own ``update()`` (Adam included).  Same ATen kernels in the same order as the reference, so the gates are
fp32 round-off: 1e-6 relative on forwards and single losses, 2e-5 at the end of a 12-16 step training curve."""
import os

import numpy as np
import pytest
import torch

from oracle import families_torch_cpu as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=True)


def sd(z, prefix="sd:"):
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


def cfg_of(z):
    out = {}
    for k in z.files:
        if k.startswith("cfg:"):
            v = z[k]
            out[k[4:]] = [int(t) for t in v] if v.ndim else (int(v) if float(v) == int(v) else float(v))
    return out


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(autouse=True)
def _one_thread():
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(prev)


# ---- ASTGCNN ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["astgcnn_cmapss_14x50_bs16", "astgcnn_ncmapss_20x50_bs6", "astgcnn_small_5x12_bs9", "astgcnn_k2_7x20_bs4"])
def test_astgcnn_forward_and_loss(name):
    z = load(name)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    st = T.AstgcnnState(sd(z))
    with torch.no_grad():
        assert rel(T.astgcnn_forward(st, x, False).numpy(), z["eval_pred"]) < 1e-6
    st = T.AstgcnnState(sd(z))
    pred = T.astgcnn_forward(st, x, True)
    assert rel(pred.detach().numpy(), z["train_pred"]) < 1e-6
    loss = torch.nn.functional.mse_loss(pred, y)
    assert abs(float(loss) - float(z["train_loss"])) < 1e-6 * abs(float(z["train_loss"]))
    loss.backward()
    for k, p in st.p.items():
        assert rel(p.grad.numpy(), z["grad:" + k]) < 1e-5, k
    for k, v in st.buf.items():
        assert rel(v.numpy(), z["sd_after:" + k]) < 1e-6, k


def test_astgcnn_update_curve():
    z = load("astgcnn_train_curve_14x50_bs20")
    st = T.AstgcnnState(sd(z, "sd0:"), lr=float(z["lr"]), weight_decay=float(z["wd"]))
    got = [T.astgcnn_update(st, torch.from_numpy(x), torch.from_numpy(y)) for x, y in zip(z["xs"], z["ys"])]
    assert np.allclose(got, z["losses"], rtol=2e-5, atol=0)


# ---- FC_STGNN -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["fcstgnn_fd001_bs5", "fcstgnn_fd002_bs3", "fcstgnn_fd003like_6p_bs2", "fcstgnn_fd004_bs6", "fcstgnn_ncmapss_bs3"])
def test_fcstgnn_forward_and_loss(name):
    z = load(name)
    cfg = cfg_of(z)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    st = T.FcstgnnState(sd(z), cfg)
    assert rel(st.pe[0, :4].numpy(), z["pe_head"]) < 1e-6
    with torch.no_grad():
        assert rel(T.fcstgnn_forward(st, x, False).numpy(), z["eval_pred"]) < 1e-6
    st = T.FcstgnnState(sd(z), cfg)
    pred = T.fcstgnn_forward(st, x, True, dropout=0.0)          # the fixtures' train mode runs with the positional dropout off
    assert rel(pred.detach().numpy(), z["train_pred"]) < 2e-6
    loss = torch.nn.functional.mse_loss(pred, y)
    assert abs(float(loss) - float(z["train_loss"])) < 2e-6 * abs(float(z["train_loss"]))


def test_fcstgnn_update_curve():
    z = load("fcstgnn_train_curve_fd004_bs10")
    st = T.FcstgnnState(sd(z, "sd0:"), cfg_of(z), lr=float(z["lr"]), weight_decay=float(z["wd"]))
    got = [T.fcstgnn_update(st, torch.from_numpy(x), torch.from_numpy(y), dropout=0.0) for x, y in zip(z["xs"], z["ys"])]
    assert np.allclose(got, z["losses"], rtol=2e-5, atol=0)


# ---- STMSGCN ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["stmsgcn_phm1_12x16_bs5", "stmsgcn_phm2_9x20_bs4", "stmsgcn_dims_7x32_bs4"])
def test_stmsgcn_forward_and_loss(name):
    z = load(name)
    cfg = cfg_of(z)
    st = T.StmsgcnState(sd(z), cfg)
    pred = T.stmsgcn_forward(st, torch.from_numpy(z["x"]))
    assert rel(pred.detach().numpy(), z["pred"]) < 1e-6
    loss = torch.nn.functional.mse_loss(pred, torch.from_numpy(z["y"]))
    assert abs(float(loss) - float(z["loss"])) < 1e-6 * abs(float(z["loss"]))
    loss.backward()
    for k, p in st.p.items():
        assert rel(p.grad.numpy(), z["grad:" + k]) < 1e-5, k


def test_stmsgcn_update_curve():
    z = load("stmsgcn_train_curve_9x20_bs6")
    st = T.StmsgcnState(sd(z, "sd0:"), cfg_of(z), lr=float(z["lr"]), weight_decay=float(z["wd"]))
    got = [T.stmsgcn_update(st, torch.from_numpy(x), torch.from_numpy(y)) for x, y in zip(z["xs"], z["ys"])]
    assert np.allclose(got, z["losses"], rtol=2e-5, atol=0)
    end = sd(z, "sd_end:")
    for k, p in st.p.items():
        assert rel(p.detach().numpy(), end["model." + k]) < 2e-5, k


# ---- HAGCN --------------------------------------------------------------------------------------------------------------------
def _hagcn_state(z):
    cfg = cfg_of(z)
    arrays = sd(z)
    has_td = any(k.startswith("TD.") for k in arrays)
    if not has_td:          # the LSTM stack is 320k of the 366k weights: most fixtures carry the node features behind it instead
        arrays.update({k: v for k, v in T.hagcn_random_arrays(cfg).items() if k.startswith("TD.")})
    return T.HagcnState(arrays, cfg, alpha=float(z["alpha"])), has_td


@pytest.mark.parametrize("name", ["hagcn_fd001_5x10_bs6", "hagcn_fd002_2x25_bs5", "hagcn_fd004_1x50_bs7", "hagcn_ncmapss_2x25_bs3", "hagcn_smalllstm_3x6_bs4"])
def test_hagcn_forward_selection_and_loss(name):
    z = load(name)
    st, has_td = _hagcn_state(z)
    x, y = torch.from_numpy(z["x"]), torch.from_numpy(z["y"])
    bs = x.shape[0]
    if has_td:
        with torch.no_grad():
            assert rel(T.hagcn_forward(st, x, False).numpy(), z["eval_pred"]) < 1e-6
        nodes = T.hagcn_nodes(st, x, True, dropout=0.0)          # the fixtures' train-mode pass runs with the LSTM dropouts off
        assert rel(nodes.detach().numpy(), z["nodes"]) < 1e-6
    else:
        nodes = torch.from_numpy(z["nodes"])
        with torch.no_grad():
            assert rel(T.hagcn_from_nodes(st, nodes, bs, False).numpy(), z["eval_pred"]) < 1e-6
    pred, kl = T.hagcn_from_nodes(st, nodes, bs, True)
    assert rel(pred.detach().numpy(), z["train_pred"]) < 2e-6
    assert abs(float(kl) - float(z["train_kl"])) < 2e-6 * abs(float(z["train_kl"]))
    loss = torch.nn.functional.mse_loss(pred, y) + st.alpha * kl
    assert abs(float(loss) - float(z["train_loss"])) < 2e-6 * abs(float(z["train_loss"]))
    loss.backward()
    for k, p in st.p.items():
        if ("grad:" + k) in z.files and (has_td or not k.startswith("TD.")):
            assert rel(p.grad.numpy(), z["grad:" + k]) < 2e-5, k


def test_time_update_runs_every_family():
    """The timing entry bench.py calls, on tiny budgets: a full update (Adam included) of each family on its reference wiring."""
    from gnn_rul_benchmarking_amd import hparams as HP
    for family, ds, did, batch in (("ASTGCNN", "NCMAPSS", "DS02", 4), ("FC_STGNN", "CMAPSS", "FD004", 3), ("HAGCN", "CMAPSS", "FD004", 2)):
        cfg = dict(HP.get_hparams_class(ds)(did).alg_hparams[family])
        r = T.time_update(family, cfg, batch, 1, warmup=1, iters=2, budget_s=20.0)
        assert r["iterations"] >= 1 and r["samples_per_s"] > 0
    cfg = dict(num_patch=9, patch_size=20, interval=2, band_width=3, gcn_dims=[16, 64, 16, 1], gru_hidden_dim=8)
    r = T.time_update("STMSGCN", cfg, 2, 1, warmup=1, iters=2, budget_s=20.0)
    assert r["iterations"] >= 1
