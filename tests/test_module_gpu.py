"""-m gpu: the drop-in module / Algorithm surface on the real kernels."""
import os

import numpy as np
import pytest
import torch

from gnn_rul_benchmarking_amd.algorithms import ST_GCN
from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def load_into(model, z, prefix):
    sd = {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix)}
    model.load_state_dict(sd)


def test_eval_forward_through_module_matches_reference():
    z = np.load(os.path.join(GOLDEN, "stgcn_cmapss_14x30_bs32.npz"))
    m = ST_GCN_model(14, 30)
    load_into(m, z, "sd:")
    m = m.to(DEV).eval()
    with torch.no_grad():
        pred = m(torch.from_numpy(z["x"]).to(DEV))
    assert pred.shape == (32, 1)
    assert rel_err(pred.cpu().numpy(), z["eval_pred"]) < 1e-4
    # the bearing loaders hand [bs, 1, N*P]; the reference reshapes, so do we
    with torch.no_grad():
        pred2 = m(torch.from_numpy(z["x"]).reshape(32, 1, 420).to(DEV))
    assert torch.equal(pred, pred2)


def test_update_matches_reference_training_curve():
    """24 x ST_GCN.update from the reference's own initial state (algorithms/algorithms.py:481-490)."""
    z = np.load(os.path.join(GOLDEN, "stgcn_train_curve_14x30_bs32.npz"))
    algo = ST_GCN({"num_patch": 14, "patch_size": 30, "dropout": 1e-12},
                  {"learning_rate": float(z["lr"]), "weight_decay": float(z["wd"])}, DEV)
    load_into(algo, z, "sd0:")
    algo.to(DEV)
    algo.train()
    xs, ys = torch.from_numpy(z["xs"]).to(DEV), torch.from_numpy(z["ys"]).to(DEV)
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(int(z["steps"]))]
    ref = z["losses"]
    assert isinstance(losses[0], float)
    assert np.max(np.abs(np.array(losses[:4]) - ref[:4]) / ref[:4]) < 1e-4
    assert np.max(np.abs(np.array(losses) - ref) / ref) < 5e-3
    sd = algo.state_dict()
    for k in z.files:
        if k.startswith("sdK:") and ("running_" in k or k.endswith(("theta.0.weight", "fc1.weight", "conv_block1.0.weight"))):
            assert rel_err(sd[k[4:]].cpu().numpy(), z[k]) < 5e-3, k
    assert int(sd["model.sg_tcn.layers.0.1.conv_block1.2.num_batches_tracked"]) == 24
    # dead branches untouched (no gradient, no weight decay): bit-identical to the initial state
    for k in z.files:
        if k.startswith("sd0:") and (".net0." in k or ".net1." in k):
            assert np.array_equal(sd[k[4:]].cpu().numpy(), z[k]), k
    algo.eval()
    with torch.no_grad():
        pred = algo.model(xs[0])
    assert rel_err(pred.cpu().numpy(), z["eval_pred_after"]) < 5e-3


def test_autograd_path_equals_fused_path():
    z = np.load(os.path.join(GOLDEN, "stgcn_train_curve_14x30_bs32.npz"))
    hp = {"learning_rate": 1e-3, "weight_decay": 1e-4}
    cfg = {"num_patch": 14, "patch_size": 30, "dropout": 0.2}
    torch.manual_seed(5)
    a = ST_GCN(cfg, hp, DEV)
    torch.manual_seed(5)
    b = ST_GCN(cfg, hp, DEV)
    load_into(a, z, "sd0:"); load_into(b, z, "sd0:")
    a.to(DEV).train(); b.to(DEV).train()
    xs, ys = torch.from_numpy(z["xs"]).to(DEV), torch.from_numpy(z["ys"]).to(DEV)
    for s in range(6):
        la = a.update(xs[s], ys[s], 1)["loss"]
        lb = b.update_reference_style(xs[s], ys[s], 1)["loss"]
        assert abs(la - lb) < 2e-5 * abs(lb)
    assert rel_err(a.model.flat_params.cpu().numpy(), b.model.flat_params.cpu().numpy()) < 1e-4
    assert rel_err(a.model._bn.cpu().numpy(), b.model._bn.cpu().numpy()) < 1e-5


def test_sync_loss_false_returns_device_tensor_and_train_mode_is_required():
    algo = ST_GCN({"num_patch": 14, "patch_size": 30, "dropout": 0.2}, {"learning_rate": 1e-4, "weight_decay": 1e-4}, DEV)
    algo.to(DEV)
    x, y = torch.rand(64, 14, 30, device=DEV), torch.rand(64, 1, device=DEV)
    algo.eval()
    with pytest.raises(RuntimeError):
        algo.update(x, y, 1)
    algo.train()
    algo.sync_loss = False
    out = algo.update(x, y, 1)["loss"]
    assert torch.is_tensor(out) and out.is_cuda and out.dim() == 0
    assert torch.isfinite(out)


def test_phm2012_shape_trains_and_unsupported_shape_raises_clearly():
    """Reference-wired PHM2012 Condition_1 hparams (configs/hparams.py:238): [bs, 1, 2560] -> 40 patches of 64."""
    algo = ST_GCN({"num_patch": 40, "patch_size": 64, "dropout": 0.2}, {"learning_rate": 1e-3, "weight_decay": 1e-4}, DEV)
    algo.to(DEV).train()
    x, y = torch.rand(50, 1, 2560, device=DEV), torch.rand(50, 1, device=DEV)
    losses = [algo.update(x, y, 1)["loss"] for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    algo.eval()
    with torch.no_grad():
        assert algo.model(x).shape == (50, 1)
    with pytest.raises(RuntimeError):                 # MPNN order k = 4 (and k > 1 on the tiled shapes) is not implemented: loud, no fallback
        ST_GCN_model(14, 30, k=4).to(DEV).eval()(torch.rand(8, 14, 30, device=DEV))
    with pytest.raises(RuntimeError):
        ST_GCN_model(160, 16, k=2).to(DEV).eval()(torch.rand(8, 160, 16, device=DEV))


def test_xjtu_and_phm_c2_shapes_train_on_the_tiled_path():
    """Reference-wired bearing configs with num_patch > 64 (configs/hparams.py:271,349): tiled path."""
    for cfg, seq in (({"num_patch": 160, "patch_size": 16, "dropout": 0.2}, 2560), ({"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, 32768)):
        torch.manual_seed(1)
        algo = ST_GCN(cfg, {"learning_rate": 1e-4, "weight_decay": 1e-4}, DEV)
        algo.to(DEV).train()
        x, y = torch.rand(20, 1, seq, device=DEV), torch.rand(20, 1, device=DEV)
        losses = [algo.update(x, y, 1)["loss"] for _ in range(6)]
        assert all(np.isfinite(losses)), (cfg, losses)
        if cfg["num_patch"] == 160:          # (with 3M parameters and 20 random samples Adam's first steps are not monotone)
            assert losses[-1] < losses[0], (cfg, losses)
        algo.eval()
        with torch.no_grad():
            assert algo.model(x).shape == (20, 1)
        # the autograd path goes through the same kernels
        algo.train()
        l2 = algo.update_reference_style(x, y, 1)["loss"]
        assert np.isfinite(l2)


def test_autograd_backward_after_a_second_forward_raises_instead_of_using_overwritten_activations():
    """The saved activations live in one workspace per batch size: a second train-mode forward of the same size overwrites them.
    The module detects that at backward time (advisor finding, round 1) instead of silently returning wrong gradients."""
    import torch
    from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model
    torch.manual_seed(0)
    m = ST_GCN_model(14, 30, dropout=0.2).to("cuda:0").train()
    x1, x2 = torch.rand(8, 14, 30, device="cuda:0"), torch.rand(8, 14, 30, device="cuda:0")
    p1 = m(x1)
    p2 = m(x2)
    with pytest.raises(RuntimeError, match="overwritten"):
        p1.sum().backward()
    p2.sum().backward()                                   # the latest forward still owns the workspace
    g2 = [p.grad.clone() for p in m.parameters() if p.grad is not None]
    m.zero_grad()
    p3 = m(x2[:4])                                        # a different batch size has its own workspace
    p4 = m(x1)
    p3.sum().backward()
    p4.sum().backward()
    assert all(torch.isfinite(g).all() for g in g2)
