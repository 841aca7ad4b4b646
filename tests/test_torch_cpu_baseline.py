"""The torch-CPU restatement that bench.py times as `cpu_baseline` (oracle/stgcn_torch_cpu.py) is pinned here: against the
24-step loss curve of the reference's own ST_GCN.update (tests/golden/stgcn_train_curve_14x30_bs32.npz, produced by
tests/golden/make_golden.py from /root/reference) and against the fp64 numpy oracle."""
import os

import numpy as np
import torch

from oracle import stgcn_oracle as O
from oracle import stgcn_torch_cpu as T

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_update_reproduces_the_reference_training_curve():
    z = np.load(os.path.join(GOLD, "stgcn_train_curve_14x30_bs32.npz"))
    N, P = int(z["num_patch"]), int(z["patch_size"])
    sd0 = {k[4:]: z[k] for k in z.files if k.startswith("sd0:")}
    st = T.State(sd0, 2, lr=float(z["lr"]), weight_decay=float(z["wd"]))
    losses = [T.update(st, torch.from_numpy(z["xs"][i]), torch.from_numpy(z["ys"][i]), N, P, dropout=1e-12) for i in range(int(z["steps"]))]
    np.testing.assert_allclose(losses, z["losses"], rtol=2e-4)
    sdK = {k[4 + 6:]: z[k] for k in z.files if k.startswith("sdK:")}
    for name, t in st.p.items():
        np.testing.assert_allclose(t.detach().numpy(), sdK[name], rtol=2e-3, atol=2e-5, err_msg=name)
    for name, t in st.buf.items():
        np.testing.assert_allclose(t.numpy(), sdK[name], rtol=1e-4, atol=1e-6, err_msg=name)


def test_forward_and_gradients_match_the_fp64_numpy_oracle():
    N, P, B = 14, 30, 48
    prm = O.random_params(N, 2, seed=5, dtype=np.float64)
    rng = np.random.default_rng(2)
    x, y = rng.uniform(0, 1, (B, N, P)), rng.uniform(0, 1, (B, 1))
    st = T.State(prm, 2, dtype=torch.float64)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    with torch.no_grad():
        ev = T.forward(st, xt, N, P, False).numpy()
    np.testing.assert_allclose(ev, O.forward(prm, x, N, P, train=False).pred, rtol=1e-9, atol=1e-11)
    pred = T.forward(st, xt, N, P, True)
    loss = torch.nn.functional.mse_loss(pred, yt)
    loss.backward()
    fc = O.forward(prm, x, N, P, train=True)
    ref_loss, dpred = O.mse_loss_and_grad(fc.pred, y)
    assert abs(float(loss) - ref_loss) < 1e-10
    g = O.backward(prm, fc, dpred, 0.0)
    for name, t in st.p.items():
        np.testing.assert_allclose(t.grad.numpy(), g[name], rtol=1e-7, atol=1e-11, err_msg=name)


def test_imposed_dropout_masks_match_the_fp64_numpy_oracle_with_dropout_on():
    """The restatement under IMPOSED masks (the HIP path's counter hash: oracle dropout_keep_mask) equals the fp64 oracle's train-mode
    forward and gradients with dropout 0.3 -- what lets the RMSE leg of bench.py train both paths with dropout on."""
    N, P, B, p, seed, step = 14, 30, 40, 0.3, 77, 5
    prm = O.random_params(N, 2, seed=6, dtype=np.float64)
    rng = np.random.default_rng(3)
    x, y = rng.uniform(0, 1, (B, N, P)), rng.uniform(0, 1, (B, 1))
    keys = [O.dropout_layer_key(seed, step, l) for l in range(2)]
    masks = [torch.from_numpy(O.dropout_keep_mask(B, N, k, p)) for k in keys]
    assert 0.6 < float(masks[0].double().mean()) < 0.8
    st = T.State(prm, 2, dtype=torch.float64)
    pred = T.forward(st, torch.from_numpy(x), N, P, True, p, masks)
    loss = torch.nn.functional.mse_loss(pred, torch.from_numpy(y))
    loss.backward()
    fc = O.forward(prm, x, N, P, train=True, dropout=p, dropout_keys=keys)
    np.testing.assert_allclose(pred.detach().numpy(), fc.pred, rtol=1e-9, atol=1e-11)
    ref_loss, dpred = O.mse_loss_and_grad(fc.pred, y)
    assert abs(float(loss) - ref_loss) < 1e-10
    g = O.backward(prm, fc, dpred, p)
    for name, t in st.p.items():
        np.testing.assert_allclose(t.grad.numpy(), g[name], rtol=1e-7, atol=1e-11, err_msg=name)


def test_time_update_reports_the_protocol_fields():
    r = T.time_update(14, 30, 64, threads=1, dropout=0.2, warmup=2, iters=3, budget_s=5.0)
    assert r["iterations"] == 3 and r["threads"] == 1 and r["samples_per_s"] > 0
    assert isinstance(T.cpu_model_name(), str) and T.cpu_model_name()
