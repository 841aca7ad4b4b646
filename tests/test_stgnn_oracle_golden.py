"""The STGNN oracle (oracle/stgnn_oracle.py) against fixtures produced by running the reference
(tests/golden/make_golden_stgnn.py): adjacency, ChebNet output, GRU output, prediction, loss, every gradient, and the
reference's own Algorithm.update for 20 steps."""
import glob
import os

import numpy as np
import pytest

from oracle import stgnn_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "stgnn_*.npz")) if "init" not in p and "curve" not in p and "trainer" not in p)


def load(path):
    z = np.load(path)
    cfg = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg:")}
    p = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("sd:")}
    return z, cfg, p


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_oracle_matches_reference_forward_and_gradients(path):
    z, cfg, p = load(path)
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    loss, grads, out, c = O.forward_backward(x, y, p, cfg["num_patch"], cfg["patch_size"], cfg["top_k"])
    bs, N, H, L = x.shape[0], cfg["num_nodes"], cfg["hidden_dim"], cfg["num_patch"]
    assert np.allclose(c["adj"], z["adj"], rtol=1e-4, atol=1e-7)
    assert (c["adj"] != 0).sum(-1).max() <= cfg["top_k"] and ((c["adj"] != 0) == (z["adj"] != 0)).all()
    assert np.allclose(c["cheb"], z["cheb"], rtol=1e-4, atol=1e-5)
    assert np.allclose(c["hs"], z["gru_out"], rtol=1e-4, atol=1e-6)
    assert np.allclose(out, z["pred"], rtol=1e-4, atol=1e-6)
    assert np.allclose(out, z["eval_pred"], rtol=1e-4, atol=1e-6)           # no dropout, no BatchNorm: eval == train
    assert abs(loss - float(z["loss"])) <= 1e-5 * abs(float(z["loss"]))
    for k in O.param_names():
        g = z["grad:" + k]
        assert np.allclose(grads[k], g, rtol=2e-3, atol=1e-6 + 1e-4 * np.abs(g).max()), k


def test_oracle_follows_the_reference_training_curve():
    z = np.load(os.path.join(GOLD, "stgnn_train_curve_1x50_bs16.npz"))
    cfg = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg:")}
    p = {k[len("sd0:model."):]: z[k].astype(np.float64) for k in z.files if k.startswith("sd0:model.")}
    state = {}
    losses = []
    for s in range(z["xs"].shape[0]):
        loss, grads, _, _ = O.forward_backward(z["xs"][s].astype(np.float64), z["ys"][s].astype(np.float64), p, cfg["num_patch"],
                                               cfg["patch_size"], cfg["top_k"])
        losses.append(loss)
        O.adam_step(p, grads, state, float(z["lr"]), float(z["wd"]))
    assert np.allclose(losses, z["losses"], rtol=2e-3)
    for k in O.param_names():
        assert np.allclose(p[k], z["sd_end:model." + k], rtol=1e-3, atol=2e-4), k
    assert np.allclose(O.forward(z["xs"][0].astype(np.float64), p, cfg["num_patch"], cfg["patch_size"], cfg["top_k"]), z["eval_pred_end"],
                       rtol=1e-3, atol=1e-4)
