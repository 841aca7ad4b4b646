"""The STNet oracle (oracle/stnet_oracle.py) against outputs of the reference itself (tests/golden/stnet_*.npz, written by
tests/golden/make_golden_stnet.py running /root/reference here): STFT magnitudes, node weights / adjacency, ChebNet output, encoder
output, LSTM output, prediction, reconstruction loss and every parameter gradient."""
import os

import numpy as np
import pytest

from oracle import stnet_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["stnet_phm_c1like_6x128_bs5", "stnet_phm_c3like_7x32_bs4", "stnet_small_3x24_bs6"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k[4:]: (z[k].tolist() if z[k].ndim else int(z[k])) for k in z.files if k.startswith("cfg:")}
    p = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("sd:")}
    return z, cfg, p


@pytest.mark.parametrize("name", CASES)
def test_forward_and_gradients_match_reference(name):
    z, cfg, p = load_case(name)
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    T, P, ns = cfg["num_patch"], cfg["patch_size"], cfg["nperseg"]
    loss, g, fw = O.loss_and_grads(p, x, y, T, P, ns)
    assert fw.mag.shape[1:] == (cfg["num_nodes"], cfg["input_dim"])
    assert rel(fw.mag, z["mag"]) < 1e-5
    assert rel(fw.node_w.reshape(z["node_w"].shape), z["node_w"]) < 1e-5
    adj = fw.mask[:, :, None] * fw.mask[:, None, :]
    assert np.array_equal(adj, z["adj"])
    assert 0.1 < fw.mask.mean() < 0.9                                   # neither an empty nor a full graph
    assert rel(fw.Yo.reshape(z["cheb_out"].shape), z["cheb_out"]) < 2e-5
    assert rel(fw.H, z["H"]) < 2e-5
    assert rel(fw.hseq, z["lstm_out"]) < 2e-5
    assert rel(fw.pred, z["pred"]) < 2e-5
    assert abs(fw.recon - float(z["recon"])) < 2e-5 * float(z["recon"])
    assert abs(loss - float(z["loss"])) < 2e-5 * abs(float(z["loss"]))
    assert rel(O.forward(p, x, T, P, ns).pred, z["eval_pred"]) < 2e-5      # no dropout, no BatchNorm: eval == train
    for k in O.param_names(len(cfg["Cheb_layers"])):
        ref = z["grad:" + k].astype(np.float64)
        if not bool(z["hasgrad:" + k]):
            assert k.startswith("cnn.") and not g[k].any()                  # the 1x1 convolution only feeds a threshold
            continue
        assert rel(g[k], ref) < 5e-4, k


def test_backward_finite_difference():
    cfg = dict(num_patch=3, patch_size=24, nperseg=6)
    rng = np.random.default_rng(0)
    p = O.random_params(3, 4, 5, [6, 4], 3, 5, seed=1)
    x, y = rng.normal(0, 1.5, (4, 72)), rng.uniform(0, 1, 4)
    loss, g, fw = O.loss_and_grads(p, x, y, **cfg)
    assert 0 < fw.mask.mean() < 1
    for k in ("chebnets.0.filters", "chebnets.1.filters", "encoder.0.weight", "decoder.6.weight", "lstm.weight_hh_l0", "linear.weight", "encoder.4.bias"):
        idx = tuple(rng.integers(0, s) for s in p[k].shape)
        eps = 1e-6 * max(1.0, abs(p[k][idx]))
        q = {n: v.copy() for n, v in p.items()}
        q[k][idx] += eps
        lp = O.loss_and_grads(q, x, y, **cfg)[0]
        q[k][idx] -= 2 * eps
        lm = O.loss_and_grads(q, x, y, **cfg)[0]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[k][idx]) < 1e-5 * max(abs(fd), 1e-3) + 1e-7 * abs(loss), (k, fd, g[k][idx])
