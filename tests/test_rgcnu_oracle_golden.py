"""The RGCNU oracle (oracle/rgcnu_oracle.py) against outputs of the reference itself (tests/golden/rgcnu_*.npz, written by
tests/golden/make_golden_rgcnu.py running /root/reference here): adjacency, spatial and temporal features, both heads, loss and every
parameter gradient -- including the adjacency-tiling quirk of Model.py:104-106."""
import os

import numpy as np
import pytest

from oracle import rgcnu_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["rgcnu_cmapss_14x50_bs7", "rgcnu_ncmapss_20x50_bs5", "rgcnu_small_5x12_bs9"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k[4:]: (float(z[k]) if k == "cfg:alpha" else int(z[k])) for k in z.files if k.startswith("cfg:")}
    p = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("sd:")}
    return z, cfg, p


@pytest.mark.parametrize("name", CASES)
def test_forward_and_gradients_match_reference(name):
    z, cfg, p = load_case(name)
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    loss, g, fw = O.loss_and_grads(p, x, y, cfg["alpha"])
    assert rel(fw.A, z["adj"]) < 1e-5
    assert rel(fw.spatial, z["spatial"]) < 1e-5
    assert rel(fw.hseq, z["temporal"]) < 1e-5
    assert rel(fw.pred, z["pred"]) < 1e-5 and rel(fw.std, z["std"]) < 1e-5
    assert abs(loss - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    assert rel(O.forward(p, x, cfg["alpha"]).pred, z["eval_pred"]) < 1e-5          # no BatchNorm: eval == train without dropout
    gmax = max(np.abs(z["grad:" + k]).max() for k in O.param_names())
    for k in O.param_names():
        ref = z["grad:" + k].astype(np.float64)
        if not bool(z["hasgrad:" + k]):
            assert k.startswith("fusion.fc2") and not g[k].any()                    # the `std` head is not in the loss
            continue
        assert np.abs(g[k] - ref).max() / max(np.abs(ref).max(), 1e-3 * gmax) < 2e-4, k


def test_adjacency_tiling_quirk_is_what_the_fixture_pins():
    """Graph (b, l) uses the adjacency of sample (b * L + l) % bs: with the natural pairing (sample b's own adjacency) the
    spatial features of the fixture are NOT reproduced."""
    z, cfg, p = load_case("rgcnu_cmapss_14x50_bs7")
    x = z["x"].astype(np.float64)
    fw = O.forward(p, x, cfg["alpha"])
    bs, N, L = x.shape
    assert (fw.gidx != np.arange(bs)[:, None]).any()
    _, Ahat = O.normalise(fw.A)
    natural = np.einsum("bij,bjl->bil", Ahat, x)                   # what gcn1 would aggregate with sample b's own adjacency
    assert rel(natural.transpose(0, 2, 1), fw.ax1) > 1e-2


def test_dropout_mask_enters_forward_and_backward_consistently():
    """Finite-difference check of the hand-derived backward with a keep mask on SCL's hidden features."""
    cfg = dict(N=5, L=6, H=4, E=3, k=3)
    rng = np.random.default_rng(0)
    p = O.random_params(cfg["N"], cfg["L"], cfg["H"], cfg["E"], cfg["k"], seed=1)
    x, y = rng.uniform(0, 1, (4, cfg["N"], cfg["L"])), rng.uniform(0, 1, 4)
    keep = (rng.uniform(size=(4, cfg["L"], cfg["N"], cfg["H"])) > 0.5) * 2.0
    loss, g, _ = O.loss_and_grads(p, x, y, 0.9, keep)
    for k in ("adj.trainable_theta1.weight", "scl.gcn2.linear.weight", "tdl.lstm.weight_hh_l0", "fusion.cnn2.weight", "scl.gcn1.linear.bias"):
        idx = tuple(rng.integers(0, s) for s in p[k].shape)
        q = {n: v.copy() for n, v in p.items()}
        eps = 1e-6
        q[k][idx] += eps
        lp = O.loss_and_grads(q, x, y, 0.9, keep)[0]
        q[k][idx] -= 2 * eps
        lm = O.loss_and_grads(q, x, y, 0.9, keep)[0]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[k][idx]) < 1e-6 + 1e-4 * abs(fd), (k, fd, g[k][idx])
