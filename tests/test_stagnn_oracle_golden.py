"""The STAGNN oracle (oracle/stagnn_oracle.py) against outputs of the reference itself (tests/golden/stagnn_*.npz, written by
tests/golden/make_golden_stagnn.py running /root/reference here): adjacency (exact), the output of every block, the prediction in
train and eval mode, the BatchNorm running statistics after the step and every parameter gradient."""
import os

import numpy as np
import pytest

from oracle import stagnn_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["stagnn_cmapss_fd001_h64_bs6", "stagnn_cmapss_fd002_h16_bs7", "stagnn_ncmapss_h32_bs5", "stagnn_small_5x12_bs9"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k[4:]: (float(z[k]) if k == "cfg:threshold" else int(z[k])) for k in z.files if k.startswith("cfg:")}
    p = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("sd:") and z[k].dtype != np.int64}
    return z, cfg, p


def grad_floor(grads):
    """Gradients that are mathematically zero (the attention bias when every pre-activation of a row has one sign: the softmax is
    shift-invariant) come out as rounding residue on both sides: compare against a floor tied to the largest gradient."""
    return 1e-7 * max(np.abs(v).max() for v in grads.values())


@pytest.mark.parametrize("name", CASES)
def test_forward_and_gradients_match_reference(name):
    z, cfg, p = load_case(name)
    heads, thr = cfg["num_heads"], cfg["threshold"]
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    loss, g, fw = O.loss_and_grads(p, x, y, heads, thr)
    assert np.array_equal(fw.adj, z["adj"]) and 0.2 < fw.adj.mean() < 0.9          # a mixed graph
    for tap, ref in ((fw.graph_out, "gat2"), (fw.tcn1_out, "tcn1"), (fw.enc1_out, "enc1"), (fw.tcn2_out, "tcn2"), (fw.enc2_out, "enc2")):
        assert rel(tap, z[ref]) < 1e-5, ref
    assert rel(fw.pred, z["pred"]) < 1e-5 and abs(loss - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    assert rel(O.forward(p, x, heads, thr, training=False).pred, z["eval_pred"]) < 1e-5
    for k, v in O.running_stats_after(p, fw).items():
        assert rel(v, z["sd_after:" + k]) < 1e-5, k
    live = [k[8:] for k in z.files if k.startswith("hasgrad:") and bool(z[k])]
    dead = [k[8:] for k in z.files if k.startswith("hasgrad:") and not bool(z[k])]
    assert live == O.live_param_names(heads) and len(dead) == 20 and all(".net0." in k or ".net1." in k for k in dead)
    floor = grad_floor({k: z["grad:" + k] for k in live})
    for k in live:
        ref = z["grad:" + k].astype(np.float64)
        assert np.abs(g[k] - ref).max() < 1e-4 * max(np.abs(ref).max(), floor / 1e-4 * 10), k


def test_backward_finite_difference():
    rng = np.random.default_rng(0)
    N, L, h, out, heads = 4, 9, 6, 3, 2
    p = O.random_params(N, L, h, out, heads, seed=1)
    x, y = rng.uniform(0, 1, (5, N, L)), rng.uniform(0, 1, 5)
    loss, g, fw = O.loss_and_grads(p, x, y, heads, 0.0)
    assert 0 < fw.adj.mean() < 1
    for k in O.live_param_names(heads):
        idx = tuple(rng.integers(0, s) for s in p[k].shape)
        eps = 1e-6
        q = {m: v.copy() for m, v in p.items()}
        q[k][idx] += eps
        lp = O.loss_and_grads(q, x, y, heads, 0.0)[0]
        q[k][idx] -= 2 * eps
        lm = O.loss_and_grads(q, x, y, heads, 0.0)[0]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[k][idx]) < 2e-5 * max(abs(fd), 1e-3) + 1e-9, (k, fd, g[k][idx])
