"""The SAGCN oracle (oracle/sagcn_oracle.py) against outputs of the reference itself (tests/golden/sagcn_*.npz, written by
tests/golden/make_golden_sagcn.py running /root/reference here): the 40 normalised statistics, the cosine adjacency, the hidden
activations, the attention, the prediction and every parameter gradient.  The two fixtures with patches longer than 16 points were
produced with the reference's unstable argsort pinned to the stable order (see the generator's `stable_argsort`); the others, the
training curve and the trainer run are the unmodified reference."""
import os

import numpy as np
import pytest

from oracle import sagcn_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["sagcn_phm_c1like_12x16_bs5", "sagcn_phm_c2like_9x20_bs4", "sagcn_xjtu_like_4x1024_bs3", "sagcn_small_3x7_bs6"]


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    cfg = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg:")}
    p = {k[3:]: z[k].astype(np.float64) for k in z.files if k.startswith("sd:")}
    return z, cfg, p


@pytest.mark.parametrize("name", CASES)
def test_forward_and_gradients_match_reference(name):
    z, cfg, p = load_case(name)
    P, n = cfg["num_patch"], cfg["patch_size"]
    assert bool(z["argsort_pinned_stable"]) == (n > 16)
    x, y = z["x"].astype(np.float64), z["y"].astype(np.float64)
    ff = O.frequency_features(x.reshape(-1, n))
    assert np.abs(ff[:, 1] - z["freq_feat32"][:, 1]).max() < 1e-6            # the bin at the median rank of the sorted power
    assert np.abs(ff[:, 7] - z["freq_feat32"][:, 7]).max() < 1e-6            # the bin of the largest amplitude
    assert (ff[:, 1] < 0).any() and (ff[:, 1] > 0).any() and (ff[:, 7] >= 0).all()
    loss, g, fw = O.loss_and_grads(p, x, y, P, n)
    assert rel(fw.feat, z["feat"]) < 1e-4
    assert rel(fw.adj, z["adj"]) < 1e-5
    assert rel(fw.h1, z["h1"]) < 1e-4 and rel(fw.h2, z["h2"]) < 1e-4 and rel(fw.h3, z["h3"]) < 1e-4
    assert rel(fw.attn, z["attn"]) < 1e-5
    assert rel(fw.pred, z["pred"]) < 1e-5
    assert abs(loss - float(z["loss"])) < 1e-5 * abs(float(z["loss"]))
    for k in O.param_names():
        assert g[k].shape == z["grad:" + k].shape and rel(g[k], z["grad:" + k]) < 1e-4, k


def test_backward_finite_difference():
    rng = np.random.default_rng(0)
    P, n, H, Ah = 4, 10, 6, 5
    p = O.random_params(P, H, Ah, seed=1)
    x, y = rng.normal(0, 0.6, (3, P * n)), rng.uniform(0, 1, 3)
    loss, g, fw = O.loss_and_grads(p, x, y, P, n)
    for k in O.param_names():
        idx = tuple(rng.integers(0, s) for s in p[k].shape)
        eps = 1e-6
        q = {m: v.copy() for m, v in p.items()}
        q[k][idx] += eps
        lp = O.loss_and_grads(q, x, y, P, n)[0]
        q[k][idx] -= 2 * eps
        lm = O.loss_and_grads(q, x, y, P, n)[0]
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[k][idx]) < 1e-5 * max(abs(fd), 1e-3) + 1e-9, (k, fd, g[k][idx])


def test_spectrum_is_mirrored_exactly_and_the_tie_rule_is_the_stable_order():
    rng = np.random.default_rng(3)
    for n in (7, 16, 20, 64):
        s = rng.normal(0, 1, (40, n))
        ff = O.frequency_features(s)
        F = np.fft.fft(s, axis=1)
        psd = np.abs(F) ** 2 / n
        assert np.allclose(ff[:, 2], psd.sum(1), rtol=1e-12)
        assert np.allclose(ff[:, 3], 1.0)                                        # every fftfreq bin lies below fs / 2
        k = np.round(np.abs(ff[:, 1]) * n).astype(int)                           # |median bin|: independent of the tie rule
        half = np.sort(np.concatenate([psd, psd], 1)[:, :n], axis=1)             # the mirrored spectrum sorted
        assert np.allclose(psd[np.arange(40), k], half[:, n // 2], rtol=1e-9)
