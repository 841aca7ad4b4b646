"""SAGCN HIP path vs the reference's golden outputs and vs the oracle (GPU)."""
import os

import numpy as np
import pytest
import torch

from oracle import sagcn_oracle as O
from test_sagcn_oracle_golden import CASES, load_case, rel

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda:0"
TOL = 1e-4


def build_model(cfg, p):
    from gnn_rul_benchmarking_amd.sagcn import SAGCN_model
    m = SAGCN_model(**cfg)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in p.items()})
    return m.to(DEV)


def grads_of(m):
    flat = m._grad_flat[:m.num_live].detach().cpu().numpy().astype(np.float64)
    return {name: flat[off:off + int(np.prod(shape))].reshape(shape) for name, (off, shape) in m._layout.items()}


def signal(bs, n, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n)[None, :]
    x = np.zeros((bs, n))
    for _ in range(3):
        fr = rng.uniform(0.02, 0.45, (bs, 1))
        x += rng.uniform(0.1, 0.5, (bs, 1)) * np.sin(2 * np.pi * fr * t + rng.uniform(0, 6.28, (bs, 1)))
    return x + 0.2 * rng.standard_normal((bs, n))


def rank_margin(x, n):
    """Relative gap between the power at the median rank and the nearest DIFFERENT power on either side, and between the two largest
    amplitudes: the two index-valued statistics are discrete decisions (fp32 kernel vs fp64 oracle); an input sitting on such an edge
    is redrawn."""
    F = np.fft.fft(x.reshape(-1, n), axis=1)
    pw = np.sort(np.abs(F) ** 2, axis=1)
    m = pw[:, n // 2:n // 2 + 1]
    below = np.where(pw < m * (1 - 1e-9), pw, -np.inf).max(1)
    above = np.where(pw > m * (1 + 1e-9), pw, np.inf).min(1)
    edge = np.minimum((m[:, 0] - below) / m[:, 0], (above - m[:, 0]) / m[:, 0])
    top = np.sort(np.abs(np.fft.rfft(x.reshape(-1, n), axis=1)), axis=1)
    return min(edge.min(), ((top[:, -1] - top[:, -2]) / top[:, -1]).min())


@pytest.mark.parametrize("name", CASES)
def test_forward_and_gradients_match_reference_golden(name):
    z, cfg, p = load_case(name)
    m = build_model(cfg, p)
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    bs = x.size(0)
    m.eval()
    with torch.no_grad():
        pred = m(x)
    assert pred.shape == (bs, 1) and rel(pred.cpu().numpy(), z["pred"]) < TOL
    assert rel(m.tap(bs, "features").cpu().numpy(), z["feat"]) < TOL
    assert rel(m.tap(bs, "h3").cpu().numpy(), z["h3"]) < TOL
    assert rel(m.tap(bs, "attention").cpu().numpy(), z["attn"]) < TOL
    m.train()
    pred2, loss = m.fused_mse_step(x, y)
    assert rel(pred2.cpu().numpy().reshape(-1, 1), z["pred"]) < TOL
    assert abs(float(loss) - float(z["loss"])) < TOL * abs(float(z["loss"]))
    g = grads_of(m)
    for k in O.param_names():
        assert rel(g[k], z["grad:" + k]) < 5e-4, k


@pytest.mark.parametrize("P,n,H,Ah,bs", [(160, 16, 100, 100, 12), (128, 20, 1000, 200, 3), (32, 1024, 1000, 100, 2), (5, 33, 7, 3, 9), (256, 3, 4, 4, 1),
                                        (1, 64, 16, 8, 4), (16, 8, 12, 6, 4), (48, 20, 30, 10, 3), (64, 64, 24, 8, 2)])
def test_training_step_matches_oracle(P, n, H, Ah, bs):
    # (patches of 2 points are accepted but not compared: their skewness is an exact zero whose rounding residue the cumulative
    #  feature c / sqrt|c| turns into 1e-4-sized noise -- in the reference's fp32 as much as here)
    cfg = dict(num_patch=P, patch_size=n, gcn_hidden_dim=H, attention_hidden_dim=Ah)
    p = O.random_params(P, H, Ah, seed=bs)
    y = np.random.default_rng(bs).uniform(0, 1, bs)
    for attempt in range(50):
        x = signal(bs, P * n, P * 10 + bs + 1000 * attempt)
        if n <= 2 or rank_margin(x, n) > 1e-5:
            break
    else:
        pytest.fail("no input away from the rank edges")
    loss, grads, fw = O.loss_and_grads(p, x, y, P, n)
    m = build_model(cfg, p).train()
    xt, yt = torch.from_numpy(x.astype(np.float32)).to(DEV), torch.from_numpy(y.astype(np.float32)).to(DEV)
    pred, l = m.fused_mse_step(xt, yt)
    assert rel(m.tap(bs, "features").cpu().numpy(), fw.feat) < TOL
    assert rel(m.tap(bs, "aggregated").cpu().numpy(), fw.ax) < TOL
    assert rel(m.tap(bs, "h3").cpu().numpy(), fw.h3) < TOL
    assert rel(m.tap(bs, "attention").cpu().numpy(), fw.attn) < TOL
    assert rel(pred.cpu().numpy().reshape(-1, 1), fw.pred) < TOL
    assert abs(float(l) - loss) < TOL * abs(loss)
    g = grads_of(m)
    for k in O.param_names():
        if not grads[k].any():          # one node: the softmax is constant, the attention layers receive exactly zero
            assert P == 1 and k.startswith("attn.") and np.abs(g[k]).max() < 1e-8, k
            continue
        assert rel(g[k], grads[k]) < 5e-4, k


def test_autograd_path_equals_fused_path():
    from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
    z, cfg, p = load_case("sagcn_phm_c2like_9x20_bs4")
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    m = build_model(cfg, p).train()
    m.fused_mse_step(x, y)
    fused = m._grad_flat[:m.num_live].clone()
    m2 = build_model(cfg, p).train()
    torch.nn.functional.mse_loss(m2(x), y).backward()
    auto = torch.cat([t.grad.reshape(-1) for t in m2._named()])
    assert torch.equal(auto, fused)
    algo = get_algorithm_class("SAGCN")(cfg, {"learning_rate": 1e-3, "weight_decay": 1e-4}, DEV)
    algo.to(DEV).train()
    before = algo.model.fc.weight.clone()
    a = algo.update(x, y, 1)["loss"]
    b = algo.update_reference_style(x, y, 1)["loss"]
    assert np.isfinite(a) and np.isfinite(b) and not torch.equal(algo.model.fc.weight, before)


def test_training_curve_matches_reference_algorithm():
    from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
    z = np.load(os.path.join(GOLD, "sagcn_train_curve_12x16_bs8.npz"))
    cfg = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg:")}
    algo = get_algorithm_class("SAGCN")(cfg, {"learning_rate": float(z["lr"]), "weight_decay": float(z["wd"])}, DEV)
    algo.load_state_dict({k[4:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd0:")})
    algo.to(DEV).train()
    xs, ys = torch.from_numpy(z["xs"]).to(DEV), torch.from_numpy(z["ys"]).to(DEV)
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(xs.size(0))]
    assert np.allclose(losses, z["losses"], rtol=1e-3), (losses, z["losses"].tolist())
    algo.eval()
    with torch.no_grad():
        assert rel(algo.model(xs[0]).cpu().numpy(), z["eval_pred_end"]) < 1e-3
    sd = algo.state_dict()
    for k in ("model.gcn1.linear.weight", "model.proj2.project_matrices.weight", "model.fc.weight"):
        assert rel(sd[k].cpu().numpy(), z["sd_end:" + k]) < 1e-3, k


def test_median_rank_tie_rule_on_exactly_mirrored_spectra():
    """Every power but DC / Nyquist appears twice: the kernel must take the bin a stable sort leaves at rank n / 2."""
    from gnn_rul_benchmarking_amd.sagcn import SAGCN_model
    for P, n in ((6, 16), (5, 20), (3, 64)):
        x = signal(4, P * n, n)
        m = SAGCN_model(P, n, 8, 4).to(DEV).eval()
        with torch.no_grad():
            m(torch.from_numpy(x.astype(np.float32)).to(DEV))
        feat = m.tap(4, "features").cpu().numpy().astype(np.float64)
        ref = O.extract_features(x, P, n)
        assert rel(feat[:, :, 13], ref[:, :, 13]) < TOL and rel(feat[:, :, 19], ref[:, :, 19]) < TOL
        assert (ref[:, :, 13] < 0).any() and (ref[:, :, 13] > 0).any()


def test_autograd_backward_after_a_second_forward_raises_instead_of_using_overwritten_activations():
    """One workspace per batch size holds the saved activations (params.ForwardTape, advisor finding of round 2): a second forward of
    the same size between a forward and its backward must raise; the latest forward still owns the workspace."""
    z, cfg, p = load_case("sagcn_phm_c2like_9x20_bs4")
    x, y = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["y"]).to(DEV)
    m = build_model(cfg, p).train()
    p1 = m(x)
    with torch.no_grad():
        m(x * 0.5)                                       # e.g. an evaluation inside the step
    with pytest.raises(RuntimeError, match="overwritten"):
        p1.sum().backward()
    p2 = m(x)
    torch.nn.functional.mse_loss(p2, y).backward()
    m2 = build_model(cfg, p).train()
    m2.fused_mse_step(x, y)
    assert torch.equal(torch.cat([t.grad.reshape(-1) for t in m._named()]), m2._grad_flat[:m2.num_live])
