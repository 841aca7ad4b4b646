"""Golden fixtures for the ST_Conv path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_stconv.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's C-MAPSS (configs/hparams.py:40: 14 nodes x 50 steps, kernel 6) and N-CMAPSS (:203: 20 x 50) wirings
and a small odd one.  Only tensors of the modules the forward actually uses (the "_1" layers, Model.py:196-206) are kept.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.ST_Conv import Model as ref_model              # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def dead(name):
    return "_layer_2." in name or ".net0." in name or ".net1." in name


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.ST_Conv_model(**cfg)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if dead(name):
                continue
            if name.endswith(".2.weight") or name.endswith(".bn.weight"):
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
            elif name.endswith(".2.bias") or name.endswith(".bn.bias"):
                p.copy_(torch.empty_like(p).uniform_(-0.3, 0.3, generator=g))
            elif not name.startswith("theta"):
                p.add_(torch.empty_like(p).uniform_(-0.05, 0.05, generator=g))
        for name, b in m.named_buffers():
            if dead(name):
                continue
            if name.endswith("running_mean"):
                b.copy_(torch.empty_like(b).uniform_(-0.2, 0.2, generator=g))
            elif name.endswith("running_var"):
                b.copy_(torch.empty_like(b).uniform_(0.5, 1.5, generator=g))
    return m


def case_forward_backward(name, cfg, bs, seed, lo=0.0, hi=1.0):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, cfg["num_nodes"], cfg["time_length"], generator=g) * (hi - lo) + lo
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    for k, v in mg.state_np(m, "sd:").items():
        if not dead(k):
            out[k] = v
    t = {}
    hs = [m.cnn_layer_1.register_forward_hook(lambda mod, i, o: t.__setitem__("cnn", o.detach().numpy().copy())),
          m.tcn_layer_1.register_forward_hook(lambda mod, i, o: t.__setitem__("tcn", o.detach().numpy().copy())),
          m.gcn_layer_1.register_forward_hook(lambda mod, i, o: (t.__setitem__("adj", i[1].detach().numpy().copy()), None)[1])]
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    for k, v in t.items():
        out["eval_" + k] = v
    m.train()
    pred = m(x)
    for h in hs:
        h.remove()
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["train_pred"] = pred.detach().numpy().copy()
    out["train_loss"] = np.float64(loss.item())
    out["train_cnn"], out["train_tcn"] = t["cnn"], t["tcn"]
    for n_, p in m.named_parameters():
        if p.grad is not None:
            out["grad:" + n_] = p.grad.numpy().copy()
        else:
            assert dead(n_), n_
    for k, v in mg.state_np(m, "sd_after:").items():
        if ("running_" in k or "num_batches" in k) and not dead(k):
            out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["eval_pred"].ravel()[:3], "loss", out["train_loss"])


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own Algorithm.update (algorithms.py:210-220) for a few steps on fixed batches."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("ST_Conv")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    g = torch.Generator().manual_seed(seed + 7)
    xs = torch.rand(steps, bs, cfg["num_nodes"], cfg["time_length"], generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd), "seed": np.int64(seed),
           "state_keys": np.array(list(algo.state_dict().keys()))}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    for k, v in mg.state_np(algo, "sd0:").items():
        out[k] = v
    algo.train()
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(steps)]
    out["losses"] = np.asarray(losses, dtype=np.float64)
    algo.eval()
    with torch.no_grad():
        out["eval_pred_end"] = algo.model(xs[0]).numpy().copy()
    for k, v in mg.state_np(algo, "sd_end:").items():
        if not dead(k):
            out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


if __name__ == "__main__":
    cm = dict(num_nodes=14, time_length=50, kernel_size=6)
    case_forward_backward("stconv_cmapss_14x50_bs12", cm, 12, seed=71)
    case_forward_backward("stconv_ncmapss_20x50_bs5", dict(cm, num_nodes=20), 5, seed=72, lo=-1.0, hi=1.0)
    case_forward_backward("stconv_small_6x11_bs9", dict(num_nodes=6, time_length=11, kernel_size=6), 9, seed=73)
    case_training_curve("stconv_train_curve_14x50_bs20", cm, 20, steps=12, seed=74, lr=1e-3, wd=1e-4)
