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


def case_trainer_cmapss(name, seed, n_train=250, n_test=80, epochs=3):
    """The reference's OWN harness (trainer.GNN_RUL_trainer) with --GNN_method ST_Conv on the synthetic C-MAPSS FD002 dataset of
    synth.py, its own hparams (configs/hparams.py:59,78) and its shuffling DataLoader; only num_epochs is patched."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_cmapss(seed, n_train, n_test)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "CMAPSS", "FD002")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": 125}, os.path.join(d, "train.pt"))
        torch.save({"samples": xte, "labels": yte, "max_ruls": 125}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="ST_Conv", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
                                      dataset_id="FD002", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(mg.ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            final = {k: v.detach().numpy().copy() for k, v in tr.algorithm.state_dict().items()}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "x_train_checksum": np.float64(xtr.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    for k in ("model.theta1", "model.fc.weight", "model.cnn_layer_1.conv.weight", "model.tcn_layer_1.conv_block2.2.running_var",
              "model.cnn_layer_1.bn.num_batches_tracked"):
        out["final:" + k] = final[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        case_trainer_cmapss("stconv_trainer_cmapss_fd002_reference_run", 9)
        sys.exit(0)
    cm = dict(num_nodes=14, time_length=50, kernel_size=6)
    case_forward_backward("stconv_cmapss_14x50_bs12", cm, 12, seed=71)
    case_forward_backward("stconv_ncmapss_20x50_bs5", dict(cm, num_nodes=20), 5, seed=72, lo=-1.0, hi=1.0)
    case_forward_backward("stconv_small_6x11_bs9", dict(num_nodes=6, time_length=11, kernel_size=6), 9, seed=73)
    case_training_curve("stconv_train_curve_14x50_bs20", cm, 20, steps=12, seed=74, lr=1e-3, wd=1e-4)
    case_trainer_cmapss("stconv_trainer_cmapss_fd002_reference_run", 9)
