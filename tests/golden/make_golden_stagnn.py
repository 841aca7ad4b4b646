"""Golden fixtures for the STAGNN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_stagnn.py [--trainer]     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's C-MAPSS wirings (configs/hparams.py:43,82,122: 14 sensors x 50 points, hidden 64 / 16 / 32, output 10,
3 heads, threshold 0), its N-CMAPSS wiring (:206: 20 sensors) and a small odd shape with a non-zero threshold.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.STAGNN import Model as ref_model                # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def windows(bs, N, L, seed):
    """C-MAPSS-like windows in [0, 1]: drifting sensors, some positively and some negatively correlated (a mixed adjacency)."""
    rng = np.random.default_rng(seed)
    t = np.linspace(0, 1, L)[None, None, :]
    slope = rng.uniform(-0.5, 0.5, (bs, N, 1))
    x = 0.5 + slope * (t - 0.5) + 0.08 * rng.standard_normal((bs, N, L))
    return np.clip(x, 0, 1).astype(np.float32)


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.STAGNN_model(**cfg)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if ".conv_block" in n_ and (n_.endswith(".2.weight") or n_.endswith(".2.bias")):
                p.add_(torch.empty_like(p).uniform_(-0.2, 0.2, generator=g))          # BatchNorm affine off its (1, 0) initialisation
        for n_, b in m.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(torch.empty_like(b).uniform_(-0.2, 0.2, generator=g))
            elif n_.endswith("running_var"):
                b.copy_(torch.empty_like(b).uniform_(0.5, 1.5, generator=g))
    return m


def case_forward_backward(name, cfg, bs, seed):
    m = build(cfg, seed)
    x = torch.from_numpy(windows(bs, cfg["num_nodes"], cfg["time_length"], seed + 7))
    y = torch.rand(bs, 1, generator=torch.Generator().manual_seed(seed + 8))
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.float64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    t = {}
    hs = [m.gcn1.register_forward_hook(lambda mod, i, o: t.update(adj=i[1].detach().numpy().copy(), gcn1=o.detach().numpy().copy())),
          m.gat1.register_forward_hook(lambda mod, i, o: t.__setitem__("gat1", o.detach().numpy().copy())),
          m.gat2.register_forward_hook(lambda mod, i, o: t.__setitem__("gat2", o.detach().numpy().copy())),
          m.tcn1.register_forward_hook(lambda mod, i, o: t.__setitem__("tcn1", o.detach().numpy().copy())),
          m.temporal_encoder1.register_forward_hook(lambda mod, i, o: t.__setitem__("enc1", o.detach().numpy().copy())),
          m.tcn2.register_forward_hook(lambda mod, i, o: t.__setitem__("tcn2", o.detach().numpy().copy())),
          m.temporal_encoder2.register_forward_hook(lambda mod, i, o: t.__setitem__("enc2", o.detach().numpy().copy()))]
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()                 # running statistics as loaded
    m.train()
    pred = m(x)
    for h in hs:
        h.remove()
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["pred"], out["loss"] = pred.detach().numpy().copy(), np.float64(loss.item())
    for k, v in t.items():
        out[k] = v
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        out["hasgrad:" + n_] = np.bool_(p.grad is not None)
    for k, v in mg.state_np(m, "sd_after:").items():           # BatchNorm running statistics after the one train-mode forward
        if "running" in k or "num_batches" in k:
            out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["pred"].ravel()[:3], "loss", out["loss"], "adj mean", t["adj"].mean())


def case_init(name, cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.STAGNN_model(**cfg)
    out = {"seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.float64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    out["param_names"] = np.array([n for n, _ in m.named_parameters()])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(out["param_names"]), "parameters,", len(m.state_dict()), "state_dict entries")


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own STAGNN.update (algorithms.py:314-323) for a few steps on fixed batches, then an eval forward."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("STAGNN")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    xs = torch.from_numpy(np.stack([windows(bs, cfg["num_nodes"], cfg["time_length"], seed + 20 + s) for s in range(steps)]))
    ys = torch.rand(steps, bs, 1, generator=torch.Generator().manual_seed(seed + 9))
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd), "seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.float64)
    for k, v in mg.state_np(algo, "sd0:").items():
        out[k] = v
    algo.train()
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(steps)]
    out["losses"] = np.asarray(losses, dtype=np.float64)
    algo.eval()
    with torch.no_grad():
        out["eval_pred_end"] = algo.model(xs[0]).numpy().copy()
    for k, v in mg.state_np(algo, "sd_end:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


def case_trainer_cmapss(name, seed, fd="FD002", n_train=300, n_test=80, epochs=3):
    """The reference's OWN harness with --GNN_method STAGNN on the synthetic C-MAPSS dataset of synth.py with its own hparams
    (configs/hparams.py:62,82: FD002 -> hidden 16, batch 100, lr 1e-3, wd 1e-4; no shuffling); num_epochs patched."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_cmapss(seed, n_train, n_test)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "CMAPSS", fd)
        os.makedirs(d)
        torch.save({"samples": torch.from_numpy(xtr), "labels": torch.from_numpy(ytr), "max_ruls": 125.0}, os.path.join(d, "train.pt"))
        torch.save({"samples": torch.from_numpy(xte), "labels": torch.from_numpy(yte), "max_ruls": 125.0}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="STAGNN", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
                                      dataset_id=fd, bearing_id=None, num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(mg.ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "STAGNN_run_0", "results.csv")).read()
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs), "fd": np.array(fd),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum())}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, np.asarray(per_epoch))


if __name__ == "__main__":
    FD1 = dict(num_nodes=14, time_length=50, hidden_dim=64, output_dim=10, num_heads=3, threshold=0)
    case_forward_backward("stagnn_cmapss_fd001_h64_bs6", FD1, 6, 1)
    FD2 = dict(num_nodes=14, time_length=50, hidden_dim=16, output_dim=10, num_heads=3, threshold=0)
    case_forward_backward("stagnn_cmapss_fd002_h16_bs7", FD2, 7, 2)
    NC = dict(num_nodes=20, time_length=50, hidden_dim=32, output_dim=10, num_heads=3, threshold=0)
    case_forward_backward("stagnn_ncmapss_h32_bs5", NC, 5, 3)
    CS = dict(num_nodes=5, time_length=12, hidden_dim=9, output_dim=4, num_heads=2, threshold=0.001)
    case_forward_backward("stagnn_small_5x12_bs9", CS, 9, 4)
    case_init("stagnn_init_fd002_seed4", FD2, 4)
    case_training_curve("stagnn_train_curve_fd002_bs20", FD2, 20, 10, 5, 1e-3, 1e-4)
    if "--trainer" in sys.argv:
        case_trainer_cmapss("stagnn_trainer_cmapss_fd002_reference_run", 12)
