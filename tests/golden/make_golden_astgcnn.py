"""Golden fixtures for the ASTGCNN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_astgcnn.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for
the shims.  Shapes: the reference's C-MAPSS wiring (configs/hparams.py:38: 14 nodes x 50 steps) and N-CMAPSS
wiring (:202: 20 x 50), plus a small odd shape and K = 2.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.ASTGCNN import Model as ref_model              # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.ASTGCNN_model(**cfg)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if ".net0." in name or ".net1." in name:
                continue
            if name.endswith(".2.weight"):
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
            elif name.endswith(".2.bias"):
                p.copy_(torch.empty_like(p).uniform_(-0.3, 0.3, generator=g))
            else:
                p.add_(torch.empty_like(p).uniform_(-0.05, 0.05, generator=g))
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.empty_like(b).uniform_(-0.2, 0.2, generator=g))
            elif name.endswith("running_var"):
                b.copy_(torch.empty_like(b).uniform_(0.5, 1.5, generator=g))
    return m


def taps(m, x):
    t = {}
    hs = [m.tcn.register_forward_hook(lambda mod, i, o: t.__setitem__("tcn_out", o.detach().numpy().copy())),
          m.gate.register_forward_hook(lambda mod, i, o: t.__setitem__("gated", o.detach().numpy().copy())),
          m.distance_module.register_forward_hook(lambda mod, i, o: t.__setitem__("adj", o.detach().numpy().copy())),
          m.chebnet.register_forward_hook(lambda mod, i, o: t.__setitem__("cheb", o.detach().numpy().copy()))]
    out = m(x)
    for h in hs:
        h.remove()
    return out, t


def case_forward_backward(name, cfg, bs, seed, lo=0.0, hi=1.0):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, cfg["num_nodes"], cfg["time_length"], generator=g) * (hi - lo) + lo
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    m.eval()
    with torch.no_grad():
        pred, t = taps(m, x)
    out["eval_pred"] = pred.numpy().copy()
    for k, v in t.items():
        out["eval_" + k] = v
    m.train()
    pred, t = taps(m, x)
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["train_pred"] = pred.detach().numpy().copy()
    out["train_loss"] = np.float64(loss.item())
    for k, v in t.items():
        out["train_" + k] = v
    for n_, p in m.named_parameters():
        if p.grad is not None:
            out["grad:" + n_] = p.grad.numpy().copy()
        else:
            assert ".net0." in n_ or ".net1." in n_, n_
    for k, v in mg.state_np(m, "sd_after:").items():
        if "running_" in k or "num_batches" in k:
            out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["eval_pred"].ravel()[:3], "loss", out["train_loss"])


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own Algorithm.update (algorithms.py:153-163) for a few steps on fixed batches."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("ASTGCNN")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    g = torch.Generator().manual_seed(seed + 7)
    xs = torch.rand(steps, bs, cfg["num_nodes"], cfg["time_length"], generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd), "seed": np.int64(seed)}
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
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


def case_trainer_cmapss(name, seed, n_train=250, n_test=80, epochs=3):
    """The reference's OWN harness (trainer.GNN_RUL_trainer) with --GNN_method ASTGCNN on the synthetic C-MAPSS FD001
    dataset of synth.py, its own hparams (configs/hparams.py:19,38: batch 100, lr 1e-3, wd 1e-4) and its own shuffling
    DataLoader (data_model_configs.py:13); only num_epochs is patched.  Records every epoch's test metrics, the results
    CSV and a few final tensors."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_cmapss(seed, n_train, n_test)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "CMAPSS", "FD001")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": 125}, os.path.join(d, "train.pt"))
        torch.save({"samples": xte, "labels": yte, "max_ruls": 125}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="ASTGCNN", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
                                      dataset_id="FD001", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(mg.ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "ASTGCNN_run_0", "results.csv")).read()
            final = {k: v.detach().numpy().copy() for k, v in tr.algorithm.state_dict().items()}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    for k in ("model.fc.weight", "model.chebnet.filters", "model.tcn.conv_block2.0.weight", "model.tcn.conv_block2.2.running_var",
              "model.tcn.conv_block1.2.num_batches_tracked"):
        out["final:" + k] = final[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        case_trainer_cmapss("astgcnn_trainer_cmapss_fd001_reference_run", 6)
        sys.exit(0)
    cm = dict(num_nodes=14, time_length=50, encoder_out_dim=50, output_dim=64, K=3)
    case_forward_backward("astgcnn_cmapss_14x50_bs16", cm, 16, seed=41)
    # N-CMAPSS is scaled to [-1, 1] (Data_read_NCMAPSS.py:230)
    case_forward_backward("astgcnn_ncmapss_20x50_bs6", dict(cm, num_nodes=20), 6, seed=42, lo=-1.0, hi=1.0)
    case_forward_backward("astgcnn_small_5x12_bs9", dict(num_nodes=5, time_length=12, encoder_out_dim=12, output_dim=8, K=3), 9, seed=43)
    case_forward_backward("astgcnn_k2_7x20_bs4", dict(num_nodes=7, time_length=20, encoder_out_dim=20, output_dim=16, K=2), 4, seed=44)
    case_training_curve("astgcnn_train_curve_14x50_bs20", cm, 20, steps=16, seed=45, lr=1e-3, wd=1e-4)
    case_trainer_cmapss("astgcnn_trainer_cmapss_fd001_reference_run", 6)
