"""Golden fixtures for the STMSGCN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_stmsgcn.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py
for the shims.  Shapes follow the reference's wired configurations (configs/hparams.py:242,275,355,390)
with fewer patches so that the fixtures stay small.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.STMSGCN import Model as ref_model              # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def build(cfg, seed, perturb=True):
    torch.manual_seed(seed)
    m = ref_model.STMSGCN_model(**cfg)
    if perturb:
        g = torch.Generator().manual_seed(seed + 1000)
        with torch.no_grad():
            for _, p in m.named_parameters():
                p.add_(torch.empty_like(p).uniform_(-0.05, 0.05, generator=g))
    return m


def case_forward_backward(name, cfg, bs, seed, x_shape=None, scale=1.0):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, cfg["num_patch"] * cfg["patch_size"], generator=g) * scale
    if x_shape is not None:
        x = x.reshape(x_shape)
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, dtype=np.int64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    taps = {}
    h = m.gru_layer.register_forward_hook(lambda mod, i, o: taps.update(cat=i[0].detach().numpy().copy(),
                                                                        gru_out=o.detach().numpy().copy()))
    pred = m(x)
    h.remove()
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["pred"] = pred.detach().numpy().copy()
    out["loss"] = np.float64(loss.item())
    out["gru_in"], out["gru_out"] = taps["cat"], taps["gru_out"]
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["pred"].ravel()[:4], "loss", out["loss"], "finite", np.isfinite(out["gru_in"]).all())


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own Algorithm.update (algorithms.py:559-571) for a few steps on fixed batches."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("STMSGCN")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    g = torch.Generator().manual_seed(seed + 7)
    xs = torch.rand(steps, bs, 1, cfg["num_patch"] * cfg["patch_size"], generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, dtype=np.int64)
    for k, v in mg.state_np(algo, "sd0:").items():
        out[k] = v
    algo.train()
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(steps)]
    out["losses"] = np.asarray(losses, dtype=np.float64)
    for k, v in mg.state_np(algo, "sd_end:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


def case_trainer_phm2012(name, seed, n_train=200, n_test=60, epochs=3):
    """The reference's OWN harness (trainer.GNN_RUL_trainer) with --GNN_method STMSGCN on the synthetic
    PHM2012/Condition_1 dataset of synth.py, its own hparams (configs/hparams.py:226,242); only num_epochs is
    patched.  Records every epoch's test metrics, the results CSV and a few final tensors."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_phm2012
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_phm2012(seed, n_train, n_test)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "PHM2012", "Condition_1")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": 1.0}, os.path.join(d, "train.pt"))
        torch.save({"samples": xte, "labels": yte, "max_ruls": 1.0}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="STMSGCN", data_path=os.path.join(tmp, "data"), dataset="PHM2012",
                                      dataset_id="Condition_1", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(mg.ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "STMSGCN_run_0", "results.csv")).read()
            final = {k: v.detach().numpy().copy() for k, v in tr.algorithm.state_dict().items()}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    for k in ("model.fc.weight", "model.gcn_layers.1.linear.weight", "model.gru_layer.gru.weight_hh_l0"):
        out["final:" + k] = final[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        case_trainer_phm2012("stmsgcn_trainer_phm2012_c1_reference_run", 5)
        sys.exit(0)
    dims = {"gcn_dims": [16, 64, 16, 1], "gru_hidden_dim": 8}
    # PHM2012 Condition_1/3 wiring (patch 16, interval 6, band 5 -> 2 nodes), fewer patches
    case_forward_backward("stmsgcn_phm1_12x16_bs5", dict(num_patch=12, patch_size=16, interval=6, band_width=5, **dims), 5, seed=31)
    # PHM2012 Condition_2 wiring (patch 20, interval 2, band 3 -> 6 nodes), input as [bs, 1, L]
    case_forward_backward("stmsgcn_phm2_9x20_bs4", dict(num_patch=9, patch_size=20, interval=2, band_width=3, **dims), 4, seed=32,
                          x_shape=(4, 1, 180))
    # XJTU Condition_1/3 wiring (patch 128, interval 3, band 5 -> 25 nodes)
    case_forward_backward("stmsgcn_xjtu1_6x128_bs3", dict(num_patch=6, patch_size=128, interval=3, band_width=5, **dims), 3, seed=33,
                          x_shape=(3, 1, 768))
    # XJTU Condition_2 wiring (patch 256, interval 6, band 10 -> 25 nodes); small-amplitude signal keeps the GRU unsaturated
    case_forward_backward("stmsgcn_xjtu2_4x256_bs3", dict(num_patch=4, patch_size=256, interval=6, band_width=10, **dims), 3, seed=34,
                          scale=0.05)
    # a non-default GCN stack
    case_forward_backward("stmsgcn_dims_7x32_bs4", dict(num_patch=7, patch_size=32, interval=2, band_width=3, gcn_dims=[8, 24, 5],
                                                        gru_hidden_dim=6), 4, seed=35, scale=0.2)
    case_training_curve("stmsgcn_train_curve_9x20_bs6", dict(num_patch=9, patch_size=20, interval=2, band_width=3, **dims),
                        6, steps=12, seed=36, lr=1e-2, wd=0.0)
    case_trainer_phm2012("stmsgcn_trainer_phm2012_c1_reference_run", 5)
