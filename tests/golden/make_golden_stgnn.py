"""Golden fixtures for the STGNN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_stgnn.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's C-MAPSS wiring (configs/hparams.py:47: one patch of 50, 14 nodes, hidden 64, K 3, top-k 10), its
N-CMAPSS wiring (:211: 5 patches of 10, 20 nodes), and a small odd shape with K = 2.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.STGNN import Model as ref_model                # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.STGNN_model(**cfg)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for _, p in m.named_parameters():
            p.add_(torch.empty_like(p).uniform_(-0.05, 0.05, generator=g))
    return m


def case_forward_backward(name, cfg, bs, seed, lo=0.0, hi=1.0):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, cfg["num_nodes"], cfg["num_patch"] * cfg["patch_size"], generator=g) * (hi - lo) + lo
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    t = {}

    def cheb_hook(mod, inputs, o):
        t["adj"] = inputs[1].detach().numpy().copy()
        t["cheb"] = o.detach().numpy().copy()

    def gru_hook(mod, inputs, o):
        t["gru_out"] = o[0].detach().numpy().copy()

    hs = [m.chebnet.register_forward_hook(cheb_hook), m.gru.register_forward_hook(gru_hook)]
    m.train()
    pred = m(x)
    for h in hs:
        h.remove()
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["pred"] = pred.detach().numpy().copy()
    out["loss"] = np.float64(loss.item())
    for k, v in t.items():
        out[k] = v
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy()
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["pred"].ravel()[:3], "loss", out["loss"])


def case_init(name, cfg, seed):
    """State dict straight after construction under torch.manual_seed(seed): pins the initialisation order."""
    torch.manual_seed(seed)
    m = ref_model.STGNN_model(**cfg)
    out = {"seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own Algorithm.update (algorithms.py:399-408) for a few steps on fixed batches."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("STGNN")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    g = torch.Generator().manual_seed(seed + 7)
    xs = torch.rand(steps, bs, cfg["num_nodes"], cfg["num_patch"] * cfg["patch_size"], generator=g)
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
    """The reference's OWN harness (trainer.GNN_RUL_trainer) with --GNN_method STGNN on the synthetic C-MAPSS FD003 dataset
    of synth.py, its own hparams (configs/hparams.py:105,128: batch 100, lr 1e-3, wd 1e-4) and its own shuffling DataLoader;
    only num_epochs is patched.  Records every epoch's test metrics, the results CSV and a few final tensors."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_cmapss(seed, n_train, n_test)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "CMAPSS", "FD003")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": 125}, os.path.join(d, "train.pt"))
        torch.save({"samples": xte, "labels": yte, "max_ruls": 125}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="STGNN", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
                                      dataset_id="FD003", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(mg.ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "STGNN_run_0", "results.csv")).read()
            final = {k: v.detach().numpy().copy() for k, v in tr.algorithm.state_dict().items()}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    for k in ("model.fc.weight", "model.fc.bias", "model.gru.bias_hh_l0", "model.gru.weight_ih_l0"):
        out["final:" + k] = final[k]
    out["final_mean:model.chebnet.filters"] = final["model.chebnet.filters"].mean(axis=1)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        case_trainer_cmapss("stgnn_trainer_cmapss_fd003_reference_run", 8)
        sys.exit(0)
    cmapss = dict(patch_size=50, num_patch=1, num_nodes=14, hidden_dim=64, K=3, top_k=10)
    ncmapss = dict(patch_size=10, num_patch=5, num_nodes=20, hidden_dim=64, K=3, top_k=10)
    small = dict(patch_size=7, num_patch=3, num_nodes=6, hidden_dim=12, K=2, top_k=4)
    case_forward_backward("stgnn_cmapss_1x50_bs9", cmapss, 9, 11)
    case_forward_backward("stgnn_ncmapss_5x10_bs5", ncmapss, 5, 12, lo=-1.0, hi=1.0)
    case_forward_backward("stgnn_small_3x7_bs6", small, 6, 13)
    case_init("stgnn_init_cmapss_seed5", cmapss, 5)
    case_training_curve("stgnn_train_curve_1x50_bs16", cmapss, 16, 20, 3, 1e-3, 1e-4)
    case_trainer_cmapss("stgnn_trainer_cmapss_fd003_reference_run", 8)
