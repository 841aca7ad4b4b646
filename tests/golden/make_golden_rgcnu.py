"""Golden fixtures for the RGCNU path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_rgcnu.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's C-MAPSS wiring (configs/hparams.py:42: 14 nodes, 50 steps, hidden 32, encoder 32, kernel 3, alpha 1), its
N-CMAPSS wiring (:205: 20 nodes) and a small odd shape.  SCL's Dropout(0.5) is switched off on the instantiated module for the
gradient fixtures (torch's Bernoulli stream cannot be reproduced by any other implementation); the dropout path is tested against
the oracle with the kernels' own counter-based mask.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.RGCNU import Model as ref_model                # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def build(cfg, seed, adj_gain=3.0):
    torch.manual_seed(seed)
    m = ref_model.RGCNU_model(**cfg)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            p.add_(torch.empty_like(p).uniform_(-0.05, 0.05, generator=g))
            if n_.startswith("adj."):
                p.mul_(adj_gain)          # spreads the tanh arguments: a non-trivial adjacency instead of values near zero
    m.scl.dropout.p = 0.0
    return m


def case_forward_backward(name, cfg, bs, seed, lo=0.0, hi=1.0):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, cfg["num_nodes"], cfg["time_length"], generator=g) * (hi - lo) + lo
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.float64(v)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    t = {}
    hs = [m.adj.register_forward_hook(lambda mod, i, o: t.__setitem__("adj", o.detach().numpy().copy())),
          m.scl.register_forward_hook(lambda mod, i, o: t.__setitem__("spatial", o.detach().numpy().copy())),
          m.tdl.register_forward_hook(lambda mod, i, o: t.__setitem__("temporal", o.detach().numpy().copy()))]
    m.train()
    pred, std = m(x, train=True)
    for h in hs:
        h.remove()
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["pred"], out["std"] = pred.detach().numpy().copy(), std.detach().numpy().copy()
    out["loss"] = np.float64(loss.item())
    for k, v in t.items():
        out[k] = v
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        out["hasgrad:" + n_] = np.bool_(p.grad is not None)
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["pred"].ravel()[:3], "loss", out["loss"], "adj>0:", float((t["adj"] > 0).mean()))


def case_init(name, cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.RGCNU_model(**cfg)
    out = {"seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.float64(v)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own RGCNU.update (algorithms.py:284-296) for a few steps on fixed batches (dropout off)."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("RGCNU")(cfg, {"learning_rate": lr, "weight_decay": wd, "lambda": 0.1}, "cpu")
    algo.model.scl.dropout.p = 0.0
    g = torch.Generator().manual_seed(seed + 7)
    xs = torch.rand(steps, bs, cfg["num_nodes"], cfg["time_length"], generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd), "seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.float64(v)
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
    """The reference's OWN harness with --GNN_method RGCNU on the synthetic C-MAPSS FD001 dataset of synth.py, its own hparams
    (batch 100, lr 1e-3, wd 1e-4) and its own shuffling DataLoader; num_epochs patched, SCL's dropout switched off after construction."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_cmapss(seed, n_train, n_test)
    orig_init = ref_model.SCL.__init__

    def init_without_dropout(self, *a, **k):
        orig_init(self, *a, **k)
        self.dropout.p = 0.0
    ref_model.SCL.__init__ = init_without_dropout
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "CMAPSS", "FD001")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": 125}, os.path.join(d, "train.pt"))
        torch.save({"samples": xte, "labels": yte, "max_ruls": 125}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="RGCNU", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
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
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "RGCNU_run_0", "results.csv")).read()
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
            ref_model.SCL.__init__ = orig_init
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum())}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, np.asarray(per_epoch))


if __name__ == "__main__":
    C = dict(num_nodes=14, time_length=50, hidden_dim=32, encoder_hidden_dim=32, kernel_size=3, alpha=1)
    case_forward_backward("rgcnu_cmapss_14x50_bs7", C, 7, 1)
    case_forward_backward("rgcnu_ncmapss_20x50_bs5", dict(C, num_nodes=20), 5, 2, lo=-1.0)
    case_forward_backward("rgcnu_small_5x12_bs9", dict(num_nodes=5, time_length=12, hidden_dim=6, encoder_hidden_dim=8, kernel_size=3, alpha=0.7), 9, 3)
    case_init("rgcnu_init_cmapss_seed4", C, 4)
    case_training_curve("rgcnu_train_curve_14x50_bs20", C, 20, 12, 5, 1e-3, 1e-4)
    case_trainer_cmapss("rgcnu_trainer_cmapss_fd001_reference_run", 6)
