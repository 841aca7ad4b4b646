"""Golden fixtures for the SAGCN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_sagcn.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's PHM2012 wirings with fewer patches and narrower layers (configs/hparams.py:235: patches of 16 points, hidden
100/100; :266: patches of 20 points -- a non power of two spectrum), the XJTU_SY patch length (:346: 1024 points) and a small odd shape.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.SAGCN import Model as ref_model                # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def signal(bs, n, seed):
    """Vibration-like snapshots inside (-1, 1) mostly, a few samples beyond the arcsin clamp."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)[None, :]
    x = np.zeros((bs, n))
    for _ in range(3):
        fr = rng.uniform(0.02, 0.45, (bs, 1))
        x += rng.uniform(0.1, 0.5, (bs, 1)) * np.sin(2 * np.pi * fr * t + rng.uniform(0, 6.28, (bs, 1)))
    return (x + 0.2 * rng.standard_normal((bs, n))).astype(np.float32)


def build(cfg, seed):
    torch.manual_seed(seed)
    return ref_model.SAGCN_model(**cfg)


class stable_argsort:
    """The reference takes `median_freq` from an UNSTABLE torch.argsort of an exactly mirrored power spectrum; for patches longer than
    16 points the order among the equal keys is whatever the sort algorithm leaves (and differs between torch's CPU and GPU sorts).
    Inside this context the reference runs with that one call pinned to the stable order -- the rule this package documents."""

    def __enter__(self):
        self.orig = torch.argsort
        torch.argsort = lambda *a, **k: self.orig(*a, **{**k, "stable": True})

    def __exit__(self, *exc):
        torch.argsort = self.orig


def case_forward_backward(name, cfg, bs, seed, pin_ties=False):
    if pin_ties:
        with stable_argsort():
            return case_forward_backward(name, cfg, bs, seed)
    m = build(cfg, seed)
    x = torch.from_numpy(signal(bs, cfg["num_patch"] * cfg["patch_size"], seed + 7))
    y = torch.rand(bs, 1, generator=torch.Generator().manual_seed(seed + 8))
    out = {"x": x.numpy().copy(), "y": y.numpy().copy(), "argsort_pinned_stable": np.bool_(torch.argsort.__name__ == "<lambda>")}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.int64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    t = {}
    hs = [m.gcn1.register_forward_hook(lambda mod, i, o: t.update(feat=i[0].detach().numpy().copy(), adj=i[1].detach().numpy().copy(),
                                                                  h1=o.detach().numpy().copy())),
          m.proj2.register_forward_hook(lambda mod, i, o: t.update(h2=i[0].detach().numpy().copy(), h3=o.detach().numpy().copy())),
          m.attn.register_forward_hook(lambda mod, i, o: t.__setitem__("attn", o.detach().numpy().copy()))]
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
    # the raw (un-normalised) 20 features of every patch in float64, for the two index-valued features' tie rule
    with torch.no_grad():
        s = x.double().reshape(bs * cfg["num_patch"], -1)
        out["freq_feat64"] = ref_model.extract_frequency_features(s).numpy().copy()
        out["freq_feat32"] = ref_model.extract_frequency_features(s.float()).numpy().copy()
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["pred"].ravel()[:3], "loss", out["loss"], "adj range", t["adj"].min(), t["adj"].max())


def case_init(name, cfg, seed):
    m = build(cfg, seed)
    out = {"seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.int64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own SAGCN.update (algorithms.py:427-436) for a few steps on fixed batches."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("SAGCN")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    xs = torch.from_numpy(np.stack([signal(bs, cfg["num_patch"] * cfg["patch_size"], seed + 20 + s) for s in range(steps)]))
    ys = torch.rand(steps, bs, 1, generator=torch.Generator().manual_seed(seed + 9))
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd), "seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.int64)
    for k, v in mg.state_np(algo, "sd0:").items():
        out[k] = v
    algo.train()
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(steps)]
    out["losses"] = np.asarray(losses, dtype=np.float64)
    algo.eval()
    with torch.no_grad():
        out["eval_pred_end"] = algo.model(xs[0]).numpy().copy()
    for k in ("model.gcn1.linear.weight", "model.proj2.project_matrices.weight", "model.fc.weight"):
        out["sd_end:" + k] = algo.state_dict()[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


def case_trainer_phm2012(name, seed, n_train=200, n_test=60, epochs=3):
    """The reference's OWN harness with --GNN_method SAGCN on the synthetic PHM2012 Condition_1 dataset of synth.py with its own hparams
    (configs/hparams.py:221,235: 160 patches of 16, hidden 100/100, batch 100, lr 1e-4, wd 1e-4; no shuffling); num_epochs patched."""
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
                                      GNN_method="SAGCN", data_path=os.path.join(tmp, "data"), dataset="PHM2012",
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
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "SAGCN_run_0", "results.csv")).read()
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum())}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, np.asarray(per_epoch))


if __name__ == "__main__":
    C1 = dict(num_patch=12, patch_size=16, gcn_hidden_dim=24, attention_hidden_dim=20)
    case_forward_backward("sagcn_phm_c1like_12x16_bs5", C1, 5, 1)
    C2 = dict(num_patch=9, patch_size=20, gcn_hidden_dim=40, attention_hidden_dim=12)
    case_forward_backward("sagcn_phm_c2like_9x20_bs4", C2, 4, 2, pin_ties=True)
    CX = dict(num_patch=4, patch_size=1024, gcn_hidden_dim=16, attention_hidden_dim=10)
    case_forward_backward("sagcn_xjtu_like_4x1024_bs3", CX, 3, 3, pin_ties=True)
    CS = dict(num_patch=3, patch_size=7, gcn_hidden_dim=5, attention_hidden_dim=6)
    case_forward_backward("sagcn_small_3x7_bs6", CS, 6, 6)
    case_init("sagcn_init_c1like_seed4", C1, 4)
    case_training_curve("sagcn_train_curve_12x16_bs8", C1, 8, 10, 5, 1e-3, 1e-4)
    if "--trainer" in sys.argv:
        case_trainer_phm2012("sagcn_trainer_phm2012_c1_reference_run", 11)
