"""Golden fixtures for the STNet path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_stnet.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's PHM2012 Condition_1 wiring with fewer patches and narrower ChebNets (configs/hparams.py:236: 20 patches of
128 points, nperseg 16 -> 9 frequency nodes x 9 frames, ChebNets [300, 200, 100], LSTM 10, auto-encoder 50), its Condition_3 wiring
(:303: patches of 32 points, nperseg 8 -> 5 nodes x 5 frames) and a small odd shape.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.STNet import Model as ref_model                # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402


def signal(bs, n, seed):
    """Vibration-like snapshots: a few tones whose amplitudes differ from patch to patch, plus noise -- so that the node
    weights straddle the 0.7 threshold and the adjacency is neither empty nor full."""
    rng = np.random.default_rng(seed)
    t = np.arange(n)[None, :]
    x = np.zeros((bs, n))
    for _ in range(4):
        fr = rng.uniform(0.02, 0.45, (bs, 1))
        amp = rng.uniform(0.0, 1.2, (bs, 1)) * (0.4 + 0.6 * np.sin(2 * np.pi * t / n * rng.integers(1, 6)) ** 2)
        x += amp * np.sin(2 * np.pi * fr * t + rng.uniform(0, 6.28, (bs, 1)))
    return (x + 0.3 * rng.standard_normal((bs, n))).astype(np.float32)


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.STNet_model(**cfg)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if not n_.startswith("cnn."):
                p.add_(torch.empty_like(p).uniform_(-0.02, 0.02, generator=g))
        m.cnn.weight.copy_(torch.tensor([0.3, 0.1]).view(1, 2, 1, 1))
        m.cnn.bias.fill_(0.1)
    return m


def case_forward_backward(name, cfg, bs, seed):
    m = build(cfg, seed)
    x = torch.from_numpy(signal(bs, cfg["num_patch"] * cfg["patch_size"], seed + 7))
    y = torch.rand(bs, 1, generator=torch.Generator().manual_seed(seed + 8))
    out = {"x": x.numpy().copy(), "y": y.numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.int64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    t = {}
    hs = [m.cnn.register_forward_hook(lambda mod, i, o: t.update(node_feat=i[0].detach().numpy().copy(), node_w=o.detach().numpy().copy())),
          m.chebnets[0].register_forward_hook(lambda mod, i, o: t.update(mag=i[0].detach().numpy().copy(), adj=i[1].detach().numpy().copy())),
          m.chebnets[-1].register_forward_hook(lambda mod, i, o: t.__setitem__("cheb_out", o.detach().numpy().copy())),
          m.encoder.register_forward_hook(lambda mod, i, o: t.__setitem__("H", o.detach().numpy().copy())),
          m.lstm.register_forward_hook(lambda mod, i, o: t.__setitem__("lstm_out", o[0].detach().numpy().copy()))]
    m.train()
    pred, recon = m(x, train=True)
    for h in hs:
        h.remove()
    loss = torch.nn.functional.mse_loss(pred, y) + recon
    m.zero_grad()
    loss.backward()
    out["pred"], out["recon"], out["loss"] = pred.detach().numpy().copy(), np.float64(recon.item()), np.float64(loss.item())
    for k, v in t.items():
        out[k] = v
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy() if p.grad is not None else np.zeros(tuple(p.shape), np.float32)
        out["hasgrad:" + n_] = np.bool_(p.grad is not None)
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["pred"].ravel()[:3], "loss", out["loss"], "recon", out["recon"], "mask mean", float((t["node_w"] > 0.7).mean()),
          "min |w - 0.7|", float(np.abs(t["node_w"] - 0.7).min()))


def case_init(name, cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.STNet_model(**cfg)
    out = {"seed": np.int64(seed)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.asarray(v, np.int64)
    for k, v in mg.state_np(m, "sd:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own STNet.update (algorithms.py:454-463) for a few steps on fixed batches."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("STNet")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    with torch.no_grad():
        algo.model.cnn.weight.copy_(torch.tensor([0.3, 0.1]).view(1, 2, 1, 1))
        algo.model.cnn.bias.fill_(0.1)
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
    for k in ("model.cnn.weight", "model.cnn.bias", "model.linear.weight"):
        out["sd_end:" + k] = algo.state_dict()[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


if __name__ == "__main__":
    C1 = dict(num_patch=6, patch_size=128, num_nodes=9, nperseg=16, input_dim=9, Cheb_layers=[40, 24, 12], lstm_hidden_dim=10, autoencoder_hidden_dim=50)
    case_forward_backward("stnet_phm_c1like_6x128_bs5", C1, 5, 1)
    C3 = dict(num_patch=7, patch_size=32, num_nodes=5, nperseg=8, input_dim=5, Cheb_layers=[30, 20, 10], lstm_hidden_dim=10, autoencoder_hidden_dim=50)
    case_forward_backward("stnet_phm_c3like_7x32_bs4", C3, 4, 2)
    CS = dict(num_patch=3, patch_size=24, num_nodes=4, nperseg=6, input_dim=5, Cheb_layers=[7, 5], lstm_hidden_dim=3, autoencoder_hidden_dim=6)
    case_forward_backward("stnet_small_3x24_bs6", CS, 6, 3)
    case_init("stnet_init_c1like_seed4", C1, 4)
    case_training_curve("stnet_train_curve_6x128_bs8", C1, 8, 10, 5, 1e-2, 1e-2)


def case_trainer_phm2012(name, seed, n_train=200, n_test=60, epochs=3):
    """The reference's OWN harness with --GNN_method STNet on the synthetic PHM2012 Condition_1 dataset of synth.py with its own hparams
    (configs/hparams.py:222,236: 20 patches of 128, ChebNets [300, 200, 100], batch 100, lr 1e-2, wd 1e-2; no shuffling); num_epochs patched."""
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
                                      GNN_method="STNet", data_path=os.path.join(tmp, "data"), dataset="PHM2012",
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
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "STNet_run_0", "results.csv")).read()
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum())}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, np.asarray(per_epoch))
