"""Golden fixtures for the FC_STGNN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_fcstgnn.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the
shims, plus ``torch.Tensor.cuda = identity`` because FC_STGNN hard-codes ``.cuda()`` (Model_Base.py:58,119,151;
SURVEY section 8c).  Shapes: the reference's five wirings (configs/hparams.py:32,69,109,149,196).  The positional-encoding
dropout (p = 0.1, hard-coded at Model.py:25) is switched off on the instantiated module so that train-mode outputs are
deterministic; nothing in the reference tree is modified.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
torch.Tensor.cuda = lambda self, *a, **k: self
from models.FC_STGNN import Model as ref_model             # noqa: E402
from algorithms.algorithms import get_algorithm_class      # noqa: E402

WIRINGS = {
    "fd001": dict(patch_size=25, num_patch=2, encoder_time_out=27, encoder_hidden_dim=8, encoder_out_dim=32, encoder_conv_kernel=2,
                  hidden_dim=8, num_sequential=6, num_node=14, num_windows=2),
    "fd002": dict(patch_size=1, num_patch=50, encoder_time_out=3, encoder_hidden_dim=8, encoder_out_dim=12, encoder_conv_kernel=2,
                  hidden_dim=8, num_sequential=10, num_node=14, num_windows=74),
    "fd003": dict(patch_size=1, num_patch=50, encoder_time_out=3, encoder_hidden_dim=8, encoder_out_dim=6, encoder_conv_kernel=2,
                  hidden_dim=24, num_sequential=25, num_node=14, num_windows=74),
    "fd004": dict(patch_size=2, num_patch=25, encoder_time_out=4, encoder_hidden_dim=8, encoder_out_dim=6, encoder_conv_kernel=2,
                  hidden_dim=8, num_sequential=10, num_node=14, num_windows=36),
    "ncmapss": dict(patch_size=2, num_patch=25, encoder_time_out=4, encoder_hidden_dim=8, encoder_out_dim=32, encoder_conv_kernel=2,
                    hidden_dim=8, num_sequential=6, num_node=20, num_windows=36),
}


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.FC_STGNN_RUL(**cfg)
    m.positional_encoding.dropout.p = 0.0
    g = torch.Generator().manual_seed(seed + 1000)
    bn = [n for n, mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm1d)]
    with torch.no_grad():
        for name, p in m.named_parameters():
            base = name.rsplit(".", 1)[0]
            if base in bn:
                if name.endswith("weight"):
                    p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
                else:
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

    def enc(mod, i, o):
        t["enc"] = o.detach().numpy().copy()

    def mpnn1(mod, i, o):
        t["pe_out"] = i[0].detach().numpy().copy()
        t["mpnn1"] = o.detach().numpy().copy()

    def mpnn2(mod, i, o):
        t["mpnn2"] = o.detach().numpy().copy()

    def adj1(mod, i, o):
        t["adj1"] = o.detach().numpy()[:4].copy()

    hs = [m.nonlin_map2.register_forward_hook(enc), m.MPNN1.register_forward_hook(mpnn1), m.MPNN2.register_forward_hook(mpnn2),
          m.MPNN1.graph_construction.register_forward_hook(adj1)]
    out = m(x)
    for h in hs:
        h.remove()
    return out, t


def state_np(m, prefix):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items() if not k.endswith("positional_encoding.pe")}


def case_forward_backward(name, cfg, bs, seed, lo=0.0, hi=1.0):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, cfg["num_node"], cfg["num_patch"] * cfg["patch_size"], generator=g) * (hi - lo) + lo
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy(), "pe_head": m.positional_encoding.pe[0, :4].numpy().copy()}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    out.update(state_np(m, "sd:"))
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
    out["train_mpnn1"] = t["mpnn1"]
    for n_, p in m.named_parameters():
        out["grad:" + n_] = p.grad.numpy().copy()
    for k, v in state_np(m, "sd_after:").items():
        if "running_" in k or "num_batches" in k:
            out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["eval_pred"].ravel()[:3], "loss", out["train_loss"])


def case_training_curve(name, cfg, bs, steps, seed, lr, wd):
    """The reference's own Algorithm.update (algorithms.py:66-76) for a few steps on fixed batches, dropout off."""
    torch.manual_seed(seed)
    algo = get_algorithm_class("FC_STGNN")(cfg, {"learning_rate": lr, "weight_decay": wd}, "cpu")
    algo.model.positional_encoding.dropout.p = 0.0
    g = torch.Generator().manual_seed(seed + 7)
    xs = torch.rand(steps, bs, cfg["num_node"], cfg["num_patch"] * cfg["patch_size"], generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out = {"xs": xs.numpy().copy(), "ys": ys.numpy().copy(), "lr": np.float64(lr), "wd": np.float64(wd), "seed": np.int64(seed),
           "state_keys": np.array(list(algo.state_dict().keys()))}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    out.update(state_np(algo, "sd0:"))
    algo.train()
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(steps)]
    out["losses"] = np.asarray(losses, dtype=np.float64)
    algo.eval()
    with torch.no_grad():
        out["eval_pred_end"] = algo.model(xs[0]).numpy().copy()
    out.update(state_np(algo, "sd_end:"))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


def case_trainer_cmapss(name, seed, n_train=250, n_test=80, epochs=3):
    """The reference's OWN harness (trainer.GNN_RUL_trainer) with --GNN_method FC_STGNN on the synthetic C-MAPSS FD004
    dataset of synth.py, its own hparams (configs/hparams.py:133,149: batch 100, lr 1e-3, wd 1e-4, 25 patches of 2) and its own
    shuffling DataLoader.  num_epochs is patched, and the algorithm class handed to the trainer switches the hard-coded
    positional-encoding dropout off after construction (torch's Bernoulli stream cannot be reproduced by any other
    implementation); nothing in the reference tree is modified."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    base = get_algorithm_class("FC_STGNN")

    class NoDropout(base):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.model.positional_encoding.dropout.p = 0.0
    NoDropout.__name__ = "FC_STGNN"
    _orig_get = ref_trainer.get_algorithm_class
    ref_trainer.get_algorithm_class = lambda n: NoDropout if n == "FC_STGNN" else _orig_get(n)
    (xtr, ytr), (xte, yte) = synthetic_cmapss(seed, n_train, n_test)
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "CMAPSS", "FD004")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": 125}, os.path.join(d, "train.pt"))
        torch.save({"samples": xte, "labels": yte, "max_ruls": 125}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="FC_STGNN", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
                                      dataset_id="FD004", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(mg.ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "FC_STGNN_run_0", "results.csv")).read()
            final = {k: v.detach().numpy().copy() for k, v in tr.algorithm.state_dict().items()}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
            ref_trainer.get_algorithm_class = _orig_get
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    for k in ("model.fc.fc4.weight", "model.MPNN1.graph_construction.mapping.weight", "model.nonlin_map.conv_block2.0.weight",
              "model.MPNN2.MPNN.bn1.running_var", "model.nonlin_map2.1.num_batches_tracked"):
        out["final:" + k] = final[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        case_trainer_cmapss("fcstgnn_trainer_cmapss_fd004_reference_run", 7)
        sys.exit(0)
    case_forward_backward("fcstgnn_fd004_bs6", WIRINGS["fd004"], 6, seed=51)
    case_forward_backward("fcstgnn_fd001_bs5", WIRINGS["fd001"], 5, seed=52)
    case_forward_backward("fcstgnn_fd002_bs3", WIRINGS["fd002"], 3, seed=53)
    # FD003 wiring (hidden_dim 24 -> 48-wide graph features) with 6 patches instead of 50: its fc1 alone is 1.2M weights at 50
    case_forward_backward("fcstgnn_fd003like_6p_bs2", dict(WIRINGS["fd003"], num_patch=6, num_windows=8), 2, seed=54)
    case_forward_backward("fcstgnn_ncmapss_bs3", WIRINGS["ncmapss"], 3, seed=55, lo=-1.0, hi=1.0)
    case_training_curve("fcstgnn_train_curve_fd004_bs10", WIRINGS["fd004"], 10, steps=12, seed=56, lr=1e-3, wd=1e-4)
    case_trainer_cmapss("fcstgnn_trainer_cmapss_fd004_reference_run", 7)
