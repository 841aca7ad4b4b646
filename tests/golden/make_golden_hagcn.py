"""Golden fixtures for the HAGCN path, produced by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden_hagcn.py     # needs /root/reference (read-only import)

Only data is written (inputs, weights, the outputs/gradients the reference produced); see make_golden.py for the shims.
Shapes: the reference's wirings (configs/hparams.py:41 FD001: 5 patches of 10; :79,119 FD002/3 and :204 N-CMAPSS: 2 of 25;
:159 FD004: 1 of 50).  The two Dropout(0.2) modules of the LSTM stack are switched off on the instantiated module so that
train-mode outputs are deterministic; nothing in the reference tree is modified.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                                   # noqa: E402  (installs the shims, sets sys.path)
from models.HAGCN import Model as ref_model                # noqa: E402

ALPHA = 100.0                                              # configs/hparams.py:22


def build(cfg, seed):
    torch.manual_seed(seed)
    m = ref_model.HAGCN_model(**cfg)
    for d in (m.TD.drop1, m.TD.drop2, m.TD.drop3):
        d.p = 0.0
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.startswith("TD."):
                continue
            p.add_(torch.empty_like(p).uniform_(-0.05, 0.05, generator=g))
            if ".rank." in name or (name.startswith("gnn") and ".mlp.2." in name):
                p.mul_(3.0)                                # spread the node scores: well separated top-k selections
    return m


def case(name, cfg, bs, num_node, seed, lo=0.0, hi=1.0, keep_td=False, keep_td_grads=False):
    m = build(cfg, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, num_node, cfg["patch_size"] * cfg["num_patch"], generator=g) * (hi - lo) + lo
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy(), "alpha": np.float64(ALPHA)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    for k, v in mg.state_np(m, "sd:").items():
        if keep_td or not k.startswith("sd:TD."):          # the LSTM stack is 320k of the 366k weights: kept in one fixture only
            out[k] = v
    t = {}

    def gin1(mod, i, o):
        t["nodes"] = i[0]
        t["adj0"] = i[1].detach().numpy().copy()
        t["gin1"] = o.detach().numpy().copy()

    def sag(l):
        def f(mod, i, o):
            t[f"xo{l}"] = o[0].detach().numpy().copy()
            t[f"adj{l}"] = o[1].detach().numpy().copy()
            t[f"kl{l}"] = np.float64(o[2].item())
        return f
    hs = [m.gin1.register_forward_hook(gin1), m.gnn1.register_forward_hook(sag(1)), m.gnn2.register_forward_hook(sag(2)),
          m.gnn3.register_forward_hook(sag(3))]
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    m.train()
    # record the indices torch.sort hands to the three SAGPool layers (Model.py:106-107): on these inputs the node scores
    # are equal to within fp32 rounding, so WHICH nodes are kept is decided by rounding noise and has to be part of the fixture
    sorted_idx = []
    _sort = torch.sort

    def recording_sort(*a, **k):
        r = _sort(*a, **k)
        sorted_idx.append(r[1].detach().numpy().copy())
        return r
    torch.sort = recording_sort
    try:
        pred, kl = m(x, train=True)
    finally:
        torch.sort = _sort
    assert len(sorted_idx) == 3
    for l, idx in enumerate(sorted_idx):
        out[f"order{l + 1}"] = idx.astype(np.int64)
    nodes = t["nodes"]
    nodes.retain_grad()
    loss = torch.nn.functional.mse_loss(pred, y) + ALPHA * kl
    m.zero_grad()
    loss.backward()
    for h in hs:
        h.remove()
    out["nodes"] = nodes.detach().numpy().copy()
    out["grad_nodes"] = nodes.grad.numpy().copy()
    for k in ("adj0", "gin1", "xo1", "adj1", "kl1", "xo2", "adj2", "kl2", "xo3", "adj3", "kl3"):
        out[k] = t[k]
    out["train_pred"] = pred.detach().numpy().copy()
    out["train_kl"] = np.float64(kl.item())
    out["train_loss"] = np.float64(loss.item())
    for n_, p in m.named_parameters():
        if keep_td_grads or not n_.startswith("TD."):      # LSTM gradients: kept in the small-LSTM fixture only
            out["grad:" + n_] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, out["eval_pred"].ravel()[:3], "kl", out["train_kl"], "loss", out["train_loss"])


def case_margins(name, dims, G, num_node, seed0, min_gap=1e-3):
    """Graph stack alone (GIN + SAGPool x 3, Model.py:165-176) on node features whose scores are NOT tied: standard-normal nodes instead
    of LSTM outputs (whose cosine adjacency is ~1 everywhere), seeds searched until every RELATIVE gap between consecutive sorted scores
    among the first k + 1 of every graph and level is >= min_gap.  There the reference's selection is a function of the arithmetic, not of
    rounding noise, and an implementation must reproduce its sort indices EXACTLY (bit-exact index work)."""
    for seed in range(seed0, seed0 + 500):
        m = build(dict(patch_size=10, num_patch=1, **dims), seed)
        g = torch.Generator().manual_seed(seed + 7)
        nodes = torch.randn(G, num_node, dims["encoder_hidden_dim"], generator=g)
        rec = []
        _sort = torch.sort

        def recording_sort(*a, **k):
            r = _sort(*a, **k)
            rec.append((r[0].detach().numpy().copy(), r[1].detach().numpy().copy()))
            return r
        torch.sort = recording_sort
        try:
            with torch.no_grad():
                adj0 = ref_model.cosine_distance(nodes)
                o1, a1, k1 = m.gnn1(m.gin1(nodes, adj0), adj0)
                o2, a2, k2 = m.gnn2(m.gin2(o1, a1), a1)
                o3, a3, k3 = m.gnn3(m.gin3(o2, a2), a2)
        finally:
            torch.sort = _sort
        # relative gaps (= gaps of the logits the softmax scores come from): 1e-3 is a thousand times their fp32 rounding noise
        gaps = [float(np.min((v[:, :k] - v[:, 1:k + 1]) / v[:, :k])) for (v, _), k in zip(rec, (10, 5, 1))]
        if min(gaps) < min_gap:
            continue
        out = {"nodes": nodes.numpy().copy(), "seed": np.int64(seed), "gaps": np.array(gaps),
               "feats": torch.cat([o1.mean(1), o2.mean(1), o3.mean(1)], -1).numpy().copy(), "kl": np.float64((k1 + k2 + k3).item())}
        for k, v in dims.items():
            out["cfg:" + k] = np.int64(v)
        for k, v in mg.state_np(m, "sd:").items():
            if not k.startswith("sd:TD.") and not k.startswith("sd:fc."):
                out[k] = v
        for l, (_, idx) in enumerate(rec):
            out[f"order{l + 1}"] = idx.astype(np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name, "seed", seed, "smallest gaps per level", gaps)
        return
    raise RuntimeError("no seed with separated scores found")


def case_init(name, cfg, seed):
    """Key order and per-tensor checksums of a freshly constructed reference model (initial-weight parity for a seed)."""
    torch.manual_seed(seed)
    m = ref_model.HAGCN_model(**cfg)
    sd = m.state_dict()
    out = {"keys": np.array(list(sd.keys())), "seed": np.int64(seed),
           "sums": np.array([float(v.double().sum()) for v in sd.values()]),
           "abssums": np.array([float(v.double().abs().sum()) for v in sd.values()]),
           "numel": np.array([v.numel() for v in sd.values()], dtype=np.int64)}
    for k, v in cfg.items():
        out["cfg:" + k] = np.int64(v)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(sd), "keys")


def case_trainer_cmapss(name, seed, n_train=250, n_test=80, epochs=3):
    """The reference's OWN harness (trainer.GNN_RUL_trainer) with --GNN_method HAGCN on the synthetic C-MAPSS FD004 dataset of
    synth.py, its own hparams (configs/hparams.py:140,159: batch 100, lr 1e-3, wd 1e-4, alpha 100, one patch of 50) and its
    shuffling DataLoader.  num_epochs is patched and the algorithm class handed to the trainer switches the LSTM-stack dropout
    off after construction (torch's Bernoulli stream cannot be reproduced elsewhere)."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    from algorithms.algorithms import get_algorithm_class
    from synth import synthetic_cmapss
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    base = get_algorithm_class("HAGCN")

    class NoDropout(base):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            for d in (self.model.TD.drop1, self.model.TD.drop2, self.model.TD.drop3):
                d.p = 0.0
    NoDropout.__name__ = "HAGCN"
    _orig_get = ref_trainer.get_algorithm_class
    ref_trainer.get_algorithm_class = lambda n: NoDropout if n == "HAGCN" else _orig_get(n)
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
                                      GNN_method="HAGCN", data_path=os.path.join(tmp, "data"), dataset="CMAPSS",
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
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
            ref_trainer.get_algorithm_class = _orig_get
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "x_train_checksum": np.float64(xtr.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "trainer":
        case_trainer_cmapss("hagcn_trainer_cmapss_fd004_reference_run", 8)
        sys.exit(0)
    dims = dict(hidden_dim=64, encoder_hidden_dim=60, output_dim=32)
    if len(sys.argv) > 1 and sys.argv[1] == "margins":
        case_margins("hagcn_margins_14x60_g6", dims, 6, 14, 300)
        sys.exit(0)
    case("hagcn_fd001_5x10_bs6", dict(patch_size=10, num_patch=5, **dims), 6, 14, seed=61, keep_td=True)
    case("hagcn_fd002_2x25_bs5", dict(patch_size=25, num_patch=2, **dims), 5, 14, seed=62)
    case("hagcn_fd004_1x50_bs7", dict(patch_size=50, num_patch=1, **dims), 7, 14, seed=63)
    case("hagcn_ncmapss_2x25_bs3", dict(patch_size=25, num_patch=2, **dims), 3, 20, seed=64, lo=-1.0, hi=1.0)
    # a narrow LSTM stack (hidden 8 / 16 / 8) so that its weights AND gradients fit a small fixture: pins the BPTT of the oracle
    case("hagcn_smalllstm_3x6_bs4", dict(patch_size=6, num_patch=3, hidden_dim=16, encoder_hidden_dim=8, output_dim=4), 4, 12, seed=66,
         keep_td=True, keep_td_grads=True)
    case_init("hagcn_init_fd004_seed65", dict(patch_size=50, num_patch=1, **dims), 65)
    case_margins("hagcn_margins_14x60_g6", dims, 6, 14, 300)
    case_trainer_cmapss("hagcn_trainer_cmapss_fd004_reference_run", 8)
