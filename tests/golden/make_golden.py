"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in this container.

    python tests/golden/make_golden.py            # needs /root/reference (read-only import)

The reference is imported, never copied: only inputs, weights and the outputs it produced are
written (as .npz).  The GPU box has no /root/reference; the tests read the .npz files only.
Shims: thop stub (utils.py:17 imports it, never calls it on this path),
torchvision stub (dataloader.py:4), np.Inf alias.  Nothing in the reference tree is modified.
"""
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
REF = os.environ.get("RUL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
warnings.filterwarnings("ignore")

import numpy as np
import torch

if "thop" not in sys.modules:
    thop = types.ModuleType("thop")
    thop.profile = lambda *a, **k: (0, 0)
    sys.modules["thop"] = thop
if "torchvision" not in sys.modules:
    tv = types.ModuleType("torchvision")
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tv.transforms
if not hasattr(np, "Inf"):
    np.Inf = np.inf

from models.ST_GCN import Model as ref_model          # noqa: E402
from algorithms.algorithms import get_algorithm_class  # noqa: E402
import utils as ref_utils                              # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from synth import synthetic_phm2012  # noqa: E402
torch.set_num_threads(1)


# nn.Dropout(0.0) in train mode returns its input, and the reference then does ``out += res``
# in place on a ReLU output (models/ST_GCN/Model.py:193-194) -> autograd raises.  The reference
# therefore cannot train with dropout exactly 0; p = 1e-12 keeps every element (keep-prob
# 1 - 1e-12, scale 1/(1-p) == 1.0f) but makes dropout out-of-place, i.e. it is "dropout off".
DROPOUT_OFF = 1e-12


def perturbed_model(num_patch, patch_size, seed, dropout=DROPOUT_OFF):
    """Reference model with weights moved away from init and non-trivial BN statistics
    (SURVEY.md section 4 hazard 5)."""
    torch.manual_seed(seed)
    m = ref_model.ST_GCN_model(num_patch, patch_size, dropout=dropout)
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if ".net0." in name or ".net1." in name:
                continue
            if name.endswith("conv_block1.2.weight") or name.endswith("conv_block2.2.weight"):
                p.copy_(torch.empty_like(p).uniform_(0.5, 1.5, generator=g))
            elif name.endswith(".2.bias"):
                p.copy_(torch.empty_like(p).uniform_(-0.3, 0.3, generator=g))
            else:
                p.add_(torch.empty_like(p).uniform_(-0.1, 0.1, generator=g))
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.empty_like(b).uniform_(-0.2, 0.2, generator=g))
            elif name.endswith("running_var"):
                b.copy_(torch.empty_like(b).uniform_(0.5, 1.5, generator=g))
    return m


def state_np(m, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in m.state_dict().items()}


def intermediates(m, x):
    """Re-walk the reference forward with the reference's own sub-modules, keeping taps."""
    bs = x.size(0)
    f = ref_model.segment_and_compute_features(x.reshape(bs * m.num_patch, m.patch_size))
    f = f.reshape(bs, m.num_patch, -1).transpose(-1, -2)
    adj = ref_model.pcc_graph_construction(f)
    out = f
    per_layer = []
    for mpnn, tcn, drop in m.sg_tcn.layers:
        res = out
        out = drop(tcn(mpnn(out, adj))) + res
        per_layer.append(out.detach().numpy().copy())
    return f.detach().numpy().copy(), adj.detach().numpy().copy(), per_layer


def case_forward_backward(name, num_patch, patch_size, bs, seed, x_shape=None, make_nan=False):
    m = perturbed_model(num_patch, patch_size, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, num_patch, patch_size, generator=g)
    if make_nan:
        x[1, 3, :] = 0.25            # constant patch -> std = 0 -> skew/kurt NaN -> whole sample NaN
    if x_shape is not None:
        x = x.reshape(x_shape)
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy(),
           "num_patch": np.int64(num_patch), "patch_size": np.int64(patch_size)}
    for k, v in state_np(m, "sd:").items():
        out[k] = v
    # eval forward + taps
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
        f, adj, per_layer = intermediates(m, x)
    out["feat"], out["adj"] = f, adj
    for l, t in enumerate(per_layer):
        out[f"eval_layer{l}"] = t
    # train-mode forward/backward (dropout p = 0 -> deterministic), MSE loss
    if not make_nan:
        m.train()
        pred = m(x)
        loss = torch.nn.functional.mse_loss(pred, y)
        m.zero_grad()
        loss.backward()
        out["train_pred"] = pred.detach().numpy().copy()
        out["train_loss"] = np.float64(loss.item())
        for n_, p in m.named_parameters():
            if p.grad is not None:
                out["grad:" + n_] = p.grad.numpy().copy()
            else:
                assert ".net0." in n_ or ".net1." in n_, n_
        for k, v in state_np(m, "sd_after:").items():
            if "running_" in k or "num_batches" in k:
                out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in out.items() if k in ("x", "eval_pred")})


def case_training_curve(name, num_patch, patch_size, bs, steps, seed, lr, wd):
    """K calls of the reference's own ST_GCN.update (algorithms/algorithms.py:481-490)."""
    ref_utils.fix_randomness(seed)
    algo_cls = get_algorithm_class("ST_GCN")
    algo = algo_cls({"num_patch": num_patch, "patch_size": patch_size, "dropout": DROPOUT_OFF},
                    {"learning_rate": lr, "weight_decay": wd, "num_epochs": 1, "batch_size": bs}, "cpu")
    src = perturbed_model(num_patch, patch_size, seed)
    algo.model.load_state_dict(src.state_dict())
    out = {"num_patch": np.int64(num_patch), "patch_size": np.int64(patch_size),
           "lr": np.float64(lr), "wd": np.float64(wd), "steps": np.int64(steps)}
    for k, v in state_np(algo, "sd0:").items():
        out[k] = v
    g = torch.Generator().manual_seed(seed + 11)
    xs = torch.rand(steps, bs, num_patch, patch_size, generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out["xs"], out["ys"] = xs.numpy().copy(), ys.numpy().copy()
    algo.train()
    losses = []
    for s in range(steps):
        losses.append(algo.update(xs[s], ys[s], 1)["loss"])
    out["losses"] = np.asarray(losses, np.float64)
    for k, v in state_np(algo, "sdK:").items():
        out[k] = v
    algo.model.eval()
    with torch.no_grad():
        out["eval_pred_after"] = algo.model(xs[0]).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


def case_init(name, num_patch, patch_size, seed):
    """Freshly constructed reference model under fix_randomness(seed): pins key names, shapes, order
    and the initial values (the drop-in consumes the torch RNG in the same order)."""
    ref_utils.fix_randomness(seed)
    m = ref_model.ST_GCN_model(num_patch, patch_size, dropout=0.2)
    out = {"num_patch": np.int64(num_patch), "patch_size": np.int64(patch_size), "seed": np.int64(seed),
           "key_order": np.array(list(m.state_dict().keys()))}
    for k, v in state_np(m, "sd:").items():
        out[k] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, len(m.state_dict()))


def case_trainer_phm2012(name, seed, n_train=200, n_test=60, epochs=3):
    """The reference's OWN harness (main.py -> trainer.GNN_RUL_trainer) on a synthetic PHM2012/Condition_1
    dataset with its own ST_GCN hparams (configs/hparams.py:223,238); only num_epochs and dropout are
    patched on the in-memory dicts (dropout 1e-12 = off, see DROPOUT_OFF).  Records every epoch's test metrics."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    import dataloader.dataloader as ref_dl
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
                                      GNN_method="ST_GCN", data_path=os.path.join(tmp, "data"), dataset="PHM2012",
                                      dataset_id="Condition_1", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            tr.model_configs["dropout"] = DROPOUT_OFF
            per_epoch = []
            orig = tr.calc_results_per_run

            def spy(run_id):
                per_epoch.append(ref_utils._calc_metrics(tr.pred_labels, tr.true_labels, tr.max_ruls))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            csv_text = open(os.path.join(tmp, "logs", "exp", "r", "ST_GCN_run_0", "results.csv")).read()
            final = {k: v.detach().numpy().copy() for k, v in tr.algorithm.state_dict().items()}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_test": np.int64(n_test), "epochs": np.int64(epochs),
           "per_epoch": np.asarray(per_epoch, np.float64), "csv_text": np.array(csv_text),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum()), "x_test_checksum": np.float64(xte.astype(np.float64).sum()),
           "batch_size": np.int64(tr.train_configs["batch_size"]), "lr": np.float64(tr.train_configs["learning_rate"])}
    for k in ("model.fc1.weight", "model.sg_tcn.layers.0.0.theta.0.weight", "model.sg_tcn.layers.1.1.conv_block2.2.running_var"):
        out["final:" + k] = final[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "per-epoch (Score_v1, Score_v2, MAE, RMSE):\n", np.asarray(per_epoch))


def case_trainer_phm2012_dict(name, seed, n_train=200, n_tests=(40, 55), epochs=3):
    """Dict-of-test-sets protocol (dataloader/dataloader.py:83-90, trainer.py:89-90,159-177,206-231): test.pt holds
    {'samples': {key: array}, 'labels': {key: array}}, train.pt['max_ruls'] is a dict with the same keys (float bearing ids, saved as
    "<int(key)>_results.csv/.pt").  The reference's own harness, as in case_trainer_phm2012."""
    import argparse
    import tempfile
    import trainer as ref_trainer
    _orig_load = torch.load
    torch.load = lambda *a, **k: _orig_load(*a, **{**k, "weights_only": False})
    (xtr, ytr), (xte, yte) = synthetic_phm2012(seed, n_train, sum(n_tests))
    keys = [3.0, 4.0]
    cut = n_tests[0]
    tx = {keys[0]: xte[:cut], keys[1]: xte[cut:]}
    ty = {keys[0]: yte[:cut], keys[1]: yte[cut:]}
    max_ruls = {keys[0]: 1.0, keys[1]: 2.5}
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, "data", "PHM2012", "Condition_1")
        os.makedirs(d)
        torch.save({"samples": xtr, "labels": ytr, "max_ruls": max_ruls}, os.path.join(d, "train.pt"))
        torch.save({"samples": tx, "labels": ty, "max_ruls": max_ruls}, os.path.join(d, "test.pt"))
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            args = argparse.Namespace(save_dir=os.path.join(tmp, "logs"), experiment_description="exp", run_description="r",
                                      GNN_method="ST_GCN", data_path=os.path.join(tmp, "data"), dataset="PHM2012",
                                      dataset_id="Condition_1", bearing_id="Testing_bearing_1", num_runs=1, device="cpu")
            tr = ref_trainer.GNN_RUL_trainer(args)
            tr.train_configs["num_epochs"] = epochs
            tr.model_configs["dropout"] = DROPOUT_OFF
            per_epoch = {k: [] for k in keys}
            orig = tr.calc_results_per_run

            def spy(run_id):
                for k in keys:
                    per_epoch[k].append(ref_utils._calc_metrics(tr.pred_labels[k], tr.true_labels[k], tr.max_ruls[k]))
                return orig(run_id)
            tr.calc_results_per_run = spy
            tr.train()
            run_dir = os.path.join(tmp, "logs", "exp", "r", "ST_GCN_run_0")
            files = sorted(f for f in os.listdir(run_dir) if f.endswith(("results.csv", "results.pt")))
            csv_text = {k: open(os.path.join(run_dir, f"{int(k)}_results.csv")).read() for k in keys}
            saved = {k: _orig_load(os.path.join(run_dir, f"{int(k)}_results.pt"), weights_only=False) for k in keys}
        finally:
            os.chdir(cwd)
            torch.load = _orig_load
    out = {"seed": np.int64(seed), "n_train": np.int64(n_train), "n_tests": np.asarray(n_tests, np.int64), "epochs": np.int64(epochs),
           "keys": np.asarray(keys, np.float64), "max_ruls": np.asarray([max_ruls[k] for k in keys], np.float64),
           "files": np.array("|".join(files)),
           "x_train_checksum": np.float64(xtr.astype(np.float64).sum())}
    for k in keys:
        out[f"per_epoch:{int(k)}"] = np.asarray(per_epoch[k], np.float64)
        out[f"csv_text:{int(k)}"] = np.array(csv_text[k])
        out[f"saved_pre:{int(k)}"] = np.asarray(saved[k]["pre"], np.float64)
        out[f"saved_max_rul:{int(k)}"] = np.float64(saved[k]["max_rul"])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, files, {k: np.asarray(v)[:, 3] for k, v in per_epoch.items()})


def case_layers(name, num_layers, seed):
    """num_layers != 2 (the reference constructor accepts it, Model.py:198): eval + train-mode gradients."""
    torch.manual_seed(seed)
    m = ref_model.ST_GCN_model(14, 30, num_layers=num_layers, dropout=DROPOUT_OFF)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n_, b in m.named_buffers():
            if n_.endswith("running_mean"):
                b.copy_(torch.empty_like(b).uniform_(-0.2, 0.2, generator=g))
            elif n_.endswith("running_var"):
                b.copy_(torch.empty_like(b).uniform_(0.5, 1.5, generator=g))
    x = torch.rand(21, 14, 30, generator=g)
    y = torch.rand(21, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy(), "num_layers": np.int64(num_layers)}
    for k, v in state_np(m, "sd:").items():
        out[k] = v
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    m.train()
    pred = m(x)
    loss = torch.nn.functional.mse_loss(pred, y)
    loss.backward()
    out["train_pred"], out["train_loss"] = pred.detach().numpy().copy(), np.float64(loss.item())
    for n_, p in m.named_parameters():
        if p.grad is not None:
            out["grad:" + n_] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def case_metrics(name, seed):
    rng = np.random.default_rng(seed)
    pred = rng.uniform(0, 1, 257)
    real = rng.uniform(0.05, 1, 257)
    s1, s2, mae, rmse = ref_utils._calc_metrics(pred, real, 125)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), pred=pred, real=real, max_rul=np.float64(125),
                        score_v1=np.float64(s1), score_v2=np.float64(s2), mae=np.float64(mae),
                        rmse=np.float64(rmse))
    print("wrote", name, s1, s2, mae, rmse)


if __name__ == "__main__":
    # BASELINE.json configs[0]: ST_GCN, C-MAPSS FD001-shaped, 14 sensors x window 30, batch 32
    case_forward_backward("stgcn_cmapss_14x30_bs32", 14, 30, 32, seed=1)
    case_forward_backward("stgcn_cmapss_14x50_bs8", 14, 50, 8, seed=2)
    # ragged: batch not a multiple of any tile, input given flat [bs, N*P] like the bearing loaders
    case_forward_backward("stgcn_16x16_bs5_flat", 16, 16, 5, seed=3, x_shape=(5, 256))
    case_forward_backward("stgcn_9x21_bs7", 9, 21, 7, seed=4)
    # reference-wired PHM2012 shape (configs/hparams.py:238), [bs, 1, 2560]
    case_forward_backward("stgcn_phm_40x64_bs4", 40, 64, 4, seed=5, x_shape=(4, 1, 2560))
    # reference-wired PHM2012 Condition_2 shape (configs/hparams.py:271): 160 patches -> tiled path
    case_forward_backward("stgcn_phm2_160x16_bs4", 160, 16, 4, seed=7, x_shape=(4, 1, 2560))
    # constant patch -> NaN propagation parity
    case_forward_backward("stgcn_nan_14x30_bs4", 14, 30, 4, seed=6, make_nan=True)
    case_training_curve("stgcn_train_curve_14x30_bs32", 14, 30, 32, steps=24, seed=8, lr=1e-3, wd=1e-4)
    case_metrics("metrics_case", 9)
    case_init("stgcn_init_14x30_seed3", 14, 30, 3)
    case_layers("stgcn_layers1_14x30_bs21", 1, 21)
    case_layers("stgcn_layers3_14x30_bs21", 3, 22)
    case_trainer_phm2012("trainer_phm2012_c1_reference_run", 5)
    case_trainer_phm2012_dict("trainer_phm2012_c1_dict_reference_run", 6)
