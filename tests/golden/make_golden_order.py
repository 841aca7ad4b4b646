"""Golden fixtures for MPNN order k > 1 (models/ST_GCN/Model.py:74-90: sum over kk of theta_kk(A^(kk+1) X)), made by RUNNING THE
REFERENCE in this container -- same rules and shims as make_golden.py (imported, never copied; only inputs, weights and outputs
are written).

    python tests/golden/make_golden_order.py        # needs /root/reference (read-only import)

No hparams row of the reference sets k (configs/hparams.py:223,238,334,349 leave the constructor default k = 1), but
``ST_GCN_model(**configs)`` (algorithms/algorithms.py:471) forwards a ``k`` entry of the model configs: the cases below construct the
reference model / algorithm exactly that way."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402  (installs the shims, imports the reference)

ref_model, ref_utils = G.ref_model, G.ref_utils


def perturbed(num_patch, patch_size, k, num_layers, seed):
    torch.manual_seed(seed)
    m = ref_model.ST_GCN_model(num_patch, patch_size, num_layers=num_layers, dropout=G.DROPOUT_OFF, k=k)
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


def case_order(name, num_patch, patch_size, k, num_layers, bs, seed):
    """Eval prediction + train-mode prediction, loss, every live gradient and the BatchNorm buffers after the step."""
    m = perturbed(num_patch, patch_size, k, num_layers, seed)
    g = torch.Generator().manual_seed(seed + 7)
    x = torch.rand(bs, num_patch, patch_size, generator=g)
    y = torch.rand(bs, 1, generator=g)
    out = {"x": x.numpy().copy(), "y": y.numpy().copy(), "num_patch": np.int64(num_patch), "patch_size": np.int64(patch_size),
           "k": np.int64(k), "num_layers": np.int64(num_layers),
           "key_order": np.array([n for n, _ in m.named_parameters()])}
    for key, v in G.state_np(m, "sd:").items():
        out[key] = v
    m.eval()
    with torch.no_grad():
        out["eval_pred"] = m(x).numpy().copy()
    m.train()
    pred = m(x)
    loss = torch.nn.functional.mse_loss(pred, y)
    m.zero_grad()
    loss.backward()
    out["train_pred"], out["train_loss"] = pred.detach().numpy().copy(), np.float64(loss.item())
    for n_, p in m.named_parameters():
        if p.grad is not None:
            out["grad:" + n_] = p.grad.numpy().copy()
        else:
            assert ".net0." in n_ or ".net1." in n_, n_
    for key, v in G.state_np(m, "sd_after:").items():
        if "running_" in key or "num_batches" in key:
            out[key] = v
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "loss", loss.item())


def case_order_curve(name, num_patch, patch_size, k, bs, steps, seed, lr, wd):
    """`steps` calls of the reference's own ST_GCN.update (algorithms/algorithms.py:481-490) with k in the model configs."""
    ref_utils.fix_randomness(seed)
    algo = G.get_algorithm_class("ST_GCN")({"num_patch": num_patch, "patch_size": patch_size, "dropout": G.DROPOUT_OFF, "k": k},
                                           {"learning_rate": lr, "weight_decay": wd, "num_epochs": 1, "batch_size": bs}, "cpu")
    algo.model.load_state_dict(perturbed(num_patch, patch_size, k, 2, seed).state_dict())
    out = {"num_patch": np.int64(num_patch), "patch_size": np.int64(patch_size), "k": np.int64(k),
           "lr": np.float64(lr), "wd": np.float64(wd), "steps": np.int64(steps)}
    for key, v in G.state_np(algo, "sd0:").items():
        out[key] = v
    g = torch.Generator().manual_seed(seed + 11)
    xs = torch.rand(steps, bs, num_patch, patch_size, generator=g)
    ys = torch.rand(steps, bs, 1, generator=g)
    out["xs"], out["ys"] = xs.numpy().copy(), ys.numpy().copy()
    algo.train()
    losses = [algo.update(xs[s], ys[s], 1)["loss"] for s in range(steps)]
    out["losses"] = np.asarray(losses, np.float64)
    for key, v in G.state_np(algo, "sdK:").items():
        out[key] = v
    algo.model.eval()
    with torch.no_grad():
        out["eval_pred_after"] = algo.model(xs[0]).numpy().copy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, losses[:3], "...", losses[-1])


if __name__ == "__main__":
    case_order("stgcn_order2_14x30_bs19", 14, 30, 2, 2, 19, seed=31)
    case_order("stgcn_order3_14x30_bs10", 14, 30, 3, 2, 10, seed=32)
    case_order("stgcn_order2_40x64_bs5", 40, 64, 2, 2, 5, seed=33)          # PHM2012 shape: the 64-lane row mapping
    case_order("stgcn_order3_9x21_layers3_bs7", 9, 21, 3, 3, 7, seed=34)
    case_order_curve("stgcn_order2_train_curve_14x30_bs16", 14, 30, 2, 16, steps=12, seed=35, lr=1e-3, wd=1e-4)
