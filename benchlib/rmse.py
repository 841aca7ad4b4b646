"""RMSE legs (the second half of BASELINE.json's metric): HIP path vs the torch-CPU restatement of the reference's update() on SURVEY section 8(d)'s teacher task.

Part of the benchmark harness behind bench.py (the driver's contract lives there).  The oracle imports in here are the `cpu_baseline` /
`rmse` checker legs only -- never the thing measured."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from .common import NUM_PATCH


def rmse_teacher_task(dev, epochs=20, n_train=49152, n_test=8192, batch=4096, max_rul=125.0, checkpoints=(36, 120, 240), dropout=0.0):
    """The RMSE half of BASELINE.json's metric on SURVEY section 8(d)'s synthetic task: a fixed random "teacher" ST_GCN (14 x 30, eval
    mode) labels ~49 k uniform windows (about FD004's training-set size); a student with another initialisation is trained for `epochs`
    passes in batches of `batch`, dropout off, (a) on the HIP path (ST_GCN.update) and (b) by the torch-CPU restatement of the reference's
    update (oracle/stgcn_torch_cpu.py) from the SAME initial weights on the same batches -- with ``dropout`` > 0 both under the SAME
    dropout masks (the HIP path's counter hash, imposed on the restatement: the reference's CPU Bernoulli stream cannot be matched by
    any GPU path); both are scored on held-out windows with
    the reference's formula RMSE = sqrt(mean((pred - y)^2)) * max_rul (utils.py:148-151) after 36, 120 and 240 optimizer steps (drift
    shows as a growing difference).  The north star asks |RMSE_hip - RMSE_cpu| <= 1e-3."""
    from gnn_rul_benchmarking_amd.algorithms import ST_GCN
    from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model
    from oracle import stgcn_torch_cpu as T
    from oracle import stgcn_oracle as O
    N, P = NUM_PATCH, 30
    g = torch.Generator(device="cpu").manual_seed(4242)
    Xtr, Xte = torch.rand(n_train, N, P, generator=g), torch.rand(n_test, N, P, generator=g)
    torch.manual_seed(100)
    teacher = ST_GCN_model(num_patch=N, patch_size=P, dropout=0.0).to(dev).eval()
    with torch.no_grad():
        ytr, yte = teacher(Xtr.to(dev)).cpu(), teacher(Xte.to(dev)).cpu()
    torch.manual_seed(7)
    algo = ST_GCN(dict(num_patch=N, patch_size=P, dropout=dropout), {"learning_rate": 1e-3, "weight_decay": 1e-4}, dev)
    algo.to(dev)
    init = {k: v.detach().cpu().numpy().copy() for k, v in algo.state_dict().items()}
    st = T.State(init, num_layers=2, lr=1e-3, weight_decay=1e-4)
    Xd, yd, Xted = Xtr.to(dev), ytr.to(dev), Xte.to(dev)
    yt = yte.reshape(-1).double()
    rmse = lambda p_: float(torch.sqrt(torch.mean((p_.double() - yt) ** 2)) * max_rul)
    threads = torch.get_num_threads()
    t_hip = t_cpu = 0.0
    hip_loss = cpu_loss = 0.0
    step, marks = 0, []
    for _ in range(epochs):
        for lo in range(0, n_train, batch):
            masks = None
            if dropout > 0.0:        # the masks the kernels are about to use: a function of (seed, step, layer, element) alone
                nb = min(batch, n_train - lo)
                masks = [torch.from_numpy(O.dropout_keep_mask(nb, N, O.dropout_layer_key(algo.model._seed, algo.model._step + 1, l), dropout))
                         for l in range(2)]
            t0 = time.perf_counter()
            algo.train()
            hip_loss = algo.update(Xd[lo:lo + batch], yd[lo:lo + batch], 1)["loss"]
            t_hip += time.perf_counter() - t0
            t0 = time.perf_counter()
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            cpu_loss = T.update(st, Xtr[lo:lo + batch], ytr[lo:lo + batch], N, P, dropout, masks)
            t_cpu += time.perf_counter() - t0
            step += 1
            if step in checkpoints:
                algo.eval()
                with torch.no_grad():
                    ph = algo.model(Xted).cpu().reshape(-1)
                    pc = T.forward(st, Xte, N, P, False).reshape(-1)
                marks.append({"steps": step, "rmse_hip": round(rmse(ph), 6), "rmse_torch_cpu": round(rmse(pc), 6),
                              "abs_diff": round(abs(rmse(ph) - rmse(pc)), 7), "train_loss_hip": round(float(hip_loss), 8),
                              "train_loss_torch_cpu": round(float(cpu_loss), 8), "max_pred_diff": round(float((ph.double() - pc.double()).abs().max()), 8)})
    torch.set_num_threads(threads)
    base = float(torch.sqrt(torch.mean((yt.mean() - yt) ** 2)) * max_rul)
    last = marks[-1]
    return {"task": f"teacher ST_GCN({N}, {P}) labels {n_train} uniform windows; student trained {epochs} epochs, batch {batch}, "
                    + (f"dropout {dropout} (the same masks on both paths), " if dropout > 0 else "dropout off, ") +
                    f"Adam lr 1e-3 wd 1e-4; scored on {n_test} held-out windows, RMSE x max_rul {max_rul:g} (reference utils.py:148-151)",
            "rmse_hip": last["rmse_hip"], "rmse_torch_cpu": last["rmse_torch_cpu"], "abs_diff": last["abs_diff"],
            "within_1e-3": bool(all(m["abs_diff"] <= 1e-3 for m in marks)), "after_steps": marks, "rmse_of_predicting_the_mean": round(base, 4),
            "final_train_loss_hip": last["train_loss_hip"], "final_train_loss_torch_cpu": last["train_loss_torch_cpu"],
            "steps": step, "seconds_hip": round(t_hip, 2), "seconds_torch_cpu": round(t_cpu, 2), "max_pred_diff": last["max_pred_diff"]}


def rmse_teacher_task_stmsgcn(dev, n_train=4000, n_test=1000, batch=100, epochs=6, max_rul=1.0, checkpoints=(36, 120, 240)):
    """The same experiment on a family WITHOUT BatchNorm and dropout (STMSGCN at the reference's PHM2012 Condition_1 wiring, 160 patches of
    16 points, the protocol's batch 100, configs/hparams.py): teacher-labelled windows, the student trained on the HIP path and by the
    torch-CPU restatement (oracle/families_torch_cpu.py) from the same weights on the same batches; 240 optimizer steps."""
    from gnn_rul_benchmarking_amd.algorithms import STMSGCN
    from gnn_rul_benchmarking_amd.stmsgcn import STMSGCN_model
    from gnn_rul_benchmarking_amd import hparams as HP
    from oracle import families_torch_cpu as T
    hp = HP.get_hparams_class("PHM2012")("Condition_1")
    cfg = dict(hp.alg_hparams["STMSGCN"])
    L = cfg["num_patch"] * cfg["patch_size"]
    g = torch.Generator(device="cpu").manual_seed(777)
    Xtr, Xte = torch.rand(n_train, 1, L, generator=g), torch.rand(n_test, 1, L, generator=g)
    torch.manual_seed(101)
    teacher = STMSGCN_model(**cfg).to(dev).eval()
    with torch.no_grad():
        ytr, yte = teacher(Xtr.to(dev)).cpu(), teacher(Xte.to(dev)).cpu()
    torch.manual_seed(8)
    tc = {"learning_rate": 1e-3, "weight_decay": 0.0}
    algo = STMSGCN(cfg, tc, dev)
    algo.to(dev)
    algo.train()
    init = {k: v.detach().cpu().numpy().copy() for k, v in algo.state_dict().items()}
    st = T.StmsgcnState(init, cfg, lr=tc["learning_rate"], weight_decay=tc["weight_decay"])
    Xd, yd, Xted = Xtr.to(dev), ytr.to(dev), Xte.to(dev)
    yt = yte.reshape(-1).double()
    rmse = lambda p_: float(torch.sqrt(torch.mean((p_.double() - yt) ** 2)) * max_rul)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    step, marks, t_cpu = 0, [], 0.0
    for _ in range(epochs):
        for lo in range(0, n_train, batch):
            hip_loss = algo.update(Xd[lo:lo + batch], yd[lo:lo + batch], 1)["loss"]
            t0 = time.perf_counter()
            cpu_loss = T.stmsgcn_update(st, Xtr[lo:lo + batch], ytr[lo:lo + batch])
            t_cpu += time.perf_counter() - t0
            step += 1
            if step in checkpoints:
                with torch.no_grad():
                    ph = algo.model(Xted).cpu().reshape(-1)
                    pc = T.stmsgcn_forward(st, Xte).reshape(-1)
                scale = float(yt.abs().max())
                marks.append({"steps": step, "rmse_hip": round(rmse(ph), 8), "rmse_torch_cpu": round(rmse(pc), 8),
                              "rel_diff": round(abs(rmse(ph) - rmse(pc)) / max(rmse(pc), 1e-30), 8), "train_loss_hip": float(hip_loss),
                              "train_loss_torch_cpu": float(cpu_loss), "max_pred_diff_over_label_scale": round(float((ph.double() - pc.double()).abs().max()) / scale, 8)})
    torch.set_num_threads(threads)
    return {"task": f"teacher STMSGCN (PHM2012 Condition_1 wiring {cfg['num_patch']} x {cfg['patch_size']}) labels {n_train} uniform windows; student "
                    f"trained {epochs} epochs at batch {batch} (Adam lr {tc['learning_rate']}, no weight decay; no BatchNorm, no dropout in this model); "
                    f"RMSE on {n_test} held-out windows in label units", "after_steps": marks, "steps": step, "seconds_torch_cpu": round(t_cpu, 2),
            "within_1e-3_relative": bool(all(m["rel_diff"] <= 1e-3 for m in marks))}


