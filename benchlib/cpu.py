"""CPU baselines: torch-CPU restatements of the reference's update() timed on the GPU box's host cores; numpy oracles for the section 8(f) families.

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


def cpu_baseline(num_patch, patch_size, dropout):
    """The reference's CPU path restated on torch-CPU (oracle/stgcn_torch_cpu.py: the same ATen kernels the reference runs,
    pinned to the reference's own training curve in tests/test_torch_cpu_baseline.py), SURVEY section 8(d) protocol:
    torch.set_num_threads(n) for n = 1 and n = all host cores (plus 16 and 64 where the host has more), 20 warm-up + 100 timed
    iterations of ST_GCN.update each, same input distribution as the GPU run; bounded sample: batch 4096 (the reference's CPU
    throughput saturates there) and the reference protocol's own batch 32 (BASELINE.json configs[0])."""
    from oracle import stgcn_torch_cpu as T
    cores = os.cpu_count() or 1
    thread_counts = sorted({1, min(16, cores), min(64, cores), cores})
    runs = []
    for n in thread_counts:
        runs.append(dict(T.time_update(num_patch, patch_size, 4096, n, dropout, warmup=20, iters=100, budget_s=14.0), what="train"))
    best = max(runs, key=lambda r: r["samples_per_s"])
    runs.append(dict(T.time_update(num_patch, patch_size, 32, 1, dropout, warmup=20, iters=100, budget_s=3.0), what="train, BASELINE.json configs[0] batch"))
    runs.append(dict(T.time_update(num_patch, patch_size, 4096, best["threads"], dropout, warmup=20, iters=100, budget_s=6.0,
                                   eval_forward=True), what="eval forward"))
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port",
            "port_of": "torch-CPU restatement of the reference's ST_GCN.update (oracle/stgcn_torch_cpu.py: the same ATen kernels in the reference's order)",
            "sample": f"{best['iterations']} ST_GCN.update iterations (after 20 warm-up) of batch 4096 ({num_patch}x{patch_size}, dropout {dropout}), "
                      f"torch-CPU restatement of the reference (oracle/stgcn_torch_cpu.py), fp32, best of thread counts {thread_counts}",
            "cpu_model": T.cpu_model_name(), "host_cpus": cores, "torch": torch.__version__, "runs": runs,
            "reference_on_survey_container": "8 vCPU Xeon 2.1 GHz: train 5.8 k samples/s at batch 32, 44.6 k/s best (BASELINE.md)"}


# SURVEY section 8d measurement configurations of the other hot-path families: (dataset, id, per-GPU batch, input shape,
# forward matmul/conv FLOPs per sample as counted there)
TORCH_CPU_FAMILIES = ("FC_STGNN", "ASTGCNN", "HAGCN", "STMSGCN")


def family_torch_cpu_baseline(family, cfg, batch, budget_s=10.0):
    """the path the reference itself takes on a CPU -- ATen kernels, autograd, torch.optim.Adam -- restated in
    oracle/families_torch_cpu.py (pinned to the reference's own fixtures, tests/test_torch_cpu_families.py) and timed on this box's host
    cores: a FULL update() at the configuration's batch, 1 thread and 16 / 64 threads, up to 100 iterations or the time budget."""
    from oracle import families_torch_cpu as T
    cores = os.cpu_count() or 1
    counts = sorted({1, min(16, cores), min(64, cores)})
    per = budget_s / len(counts)
    runs = [T.time_update(family, dict(cfg), batch, th, warmup=3, iters=100, budget_s=per) for th in counts]
    best = max(runs, key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port", "port_of": "torch-CPU restatement of the reference's update() (oracle/families_torch_cpu.py: the same ATen kernels in the reference's order)",
            "cpu_model": T.cpu_model_name(), "host_cpus": cores, "torch": torch.__version__, "runs": runs,
            "sample": f"full update() (train forward + loss + backward + torch.optim.Adam) of oracle/families_torch_cpu.py at batch {batch}, "
                      f"fp32 ATen kernels, threads {counts}, <= {per:.1f} s or 100 iterations each (iterations timed: "
                      f"{[r['iterations'] for r in runs]})",
            "reference_on_survey_container": BASELINE_MD_FAMILY.get(family)}


# the reference's own update() on the survey container (8 vCPU Xeon 2.1 GHz), samples/s
BASELINE_MD_FAMILY = {"FC_STGNN": "1 648 samples/s at batch 256", "ASTGCNN": "19 336 samples/s at batch 512", "HAGCN": "305 samples/s at batch 256 (FD001 wiring)",
                      "STMSGCN": "67 samples/s at batch 128"}


def family_cpu_baseline(family, cfg, shape, budget_s=10.0, model=None, batch=None):
    """The family's CPU baseline on this box's host cores, bounded sample: the torch-CPU restatement for the BASELINE.json families,
    the numpy oracle (train-step restatement, ``kind: port``) for the section 8(f) families."""
    import numpy as np
    if family in TORCH_CPU_FAMILIES and batch is not None:
        return family_torch_cpu_baseline(family, cfg, batch, budget_s)
    rng = np.random.default_rng(0)
    if family == "HAGCN" and model is None:
        return None
    if family == "ASTGCNN":
        from oracle import astgcnn_oracle as O
        p = O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["output_dim"], cfg["K"])
        bs = 64
        x, y = rng.uniform(-1, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y)
    elif family == "FC_STGNN":
        from oracle import fcstgnn_oracle as O
        c = O.Config(**cfg)
        p = O.random_params(c)
        bs = 32
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, c)
    elif family == "STMSGCN":
        from oracle import stmsgcn_oracle as O
        c = O.Config(cfg["num_patch"], cfg["patch_size"], cfg["interval"], cfg["band_width"], cfg["gcn_dims"], cfg["gru_hidden_dim"])
        p = O.random_params(c)
        bs = 2
        x, y = rng.uniform(0, 1, (bs, shape[1])), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, c)
    elif family == "STGNN":
        from oracle import stgnn_oracle as O
        p = O.random_params(cfg["num_patch"], cfg["patch_size"], cfg["num_nodes"], cfg["hidden_dim"], cfg["K"])
        bs = 256
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.forward_backward(x, y, p, cfg["num_patch"], cfg["patch_size"], cfg["top_k"])
    elif family == "RGCNU":
        from oracle import rgcnu_oracle as O
        p = O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["hidden_dim"], cfg["encoder_hidden_dim"], cfg["kernel_size"])
        bs = 32
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["alpha"])
    elif family == "STNet":
        from oracle import stnet_oracle as O
        p = O.random_params(cfg["num_patch"], cfg["num_nodes"], cfg["input_dim"], cfg["Cheb_layers"], cfg["lstm_hidden_dim"], cfg["autoencoder_hidden_dim"])
        bs = 8
        x, y = rng.normal(0, 1, (bs, shape[1])), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["num_patch"], cfg["patch_size"], cfg["nperseg"])
    elif family == "STAGNN":
        from oracle import stagnn_oracle as O
        p = O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["hidden_dim"], cfg["output_dim"], cfg["num_heads"])
        bs = 64
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["num_heads"], cfg["threshold"])
    elif family == "SAGCN":
        from oracle import sagcn_oracle as O
        p = O.random_params(cfg["num_patch"], cfg["gcn_hidden_dim"], cfg["attention_hidden_dim"])
        bs = 8
        x, y = rng.uniform(-0.5, 0.5, (bs, shape[1])), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["num_patch"], cfg["patch_size"])
    elif family == "HAGCN":
        # the oracle restates the model in blocks (Bi-LSTM stack, graph stack, head): one train step = their forwards and backwards
        # in sequence, on the bench model's own parameters (models/HAGCN/Model.py:149-195, algorithms.py:222-248 with alpha = 100)
        from oracle import hagcn_oracle as O
        p = {k[6:] if k.startswith("model.") else k: v.detach().double().cpu().numpy() for k, v in model.state_dict().items()}
        ps, npatch = cfg["patch_size"], cfg["num_patch"]
        bs = 4
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, (bs, 1))

        def run():
            pred, kl, fw = O.forward(p, x, ps, npatch)
            dpred = 2.0 * (pred - y) / bs
            f2 = fw.feats.reshape(bs, -1)
            h = np.maximum(f2 @ p["fc.0.weight"].T + p["fc.0.bias"], 0.0)
            dh = (dpred @ p["fc.2.weight"]) * (h > 0)
            dfeats = (dh @ p["fc.0.weight"]).reshape(fw.feats.shape)
            _, dx0 = O.graph_backward(p, fw, dfeats, 100.0)
            O.td_backward(p, x, ps, npatch, dx0)
    else:
        return None
    # the oracle's cost sits in numpy's BLAS / einsum calls: timed with 1 BLAS thread and with all host cores, the better one quoted
    from threadpoolctl import threadpool_limits
    from oracle import stgcn_torch_cpu as T
    cores = os.cpu_count() or 1
    runs = []
    for th in sorted({1, cores}):
        with threadpool_limits(limits=th):
            run()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s / 2 and n < 50:
                run()
                n += 1
            el = time.perf_counter() - t0
        runs.append({"threads": th, "samples_per_s": round(bs * n / el, 2), "steps": n})
    best = max(runs, key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port", "cpu_model": T.cpu_model_name(),
            "host_cpus": cores, "runs": runs,
            "sample": f"train steps (forward + loss + backward, no optimizer) of oracle/{family.lower()}_oracle.py, batch {bs}, fp64 numpy; "
                      f"BLAS threads 1 and {cores}, <= {budget_s / 2:.0f} s each"}


