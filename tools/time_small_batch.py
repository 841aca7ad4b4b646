"""ST_GCN.update at small batches (us per step): auto (the matrix-core chain as ten launches), the same with F_1 .. G_0 as one launch
(mx_persist), both also replayed from a hipGraph, the fp32 chain and its cooperative single launch."""
import sys
import time

import torch

sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd import _lib                      # noqa: E402
from gnn_rul_benchmarking_amd.algorithms import ST_GCN         # noqa: E402

import os
dev = torch.device("cuda:0")
NP, PS = int(os.environ.get("NP", 14)), int(os.environ.get("PS", 30))      # NP=40 PS=64: the PHM2012 wiring
for B in [int(v) for v in sys.argv[1:]] or [100, 256, 1024, 2048, 4096]:
    row = []
    for name, path in (("auto", _lib.STEP_AUTO), ("auto+graph", _lib.STEP_AUTO), ("mx_persist", _lib.STEP_MX_PERSIST),
                       ("mx_persist+graph", _lib.STEP_MX_PERSIST), ("chain", _lib.STEP_CHAIN), ("coop", _lib.STEP_COOP)):
        torch.manual_seed(0)
        a = ST_GCN({"num_patch": NP, "patch_size": PS, "dropout": 0.2}, {"learning_rate": 1e-4, "weight_decay": 1e-4}, dev)
        a.to(dev).train()
        a.sync_loss = False
        a.model.step_path = path
        if name.endswith('graph'):
            a.enable_graphs()
        X, y = torch.rand(B, NP, PS, device=dev), torch.rand(B, 1, device=dev)
        try:
            for _ in range(20):
                a.update(X, y, 1)
        except RuntimeError as e:
            row.append(f"{name}: n/a")
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(200):
                a.update(X, y, 1)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200 * 1e6)
        row.append(f"{name}: {sorted(ts)[2]:7.1f} us")
    print(f"batch {B:6d}  " + "   ".join(row), flush=True)
