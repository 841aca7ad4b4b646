"""XJTU-SY 1024 x 32 on the tiled path: ms per update() and per eval forward at a batch (default 1024): python tools/time_tiled_step.py [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from benchlib.common import event_time_ms
from gnn_rul_benchmarking_amd.algorithms import ST_GCN

dev = torch.device("cuda:0")
for B in ([int(a) for a in sys.argv[1:]] or [1024]):
    torch.manual_seed(0)
    algo = ST_GCN({"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
    algo.to(dev).train()
    algo.sync_loss = False
    x, y = torch.rand(B, 1, 32768, device=dev), torch.rand(B, 1, device=dev)
    ms = min(event_time_ms(lambda: algo.update(x, y, 1), 10, warm=3) for _ in range(3))
    algo.eval()
    with torch.no_grad():
        ems = min(event_time_ms(lambda: algo.model(x), 10, warm=3) for _ in range(3))
    print(f"batch {B}: train {ms:.4f} ms/step, eval {ems:.4f} ms/batch", flush=True)
    del algo, x, y
