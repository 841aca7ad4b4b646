import os, sys, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import ST_GCN
from benchlib.common import event_time_ms
dev = torch.device("cuda:0")
for B in (100, 1024):
    torch.manual_seed(0)
    algo = ST_GCN({"num_patch": 160, "patch_size": 16, "dropout": 0.3}, {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
    algo.to(dev).train(); algo.sync_loss = False
    x, y = torch.rand(B, 160, 16, device=dev), torch.rand(B, 1, device=dev)
    ms = min(event_time_ms(lambda: algo.update(x, y, 1), 30, warm=5) for _ in range(4))
    print(os.environ.get("RULGNN_LIB", "default"), "160x16 batch", B, round(ms, 4))
