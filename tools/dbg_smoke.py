import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
import os
os.environ["AMD_LOG_LEVEL"] = os.environ.get("AMD_LOG_LEVEL", "0")
from gnn_rul_benchmarking_amd.algorithms import ST_GCN
dev = torch.device("cuda:0")
torch.manual_seed(0)
algo = ST_GCN({"num_patch": 14, "patch_size": 30, "dropout": 0.2}, {"learning_rate": 1e-3, "weight_decay": 1e-4}, dev)
print("constructed on", algo.model.flat_params.device)
algo.to(dev)
print("moved to", algo.model.flat_params.device, algo.model._bn.device)
x = torch.rand(64, 14, 30, device=dev)
algo.eval()
with torch.no_grad():
    try:
        p = algo.model(x); torch.cuda.synchronize(); print("ok", p[:3].view(-1))
    except Exception as e:
        print("ERR", e)
x = torch.rand(65, 14, 30, device=dev)
with torch.no_grad():
    p = algo.model(x); torch.cuda.synchronize(); print("ok65", p[:3].view(-1))
