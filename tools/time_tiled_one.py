"""One tiled ST_GCN configuration for profiling (development aid): XJTU c1 1024x32, batch 1024."""
import sys, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import ST_GCN
dev = torch.device("cuda:0")
torch.manual_seed(0)
algo = ST_GCN({"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
algo.to(dev).train(); algo.sync_loss = False
import os; B = int(os.environ.get("TB", "1024")); x, y = torch.rand(B, 1, 32768, device=dev), torch.rand(B, 1, device=dev)
for _ in range(12): algo.update(x, y, 1)
torch.cuda.synchronize()
