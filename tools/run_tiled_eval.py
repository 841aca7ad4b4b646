"""Eval forwards of ST_GCN at XJTU-SY 1024 x 32 (tiled path) for a kernel trace: python tools/run_tiled_eval.py [batch]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_rul_benchmarking_amd.algorithms import ST_GCN
dev = torch.device("cuda:0")
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
algo = ST_GCN({"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
algo.to(dev).eval()
x = torch.rand(B, 1, 32768, device=dev)
with torch.no_grad():
    for _ in range(12):
        algo.model(x)
torch.cuda.synchronize()
