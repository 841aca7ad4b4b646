"""development aid: six ST_GCN.update calls at batch 100 in the single-launch form (RULGNN_STEP_MX_PERSIST); with a library built with
-DMXP_TRACE (tools/build_variants.py stgcn_train_mx.hip trace:-DMXP_TRACE; RULGNN_LIB=variants/librulgnn_trace.so) workgroup 0 prints its
in-kernel timestamps per phase (100 MHz ticks between the marks of csrc/stgcn_train_mx.hip): profiles/r06_notes.md section 4."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_rul_benchmarking_amd.algorithms import ST_GCN
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = ST_GCN({"num_patch": 14, "patch_size": 30, "dropout": 0.2}, {"learning_rate": 1e-4, "weight_decay": 1e-4}, dev)
a.to(dev).train(); a.sync_loss = False
X, y = torch.rand(100, 14, 30, device=dev), torch.rand(100, 1, device=dev)
a.model.step_path = 4
for _ in range(6): a.update(X, y, 1); torch.cuda.synchronize()
