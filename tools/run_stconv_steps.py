"""ST_Conv training steps at one shape for a profiler run: python tools/run_stconv_steps.py [nodes] [batch] [steps]"""
import sys, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import ST_Conv
nodes, bs, steps = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 14), (2, 100), (3, 50)))
dev = torch.device("cuda:0")
torch.manual_seed(0)
algo = ST_Conv(dict(num_nodes=nodes, time_length=50, kernel_size=6), {"learning_rate": 1e-4, "weight_decay": 1e-4}, dev)
algo.to(dev).train()
algo.sync_loss = False
x, y = torch.rand(bs, nodes, 50, device=dev), torch.rand(bs, 1, device=dev)
for _ in range(steps): algo.update(x, y, 1)
torch.cuda.synchronize()
