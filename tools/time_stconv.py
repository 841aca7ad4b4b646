"""Training-step and eval-forward timing of ST_Conv at the reference wirings (development aid)."""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import ST_Conv

dev = torch.device("cuda:0")
for name, nodes, bs in [("CMAPSS 14x50 bs100", 14, 100), ("NCMAPSS 20x50 bs100", 20, 100), ("CMAPSS 14x50 bs4096", 14, 4096),
                        ("CMAPSS 14x50 bs65536", 14, 65536)]:
    torch.manual_seed(0)
    algo = ST_Conv(dict(num_nodes=nodes, time_length=50, kernel_size=6), {"learning_rate": 1e-4, "weight_decay": 1e-4}, dev)
    algo.to(dev).train()
    algo.sync_loss = False
    x, y = torch.rand(bs, nodes, 50, device=dev), torch.rand(bs, 1, device=dev)
    for _ in range(3): algo.update(x, y, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): algo.update(x, y, 1)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / n
    algo.eval()
    with torch.no_grad():
        for _ in range(3): algo.model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): algo.model(x)
        torch.cuda.synchronize(); ev = (time.perf_counter() - t0) / n
    print(f"{name:24s}: train {tr*1e3:9.3f} ms/step ({bs/tr:11.0f} samples/s)   eval {ev*1e3:9.3f} ms ({bs/ev:11.0f} samples/s)", flush=True)
