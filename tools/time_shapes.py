"""Training-step and eval-forward timing of ST_GCN at the reference-wired shapes (development aid)."""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import ST_GCN

dev = torch.device("cuda:0")
for name, cfg, seq, bs in [("CMAPSS-ext 14x30", {"num_patch": 14, "patch_size": 30, "dropout": 0.2}, 420, 100),
                           ("PHM2012 c1 40x64", {"num_patch": 40, "patch_size": 64, "dropout": 0.2}, 2560, 100),
                           ("PHM2012 c2 160x16", {"num_patch": 160, "patch_size": 16, "dropout": 0.2}, 2560, 100),
                           ("XJTU c1 1024x32", {"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, 32768, 100),
                           ("XJTU c2 2048x16", {"num_patch": 2048, "patch_size": 16, "dropout": 0.2}, 32768, 100),
                           ("XJTU c1 1024x32 bs1024", {"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, 32768, 1024)]:
    torch.manual_seed(0)
    algo = ST_GCN(cfg, {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
    algo.to(dev).train()
    algo.sync_loss = False
    x, y = torch.rand(bs, 1, seq, device=dev), torch.rand(bs, 1, device=dev)
    for _ in range(3): algo.update(x, y, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): algo.update(x, y, 1)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / n
    algo.eval()
    with torch.no_grad():
        for _ in range(3): algo.model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): algo.model(x)
        torch.cuda.synchronize(); ev = (time.perf_counter() - t0) / n
    print(f"{name:26s} batch {bs:5d}: train {tr*1e3:8.3f} ms/step ({bs/tr:10.0f} samples/s)   eval {ev*1e3:8.3f} ms ({bs/ev:10.0f} samples/s)")
