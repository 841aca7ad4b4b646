"""Training-step and eval-forward timing of STMSGCN at the reference-wired shapes (development aid).
    python tools/time_stmsgcn.py [name-filter]"""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import STMSGCN

dev = torch.device("cuda:0")
D = {"gcn_dims": [16, 64, 16, 1], "gru_hidden_dim": 8}
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, cfg, bs in [("PHM2012 c1 160x16 n2", dict(num_patch=160, patch_size=16, interval=6, band_width=5, **D), 100),
                      ("PHM2012 c2 128x20 n6", dict(num_patch=128, patch_size=20, interval=2, band_width=3, **D), 100),
                      ("XJTU c1 256x128 n25", dict(num_patch=256, patch_size=128, interval=3, band_width=5, **D), 100),
                      ("XJTU c2 128x256 n25", dict(num_patch=128, patch_size=256, interval=6, band_width=10, **D), 100),
                      ("XJTU c1 256x128 n25 bs128", dict(num_patch=256, patch_size=128, interval=3, band_width=5, **D), 128),
                      ("XJTU c1 256x128 n25 bs1024", dict(num_patch=256, patch_size=128, interval=3, band_width=5, **D), 1024)]:
    if flt not in name:
        continue
    torch.manual_seed(0)
    algo = STMSGCN(cfg, {"learning_rate": 1e-4, "weight_decay": 0.0}, dev)
    algo.to(dev).train()
    algo.sync_loss = False
    x, y = torch.rand(bs, 1, cfg["num_patch"] * cfg["patch_size"], device=dev), torch.rand(bs, 1, device=dev)
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
    print(f"{name:28s} batch {bs:5d}: train {tr*1e3:9.3f} ms/step ({bs/tr:10.0f} samples/s)   eval {ev*1e3:9.3f} ms ({bs/ev:10.0f} samples/s)", flush=True)
