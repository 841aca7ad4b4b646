"""Training-step and eval-forward timing of FC_STGNN at the reference wirings (development aid).
    python tools/time_fcstgnn.py [name-filter]"""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import FC_STGNN
from gnn_rul_benchmarking_amd.hparams import get_hparams_class

dev = torch.device("cuda:0")
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, ds, did, bs in [("FD004 bs100", "CMAPSS", "FD004", 100), ("FD004 bs256", "CMAPSS", "FD004", 256),
                          ("FD001 bs100", "CMAPSS", "FD001", 100), ("FD002 bs100", "CMAPSS", "FD002", 100),
                          ("FD003 bs100", "CMAPSS", "FD003", 100), ("NCMAPSS bs100", "NCMAPSS", None, 100),
                          ("FD004 bs4096", "CMAPSS", "FD004", 4096)]:
    if flt not in name:
        continue
    cfg = get_hparams_class(ds)(did).alg_hparams["FC_STGNN"]
    torch.manual_seed(0)
    algo = FC_STGNN(cfg, {"learning_rate": 1e-4, "weight_decay": 1e-4}, dev)
    algo.to(dev).train()
    algo.sync_loss = False
    x, y = torch.rand(bs, cfg["num_node"], cfg["num_patch"] * cfg["patch_size"], device=dev), torch.rand(bs, 1, device=dev)
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
    print(f"{name:16s}: train {tr*1e3:9.3f} ms/step ({bs/tr:10.0f} samples/s)   eval {ev*1e3:9.3f} ms ({bs/ev:10.0f} samples/s)", flush=True)
