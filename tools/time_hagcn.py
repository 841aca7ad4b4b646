"""Training-step and eval-forward timing of HAGCN at the reference wirings (development aid); splits the step into the
LSTM stack (torch / MIOpen) and the rest (HIP graph stack + fc + optimizer)."""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import HAGCN
from gnn_rul_benchmarking_amd.hparams import get_hparams_class

dev = torch.device("cuda:0")
for name, ds, did, bs in [("FD001 5x10 bs100", "CMAPSS", "FD001", 100), ("FD002 2x25 bs100", "CMAPSS", "FD002", 100),
                          ("FD004 1x50 bs100", "CMAPSS", "FD004", 100), ("FD004 1x50 bs256", "CMAPSS", "FD004", 256),
                          ("NCMAPSS 2x25 bs100", "NCMAPSS", None, 100)]:
    h = get_hparams_class(ds)(did)
    cfg = h.alg_hparams["HAGCN"]
    nodes = 20 if ds == "NCMAPSS" else 14
    torch.manual_seed(0)
    algo = HAGCN(cfg, h.train_params["HAGCN"], dev)
    algo.to(dev).train()
    algo.sync_loss = False
    x, y = torch.rand(bs, nodes, 50, device=dev), torch.rand(bs, 1, device=dev)
    for _ in range(3): algo.update(x, y, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 5
    for _ in range(n): algo.update(x, y, 1)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / n
    algo.eval()
    with torch.no_grad():
        for _ in range(2): algo.model(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): algo.model(x)
        torch.cuda.synchronize(); ev = (time.perf_counter() - t0) / n
        # graph stack alone on the LSTM output
        G = bs * cfg["num_patch"]
        nd = torch.rand(G, nodes, cfg["encoder_hidden_dim"], device=dev)
        for _ in range(2): algo.model.graph_stack(nd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): algo.model.graph_stack(nd)
        torch.cuda.synchronize(); gs = (time.perf_counter() - t0) / 20
    print(f"{name:20s}: train {tr*1e3:9.2f} ms/step ({bs/tr:8.0f} samples/s)  eval {ev*1e3:9.2f} ms ({bs/ev:8.0f} samples/s)  "
          f"graph stack alone {gs*1e3:7.3f} ms", flush=True)
