"""Training-step time of a family, eager vs hipGraph replay, second stream on / off (development aid).
usage: python tools/time_graphs.py FC_STGNN|ASTGCNN|ST_GCN [batch]"""
import sys, time
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd import algorithms as A
from gnn_rul_benchmarking_amd.hparams import get_hparams_class

fam = sys.argv[1]
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda:0")
ds, did, shape = {"FC_STGNN": ("CMAPSS", "FD004", (14, 50)), "ASTGCNN": ("NCMAPSS", None, (20, 50)), "ST_GCN": ("CMAPSS", "FD004", (14, 30))}[fam]
h = get_hparams_class(ds)(did) if fam != "ST_GCN" else get_hparams_class(ds)(did, window=30)
for graphs in (False, True):
    for aux in (False, True):
        torch.manual_seed(0)
        algo = getattr(A, fam)(h.alg_hparams[fam], h.train_params[fam], dev)
        algo.to(dev).train()
        algo.sync_loss = False
        if hasattr(algo.model, "side_stream"):
            algo.model.side_stream.enabled = aux
        elif aux:
            continue
        if graphs:
            algo.enable_graphs()
        x, y = torch.rand(bs, *shape, device=dev), torch.rand(bs, 1, device=dev)
        for _ in range(8): out = algo.update(x, y, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 200
        for _ in range(n): out = algo.update(x, y, 1)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{fam} bs {bs} graphs={graphs} aux={aux}: {dt * 1e3:.3f} ms/step  loss {float(out['loss']):.6f}", flush=True)
