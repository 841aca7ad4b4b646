import sys, os, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import ST_GCN
from benchlib.common import event_time_ms
dev = torch.device("cuda:0")
for aux in (True, False):
    torch.manual_seed(0)
    algo = ST_GCN({"num_patch": 1024, "patch_size": 32, "dropout": 0.3}, {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
    algo.to(dev).train(); algo.sync_loss = False
    algo.model.side_stream.enabled = aux
    x, y = torch.rand(1024, 1024, 32, device=dev), torch.rand(1024, 1, device=dev)
    ms = min(event_time_ms(lambda: algo.update(x, y, 1), 10, warm=3) for _ in range(4))
    print(os.environ.get("RULGNN_LIB", "default"), "aux" if aux else "single stream", round(ms, 4))
