import sys, time, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd.algorithms import RGCNU
dev = torch.device("cuda:0")
for nodes, bs in [(14, 256), (20, 256), (20, 100)]:
    torch.manual_seed(0)
    algo = RGCNU(dict(num_nodes=nodes, time_length=50, hidden_dim=32, encoder_hidden_dim=32, kernel_size=3, alpha=1), {"learning_rate": 1e-4, "weight_decay": 1e-4}, dev)
    algo.to(dev).train(); algo.sync_loss = False
    x, y = torch.rand(bs, nodes, 50, device=dev), torch.rand(bs, 1, device=dev)
    for _ in range(5): algo.update(x, y, 1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): algo.update(x, y, 1)
    torch.cuda.synchronize(); tr = (time.perf_counter() - t0) / 30
    print(f"RGCNU {nodes} nodes bs{bs}: {tr*1e3:.3f} ms/step ({bs/tr:.0f} samples/s)", flush=True)
