"""Eager vs hipGraph replay of Algorithm.update for the family configurations of bench.py: python tools/time_family_graphs.py FAMILY [steps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
from gnn_rul_benchmarking_amd import hparams as HP

fam = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ds, did, B, shape, _ = bench.FAMILY_CONFIGS[fam]
hp = HP.get_hparams_class(ds)(did)
dev = torch.device("cuda:0")
torch.manual_seed(0)
algo = get_algorithm_class(fam)(hp.alg_hparams[fam], hp.train_params[fam], dev)
algo.to(dev); algo.train(); algo.sync_loss = False
g = torch.Generator(device=dev).manual_seed(1)
Xs = [torch.rand(B, *shape, device=dev, generator=g) for _ in range(2)]
ys = [torch.rand(B, 1, device=dev, generator=g) for _ in range(2)]
def run(n):
    for i in range(n):
        last = algo.update(Xs[i % 2], ys[i % 2], 1)["loss"]
    torch.cuda.synchronize()
    return last
run(10)
t0 = time.perf_counter(); run(steps); e = (time.perf_counter() - t0) / steps
print(fam, "eager  ms/step", round(e * 1e3, 4))
if algo.supports_graphs:
    algo.enable_graphs()
    run(10)
    t0 = time.perf_counter(); l = run(steps); e = (time.perf_counter() - t0) / steps
    print(fam, "graphs ms/step", round(e * 1e3, 4), float(l))
