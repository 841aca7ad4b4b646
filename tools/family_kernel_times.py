"""Per-kernel device time of one family's training step (HIP activity tracer): python tools/family_kernel_times.py SAGCN [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
from gnn_rul_benchmarking_amd import hparams as HP

fam = sys.argv[1]
ds, did, B, shape, _ = bench.FAMILY_CONFIGS[fam]
if len(sys.argv) > 2:
    B = int(sys.argv[2])
hp = HP.get_hparams_class(ds)(did)
torch.manual_seed(0)
algo = get_algorithm_class(fam)(hp.alg_hparams[fam], hp.train_params[fam], "cuda:0")
algo.to("cuda:0")
algo.train()
X, y = torch.rand(B, *shape, device="cuda:0"), torch.rand(B, 1, device="cuda:0")
for _ in range(3):
    algo.update(X, y, 1)
kt = bench.kernel_times(lambda i: algo.update(X, y, 1))
tot = sum(c * u for c, u in kt.values())
print(f"{fam} batch {B}: {tot:.1f} us of kernels per step, {sum(c for c, _ in kt.values()):.0f} launches")
for k, (c, u) in sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:25]:
    print(f"  {c * u:9.1f} us  {c:5.1f} x {u:8.1f}  {bench.kernel_short_name(k)}")
