"""Time the fused eval forward kernels (exact fp32 vs matrix-core) at C-MAPSS shape; prints us per call and the HBM-roofline fraction
on SURVEY 8(d)'s algorithmic bytes (4 N P + 4 per sample)."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd import _lib, params as PL   # noqa: E402
from oracle import stgcn_oracle as O                       # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
import os
N, P, L = int(os.environ.get("NP", 14)), int(os.environ.get("PS", 30)), 2
prm_np, bn_np = PL.pack_numpy(O.random_params(N, L, seed=1), N, L)
prm, bn = torch.from_numpy(prm_np).to(dev), torch.from_numpy(bn_np).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B in [int(a) for a in sys.argv[1:]] or [4096, 65536, 1048576]:
    x = torch.rand(B, N * P, device=dev)
    out = torch.empty(B, device=dev)
    shp = _lib.StgcnShape(B, N, P, L, 1)
    res = {}
    for name, path in (("exact", _lib.EVAL_EXACT), ("mx", _lib.EVAL_MX))[:2 if N <= 47 else 1]:
        def call():
            _lib.check(lib.rulgnn_stgcn_forward_path_f32(C.byref(shp), x.data_ptr(), prm.data_ptr(), bn.data_ptr(), out.data_ptr(), None, 0,
                                                         path, st), name)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        reps = 50 if B <= 65536 else 10
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        us = float(np.median(ts))
        res[name] = us
        print(f"B={B:8d} {name:6s} {us:9.2f} us  {B / us:8.1f} M samples/s  {B * (4 * N * P + 4) / us / 1e6:6.3f} TB/s = {B * (4 * N * P + 4) / us / 8e6:.3f} of 8 TB/s",
              flush=True)
