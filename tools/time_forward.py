"""Ad-hoc timing of the fused eval forward (development aid; bench.py is the judged harness)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd import _lib, params as PL
from oracle import stgcn_oracle as O

lib = _lib.load()
N, P, L = 14, int(sys.argv[2]) if len(sys.argv) > 2 else 30, 2
for B in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["65536", "1048576"])]:
    prm = O.random_params(N, L, seed=1)
    flat, bn = PL.pack_numpy(prm, N, L)
    dev = torch.device("cuda:0")
    x = torch.rand(B, N * P, device=dev)
    fp, bp = torch.from_numpy(flat).to(dev), torch.from_numpy(bn).to(dev)
    out = torch.empty(B, device=dev)
    shp = _lib.StgcnShape(B, N, P, L, 1)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        _lib.check(lib.rulgnn_stgcn_forward_f32(C.byref(shp), x.data_ptr(), fp.data_ptr(), bp.data_ptr(), out.data_ptr(), None, 0, st), "fwd")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20
    e0.record()
    for _ in range(iters): run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    byts = B * (N * P * 4 + 4)
    print(f"B={B} P={P}: {ms*1e3:.1f} us/launch  {B/ms/1e3:.1f} Msamples/s  {byts/ms/1e6:.1f} GB/s algorithmic ({byts/ms/1e6/8000*100:.1f}% of 8 TB/s)")
