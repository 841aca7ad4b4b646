import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd import _lib, params as PL
from oracle import stgcn_oracle as O
lib = _lib.load(); dev = torch.device("cuda:0")
N, P, L = 14, 30, 2
prm_np, bn_np = PL.pack_numpy(O.random_params(N, L, seed=1), N, L)
prm, bn = torch.from_numpy(prm_np).to(dev), torch.from_numpy(bn_np).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for B in (65536, 262144, 1048576):
    x = torch.rand(B, N * P, device=dev); out = torch.empty(B, device=dev)
    shp = _lib.StgcnShape(B, N, P, L, 1)
    def call():
        _lib.check(lib.rulgnn_stgcn_forward_path_f32(C.byref(shp), x.data_ptr(), prm.data_ptr(), bn.data_ptr(), out.data_ptr(), None, 0, _lib.EVAL_MX, st), "mx")
    for reps in (5, 20, 100, 1000, 20, 5):
        if reps * B > 3e8: continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): call()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        print(f"B={B} reps={reps:5d}: " + " ".join(f"{t:7.2f}" for t in ts), flush=True)
