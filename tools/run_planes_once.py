"""A few launches of the pre-split product at one shape (for the PMC passes): python tools/run_planes_once.py [M N K la lb split]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_rul_benchmarking_amd import _lib

a = sys.argv[1:]
M, N, K = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (10240, 1024, 1024)
la, lb = (a[3], a[4]) if len(a) >= 5 else ("k", "k")
split = int(a[5]) if len(a) >= 6 else 0
lib = _lib.load()
dev = "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M * K, device=dev)
B = torch.randn(N * K, device=dev)
Cm = torch.empty(M, N, device=dev)
pa, pb = torch.zeros(64, device=dev), torch.zeros(64, device=dev)
lib.rulgnn_absmax_partials_f32(A.data_ptr(), A.numel(), pa.data_ptr(), 64, st())
lib.rulgnn_absmax_partials_f32(B.data_ptr(), B.numel(), pb.data_ptr(), 64, st())
sAm, sAk = (K, 1) if la == "k" else (1, M)
sBn, sBk = (K, 1) if lb == "k" else (1, N)
nb = lib.rulgnn_sgemm_scaled_workspace_bytes(M, N, K, split)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
used = C.c_int32(-1)
for _ in range(6):
    lib.rulgnn_sgemm_scaled_ws_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, 0, pa.data_ptr(), 64, pb.data_ptr(), 64, split,
                                   ws.data_ptr(), nb, C.byref(used), st())
torch.cuda.synchronize()
print("used", used.value)
