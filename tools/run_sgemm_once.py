"""One shape of the matrix-core GEMM a few times (for rocprofv3 / PMC passes): python tools/run_sgemm_once.py M N K [a_layout b_layout]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_rul_benchmarking_amd import _lib
M, N, K = (int(v) for v in sys.argv[1:4])
la, lb = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else ("k", "k")
lib = _lib.load()
A, B, Cm = torch.randn(M * K, device="cuda:0"), torch.randn(N * K, device="cuda:0"), torch.empty(M, N, device="cuda:0")
sAm, sAk = (K, 1) if la == "k" else (1, M)
sBn, sBk = (K, 1) if lb == "k" else (1, N)
for _ in range(5):
    lib.rulgnn_sgemm_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
torch.cuda.synchronize()
