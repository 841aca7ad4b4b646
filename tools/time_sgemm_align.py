"""The large-tile GEMM needs 16-byte aligned operands: what a misaligned weight pointer costs on STNet's products (development aid)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_rul_benchmarking_amd import _lib
lib = _lib.load(); dev = "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, M, N, K in [("cheb2 fwd [18000,900]x[900,200]", 18000, 200, 900), ("cheb3 fwd [18000,600]x[600,100]", 18000, 100, 600)]:
    for off in (0, 1):
        A = torch.randn(M * K, device=dev); Bf = torch.randn(N * K + 4, device=dev); Cm = torch.empty(M, N, device=dev)
        bp = Bf.data_ptr() + 4 * off
        call = lambda: lib.rulgnn_sgemm_f32(A.data_ptr(), K, 1, bp, 1, N, Cm.data_ptr(), N, M, N, K, 0, st())
        for _ in range(3): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:36s} B offset {off} floats: {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
