import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_rul_benchmarking_amd import _lib
lib = _lib.load(); dev = "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SH = [("STNet cheb1 fwd [18000,27]x[27,300]", 18000, 300, 27, "k", "r"),
      ("STNet cheb1 fwd padded K=28", 18000, 300, 28, "k", "r"),
      ("STNet cheb2 fwd [18000,900]x[900,200]", 18000, 200, 900, "k", "r"),
      ("STNet cheb3 fwd [18000,600]x[600,100]", 18000, 100, 600, "k", "r"),
      ("STNet cheb3 dx  [18000,100]x[100,600]", 18000, 600, 100, "k", "k"),
      ("STNet cheb2 dx  [18000,200]x[200,900]", 18000, 900, 200, "k", "k"),
      ("STNet cheb1 dx  [18000,300]x[300,27]", 18000, 27, 300, "k", "k")]
for name, M, N, K, la, lb in SH:
    A = torch.randn(M * K, device=dev); B = torch.randn(N * K, device=dev); Cm = torch.empty(M, N, device=dev)
    sAm, sAk = (K, 1) if la == "k" else (1, M)
    sBn, sBk = (K, 1) if lb == "k" else (1, N)
    call = lambda: lib.rulgnn_sgemm_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, 0, st())
    for _ in range(3): call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): call()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:48s} {ms*1e3:9.1f} us {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")
