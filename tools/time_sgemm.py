"""Times the fp32 matrix-core GEMM alone on the shapes of the SAGCN / STNet / tiled ST_GCN steps: python tools/time_sgemm.py"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_rul_benchmarking_amd import _lib

lib = _lib.load()
dev = "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [("SAGCN feature axis  [12800,1000]x[1000,1000]", 12800, 1000, 1000, "k", "k"),
          ("SAGCN node axis     [128,128]x[128,100000]", 128, 100000, 128, "k", "r"),
          ("SAGCN d input       [12800,1000]x[1000,1000]^T", 12800, 1000, 1000, "k", "r"),
          ("SAGCN weight grad   [1000,12800]x[12800,1000]", 1000, 1000, 12800, "r", "r"),
          ("STNet ChebNet 2     [18000,900]x[900,200]", 18000, 200, 900, "k", "r"),
          ("square 4096", 4096, 4096, 4096, "k", "k")]
for name, M, N, K, la, lb in SHAPES:
    A = torch.randn(M * K, device=dev)
    B = torch.randn(N * K, device=dev)
    Cm = torch.empty(M, N, device=dev)
    sAm, sAk = (K, 1) if la == "k" else (1, M)
    sBn, sBk = (K, 1) if lb == "k" else (1, N)
    call = lambda: lib.rulgnn_sgemm_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, 0, st())
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:52s} {ms * 1e3:9.1f} us  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s")
