"""Times the scaled product on pre-split operands (csrc/sgemm_planes.hip) against the in-loop split of round 5 (sgemm_f16x2v_kernel) on the
tiled ST_GCN path's five large contractions (XJTU-SY 1024 x 32, batch 1024): python tools/time_sgemm_planes.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gnn_rul_benchmarking_amd import _lib

lib = _lib.load()
dev = "cuda:0"
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [("theta(A.X)     [10240,1024]x[1024,1024]  k,k", 10240, 1024, 1024, "k", "k", 0),
          ("d(A.X)=dH.theta [10240,1024]x[1024,1024] k,r", 10240, 1024, 1024, "k", "r", 0),
          ("d theta=dH^T.AX [1024,10240]x[10240,1024] r,r split-K", 1024, 1024, 10240, "r", "r", 1),
          ("square 4096 k,k", 4096 + 64, 4096, 4096, "k", "k", 0)]


def timed(call, n=20):
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, M, N, K, la, lb, split in SHAPES:
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
    new = lambda: lib.rulgnn_sgemm_scaled_ws_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, 0, pa.data_ptr(), 64,
                                                 pb.data_ptr(), 64, split, ws.data_ptr(), nb, C.byref(used), st())
    if split:
        nb2 = lib.rulgnn_sgemm_splitk_workspace_bytes(M, N, K)
        ws2 = torch.empty(nb2, dtype=torch.uint8, device=dev)
        old = lambda: lib.rulgnn_sgemm_splitk_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, None, ws2.data_ptr(), nb2, st())
    else:
        old = lambda: lib.rulgnn_sgemm_scaled_f32(A.data_ptr(), sAm, sAk, B.data_ptr(), sBn, sBk, Cm.data_ptr(), N, M, N, K, 0, pa.data_ptr(), 64,
                                                  pb.data_ptr(), 64, st())
    t_new, t_old = timed(new), timed(old)
    fl = 2.0 * M * N * K
    print(f"{name:58s} pre-split (used={used.value}) {t_new * 1e3:8.1f} us {fl / t_new / 1e9:7.1f} TF | round-5 kernel {t_old * 1e3:8.1f} us {fl / t_old / 1e9:7.1f} TF"
          + ("  (bf16x3 split-K: the old entry takes no scales)" if split else ""), flush=True)
