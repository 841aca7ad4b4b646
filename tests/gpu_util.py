"""Helpers for the -m gpu parity tests: call the C-ABI with torch device buffers."""
import ctypes as C
import glob
import os

import numpy as np
import torch

from gnn_rul_benchmarking_amd import _lib, params as PL

from conftest import GOLDEN


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    sd = {k[3:]: z[k] for k in z.files if k.startswith("sd:")}
    return z, sd


FB_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "stgcn_*x*_bs*.npz"))
                  if "train_curve" not in p)


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def shape_struct(batch, N, P, L=2, k=1):
    return _lib.StgcnShape(batch, N, P, L, k)


def abi_forward(x_np, flat_np, bn_np, N, P, L=2):
    """rulgnn_stgcn_forward_f32 on cuda:0; returns pred as numpy [B]."""
    lib = _lib.load()
    dev = torch.device("cuda:0")
    B = x_np.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(x_np.reshape(B, -1), np.float32)).to(dev)
    prm = torch.from_numpy(flat_np).to(dev)
    bn = torch.from_numpy(bn_np).to(dev)
    out = torch.full((B,), float("nan"), device=dev)
    shp = shape_struct(B, N, P, L)
    rc = lib.rulgnn_stgcn_forward_f32(C.byref(shp), x.data_ptr(), prm.data_ptr(), bn.data_ptr(), out.data_ptr(),
                                      stream_ptr())
    _lib.check(rc, "rulgnn_stgcn_forward_f32")
    torch.cuda.synchronize()
    return out.cpu().numpy()


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
