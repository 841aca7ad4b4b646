"""-m gpu: the fused HIP eval forward (through the C-ABI) vs the reference's golden outputs and
vs the numpy oracle on seeded inputs.  Tolerance: 1e-4 relative fp32 (BASELINE.json north_star)."""
import numpy as np
import pytest

from gnn_rul_benchmarking_amd import params as PL
from oracle import stgcn_oracle as O

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.mark.parametrize("name", __import__("gpu_util").FB_CASES)
def test_forward_matches_reference_golden(name):
    import gpu_util as G
    z, sd = G.load_case(name)
    N, P = int(z["num_patch"]), int(z["patch_size"])
    flat, bn = PL.pack_numpy(sd, N, 2)
    pred = G.abi_forward(z["x"], flat, bn, N, P)
    ref = z["eval_pred"][:, 0]
    assert np.array_equal(np.isnan(pred), np.isnan(ref)), "NaN must appear in exactly the reference's places"
    ok = ~np.isnan(ref)
    assert G.rel_err(pred[ok], ref[ok]) < TOL


@pytest.mark.parametrize("N,P,B", [(14, 30, 1), (14, 30, 3), (14, 30, 4), (14, 30, 1027), (14, 50, 257),
                                   (16, 30, 65), (3, 6, 9), (5, 7, 33), (14, 31, 19), (16, 16, 40),
                                   (24, 20, 37), (32, 8, 11), (40, 64, 9), (64, 10, 6),
                                   # num_patch > 64: tiled path (PHM2012 Condition_2, XJTU-SY shapes)
                                   (65, 8, 5), (160, 16, 7), (200, 6, 3), (1024, 32, 4), (2048, 16, 2),
                                   # ... with batch x 10 rows enough for the pre-split product kernel (csrc/sgemm_planes.hip: 160 x 256 / 128 x 256 tiles)
                                   (1024, 32, 512), (256, 8, 1024)])
def test_forward_matches_oracle_seeded(N, P, B):
    import gpu_util as G
    rng = np.random.default_rng(N * 1000 + P * 10 + B)
    prm = O.random_params(N, 2, seed=B)
    x = rng.uniform(0, 1, (B, N, P)).astype(np.float32)
    flat, bn = PL.pack_numpy(prm, N, 2)
    pred = G.abi_forward(x, flat, bn, N, P)
    ref = O.forward(prm, x.astype(np.float64), N, P).pred[:, 0]
    assert np.isfinite(ref).all()
    assert G.rel_err(pred, ref) < TOL


def test_forward_full_size_batch_split_invariance():
    """BASELINE-size batch: the eval forward is per-sample, so any split of the batch must give
    bit-identical predictions (size-independent property; the oracle checks a slice)."""
    import gpu_util as G
    N, P, B = 14, 30, 65536 + 3
    rng = np.random.default_rng(7)
    prm = O.random_params(N, 2, seed=3)
    x = rng.uniform(0, 1, (B, N, P)).astype(np.float32)
    flat, bn = PL.pack_numpy(prm, N, 2)
    full = G.abi_forward(x, flat, bn, N, P)
    parts = np.concatenate([G.abi_forward(x[:1001], flat, bn, N, P), G.abi_forward(x[1001:], flat, bn, N, P)])
    assert np.array_equal(full, parts)
    ref = O.forward(prm, x[:512].astype(np.float64), N, P).pred[:, 0]
    assert G.rel_err(full[:512], ref) < TOL


def test_forward_rejects_unsupported():
    import ctypes as C
    import torch
    import gpu_util as G
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    t = torch.zeros(16, device="cuda:0")
    for shp in (G.shape_struct(4, 8192, 32), G.shape_struct(4, 14, 30, 2, k=4), G.shape_struct(4, 160, 16, 2, k=2)):
        rc = lib.rulgnn_stgcn_forward_f32(C.byref(shp), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 0, G.stream_ptr())
        assert rc == -2
    rc = lib.rulgnn_stgcn_forward_f32(C.byref(G.shape_struct(4, 1024, 32)), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 0, G.stream_ptr())
    assert rc == -3          # tiled path without its workspace
    shp = G.shape_struct(4, 14, 30)
    assert lib.rulgnn_stgcn_forward_f32(C.byref(shp), None, t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 0, G.stream_ptr()) == -1


def test_tiled_forward_with_plane_writing_aggregation_agrees_with_the_split_pass_form_and_places_nan_alike():
    """XJTU-SY 1024 x 32 at batch 512 (5 120 rows: the pre-split product kernel with the A planes written by the aggregation itself,
    scale from the bound 16 max |X|) against the same windows run as two batches of 256 (too few tiles: the in-loop split of round 5):
    same function to 1e-4, and a window with a constant patch -- NaN statistics, models/ST_GCN/Model.py:7-52 -- gives NaN for exactly
    that sample in both."""
    import gpu_util as G
    N, P, B = 1024, 32, 512
    rng = np.random.default_rng(99)
    prm = O.random_params(N, 2, seed=4)
    x = rng.uniform(0, 1, (B, N, P)).astype(np.float32)
    x[7, 100, :] = 0.25                                           # a constant patch
    x[300] *= 40.0                                                # one window far above the others (the scale bound follows it)
    flat, bn = PL.pack_numpy(prm, N, 2)
    fused = G.abi_forward(x, flat, bn, N, P)
    halves = np.concatenate([G.abi_forward(x[:256], flat, bn, N, P), G.abi_forward(x[256:], flat, bn, N, P)])
    assert np.isnan(fused[7]) and np.isnan(halves[7])
    ok = np.ones(B, bool); ok[7] = False
    assert np.isfinite(fused[ok]).all() and np.isfinite(halves[ok]).all()
    assert G.rel_err(fused[ok], halves[ok]) < TOL
    assert G.elem_gate(fused[ok], halves[ok], rtol=1e-4, floor=1e-5) <= 1.0
