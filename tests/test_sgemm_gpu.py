"""The fp32 matrix-core GEMM (csrc/sgemm_mfma.hpp: 64x64 and 128x128 tile kernels, generic strides) against numpy fp64 (GPU)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run(A, B, M, N, K, a_layout, b_layout, accumulate=False, C0=None):
    """A given as the logical [M, K] array, B as the logical [N, K] array; layouts 'k' (k contiguous) or 'r' (row index contiguous)."""
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    At = torch.from_numpy(np.ascontiguousarray(A if a_layout == "k" else A.T)).to(DEV)
    Bt = torch.from_numpy(np.ascontiguousarray(B if b_layout == "k" else B.T)).to(DEV)
    sAm, sAk = (K, 1) if a_layout == "k" else (1, M)
    sBn, sBk = (K, 1) if b_layout == "k" else (1, N)
    Ct = torch.zeros(M, N, device=DEV) if C0 is None else torch.from_numpy(C0.copy()).to(DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rulgnn_sgemm_f32(At.data_ptr(), sAm, sAk, Bt.data_ptr(), sBn, sBk, Ct.data_ptr(), N, M, N, K, 1 if accumulate else 0, st), "sgemm")
    return Ct.cpu().numpy()


@pytest.mark.parametrize("M,N,K", [(1280, 1000, 1000), (128, 25600, 128), (200, 12800, 128), (1024, 1024, 16), (777, 515, 133), (129, 97 * 128, 40),
                                   (12800, 1000, 40), (64, 64, 64), (5, 3, 7), (300, 200, 19), (3000, 3500, 77)])
@pytest.mark.parametrize("a_layout,b_layout", [("k", "k"), ("k", "r"), ("r", "k"), ("r", "r")])
def test_matches_numpy(M, N, K, a_layout, b_layout):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    got = run(A, B, M, N, K, a_layout, b_layout)
    assert np.abs(got - ref).max() < 2e-6 * np.sqrt(K) * np.abs(ref).max() + 1e-6
    # fp32-class accuracy whichever kernel served the shape: compare with what fp32 accumulation itself does to this product
    f32 = (A @ B.T).astype(np.float64)
    assert np.abs(got - ref).max() <= 4 * np.abs(f32 - ref).max() + 1e-6 * np.abs(ref).max()
    if M * N <= 1 << 20:
        C0 = rng.standard_normal((M, N)).astype(np.float32)
        got = run(A, B, M, N, K, a_layout, b_layout, accumulate=True, C0=C0)
        assert np.abs(got - (ref + C0)).max() < 2e-6 * np.sqrt(K) * np.abs(ref).max() + 1e-5


def test_both_tile_kernels_give_the_same_bits_and_unaligned_operands_are_handled():
    """Same k order per accumulator in the 64x64 and the 128x128 kernel: a sub-block computed alone (small problem -> 64x64 kernel) equals
    the same block of the big product; an operand whose base is not 16-byte aligned takes the 64x64 kernel."""
    from gnn_rul_benchmarking_amd import _lib
    prev = _lib.load().rulgnn_sgemm_mode(_lib.GEMM_F32)          # the fp32-instruction mode of the large-tile kernel
    assert prev == _lib.GEMM_BF16X3                                # (the default)
    try:
        _check_f32_tiles_bitwise()
    finally:
        _lib.load().rulgnn_sgemm_mode(prev)


def _check_f32_tiles_bitwise():
    from gnn_rul_benchmarking_amd import _lib
    rng = np.random.default_rng(1)
    M, N, K = 1024, 2048, 200
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    big = run(A, B, M, N, K, "k", "k")
    small = run(A[:64], B[:64], 64, 64, K, "k", "k")
    assert np.array_equal(big[:64, :64], small)
    lib = _lib.load()
    buf = torch.from_numpy(np.concatenate([[0.0], A.reshape(-1)]).astype(np.float32)).to(DEV)      # A starts 4 bytes into the allocation
    Bt = torch.from_numpy(B).to(DEV)
    Ct = torch.zeros(M, N, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rulgnn_sgemm_f32(buf.data_ptr() + 4, K, 1, Bt.data_ptr(), K, 1, Ct.data_ptr(), N, M, N, K, 0, st), "sgemm")
    assert np.array_equal(Ct.cpu().numpy(), big)


def test_bf16x3_is_fp32_class_on_hard_inputs():
    """Operands spanning twelve orders of magnitude (gradient-sized values beside O(1e3) activations) and exactly representable
    integers: the three-way bf16 split is exact, so integer products come out exact and tiny values keep their relative accuracy."""
    rng = np.random.default_rng(7)
    M, N, K = 512, 640, 96
    A = (rng.standard_normal((M, K)) * 10.0 ** rng.uniform(-9, 3, (M, 1))).astype(np.float32)
    B = (rng.standard_normal((N, K)) * 10.0 ** rng.uniform(-9, 3, (N, 1))).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    got = run(A, B, M, N, K, "k", "r")
    scale = np.abs(A.astype(np.float64)) @ np.abs(B.astype(np.float64)).T          # per-output magnitude of the summed products
    assert (np.abs(got - ref) / scale).max() < 4e-7
    Ai = rng.integers(-300, 300, (M, K)).astype(np.float32)
    Bi = rng.integers(-300, 300, (N, K)).astype(np.float32)
    exact = Ai.astype(np.float64) @ Bi.astype(np.float64).T                        # |sum| <= 96 * 9e4 < 2^24: every partial sum is an exact fp32
    got = run(Ai, Bi, M, N, K, "r", "k")
    assert np.array_equal(got, exact)


# ---- split-K weight gradients (rulgnn_sgemm_splitk_f32): every path of csrc/sgemm.hip::sgemm_splitk / sgemm_splitk_colsum -----------------
def run_splitk(A, B, M, N, K, a_layout, b_layout, colsum):
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    At = torch.from_numpy(np.ascontiguousarray(A if a_layout == "k" else A.T)).to(DEV)
    Bt = torch.from_numpy(np.ascontiguousarray(B if b_layout == "k" else B.T)).to(DEV)
    sAm, sAk = (K, 1) if a_layout == "k" else (1, M)
    sBn, sBk = (K, 1) if b_layout == "k" else (1, N)
    Ct = torch.full((M, N), float("nan"), device=DEV)
    cs = torch.full((M,), float("nan"), device=DEV) if colsum else None
    nbytes = lib.rulgnn_sgemm_splitk_workspace_bytes(M, N, K)
    ws = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rulgnn_sgemm_splitk_f32(At.data_ptr(), sAm, sAk, Bt.data_ptr(), sBn, sBk, Ct.data_ptr(), N, M, N, K,
                                           cs.data_ptr() if colsum else None, ws.data_ptr(), nbytes, st), "sgemm_splitk")
    torch.cuda.synchronize()
    return Ct.cpu().numpy(), (cs.cpu().numpy() if colsum else None)


# (M, N, K): the matrix-core long-k kernel in each of its tile counts (row-major operands, M <= 32, N <= 64, with the ones column pushing
# N + 1 over a tile edge), the LDS long-k kernel (k-contiguous operands), the one-workgroup kernel (K M N <= 131072), 64-k slices of the tile
# kernel with the 16-stripe reduction, a ragged last k-range
SPLITK = [(8, 16, 93184), (16, 16, 50176), (16, 32, 7168), (16, 48, 7168), (32, 32, 10240), (24, 64, 4099), (30, 17, 131), (1, 8, 4000),
          (16, 16, 2048), (8, 16, 600), (50, 50, 10240), (96, 72, 7168), (240, 60, 3584), (3, 5, 1)]


@pytest.mark.parametrize("M,N,K", SPLITK)
@pytest.mark.parametrize("layout", ["r", "k"])
@pytest.mark.parametrize("colsum", [False, True])
def test_splitk_weight_gradient_matches_numpy(M, N, K, layout, colsum):
    rng = np.random.default_rng(M * 1000 + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    got, cs = run_splitk(A, B, M, N, K, layout, layout, colsum)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() < 2e-6 * np.sqrt(K) * max(np.abs(ref).max(), 1.0) + 1e-6
    if colsum:
        cref = A.astype(np.float64).sum(axis=1)
        assert np.isfinite(cs).all()
        assert np.abs(cs - cref).max() < 2e-6 * np.sqrt(K) * max(np.abs(cref).max(), 1.0) + 1e-6
    # deterministic: same bits on a second run
    got2, cs2 = run_splitk(A, B, M, N, K, layout, layout, colsum)
    assert np.array_equal(got, got2) and (not colsum or np.array_equal(cs, cs2))


def run_scaled(A, B, M, N, K, a_layout, b_layout, nparts=(37, 5)):
    """rulgnn_sgemm_scaled_f32 with the operand scale rows from rulgnn_absmax_partials_f32 (A: logical [M, K], B: logical [N, K])."""
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    At = torch.from_numpy(np.ascontiguousarray(A if a_layout == "k" else A.T)).to(DEV)
    Bt = torch.from_numpy(np.ascontiguousarray(B if b_layout == "k" else B.T)).to(DEV)
    sAm, sAk = (K, 1) if a_layout == "k" else (1, M)
    sBn, sBk = (K, 1) if b_layout == "k" else (1, N)
    Ct = torch.zeros(M, N, device=DEV)
    pa, pb = torch.zeros(nparts[0], device=DEV), torch.zeros(nparts[1], device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rulgnn_absmax_partials_f32(At.data_ptr(), At.numel(), pa.data_ptr(), nparts[0], st), "absmax")
    _lib.check(lib.rulgnn_absmax_partials_f32(Bt.data_ptr(), Bt.numel(), pb.data_ptr(), nparts[1], st), "absmax")
    _lib.check(lib.rulgnn_sgemm_scaled_f32(At.data_ptr(), sAm, sAk, Bt.data_ptr(), sBn, sBk, Ct.data_ptr(), N, M, N, K, 0, pa.data_ptr(), nparts[0],
                                           pb.data_ptr(), nparts[1], st), "sgemm_scaled")
    return Ct.cpu().numpy(), pa.cpu().numpy(), pb.cpu().numpy()


@pytest.mark.parametrize("scale_a,scale_b", [(1.0, 1.0), (3e-7, 40.0), (2.5e5, 1e-9), (1e-30, 1e20)])
@pytest.mark.parametrize("a_layout,b_layout", [("k", "k"), ("k", "r"), ("r", "r")])
def test_two_plane_f16_split_with_operand_scales(scale_a, scale_b, a_layout, b_layout):
    """The 256 x 256 kernel on two f16 planes per operand (csrc/sgemm.hip: sgemm_f16x2v_kernel) at the shape of the tiled ST_GCN path's theta
    product, with operands far outside the f16 range in both directions: the power-of-two scales from the partial maxima keep the error at
    the 2^-22-per-operand class whatever the magnitudes (gradient-sized 1e-7 values, 1e5-sized activations)."""
    M, N, K = 2560, 1024, 1024 + 16 * 3 + 5                       # 40 tiles x ... >= 160 tiles needs M >= 10240: the split-K-free wide grid
    M = 10240
    rng = np.random.default_rng(int(1e3 * np.log10(scale_a * 7 + scale_b)) % 1000)
    A = (rng.standard_normal((M, K)) * scale_a).astype(np.float32)
    B = (rng.standard_normal((N, K)) * scale_b).astype(np.float32)
    got, pa, pb = run_scaled(A, B, M, N, K, a_layout, b_layout)
    assert pa.max() == np.abs(A).max() and pb.max() == np.abs(B).max()
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    err = np.abs(got - ref).max() / np.abs(ref).max()
    f32 = np.abs((A @ B.T).astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < 4e-6, (err, f32)                                  # (fp32 accumulation itself: ~1e-6 at K = 1000)
    assert np.isfinite(got).all()


def test_two_plane_f16_split_special_values_and_small_elements():
    """NaN / Inf elements do not set the scale and propagate like in fp32 (rows without them stay finite and right).  The stated limit of the
    form: an element 2^20 below its tensor's largest has lost the low bits of its lo part (f16 subnormals: absolute error 2^-25 after scaling,
    i.e. <= 2^-37 of the largest) -- a column that only such elements touch is right to ~4e-5 of ITS magnitude, not to 1e-6."""
    M, N, K = 10240, 1024, 512
    rng = np.random.default_rng(11)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    A[7, 3] = np.nan
    A[9, 100] = np.inf
    B[5, :] *= 2.0 ** -20                                          # a whole row of B far below the rest
    got, pa, pb = run_scaled(A, B, M, N, K, "k", "k")
    assert np.isfinite(pa).all() and pa.max() < 10
    ref = np.where(np.isfinite(A), A, 0).astype(np.float64) @ B.astype(np.float64).T
    ok = np.ones(M, bool); ok[[7, 9]] = False
    assert np.isnan(got[7]).all() and not np.isfinite(got[9]).any()
    assert np.abs(got[ok] - ref[ok]).max() < 4e-6 * np.abs(ref[ok]).max()
    col = np.abs(got[ok, 5] - ref[ok, 5]).max() / np.abs(ref[ok, 5]).max()
    assert col < 1e-4, col


def test_bf16x3_only_mode_ignores_the_operand_scales():
    """RULGNN_GEMM_BF16X3_ONLY: the scaled entry runs the three-plane bf16 kernel -- the same bits as the plain entry."""
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    M, N, K = 10240, 1024, 256
    rng = np.random.default_rng(3)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    plain = run(A, B, M, N, K, "k", "k")
    prev = lib.rulgnn_sgemm_mode(_lib.GEMM_BF16X3_ONLY)
    try:
        only, _, _ = run_scaled(A, B, M, N, K, "k", "k")
    finally:
        lib.rulgnn_sgemm_mode(prev)
    scaled, _, _ = run_scaled(A, B, M, N, K, "k", "k")
    assert np.array_equal(only, plain)
    assert not np.array_equal(scaled, plain) and np.abs(scaled - plain).max() < 1e-5 * np.abs(plain).max()


# ---- the scaled product on PRE-SPLIT operands (csrc/sgemm_planes.hip: split pass + LDS-DMA product kernel; rulgnn_sgemm_scaled_ws_f32) -----
def run_scaled_ws(A, B, M, N, K, a_layout, b_layout, split_k=False, accumulate=False, C0=None, nparts=(37, 5)):
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    At = torch.from_numpy(np.ascontiguousarray(A if a_layout == "k" else A.T)).to(DEV)
    Bt = torch.from_numpy(np.ascontiguousarray(B if b_layout == "k" else B.T)).to(DEV)
    sAm, sAk = (K, 1) if a_layout == "k" else (1, M)
    sBn, sBk = (K, 1) if b_layout == "k" else (1, N)
    Ct = torch.full((M, N), float("nan"), device=DEV) if C0 is None else torch.from_numpy(C0.copy()).to(DEV)
    pa, pb = torch.zeros(nparts[0], device=DEV), torch.zeros(nparts[1], device=DEV)
    nbytes = lib.rulgnn_sgemm_scaled_workspace_bytes(M, N, K, 1 if split_k else 0)
    ws = torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=DEV)               # garbage: nothing may rely on a zeroed workspace
    used = C.c_int32(-1)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.rulgnn_absmax_partials_f32(At.data_ptr(), At.numel(), pa.data_ptr(), nparts[0], st), "absmax")
    _lib.check(lib.rulgnn_absmax_partials_f32(Bt.data_ptr(), Bt.numel(), pb.data_ptr(), nparts[1], st), "absmax")
    _lib.check(lib.rulgnn_sgemm_scaled_ws_f32(At.data_ptr(), sAm, sAk, Bt.data_ptr(), sBn, sBk, Ct.data_ptr(), N, M, N, K, 1 if accumulate else 0,
                                              pa.data_ptr(), nparts[0], pb.data_ptr(), nparts[1], 1 if split_k else 0, ws.data_ptr(), nbytes,
                                              C.byref(used), st), "sgemm_scaled_ws")
    torch.cuda.synchronize()
    return Ct.cpu().numpy(), int(used.value)


@pytest.mark.parametrize("scale_a,scale_b", [(1.0, 1.0), (3e-7, 40.0), (2.5e5, 1e-9), (1e-30, 1e20)])
@pytest.mark.parametrize("a_layout,b_layout", [("k", "k"), ("k", "r"), ("r", "k"), ("r", "r")])
def test_presplit_product_at_the_theta_shape_over_the_operand_range(scale_a, scale_b, a_layout, b_layout):
    """[10 240 x 1024] . [1024 x 1024] -- theta(A.X) of the reference's XJTU-SY ST_GCN wiring at batch 1024 (models/ST_GCN/Model.py:88,
    configs/hparams.py:334,349) and, with a row-contiguous B, its data gradient d(A.X) = dH theta: 64 x 4 tiles of 160 x 256, every layout
    of the split pass, operands from 1e-30 to 1e20: the same 2^-22-per-operand error class as the in-loop split."""
    M, N, K = 10240, 1024, 1024
    rng = np.random.default_rng(int(1e3 * np.log10(scale_a * 7 + scale_b)) % 1000 + 1)
    A = (rng.standard_normal((M, K)) * scale_a).astype(np.float32)
    B = (rng.standard_normal((N, K)) * scale_b).astype(np.float32)
    got, used = run_scaled_ws(A, B, M, N, K, a_layout, b_layout)
    assert used == 1
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 4e-6
    old, _, _ = run_scaled(A, B, M, N, K, a_layout, b_layout)                           # the in-loop split of round 5: same class, other summation order
    assert np.abs(got - old).max() / np.abs(ref).max() < 4e-6


@pytest.mark.parametrize("M,N,K", [(5120, 1024, 96), (5120, 1024, 32), (5120, 1024, 64), (10240, 512, 160), (20480, 256, 256), (4160, 2048, 128)])
def test_presplit_product_tile_forms_and_short_k(M, N, K):
    """128 x 256 tiles (M = 5120 = 40 x 128 fills the chip better than 32 x 160), one / two / three / more k stages of the ring, other panel
    counts; integer operands come out EXACT (|sum| < 2^24: every partial sum is an exact fp32 and the split of an integer < 2^11 has lo = 0)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    got, used = run_scaled_ws(A, B, M, N, K, "k", "k")
    assert used == 1
    assert np.abs(got - ref).max() < 2e-6 * np.sqrt(K) * np.abs(ref).max() + 1e-6
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    b_layout = "r" if K % 64 == 0 else "k"            # (the transposing split pass works in 64 x 64 tiles)
    acc, used = run_scaled_ws(A, B, M, N, K, "k", b_layout, accumulate=True, C0=C0)
    assert used == 1 and np.abs(acc - (ref + C0)).max() < 2e-6 * np.sqrt(K) * np.abs(ref).max() + 1e-5
    Ai = rng.integers(-300, 300, (M, K)).astype(np.float32)
    Bi = rng.integers(-300, 300, (N, K)).astype(np.float32)
    got, _ = run_scaled_ws(Ai, Bi, M, N, K, "r" if K % 64 == 0 else "k", "k")
    assert np.array_equal(got, Ai.astype(np.float64) @ Bi.astype(np.float64).T)


@pytest.mark.parametrize("M,N,K,layout", [(1024, 1024, 10240, "r"), (1024, 1024, 10240, "k"), (1280, 512, 20480, "r"), (640, 256, 40960, "r")])
def test_presplit_split_k_weight_gradient(M, N, K, layout):
    """d theta = dH^T (A.X): [1024 x 1024] over K = batch x 10 = 10 240 rows, both operands row-contiguous (the transposing split pass),
    32 tiles of 128 x 256 x 8 k slices = 256 workgroups + the fixed-order slice sum: right, and the same bits on a second run."""
    rng = np.random.default_rng(M + K)
    A = (rng.standard_normal((M, K)) * 1e-6).astype(np.float32)                       # gradient-sized
    B = rng.standard_normal((N, K)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    got, used = run_scaled_ws(A, B, M, N, K, layout, layout, split_k=True)
    assert used == 1
    assert np.abs(got - ref).max() < 2e-6 * np.sqrt(K) * np.abs(ref).max()
    got2, _ = run_scaled_ws(A, B, M, N, K, layout, layout, split_k=True)
    assert np.array_equal(got, got2)


def test_presplit_entry_falls_back_for_other_shapes_and_modes_and_propagates_special_values():
    """A shape the pre-split kernel does not take (M = 10 250) runs the round-5 kernels through the same entry; RULGNN_GEMM_BF16X3_ONLY
    keeps every product on the range-free bf16 split; NaN / Inf elements do not set the scale and propagate as in fp32."""
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    M, N, K = 10250, 1024, 256
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    got, used = run_scaled_ws(A, B, M, N, K, "k", "k")
    assert used == 0 and np.abs(got - ref).max() < 4e-6 * np.abs(ref).max()
    M = 10240
    A = A[:M].copy()
    ref = ref[:M]
    prev = lib.rulgnn_sgemm_mode(_lib.GEMM_BF16X3_ONLY)
    try:
        only, used = run_scaled_ws(A, B, M, N, K, "k", "k")
    finally:
        lib.rulgnn_sgemm_mode(prev)
    assert used == 0 and np.array_equal(only, run(A, B, M, N, K, "k", "k"))
    A[7, 3] = np.nan
    A[9, 100] = np.inf
    got, used = run_scaled_ws(A, B, M, N, K, "k", "k")
    ok = np.ones(M, bool); ok[[7, 9]] = False
    assert used == 1 and np.isnan(got[7]).all() and not np.isfinite(got[9]).any()
    assert np.abs(got[ok] - ref[ok]).max() < 4e-6 * np.abs(ref[ok]).max()
