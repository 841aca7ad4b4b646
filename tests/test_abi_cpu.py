"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/rulgnn.h declares,
and its host-only entry points behave (no kernels are launched here)."""
import ctypes as C
import os
import re

from gnn_rul_benchmarking_amd import _lib, build, params as PL

from conftest import ROOT


def test_library_builds_and_loads():
    build.build()
    lib = _lib.load()
    assert lib.rulgnn_version() >= 100


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "rulgnn.h")).read()
    declared = set(re.findall(r"\b(rulgnn_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    raw = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name


def test_param_count_and_layout_agree_with_library():
    lib = _lib.load()
    for N, L in [(14, 2), (40, 2), (16, 3), (2, 1)]:
        assert lib.rulgnn_stgcn_param_count(N, L) == PL.param_count(N, L)
        for k in (1, 2, 3):
            assert lib.rulgnn_stgcn_param_count_order(N, L, k) == PL.param_count(N, L, k)
        lay = PL.live_param_layout(N, L)
        last_off, last_shape = list(lay.values())[-1]
        assert last_off + 1 == PL.param_count(N, L)
    assert PL.param_count(14, 2) == 1525       # 1,525 live parameters


def test_error_codes_have_messages():
    for code in (0, -1, -2, -3, -4, -5):
        assert "unknown" not in _lib.strerror(code)
    assert "unknown" in _lib.strerror(-99)


def test_shape_validation_without_gpu():
    lib = _lib.load()
    bad = _lib.StgcnShape(4, 8192, 32, 2, 1)          # beyond every path
    assert lib.rulgnn_stgcn_forward_f32(C.byref(bad), None, None, None, None, None, 0, None) == -2
    big = _lib.StgcnShape(4, 1024, 32, 2, 1)          # XJTU-sized num_patch: tiled path, needs a workspace
    assert lib.rulgnn_stgcn_forward_workspace_bytes(C.byref(big)) > 0
    assert lib.rulgnn_stgcn_forward_workspace_bytes(C.byref(_lib.StgcnShape(4, 14, 30, 2, 1))) == 0
    bad = _lib.StgcnShape(4, 14, 30, 2, 4)            # MPNN order: 1..3 (2, 3 on the row-mapped kernels, num_patch <= 64)
    assert lib.rulgnn_stgcn_forward_f32(C.byref(bad), None, None, None, None, None, 0, None) == -2
    assert lib.rulgnn_stgcn_forward_f32(C.byref(_lib.StgcnShape(4, 160, 16, 2, 2)), None, None, None, None, None, 0, None) == -2
    # k > 1 on a shape the fused kernels cannot hold (a 4096-point window; four layers) must not fall through to the tiled kernels
    for shp in (_lib.StgcnShape(4, 14, 4096, 2, 2), _lib.StgcnShape(4, 14, 30, 4, 2)):
        assert lib.rulgnn_stgcn_train_workspace_bytes(C.byref(shp)) == 0
    assert lib.rulgnn_stgcn_forward_f32(C.byref(_lib.StgcnShape(4, 14, 4096, 2, 2)), None, None, None, None, None, 0, None) == -2
    assert lib.rulgnn_stgcn_train_workspace_bytes(C.byref(_lib.StgcnShape(4, 14, 4096, 2, 1))) > 0
    assert lib.rulgnn_stgcn_forward_f32(C.byref(_lib.StgcnShape(4, 14, 30, 2, 0)), None, None, None, None, None, 0, None) == -1
    assert lib.rulgnn_stgcn_forward_f32(C.byref(_lib.StgcnShape(4, 14, 30, 2, 3)), None, None, None, None, None, 0, None) == -1   # valid shape, null pointers
    ok = _lib.StgcnShape(4, 14, 30, 2, 1)
    assert lib.rulgnn_stgcn_forward_f32(C.byref(ok), None, None, None, None, None, 0, None) == -1   # null pointers
    empty = _lib.StgcnShape(0, 14, 30, 2, 1)
    assert lib.rulgnn_stgcn_forward_f32(C.byref(empty), None, None, None, None, None, 0, None) == 0  # empty batch is a no-op


def test_single_hip_runtime_even_when_library_is_loaded_before_torch():
    """Regression: loading librulgnn.so before torch used to map /opt/rocm's libamdhip64 next to the
    one bundled with torch (two HIP runtimes -> every launch on a torch stream failed)."""
    import subprocess
    import sys
    code = "\n".join([
        "import sys, os; sys.path.insert(0, %r)" % ROOT,
        "from gnn_rul_benchmarking_amd import _lib",
        "_lib.load()",
        "import torch",
        "libs = {l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}",
        "print(len({os.path.realpath(p) for p in libs}))"])
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "1", out.stdout
