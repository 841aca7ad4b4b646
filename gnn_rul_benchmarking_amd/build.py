"""Build librulgnn.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m gnn_rul_benchmarking_amd.build

hipcc cross-compiles without a GPU; the resulting .so sits next to this file so that it travels
with the source tree (it is git-ignored, not gpurun-ignored)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "librulgnn.so")
SOURCES = ["stgcn_forward.hip", "stgcn_train.hip", "stgcn_tiled.hip", "stmsgcn.hip", "astgcnn.hip", "stconv.hip", "stgnn.hip", "fcstgnn.hip", "hagcn.hip", "bilstm.hip", "gru.hip", "metrics.hip", "optim.hip", "rulgnn_api.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def _deps() -> list[str]:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".h"))]
    deps.append(os.path.join(PKG_DIR, "..", "include", "rulgnn.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def needs_build() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    missing = [s for s in srcs if not os.path.exists(s)]
    if missing:
        raise RuntimeError(f"missing HIP sources: {missing}")
    tmp = LIB_PATH + ".tmp"
    cmd = [_hipcc()] + FLAGS + srcs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
