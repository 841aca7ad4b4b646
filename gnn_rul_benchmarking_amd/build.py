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
SOURCES = ["sgemm.hip", "sgemm_planes.hip", "peer_comm.hip", "stgcn_forward.hip", "stgcn_forward_mx.hip", "stgcn_train.hip", "stgcn_train_mx.hip", "stgcn_train_mxw.hip", "stgcn_tiled.hip", "stmsgcn.hip", "astgcnn.hip", "stconv.hip", "stgnn.hip", "rgcnu.hip", "stnet.hip", "sagcn.hip", "stagnn.hip", "fcstgnn.hip", "hagcn.hip", "bilstm.hip", "gru.hip", "metrics.hip", "optim.hip", "rulgnn_api.hip"]


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


OBJ_DIR = os.path.join(PKG_DIR, "..", "build", "obj")
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-c"]
# per-source additions.  stgcn_forward_mx.hip: the SLP vectoriser pairs scalar fp32 adds into v_pk_add_f32 and pays for it in
# v_mov register shuffles (same FLOP rate on gfx950): measured 80 -> 73 us at batch 65536.
EXTRA_FLAGS = {"stgcn_forward_mx.hip": ["-fno-slp-vectorize"], "stgcn_train_mx.hip": ["-fno-slp-vectorize"], "stgcn_train_mxw.hip": ["-fno-slp-vectorize"]}


def _compile_one(hipcc: str, src: str, obj: str, verbose: bool) -> None:
    cmd = [hipcc] + COMPILE_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + [src, "-o", obj + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed on {os.path.basename(src)}:\n" + res.stdout + res.stderr)
    os.replace(obj + ".tmp", obj)


def build(force: bool = False, verbose: bool = False, jobs: int | None = None) -> str:
    """Compile every HIP source to an object (in parallel, only what is out of date) and link one shared library;
    returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    missing = [s for s in srcs if not os.path.exists(s)]
    if missing:
        raise RuntimeError(f"missing HIP sources: {missing}")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [d for d in _deps() if not d.endswith(".hip")]
    newest_header = max(os.path.getmtime(h) for h in headers if os.path.exists(h))
    objs, todo = [], []
    for src in srcs:
        obj = os.path.join(OBJ_DIR, os.path.basename(src) + ".o")
        objs.append(obj)
        stale = force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header)
        if stale:
            todo.append((src, obj))
    with ThreadPoolExecutor(max_workers=jobs or min(6, os.cpu_count() or 1)) as pool:
        for fut in [pool.submit(_compile_one, hipcc, src, obj, verbose) for src, obj in todo]:
            fut.result()
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + objs + ["-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
