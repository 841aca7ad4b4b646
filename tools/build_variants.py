"""Build librulgnn variants that differ in -D flags of ONE source (development aid for kernel tuning).
    python tools/build_variants.py stgcn_train.hip name1:-DA=1,-DB=2 name2:-DA=3 ...
Writes variants/librulgnn_<name>.so (git-ignored, travels with gpurun); run with RULGNN_LIB=variants/librulgnn_<name>.so."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gnn_rul_benchmarking_amd.build import SOURCES, CSRC, COMPILE_FLAGS, EXTRA_FLAGS, _hipcc
OBJ = "/tmp/rulgnn_objs"
OUT = os.path.join(ROOT, "variants")
os.makedirs(OBJ, exist_ok=True); os.makedirs(OUT, exist_ok=True)
cflags = [f for f in COMPILE_FLAGS if f != "-c"]
target = sys.argv[1]
def cc(src, out, extra=()):
    subprocess.run([_hipcc()] + cflags + EXTRA_FLAGS.get(src, []) + list(extra) + ["-c", os.path.join(CSRC, src), "-o", out], check=True)
others = [s for s in SOURCES if s != target]
def obj(s): return os.path.join(OBJ, s.replace(".hip", ".o"))
todo = [s for s in others if not os.path.exists(obj(s)) or os.path.getmtime(obj(s)) < os.path.getmtime(os.path.join(CSRC, s))]
with ThreadPoolExecutor(4) as ex:
    list(ex.map(lambda s: cc(s, obj(s)), todo))
def variant(spec):
    name, _, defs = spec.partition(":")
    o = os.path.join(OBJ, f"{target}.{name}.o")
    cc(target, o, [d for d in defs.split(",") if d])
    lib = os.path.join(OUT, f"librulgnn_{name}.so")
    subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc"] + [obj(s) for s in others] + [o, "-o", lib], check=True)
    return lib
with ThreadPoolExecutor(4) as ex:
    for lib in ex.map(variant, sys.argv[2:]):
        print(lib)
