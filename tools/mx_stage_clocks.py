"""Stage clocks of the fused ST_GCN eval forward (development aid).
    python tools/build_variants.py stgcn_forward_mx.hip clk:-DMX_STAGE_CLOCKS=1 clk2:-DMX_STAGE_CLOCKS=2 clk3:-DMX_STAGE_CLOCKS=3
    (1: a stamp per stage, 2: two per iteration, 3: wavefront lifetime only -- the one that does not perturb the loop)
    RULGNN_LIB=variants/librulgnn_clk.so python tools/mx_stage_clocks.py [batch]
Prints, per stage of a 4-sample tile, the s_memtime ticks a wavefront spends between the stage's boundaries (issue time: a stall is
charged to the stage whose instruction waits), averaged over all tiles of all wavefronts, beside the instruction counts of the ISA."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
from gnn_rul_benchmarking_amd import _lib, params as PL
from oracle import stgcn_oracle as O
lib = _lib.load(); dev = torch.device("cuda:0")
N, P, L = 14, 30, 2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1048576
prm_np, bn_np = PL.pack_numpy(O.random_params(N, L, seed=1), N, L)
prm, bn = torch.from_numpy(prm_np).to(dev), torch.from_numpy(bn_np).to(dev)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.rand(B, N * P, device=dev); out = torch.empty(B, device=dev)
shp = _lib.StgcnShape(B, N, P, L, 1)
def call():
    _lib.check(lib.rulgnn_stgcn_forward_path_f32(C.byref(shp), x.data_ptr(), prm.data_ptr(), bn.data_ptr(), out.data_ptr(), None, 0, _lib.EVAL_MX, st), "mx")
fn = getattr(lib, "rulgnn_debug_mx_stage_clocks", None)
if fn is None:
    sys.exit("this library has no stage clocks: build the variant with -DMX_STAGE_CLOCKS")
buf = (C.c_ulonglong * 22)()
import time
t0 = time.time()
while time.time() - t0 < 1.5:          # clocks ramp for tens of milliseconds after an idle period
    for _ in range(20): call()
    torch.cuda.synchronize()
fn(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 10
print(f"B={B}: {us:.1f} us per launch (this build, the ten launches the clocks below are from)")
fn(buf, 0)
names = ["wait: tile landed (vmcnt/lgkmcnt 0) + pending store", "patch load, next-tile request, statistics", "Pearson: LDS round trip, centre/normalise, 14 f32 MFMA",
         "adjacency split + D-layout reads", "L0 T", "L0 Hp", "L0 conv1 operands (leaky, split, shift tile)", "L0 conv1 MFMA", "L0 conv2 operands", "L0 conv2 MFMA",
         "L0 residual + L1 T", "L1 Hp", "L1 conv1 operands", "L1 conv1 MFMA", "L1 conv2 operands", "L1 conv2 MFMA", "L1 residual", "head"]
tiles = buf[20]
tot = sum(buf[i] for i in range(18))
waves = buf[18]
print(f"{tiles} tile passes by {waves} wavefronts in 10 launches; {tot / max(tiles, 1):.0f} ticks per tile and wavefront inside the loop;")
print(f"a wavefront lives {buf[19] / waves:.0f} shader ticks = {buf[21] / waves:.0f} ticks of the 100 MHz wall clock from entry to its last tile: "
      f"{buf[19] / max(buf[21], 1) * 100:.0f} MHz under THIS kernel ({buf[21] / waves / 100:.1f} us of the launch's {us:.1f})")
if tot:
    for i, n in enumerate(names):
        print(f"  {i:2d} {n:60s} {buf[i] / tiles:8.1f}  {100.0 * buf[i] / tot:5.1f} %")
else:
    print(f"lifetime form: {B // 4 / waves * 10:.1f} tiles per wavefront -> {buf[19] / waves / (B // 4 / waves * 10):.0f} ticks per tile and wavefront, prologue included")
