"""Per-phase HIP-event timing of the ST_GCN training step (the chain RULGNN_STEP_AUTO picks) at several batch sizes.
    python tools/time_train_phases.py [batch ...]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gnn_rul_benchmarking_amd import _lib  # noqa: E402
from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model  # noqa: E402

N, P = int(os.environ.get("NP", 14)), int(os.environ.get("PS", 30))
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = ST_GCN_model(num_patch=N, patch_size=P, dropout=float(os.environ.get("DROPOUT", 0.2))).to(dev)
model.train()
lib = _lib.load()
L = model.num_layers
names = [f"F{i}" for i in range(2 * L)] + ["TOP"] + [f"G{2 * L - 1 - j}" for j in range(2 * L)]
for B in [int(b) for b in sys.argv[1:]] or [65536]:
    X = torch.rand(B, N, P, device=dev)
    y = torch.rand(B, 1, device=dev)
    model.fused_mse_step(X, y)
    shp = model._shape(B)
    a = model._train_args(shp, X.reshape(B, -1).contiguous(), y.reshape(-1).contiguous(), None, model._step)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def chain(evs=None):
        _lib.check(lib.rulgnn_stgcn_train_phase_f32(C.byref(shp), C.byref(a), -1, st()), "prepare")
        if evs is not None:
            evs[0].record()
        for ph in range(len(names)):
            _lib.check(lib.rulgnn_stgcn_train_phase_f32(C.byref(shp), C.byref(a), ph, st()), "phase")
            if evs is not None:
                evs[ph + 1].record()
    for _ in range(3):
        chain()
    torch.cuda.synchronize()
    iters, acc = 20, [0.0] * len(names)
    for _ in range(iters):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        chain(evs)
        torch.cuda.synchronize()
        for ph in range(len(names)):
            acc[ph] += evs[ph].elapsed_time(evs[ph + 1])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        model.fused_mse_step(X, y)
    e1.record()
    torch.cuda.synchronize()
    print(f"batch {B}: step {e0.elapsed_time(e1) / iters * 1e3:.1f} us | " + " ".join(f"{n} {t / iters * 1e3:.1f}" for n, t in zip(names, acc)) +
          f" | sum {sum(acc) / iters * 1e3:.1f}", flush=True)
