"""Constants and timing helpers shared by the benchmark legs.

Part of the benchmark harness behind bench.py (the driver's contract lives there).  The oracle imports in here are the `cpu_baseline` /
`rmse` checker legs only -- never the thing measured."""
from __future__ import annotations

import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
NUM_PATCH = 14                 # C-MAPSS: 14 sensors kept (Data_read_CMAPSS.py:76)
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 matrix peak (MI355X_MICROARCH.md)




def event_time_ms(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def algorithmic_bytes_per_sample(N, P):
    """the window is read once and one float is written; the 6.1 KB of weights amortise over the batch."""
    return 4 * N * P + 4


def forward_flops_per_sample(N, P, L=2):
    """Useful FLOPs of one ST_GCN forward per sample (matmul / conv FLOPs counted with FlopCounterMode on the reference
    -- 41,660 at 14 patches, 157,880 at 40 -- plus ~20 N P for the patch statistics); other shapes: the same terms by formula."""
    mm = {14: 41660, 40: 157880}.get(N)
    if mm is None or L != 2:
        per_layer = 2 * 10 * 10 * N + 2 * 10 * N * N + 2 * (2 * 10 * 10 * 2 * N)
        mm = L * per_layer + 2 * 2 * 10 * 10 * N + 2 * N * N + 2 * N
    return mm + 20 * N * P


def compute_leg(flops_per_sample, samples_per_s, what):
    """The compute-side roofline beside an HBM fraction: useful FLOPs per second against the fp32 matrix / vector peak (157.3 TFLOP/s).
    At ~30 FLOP per byte these shapes sit above the fp32 machine balance (157.3 TF / 8 TB/s ~ 20): the HBM fraction is the contract, this is
    the bound the kernels actually run against."""
    tf = flops_per_sample * samples_per_s / 1e12
    return {"flops_per_sample": int(flops_per_sample), "achieved_tflops": round(tf, 2), "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
            "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "counts": what}


def kernel_short_name(name):
    """'void rulgnn::(anonymous namespace)::fc_graph_bwd_kernel<2>(rulgnn::...)' -> 'fc_graph_bwd_kernel<2>'"""
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rulgnn::", "").strip()


def kernel_times(step_fn, steps=10):
    """Per-kernel device time of `steps` calls of step_fn, measured live through the HIP activity tracer (torch.profiler / roctracer;
    it records every kernel this process launches, the library's included): {kernel name: (launches per step, average us)}."""
    from torch.profiler import ProfilerActivity, profile

    def once():
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(steps):
                step_fn(i)
            torch.cuda.synchronize()
        out = {}
        for e in prof.key_averages():
            if e.device_time_total > 0 and e.count > 0:
                out[e.key] = (e.count / steps, e.device_time_total / e.count)
        return out

    # the tracer now and then hands back a fraction of a short run's records (seen with two streams: 0.2 launches per step of a kernel
    # that runs once per step): a capture in which a kernel's count is not a whole number of launches per step is taken again
    out = {}
    for _ in range(3):
        out = once()
        if out and all(abs(c - round(c)) < 1e-9 and c >= 1 for c, _ in out.values()):
            break
    return out


