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
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X fp32 matrix / vector peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 matrix peak (MI355X_MICROARCH.md)
F16_MFMA_PEAK_TFLOPS = 2500.0     # dense f16 matrix peak (same pipe, same rate)
# fp32-class products out of split half-precision operands: the ceiling of the FORMULATION is the pipe's peak / the matrix instructions one
# product block costs (two f16 planes: hi hi + hi lo + lo hi = 3; three bf16 planes: 6)
F16X2_SPLIT_PEAK_TFLOPS = round(F16_MFMA_PEAK_TFLOPS / 3.0, 1)
BF16X3_SPLIT_PEAK_TFLOPS = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)




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


def mx_instruction_census(N, L=2):
    """Matrix instructions of the fused matrix-core forward per sample (csrc/stgcn_forward_mx.hip, DESIGN section 3.0: per 4-sample tile
    80 v_mfma_f32_16x16x32_f16 -- 10 per layer and sample for the three products of every split operand pair -- and 14 fp32 MFMAs of the
    Pearson Gram matrix; the wide kernel of 16 <= num_patch <= 47 runs the same chains over 2 | 3 column tiles)."""
    tiles = 1 if N <= 15 else (N + 15) // 16
    return {"f16_16x16x32_per_sample": 10 * L * tiles, "fp32_per_sample": 3.5 * tiles}


def compute_leg(flops_per_sample, samples_per_s, what, num_patch=None, passes=1.0):
    """The compute side beside an HBM fraction, on TWO bases, each named:
      * `frac` (peak_basis: fp32): USEFUL fp32-class FLOPs per second (every product counted once) against the fp32 vector / matrix peak
        (157.3 TFLOP/s) -- the machine-balance argument: at ~30 FLOP per byte these shapes sit above 157.3 TF / 8 TB/s ~ 20;
      * `mfma_pipe`: the f16 matrix instructions the kernels really ISSUE (three per product of split operands; `passes` forward-equivalents
        per sample: 1 for the eval forward, ~9 for the recomputing training chain) as f16 MFMA FLOPs per second against the dense f16 matrix
        peak (2500 TFLOP/s) -- the occupancy of the pipe they run on, from the instruction census; the PMC measurement of the same thing
        (SQ_VALU_MFMA_BUSY_CYCLES against the kernel's shader cycles x 1024 SIMDs) is `mfma_util` where a committed profile covers the kernel."""
    tf = flops_per_sample * samples_per_s / 1e12
    out = {"flops_per_sample": int(flops_per_sample), "achieved_tflops": round(tf, 2), "peak_tflops": FP32_MFMA_PEAK_TFLOPS,
           "peak_basis": "fp32 vector / matrix peak (157.3 TFLOP/s); useful fp32-class FLOPs, every product counted once",
           "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "counts": what}
    if num_patch is not None and num_patch <= 47:
        c = mx_instruction_census(num_patch)
        issued = c["f16_16x16x32_per_sample"] * passes * 2 * 16 * 16 * 32 * samples_per_s / 1e12
        out["mfma_pipe"] = {"issued_f16_mfma_tflops": round(issued, 1), "peak_tflops": F16_MFMA_PEAK_TFLOPS,
                            "peak_basis": "dense f16 matrix peak (2500 TFLOP/s, MI355X_MICROARCH.md): v_mfma_f32_16x16x32_f16 issued, 3 per split product",
                            "frac": round(issued / F16_MFMA_PEAK_TFLOPS, 4),
                            "f16_mfma_per_sample": round(c["f16_16x16x32_per_sample"] * passes, 1)}
    return out


def mfma_util_from_profile(kernel_substr, files=("r06_forward_bs1048576_sq_counters.txt", "r06_train_step_sq_counters.txt", "r06_sgemm_planes_sq_counters.txt",
                                                 "r06_stgcn_tiled_xjtu_sq_counters.txt", "r06_fcstgnn_sq_counters.txt")):
    """MFMA utilisation of a kernel from a committed PMC summary (profiles/r06_*_sq_counters.txt, tools/pmc_kernel_report.py format):
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs = the kernel's shader cycles) / 1024 SIMDs.
    None when no committed profile holds both counters for a kernel of that name."""
    import re as _re
    for name in files:
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        cur, vals = None, {}
        for line in open(path):
            if not line.startswith(" "):
                cur = line.strip()
                continue
            m = _re.match(r"\s+(\S+)\s+([0-9.eE+-]+)", line)
            if m and cur and kernel_substr in cur:
                vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
        for k, v in vals.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"] > 0:
                return {"mfma_util": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4),
                        "source": "profiles/" + name, "kernel": k[:120],
                        "formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)"}
    return None


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


