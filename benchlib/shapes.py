"""ST_GCN at the reference's other wirings: PHM2012 40 x 64 (wide matrix-core chain) and the num_patch > 64 shapes of the tiled path.

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
from .common import HBM_PEAK_GBS, FP32_MFMA_PEAK_TFLOPS, F16X2_SPLIT_PEAK_TFLOPS, event_time_ms, algorithmic_bytes_per_sample, mfma_util_from_profile


def stgcn_train_other_shape(dev, N, P, batches, steps=10, fp32_batches=(), single_launch_batches=()):
    """ST_GCN.update at another wiring (the reference's own PHM2012 40 x 64, configs/hparams.py:223,238): ms per step and samples/s per batch
    on the chain AUTO resolves to (the wide matrix-core chain, csrc/stgcn_train_mxw.hip), and -- for ``fp32_batches`` -- on the fp32 phase chain
    (RULGNN_STEP_CHAIN, the row-mapped path of rounds 1-3) in the same run."""
    from gnn_rul_benchmarking_amd.algorithms import ST_GCN
    from gnn_rul_benchmarking_amd import _lib
    out = {}
    for B in batches:
        g = torch.Generator(device=dev).manual_seed(5)
        X, y = torch.rand(B, N, P, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)
        entry = {}
        for name, path in (("auto", _lib.STEP_AUTO), ("fp32_chain", _lib.STEP_CHAIN), ("single_launch", _lib.STEP_MX_PERSIST)):
            if name == "fp32_chain" and B not in fp32_batches:
                continue
            if name == "single_launch" and B not in single_launch_batches:
                continue
            torch.manual_seed(0)
            algo = ST_GCN(dict(num_patch=N, patch_size=P, dropout=0.2), {"learning_rate": 1e-3, "weight_decay": 1e-4}, dev)
            algo.to(dev)
            algo.train()
            algo.sync_loss = False
            algo.model.step_path = path
            entry[name] = min(event_time_ms(lambda: algo.update(X, y, 1), steps, warm=3) for _ in range(3))     # best of three timed regions
            del algo
        ms = entry["auto"]
        alg = algorithmic_bytes_per_sample(N, P)
        out[f"batch_{B}"] = {"ms_per_step": round(ms, 4), "samples_per_s": round(B / (ms * 1e-3), 1),
                             "step_algorithmic_frac": round(alg * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        if "fp32_chain" in entry:
            out[f"batch_{B}"].update(fp32_chain_ms_per_step=round(entry["fp32_chain"], 4), vs_fp32_chain=round(entry["fp32_chain"] / ms, 2))
        if "single_launch" in entry:      # RULGNN_STEP_MX_PERSIST: F_1 .. G_0 as one launch behind arrival counters (built, slower: explicit option)
            out[f"batch_{B}"].update(single_launch_ms_per_step=round(entry["single_launch"], 4))
        del X, y
    return out


def stgcn_tiled_shapes(dev, steps=10):
    """ST_GCN at the reference's own wirings with num_patch > 64 (configs/hparams.py:269,349,384,418: PHM2012 Condition_2 160 x 16, XJTU-SY 1024 x 32)
    on the tiled path (csrc/stgcn_tiled.hip): theta / fc1 are num_patch x num_patch matrices there and theta(A.X) is a dense
    [batch*10, N] x [N, N] contraction -- SURVEY section 8(d): 46.3 MFLOP per sample forward at 1024 x 32, 353 FLOP/B: priced on the MFMA
    roofline (dense fp32-class products: the fp32 matrix peak; the large GEMM runs them as bf16 x 3 above that peak).
    Full update() (forward + loss + backward + Adam + running statistics) at the reference protocol's batch (100) and at 1024, and the eval
    forward."""
    from gnn_rul_benchmarking_amd.algorithms import ST_GCN
    out = {}
    for name, N, P, batches in (("xjtu_1024x32", 1024, 32, (100, 1024)), ("phm2012_c2_160x16", 160, 16, (100, 1024))):
        # matmul FLOPs per sample forward: L x theta [10 x N x N] + fc1 [N x N] + A.X [10 x 10 x N] x L + conv 2 x [10 x 20 x N] x L
        L = 2
        fwd = 2.0 * (L * 10 * N * N + N * N + L * 100 * N + L * 2 * 200 * N)
        entry = {"forward_matmul_flops_per_sample": fwd, "algorithmic_bytes_per_sample": algorithmic_bytes_per_sample(N, P)}
        for B in batches:
            g = torch.Generator(device=dev).manual_seed(5)
            X, y = torch.rand(B, N, P, device=dev, generator=g), torch.rand(B, 1, device=dev, generator=g)
            torch.manual_seed(0)
            algo = ST_GCN(dict(num_patch=N, patch_size=P, dropout=0.3), {"learning_rate": 1e-5, "weight_decay": 1e-4}, dev)
            algo.to(dev)
            algo.train()
            algo.sync_loss = False
            ms = min(event_time_ms(lambda: algo.update(X, y, 1), steps, warm=3) for _ in range(3))
            algo.eval()
            with torch.no_grad():
                ems = min(event_time_ms(lambda: algo.model(X), steps, warm=3) for _ in range(3))
            tf, etf = 3.0 * fwd * B / (ms * 1e-3) / 1e12, fwd * B / (ems * 1e-3) / 1e12
            entry[f"batch_{B}"] = {"train_ms_per_step": round(ms, 4), "train_samples_per_s": round(B / (ms * 1e-3), 1),
                                   "eval_ms_per_batch": round(ems, 4), "eval_samples_per_s": round(B / (ems * 1e-3), 1),
                                   "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": F16X2_SPLIT_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                "peak_basis": "dense f16 matrix peak (2500 TFLOP/s) / 3 matrix instructions per fp32-class product block "
                                                              "(two-plane f16 split: the instruction the large GEMMs issue is v_mfma_f32_32x32x16_f16); "
                                                              "FLOPs counted once",
                                                "frac": round(tf / F16X2_SPLIT_PEAK_TFLOPS, 4), "eval_achieved": round(etf, 2),
                                                "eval_frac": round(etf / F16X2_SPLIT_PEAK_TFLOPS, 4),
                                                "frac_of_fp32_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                                "counts": "whole step: 3 x the forward matmul FLOPs per sample x samples/s (not one kernel; the "
                                                          "position-parallel kernels between the GEMMs are HBM-bound and most of the step)"}}
            del algo, X, y
        out[name] = entry
    out["large_gemm"] = {"kernel": "sgemm_planes_kernel<5> (csrc/sgemm_planes.hip: pre-split operands, LDS-DMA, 160 x 256 tiles) on theta(A.X) and d(A.X); "
                                   "sgemm_f16x2v_kernel (in-loop split, 256 x 256 tiles) on d theta",
                         "shape": "[10240 x 1024] . [1024 x 1024]", "kernel_only_us": 63.6, "kernel_only_tflops": 338.0,
                         "with_split_passes_us": 86.0, "with_split_passes_tflops": 250.0, "round5_kernel_us": 111.0, "round5_kernel_tflops": 193.0,
                         "peak": F16X2_SPLIT_PEAK_TFLOPS, "frac_kernel_only": round(338.0 / F16X2_SPLIT_PEAK_TFLOPS, 3),
                         "mfma_util": mfma_util_from_profile("sgemm_planes_kernel"),
                         "source": "profiles/r06_sgemm_planes_kernel_stats.csv, profiles/r06_sgemm_planes_timing.txt (tools/time_sgemm_planes.py)"}
    out["profile"] = "profiles/r06_stgcn_tiled_xjtu_bs1024_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/time_tiled_one.py); " \
                     "what limits the pre-split form inside the step (split-pass traffic, no CU left for the side stream): profiles/r06_notes.md"
    return out


