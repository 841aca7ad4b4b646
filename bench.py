#!/usr/bin/env python
"""bench.py -- headline benchmark: ST_GCN training samples/s on C-MAPSS FD004-shaped batches.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one ``ST_GCN.update`` (reference algorithms/algorithms.py:481-490): train-mode forward
(BatchNorm batch statistics, dropout), MSE, backward, Adam -- on one synthetic batch already
resident in HBM.  For N > 1 each rank owns a fixed per-GPU batch (weak scaling) and the gradient
bucket is all-reduced over RCCL once per step (gnn_rul_benchmarking_amd/dp.py).

`--gpus N` from a bare shell (no WORLD_SIZE in the environment) re-launches itself under torch.distributed.run with N
processes on 127.0.0.1.  The step loop is timed `--reps` times (each repetition = EXACTLY `--steps` steps between barriers +
synchronize, max over ranks) and the MEDIAN repetition is reported.  For N > 1 the weak-scaling run (fixed per-GPU batch) is the
headline and a strong-scaling run (the same global batch as N = 1, split over the ranks) is reported beside it.

Rank 0 prints ONE JSON line.  Besides the driver contract it carries
  roofline       dominant kernel of the step (picked live by HIP-event timing of each phase kernel).  `achieved` / `frac` are on
                 SURVEY section 8(d)'s ALGORITHMIC bytes (4 N P + 4 = 1684 B per sample at 14x30: the window in, one float out);
                 `frac_traffic` is the same launch priced on the bytes the phase really moves (its inter-phase tensors included);
                 `step_algorithmic_frac` prices the whole step on the algorithmic bytes, `traffic_over_algorithmic` = PMC bytes of
                 all phases / algorithmic bytes
  roofline_forward   the fused eval forward kernel (the north-star kernel), same measurement, at the bench batch and at 1M
  cpu_baseline   the reference's CPU path restated on torch-CPU (oracle/stgcn_torch_cpu.py, pinned to the reference's own
                 training curve) timed on this box's host cores at 1 and all threads, 20 warm-up + 100 iterations
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
NUM_PATCH = 14                 # C-MAPSS: 14 sensors kept (Data_read_CMAPSS.py:76)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the timed step loop; the median is reported")
    ap.add_argument("--batch", type=int, default=65536, help="per-GPU batch (weak scaling) = global batch of the strong-scaling run")
    ap.add_argument("--sync-bn", action="store_true",
                    help="data parallel with synchronised BatchNorm: the N-GPU step is the 1-GPU function of the global batch "
                         "(8 more 160-byte all-reduces per step); default = local statistics like torch DDP")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both",
                    help="N > 1: weak = fixed per-GPU batch (headline), strong = fixed global batch split over the ranks")
    ap.add_argument("--patch-size", type=int, default=30, help="window length (BASELINE.json: 30)")
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true", help="skip the other four BASELINE.json configurations in the default line")
    ap.add_argument("--no-rmse", action="store_true", help="skip the teacher-task RMSE leg (the other half of BASELINE.json's metric)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--isolated-phases", action="store_true", help="also time every phase kernel re-run back to back (MALL-warm)")
    ap.add_argument("--sync-loss", action="store_true", help="loss.item() every step like the reference")
    ap.add_argument("--force-dp", action="store_true", help="use the data-parallel step even for world_size 1 (exercises RCCL)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="--family FC_STGNN only: bf16 = BASELINE.json's 'FC_STGNN ... bf16' variant (bf16 operands on the row-projection "
                         "matrix-core GEMMs, fp32 accumulate / BatchNorm / graphs / weight gradients); reported separately, it does not meet 1e-4")
    ap.add_argument("--family", default="ST_GCN", choices=["ST_GCN", "ASTGCNN", "FC_STGNN", "STMSGCN", "HAGCN", "STGNN", "RGCNU", "STNet", "SAGCN", "STAGNN"],
                    help="ST_GCN (default) is the headline benchmark; the others run the same contract on the SURVEY section 8d "
                         "configuration of that model family")
    return ap.parse_args()


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


def phase_names(L):
    return [f"F{i}" for i in range(2 * L)] + ["TOP"] + [f"G{2 * L - 1 - j}" for j in range(2 * L)]


def phase_bytes_per_sample(name, N, P, L, chain="mx"):
    """Algorithmic HBM bytes per sample of one phase kernel.  T = one [10, N] fp32 state tensor per sample
    (packed: only the N patch lanes of a row are stored) = 560 B at N = 14; t = d X_L as (value, arg-max channel) per (sample, patch).

    chain "mx" (matrix-core chain, csrc/stgcn_train_mx.hip: every phase recomputes from the layer input): A = the adjacency's 55
    unique entries = 220 B; what crosses HBM between phases is X_l, the gated x-hat Q_l of BatchNorm 2l-1 (l >= 1), d(x0 + H) and d X_l.
    chain "fp32" (row-mapped chain, csrc/stgcn_train.hip): A = 400 B lane layout, plus the saved H, z1, o0, z2 of every layer."""
    T = 10 * N * 4
    TOPG = 2 * N * 4                                   # d X_L: (value, arg-max channel) per (sample, patch) instead of ten rows
    if chain == "mx":
        A = 55 * 4
        if name == "F0":
            return N * P * 4 + T + A                   # read the window; write X_0, adjacency
        if name == "TOP":
            return T + A + TOPG + 8                    # X_{L-1}, A; write d X_L; y in, pred out
        i = int(name[1:])
        l, blk = divmod(i, 2)
        din = TOPG if l == L - 1 else T
        if name[0] == "F":
            return (T + A) if blk == 1 else (T + A + 2 * T)        # F_{2l+1}: X_l, A;  F_{2l}, l >= 1: X_{l-1}, A; write X_l, Q_l
        if blk == 1:
            return T + A + din + T                     # G_{2l+1}: X_l, A, d X_{l+1}; write d(x0 + H)
        return (T + A + T) if l == 0 else (T + A + T + din + T + T)   # G_{2l}: X_l, A, d(x0 + H) (+ d X_{l+1}, Q_l in; d X_l out)
    A = 10 * 10 * 4
    if name == "F0":
        return N * P * 4 + 3 * T + A                   # read the window; write X0, adjacency, H, z1
    if name == "TOP":
        return 3 * T + TOPG + 8                        # X_{L-1}, o0, z2; write dX_L; y in, pred out
    i = int(name[1:])
    l, blk = divmod(i, 2)
    din = TOPG if l == L - 1 else T                    # the gradient entering the top layer is the sparse one
    if name[0] == "F":
        if blk == 1:
            return 4 * T                               # F_{2l+1}: H, z1; write o0, z2
        return 6 * T + A                               # F_{2l}, l >= 1: X_{l-1}, A, o0, z2; write X_l, H, z1
    if blk == 1:
        return 4 * T + din                             # G_{2l+1}: z1, o0, z2, dX_{l+1}; write d(x0+H)
    return (4 * T + A) if l == 0 else (6 * T + A + din)   # G_{2l}: X_l, A, H, z1, d(x0+H) (+ dX in/out, z2 of the layer below)


def _traffic_profile(chain="mx"):
    """The committed PMC summary (FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh) of the given phase chain ("mx" = the
    matrix-core chain of round 4, "fp32" = the row-mapped chain; summaries without a "chain" entry predate the former): newest round first."""
    for name in ("r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_g_hbm_traffic.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
            if t.get("chain", "fp32") != chain:
                continue
            t["file"] = "profiles/" + name
            return t
        except Exception:
            continue
    return None


def measured_traffic(kernel_key, N, P, B, chain="mx"):
    """HBM bytes per launch from the committed PMC summary, scaled to this batch; None when the profiled workload does not match."""
    t = _traffic_profile(chain)
    if not t:
        return None
    w = t["workload"]
    if (w["num_patch"], w["patch_size"]) != (N, P) or kernel_key not in t["kernels"]:
        return None
    return round(t["kernels"][kernel_key]["hbm_bytes_per_sample"] * B)


def forward_traffic(N, P, B):
    """HBM bytes per launch of the fused eval forward from its own PMC passes (tools/profile_forward.sh ->
    profiles/r0N_forward_bs<B>_hbm_traffic.json: the forward profiled ALONE -- the EVAL entry of the train-step profile also counts
    the bench's other launches of that name), or None when this batch was not profiled."""
    for rnd in ("r05", "r04", "r03", "r02"):       # newest round first
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_forward_bs{B}_hbm_traffic.json")))
            w = t["workload"]
            if (w["num_patch"], w["patch_size"], w["batch"]) == (N, P, B):
                return round(t["kernels"]["EVAL"]["hbm_bytes_per_launch"])
        except Exception:
            continue
    return None


def algorithmic_bytes_per_sample(N, P):
    """the window is read once and one float is written; the 6.1 KB of weights amortise over the batch."""
    return 4 * N * P + 4


def time_eval_forward(model, X, iters=20, reps=5, settle_ms=20.0):
    """Median over `reps` event-timed groups of `iters` launches of the fused eval forward (one kernel per call), taken in steady
    state: the kernel is launched back to back for `settle_ms` first.  The clock of an MI355X that was idle (or in another kernel
    mix) takes 5-10 ms of this kernel to settle: the first 1-2 ms of launches run ~10 % slower (tools/time_forward_steady.py:
    55 us -> 50.3 us at batch 65536 after 100 launches, flat from there to 1000)."""
    import statistics
    model.eval()
    with torch.no_grad():
        one = event_time_ms(lambda: model(X), 3)
        event_time_ms(lambda: model(X), max(3, int(settle_ms / max(one, 1e-3))), warm=0)
        ts = [event_time_ms(lambda: model(X), iters) for _ in range(reps)]
    model.train()
    return statistics.median(ts)


def roofline_measurements(model, X, y, step_ms, iters=10, isolated=False, big_forward=True):
    """HIP-event timing (on torch's current stream = the stream the kernels are launched on) of every
    phase kernel of the training step and of the fused eval forward kernel."""
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    B, N, P, L = X.size(0), model.num_patch, model.patch_size, model.num_layers
    alg = algorithmic_bytes_per_sample(N, P)
    x2d = X.reshape(B, -1).contiguous()
    yv = y.reshape(-1).contiguous()
    shp = model._shape(B)
    model.fused_mse_step(X, y)                       # leaves a valid cache / cells in the workspace
    a = model._train_args(shp, x2d, yv, None, model._step)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    names = phase_names(L)
    resolved = lib.rulgnn_stgcn_train_step_resolve(C.byref(shp), C.c_void_p(x2d.data_ptr()), int(model.step_path))
    chain_kind = "mx" if resolved == _lib.STEP_MX else "fp32"
    # in-step timing: the phases run in the order of the real step with an event between each, so every kernel sees the
    # cache state its predecessor leaves (re-running ONE phase back to back keeps its ~250 MB working set warm in the
    # 256-MB MALL and reads 8-15 % faster than the same kernel does inside the step).  Phase -1 = the step's prepare kernel: the
    # reduction cells are cleared, so the phases run on valid BatchNorm statistics.
    def chain(evs=None):
        _lib.check(lib.rulgnn_stgcn_train_phase_f32(C.byref(shp), C.byref(a), -1, st()), "prepare")
        if evs is not None:
            evs[0].record()
        for ph in range(len(names)):
            _lib.check(lib.rulgnn_stgcn_train_phase_f32(C.byref(shp), C.byref(a), ph, st()), "phase")
            if evs is not None:
                evs[ph + 1].record()
    for _ in range(2):
        chain()
    torch.cuda.synchronize()
    acc = [0.0] * len(names)
    for _ in range(iters):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        chain(evs)
        torch.cuda.synchronize()
        for ph in range(len(names)):
            acc[ph] += evs[ph].elapsed_time(evs[ph + 1])
    per = {name: {"ms": acc[ph] / iters, "bytes_per_sample": phase_bytes_per_sample(name, N, P, L, chain_kind)} for ph, name in enumerate(names)}
    iso = None
    if isolated:
        iso = {}
        for ph, name in enumerate(names):
            def run(ph=ph):
                _lib.check(lib.rulgnn_stgcn_train_phase_f32(C.byref(shp), C.byref(a), ph, st()), "phase")
            iso[name] = round(event_time_ms(run, iters) * 1e3, 1)
    # The kernels' own durations inside REAL steps (device timestamps of the HIP activity tracer: what rocprofv3 --kernel-trace reports,
    # profiles/r0N_train_step_kernel_stats.csv): the event intervals above carry ~5 us of launch / event overhead per phase, which is
    # 10-20 % of a 25-50 us kernel.  The dominant kernel and its roofline are taken from these where the tracer delivers them.
    traced = {}
    try:
        import re
        kt = kernel_times(lambda i: model.fused_mse_step(X, y), steps=10)
        for kname_, (cnt, us) in kt.items():
            short = kernel_short_name(kname_)
            mm = re.search(r"stgcn_train_mx_kernel<(\d+), (\d), (\d), (\d+)>", short) or re.search(r"stgcn_train_phase_kernel<(\d+), (\d), (\d), (\d)", short)
            if mm and "mx_kernel" in short:
                ph = {"0": "F", "1": "TOP", "2": "G"}[mm.group(2)] + (mm.group(3) if mm.group(2) != "1" else "")
            elif mm:
                ph = {"0": "F", "1": "TOP", "2": "G"}[mm.group(3)] + (mm.group(4) if mm.group(3) != "1" else "")
            elif "stgcn_train_f0_mx_kernel" in short:
                ph = "F0"
            else:
                continue
            if abs(cnt - 1.0) < 1e-9:
                traced[ph] = us
    except Exception:
        traced = {}
    if set(traced) == set(names):
        for ph in names:
            per[ph]["event_ms"] = per[ph]["ms"]
            per[ph]["ms"] = traced[ph] * 1e-3
    dom = max(per, key=lambda k: per[k]["ms"])
    d = per[dom]
    ach = alg * B / (d["ms"] * 1e-3) / 1e9                               # algorithmic bytes of the launch / its duration
    ach_traffic = d["bytes_per_sample"] * B / (d["ms"] * 1e-3) / 1e9      # the bytes this phase really moves
    prof = _traffic_profile(chain_kind)
    total_traffic = None
    if prof and (prof["workload"]["num_patch"], prof["workload"]["patch_size"]) == (N, P):
        total_traffic = sum(k["hbm_bytes_per_sample"] for n_, k in prof["kernels"].items() if n_ in names)
    if chain_kind == "mx":
        kname = "stgcn_train_f0_mx_kernel (F0)" if dom == "F0" else f"stgcn_train_mx_kernel<{dom}>"
    else:
        kname = f"stgcn_train_phase_kernel<{dom}>"
    algorithmic_step = sum(v["bytes_per_sample"] for v in per.values())
    roof = {"bound": "hbm", "kernel": kname, "chain": "matrix-core chain, activations recomputed (RULGNN_STEP_MX)" if chain_kind == "mx"
            else "row-mapped fp32 chain (RULGNN_STEP_CHAIN)", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom, N, P, B, chain_kind),
            "algorithmic_bytes_per_sample": alg,
            "frac_traffic": round(ach_traffic / HBM_PEAK_GBS, 4), "phase_bytes_per_sample": d["bytes_per_sample"],
            "step_algorithmic_frac": round(alg * B / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "step_accounted_bytes_per_sample": algorithmic_step,
            "accounted_over_algorithmic": round(algorithmic_step / alg, 2),
            "traffic_over_algorithmic": round(total_traffic / alg, 2) if total_traffic else None,
            "step_traffic_bytes_per_sample": round(total_traffic, 1) if total_traffic else None,
            "traffic_source": prof["file"] if prof else None,
            "us_per_launch": round(d["ms"] * 1e3, 1),
            "phase_us": {k: round(v["ms"] * 1e3, 1) for k, v in per.items()},
            "timing": ("kernel durations inside real steps from the HIP activity tracer (device timestamps, 10 steps); phase_us_events = HIP events "
                       "between consecutive phases launched one by one, ~5 us of launch / event overhead each") if traced and set(traced) == set(names)
                      else "HIP events between consecutive phases of the step (in-step cache state)"}
    if traced and set(traced) == set(names):
        roof["phase_us_events"] = {k: round(v["event_ms"] * 1e3, 1) for k, v in per.items()}
        roof["phase_kernel_time_sum_us"] = round(sum(v["ms"] for v in per.values()) * 1e3, 1)
    if iso:
        roof["phase_us_isolated"] = iso
    # the north-star kernel: fused eval forward, one launch per call
    fms = time_eval_forward(model, X)
    fach = alg * B / (fms * 1e-3) / 1e9
    roof_f = {"bound": "hbm", "kernel": "stgcn_forward_mx_kernel", "achieved": round(fach, 1), "peak": HBM_PEAK_GBS,
              "unit": "GB/s", "frac": round(fach / HBM_PEAK_GBS, 4), "traffic": forward_traffic(N, P, B),
              "algorithmic_bytes_per_sample": alg, "batch": B,
              "us_per_launch": round(fms * 1e3, 1), "samples_per_s": round(B / (fms * 1e-3), 1),
              "compute": compute_leg(forward_flops_per_sample(N, P, L), B / (fms * 1e-3), "one eval forward per sample")}
    roof["compute"] = compute_leg(3 * forward_flops_per_sample(N, P, L), B / (step_ms * 1e-3),
                                  "3 x the forward FLOPs per sample (SURVEY section 8d), whole step; recomputed products not counted")
    if big_forward:
        BB = 1 << 20
        g = torch.Generator(device=X.device).manual_seed(99)
        Xb = torch.rand(BB, N, P, device=X.device, generator=g)
        bms = time_eval_forward(model, Xb, iters=5)
        bach = alg * BB / (bms * 1e-3) / 1e9
        roof_f["at_1M"] = {"batch": BB, "us_per_launch": round(bms * 1e3, 1), "achieved": round(bach, 1),
                           "frac": round(bach / HBM_PEAK_GBS, 4), "samples_per_s": round(BB / (bms * 1e-3), 1),
                           "traffic": forward_traffic(N, P, BB),
                           "compute": compute_leg(forward_flops_per_sample(N, P, L), BB / (bms * 1e-3), "one eval forward per sample")}
        del Xb
        # the reference's C-MAPSS window is 50 points (Data_Process/Data_read_CMAPSS.py:330): the same kernel at 14 x 50
        from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model as _M
        torch.manual_seed(2)
        m50 = _M(num_patch=N, patch_size=50).to(X.device)
        X50 = torch.rand(1 << 19, N, 50, device=X.device, generator=g)
        cms = time_eval_forward(m50, X50, iters=5)
        calg = algorithmic_bytes_per_sample(N, 50)
        cach = calg * X50.size(0) / (cms * 1e-3) / 1e9
        roof_f["cmapss_14x50"] = {"kernel": "stgcn_forward_mx_kernel<2, 14, 50>", "batch": X50.size(0), "algorithmic_bytes_per_sample": calg,
                                  "us_per_launch": round(cms * 1e3, 1), "achieved": round(cach, 1), "frac": round(cach / HBM_PEAK_GBS, 4),
                                  "samples_per_s": round(X50.size(0) / (cms * 1e-3), 1),
                                  "compute": compute_leg(forward_flops_per_sample(N, 50, L), X50.size(0) / (cms * 1e-3), "one eval forward per sample")}
        del X50, m50
        # the reference's own ST_GCN wiring on PHM2012 (configs/hparams.py:238: 40 patches of 64 points): the wide matrix-core kernel
        # (stgcn_forward_mxw_kernel) followed by the scanning launch of the exact kernel, both inside the timed region
        from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model
        WN, WP, WB = 40, 64, 1 << 17
        torch.manual_seed(1)
        wide = ST_GCN_model(num_patch=WN, patch_size=WP).to(X.device)
        Xw = torch.rand(WB, WN, WP, device=X.device, generator=g)
        wms = time_eval_forward(wide, Xw, iters=5)
        walg = algorithmic_bytes_per_sample(WN, WP)
        wach = walg * WB / (wms * 1e-3) / 1e9
        roof_f["phm2012_40x64"] = {"kernel": "stgcn_forward_mxw_kernel + stgcn_forward_fixup_kernel", "batch": WB,
                                   "algorithmic_bytes_per_sample": walg, "us_per_call": round(wms * 1e3, 1), "achieved": round(wach, 1),
                                   "frac": round(wach / HBM_PEAK_GBS, 4), "samples_per_s": round(WB / (wms * 1e-3), 1),
                                   "compute": compute_leg(forward_flops_per_sample(WN, WP, 2), WB / (wms * 1e-3), "one eval forward per sample")}
        del Xw, wide
    return roof, roof_f


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


def rmse_teacher_task(dev, epochs=20, n_train=49152, n_test=8192, batch=4096, max_rul=125.0, checkpoints=(36, 120, 240)):
    """The RMSE half of BASELINE.json's metric on SURVEY section 8(d)'s synthetic task: a fixed random "teacher" ST_GCN (14 x 30, eval
    mode) labels ~49 k uniform windows (about FD004's training-set size); a student with another initialisation is trained for `epochs`
    passes in batches of `batch`, dropout off, (a) on the HIP path (ST_GCN.update) and (b) by the torch-CPU restatement of the reference's
    update (oracle/stgcn_torch_cpu.py) from the SAME initial weights on the same batches; both are scored on held-out windows with
    the reference's formula RMSE = sqrt(mean((pred - y)^2)) * max_rul (utils.py:148-151) after 36, 120 and 240 optimizer steps (drift
    shows as a growing difference).  The north star asks |RMSE_hip - RMSE_cpu| <= 1e-3."""
    from gnn_rul_benchmarking_amd.algorithms import ST_GCN
    from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model
    from oracle import stgcn_torch_cpu as T
    N, P = NUM_PATCH, 30
    g = torch.Generator(device="cpu").manual_seed(4242)
    Xtr, Xte = torch.rand(n_train, N, P, generator=g), torch.rand(n_test, N, P, generator=g)
    torch.manual_seed(100)
    teacher = ST_GCN_model(num_patch=N, patch_size=P, dropout=0.0).to(dev).eval()
    with torch.no_grad():
        ytr, yte = teacher(Xtr.to(dev)).cpu(), teacher(Xte.to(dev)).cpu()
    torch.manual_seed(7)
    algo = ST_GCN(dict(num_patch=N, patch_size=P, dropout=0.0), {"learning_rate": 1e-3, "weight_decay": 1e-4}, dev)
    algo.to(dev)
    init = {k: v.detach().cpu().numpy().copy() for k, v in algo.state_dict().items()}
    st = T.State(init, num_layers=2, lr=1e-3, weight_decay=1e-4)
    Xd, yd, Xted = Xtr.to(dev), ytr.to(dev), Xte.to(dev)
    yt = yte.reshape(-1).double()
    rmse = lambda p_: float(torch.sqrt(torch.mean((p_.double() - yt) ** 2)) * max_rul)
    threads = torch.get_num_threads()
    t_hip = t_cpu = 0.0
    hip_loss = cpu_loss = 0.0
    step, marks = 0, []
    for _ in range(epochs):
        for lo in range(0, n_train, batch):
            t0 = time.perf_counter()
            algo.train()
            hip_loss = algo.update(Xd[lo:lo + batch], yd[lo:lo + batch], 1)["loss"]
            t_hip += time.perf_counter() - t0
            t0 = time.perf_counter()
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            cpu_loss = T.update(st, Xtr[lo:lo + batch], ytr[lo:lo + batch], N, P, 0.0)
            t_cpu += time.perf_counter() - t0
            step += 1
            if step in checkpoints:
                algo.eval()
                with torch.no_grad():
                    ph = algo.model(Xted).cpu().reshape(-1)
                    pc = T.forward(st, Xte, N, P, False).reshape(-1)
                marks.append({"steps": step, "rmse_hip": round(rmse(ph), 6), "rmse_torch_cpu": round(rmse(pc), 6),
                              "abs_diff": round(abs(rmse(ph) - rmse(pc)), 7), "train_loss_hip": round(float(hip_loss), 8),
                              "train_loss_torch_cpu": round(float(cpu_loss), 8), "max_pred_diff": round(float((ph.double() - pc.double()).abs().max()), 8)})
    torch.set_num_threads(threads)
    base = float(torch.sqrt(torch.mean((yt.mean() - yt) ** 2)) * max_rul)
    last = marks[-1]
    return {"task": f"teacher ST_GCN({N}, {P}) labels {n_train} uniform windows; student trained {epochs} epochs, batch {batch}, dropout off, "
                    f"Adam lr 1e-3 wd 1e-4; scored on {n_test} held-out windows, RMSE x max_rul {max_rul:g} (reference utils.py:148-151)",
            "rmse_hip": last["rmse_hip"], "rmse_torch_cpu": last["rmse_torch_cpu"], "abs_diff": last["abs_diff"],
            "within_1e-3": bool(all(m["abs_diff"] <= 1e-3 for m in marks)), "after_steps": marks, "rmse_of_predicting_the_mean": round(base, 4),
            "final_train_loss_hip": last["train_loss_hip"], "final_train_loss_torch_cpu": last["train_loss_torch_cpu"],
            "steps": step, "seconds_hip": round(t_hip, 2), "seconds_torch_cpu": round(t_cpu, 2), "max_pred_diff": last["max_pred_diff"]}


def rmse_teacher_task_stmsgcn(dev, n_train=4000, n_test=1000, batch=100, epochs=6, max_rul=1.0, checkpoints=(36, 120, 240)):
    """The same experiment on a family WITHOUT BatchNorm and dropout (STMSGCN at the reference's PHM2012 Condition_1 wiring, 160 patches of
    16 points, the protocol's batch 100, configs/hparams.py): teacher-labelled windows, the student trained on the HIP path and by the
    torch-CPU restatement (oracle/families_torch_cpu.py) from the same weights on the same batches; 240 optimizer steps."""
    from gnn_rul_benchmarking_amd.algorithms import STMSGCN
    from gnn_rul_benchmarking_amd.stmsgcn import STMSGCN_model
    from gnn_rul_benchmarking_amd import hparams as HP
    from oracle import families_torch_cpu as T
    hp = HP.get_hparams_class("PHM2012")("Condition_1")
    cfg = dict(hp.alg_hparams["STMSGCN"])
    L = cfg["num_patch"] * cfg["patch_size"]
    g = torch.Generator(device="cpu").manual_seed(777)
    Xtr, Xte = torch.rand(n_train, 1, L, generator=g), torch.rand(n_test, 1, L, generator=g)
    torch.manual_seed(101)
    teacher = STMSGCN_model(**cfg).to(dev).eval()
    with torch.no_grad():
        ytr, yte = teacher(Xtr.to(dev)).cpu(), teacher(Xte.to(dev)).cpu()
    torch.manual_seed(8)
    tc = {"learning_rate": 1e-3, "weight_decay": 0.0}
    algo = STMSGCN(cfg, tc, dev)
    algo.to(dev)
    algo.train()
    init = {k: v.detach().cpu().numpy().copy() for k, v in algo.state_dict().items()}
    st = T.StmsgcnState(init, cfg, lr=tc["learning_rate"], weight_decay=tc["weight_decay"])
    Xd, yd, Xted = Xtr.to(dev), ytr.to(dev), Xte.to(dev)
    yt = yte.reshape(-1).double()
    rmse = lambda p_: float(torch.sqrt(torch.mean((p_.double() - yt) ** 2)) * max_rul)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    step, marks, t_cpu = 0, [], 0.0
    for _ in range(epochs):
        for lo in range(0, n_train, batch):
            hip_loss = algo.update(Xd[lo:lo + batch], yd[lo:lo + batch], 1)["loss"]
            t0 = time.perf_counter()
            cpu_loss = T.stmsgcn_update(st, Xtr[lo:lo + batch], ytr[lo:lo + batch])
            t_cpu += time.perf_counter() - t0
            step += 1
            if step in checkpoints:
                with torch.no_grad():
                    ph = algo.model(Xted).cpu().reshape(-1)
                    pc = T.stmsgcn_forward(st, Xte).reshape(-1)
                scale = float(yt.abs().max())
                marks.append({"steps": step, "rmse_hip": round(rmse(ph), 8), "rmse_torch_cpu": round(rmse(pc), 8),
                              "rel_diff": round(abs(rmse(ph) - rmse(pc)) / max(rmse(pc), 1e-30), 8), "train_loss_hip": float(hip_loss),
                              "train_loss_torch_cpu": float(cpu_loss), "max_pred_diff_over_label_scale": round(float((ph.double() - pc.double()).abs().max()) / scale, 8)})
    torch.set_num_threads(threads)
    return {"task": f"teacher STMSGCN (PHM2012 Condition_1 wiring {cfg['num_patch']} x {cfg['patch_size']}) labels {n_train} uniform windows; student "
                    f"trained {epochs} epochs at batch {batch} (Adam lr {tc['learning_rate']}, no weight decay; no BatchNorm, no dropout in this model); "
                    f"RMSE on {n_test} held-out windows in label units", "after_steps": marks, "steps": step, "seconds_torch_cpu": round(t_cpu, 2),
            "within_1e-3_relative": bool(all(m["rel_diff"] <= 1e-3 for m in marks))}


def stgcn_train_other_shape(dev, N, P, batches, steps=10, fp32_batches=()):
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
        for name, path in (("auto", _lib.STEP_AUTO), ("fp32_chain", _lib.STEP_CHAIN)):
            if name == "fp32_chain" and B not in fp32_batches:
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
                                   "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                                "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 4), "eval_achieved": round(etf, 2),
                                                "eval_frac": round(etf / FP32_MFMA_PEAK_TFLOPS, 4),
                                                "counts": "whole step: 3 x the forward matmul FLOPs per sample x samples/s (not one kernel)"}}
            del algo, X, y
        out[name] = entry
    out["profile"] = "profiles/r05_stgcn_tiled_xjtu_bs1024_kernel_stats.csv (rocprofv3 --kernel-trace --stats of tools/time_tiled_one.py: 5 large GEMMs at " \
                     "~147 TFLOP/s = 36 % of the step, the position-parallel kernels between them the rest)"
    return out


def cpu_baseline(num_patch, patch_size, dropout):
    """The reference's CPU path restated on torch-CPU (oracle/stgcn_torch_cpu.py: the same ATen kernels the reference runs,
    pinned to the reference's own training curve in tests/test_torch_cpu_baseline.py), SURVEY section 8(d) protocol:
    torch.set_num_threads(n) for n = 1 and n = all host cores (plus 16 and 64 where the host has more), 20 warm-up + 100 timed
    iterations of ST_GCN.update each, same input distribution as the GPU run; bounded sample: batch 4096 (the reference's CPU
    throughput saturates there) and the reference protocol's own batch 32 (BASELINE.json configs[0])."""
    from oracle import stgcn_torch_cpu as T
    cores = os.cpu_count() or 1
    thread_counts = sorted({1, min(16, cores), min(64, cores), cores})
    runs = []
    for n in thread_counts:
        runs.append(dict(T.time_update(num_patch, patch_size, 4096, n, dropout, warmup=20, iters=100, budget_s=14.0), what="train"))
    best = max(runs, key=lambda r: r["samples_per_s"])
    runs.append(dict(T.time_update(num_patch, patch_size, 32, 1, dropout, warmup=20, iters=100, budget_s=3.0), what="train, BASELINE.json configs[0] batch"))
    runs.append(dict(T.time_update(num_patch, patch_size, 4096, best["threads"], dropout, warmup=20, iters=100, budget_s=6.0,
                                   eval_forward=True), what="eval forward"))
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port",
            "sample": f"{best['iterations']} ST_GCN.update iterations (after 20 warm-up) of batch 4096 ({num_patch}x{patch_size}, dropout {dropout}), "
                      f"torch-CPU restatement of the reference (oracle/stgcn_torch_cpu.py), fp32, best of thread counts {thread_counts}",
            "cpu_model": T.cpu_model_name(), "host_cpus": cores, "torch": torch.__version__, "runs": runs,
            "reference_on_survey_container": "8 vCPU Xeon 2.1 GHz: train 5.8 k samples/s at batch 32, 44.6 k/s best (BASELINE.md)"}


# SURVEY section 8d measurement configurations of the other hot-path families: (dataset, id, per-GPU batch, input shape,
# forward matmul/conv FLOPs per sample as counted there)
FAMILY_CONFIGS = {
    "ASTGCNN": ("NCMAPSS", None, 512, (20, 50), 1.22e6),
    "FC_STGNN": ("CMAPSS", "FD004", 256, (14, 50), 3.28e6),
    "HAGCN": ("CMAPSS", "FD004", 256, (14, 50), 0.99e6 + 8.7e6),
    # SURVEY 8d counts 185 MFLOP per sample as the reference WRITES the model (dense diag_embed products for the normalisation); the kernels
    # scale rows / columns instead and execute ~121 MFLOP: the whole-step estimate is priced on the work actually done
    "STMSGCN": ("XJTU_SY", "Condition_1", 128, (1, 32768), 121e6),
    # SURVEY 8f rank 3; forward FLOPs per sample: ChebNet projection 14*150*64*2, graph terms 2*14*14*50*2 + cdist 14*14*50*3,
    # GRU input projection 14*64*192*2 (one step, h0 = 0), fc 896*2
    "STGNN": ("CMAPSS", "FD004", 256, (14, 50), 0.68e6),
    # SURVEY 8f rank 3; forward FLOPs per sample: adjacency 2*(2*14*14*50) + 2*(2*14*14*14), 50 graphs x (2*14*14*(1+32) + 2*14*32*32),
    # LSTM 50 steps x 2*4*32*(14+32), fusion 2*32*14*50 + 2*32*32*3*50 + 2*2*1600
    "RGCNU": ("CMAPSS", "FD004", 256, (14, 50), 0.09e6 + 2.08e6 + 0.59e6 + 0.36e6),
    # SURVEY 8f rank 3 (PHM2012 Condition_1 wiring at the reference protocol's batch); forward FLOPs per sample: the three ChebNet GEMMs
    # over 20 x 9 node rows 180 * 2 * (27*300 + 900*200 + 600*100), auto-encoder 20 * 2 * (2*900*50 + 6*50*50), LSTM, head
    "STNet": ("PHM2012", "Condition_1", 100, (1, 2560), 89.3e6 + 4.2e6 + 0.05e6),
    # SURVEY 8f rank 3 (PHM2012 Condition_2 wiring: 128 patches of 20 points, hidden 1000 / 200, the reference protocol's batch); forward
    # FLOPs per sample: gcn1 2*128*40*1000, two projection layers 2 * (2*128*128*1000 + 2*128*1000*1000), attention 2 * 2*200*128*1000
    "SAGCN": ("PHM2012", "Condition_2", 100, (1, 2560), 10.2e6 + 2 * (32.8e6 + 256e6) + 102.4e6),
    # SURVEY 8f rank 3 (C-MAPSS FD001 wiring: hidden 64, 3 heads); forward FLOPs per sample: covariance 2*14*14*50, GCNs 2*14*14*(50+64) +
    # 2*14*64*(50+64), six attention heads 6 * (2*14*64*64 + 2*14*14*64), tcn1 2*64*64*(2*14 + 14 + 2*64), encoder 2*3*64*64, tcn2
    # 2*10*64*(2*64 + 64 + 2*10), head
    "STAGNN": ("CMAPSS", "FD001", 256, (14, 50), 0.02e6 + 0.25e6 + 0.84e6 + 1.39e6 + 0.03e6 + 0.27e6),
}
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X fp32 matrix peak
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 matrix peak (MI355X_MICROARCH.md)


def kernel_short_name(name):
    """'void rulgnn::(anonymous namespace)::fc_graph_bwd_kernel<2>(rulgnn::...)' -> 'fc_graph_bwd_kernel<2>'"""
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("rulgnn::", "").strip()


def family_traffic(family, kernel_short):
    """HBM bytes per launch of a family's kernel from the committed PMC summary (profiles/r0N_family_hbm_traffic.json, the newest round
    that has the kernel; written by tools/family_traffic_report.py from separate FETCH_SIZE / WRITE_SIZE passes), or None."""
    for tag in ("r05", "r04", "r03", "r02"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_family_hbm_traffic.json")))
            for k, v in t["families"][family]["kernels"].items():
                if k == kernel_short:
                    return round(v["hbm_bytes_per_launch"])
        except Exception:
            pass
    return None


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


def _stmsgcn_nodes(cfg):
    """Graph nodes of STMSGCN = energy bands of a patch's lagged spectrum (models/STMSGCN/Model.py:7-31)."""
    return (cfg["patch_size"] - cfg["interval"]) // cfg["band_width"]


def _gcn_stack_flops(n, dims):
    """Forward matmul FLOPs of STMSGCN's GCN stack for ONE graph of n nodes (models/STMSGCN/Model.py:84-112 without the dense
    diag products): per layer the Gram matrix x x^T and A.x (2 n^2 f each) and the Linear (2 n f_in f_out)."""
    f = [1] + list(dims)
    return sum(2 * 2 * n * n * f[l] + 2 * n * f[l] * f[l + 1] for l in range(len(dims)))


# Work model of the kernel that dominates each family's step: substring of the kernel name -> f(cfg, batch, launches per step)
# = (algorithmic FLOPs of ONE launch, how they are counted).  Matmul-type FLOPs only, as SURVEY 8(d) counts them.
def dominant_kernel_work(family, name, cfg, B, shape, per_step):
    if family == "FC_STGNN" and "fc_graph_bwd" in name:
        Q, D2 = 2 * cfg["num_node"], 2 * cfg["hidden_dim"]
        graphs = B * ((cfg["num_patch"] - 1) + (cfg["num_patch"] - 2) // 2 + 1)            # window 2, stride 1 and 2 (Model_Base.py:175-225)
        return graphs / per_step * 6 * Q * Q * D2, ("backward of one window graph: dA = dAX X'^T, dX' = A^T dAX, dM = (dS + dS^T) M, "
                                                    "2 Q^2 D FLOPs each (Q = 28 nodes, D = 16); averaged over the two window blocks")
    if family == "FC_STGNN" and ("fc_graph_kernel" in name or "fc_graph_mx" in name or "fc_block_mx" in name):
        Q, D2 = 2 * cfg["num_node"], 2 * cfg["hidden_dim"]
        graphs = B * ((cfg["num_patch"] - 1) + (cfg["num_patch"] - 2) // 2 + 1)
        extra = (2 * Q * D2 * D2 + 2 * Q * D2 * (D2 // 2)) if "fc_block_mx" in name else 0            # mapping + the block's Linear
        return graphs / per_step * (4 * Q * Q * D2 + extra), ("forward of one window graph: S = M M^T and A X', 2 Q^2 D FLOPs each (Q = 28 nodes, D = 16)"
                                                              + ("; plus the mapping F W_map^T and the block's Linear" if extra else "")
                                                              + "; averaged over the two window blocks")
    if family == "HAGCN" and ("lstm_forward_kernel" in name or "lstm_backward_kernel" in name):
        T = B * shape[0]
        H = cfg["encoder_hidden_dim"] * (2 if "<128" in name else 1)                          # layers 1, 3: H; layer 2: 2H (Model.py:41-56)
        return T * 2 * cfg["num_patch"] * (2 * 4 * H * H), ("recurrent matvec 4H x H per step, direction and sequence over batch x nodes = "
                                                            f"{T} SEQUENTIAL steps (H = {H}): a latency-bound recurrence, not a throughput kernel")
    if family == "ASTGCNN" and ("ast_front_kernel" in name or "ast_graph_bwd_kernel" in name):
        # round 4: the GEMM launches around the graph stage live inside these two kernels (csrc/astgcnn.hip): gate projection x theta^T and
        # P projection (2 N E^2 each), the filter product (2 K E O per sample) -- and their backward counterparts d G += d PX P, DT = D Fcat^T
        N, E, K, O = cfg["num_nodes"], cfg["encoder_out_dim"], cfg["K"], cfg["output_dim"]
        graph = 3 * N * N * E + 2 * N * N * N + (K - 1) * 2 * N * N * E                         # cdist, one N^3 Laplacian term, Chebyshev recursion
        if "bwd" in name:
            return B * (2 * graph + 2 * N * E * E + 2 * K * E * O), (f"per sample ({N} nodes, {E} features, K = {K}): graph backward = 2 x (pairwise distances 3 N^2 E + "
                                                                     "(K - 1) Chebyshev products 2 N^2 E), d G += d PX P 2 N E^2, DT = D Fcat^T 2 K E O")
        return B * (graph + 4 * N * E * E + 2 * K * E * O), (f"per sample ({N} nodes, {E} features, K = {K}): gate and P projections 2 x 2 N E^2, pairwise "
                                                             "distances 3 N^2 E, (K - 1) Chebyshev products 2 N^2 E, filter product 2 K E O")
    if family == "ASTGCNN" and "tcn_conv" in name:
        N, T = cfg["num_nodes"], cfg["time_length"]
        taps = 6                                                                                # kernel_size of the reference TCN (models/ASTGCNN/Model.py:236)
        return B * 2 * N * N * taps * T * (2 if "bwd" in name else 1), (f"causal convolution {N} -> {N} channels, {taps} taps, {T} steps per sample"
                                                                        + ("; backward: data and weight gradient" if "bwd" in name else ""))
    if family == "SAGCN" and "sgemm_" in name and "reduce" not in name:
        # every matrix product of one step (csrc/sagcn.hip::sagcn_run) as (what, M, N, K, A contiguous along k, B contiguous along k, split-K),
        # mapped to the kernel instance that serves it by the dispatch rules of csrc/sgemm_mfma.hpp (restated here)
        P, H, Ah = cfg["num_patch"], cfg["gcn_hidden_dim"], cfg["attention_hidden_dim"]
        R, BH = P * B, B * H
        gemms = [("gcn1", R, H, 40, 1, 1, 0)]
        for _ in range(2):
            gemms += [("node axis", P, BH, P, 1, 0, 0), ("feature axis", R, H, H, 1, 1, 0), ("d node-mixed", R, H, H, 1, 0, 0),
                      ("feature-axis weight gradient", H, H, R, 0, 0, 1), ("node-axis weight gradient", P, P, BH, 1, 1, 1), ("d input", P, BH, P, 0, 0, 0)]
        gemms += [("attention tanh layer", Ah, BH, P, 1, 0, 0), ("attention logits", P, BH, Ah, 1, 0, 0), ("softmax-layer weight gradient", P, Ah, BH, 1, 1, 1),
                  ("d tanh", Ah, BH, P, 0, 0, 0), ("tanh-layer weight gradient", Ah, P, BH, 1, 1, 1), ("d h3", P, BH, Ah, 0, 0, 0)]

        def instance(M, N, K, ak, bk, split):
            slices = 1
            if split:
                big = M > 96 and N > 96
                t = 128 if big else 64
                tiles = -(-M // t) * -(-N // t)
                slices = max(1, min((768 if big else 1024) // tiles, -(-K // 256), 256))
            fl = f"<{'true' if ak else 'false'}, {'true' if bk else 'false'}"
            if M > 192 and N > 192 and -(-M // 256) * -(-N // 256) * slices >= 160:
                return "sgemm_bf16x3v_kernel" + fl
            if M > 96 and N > 96 and K >= 16 and -(-M // 128) * -(-N // 128) * slices >= 96:
                return "sgemm_bf16x3_kernel" + fl
            return "other"
        mine = [(w, 2.0 * M * N * K) for (w, M, N, K, ak, bk, sp) in gemms if name.replace("rulgnn::", "").replace("void ", "").startswith(instance(M, N, K, ak, bk, sp))]
        if mine:
            flops = sum(f for _, f in mine)
            return flops / per_step, (f"matrix products of one step served by this kernel instance ({', '.join(sorted(set(w for w, _ in mine)))}): "
                                      f"{flops / 1e9:.1f} GFLOP over {len(mine)} products, {per_step:.0f} launches counted (P = {P}, H = {H}, Ah = {Ah}, batch {B})")
    if family == "STMSGCN" and "msg_gcn_backward" in name:
        n = _stmsgcn_nodes(cfg)
        return B * cfg["num_patch"] * 2 * _gcn_stack_flops(n, cfg["gcn_dims"]), (f"backward of the 4-layer GCN stack of every (sample, patch) graph ({n} nodes): "
                                                                                  "2 x its forward FLOPs (Gram matrix, A.x and Linear per layer)")
    if family == "STMSGCN" and "msg_features" in name:
        n = _stmsgcn_nodes(cfg)
        return B * cfg["num_patch"] * _gcn_stack_flops(n, cfg["gcn_dims"]), "forward of the GCN stack per graph (the DFT is not counted)"
    return None, None


TORCH_CPU_FAMILIES = ("FC_STGNN", "ASTGCNN", "HAGCN", "STMSGCN")


def family_torch_cpu_baseline(family, cfg, batch, budget_s=10.0):
    """the path the reference itself takes on a CPU -- ATen kernels, autograd, torch.optim.Adam -- restated in
    oracle/families_torch_cpu.py (pinned to the reference's own fixtures, tests/test_torch_cpu_families.py) and timed on this box's host
    cores: a FULL update() at the configuration's batch, 1 thread and 16 / 64 threads, up to 100 iterations or the time budget."""
    from oracle import families_torch_cpu as T
    cores = os.cpu_count() or 1
    counts = sorted({1, min(16, cores), min(64, cores)})
    per = budget_s / len(counts)
    runs = [T.time_update(family, dict(cfg), batch, th, warmup=3, iters=100, budget_s=per) for th in counts]
    best = max(runs, key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "torch-cpu restatement",
            "cpu_model": T.cpu_model_name(), "host_cpus": cores, "torch": torch.__version__, "runs": runs,
            "sample": f"full update() (train forward + loss + backward + torch.optim.Adam) of oracle/families_torch_cpu.py at batch {batch}, "
                      f"fp32 ATen kernels, threads {counts}, <= {per:.1f} s or 100 iterations each (iterations timed: "
                      f"{[r['iterations'] for r in runs]})",
            "reference_on_survey_container": BASELINE_MD_FAMILY.get(family)}


# the reference's own update() on the survey container (8 vCPU Xeon 2.1 GHz), samples/s
BASELINE_MD_FAMILY = {"FC_STGNN": "1 648 samples/s at batch 256", "ASTGCNN": "19 336 samples/s at batch 512", "HAGCN": "305 samples/s at batch 256 (FD001 wiring)",
                      "STMSGCN": "67 samples/s at batch 128"}


def family_cpu_baseline(family, cfg, shape, budget_s=10.0, model=None, batch=None):
    """The family's CPU baseline on this box's host cores, bounded sample: the torch-CPU restatement for the BASELINE.json families,
    the numpy oracle (train-step restatement, ``kind: port``) for the section 8(f) families."""
    import numpy as np
    if family in TORCH_CPU_FAMILIES and batch is not None:
        return family_torch_cpu_baseline(family, cfg, batch, budget_s)
    rng = np.random.default_rng(0)
    if family == "HAGCN" and model is None:
        return None
    if family == "ASTGCNN":
        from oracle import astgcnn_oracle as O
        p = O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["output_dim"], cfg["K"])
        bs = 64
        x, y = rng.uniform(-1, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y)
    elif family == "FC_STGNN":
        from oracle import fcstgnn_oracle as O
        c = O.Config(**cfg)
        p = O.random_params(c)
        bs = 32
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, c)
    elif family == "STMSGCN":
        from oracle import stmsgcn_oracle as O
        c = O.Config(cfg["num_patch"], cfg["patch_size"], cfg["interval"], cfg["band_width"], cfg["gcn_dims"], cfg["gru_hidden_dim"])
        p = O.random_params(c)
        bs = 2
        x, y = rng.uniform(0, 1, (bs, shape[1])), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, c)
    elif family == "STGNN":
        from oracle import stgnn_oracle as O
        p = O.random_params(cfg["num_patch"], cfg["patch_size"], cfg["num_nodes"], cfg["hidden_dim"], cfg["K"])
        bs = 256
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.forward_backward(x, y, p, cfg["num_patch"], cfg["patch_size"], cfg["top_k"])
    elif family == "RGCNU":
        from oracle import rgcnu_oracle as O
        p = O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["hidden_dim"], cfg["encoder_hidden_dim"], cfg["kernel_size"])
        bs = 32
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["alpha"])
    elif family == "STNet":
        from oracle import stnet_oracle as O
        p = O.random_params(cfg["num_patch"], cfg["num_nodes"], cfg["input_dim"], cfg["Cheb_layers"], cfg["lstm_hidden_dim"], cfg["autoencoder_hidden_dim"])
        bs = 8
        x, y = rng.normal(0, 1, (bs, shape[1])), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["num_patch"], cfg["patch_size"], cfg["nperseg"])
    elif family == "STAGNN":
        from oracle import stagnn_oracle as O
        p = O.random_params(cfg["num_nodes"], cfg["time_length"], cfg["hidden_dim"], cfg["output_dim"], cfg["num_heads"])
        bs = 64
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["num_heads"], cfg["threshold"])
    elif family == "SAGCN":
        from oracle import sagcn_oracle as O
        p = O.random_params(cfg["num_patch"], cfg["gcn_hidden_dim"], cfg["attention_hidden_dim"])
        bs = 8
        x, y = rng.uniform(-0.5, 0.5, (bs, shape[1])), rng.uniform(0, 1, bs)
        run = lambda: O.loss_and_grads(p, x, y, cfg["num_patch"], cfg["patch_size"])
    elif family == "HAGCN":
        # the oracle restates the model in blocks (Bi-LSTM stack, graph stack, head): one train step = their forwards and backwards
        # in sequence, on the bench model's own parameters (models/HAGCN/Model.py:149-195, algorithms.py:222-248 with alpha = 100)
        from oracle import hagcn_oracle as O
        p = {k[6:] if k.startswith("model.") else k: v.detach().double().cpu().numpy() for k, v in model.state_dict().items()}
        ps, npatch = cfg["patch_size"], cfg["num_patch"]
        bs = 4
        x, y = rng.uniform(0, 1, (bs,) + shape), rng.uniform(0, 1, (bs, 1))

        def run():
            pred, kl, fw = O.forward(p, x, ps, npatch)
            dpred = 2.0 * (pred - y) / bs
            f2 = fw.feats.reshape(bs, -1)
            h = np.maximum(f2 @ p["fc.0.weight"].T + p["fc.0.bias"], 0.0)
            dh = (dpred @ p["fc.2.weight"]) * (h > 0)
            dfeats = (dh @ p["fc.0.weight"]).reshape(fw.feats.shape)
            _, dx0 = O.graph_backward(p, fw, dfeats, 100.0)
            O.td_backward(p, x, ps, npatch, dx0)
    else:
        return None
    # the oracle's cost sits in numpy's BLAS / einsum calls: timed with 1 BLAS thread and with all host cores, the better one quoted
    from threadpoolctl import threadpool_limits
    from oracle import stgcn_torch_cpu as T
    cores = os.cpu_count() or 1
    runs = []
    for th in sorted({1, cores}):
        with threadpool_limits(limits=th):
            run()
            n, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s / 2 and n < 50:
                run()
                n += 1
            el = time.perf_counter() - t0
        runs.append({"threads": th, "samples_per_s": round(bs * n / el, 2), "steps": n})
    best = max(runs, key=lambda r: r["samples_per_s"])
    return {"value": best["samples_per_s"], "unit": "samples/s", "cores": best["threads"], "kind": "port", "cpu_model": T.cpu_model_name(),
            "host_cpus": cores, "runs": runs,
            "sample": f"train steps (forward + loss + backward, no optimizer) of oracle/{family.lower()}_oracle.py, batch {bs}, fp64 numpy; "
                      f"BLAS threads 1 and {cores}, <= {budget_s / 2:.0f} s each"}


def family_main(args, world, rank, dev, use_dist, dist):
    out = family_line(args, args.family, world, rank, dev, use_dist, dist)
    return json.dumps(out) if out is not None else None


def family_line(args, family, world, rank, dev, use_dist, dist, batch=None, cpu_budget_s=10.0):
    """The bench contract for one of the other model families on its SURVEY section 8d configuration: returns the line as a dict
    (rank 0) or None."""
    import copy
    args = copy.copy(args)
    args.family = family
    from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
    from gnn_rul_benchmarking_amd.dp import DataParallel
    from gnn_rul_benchmarking_amd import hparams as HP
    ds, did, B, shape, fwd_flops = FAMILY_CONFIGS[args.family]
    if batch is not None:
        B = batch
    hp = HP.get_hparams_class(ds)(did)
    cfg, train_cfg = hp.alg_hparams[args.family], hp.train_params[args.family]
    torch.manual_seed(0)
    algo = get_algorithm_class(args.family)(cfg, train_cfg, dev)
    algo.to(dev)
    algo.train()
    algo.sync_loss = bool(args.sync_loss)
    if args.dtype != "f32":
        if args.family != "FC_STGNN":
            raise SystemExit("--dtype bf16 exists for --family FC_STGNN only")
        algo.model.compute_dtype = args.dtype
    replicas = args.family == "HAGCN"                  # its LSTM recurs along batch*nodes: not sample-shardable
    if use_dist and not replicas:
        algo.attach_data_parallel(DataParallel())
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    Xs = [torch.rand(B, *shape, device=dev, generator=g) for _ in range(2)]
    ys = [torch.rand(B, 1, device=dev, generator=g) for _ in range(2)]

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    variant_error = None
    if args.dtype != "f32" and rank == 0:
        # the variant's error, measured live on the first batch against the fp32 path with the same weights (train-mode forward)
        preds = {}
        for dt in ("f32", args.dtype):
            algo.model.compute_dtype = dt
            step0 = algo.model._step
            preds[dt] = algo.model.fused_mse_step(Xs[0], ys[0])[0].clone()
            algo.model._step = step0                       # same dropout mask for both
        algo.model.compute_dtype = args.dtype
        err = float((preds[args.dtype] - preds["f32"]).abs().max() / preds["f32"].abs().max())
        variant_error = {"pred_max_rel_error_vs_f32": err, "meets_1e-4_gate": bool(err < 1e-4),
                         "note": "bf16 operands on the row projections only; tests/test_fcstgnn_gpu.py bounds it against the fp64 oracle"}
    last = None
    for i in range(args.warmup):
        last = algo.update(Xs[i % 2], ys[i % 2], 1)["loss"]
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        last = algo.update(Xs[i % 2], ys[i % 2], 1)["loss"]
    sync()
    el = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    if rank != 0:
        return None
    rate = world * B * args.steps / el
    tf = 3.0 * fwd_flops * rate / 1e12
    step_ms = el / args.steps * 1e3
    roof = {"bound": "mfma", "achieved": round(tf, 3), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / FP32_MFMA_PEAK_TFLOPS, 5), "traffic": None,
            "note": "whole step: 3 x SURVEY section 8d forward FLOPs per sample x samples/s (not one kernel)"}
    if not args.no_roofline:
        # the dominant kernel of the step, measured live (per-kernel device time over 10 more steps), priced on its own matmul FLOPs
        kt = kernel_times(lambda i: algo.update(Xs[i % 2], ys[i % 2], 1))
        tot = sum(c * us for c, us in kt.values())
        dom = max(kt, key=lambda k: kt[k][0] * kt[k][1])
        per_step, us = kt[dom]
        work, how = dominant_kernel_work(args.family, dom, cfg, B, shape, per_step)
        generic_above = None
        if not work:
            # the largest share belongs to a generic GEMM instance that serves several shapes of the step: price the largest NAMED
            # kernel that has a FLOP model instead, and say which launches stand above it
            ranked = sorted(kt, key=lambda k: -kt[k][0] * kt[k][1])
            for k in ranked:
                w2, h2 = dominant_kernel_work(args.family, k, cfg, B, shape, kt[k][0])
                if w2:
                    generic_above = [kernel_short_name(x) for x in ranked[:ranked.index(k)]]
                    dom, (per_step, us), work, how = k, kt[k], w2, h2
                    break
        short = kernel_short_name(dom)
        top = sorted(kt.items(), key=lambda kv: -kv[1][0] * kv[1][1])[:5]
        whole = roof
        roof = {"bound": "mfma", "kernel": short, "launches_per_step": round(per_step, 2), "us_per_launch": round(us, 2),
                "share_of_step_kernel_time": round(per_step * us / tot, 4), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "traffic": family_traffic(args.family, short),
                "timing": "per-kernel device time from the HIP activity tracer over 10 steps inside bench.py",
                "step_kernel_time_us": round(tot, 1), "launches_in_step": round(sum(c for c, _ in kt.values()), 1),
                "top_kernels": [{"kernel": kernel_short_name(k),
                                 "launches_per_step": round(c, 2), "us_per_launch": round(u, 2)} for k, (c, u) in top],
                "whole_step_estimate": {"achieved": whole["achieved"], "frac": whole["frac"], "note": whole["note"]}}
        if work:
            ach = work / (us * 1e-6) / 1e12
            peak = FP32_MFMA_PEAK_TFLOPS
            if "bf16x3" in short:
                # fp32-class products out of six bf16 matrix instructions each (csrc/sgemm_mfma.hpp): the ceiling is a sixth of the dense bf16 peak
                peak = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)
                roof["peak_note"] = "dense bf16 matrix peak (2500 TFLOP/s, MI355X_MICROARCH.md) / 6 products per fp32-class product; FLOPs counted once"
            roof.update({"achieved": round(ach, 4), "peak": peak, "frac": round(ach / peak, 5), "flops_per_launch": round(work), "work_model": how})
            if generic_above:
                roof["generic_gemm_instances_with_larger_share"] = generic_above
        else:
            roof.update({"achieved": whole["achieved"], "frac": whole["frac"],
                         "work_model": "no per-kernel FLOP model for this kernel (a generic GEMM serving several shapes): the whole-step estimate is quoted"})
    out = {"metric": f"training samples/sec, {args.family} ({ds} {did or ''} wiring)".replace("  ", " "), "value": round(rate, 1),
           "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(step_ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": f"{args.family}.update (fwd+loss+bwd+Adam), input [{B}, {shape[0]}, {shape[1]}], hparams {cfg}",
                      "per_gpu_batch": B, "global_batch": world * B,
                      "parallelism": f"replicas{world}" if replicas else f"dp{world}"},
           "final_loss": round(float(last), 6),
           "roofline": roof}
    if variant_error is not None:
        out["variant_error"] = variant_error
    if world == 1 and not args.no_cpu_baseline:
        cb = family_cpu_baseline(args.family, cfg, shape, budget_s=cpu_budget_s, model=algo.model, batch=B)
        if cb:
            out["cpu_baseline"] = cb
    return out


def finish(line, use_dist, dist):
    """Tear the process group down, then print rank 0's JSON line as the LAST thing on stdout: RCCL's version banner
    (NCCL_DEBUG=VERSION is exported in this image) sits in the C stdio buffer and would otherwise land after the line."""
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(line, flush=True)


def self_spawn(args):
    """`python bench.py --gpus N` from a bare shell: re-launch under torch.distributed.run, one process per GPU, rendezvous
    on 127.0.0.1 (the container hostname may not resolve); returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def timed_repetitions(step_fn, steps, warmup, reps, use_dist, dist, dev):
    """`warmup` untimed steps, then `reps` repetitions of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both
    sides, the MAX over ranks taken per repetition; returns the list of per-repetition seconds (same on every rank)."""
    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
    k = 0
    for _ in range(warmup):
        step_fn(k)
        k += 1
    out = []
    for _ in range(reps):
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step_fn(k)
            k += 1
        sync()
        el = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        out.append(el)
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_spawn(args))
    args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ST_GCN path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import __graft_entry__ as entry
    if rank == 0:
        entry.build()
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dp
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        dist.barrier()

    if args.family != "ST_GCN":
        if args.batch != 65536:
            line = family_line(args, args.family, world, rank, dev, use_dist, dist, batch=args.batch)
            line = json.dumps(line) if line is not None else None
        else:
            line = family_main(args, world, rank, dev, use_dist, dist)
        finish(line, use_dist, dist)
        return

    import statistics
    from gnn_rul_benchmarking_amd.algorithms import ST_GCN
    from gnn_rul_benchmarking_amd.dp import DataParallel
    from gnn_rul_benchmarking_amd import hparams as HP

    hp = HP.get_hparams_class("CMAPSS")("FD004")
    train_cfg = hp.train_params["ST_GCN"]
    model_cfg = dict(hp.alg_hparams["ST_GCN"], patch_size=args.patch_size, dropout=args.dropout)
    torch.manual_seed(0)
    algo = ST_GCN(model_cfg, train_cfg, dev)
    algo.to(dev)
    algo.train()
    algo.sync_loss = bool(args.sync_loss)
    if use_dist:
        algo.attach_data_parallel(DataParallel(sync_bn=args.sync_bn))

    def run(B):
        """Per-rank batch B: returns (per-repetition seconds, last loss, the batches)."""
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        nbuf = 4                                        # distinct batches, cycled (all resident in HBM)
        Xs = [torch.rand(B, NUM_PATCH, args.patch_size, device=dev, generator=g) for _ in range(nbuf)]
        ys = [torch.rand(B, 1, device=dev, generator=g) for _ in range(nbuf)]
        last = [None]

        def step(k):
            last[0] = algo.update(Xs[k % nbuf], ys[k % nbuf], 1)["loss"]
        els = timed_repetitions(step, args.steps, args.warmup, max(1, args.reps), use_dist, dist, dev)
        loss = float(last[0])
        if not (loss == loss):
            raise SystemExit("training diverged to NaN")
        return els, loss, Xs, ys

    B = args.batch
    weak = args.scaling != "strong" or world == 1
    els, final_loss, Xs, ys = run(B if weak else max(1, B // world))
    per_rank = B if weak else max(1, B // world)
    el = statistics.median(els)
    strong = None
    if world > 1 and args.scaling == "both":
        sb = max(1, B // world)
        sels, _, _, _ = run(sb)
        sel = statistics.median(sels)
        strong = {"global_batch": sb * world, "per_gpu_batch": sb, "value": round(world * sb * args.steps / sel, 1), "unit": "samples/s",
                  "ms_per_step": round(sel / args.steps * 1e3, 4), "ms_per_step_repetitions": [round(e / args.steps * 1e3, 4) for e in sels]}

    # a step the f16 range guard rejected would have been DROPPED (loss kept on the device): never silently
    algo.check_guard()
    from gnn_rul_benchmarking_amd import _lib as _L
    on_mx = algo.model._last_chain == _L.STEP_MX
    dtype_name = "f32 (f16x2-split MFMA operands, fp32 accumulate)" if on_mx else "f32"
    line = None
    line_out = None
    if rank == 0:
        total = world * per_rank * args.steps
        out = {
            "metric": "training samples/sec, C-MAPSS FD004-shaped ST_GCN", "value": round(total / el, 1),
            "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(el / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
            "config": {"workload": f"ST_GCN.update (fwd+MSE+bwd+Adam), C-MAPSS FD004-shaped windows "
                                   f"[{NUM_PATCH} sensors x {args.patch_size}], per-GPU batch {per_rank}, dropout {args.dropout}, "
                                   f"lr {train_cfg['learning_rate']}, wd {train_cfg['weight_decay']}",
                       "per_gpu_batch": per_rank, "global_batch": world * per_rank, "num_patch": NUM_PATCH,
                       "patch_size": args.patch_size, "parallelism": f"dp{world}" + ("+syncbn" if args.sync_bn and use_dist else ""),
                       "batchnorm": ("synchronised over the ranks (global-batch statistics: the 1-GPU function)" if args.sync_bn
                                     else "local per-rank statistics (DDP default)") if use_dist else "single process",
                       "loss_readback": "every step" if args.sync_loss else "end of run (device-side loss each step)"},
            "repetitions": len(els), "ms_per_step_repetitions": [round(e / args.steps * 1e3, 4) for e in els],
            "timing": f"median of {len(els)} repetitions of {args.steps} steps, each between barrier + synchronize, max over ranks",
            "final_loss": round(final_loss, 6),
        }
        if strong is not None:
            out["strong_scaling"] = strong
        if not args.no_roofline:
            roof, roof_f = roofline_measurements(algo.model, Xs[0], ys[0], el / args.steps * 1e3, isolated=args.isolated_phases)
            out["roofline"] = roof
            out["roofline_forward"] = roof_f
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(NUM_PATCH, args.patch_size, args.dropout)
        if world == 1 and not args.no_roofline:
            # the reference reads the loss back every step (algorithms/algorithms.py:490); the headline keeps it on the device
            algo.sync_loss = True
            sl = statistics.median(timed_repetitions(lambda k: algo.update(Xs[k % 4], ys[k % 4], 1), args.steps, 2, 3, use_dist, dist, dev))
            algo.sync_loss = bool(args.sync_loss)
            out["with_per_step_loss_readback"] = {"ms_per_step": round(sl / args.steps * 1e3, 4), "value": round(per_rank * args.steps / sl, 1),
                                                  "unit": "samples/s", "note": "ST_GCN.update returning loss.item() every step like the reference"}
            fp = stgcn_train_other_shape(dev, NUM_PATCH, args.patch_size, [per_rank], fp32_batches=(per_rank,))[f"batch_{per_rank}"]
            out["fp32_chain_ms_per_step"] = fp["fp32_chain_ms_per_step"]
            out["fp32_chain"] = {"ms_per_step": fp["fp32_chain_ms_per_step"], "value": round(per_rank / (fp["fp32_chain_ms_per_step"] * 1e-3), 1),
                                 "unit": "samples/s", "vs_fp32_chain": fp["vs_fp32_chain"],
                                 "note": "the same step with every product in plain fp32 FMAs (RULGNN_STEP_CHAIN, the path a guard trip falls back to), "
                                         "same batch, dropout 0.2, timed beside the headline"}
            out["train_phm2012_40x64"] = dict(stgcn_train_other_shape(dev, 40, 64, [100, 16384, 65536], fp32_batches=(16384, 65536)),
                                              workload="ST_GCN.update at the reference's own PHM2012 wiring (40 patches x 64 points, configs/hparams.py:223,238): "
                                                       "wide matrix-core chain (one sample per wavefront in three column tiles, activations recomputed), the "
                                                       "fp32 phase chain of rounds 1-3 timed beside it",
                                              algorithmic_bytes_per_sample=algorithmic_bytes_per_sample(40, 64))
            out["train_reference_wirings_tiled"] = stgcn_tiled_shapes(dev)
        line_out = out
    if world == 1 and rank == 0 and not args.no_families and args.family == "ST_GCN":
        # the other four BASELINE.json configurations, each on its SURVEY section 8d wiring: same contract, compact
        del Xs, ys
        fams = {}
        for fam in ("FC_STGNN", "ASTGCNN", "HAGCN", "STMSGCN"):
            import copy
            fa = copy.copy(args)
            fa.steps, fa.warmup = (20, 5) if fam != "HAGCN" else (10, 3)
            d = family_line(fa, fam, world, rank, dev, False, dist, cpu_budget_s=9.0)
            r = d["roofline"]
            fams[fam] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                         "workload": d["config"]["workload"], "per_gpu_batch": d["config"]["per_gpu_batch"], "final_loss": d["final_loss"],
                         "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch", "launches_per_step",
                                                         "launches_in_step", "step_kernel_time_us", "share_of_step_kernel_time", "whole_step_estimate",
                                                         "work_model") if k in r},
                         "cpu_baseline": {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "runs")} if "cpu_baseline" in d else None}
        # BASELINE.json configs[1] is written "FC_STGNN ... bf16": the switch exists (bf16 operands on the row projections that run as GEMM
        # launches, fp32 everything else) and is timed here beside the fp32 path the entry above reports; at the FD004 sizes those projections
        # are fused kernels, so the variant is the same step -- stated, not hidden
        fb = copy.copy(args)
        fb.steps, fb.warmup, fb.dtype, fb.no_cpu_baseline, fb.no_roofline = 20, 5, "bf16", True, True
        db = family_line(fb, "FC_STGNN", world, rank, dev, False, dist)
        fams["FC_STGNN"]["bf16_variant"] = {"ms_per_step": db["ms_per_step"], "value": db["value"], "unit": db["unit"],
                                            "error_vs_f32": db.get("variant_error"),
                                            "note": "compute_dtype='bf16' (rulgnn_fcstgnn_args.compute_dtype): operands of the GEMM-launch row projections "
                                                    "rounded to bf16; tests/test_fcstgnn_gpu.py bounds it against the fp64 oracle"}
        line_out["families"] = fams
    if world == 1 and rank == 0 and not args.no_rmse:
        # (after the family lines: the torch-CPU halves of these legs leave the host's thread pool warm and spinning, which the
        # launch-bound FC_STGNN step right behind them paid for with 1.4 instead of 0.43 ms)
        line_out["rmse"] = rmse_teacher_task(dev)
        line_out["rmse"]["bn_free_family"] = rmse_teacher_task_stmsgcn(dev)
    if rank == 0:
        line = json.dumps(line_out)
    finish(line, use_dist, dist)


if __name__ == "__main__":
    main()
