"""Roofline legs of the ST_GCN line: per-phase algorithmic bytes, committed PMC traffic, live kernel timing (bench.py: `roofline`, `roofline_forward`).

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
import ctypes as C

from .common import (HBM_PEAK_GBS, NUM_PATCH, FP32_MFMA_PEAK_TFLOPS, event_time_ms, algorithmic_bytes_per_sample, forward_flops_per_sample, compute_leg,
                     kernel_short_name, kernel_times, mfma_util_from_profile)


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
    for name in ("r06_hbm_traffic.json", "r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_g_hbm_traffic.json"):
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
    for rnd in ("r06", "r05", "r04", "r03", "r02"):       # newest round first
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_forward_bs{B}_hbm_traffic.json")))
            w = t["workload"]
            if (w["num_patch"], w["patch_size"], w["batch"]) == (N, P, B):
                return round(t["kernels"]["EVAL"]["hbm_bytes_per_launch"])
        except Exception:
            continue
    return None


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
    # (under an external profiler -- rocprofv3 owns the activity tracer -- torch's profiler returns microsecond-long stubs: a traced
    # duration below a third of the event interval of the same phase is not believed, the event timing stands)
    if set(traced) == set(names) and any(traced[ph] * 1e-3 < per[ph]["ms"] / 3 for ph in names):
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
              "compute": compute_leg(forward_flops_per_sample(N, P, L), B / (fms * 1e-3), "one eval forward per sample", num_patch=N)}
    roof_f["mfma_util"] = mfma_util_from_profile("stgcn_forward_mx_kernel")
    roof["compute"] = compute_leg(3 * forward_flops_per_sample(N, P, L), B / (step_ms * 1e-3),
                                  "3 x the forward FLOPs per sample (SURVEY section 8d), whole step; recomputed products not counted",
                                  num_patch=N, passes=9.0 if chain_kind == "mx" else 0.0)
    roof["mfma_util"] = mfma_util_from_profile("stgcn_train_mx_kernel<2, 2, 2")
    if big_forward:
        BB = 1 << 20
        g = torch.Generator(device=X.device).manual_seed(99)
        Xb = torch.rand(BB, N, P, device=X.device, generator=g)
        bms = time_eval_forward(model, Xb, iters=5)
        bach = alg * BB / (bms * 1e-3) / 1e9
        roof_f["at_1M"] = {"batch": BB, "us_per_launch": round(bms * 1e3, 1), "achieved": round(bach, 1),
                           "frac": round(bach / HBM_PEAK_GBS, 4), "samples_per_s": round(BB / (bms * 1e-3), 1),
                           "traffic": forward_traffic(N, P, BB),
                           "compute": compute_leg(forward_flops_per_sample(N, P, L), BB / (bms * 1e-3), "one eval forward per sample", num_patch=N)}
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
                                  "compute": compute_leg(forward_flops_per_sample(N, 50, L), X50.size(0) / (cms * 1e-3), "one eval forward per sample", num_patch=N)}
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
                                   "compute": compute_leg(forward_flops_per_sample(WN, WP, 2), WB / (wms * 1e-3), "one eval forward per sample", num_patch=WN)}
        del Xw, wide
    return roof, roof_f


