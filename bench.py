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

# the legs of the line live in benchlib/ (re-exported here: tests and tools import them from `bench`)
from benchlib.common import (HBM_PEAK_GBS, NUM_PATCH, FP32_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS, event_time_ms,  # noqa: E402,F401
                             algorithmic_bytes_per_sample, forward_flops_per_sample, compute_leg)
from benchlib.roofline import (phase_names, phase_bytes_per_sample, measured_traffic, forward_traffic, time_eval_forward,  # noqa: E402,F401
                               roofline_measurements)
from benchlib.rmse import rmse_teacher_task, rmse_teacher_task_stmsgcn  # noqa: E402,F401
from benchlib.shapes import stgcn_train_other_shape, stgcn_tiled_shapes  # noqa: E402,F401
from benchlib.cpu import cpu_baseline, family_cpu_baseline, family_torch_cpu_baseline  # noqa: E402,F401
from benchlib.common import kernel_times, kernel_short_name  # noqa: E402,F401
from benchlib.families import FAMILY_CONFIGS, family_line, family_main, dominant_kernel_work  # noqa: E402,F401


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
    ap.add_argument("--bn-collective", choices=["group", "peer"], default="group",
                    help="with --sync-bn: the BatchNorm reductions through the process group (RCCL: a host callback per collective) or as "
                         "device-side one-shot all-reduces over IPC-mapped mailboxes (csrc/peer_comm.hip; one node)")
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both",
                    help="N > 1: weak = fixed per-GPU batch (headline), strong = fixed global batch split over the ranks")
    ap.add_argument("--patch-size", type=int, default=30, help="window length (BASELINE.json: 30)")
    ap.add_argument("--dropout", type=float, default=0.2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-families", action="store_true", help="skip the other four BASELINE.json configurations in the default line")
    ap.add_argument("--no-rmse", action="store_true", help="skip the teacher-task RMSE leg (the other half of BASELINE.json's metric)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-shapes", action="store_true",
                    help="skip the legs at other batches / windows (batch 100, 14 x 50, 40 x 64, the tiled wirings): a rocprofv3 --stats summary of such a "
                         "run averages every kernel over the headline batch only (tools/profile_r06.sh)")
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


def data_parallel_diagnostics(algo, Xs, ys, args, dist, dev, world, step_ms):
    """What the first multi-GPU run needs to be diagnostic (every rank takes part: collectives inside): the backend and its rank count,
    the step's collectives and their sizes, the latency of ONE bucket all-reduce and ONE BatchNorm-cell reduction measured alone on this
    world (HIP events around 50 back-to-back calls, max over ranks), and the host time to ENQUEUE a step (no synchronisation: it must stay
    below the device step time or the host, not the GPUs, sets the rate)."""
    import time as _t
    m, dp = algo.model, algo.dp
    def max_over_ranks(v):
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    bucket = m.bucket.clone()
    def timed(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize(); dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return max_over_ranks(e0.elapsed_time(e1) / n * 1e3)
    bucket_us = timed(lambda: dist.all_reduce(bucket, op=dist.ReduceOp.SUM))
    cells = torch.zeros(20, dtype=torch.float64, device=dev)
    group_us = timed(lambda: dist.all_reduce(cells, op=dist.ReduceOp.SUM))
    peer_us = timed(lambda: dp.peer(cells)) if dp.peer is not None else None
    # host enqueue time of a step: the loop below returns as soon as everything is queued
    torch.cuda.synchronize(); dist.barrier()
    n = max(10, args.steps)
    t0 = _t.perf_counter()
    for k in range(n):
        algo.update(Xs[k % len(Xs)], ys[k % len(ys)], 1)
    host_ms = (_t.perf_counter() - t0) / n * 1e3
    torch.cuda.synchronize()
    host_ms = max_over_ranks(host_ms)
    if dp.peer is not None:
        dp.peer.check()
    L = m.num_layers
    return {"backend": dist.get_backend(), "ranks_in_group": dist.get_world_size(), "world_size_env": world,
            "batchnorm": ("syncbn via " + ("device-side one-shot all-reduce over IPC mailboxes (csrc/peer_comm.hip)" if dp.peer is not None
                                          else "process-group all-reduce (host callback per collective)")) if args.sync_bn else "local statistics",
            "collectives_per_step": {"bucket_all_reduce": 1, "bucket_bytes": int(m.bucket.numel() * 4),
                                     "batchnorm_cell_all_reduces": 4 * L if args.sync_bn else 0, "cell_bytes": 160},
            "bucket_all_reduce_us": round(bucket_us, 1), "cell_all_reduce_us_process_group": round(group_us, 1),
            "cell_all_reduce_us_peer_mailboxes": round(peer_us, 1) if peer_us is not None else None,
            "host_enqueue_ms_per_step": round(host_ms, 4), "device_ms_per_step": round(step_ms, 4),
            "host_bound": bool(host_ms > step_ms),
            "timing": "collectives: HIP events around 50 back-to-back calls on this world, max over ranks; host: wall clock of enqueueing "
                      f"{n} steps without synchronisation, max over ranks"}


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
    # Test hook (tests/test_bench_multirank_gpu.py): RULGNN_BENCH_SHARE_GPU=1 lets the ranks of an N > 1 launch share the visible GPUs
    # over the gloo backend, so that the multi-rank path of this script (sharding, barriers, max over ranks, one JSON line) can be
    # exercised on a one-GPU box.  Never set by the driver; the numbers of such a run are not a measurement.
    share_gpu = os.environ.get("RULGNN_BENCH_SHARE_GPU") == "1"
    if local_rank >= torch.cuda.device_count() and not share_gpu:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({torch.cuda.device_count()} visible)")
    dev_index = local_rank % torch.cuda.device_count() if share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

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
        if share_gpu:
            dist.init_process_group("gloo")
        else:
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
        algo.attach_data_parallel(DataParallel(sync_bn=args.sync_bn, bn_collective=args.bn_collective))

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
    dp_diag = None
    if use_dist:
        dp_diag = data_parallel_diagnostics(algo, Xs, ys, args, dist, dev, world, el / args.steps * 1e3)
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
        if dp_diag is not None:
            out["data_parallel"] = dp_diag
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
        if world == 1 and not args.no_roofline and not args.no_other_shapes:
            # the reference protocol's batch (configs/hparams.py:16-27: batch_size 100): ten dependent launches at their latency floor
            out["train_batch_100"] = dict(stgcn_train_other_shape(dev, NUM_PATCH, args.patch_size, [100], steps=50, fp32_batches=(100,),
                                                                   single_launch_batches=(100,)),
                                          workload="ST_GCN.update at the reference protocol's batch (100) at 14 x 30, dropout 0.2: the matrix-core chain as ten "
                                                   "launches (what RULGNN_STEP_AUTO runs), the fp32 chain, and F_1 .. G_0 as ONE launch behind arrival counters "
                                                   "(RULGNN_STEP_MX_PERSIST: built, measured slower, explicit option only -- profiles/r06_notes.md section 4)")
            # the reference's real C-MAPSS window is 50 points (Data_Process/Data_read_CMAPSS.py:330; BASELINE.json names 30): the same step at 14 x 50
            out["train_cmapss_14x50"] = dict(stgcn_train_other_shape(dev, NUM_PATCH, 50, [per_rank], fp32_batches=(per_rank,)),
                                             workload="ST_GCN.update at the reference's own C-MAPSS window (14 sensors x 50 points), same batch, dropout 0.2: "
                                                      "matrix-core chain, fp32 chain beside it",
                                             algorithmic_bytes_per_sample=algorithmic_bytes_per_sample(NUM_PATCH, 50))
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
        import copy
        # BASELINE.json configs[1] is written "FC_STGNN ... bf16": the variant (every product of the window-graph kernels, forward and backward,
        # on v_mfma_f32_32x32x16_bf16 with bf16-rounded operands; fp32 accumulate / softmax / BatchNorm / weight gradients / Adam) is timed here
        # beside the fp32 path the entry above reports; it does NOT meet the 1e-4 gate (its error is in the line) -- the parity claim is fp32
        fb = copy.copy(args)
        fb.steps, fb.warmup, fb.dtype, fb.no_cpu_baseline, fb.no_roofline = 100, 10, "bf16", True, True
        db = family_line(fb, "FC_STGNN", world, rank, dev, False, dist)
        bf16_variant = {"ms_per_step": db["ms_per_step"], "value": db["value"], "unit": db["unit"],
                                            "error_vs_f32": db.get("variant_error"),
                                            "dtype": "bf16 operands (v_mfma_f32_32x32x16_bf16), fp32 accumulate",
                                            "note": "compute_dtype='bf16' (rulgnn_fcstgnn_args.compute_dtype): the products of the window-graph kernels "
                                                    "(mapping, M M^T, A.X, the block projection and their backward forms: csrc/fcstgnn.hip fc_prod<true>) and the "
                                                    "GEMM-launch row projections on bf16 operands; does not meet the 1e-4 gate -- "
                                                    "tests/test_fcstgnn_gpu.py bounds it against the fp64 oracle at FD004 batch 256 (<= 1e-2)"}
        # (timed FIRST among the family legs, like the fp32 leg it is compared with: behind the torch-CPU baselines of the later families the
        # host's thread pool is still spinning and this launch-bound step paid 0.37-0.41 ms for it instead of 0.31)
        fams = {}
        for fam in ("FC_STGNN", "ASTGCNN", "HAGCN", "STMSGCN"):
            import copy
            fa = copy.copy(args)
            # (sub-millisecond steps: 100 of them, so that the closing synchronize -- ~30-50 us -- is a fraction of a percent of the region)
            fa.steps, fa.warmup = {"HAGCN": (10, 3), "STMSGCN": (40, 5)}.get(fam, (100, 10))
            d = family_line(fa, fam, world, rank, dev, False, dist, cpu_budget_s=9.0)
            r = d["roofline"]
            fams[fam] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                         "workload": d["config"]["workload"], "per_gpu_batch": d["config"]["per_gpu_batch"], "final_loss": d["final_loss"],
                         "roofline": {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "peak_basis", "mfma_util", "unit", "frac", "traffic", "us_per_launch", "launches_per_step",
                                                         "launches_in_step", "step_kernel_time_us", "share_of_step_kernel_time", "whole_step_estimate",
                                                         "work_model") if k in r},
                         "cpu_baseline": {k: d["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "port_of", "sample", "runs") if k in d["cpu_baseline"]} if "cpu_baseline" in d else None}
        fams["FC_STGNN"]["bf16_variant"] = bf16_variant
        line_out["families"] = fams
    if world == 1 and rank == 0 and not args.no_rmse:
        # (after the family lines: the torch-CPU halves of these legs leave the host's thread pool warm and spinning, which the
        # launch-bound FC_STGNN step right behind them paid for with 1.4 instead of 0.43 ms)
        line_out["rmse"] = rmse_teacher_task(dev)
        # ... and with dropout ON (the reference's protocol trains with dropout 0.2: configs/hparams.py), both paths under the same masks
        rd = rmse_teacher_task(dev, epochs=10, checkpoints=(36, 120), dropout=0.2)
        line_out["rmse"]["with_dropout"] = {k: rd[k] for k in ("task", "rmse_hip", "rmse_torch_cpu", "abs_diff", "within_1e-3", "after_steps", "steps")}
        line_out["rmse"]["bn_free_family"] = rmse_teacher_task_stmsgcn(dev)
    if rank == 0:
        line = json.dumps(line_out)
    finish(line, use_dist, dist)


if __name__ == "__main__":
    main()

