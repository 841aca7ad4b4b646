"""`bench.py --family NAME` and the `families` entry of the default line: the other model families on their SURVEY section 8(d) configurations.

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
from .common import (FP32_MFMA_PEAK_TFLOPS, BF16_MFMA_PEAK_TFLOPS, F16X2_SPLIT_PEAK_TFLOPS, HBM_PEAK_GBS, kernel_short_name, kernel_times,
                     mfma_util_from_profile)
from .cpu import family_cpu_baseline


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


def _stmsgcn_nodes(cfg):
    """Graph nodes of STMSGCN = energy bands of a patch's lagged spectrum (models/STMSGCN/Model.py:7-31)."""
    return (cfg["patch_size"] - cfg["interval"]) // cfg["band_width"]


def _gcn_stack_flops(n, dims):
    """Forward matmul FLOPs of STMSGCN's GCN stack for ONE graph of n nodes (models/STMSGCN/Model.py:84-112 without the dense
    diag products): per layer the Gram matrix x x^T and A.x (2 n^2 f each) and the Linear (2 n f_in f_out)."""
    f = [1] + list(dims)
    return sum(2 * 2 * n * n * f[l] + 2 * n * f[l] * f[l + 1] for l in range(len(dims)))


def _fc_blocks(per_step):
    return "both window blocks in the one launch (blockIdx.y)" if per_step < 1.5 else "averaged over the two window blocks' launches"


# Work model of the kernel that dominates each family's step: substring of the kernel name -> f(cfg, batch, launches per step)
# = (algorithmic FLOPs of ONE launch, how they are counted).  Matmul-type FLOPs only, as SURVEY 8(d) counts them.
def dominant_kernel_work(family, name, cfg, B, shape, per_step):
    if family == "FC_STGNN" and "fc_graph_bwd" in name:
        Q, D2 = 2 * cfg["num_node"], 2 * cfg["hidden_dim"]
        graphs = B * ((cfg["num_patch"] - 1) + (cfg["num_patch"] - 2) // 2 + 1)            # window 2, stride 1 and 2 (Model_Base.py:175-225)
        return graphs / per_step * 6 * Q * Q * D2, ("backward of one window graph: dA = dAX X'^T, dX' = A^T dAX, dM = (dS + dS^T) M, "
                                                    "2 Q^2 D FLOPs each (Q = 28 nodes, D = 16); " + _fc_blocks(per_step))
    if family == "FC_STGNN" and ("fc_graph_kernel" in name or "fc_graph_mx" in name or "fc_block_mx" in name):
        Q, D2 = 2 * cfg["num_node"], 2 * cfg["hidden_dim"]
        graphs = B * ((cfg["num_patch"] - 1) + (cfg["num_patch"] - 2) // 2 + 1)
        extra = (2 * Q * D2 * D2 + 2 * Q * D2 * (D2 // 2)) if "fc_block_mx" in name else 0            # mapping + the block's Linear
        return graphs / per_step * (4 * Q * Q * D2 + extra), ("forward of one window graph: S = M M^T and A X', 2 Q^2 D FLOPs each (Q = 28 nodes, D = 16)"
                                                              + ("; plus the mapping F W_map^T and the block's Linear" if extra else "")
                                                              + "; " + _fc_blocks(per_step))
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
                         "note": "bf16 operands on every product of the window-graph kernels (forward and backward) and on the GEMM-launch row projections; fp32 "
                                 "accumulate / softmax / BatchNorm / weight gradients / Adam; tests/test_fcstgnn_gpu.py bounds it against the fp64 oracle at full size"}
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
            roof["peak_basis"] = "fp32 matrix peak (157.3 TFLOP/s): the kernel issues v_mfma_f32_*_f32"
            if "f16x2" in short or "sgemm_planes" in short:
                peak = F16X2_SPLIT_PEAK_TFLOPS
                roof["peak_basis"] = "dense f16 matrix peak (2500 TFLOP/s) / 3 matrix instructions per fp32-class product block; FLOPs counted once"
            if args.dtype == "bf16" and ("fc_block_mx" in short or "fc_graph_bwd_mx" in short):
                peak = BF16_MFMA_PEAK_TFLOPS
                roof["peak_basis"] = "dense bf16 matrix peak (2500 TFLOP/s): the kernel issues v_mfma_f32_32x32x16_bf16"
            if "bf16x3" in short:
                # fp32-class products out of six bf16 matrix instructions each (csrc/sgemm_mfma.hpp): the ceiling is a sixth of the dense bf16 peak
                peak = round(BF16_MFMA_PEAK_TFLOPS / 6.0, 1)
                roof["peak_basis"] = "dense bf16 matrix peak (2500 TFLOP/s, MI355X_MICROARCH.md) / 6 matrix instructions per fp32-class product block; FLOPs counted once"
            roof.update({"achieved": round(ach, 4), "peak": peak, "frac": round(ach / peak, 5), "flops_per_launch": round(work), "work_model": how})
            if generic_above:
                roof["generic_gemm_instances_with_larger_share"] = generic_above
            mu = mfma_util_from_profile(short.split("<")[0]) if args.dtype == "f32" else None
            if mu:
                roof["mfma_util"] = mu
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


