"""-m gpu: data-parallel steps with the REAL HIP kernels in two processes (both on cuda:0, gloo backend: the bucket and the BatchNorm
cells travel through host staging) against single-process steps -- the rank-count dependent paths that a world-size-1 RCCL test or an
oracle-backed double cannot reach (VERDICT r3 missing 6): global / local counts, the BatchNorm parameter-gradient
scale (rank 0 only), zero joins of an empty shard, the order of the collectives on ragged shards.

* synchronised BatchNorm: the two-rank step IS the single-process step of the concatenated batch;
* local statistics (DDP default): the two-rank step is the sum of the two shards' single-process ``fused_mse_step`` buckets;
* ST_GCN (matrix-core chain at 14 x 30, its wide form at 20 x 30 and 40 x 64, fp32 chain at 21 x 30, tiled path with the overlapped all-reduce at 72 x 8), FC_STGNN, ASTGCNN."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(family, cfg, dev):
    from gnn_rul_benchmarking_amd import algorithms as A
    hp = {"learning_rate": 1e-3, "weight_decay": 1e-4}
    torch.manual_seed(3)
    algo = getattr(A, family)(dict(cfg), hp, dev)
    algo.to(dev)
    algo.train()
    return algo


def _data(cfg_shape, B, dev, label_scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(17)
    X = torch.rand((B,) + tuple(cfg_shape), generator=g).to(dev)
    y = (torch.rand((B, 1), generator=g) * label_scale).to(dev)
    return X, y


def _worker(rank, world, port, family, cfg, shape, B, sync_bn, overlap_min, out, scale=1.0, sync_loss=True, bn_collective="group"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_rul_benchmarking_amd.dp import DataParallel, shard_bounds
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        algo = _make(family, cfg, dev)
        dp = DataParallel(sync_bn=sync_bn, bn_collective=bn_collective)
        if overlap_min is not None:
            dp.OVERLAP_MIN_BYTES = overlap_min
        algo.attach_data_parallel(dp)
        X, y = _data(shape, B, dev, scale)
        lo, hi = shard_bounds(B, world, rank)
        algo.sync_loss = sync_loss
        losses = [algo.update(X[lo:hi], y[lo:hi], 1, global_batch=B, sample_offset=lo)["loss"] for _ in range(2)]
        losses = [float(v) for v in losses]
        torch.cuda.synchronize()
        bn = getattr(algo.model, "_bn", None)
        out[rank] = {"loss": losses, "flat": algo.model.flat_params.detach().cpu().numpy(),
                     "bn": bn.detach().cpu().numpy() if bn is not None else np.zeros(1, np.float32),
                     "shard": hi - lo, "regions": getattr(dp, "last_overlap_regions", None),
                     "step_path": int(getattr(algo.model, "step_path", -1)),
                     "trips": algo.model.guard_trips() if hasattr(algo.model, "guard_trips") else 0,
                     "peer_collectives": dp.peer.collectives() if dp.peer is not None else 0}
        if dp.peer is not None:
            dp.peer.check()
            dp.peer.close()
    finally:
        dist.destroy_process_group()


def _single_process(family, cfg, shape, B, sync_bn, world=2):
    """What the two ranks must reproduce, computed in this process."""
    from gnn_rul_benchmarking_amd.dp import shard_bounds
    dev = torch.device("cuda:0")
    algo = _make(family, cfg, dev)
    X, y = _data(shape, B, dev)
    m = algo.model
    batch_coupled = hasattr(m, "_after_train_forward")             # BatchNorm families
    by_offset = getattr(m, "dropout_by_sample_offset", False)      # RGCNU: dropout masks (and its adjacency pairing) follow the shard
    losses = []
    for _ in range(2):
        if sync_bn or not (batch_coupled or by_offset):
            losses.append(algo.update(X, y, 1)["loss"])                      # the function of the concatenated batch
            continue
        # shard-local statistics / pairings: every shard's own step, buckets summed, one optimizer step -- what dp.step does, without the
        # collective
        total = None
        step0 = getattr(m, "_step", 0)
        for r in range(world):
            lo, hi = shard_bounds(B, world, r)
            if hi == lo:
                continue
            if hasattr(m, "_step"):
                m._step = step0
            if batch_coupled:
                m.fused_mse_step(X[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo, update_running_stats=False, moments_to_bucket=True)
            else:
                m.fused_mse_step(X[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
            total = m.bucket.clone() if total is None else total + m.bucket
        m.bucket.copy_(total)
        algo.optimizer.step(from_bucket=True)
        if batch_coupled:
            m._after_train_forward(B, from_bucket_moments=True)
        losses.append(float(m.bucket[m.num_live]))
    torch.cuda.synchronize()
    bn = getattr(m, "_bn", None)
    return {"loss": losses, "flat": m.flat_params.detach().cpu().numpy(), "bn": bn.detach().cpu().numpy() if bn is not None else np.zeros(1, np.float32)}


STGCN_MX = ("ST_GCN", dict(num_patch=14, patch_size=30, dropout=0.2), (14, 30))
STGCN_MXW = ("ST_GCN", dict(num_patch=20, patch_size=30, dropout=0.2), (20, 30))          # the wide matrix-core chain (two column tiles)
STGCN_MXW40 = ("ST_GCN", dict(num_patch=40, patch_size=64, dropout=0.2), (40, 64))        # PHM2012's wiring (three column tiles)
STGCN_FP32 = ("ST_GCN", dict(num_patch=21, patch_size=30, dropout=0.2), (21, 30))         # 630 floats per window: not 16-byte pieces -> fp32 chain
STGCN_TILED = ("ST_GCN", dict(num_patch=72, patch_size=8, dropout=0.2), (72, 8))
STGCN_ORDER2 = ("ST_GCN", dict(num_patch=14, patch_size=30, dropout=0.2, k=2), (14, 30))  # MPNN order 2 (Model.py:74-90): fp32 chain, larger bucket
STGCN_ORDER3_W = ("ST_GCN", dict(num_patch=24, patch_size=16, dropout=0.2, k=3), (24, 16))  # order 3 on the 64-lane row mapping
def _hp_case(family):
    """The reference's FD004 wiring of the family (configs/hparams.py), as this package restates it."""
    from gnn_rul_benchmarking_amd import hparams as HP
    return (family, dict(HP.get_hparams_class("CMAPSS")("FD004").alg_hparams[family]), (14, 50))


def _run(case, B, sync_bn, overlap_min=None):
    family, cfg, shape = case
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), family, cfg, shape, B, sync_bn, overlap_min, out), nprocs=2, join=True)
    return out[0], out[1], _single_process(family, cfg, shape, B, sync_bn)


def _check(r0, r1, ref, tol):
    assert np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["bn"], r1["bn"])          # replicas stay identical
    assert r0["loss"] == r1["loss"]
    assert np.allclose(r0["loss"], ref["loss"], rtol=1e-5, atol=1e-7)
    # Adam's first steps move a parameter by ~lr whatever the gradient's size: compare on that scale
    assert np.max(np.abs(r0["flat"] - ref["flat"])) < tol
    assert np.allclose(r0["bn"], ref["bn"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("B", [37, 1])
@pytest.mark.parametrize("sync_bn", [False, True])
def test_stgcn_two_processes_equal_the_single_process_step(B, sync_bn):
    """Matrix-core chain (14 x 30): unequal shards (19 + 18) and an empty shard (1 + 0)."""
    r0, r1, ref = _run(STGCN_MX, B, sync_bn)
    assert r1["shard"] == B // 2
    _check(r0, r1, ref, 2e-4)


@pytest.mark.parametrize("sync_bn", [False, True])
def test_stgcn_fp32_chain_two_processes(sync_bn):
    r0, r1, ref = _run(STGCN_FP32, 23, sync_bn)
    _check(r0, r1, ref, 2e-4)


@pytest.mark.parametrize("case,B", [(STGCN_ORDER2, 37), (STGCN_ORDER2, 1), (STGCN_ORDER3_W, 11)])
@pytest.mark.parametrize("sync_bn", [False, True])
def test_stgcn_mpnn_order_above_one_two_processes(case, B, sync_bn):
    """k = 2, 3: the bucket grows by (k - 1)(N^2 + N) floats per layer; everything else as at k = 1."""
    r0, r1, ref = _run(case, B, sync_bn)
    _check(r0, r1, ref, 2e-4)


@pytest.mark.parametrize("case,B", [(STGCN_MXW, 23), (STGCN_MXW40, 13), (STGCN_MXW40, 1)])
@pytest.mark.parametrize("sync_bn", [False, True])
def test_stgcn_wide_matrix_core_chain_two_processes(case, B, sync_bn):
    r0, r1, ref = _run(case, B, sync_bn)
    _check(r0, r1, ref, 2e-4)


@pytest.mark.parametrize("B", [9, 1])
def test_stgcn_tiled_path_overlapped_all_reduce_two_processes(B):
    """num_patch > 64: the bucket leaves in gradient-ready regions on a side stream; an empty shard replays the same collectives."""
    r0, r1, ref = _run(STGCN_TILED, B, False, overlap_min=1024)
    assert r0["regions"] is not None and r0["regions"] == r1["regions"] and len(r0["regions"]) >= 3
    _check(r0, r1, ref, 2e-4)


@pytest.mark.parametrize("family", ["FC_STGNN", "ASTGCNN"])
@pytest.mark.parametrize("B", [11, 1])
def test_synchronised_batchnorm_families_two_processes(family, B):
    r0, r1, ref = _run(_hp_case(family), B, True)
    _check(r0, r1, ref, 3e-4)


# ---- the f16 range guard under data parallelism -------------------------------------------------------------------
def _run_scaled(case, B, sync_bn, scale, sync_loss):
    family, cfg, shape = case
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), family, cfg, shape, B, sync_bn, None, out, scale, sync_loss), nprocs=2, join=True)
    return out[0], out[1]


@pytest.mark.parametrize("sync_bn", [False, True])
@pytest.mark.parametrize("B", [37, 1])
def test_guard_trip_retries_on_the_fp32_chain_on_every_rank(sync_bn, B):
    """Labels of 1e15 (windows in [0, 1)): d loss / d pred ~ 1e14 leaves the f16 range of the matrix-core backward for certain, while the fp32
    phases take the same step with finite results (loss ~ 1e29, Adam's g^2 ~ 1e30).  Per-step loss read-back: the NaN travels in the all-reduced bucket, EVERY rank (also one whose shard is
    empty, B = 1) skips the guarded optimizer, repeats the step on the fp32 phases -- under synchronised BatchNorm too: the launch form
    is passed down (rulgnn_stgcn_train_fwdbwd_syncbn_path_f32) -- and ends with finite, identical replicas that equal a model that was
    on the fp32 chain all along."""
    from gnn_rul_benchmarking_amd import _lib
    r0, r1 = _run_scaled(STGCN_MX, B, sync_bn, 1.0e15, True)
    assert r0["step_path"] == r1["step_path"] == _lib.STEP_CHAIN
    assert np.all(np.isfinite(r0["loss"])) and r0["loss"] == r1["loss"]
    assert np.all(np.isfinite(r0["flat"])) and np.array_equal(r0["flat"], r1["flat"]) and np.array_equal(r0["bn"], r1["bn"])
    assert np.all(np.isfinite(r0["bn"]))
    # the single-process fp32-chain run of the same global batch (synchronised BatchNorm = the function of the concatenated batch)
    if sync_bn:
        dev = torch.device("cuda:0")
        ref = _make(*STGCN_MX[:2], dev)
        ref.model.step_path = _lib.STEP_CHAIN
        X, y = _data(STGCN_MX[2], B, dev, 1.0e15)
        want = [ref.update(X, y, 1)["loss"] for _ in range(2)]
        assert np.allclose(r0["loss"], want, rtol=1e-5)
        assert np.max(np.abs(r0["flat"] - ref.model.flat_params.detach().cpu().numpy())) < 2e-4


def test_guard_trip_without_loss_readback_is_counted_on_every_rank():
    """``sync_loss=False``: no retry is possible; the dropped steps are counted from the all-reduced loss, the same number on both ranks."""
    r0, r1 = _run_scaled(STGCN_MX, 37, False, 1.0e15, False)
    assert r0["trips"] == r1["trips"] == 2 and np.all(np.isnan(r0["loss"]))
    assert np.array_equal(r0["flat"], r1["flat"]) and np.all(np.isfinite(r0["flat"]))


# ---- sharded evaluation -------------------------------------------------------------------------------------------
def _eval_worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_rul_benchmarking_amd.dataloader import DeviceBatchLoader
        from gnn_rul_benchmarking_amd.dp import DataParallel
        from gnn_rul_benchmarking_amd.trainer import gather_shards, sharded_metrics
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        algo = _make(*STGCN_MX[:2], dev)
        X, y = _data(STGCN_MX[2], n, dev)
        dl = DeviceBatchLoader(X.cpu(), y.cpu(), 16, False, False, dev, rank, world, shard_samples=True)
        model = algo.model
        model.eval()
        preds, reals = [], []
        with torch.no_grad():
            for xb, yb, _, _ in dl:
                preds.append(model(xb).view(-1)); reals.append(yb.view(-1))
        pred = torch.cat(preds) if preds else torch.empty(0, device=dev)
        real = torch.cat(reals) if reals else torch.empty(0, device=dev)
        dp = DataParallel()
        m = sharded_metrics(pred, real, 125.0, dp)
        full = gather_shards(pred, dl, dp)
        out[rank] = {"metrics": m, "full": full.cpu().numpy(), "shard": dl.shard}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [75, 1])
def test_sharded_evaluation_equals_the_single_process_evaluation(n):
    """Each rank evaluates its contiguous shard of the test set with the real eval kernel, reduces it on the device to four fp64 sums, one
    all-reduce: the metrics equal the single-process ``device_metrics`` of the whole set to 1e-12, and the gathered predictions are the
    single-process predictions bit for bit (an eval forward does not depend on its batch mates)."""
    from gnn_rul_benchmarking_amd.metrics import device_metrics
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_eval_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    dev = torch.device("cuda:0")
    algo = _make(*STGCN_MX[:2], dev)
    X, y = _data(STGCN_MX[2], n, dev)
    algo.model.eval()
    with torch.no_grad():
        pred = algo.model(X).view(-1)
    want = device_metrics(pred, y.view(-1), 125.0)
    assert out[0]["metrics"] == out[1]["metrics"]
    assert np.allclose(out[0]["metrics"], want, rtol=1e-12, atol=0)
    assert np.array_equal(out[0]["full"], pred.cpu().numpy()) and np.array_equal(out[1]["full"], out[0]["full"])
    assert out[0]["shard"] == (0, (n + 1) // 2) and out[1]["shard"] == ((n + 1) // 2, n)


# ---- the families without BatchNorm, and the local-statistics legs of the BatchNorm families -----------------------------
STMSGCN_PHM = ("STMSGCN", dict(num_patch=9, patch_size=20, interval=2, band_width=3, gcn_dims=[16, 64, 16, 1], gru_hidden_dim=8), (1, 180))


@pytest.mark.parametrize("B", [9, 1])
def test_stmsgcn_two_processes_equal_the_single_process_step(B):
    """No BatchNorm, no dropout: the two-rank step IS the single-process step of the whole batch ("best model for exact
    data-parallel parity"); 5 + 4 samples and an empty shard."""
    r0, r1, ref = _run(STMSGCN_PHM, B, False)
    _check(r0, r1, ref, 2e-4)


@pytest.mark.parametrize("B", [11, 1])
def test_rgcnu_two_processes_dropout_by_sample_offset(B):
    """RGCNU: no BatchNorm; its dropout masks are indexed by the GLOBAL sample (sample_offset) and its adjacency pairing follows the shard
    (rgcnu.py): the two-rank step equals the sum of the two shards' own steps."""
    r0, r1, ref = _run(_hp_case("RGCNU"), B, False)
    _check(r0, r1, ref, 3e-4)


@pytest.mark.parametrize("family", ["FC_STGNN", "ASTGCNN"])
@pytest.mark.parametrize("B", [11, 1])
def test_local_batchnorm_families_two_processes(family, B):
    """DDP's default (rank-local batch statistics, moments averaged for the running statistics) for the other two BatchNorm families."""
    r0, r1, ref = _run(_hp_case(family), B, False)
    _check(r0, r1, ref, 3e-4)


# ---- device-side one-shot all-reduce of the BatchNorm cells (csrc/peer_comm.hip): no host callback between the phases of a step ----------
def _run_peer(case, B):
    family, cfg, shape = case
    mgr = mp.Manager()
    out_g, out_p = mgr.dict(), mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), family, cfg, shape, B, True, None, out_g), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), family, cfg, shape, B, True, None, out_p, 1.0, True, "peer"), nprocs=2, join=True)
    return out_g, out_p


@pytest.mark.timeout(600)
@pytest.mark.parametrize("case,B,collectives_per_step", [(STGCN_MX, 37, 8), (STGCN_MX, 1, 8), (STGCN_MXW40, 6, 8), (STGCN_FP32, 23, 8),
                                                         (_hp_case("FC_STGNN"), 9, 14), (_hp_case("ASTGCNN"), 11, 4)])
def test_peer_mailbox_all_reduce_equals_the_process_group_collectives(case, B, collectives_per_step):
    """``DataParallel(sync_bn=True, bn_collective="peer")``: the 4 L (ST_GCN), 14 (FC_STGNN), 4 (ASTGCNN) BatchNorm reductions of a step
    as single-workgroup launches over IPC-mapped mailboxes (two processes on one GPU here) -- BIT-equal to the same two-rank step through
    the process group's all-reduce (two ranks: a + b in either order is the same double), replicas identical, an empty shard (B = 1) joins
    with zeros, and no collective timed out."""
    g, p = _run_peer(case, B)
    for r in (0, 1):
        assert np.array_equal(p[r]["flat"], g[r]["flat"]) and np.array_equal(p[r]["bn"], g[r]["bn"]) and p[r]["loss"] == g[r]["loss"]
        assert p[r]["peer_collectives"] == 2 * collectives_per_step and g[r]["peer_collectives"] == 0
    assert np.array_equal(p[0]["flat"], p[1]["flat"])


def _peer_stress_worker(rank, world, port, iters, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_rul_benchmarking_amd.dp import PeerAllReduce
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        peer = PeerAllReduce()
        bad = 0
        g = torch.Generator(device="cpu").manual_seed(5)
        base = torch.rand(iters, 128, generator=g, dtype=torch.float64)          # the same table on every rank
        for k in range(iters):
            n = 1 + (k * 37) % 128                                               # every count 1..128, both parities of the slot ring
            mine = (base[k, :n] * (rank + 1)).to(dev)
            peer(mine)
            want = base[k, :n] * sum(r + 1 for r in range(world))
            if k % 50 == 0 and rank == 1:
                torch.cuda.synchronize()                                         # let the ranks drift apart now and then
            bad += int(not torch.equal(mine.cpu(), want))
        torch.cuda.synchronize()
        peer.check()
        out[rank] = {"bad": bad, "collectives": peer.collectives()}
        peer.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_peer_mailbox_all_reduce_hundreds_of_back_to_back_collectives():
    """600 one-shot all-reduces of 1..128 doubles back to back from two processes (no host synchronisation in between except where
    one rank deliberately lags): every sum exact (k * (1 + 2) of the same table: the same doubles in rank order), no collective timed
    out -- the two-parity slot ring never hands a rank the data of the wrong collective."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_peer_stress_worker, args=(2, _free_port(), 600, out), nprocs=2, join=True)
    assert out[0] == {"bad": 0, "collectives": 600} and out[1] == {"bad": 0, "collectives": 600}


def test_peer_mailbox_all_reduce_gives_up_on_a_missing_peer_within_its_timeout():
    """A world of two whose second rank never arrives (its mailbox is a second local allocation nobody writes): with a 50-ms timeout
    the collective returns NaN -- not a hung GPU -- and the communicator reports which collective gave up; the next one (the peer still
    missing) does the same; the argument checks of the timeout setter."""
    import ctypes as C
    import time
    from gnn_rul_benchmarking_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    hb = lib.rulgnn_peer_handle_bytes()
    boxes = (C.c_void_p * 2)()
    for r in range(2):
        box, handle = C.c_void_p(), (C.c_ubyte * hb)()
        _lib.check(lib.rulgnn_peer_mailbox_alloc(C.byref(box), handle), "rulgnn_peer_mailbox_alloc")
        boxes[r] = box.value
    comm = lib.rulgnn_peer_comm_create(0, 2, boxes)
    assert comm
    assert lib.rulgnn_peer_comm_set_timeout_ms(comm, 0) == _lib.EINVAL and lib.rulgnn_peer_comm_set_timeout_ms(comm, 10 ** 7) == _lib.EINVAL
    assert lib.rulgnn_peer_comm_set_timeout_ms(comm, 50) == _lib.OK
    buf = torch.arange(20, dtype=torch.float64, device="cuda:0")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    for q in (1, 2):
        assert lib.rulgnn_peer_allreduce_f64(comm, buf.data_ptr(), 20, st) == _lib.OK
        torch.cuda.synchronize()
        assert bool(torch.isnan(buf).all())
        assert lib.rulgnn_peer_comm_status(comm) == q
        buf.copy_(torch.arange(20, dtype=torch.float64))
    assert time.perf_counter() - t0 < 5.0
    lib.rulgnn_peer_comm_destroy(comm)
    for r in range(2):
        assert lib.rulgnn_peer_mailbox_free(boxes[r]) == _lib.OK
