"""world_size-2 gloo tests (CPU) of the data-parallel step: sharding, the single bucket all-reduce,
loss / gradient / BatchNorm-moment semantics, replica consistency.  The HIP model is replaced by a
test double that fills the same bucket with the numpy oracle (the oracle is test infrastructure)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gnn_rul_benchmarking_amd import params as PL
from gnn_rul_benchmarking_amd.dp import DataParallel, shard_bounds
from oracle import stgcn_oracle as O

N, P, L = 14, 30, 2


def test_shard_bounds_cover_ragged_batches():
    for n in (0, 1, 7, 100, 101, 65536 + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


class OracleModel:
    """Duck-types the slice of ST_GCN_model that dp.DataParallel touches."""

    def __init__(self, prm, dropout=0.2, seed=7):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.num_live = PL.param_count(N, L)
        self.bucket = torch.zeros(self.num_live + 1 + PL.bn_buffer_count(L), dtype=torch.float32)
        flat, bn = PL.pack_numpy(prm, N, L)
        self.flat_params, self._bn = torch.from_numpy(flat.copy()), torch.from_numpy(bn.copy())
        self._nbt = torch.zeros(2 * L, dtype=torch.int64)
        self.dropout, self.seed, self._step = dropout, seed, 0

    def fused_mse_step(self, X, y, global_batch=None, sample_offset=0, update_running_stats=True, moments_to_bucket=False):
        self._step += 1
        keys = [O.dropout_layer_key(self.seed, self._step, l) for l in range(L)]
        x = X.numpy().astype(np.float64)
        fc = O.forward(self.prm, x, N, P, L, train=True, dropout=self.dropout, dropout_keys=keys, sample_offset=sample_offset)
        loss, dpred = O.mse_loss_and_grad(fc.pred, y.numpy().astype(np.float64), global_batch)
        g = O.backward(self.prm, fc, dpred, self.dropout)
        for name, (off, shape) in PL.live_param_layout(N, L).items():
            self.bucket[off:off + int(np.prod(shape))] = torch.from_numpy(g[name].reshape(-1).astype(np.float32))
        self.bucket[self.num_live] = loss
        assert moments_to_bucket and not update_running_stats
        w = x.shape[0] / float(global_batch)
        tail = self.bucket[self.num_live + 1:]
        for l in range(L):
            lc = fc.layers[l]
            for b, z in enumerate((lc.z1, lc.z2)):
                base = ((l * 2 + b) * 2) * 10
                tail[base:base + 10] = torch.from_numpy((w * z.mean(axis=(0, 2))).astype(np.float32))
                tail[base + 10:base + 20] = torch.from_numpy((w * (z * z).mean(axis=(0, 2))).astype(np.float32))
        self.last_fc = fc
        return None, self.bucket[self.num_live]

    def _after_train_forward(self, batch, from_bucket_moments=False):
        assert from_bucket_moments
        self.global_moments = self.bucket[self.num_live + 1:].clone()
        self._nbt += 1


class SgdFromBucket:
    def __init__(self, model):
        self.model = model

    def step(self, from_bucket=False):
        assert from_bucket
        self.model.flat_params -= 0.1 * self.model.bucket[:self.model.num_live]


def _worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        prm = O.random_params(N, L, seed=rank * 13)          # ranks start DIFFERENT on purpose
        x = torch.from_numpy(rng.uniform(0, 1, (B, N, P)).astype(np.float32))
        y = torch.from_numpy(rng.uniform(0, 1, (B, 1)).astype(np.float32))
        model = OracleModel(prm)
        dp = DataParallel()
        assert (dp.rank, dp.world_size) == (rank, world)
        dp.broadcast_model(model)                              # now identical to rank 0
        model.prm = {k: np.asarray(v, np.float64) for k, v in O.random_params(N, L, seed=0).items()}
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "flat": model.flat_params.clone().numpy(),
                     "moments": model.global_moments.numpy(), "pred": model.last_fc.pred.copy(), "lo": lo, "hi": hi}
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("B", [64, 65])
def test_data_parallel_step_world2_gloo(B):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    # replicas hold the same reduced bucket and therefore the same parameters after the step
    assert np.array_equal(r0["bucket"], r1["bucket"])
    assert np.array_equal(r0["flat"], r1["flat"])
    # the reduced loss is the global-batch MSE of the (locally normalised) predictions
    prm = O.random_params(N, L, seed=0)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (B, N, P)).astype(np.float32)
    y = rng.uniform(0, 1, (B, 1)).astype(np.float32)
    pred = np.concatenate([r0["pred"], r1["pred"]])
    assert abs(r0["loss"] - float(np.mean((pred - y) ** 2))) < 1e-5
    # BatchNorm 0 sees the same z in every sharding: its reduced moments equal the single-process batch statistics
    fc = O.forward(prm, x.astype(np.float64), N, P, L, train=True, dropout=0.0)
    z = fc.layers[0].z1
    assert np.allclose(r0["moments"][0:10], z.mean(axis=(0, 2)), rtol=1e-5, atol=1e-6)
    assert np.allclose(r0["moments"][10:20], (z * z).mean(axis=(0, 2)), rtol=1e-5, atol=1e-6)
    # dropout masks do not depend on the sharding: shard 1 used stream positions [lo, hi) of the global batch
    keys = [O.dropout_layer_key(7, 1, l) for l in range(L)]
    full = O.dropout_keep_mask(B, N, keys[0], 0.2)
    part = O.dropout_keep_mask(r1["hi"] - r1["lo"], N, keys[0], 0.2, sample_offset=r1["lo"])
    assert np.array_equal(full[r1["lo"]:r1["hi"]], part)


def test_loader_shards_partition_every_batch():
    from gnn_rul_benchmarking_amd.dataloader import DeviceBatchLoader
    n = 103
    X = torch.arange(n, dtype=torch.float32).view(n, 1, 1).repeat(1, 2, 3)
    y = torch.arange(n, dtype=torch.float32).view(n, 1)
    seen = {}
    for rank in range(2):
        torch.manual_seed(5)                                    # every rank walks the same shuffled batches
        dl = DeviceBatchLoader(X, y, 25, True, False, "cpu", rank, 2)
        seen[rank] = [(yb.view(-1).tolist(), gb, lo) for _, yb, gb, lo in dl]
    assert len(seen[0]) == len(seen[1]) == 5
    allv = []
    for (a, gb, lo0), (b, gb1, lo1) in zip(seen[0], seen[1]):
        assert gb == gb1 == len(a) + len(b) and lo0 == 0 and lo1 == len(a)
        allv += a + b
    assert sorted(allv) == list(range(n))
    torch.manual_seed(5)
    single = [yb.view(-1).tolist() for _, yb, _, _ in DeviceBatchLoader(X, y, 25, True, False, "cpu")]
    assert sum(single, []) == allv                               # same global batches as a single process


# ---- STMSGCN: no BatchNorm / dropout, so the sharded step must reproduce the single-process step exactly ----
from oracle import stmsgcn_oracle as MO   # noqa: E402

MCFG = MO.Config(4, 20, 2, 3, [6, 9, 2], 4)


class StmsgcnOracleModel:
    """Duck-types the slice of STMSGCN_model that dp.DataParallel touches (no _after_train_forward: not batch-coupled)."""

    def __init__(self, prm):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names = MO.param_names(MCFG)
        self.num_live = sum(self.prm[k].size for k in self.names)
        self.bucket = torch.zeros(self.num_live + 1, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in self.names]).astype(np.float32))

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None):
        x, yy = X.numpy().astype(np.float64).reshape(X.size(0), -1), y.numpy().astype(np.float64).reshape(-1)
        loss, grads, _ = MO.loss_and_grads(self.prm, x, yy, MCFG)
        scale = x.shape[0] / float(global_batch)               # the kernels divide by the global batch
        self.bucket[:self.num_live] = torch.from_numpy(
            (scale * np.concatenate([grads[k].reshape(-1) for k in self.names])).astype(np.float32))
        self.bucket[self.num_live] = scale * loss
        return None, self.bucket[self.num_live]


def _stmsgcn_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(1)
        x = torch.from_numpy(rng.uniform(0, 0.5, (B, 1, MCFG.num_patch * MCFG.patch_size)).astype(np.float32))
        y = torch.from_numpy(rng.uniform(0, 1, (B, 1)).astype(np.float32))
        model = StmsgcnOracleModel(MO.random_params(MCFG, seed=5))
        dp = DataParallel()
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "flat": model.flat_params.clone().numpy()}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [7, 1])       # B = 1: the ragged last batch is smaller than the world, rank 1's shard is EMPTY
def test_stmsgcn_data_parallel_step_equals_single_process_world2_gloo(B):
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_stmsgcn_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert np.array_equal(out[0]["bucket"], out[1]["bucket"]) and np.array_equal(out[0]["flat"], out[1]["flat"])
    rng = np.random.default_rng(1)
    x = rng.uniform(0, 0.5, (B, 1, MCFG.num_patch * MCFG.patch_size)).astype(np.float32).astype(np.float64).reshape(B, -1)
    y = rng.uniform(0, 1, (B, 1)).astype(np.float32).astype(np.float64).reshape(-1)
    prm = MO.random_params(MCFG, seed=5)
    loss, grads, _ = MO.loss_and_grads(prm, x, y, MCFG)
    full = np.concatenate([grads[k].reshape(-1) for k in MO.param_names(MCFG)])
    assert abs(out[0]["loss"] - loss) < 1e-6 * abs(loss)
    assert np.allclose(out[0]["bucket"][:-1], full, rtol=1e-5, atol=1e-9)


# ---- ASTGCNN: BatchNorm-coupled like ST_GCN (local normalisation, global moments for the running statistics) ----
from oracle import astgcnn_oracle as AO   # noqa: E402

AN, AT, AOUT = 5, 12, 8


class AstgcnnOracleModel:
    """Duck-types the slice of ASTGCNN_model that dp.DataParallel touches."""

    def __init__(self, prm):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names = AO.live_param_names()
        self.num_live = sum(self.prm[k].size for k in self.names)
        self.bucket = torch.zeros(self.num_live + 1 + 4 * AN, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.global_moments = None

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None, sample_offset=0, update_running_stats=True,
                       moments_to_bucket=False):
        assert moments_to_bucket and not update_running_stats
        x, yy = X.numpy().astype(np.float64), y.numpy().astype(np.float64).reshape(-1)
        loss, grads, fw = AO.loss_and_grads(self.prm, x, yy, global_batch=global_batch)
        self.bucket[:self.num_live] = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.bucket[self.num_live] = loss
        w = x.shape[0] / float(global_batch)
        tail = self.bucket[self.num_live + 1:]
        for i, z in enumerate((fw.z1, fw.z2)):
            tail[(2 * i) * AN:(2 * i + 1) * AN] = torch.from_numpy((w * z.mean(axis=(0, 2))).astype(np.float32))
            tail[(2 * i + 1) * AN:(2 * i + 2) * AN] = torch.from_numpy((w * (z * z).mean(axis=(0, 2))).astype(np.float32))
        self.last = fw
        return None, self.bucket[self.num_live]

    def _after_train_forward(self, batch, from_bucket_moments=False):
        assert from_bucket_moments
        self.global_moments = self.bucket[self.num_live + 1:].clone()


def _astgcnn_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(2)
        x = torch.from_numpy(rng.uniform(0, 1, (B, AN, AT)).astype(np.float32))
        y = torch.from_numpy(rng.uniform(0, 1, (B, 1)).astype(np.float32))
        model = AstgcnnOracleModel(AO.random_params(AN, AT, output_dim=AOUT, seed=9))
        dp = DataParallel()
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "flat": model.flat_params.clone().numpy(),
                     "moments": model.global_moments.numpy(), "pred": model.last.pred.copy()}
    finally:
        dist.destroy_process_group()


def test_astgcnn_data_parallel_step_world2_gloo():
    B, world = 11, 2
    out = mp.Manager().dict()
    mp.spawn(_astgcnn_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert np.array_equal(out[0]["bucket"], out[1]["bucket"]) and np.array_equal(out[0]["flat"], out[1]["flat"])
    rng = np.random.default_rng(2)
    x = rng.uniform(0, 1, (B, AN, AT)).astype(np.float32).astype(np.float64)
    y = rng.uniform(0, 1, (B, 1)).astype(np.float32).astype(np.float64)
    # the reduced loss is the global-batch MSE of the (locally normalised) shard predictions
    pred = np.concatenate([out[0]["pred"], out[1]["pred"]])
    assert abs(out[0]["loss"] - float(np.mean((pred - y) ** 2))) < 1e-5
    # BatchNorm 1 sees the same conv output in every sharding: its reduced moments equal the single-process statistics
    fw = AO.forward(AO.random_params(AN, AT, output_dim=AOUT, seed=9), x, train=True)
    assert np.allclose(out[0]["moments"][:AN], fw.z1.mean(axis=(0, 2)), rtol=1e-5, atol=1e-6)
    assert np.allclose(out[0]["moments"][AN:2 * AN], (fw.z1 ** 2).mean(axis=(0, 2)), rtol=1e-5, atol=1e-6)


# ---- FC_STGNN: seven BatchNorms + dropout stream offset, same bucket scheme ----
from oracle import fcstgnn_oracle as FO   # noqa: E402

FCFG = FO.Config(patch_size=2, num_patch=5, encoder_time_out=4, encoder_hidden_dim=3, encoder_out_dim=2, encoder_conv_kernel=2,
                 hidden_dim=2, num_sequential=1, num_node=3, num_windows=6)


class FcstgnnOracleModel:
    """Duck-types the slice of FC_STGNN_RUL that dp.DataParallel touches."""

    def __init__(self, prm):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names = FO.param_names(FCFG)
        self.num_live = sum(self.prm[k].size for k in self.names)
        self.nbn = 2 * sum(FO.bn_channels(FCFG).values())
        self.bucket = torch.zeros(self.num_live + 1 + self.nbn, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.global_moments = None

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None, sample_offset=0, update_running_stats=True,
                       moments_to_bucket=False):
        assert moments_to_bucket and not update_running_stats
        x, yy = X.numpy().astype(np.float64), y.numpy().astype(np.float64).reshape(-1)
        loss, grads, fw = FO.loss_and_grads(self.prm, x, yy, FCFG, global_batch=global_batch)
        self.bucket[:self.num_live] = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.bucket[self.num_live] = loss
        w = x.shape[0] / float(global_batch)
        tail, o = self.bucket[self.num_live + 1:], 0
        for name, t in FO.bn_tapes(fw).items():
            c = t.mean.size
            tail[o:o + c] = torch.from_numpy((w * t.mean).astype(np.float32))
            tail[o + c:o + 2 * c] = torch.from_numpy((w * (t.var + t.mean ** 2)).astype(np.float32))
            o += 2 * c
        self.sample_offset = sample_offset
        return None, self.bucket[self.num_live]

    def _after_train_forward(self, batch, from_bucket_moments=False):
        assert from_bucket_moments
        self.global_moments = self.bucket[self.num_live + 1:].clone()


def _fcstgnn_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(4)
        x = torch.from_numpy(rng.uniform(0, 1, (B, FCFG.num_node, FCFG.num_patch * FCFG.patch_size)).astype(np.float32))
        y = torch.from_numpy(rng.uniform(0, 1, (B, 1)).astype(np.float32))
        model = FcstgnnOracleModel(FO.random_params(FCFG, seed=6))
        dp = DataParallel()
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "flat": model.flat_params.clone().numpy(),
                     "moments": model.global_moments.numpy(), "offset": model.sample_offset, "lo": lo}
    finally:
        dist.destroy_process_group()


def test_fcstgnn_data_parallel_step_world2_gloo():
    B, world = 9, 2
    out = mp.Manager().dict()
    mp.spawn(_fcstgnn_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert np.array_equal(out[0]["bucket"], out[1]["bucket"]) and np.array_equal(out[0]["flat"], out[1]["flat"])
    assert out[1]["offset"] == out[1]["lo"] > 0                         # the dropout stream of shard 1 starts at its first sample
    # the first BatchNorm sees the same conv output in every sharding: reduced moments == single-process batch statistics
    rng = np.random.default_rng(4)
    x = rng.uniform(0, 1, (B, FCFG.num_node, FCFG.num_patch * FCFG.patch_size)).astype(np.float32).astype(np.float64)
    fw = FO.forward(FO.random_params(FCFG, seed=6), x, FCFG, train=True)
    c = FCFG.encoder_hidden_dim
    assert np.allclose(out[0]["moments"][:c], fw.bna.mean, rtol=1e-5, atol=1e-6)
    assert np.allclose(out[0]["moments"][c:2 * c], fw.bna.var + fw.bna.mean ** 2, rtol=1e-5, atol=1e-6)


# ---- ST_Conv: three BatchNorms, same bucket scheme as ASTGCNN ----
from oracle import stconv_oracle as CO   # noqa: E402

CN, CT = 5, 9


class StconvOracleModel:
    def __init__(self, prm):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names = CO.live_param_names()
        self.num_live = sum(self.prm[k].size for k in self.names)
        self.bucket = torch.zeros(self.num_live + 1 + 6 * CN, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in self.names]).astype(np.float32))

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None, sample_offset=0, update_running_stats=True,
                       moments_to_bucket=False):
        assert moments_to_bucket and not update_running_stats
        x, yy = X.numpy().astype(np.float64), y.numpy().astype(np.float64).reshape(-1)
        loss, grads, fw = CO.loss_and_grads(self.prm, x, yy, global_batch=global_batch)
        self.bucket[:self.num_live] = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.bucket[self.num_live] = loss
        w = x.shape[0] / float(global_batch)
        tail = self.bucket[self.num_live + 1:]
        for i, t in enumerate((fw.bn1, fw.bn2, fw.bnc)):
            tail[(2 * i) * CN:(2 * i + 1) * CN] = torch.from_numpy((w * t.mean).astype(np.float32))
            tail[(2 * i + 1) * CN:(2 * i + 2) * CN] = torch.from_numpy((w * (t.var + t.mean ** 2)).astype(np.float32))
        return None, self.bucket[self.num_live]

    def _after_train_forward(self, batch, from_bucket_moments=False):
        assert from_bucket_moments
        self.global_moments = self.bucket[self.num_live + 1:].clone()


def _stconv_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(3)
        x = torch.from_numpy(rng.uniform(0, 1, (B, CN, CT)).astype(np.float32))
        y = torch.from_numpy(rng.uniform(0, 1, (B, 1)).astype(np.float32))
        model = StconvOracleModel(CO.random_params(CN, CT, seed=8))
        dp = DataParallel()
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "flat": model.flat_params.clone().numpy(),
                     "moments": model.global_moments.numpy()}
    finally:
        dist.destroy_process_group()


def test_stconv_data_parallel_step_world2_gloo():
    B, world = 9, 2
    out = mp.Manager().dict()
    mp.spawn(_stconv_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert np.array_equal(out[0]["bucket"], out[1]["bucket"]) and np.array_equal(out[0]["flat"], out[1]["flat"])
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, (B, CN, CT)).astype(np.float32).astype(np.float64)
    fw = CO.forward(CO.random_params(CN, CT, seed=8), x, train=True)
    # the first TCN BatchNorm and the CNN BatchNorm see sharding-independent inputs: reduced moments == single-process statistics
    assert np.allclose(out[0]["moments"][:CN], fw.bn1.mean, rtol=1e-5, atol=1e-6)
    assert np.allclose(out[0]["moments"][4 * CN:5 * CN], fw.bnc.mean, rtol=1e-5, atol=1e-6)


# ---- STGNN: no BatchNorm, no dropout -- the plain [gradient | loss] bucket, through the Algorithm class ----
from oracle import stgnn_oracle as GO   # noqa: E402

SG_CFG = dict(patch_size=6, num_patch=3, num_nodes=5, hidden_dim=8, K=3, top_k=3)


class StgnnOracleModel:
    """Duck-types the slice of STGNN_model that STGNN.update / dp.DataParallel touch (oracle-backed: the HIP path needs a GPU)."""

    def __init__(self, prm):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names = GO.param_names()
        self.num_live = sum(self.prm[k].size for k in self.names)
        self.bucket = torch.zeros(self.num_live + 1, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in self.names]).astype(np.float32))

    def _sync(self):
        o = 0
        for k in self.names:
            n = self.prm[k].size
            self.prm[k] = self.flat_params[o:o + n].numpy().astype(np.float64).reshape(self.prm[k].shape)
            o += n

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None):
        self._sync()
        x, yy = X.numpy().astype(np.float64), y.numpy().astype(np.float64)
        loss, grads, _, _ = GO.forward_backward(x, yy, self.prm, SG_CFG["num_patch"], SG_CFG["patch_size"], SG_CFG["top_k"])
        scale = x.shape[0] / float(global_batch if global_batch is not None else x.shape[0])    # the kernels divide by the global batch
        self.bucket[:self.num_live] = torch.from_numpy((scale * np.concatenate([grads[k].reshape(-1) for k in self.names])).astype(np.float32))
        self.bucket[self.num_live] = scale * loss
        return None, self.bucket[self.num_live]


def _stgnn_algo(perturb=0.0):
    """The package's STGNN Algorithm with its model / optimizer swapped for the CPU doubles."""
    from gnn_rul_benchmarking_amd.algorithms import STGNN
    torch.manual_seed(4)
    algo = STGNN(SG_CFG, {"learning_rate": 1e-2, "weight_decay": 1e-4}, "cpu")
    prm = GO.random_params(SG_CFG["num_patch"], SG_CFG["patch_size"], SG_CFG["num_nodes"], SG_CFG["hidden_dim"], SG_CFG["K"], seed=8)
    double = StgnnOracleModel(prm)
    double.flat_params += perturb
    del algo.model                                        # registered sub-module: replace it with the plain-object double
    object.__setattr__(algo, "model", double)
    algo.optimizer = SgdFromBucket(double)
    return algo


def _stgnn_batch(B):
    g = torch.Generator().manual_seed(21)
    return torch.rand(B, SG_CFG["num_nodes"], SG_CFG["num_patch"] * SG_CFG["patch_size"], generator=g), torch.rand(B, 1, generator=g)


def _stgnn_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        algo = _stgnn_algo(perturb=0.5 * rank)            # replicas start different: attach must broadcast rank 0's weights
        algo.attach_data_parallel(DataParallel())
        x, y = _stgnn_batch(B)
        lo, hi = shard_bounds(B, world, rank)
        losses = [float(algo.update(x[lo:hi], y[lo:hi], 1, global_batch=B, sample_offset=lo)["loss"]) for _ in range(3)]
        out[rank] = {"losses": losses, "flat": algo.model.flat_params.clone().numpy()}
    finally:
        dist.destroy_process_group()


def test_stgnn_data_parallel_steps_equal_single_process_world2_gloo():
    B, world = 9, 2
    out = mp.Manager().dict()
    mp.spawn(_stgnn_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert np.array_equal(out[0]["flat"], out[1]["flat"])                  # replicas stay bit-identical
    algo = _stgnn_algo()
    x, y = _stgnn_batch(B)
    single = []
    for _ in range(3):                                                     # the single-process reference: same double, full batch
        _, loss = algo.model.fused_mse_step(x, y, global_batch=B)
        single.append(float(loss))
        algo.optimizer.step(from_bucket=True)
    assert np.allclose(out[0]["losses"], single, rtol=1e-5) and np.allclose(out[1]["losses"], single, rtol=1e-5)
    assert np.allclose(out[0]["flat"], algo.model.flat_params.numpy(), rtol=1e-4, atol=1e-6)


# ---- synchronised BatchNorm: the data-parallel step IS the single-process full-batch step ------------------------------------
class SyncOracleModel(OracleModel):
    """Test double of ST_GCN_model.fused_mse_step_syncbn: the oracle with its BatchNorm reductions routed through the caller's
    all-reduce, in the order the phase kernels issue them (2 L forward pairs, then 2 L backward pairs)."""
    SYNC_BN_PAIRS_PER_LAYER = 4
    num_layers = L

    def sync_bn_schedule(self):
        return [20] * (self.SYNC_BN_PAIRS_PER_LAYER * self.num_layers)

    def __init__(self, prm, dropout=0.2, seed=7):
        super().__init__(prm, dropout, seed)
        self._bn_batch = torch.zeros(PL.bn_buffer_count(L), dtype=torch.float32)
        self.reductions = 0

    def fused_mse_step_syncbn(self, X, y, global_batch, sample_offset, bn_param_grad_scale, allreduce):
        self._step += 1
        keys = [O.dropout_layer_key(self.seed, self._step, l) for l in range(L)]

        def reduce(v):
            t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64))
            assert t.numel() == 20
            allreduce(t)
            self.reductions += 1
            return t.numpy()
        cnt = float(global_batch * N)
        x = X.numpy().astype(np.float64)
        fc = O.forward(self.prm, x, N, P, L, train=True, dropout=self.dropout, dropout_keys=keys, sample_offset=sample_offset,
                       stat_reduce=reduce, stat_count=cnt)
        loss, dpred = O.mse_loss_and_grad(fc.pred, y.numpy().astype(np.float64), global_batch)
        g = O.backward(self.prm, fc, dpred, self.dropout, stat_reduce=reduce, stat_count=cnt)
        for name, (off, shape) in PL.live_param_layout(N, L).items():
            v = g[name].reshape(-1)
            if name.endswith(".2.weight") or name.endswith(".2.bias"):
                v = v * bn_param_grad_scale        # global sums on every rank: one rank contributes them (rulgnn.h)
            self.bucket[off:off + int(np.prod(shape))] = torch.from_numpy(v.astype(np.float32))
        self.bucket[self.num_live] = loss
        for l in range(L):
            for b in range(2):
                base = ((l * 2 + b) * 2) * 10
                self._bn_batch[base:base + 10] = torch.from_numpy(fc.layers[l].bn_mean[b].astype(np.float32))
                self._bn_batch[base + 10:base + 20] = torch.from_numpy(fc.layers[l].bn_var[b].astype(np.float32))
        self.last_fc = fc
        return None, self.bucket[self.num_live]

    def _after_train_forward(self, batch, from_bucket_moments=False, from_bucket_stats=False):
        assert from_bucket_stats and not from_bucket_moments
        self.global_stats = self.bucket[self.num_live + 1:].clone()
        self._nbt += 1


def _sync_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        x = torch.from_numpy(rng.uniform(0, 1, (B, N, P)).astype(np.float32))
        y = torch.from_numpy(rng.uniform(0, 1, (B, 1)).astype(np.float32))
        model = SyncOracleModel(O.random_params(N, L, seed=0))
        dp = DataParallel(sync_bn=True)
        dp.broadcast_model(model)
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "flat": model.flat_params.clone().numpy(),
                     "stats": model.global_stats.numpy(), "pred": getattr(model, "last_fc", None) and model.last_fc.pred.copy(),
                     "reductions": model.reductions, "step": model._step}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [64, 65, 1])
def test_sync_batchnorm_step_equals_the_single_process_full_batch_step_world2_gloo(B):
    """B = 1: the second rank's shard is empty and it still joins every collective."""
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_sync_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert np.array_equal(r0["bucket"], r1["bucket"]) and np.array_equal(r0["flat"], r1["flat"])
    assert r0["step"] == r1["step"] == 1
    assert r0["reductions"] == 4 * L and r1["reductions"] == (4 * L if B > 1 else 0)
    # the single-process step on the whole batch
    prm = O.random_params(N, L, seed=0)
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 1, (B, N, P)).astype(np.float32).astype(np.float64)
    y = rng.uniform(0, 1, (B, 1)).astype(np.float32).astype(np.float64)
    keys = [O.dropout_layer_key(7, 1, l) for l in range(L)]
    fc = O.forward(prm, x, N, P, L, train=True, dropout=0.2, dropout_keys=keys)
    loss, dpred = O.mse_loss_and_grad(fc.pred, y)
    g = O.backward(prm, fc, dpred, 0.2)
    pred = np.concatenate([r["pred"] for r in (r0, r1) if r["pred"] is not None])
    assert np.allclose(pred, fc.pred, rtol=1e-9, atol=1e-12)            # same function, not just the same loss
    assert abs(r0["loss"] - loss) < 1e-6 * max(abs(loss), 1e-3)
    for name, (off, shape) in PL.live_param_layout(N, L).items():
        ref = g[name].reshape(-1)
        got = r0["bucket"][off:off + ref.size]
        assert np.allclose(got, ref, rtol=2e-5, atol=2e-6 * max(np.abs(ref).max(), 1e-6)), name
    for l in range(L):
        for b in range(2):
            base = ((l * 2 + b) * 2) * 10
            assert np.allclose(r0["stats"][base:base + 10], fc.layers[l].bn_mean[b], rtol=1e-5, atol=1e-7)
            assert np.allclose(r0["stats"][base + 10:base + 20], fc.layers[l].bn_var[b], rtol=1e-5, atol=1e-7)


# ---- RGCNU: dropout without BatchNorm -- the plain [gradient | loss] bucket, masks indexed by the global sample -----------------
from oracle import rgcnu_oracle as RO   # noqa: E402

RG_CFG = dict(num_nodes=5, time_length=8, hidden_dim=6, encoder_hidden_dim=4, kernel_size=3, alpha=0.8)


class RgcnuOracleModel:
    """Duck-types the slice of RGCNU_model that RGCNU.update / dp.DataParallel touch (oracle-backed: the HIP path needs a GPU)."""
    dropout_by_sample_offset = True
    training = True

    def __init__(self, prm, seed=5, p=0.5):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names = RO.param_names()
        self.num_live = sum(self.prm[k].size for k in self.names)
        self.bucket = torch.zeros(self.num_live + 1, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in self.names]).astype(np.float32))
        self._seed, self._step, self.p = seed, 0, p
        self.offsets = []

    def _sync(self):
        o = 0
        for k in self.names:
            n = self.prm[k].size
            self.prm[k] = self.flat_params[o:o + n].numpy().astype(np.float64).reshape(self.prm[k].shape)
            o += n

    def keep(self, bs, offset):
        N, L, H = RG_CFG["num_nodes"], RG_CFG["time_length"], RG_CFG["hidden_dim"]
        n = bs * L * N * H
        ctr = (np.arange(n, dtype=np.uint64) + np.uint64(offset * L * N * H)).astype(np.uint32)
        with np.errstate(over="ignore"):
            k = O._lowbias32(ctr ^ np.uint32(O.dropout_layer_key(self._seed, self._step, 0))) >= np.uint32(O.dropout_threshold(self.p))
        return k.reshape(bs, L, N, H) / (1.0 - self.p)

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None, sample_offset=0):
        self._sync()
        self._step += 1
        self.offsets.append(int(sample_offset))
        x, yy = X.numpy().astype(np.float64), y.numpy().astype(np.float64)
        loss, grads, fw = RO.loss_and_grads(self.prm, x, yy, RG_CFG["alpha"], self.keep(x.shape[0], sample_offset), global_batch)
        self.bucket[:self.num_live] = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.bucket[self.num_live] = loss
        self.last_pred = fw.pred.copy()
        return None, self.bucket[self.num_live]


def _rgcnu_algo(perturb=0.0):
    from gnn_rul_benchmarking_amd.algorithms import RGCNU
    torch.manual_seed(4)
    algo = RGCNU(RG_CFG, {"learning_rate": 1e-2, "weight_decay": 1e-4, "lambda": 0.1}, "cpu")
    c = RG_CFG
    double = RgcnuOracleModel(RO.random_params(c["num_nodes"], c["time_length"], c["hidden_dim"], c["encoder_hidden_dim"], c["kernel_size"], seed=8))
    double.flat_params += perturb
    del algo.model
    object.__setattr__(algo, "model", double)
    algo.optimizer = SgdFromBucket(double)
    return algo


def _rgcnu_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        algo = _rgcnu_algo(perturb=0.5 * rank)
        algo.attach_data_parallel(DataParallel())
        g = torch.Generator().manual_seed(31)
        x, y = torch.rand(B, RG_CFG["num_nodes"], RG_CFG["time_length"], generator=g), torch.rand(B, 1, generator=g)
        lo, hi = shard_bounds(B, world, rank)
        loss = float(algo.update(x[lo:hi], y[lo:hi], 1, global_batch=B, sample_offset=lo)["loss"])
        out[rank] = {"loss": loss, "flat": algo.model.flat_params.clone().numpy(), "offsets": algo.model.offsets,
                     "pred": algo.model.last_pred, "keep": algo.model.keep(hi - lo, lo), "lo": lo, "hi": hi}
    finally:
        dist.destroy_process_group()


def test_rgcnu_data_parallel_step_world2_gloo():
    """Replicas stay identical, the reduced loss is the global-batch MSE of the shard predictions, and the dropout masks are those
    of the global sample indices (the shards' masks tile the full batch's mask)."""
    B, world = 9, 2
    out = mp.Manager().dict()
    mp.spawn(_rgcnu_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert np.array_equal(r0["flat"], r1["flat"])
    assert r0["offsets"] == [0] and r1["offsets"] == [r1["lo"]]
    g = torch.Generator().manual_seed(31)
    x, y = torch.rand(B, RG_CFG["num_nodes"], RG_CFG["time_length"], generator=g), torch.rand(B, 1, generator=g)
    pred = np.concatenate([r0["pred"], r1["pred"]])
    assert abs(r0["loss"] - float(np.mean((pred - y.numpy()) ** 2))) < 1e-6
    full = _rgcnu_algo().model
    full._step = 1
    assert np.array_equal(np.concatenate([r0["keep"], r1["keep"]]), full.keep(B, 0))


# ---- SAGCN and STAGNN: the plain [gradient | loss] bucket -----------------------------------------------------------------------
from oracle import sagcn_oracle as SGO    # noqa: E402
from oracle import stagnn_oracle as TGO   # noqa: E402

SAGCN_CFG = dict(num_patch=4, patch_size=10, gcn_hidden_dim=6, attention_hidden_dim=5)
STAGNN_CFG = dict(num_nodes=4, time_length=9, hidden_dim=6, output_dim=3, num_heads=2, threshold=0)


class BucketOracleModel:
    """Duck-types the slice of SAGCN_model / STAGNN_model that their Algorithm.update and dp.DataParallel touch (oracle-backed: the HIP
    path needs a GPU).  `step_fn(prm, x, y, global_batch) -> (loss, grads, fw)`."""
    training = True

    def __init__(self, prm, names, step_fn):
        self.prm = {k: np.asarray(v, np.float64) for k, v in prm.items()}
        self.names, self.step_fn = names, step_fn
        self.num_live = sum(self.prm[k].size for k in names)
        self.bucket = torch.zeros(self.num_live + 1, dtype=torch.float32)
        self.flat_params = torch.from_numpy(np.concatenate([self.prm[k].reshape(-1) for k in names]).astype(np.float32))

    def fused_mse_step(self, X, y, optimizer=None, global_batch=None):
        o = 0
        for k in self.names:
            n = self.prm[k].size
            self.prm[k] = self.flat_params[o:o + n].numpy().astype(np.float64).reshape(self.prm[k].shape)
            o += n
        loss, grads, fw = self.step_fn(self.prm, X.numpy().astype(np.float64), y.numpy().astype(np.float64), global_batch)
        self.bucket[:self.num_live] = torch.from_numpy(np.concatenate([grads[k].reshape(-1) for k in self.names]).astype(np.float32))
        self.bucket[self.num_live] = loss
        self.last_pred = fw.pred.copy()
        return None, self.bucket[self.num_live]


def _bucket_algo(family, perturb=0.0):
    from gnn_rul_benchmarking_amd import algorithms as ALG
    torch.manual_seed(4)
    if family == "SAGCN":
        c = SAGCN_CFG
        algo = ALG.SAGCN(c, {"learning_rate": 1e-3, "weight_decay": 1e-4}, "cpu")
        double = BucketOracleModel(SGO.random_params(c["num_patch"], c["gcn_hidden_dim"], c["attention_hidden_dim"], seed=8), SGO.param_names(),
                                   lambda p, x, y, gb: SGO.loss_and_grads(p, x, y, c["num_patch"], c["patch_size"], gb))
    else:
        c = STAGNN_CFG
        algo = ALG.STAGNN(c, {"learning_rate": 1e-3, "weight_decay": 1e-4}, "cpu")
        double = BucketOracleModel(TGO.random_params(c["num_nodes"], c["time_length"], c["hidden_dim"], c["output_dim"], c["num_heads"], seed=8),
                                   TGO.live_param_names(c["num_heads"]), lambda p, x, y, gb: TGO.loss_and_grads(p, x, y, c["num_heads"], c["threshold"], gb))
    double.flat_params += perturb
    del algo.model
    object.__setattr__(algo, "model", double)
    algo.optimizer = SgdFromBucket(double)
    return algo


def _bucket_inputs(family, B):
    g = torch.Generator().manual_seed(31)
    if family == "SAGCN":
        return torch.rand(B, SAGCN_CFG["num_patch"] * SAGCN_CFG["patch_size"], generator=g) - 0.5, torch.rand(B, 1, generator=g)
    return torch.rand(B, STAGNN_CFG["num_nodes"], STAGNN_CFG["time_length"], generator=g), torch.rand(B, 1, generator=g)


def _bucket_worker(rank, world, port, family, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        algo = _bucket_algo(family, perturb=0.5 * rank)            # ranks start DIFFERENT on purpose
        algo.attach_data_parallel(DataParallel())
        x, y = _bucket_inputs(family, B)
        lo, hi = shard_bounds(B, world, rank)
        loss = float(algo.update(x[lo:hi], y[lo:hi], 1, global_batch=B, sample_offset=lo)["loss"])
        out[rank] = {"loss": loss, "flat": algo.model.flat_params.clone().numpy(), "pred": algo.model.last_pred}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("family", ["SAGCN", "STAGNN"])
def test_bucket_families_data_parallel_step_world2_gloo(family):
    """Replicas stay identical and the reduced loss is the global-batch MSE of the shard predictions.  SAGCN's samples are independent:
    the data-parallel step equals the single-process step; STAGNN normalises with rank-local BatchNorm statistics (stated in stagnn.py),
    so only the first two properties hold for it."""
    B, world = 7, 2
    out = mp.Manager().dict()
    mp.spawn(_bucket_worker, args=(world, _free_port(), family, B, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert np.array_equal(r0["flat"], r1["flat"])
    x, y = _bucket_inputs(family, B)
    pred = np.concatenate([r0["pred"], r1["pred"]])
    assert abs(r0["loss"] - float(np.mean((pred - y.numpy()) ** 2))) < 1e-6
    if family == "SAGCN":
        single = _bucket_algo(family)
        loss = float(single.update(x, y, 1)["loss"])
        single.optimizer.step(from_bucket=True)          # (the oracle-backed double leaves the optimizer to the caller)
        assert abs(loss - r0["loss"]) < 1e-6 and np.allclose(single.model.flat_params.numpy(), r0["flat"], rtol=1e-5, atol=1e-7)


# ---- large-bucket overlap: every rank issues the same collectives, also with an empty shard (ragged last batch smaller than the world) ----
class ReadyRegionsModel:
    """Duck-types a model whose backward reports final gradient regions (ST_GCN's tiled path): a bucket of 4096 'gradients' + loss +
    8 moments, two reported regions (in backward order) and what they leave."""
    reports_ready_gradients = True

    def __init__(self):
        self.num_live = 4096
        self.bucket = torch.zeros(self.num_live + 1 + 8, dtype=torch.float32)
        self.flat_params = torch.zeros(self.num_live)
        self._bn = torch.zeros(8)
        self._nbt = torch.zeros(2, dtype=torch.int64)
        self._step = 0

    def ready_regions(self):
        return [(3000, 1096), (1000, 500)]

    def fused_mse_step(self, X, y, global_batch=None, sample_offset=0, update_running_stats=True, moments_to_bucket=False, grad_ready=None):
        assert moments_to_bucket and not update_running_stats and grad_ready is not None
        self._step += 1
        b = X.shape[0]
        self.bucket[:self.num_live] = float(X.sum()) + torch.arange(self.num_live, dtype=torch.float32) * 1e-3
        for off, cnt in self.ready_regions():
            grad_ready(off, cnt)
        self.bucket[self.num_live] = float(b)
        self.bucket[self.num_live + 1:] = b / float(global_batch)
        return None, self.bucket[self.num_live]

    def _after_train_forward(self, batch, from_bucket_moments=False):
        assert from_bucket_moments
        self._nbt += 1


def _overlap_worker(rank, world, port, B, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = torch.arange(B * 3, dtype=torch.float32).reshape(B, 3) + 1.0
        y = torch.zeros(B, 1)
        model = ReadyRegionsModel()
        dp = DataParallel()
        dp.OVERLAP_MIN_BYTES = 1024
        lo, hi = shard_bounds(B, world, rank)
        loss = dp.step(model, SgdFromBucket(model), x[lo:hi], y[lo:hi], global_batch=B, sample_offset=lo)
        out[rank] = {"loss": float(loss), "bucket": model.bucket.clone().numpy(), "regions": list(dp.last_overlap_regions),
                     "step": model._step, "shard": hi - lo}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("B", [1, 5])
def test_overlapped_all_reduce_issues_the_same_collectives_on_every_rank_world2_gloo(B):
    """with B = 1 rank 1's shard is empty; it used to issue ONE all-reduce over the whole bucket against rank 0's
    per-region all-reduces.  Both ranks now walk the model's region schedule and its complement."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r1["shard"] == B // 2 and r0["step"] == r1["step"] == 1
    assert r0["regions"] == r1["regions"] == [(0, 1000), (1000, 1500), (1500, 3000), (3000, 4096), (4096, 4105)]
    assert np.array_equal(r0["bucket"], r1["bucket"])
    assert r0["loss"] == float(B) and np.allclose(r0["bucket"][4097:], 1.0)
    # every element was summed exactly once
    x = np.arange(B * 3, dtype=np.float32).reshape(B, 3) + 1.0
    n0 = (B + 1) // 2
    want = sum(float(part.sum()) + np.arange(4096, dtype=np.float32) * 1e-3 for part in (x[:n0], x[n0:]) if part.shape[0] > 0)
    assert np.allclose(r0["bucket"][:4096], want, rtol=1e-6)


# ---- sharded evaluation: metrics of a test set spread over the ranks -------------------------------------------
def _sharded_eval_worker(rank, world, port, n, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gnn_rul_benchmarking_amd.dataloader import DeviceBatchLoader
        from gnn_rul_benchmarking_amd.dp import DataParallel
        from gnn_rul_benchmarking_amd.trainer import gather_shards, sharded_metrics
        g = torch.Generator().manual_seed(5)
        X, y = torch.rand(n, 3, 4, generator=g), torch.rand(n, 1, generator=g)
        pred_full = (y.view(-1) + 0.05 * torch.randn(n, generator=g)).clamp_min(0.0)
        dl = DeviceBatchLoader(X, y, 4, False, False, "cpu", rank, world, shard_samples=True)
        lo, hi = dl.shard
        seen = [yb for _, yb, _, _ in dl]
        got_y = torch.cat(seen).view(-1) if seen else torch.empty(0)
        assert torch.equal(got_y, y.view(-1)[lo:hi]) and dl.global_n == n
        dp = DataParallel()
        m = sharded_metrics(pred_full[lo:hi].numpy(), y.view(-1)[lo:hi].numpy(), 125.0, dp)
        full = gather_shards(pred_full[lo:hi], dl, dp)
        out[rank] = {"metrics": m, "full_equal": bool(torch.equal(full, pred_full)), "shard": (lo, hi)}
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [37, 1, 2])
def test_sharded_test_set_metrics_equal_the_single_process_metrics_world2_gloo(n):
    """Contiguous shards (19 + 18; 1 + 0: an EMPTY shard joins with zeros), four sums + count in one all-reduce, the reference's closing
    divisions on every rank: equal to ``_calc_metrics`` of the whole set to 1e-12; the gathered prediction vector is the original."""
    from gnn_rul_benchmarking_amd.metrics import _calc_metrics
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sharded_eval_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    g = torch.Generator().manual_seed(5)
    X, y = torch.rand(n, 3, 4, generator=g), torch.rand(n, 1, generator=g)
    pred_full = (y.view(-1) + 0.05 * torch.randn(n, generator=g)).clamp_min(0.0)
    want = _calc_metrics(pred_full.numpy().astype(np.float64), y.view(-1).numpy().astype(np.float64), 125.0)
    assert out[0]["metrics"] == out[1]["metrics"]
    assert np.allclose(out[0]["metrics"], want, rtol=1e-12, atol=0)
    assert out[0]["full_equal"] and out[1]["full_equal"]
    assert out[0]["shard"][1] == out[1]["shard"][0] and out[1]["shard"][1] == n


@pytest.mark.parametrize("drop_last", [True, False])
def test_a_test_set_that_cannot_be_sharded_is_walked_whole_on_every_rank(drop_last, tmp_path):
    """a test loader that was asked to shard but cannot (``drop_last=True``), or a model whose eval forward
    depends on the batch composition (``eval_sample_independent = False``: RGCNU, HAGCN -> ``shard_test_sets=False``), must give EVERY
    rank the reference's batches of the whole set -- never a per-batch slice (rank 0 would report metrics of 1 / world of the set)."""
    from types import SimpleNamespace
    from gnn_rul_benchmarking_amd.dataloader import DeviceBatchLoader, data_generator
    from gnn_rul_benchmarking_amd.algorithms import HAGCN, RGCNU, ST_GCN
    g = torch.Generator().manual_seed(5)
    n, bs = 11, 4
    X, y = torch.rand(n, 3, 4, generator=g), torch.rand(n, 1, generator=g)
    one = [yb.clone() for _, yb, _, _ in DeviceBatchLoader(X, y, bs, False, drop_last, "cpu")]
    for rank in range(2):
        dl = DeviceBatchLoader(X, y, bs, False, drop_last, "cpu", rank, 2, shard_samples=True)
        got = [(yb, gb, lo) for _, yb, gb, lo in dl]
        if drop_last:                     # not shardable: the whole set, in the single-process batches
            assert not dl.shard_samples and dl.shard == (0, n)
            assert len(got) == len(one) and all(torch.equal(a[0], b) and a[1] == bs and a[2] == 0 for a, b in zip(got, one))
        else:                             # sharded: this rank's contiguous shard only
            assert dl.shard_samples and sum(a[0].shape[0] for a in got) == dl.shard[1] - dl.shard[0] in (5, 6)
    # the model-level opt-out, through data_generator
    assert getattr(ST_GCN.model_class, "eval_sample_independent", True)
    assert RGCNU.model_class.eval_sample_independent is False and HAGCN.model_class.eval_sample_independent is False
    torch.save({"samples": X.numpy(), "labels": y.view(-1).numpy(), "max_ruls": 125.0}, tmp_path / "train.pt")
    torch.save({"samples": X.numpy(), "labels": y.view(-1).numpy(), "max_ruls": 125.0}, tmp_path / "test.pt")
    cfg = SimpleNamespace(shuffle=False, drop_last=False)
    for rank in range(2):
        _, whole, _ = data_generator(str(tmp_path), cfg, {"batch_size": bs}, "cpu", rank, 2, shard_test_sets=False)
        assert not whole.shard_samples and whole.n == n and [b[0].shape[0] for b in whole] == [4, 4, 3]
        _, sharded, _ = data_generator(str(tmp_path), cfg, {"batch_size": bs}, "cpu", rank, 2)
        assert sharded.shard_samples and sharded.n in (5, 6)


def test_data_parallel_options_are_validated_and_the_peer_collective_is_opt_in(tmp_path):
    """``DataParallel(bn_collective=...)``: only "group" (default: every collective through torch.distributed) and "peer" (the device-side
    one-shot all-reduce of csrc/peer_comm.hip, built only together with ``sync_bn=True``); anything else raises before any collective;
    the C entry that the "peer" form hands to the library has the callback's signature and rejects bad arguments without a GPU."""
    import ctypes as C
    from gnn_rul_benchmarking_amd import _lib
    from gnn_rul_benchmarking_amd.dp import DataParallel
    store = str(tmp_path / "store")
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=0, world_size=1)
    try:
        with pytest.raises(ValueError):
            DataParallel(bn_collective="ring")
        assert DataParallel().peer is None and DataParallel(sync_bn=True).peer is None          # "group" never touches the mailboxes
        assert DataParallel(sync_bn=False, bn_collective="peer").peer is None                   # ... nor does "peer" without sync_bn
    finally:
        dist.destroy_process_group()
    lib = _lib.load()
    assert lib.rulgnn_peer_handle_bytes() == 64 and lib.rulgnn_peer_mailbox_bytes() > 2 * 8 * 128 * 8
    assert lib.rulgnn_peer_comm_create(0, 9, (C.c_void_p * 9)()) is None                           # world beyond 8
    assert lib.rulgnn_peer_comm_create(0, 2, (C.c_void_p * 2)()) is None                           # null mailboxes
    assert lib.rulgnn_peer_allreduce_f64(None, None, 4, None) == _lib.EINVAL
    assert lib.rulgnn_peer_comm_collectives(None) == -1
