"""-m gpu: synchronised BatchNorm (rulgnn_stgcn_train_fwdbwd_syncbn_f32).  Two data-parallel ranks are
emulated on ONE GPU: two replicas, two host threads, two HIP streams; the all-reduce callback the library issues between its
phase kernels is a two-party rendezvous that sums the ranks' 20 reduction cells.  The sharded step must be the single-call
step on the whole batch: same predictions, loss, BatchNorm statistics and (summed over the ranks) the same gradient."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _replica(N, P, L, dropout, state=None, seed=5, k=1):
    from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model
    torch.manual_seed(3)
    m = ST_GCN_model(N, P, num_layers=L, dropout=dropout, k=k)
    if state is not None:
        m.load_state_dict(state)
    m = m.to(DEV).train()
    m._seed = seed
    return m


class TwoPartySum:
    """In-place SUM over two 'ranks' that live in two threads of this process."""

    def __init__(self, numel=20):
        self.numel = numel
        self.barrier = threading.Barrier(2, timeout=60)
        self.slots = [None, None]
        self.calls = [0, 0]

    def __call__(self, rank, view):
        assert view.dtype == torch.float64 and view.numel() == self.numel and view.is_cuda
        torch.cuda.current_stream().synchronize()              # the producing phase kernel has finished
        self.slots[rank] = view.clone()
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()
        total = self.slots[0] + self.slots[1]
        view.copy_(total)
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()                                     # both have read both slots before either overwrites its own
        self.calls[rank] += 1


@pytest.mark.parametrize("N,P,L,B,split,p,K", [(14, 30, 2, 96, 48, 0.2, 1), (14, 30, 2, 4099, 1500, 0.2, 1), (14, 30, 1, 37, 36, 0.0, 1),
                                               (40, 64, 2, 21, 8, 0.2, 1), (14, 50, 3, 130, 64, 0.1, 1),
                                               (14, 30, 2, 96, 40, 0.2, 2), (40, 64, 2, 21, 8, 0.2, 3)])
def test_two_shards_with_synchronised_batchnorm_equal_the_full_batch_step(N, P, L, B, split, p, K):
    g = torch.Generator(device=DEV).manual_seed(B)
    x = torch.rand(B, N, P, device=DEV, generator=g)
    y = torch.rand(B, 1, device=DEV, generator=g)
    full = _replica(N, P, L, p, k=K)
    state = {k: v.clone() for k, v in full.state_dict().items()}
    pred_f, loss_f = full.fused_mse_step(x, y)
    pred_f, loss_f = pred_f.clone(), float(loss_f)
    grad_f, bn_f = full.bucket[:full.num_live].clone(), full._bn_batch.clone()

    ranks = [_replica(N, P, L, p, state, k=K), _replica(N, P, L, p, state, k=K)]
    bounds = [(0, split), (split, B)]
    comm = TwoPartySum()
    errors = []

    def run(rank):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
                lo, hi = bounds[rank]
                ranks[rank].fused_mse_step_syncbn(x[lo:hi], y[lo:hi], B, lo, 1.0 if rank == 0 else 0.0, lambda v: comm(rank, v))
                torch.cuda.current_stream().synchronize()
        except BaseException as e:                  # pragma: no cover
            errors.append(e)
            comm.barrier.abort()
    torch.cuda.synchronize()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errors, errors
    assert comm.calls == [4 * L, 4 * L]
    torch.cuda.synchronize()

    pred = torch.cat([ranks[0]._pred_buf[:split], ranks[1]._pred_buf[:B - split]])
    scale = float(pred_f.abs().max())
    assert float((pred - pred_f).abs().max()) < 2e-6 * scale
    nl = full.num_live
    loss = float(ranks[0].bucket[nl] + ranks[1].bucket[nl])
    assert abs(loss - loss_f) < 1e-5 * abs(loss_f)
    for r in ranks:                                  # every rank holds the statistics of the GLOBAL batch
        assert torch.allclose(r._bn_batch, bn_f, rtol=1e-5, atol=1e-7)
    grad = ranks[0].bucket[:nl] + ranks[1].bucket[:nl]
    from gnn_rul_benchmarking_amd import params as PL
    for name, (off, shape) in PL.live_param_layout(N, L, K).items():
        n = int(np.prod(shape))
        ref, got = grad_f[off:off + n], grad[off:off + n]
        assert float((got - ref).abs().max()) < 2e-5 * max(float(ref.abs().max()), 1e-6), name


def test_synchronised_batchnorm_differs_from_local_statistics_and_rejects_bad_arguments():
    """The local-BN shard step is a different function (that is the point of the option), and the ABI rejects what it documents."""
    import ctypes as C
    from gnn_rul_benchmarking_amd import _lib
    N, P, L, B = 14, 30, 2, 64
    g = torch.Generator(device=DEV).manual_seed(1)
    x, y = torch.rand(B, N, P, device=DEV, generator=g), torch.rand(B, 1, device=DEV, generator=g)
    full = _replica(N, P, L, 0.0)
    pred_f = full.fused_mse_step(x, y)[0].clone()
    local = _replica(N, P, L, 0.0, {k: v.clone() for k, v in full.state_dict().items()})
    pred_l = local.fused_mse_step(x[:32], y[:32], global_batch=B)[0][:32].clone()
    assert float((pred_l - pred_f[:32]).abs().max()) > 1e-4 * float(pred_f.abs().max())
    lib = _lib.load()
    shp = local._shape(32)
    a = local._train_args(shp, x[:32].reshape(32, -1).contiguous(), y[:32].reshape(-1).contiguous(), None, 1, B, 0, False)
    cb = _lib.ALLREDUCE_F64_FN(lambda u, b, c, s: 0)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.rulgnn_stgcn_train_fwdbwd_syncbn_f32(C.byref(shp), C.byref(a), 2.0, cb, None, st) == _lib.EINVAL
    a.bn_moment_weight = 0.5
    assert lib.rulgnn_stgcn_train_fwdbwd_syncbn_f32(C.byref(shp), C.byref(a), 1.0, cb, None, st) == _lib.EINVAL
    a.bn_moment_weight = 0.0
    fail = _lib.ALLREDUCE_F64_FN(lambda u, b, c, s: 1)
    assert lib.rulgnn_stgcn_train_fwdbwd_syncbn_f32(C.byref(shp), C.byref(a), 1.0, fail, None, st) == _lib.ECALLBACK
    torch.cuda.synchronize()


def _family_replica(name, state=None):
    """FC_STGNN (seven BatchNorm layers, positional-encoding dropout) / ASTGCNN (two) at their C-MAPSS wirings."""
    from gnn_rul_benchmarking_amd import hparams as HP
    torch.manual_seed(3)
    if name == "FC_STGNN":
        from gnn_rul_benchmarking_amd.fcstgnn import FC_STGNN_RUL as M
        cfg = HP.get_hparams_class("CMAPSS")("FD004").alg_hparams[name]
    else:
        from gnn_rul_benchmarking_amd.astgcnn import ASTGCNN_model as M
        cfg = HP.get_hparams_class("NCMAPSS")("DS02").alg_hparams[name]
    m = M(**cfg)
    if state is not None:
        m.load_state_dict(state)
    m = m.to(DEV).train()
    m._seed = 5
    return m


@pytest.mark.parametrize("name,shape,B,split", [("FC_STGNN", (14, 50), 24, 10), ("FC_STGNN", (14, 50), 9, 8), ("ASTGCNN", (20, 50), 96, 40),
                                                ("ASTGCNN", (20, 50), 33, 32)])
def test_family_two_shards_with_synchronised_batchnorm_equal_the_full_batch_step(name, shape, B, split):
    """rulgnn_fcstgnn_fwdbwd_syncbn_f32 / rulgnn_astgcnn_fwdbwd_syncbn_f32 (round 3): same contract as the ST_GCN entry."""
    g = torch.Generator(device=DEV).manual_seed(B)
    x = torch.rand(B, *shape, device=DEV, generator=g)
    y = torch.rand(B, 1, device=DEV, generator=g)
    full = _family_replica(name)
    state = {k: v.clone() for k, v in full.state_dict().items()}
    pred_f, loss_f = full.fused_mse_step(x, y, update_running_stats=False)
    pred_f, loss_f = pred_f.clone(), float(loss_f)
    nl = full.num_live
    grad_f, bn_f = full.bucket[:nl].clone(), full._bn_batch.clone()

    ranks = [_family_replica(name, state), _family_replica(name, state)]
    bounds = [(0, split), (split, B)]
    schedule = ranks[0].sync_bn_schedule()
    comm = TwoPartySum(schedule[0])
    errors = []

    def run(rank):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=DEV)):
                lo, hi = bounds[rank]
                ranks[rank].fused_mse_step_syncbn(x[lo:hi], y[lo:hi], B, lo, 1.0 if rank == 0 else 0.0, lambda v: comm(rank, v))
                torch.cuda.current_stream().synchronize()
        except BaseException as e:                  # pragma: no cover
            errors.append(e)
            comm.barrier.abort()
    torch.cuda.synchronize()
    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not errors, errors
    assert comm.calls == [len(schedule), len(schedule)]
    torch.cuda.synchronize()

    pred = torch.cat([ranks[0]._pred_buf[:split], ranks[1]._pred_buf[:B - split]])
    assert float((pred - pred_f[:B]).abs().max()) < 1e-5 * float(pred_f.abs().max())
    loss = float(ranks[0].bucket[nl] + ranks[1].bucket[nl])
    assert abs(loss - loss_f) < 2e-5 * abs(loss_f)
    for r in ranks:                                  # every rank holds the statistics of the GLOBAL batch
        assert torch.allclose(r._bn_batch, bn_f, rtol=2e-5, atol=1e-6)
    grad = ranks[0].bucket[:nl] + ranks[1].bucket[:nl]
    gmax = float(grad_f.abs().max())
    for pname, (off, shp) in full._layout.items():
        n = int(np.prod(shp))
        ref, got = grad_f[off:off + n], grad[off:off + n]
        # (a bias in front of a BatchNorm has a mathematically zero gradient: rounding residue, compared on the gradient's scale)
        assert float((got - ref).abs().max()) < 2e-4 * max(float(ref.abs().max()), 1e-3 * gmax), pname
    # and the local-statistics shard step is a different function
    local = _family_replica(name, state)
    pred_l = local.fused_mse_step(x[:split], y[:split], global_batch=B, update_running_stats=False)[0][:split]
    assert float((pred_l - pred_f[:split]).abs().max()) > 1e-5 * float(pred_f.abs().max())
