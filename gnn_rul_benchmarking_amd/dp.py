"""Data parallelism for the ST_GCN path: one process per GPU, gradients all-reduced over RCCL.

The reference is single-process (no torch.distributed anywhere); this is the
new part.  Training samples are independent except for BatchNorm's batch statistics, so the
global batch is sharded by sample with NO data-path collective; per step there is exactly ONE
all-reduce, over a single flat fp32 bucket:

    [ gradient of the live parameters (P floats) | loss (1) | BatchNorm batch moments (2L*2*10) ]

* gradient: every rank's kernels already scale d(loss)/d(pred) by 1/global_batch, so a SUM over
  ranks is the gradient of the global-batch MSE -- no division afterwards;
* loss: sum over ranks of sum_i (pred_i - y_i)^2 / global_batch;
* BatchNorm: normalisation uses each rank's LOCAL batch statistics (the torch DDP default); the
  running statistics are updated from the all-reduced global-batch moments, so replicas stay
  bit-identical without a separate buffer broadcast.
* ``DataParallel(sync_bn=True)`` (ST_GCN, num_patch <= 64): SYNCHRONISED BatchNorm -- SURVEY.md section 8e's "per-BN
  all-reduce of [sum x, sum x^2, count] in forward and the matching [sum dy, sum dy*xhat] in backward".  The step then
  computes exactly the single-GPU function of the concatenated batch (tests/test_dp_cpu.py, tests/test_syncbn_gpu.py),
  at the price of 4 L more latency-bound all-reduces of 20 doubles between the phase kernels (8 at L = 2).

For C-MAPSS shapes the bucket is 6.4 KB: the collective is latency-bound, so
it is issued once, on the compute stream, directly on the kernels' output buffer (no copy)."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n samples for `rank`; the first n % world_size ranks get one extra
    (ragged last batch, drop_last=False in the reference's loaders)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class PeerAllReduce:
    """The synchronised-BatchNorm collectives as one-shot all-reduces over peer-mapped mailboxes (csrc/peer_comm.hip,
    include/rulgnn.h "One-shot all-reduce ..."): one single-workgroup launch per collective on the compute stream, NO host callback
    between the phases of a step -- through ``torch.distributed`` each of the 4 L collectives of a step is a ctypes -> Python -> RCCL
    round trip (~50 us of a 350-us step).  Sums in rank order: bit-identical on every rank.

    Set-up exchanges the mailboxes' IPC handles through ``torch.distributed`` (any backend); the ranks must be on one node with peer
    access between their devices (or on one device).  An instance is passed to ``fused_mse_step_syncbn`` in place of the Python
    all-reduce (``c_callback`` / ``c_user``: _lib.allreduce_callback) and is itself callable on a float64 device tensor.

    ``timeout_s``: how long a collective spins for a peer before it gives up (NaN in the buffer -> a NaN loss on every rank through the
    bucket all-reduce, and ``check()`` raises on the rank that waited)."""

    def __init__(self, process_group=None, timeout_s: float = 20.0):
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        self._lib, self._C = lib, C
        self.rank, self.world_size = dist.get_rank(process_group), dist.get_world_size(process_group)
        hb = lib.rulgnn_peer_handle_bytes()
        handle = (C.c_ubyte * hb)()
        self._box = C.c_void_p()
        _lib.check(lib.rulgnn_peer_mailbox_alloc(C.byref(self._box), handle), "rulgnn_peer_mailbox_alloc")
        handles = [None] * self.world_size
        dist.all_gather_object(handles, bytes(handle), group=process_group)
        ptrs = (C.c_void_p * self.world_size)()
        self._opened = []
        for r, h in enumerate(handles):
            if r == self.rank:
                ptrs[r] = self._box.value
                continue
            peer = C.c_void_p()
            _lib.check(lib.rulgnn_peer_mailbox_open((C.c_ubyte * hb).from_buffer_copy(h), C.byref(peer)), "rulgnn_peer_mailbox_open")
            self._opened.append(peer)
            ptrs[r] = peer.value
        self.c_user = lib.rulgnn_peer_comm_create(self.rank, self.world_size, ptrs)
        if not self.c_user:
            raise RuntimeError("rulgnn_peer_comm_create failed")
        _lib.check(lib.rulgnn_peer_comm_set_timeout_ms(self.c_user, int(round(timeout_s * 1000.0))), "rulgnn_peer_comm_set_timeout_ms")
        self.c_callback = C.cast(lib.rulgnn_peer_allreduce_f64, C.c_void_p).value
        dist.barrier(group=process_group)              # every mailbox is mapped everywhere before the first push

    def __call__(self, cells: torch.Tensor) -> None:
        from . import _lib
        if cells.dtype != torch.float64 or not cells.is_cuda or not cells.is_contiguous():
            raise RuntimeError("PeerAllReduce sums contiguous float64 device tensors")
        st = self._C.c_void_p(torch.cuda.current_stream(cells.device).cuda_stream)
        _lib.check(self._lib.rulgnn_peer_allreduce_f64(self.c_user, cells.data_ptr(), cells.numel(), st), "rulgnn_peer_allreduce_f64")

    def collectives(self) -> int:
        return int(self._lib.rulgnn_peer_comm_collectives(self.c_user))

    def check(self) -> None:
        """Raise if a collective of this rank gave up waiting for a peer (synchronises with the device)."""
        e = int(self._lib.rulgnn_peer_comm_status(self.c_user))
        if e != 0:
            raise RuntimeError(f"one-shot all-reduce number {e} timed out waiting for a peer rank (the step's results are NaN)")

    def close(self) -> None:
        if getattr(self, "c_user", None):
            torch.cuda.synchronize()
            self._lib.rulgnn_peer_comm_destroy(self.c_user)
            self.c_user = None
            for p in self._opened:
                self._lib.rulgnn_peer_mailbox_close(p)
            self._lib.rulgnn_peer_mailbox_free(self._box)


class DataParallel:
    def __init__(self, process_group=None, sync_bn=False, bn_collective="group"):
        """``bn_collective`` (with ``sync_bn=True``): "group" = every BatchNorm reduction through ``torch.distributed.all_reduce`` on the
        process group (RCCL on GPUs: a host callback per collective), "peer" = the device-side one-shot all-reduce of ``PeerAllReduce``
        (one node, peer access; no host work between the phases of a step)."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if bn_collective not in ("group", "peer"):
            raise ValueError("bn_collective is 'group' or 'peer'")
        self.group = process_group
        self.sync_bn = bool(sync_bn)
        self.rank = dist.get_rank(process_group)
        self.world_size = dist.get_world_size(process_group)
        self.bn_collective = bn_collective
        self.peer = PeerAllReduce(process_group) if (self.sync_bn and bn_collective == "peer") else None

    def broadcast_model(self, model) -> None:
        """Make every replica identical to rank 0 (parameters and BatchNorm statistics)."""
        dist.broadcast(model.flat_params, 0, group=self.group)
        if getattr(model, "_bn", None) is not None:            # models without BatchNorm (STMSGCN, STGNN) have no such buffers
            dist.broadcast(model._bn, 0, group=self.group)
        if hasattr(model, "_flush_nbt"):
            model._flush_nbt()
        if getattr(model, "_nbt", None) is not None:
            dist.broadcast(model._nbt, 0, group=self.group)

    @staticmethod
    def _empty_shard_bookkeeping(model, global_batch) -> None:
        """What a rank with an EMPTY shard must still advance so that it stays in lockstep with the ranks that ran kernels: the dropout
        stream position, and the chain the step resolved to (a function of the shape and ``step_path`` alone): with it the guarded /
        unguarded choice of the optimizer kernels behind the all-reduce is the same on every rank."""
        if hasattr(model, "_step"):
            model._step += 1
        resolve = getattr(model, "resolve_chain_for_empty_shard", None)
        if resolve is not None:
            resolve(global_batch)

    @staticmethod
    def _note_guard(model, loss) -> None:
        if getattr(model, "guard_tensor", None) is not None:
            model.note_data_parallel_loss(loss)

    @staticmethod
    def _optimizer_step(model, optimizer) -> None:
        """The optimizer over the all-reduced bucket.  ST_GCN's matrix-core training chain reports an f16 range violation as a NaN
        loss (include/rulgnn.h, RULGNN_STEP_MX): summed over the ranks it makes every rank skip this step (and the running-statistics
        update) alike -- guarded kernels -- and ``ST_GCN.update`` then repeats the step on the fp32 chain."""
        guard = getattr(model, "guard_tensor", None)
        if guard is not None:
            optimizer.step(from_bucket=True, guard=guard)
        else:
            optimizer.step(from_bucket=True)

    def _optimizer_and_stats(self, model, optimizer, global_batch, **source) -> None:
        """Optimizer step + BatchNorm running statistics behind the all-reduce: one launch where the model offers it (ST_GCN)."""
        fused = getattr(model, "fused_optimizer_and_running_stats", None)
        if fused is not None and fused(optimizer, global_batch, **source):
            return
        self._optimizer_step(model, optimizer)
        model._after_train_forward(global_batch, **source)

    def all_reduce_bucket(self, bucket: torch.Tensor) -> None:
        dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)

    # ---- large buckets: all-reduce in gradient-ready order, overlapped with the rest of the backward -------------------------
    OVERLAP_MIN_BYTES = 1 << 20          # below this one blocking all-reduce is latency-bound either way

    @staticmethod
    def _complement(regions, numel):
        """What ``regions`` ((lo, hi) pairs, disjoint) leave of [0, numel), in ascending order."""
        pos, rest = 0, []
        for lo, hi in sorted(regions):
            if lo < pos:
                raise RuntimeError("overlapping gradient-ready regions")
            if lo > pos:
                rest.append((pos, lo))
            pos = hi
        if pos < numel:
            rest.append((pos, numel))
        return rest

    def _overlapped_step(self, model, X_shard, y_shard, global_batch, sample_offset):
        """``fused_mse_step`` of a model whose backward reports final gradient regions (ST_GCN's tiled path: theta / fc1 are
        num_patch x num_patch, a 12.6 MB bucket at XJTU-SY).  Each reported region is all-reduced on a side stream as soon as the
        kernels that finalise it have been enqueued -- behind an event recorded on the compute stream -- while the compute stream
        carries on with the layers below; whatever was not reported goes out when the step has been enqueued; the compute stream
        then waits for the side stream.  Every bucket element is summed over the ranks exactly once.

        Every rank must issue the SAME collectives in the same order: the regions are ``model.ready_regions()`` -- a function of the
        shape alone, checked against what the kernels report -- followed by their complement, and a rank whose shard is empty (ragged
        last batch smaller than the world, ``drop_last=False`` in the reference's loaders) replays exactly that sequence on a zero
        bucket instead of one all-reduce over the whole bucket (which would pair its single collective with the first slice of the
        others: a hang, or silently wrong sums)."""
        bucket = model.bucket
        on_gpu = bucket.is_cuda
        expected = [(int(o), int(o) + int(c)) for o, c in model.ready_regions()]
        if on_gpu:
            compute = torch.cuda.current_stream()
            if getattr(self, "_comm_stream", None) is None or self._comm_stream.device != bucket.device:
                self._comm_stream = torch.cuda.Stream(device=bucket.device)
            comm = self._comm_stream
        done = []

        def launch(offset, count):
            if on_gpu:
                ev = torch.cuda.Event()
                ev.record(compute)
                comm.wait_event(ev)
                with torch.cuda.stream(comm):
                    dist.all_reduce(bucket[offset:offset + count], op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(bucket[offset:offset + count], op=dist.ReduceOp.SUM, group=self.group)
            done.append((offset, offset + count))

        if X_shard.size(0) == 0:
            bucket.zero_()
            self._empty_shard_bookkeeping(model, global_batch)
            for lo, hi in expected:
                launch(lo, hi - lo)
        else:
            model.fused_mse_step(X_shard, y_shard, global_batch=global_batch, sample_offset=sample_offset, update_running_stats=False,
                                 moments_to_bucket=True, grad_ready=launch)
            if done != expected:
                raise RuntimeError(f"gradient-ready regions {done} differ from the model's schedule {expected}")
        for lo, hi in self._complement(done, bucket.numel()):
            launch(lo, hi - lo)
        if on_gpu:
            fin = torch.cuda.Event()
            fin.record(comm)
            compute.wait_event(fin)
        self.last_overlap_regions = sorted(done)      # (tests / diagnostics)

    def _sync_bn_step(self, model, optimizer, X_shard, y_shard, global_batch, sample_offset):
        """Synchronised-BatchNorm step: the model's phase chain calls back for every BatchNorm reduction pair; an empty shard
        joins the same collectives with zeros."""
        b = X_shard.size(0)
        if not hasattr(model, "fused_mse_step_syncbn"):
            raise RuntimeError(f"{type(model).__name__} has no synchronised-BatchNorm step (ST_GCN, FC_STGNN and ASTGCNN do); "
                               "use sync_bn=False")
        schedule = model.sync_bn_schedule()
        # rank 0 must hold data whenever the batch is not empty (shard_bounds()): it alone contributes the BatchNorm scale / shift
        # gradients.  A violation is raised AFTER this rank has joined every collective of the step with zeros -- raising here would
        # leave the other ranks waiting in theirs forever.
        violated = b == 0 and self.rank == 0 and global_batch > 0
        reduce_cells = self.peer if self.peer is not None else (lambda cells: dist.all_reduce(cells, op=dist.ReduceOp.SUM, group=self.group))
        if b == 0:
            for n in schedule:
                zero = torch.zeros(n, dtype=torch.float64, device=model.bucket.device)
                reduce_cells(zero)
            model.bucket.zero_()
            self._empty_shard_bookkeeping(model, global_batch)
        else:
            # the BatchNorm scale / shift gradients come out of the all-reduced cells, i.e. they are already the global sums on every
            # rank: rank 0 (never empty under shard_bounds) contributes them to the bucket, the others contribute zero
            model.fused_mse_step_syncbn(X_shard, y_shard, global_batch, sample_offset, 1.0 if self.rank == 0 else 0.0, reduce_cells)
            # every rank holds the same global (mean, variance); weighted by the shard fraction they SUM to themselves over the
            # ranks, which also hands them to a rank whose shard was empty -- one bucket all-reduce as in the local-BN step
            tail = model.bucket[model.num_live + 1:model.num_live + 1 + model._bn_batch.numel()]
            torch.mul(model._bn_batch.reshape(-1), float(b) / float(global_batch), out=tail)
        self.all_reduce_bucket(model.bucket)
        if violated:
            raise RuntimeError("synchronised BatchNorm expects shard_bounds() sharding: rank 0 holds data whenever the batch is not empty")
        self._optimizer_and_stats(model, optimizer, global_batch, from_bucket_stats=True)
        self._note_guard(model, model.bucket[model.num_live])
        return model.bucket[model.num_live]

    def step(self, model, optimizer, X_shard, y_shard, global_batch=None, sample_offset=None):
        """One data-parallel ``Algorithm.update`` on this rank's shard.  ``global_batch`` defaults to
        world_size * len(shard) (equal shards); pass it (and ``sample_offset``) for ragged batches."""
        b = X_shard.size(0)
        if global_batch is None:
            global_batch = b * self.world_size
        if sample_offset is None:
            sample_offset = b * self.rank
        batch_coupled = hasattr(model, "_after_train_forward")      # BatchNorm / dropout state (ST_GCN); STMSGCN has none
        if self.sync_bn and batch_coupled:
            return self._sync_bn_step(model, optimizer, X_shard, y_shard, global_batch, sample_offset)
        # decided from what every rank shares (the model and its bucket), never from this rank's shard: ranks that disagreed would issue
        # different collectives
        overlap = batch_coupled and getattr(model, "reports_ready_gradients", False) and \
            model.bucket.numel() * 4 >= self.OVERLAP_MIN_BYTES
        if overlap:
            self._overlapped_step(model, X_shard, y_shard, global_batch, sample_offset)      # an empty shard replays the same collectives
            self._optimizer_and_stats(model, optimizer, global_batch, from_bucket_moments=True)
            self._note_guard(model, model.bucket[model.num_live])
            return model.bucket[model.num_live]
        if b == 0:
            # Ragged last batch smaller than the world (drop_last=False: n % batch_size can be 1..world_size-1): this rank's
            # shard is empty.  It launches no kernel, contributes a zero bucket, and still takes part in the all-reduce, the
            # optimizer step and the running-statistics update, so that replicas stay identical and nobody waits forever.
            model.bucket.zero_()
            self._empty_shard_bookkeeping(model, global_batch)
        elif batch_coupled:
            model.fused_mse_step(X_shard, y_shard, global_batch=global_batch, sample_offset=sample_offset,
                                 update_running_stats=False, moments_to_bucket=True)
        elif getattr(model, "dropout_by_sample_offset", False):      # dropout without BatchNorm (RGCNU): masks indexed by global sample
            model.fused_mse_step(X_shard, y_shard, global_batch=global_batch, sample_offset=sample_offset)
        else:
            model.fused_mse_step(X_shard, y_shard, global_batch=global_batch)
        self.all_reduce_bucket(model.bucket)
        if batch_coupled:
            self._optimizer_and_stats(model, optimizer, global_batch, from_bucket_moments=True)
        else:
            self._optimizer_step(model, optimizer)
        self._note_guard(model, model.bucket[model.num_live])
        return model.bucket[model.num_live]
