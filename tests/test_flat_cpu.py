"""The base class of the model modules (gnn_rul_benchmarking_amd/flat.py): flat parameter views, bucket, workspace cache -- host logic,
no GPU."""
import pytest
import torch
import torch.nn as nn

from gnn_rul_benchmarking_amd import params as PL
from gnn_rul_benchmarking_amd.flat import FlatModule


class _Toy(FlatModule):
    bucket_tail = 3
    workspace_slots = 2

    def __init__(self, order=None):
        super().__init__()
        self.a = nn.Linear(3, 2)
        self.b = nn.Linear(2, 1)
        self.bn = nn.BatchNorm1d(2)
        if order is not None:
            self.flat_order = order
        self._bn = None
        self._track_batchnorm_counters()
        self._init_flat()

    def _reflatten_buffers(self, dev):
        bufs = dict(self.named_buffers())
        self._bn = torch.empty(4, dtype=torch.float32, device=dev)
        self._nbt = torch.zeros(1, dtype=torch.int64, device=dev)
        for i, leaf in enumerate(("running_mean", "running_var")):
            self._bn[2 * i:2 * i + 2].copy_(bufs[f"bn.{leaf}"])
            self._set_buffer(f"bn.{leaf}", self._bn[2 * i:2 * i + 2])
        self._nbt[0].copy_(bufs["bn.num_batches_tracked"])
        self._set_buffer("bn.num_batches_tracked", self._nbt[0])


def test_parameters_are_views_of_one_flat_buffer_in_layout_order():
    torch.manual_seed(0)
    ref = [p.detach().clone() for p in nn.Sequential(nn.Linear(3, 2), nn.Linear(2, 1)).parameters()]
    torch.manual_seed(0)
    m = _Toy()
    assert m.num_live == 6 + 2 + 2 + 1 + 2 + 2 and m.flat_params.numel() == m.num_live     # a.w a.b b.w b.b bn.weight bn.bias
    assert m.bucket.numel() == m.num_live + 3
    off = 0
    for (name, p), (o, n, shape) in zip(m._named_live(), m._slices):
        assert o == off and tuple(p.shape) == shape
        assert p.data_ptr() == m.flat_params.data_ptr() + 4 * o                            # a view, not a copy
        off += n
    for p, r in zip(list(m.parameters())[:4], ref):                                        # same RNG consumption as the plain modules
        assert torch.equal(p.detach(), r)
    m.flat_params.fill_(0.5)
    assert all(float(p.detach().min()) == 0.5 for p in m.parameters())


def test_flat_order_overrides_named_parameters_order():
    m = _Toy(order=["b.weight", "b.bias", "a.weight", "a.bias", "bn.weight", "bn.bias"])
    assert list(m._layout) == ["b.weight", "b.bias", "a.weight", "a.bias", "bn.weight", "bn.bias"]
    assert m.b.weight.data_ptr() == m.flat_params.data_ptr() and m.a.weight.data_ptr() == m.flat_params.data_ptr() + 4 * 3


def test_noop_apply_keeps_the_buffers_and_a_real_conversion_rebuilds_them():
    m = _Toy()
    flat0, hits = m.flat_params, []
    m._reflatten_listeners = [lambda: hits.append(1)]
    m.to(torch.device("cpu")).float()                                                      # the trainers' per-epoch no-op
    assert m.flat_params is flat0 and not hits
    m.double()                                                                             # converts tensor by tensor: views are gone
    assert m.flat_params is not flat0 and m.flat_params.dtype == torch.float32 and hits    # rebuilt (fp32 storage), listeners told
    assert PL.flat_views_intact(m)
    assert m.bn.running_mean.data_ptr() == m._bn.data_ptr()


def test_batchnorm_counter_is_flushed_when_somebody_looks():
    m = _Toy()
    m._nbt_pending = 5
    assert int(m.state_dict()["bn.num_batches_tracked"]) == 5 and m._nbt_pending == 0


def test_workspace_cache_evicts_the_oldest_size_unless_pinned():
    m = _Toy()
    sizes = []
    for B in (4, 8, 16):
        ws, pred = m._workspace_entry(B, lambda: 64, "unsupported")
        assert ws.numel() == 64 and pred.numel() == B
        sizes.append(B)
    assert list(m._bufs) == [8, 16]
    m._pin_bufs = True
    m._workspace_entry(32, lambda: 64, "unsupported")
    assert list(m._bufs) == [8, 16, 32]
    with pytest.raises(RuntimeError, match="not covered"):
        m._workspace_entry(64, lambda: 0, "configuration not covered")
    ent = m._workspace_entry(7, lambda: 16, "x", make=lambda dev: (torch.zeros(2), torch.zeros(3)))
    assert len(ent) == 3 and ent[2].numel() == 3


def test_adam_block_and_side_stream_defaults():
    m = _Toy()
    assert m._adam_args(None) is None
    s = PL.SideStream()
    assert s.pointer(torch.device("cpu"), training=False) is None
    s.enabled = False
    assert s.pointer(torch.device("cpu"), training=True) is None
