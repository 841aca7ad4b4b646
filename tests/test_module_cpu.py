"""CPU-side checks of the drop-in module surface (no kernels run): state_dict keys / order / shapes
and initial values equal the reference's, flat-buffer views stay coherent, the registry behaves
like the reference's, and the compute path refuses to run without a GPU."""
import os

import numpy as np
import pytest
import torch

from gnn_rul_benchmarking_amd import params as PL
from gnn_rul_benchmarking_amd.algorithms import ST_GCN, get_algorithm_class
from gnn_rul_benchmarking_amd.stgcn import ST_GCN_model

from conftest import GOLDEN


def _seed(s):
    import random
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


def test_state_dict_matches_reference_keys_order_shapes_and_init_values():
    z = np.load(os.path.join(GOLDEN, "stgcn_init_14x30_seed3.npz"))
    _seed(int(z["seed"]))
    m = ST_GCN_model(14, 30, dropout=0.2)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in z["key_order"]]
    assert len(sd) == 52
    for k, v in sd.items():
        ref = z["sd:" + k]
        assert tuple(v.shape) == tuple(ref.shape), k
        assert np.array_equal(v.cpu().numpy(), ref), k        # same RNG consumption order -> same init


def test_live_parameters_are_views_of_the_flat_buffer():
    m = ST_GCN_model(14, 30)
    flat = m.flat_params
    assert flat.numel() == 1525 == PL.param_count(14, 2)
    table = dict(m.named_parameters())
    for name, (off, shape) in PL.live_param_layout(14, 2).items():
        p = table[name]
        assert p.data_ptr() == flat.data_ptr() + 4 * off
        assert tuple(p.shape) == shape
    # writing through load_state_dict lands in the flat buffer
    sd = {k: torch.full_like(v, 0.25) if v.is_floating_point() else v for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    assert torch.all(flat == 0.25)
    assert torch.all(m._bn == 0.25)
    # dead branches exist, are parameters, and are not part of the flat buffer
    dead = [n for n, _ in m.named_parameters() if ".net0." in n or ".net1." in n]
    assert len(dead) == 20
    # a dtype/device round trip keeps the views coherent
    m.double().float()
    assert dict(m.named_parameters())["fc1.weight"].data_ptr() == m.flat_params.data_ptr() + 4 * PL.live_param_layout(14, 2)["fc1.weight"][0]


def test_registry_contract():
    assert get_algorithm_class("ST_GCN") is ST_GCN
    with pytest.raises(NotImplementedError, match="Algorithm not found: FC_STGNNX"):
        get_algorithm_class("FC_STGNNX")
    with pytest.raises(NotImplementedError):
        get_algorithm_class("Algorithm")
    algo = ST_GCN({"num_patch": 14, "patch_size": 30, "dropout": 0.2}, {"learning_rate": 1e-4, "weight_decay": 1e-4}, "cpu")
    keys = list(algo.state_dict().keys())
    assert all(k.startswith("model.") for k in keys) and len(keys) == 52      # utils.py:111-120 checkpoint format
    assert algo.hparams["learning_rate"] == 1e-4
    assert isinstance(algo.mse, torch.nn.MSELoss)


def test_no_cpu_fallback():
    m = ST_GCN_model(14, 30)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(4, 14, 30))
    m.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(4, 14, 30))
    algo = ST_GCN({"num_patch": 14, "patch_size": 30}, {"learning_rate": 1e-4, "weight_decay": 1e-4}, "cpu")
    algo.train()
    with pytest.raises(RuntimeError):
        algo.update(torch.rand(4, 14, 30), torch.rand(4, 1), 1)


def test_bad_input_shape_raises_like_reference_reshape():
    m = ST_GCN_model(14, 30)
    with pytest.raises(RuntimeError):
        m(torch.rand(4, 14, 31))


@pytest.mark.parametrize("family", ["ST_GCN", "STMSGCN", "ASTGCNN", "FC_STGNN", "ST_Conv", "STGNN", "HAGCN"])
def test_noop_to_keeps_the_flat_buffers_and_a_real_move_notifies_listeners(family):
    """The trainers call ``algorithm.model.to(device)`` every epoch (reference trainer.py:135): on an unchanged device that must
    not reallocate the flat parameter / BatchNorm / bucket buffers -- captured hipGraphs and Adam state point into them."""
    from gnn_rul_benchmarking_amd import hparams as H
    cfgs = {
        "ST_GCN": dict(num_patch=14, patch_size=30, dropout=0.2),
        "STMSGCN": dict(num_patch=6, patch_size=20, interval=2, band_width=3, gcn_dims=[4, 6, 2, 1], gru_hidden_dim=4),
        "ASTGCNN": dict(num_nodes=5, time_length=12, encoder_out_dim=12, output_dim=8, K=3),
        "FC_STGNN": H.get_hparams_class("CMAPSS")("FD004").alg_hparams["FC_STGNN"],
        "ST_Conv": H.get_hparams_class("CMAPSS")("FD004").alg_hparams["ST_Conv"],
        "STGNN": H.get_hparams_class("CMAPSS")("FD004").alg_hparams["STGNN"],
        "HAGCN": H.get_hparams_class("CMAPSS")("FD004").alg_hparams["HAGCN"],
    }
    algo = get_algorithm_class(family)(cfgs[family], {"learning_rate": 1e-3, "weight_decay": 0.0, "alpha": 100}, "cpu")
    model = algo.model
    flat, bucket = model.flat_params, model._grad_flat
    before = {k: v.clone() for k, v in model.state_dict().items()}
    calls = []
    model._reflatten_listeners = [lambda: calls.append(1)]
    model.to("cpu")
    model.float()
    algo.to(torch.device("cpu"))
    assert model.flat_params is flat and model._grad_flat is bucket and not calls
    model._reflatten()                                   # what a real device move ends with
    assert model.flat_params is not flat and calls == [1]
    after = model.state_dict()
    assert list(after) == list(before) and all(torch.equal(after[k], before[k]) for k in before)


def test_forward_tape_tokens():
    """params.ForwardTape: the token of a batch size changes with every forward that writes its workspace; a stale token or an evicted
    workspace raises (the four flat-parameter models of round 2 use it in their autograd Functions)."""
    from gnn_rul_benchmarking_amd.params import ForwardTape
    tape, bufs = ForwardTape(), {8: object(), 4: object()}
    t8 = tape.mark(8)
    t4 = tape.mark(4)
    tape.check(8, t8, bufs, "M")
    tape.check(4, t4, bufs, "M")
    tape.mark(8)
    with pytest.raises(RuntimeError, match="overwritten"):
        tape.check(8, t8, bufs, "M")
    tape.check(4, t4, bufs, "M")
    del bufs[4]
    with pytest.raises(RuntimeError, match="evicted"):
        tape.check(4, t4, bufs, "M")


@pytest.mark.parametrize("name", ["stgcn_order2_14x30_bs19", "stgcn_order3_9x21_layers3_bs7"])
def test_mpnn_order_above_one_module_surface_matches_the_reference(name):
    """ST_GCN_model(..., k = 2 / 3): the parameters, their order and shapes are the reference's (``named_parameters()`` of its own model,
    tests/golden/make_golden_order.py); the live ones are views of the flat buffer in that order; its state_dict loads strictly."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N, P, L, K = int(z["num_patch"]), int(z["patch_size"]), int(z["num_layers"]), int(z["k"])
    m = ST_GCN_model(N, P, num_layers=L, dropout=0.2, k=K)
    assert [n for n, _ in m.named_parameters()] == [str(k) for k in z["key_order"]]
    table = dict(m.named_parameters())
    assert m.flat_params.numel() == PL.param_count(N, L, K)
    off_prev = -1
    for pname, (off, shape) in PL.live_param_layout(N, L, K).items():
        p = table[pname]
        assert p.data_ptr() == m.flat_params.data_ptr() + 4 * off and tuple(p.shape) == shape and off > off_prev
        off_prev = off
    live = [n for n in z["key_order"].tolist() if ".net0." not in n and ".net1." not in n]
    assert live == list(PL.live_param_layout(N, L, K).keys())
    sd = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd:")}
    res = m.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for pname in live:
        assert np.array_equal(table[pname].detach().numpy(), z["sd:" + pname]), pname
