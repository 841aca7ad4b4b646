"""HAGCN host side without a GPU: state_dict surface, init parity with the reference, C-ABI shape rules, hparams."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from gnn_rul_benchmarking_amd import _lib
from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
from gnn_rul_benchmarking_amd.hagcn import HAGCN_model
from gnn_rul_benchmarking_amd.hparams import get_hparams_class

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_state_dict_and_initial_weights_equal_the_reference_for_the_same_seed():
    z = np.load(os.path.join(GOLD, "hagcn_init_fd004_seed65.npz"))
    cfg = {k[4:]: int(z[k]) for k in z.files if k.startswith("cfg:")}
    assert cfg == get_hparams_class("CMAPSS")("FD004").alg_hparams["HAGCN"]
    torch.manual_seed(int(z["seed"]))
    algo = get_algorithm_class("HAGCN")(cfg, get_hparams_class("CMAPSS")("FD004").train_params["HAGCN"], "cpu")
    assert algo.alpha == 100
    sd = algo.model.state_dict()
    assert list(sd.keys()) == list(z["keys"]) and len(sd) == 67                   # HAGCN 67 entries
    assert sum(v.numel() for v in sd.values()) == int(z["numel"].sum()) == 365770
    for i, (k, v) in enumerate(sd.items()):
        assert abs(float(v.double().sum()) - z["sums"][i]) < 1e-9 * max(1.0, abs(z["sums"][i])), k
        assert abs(float(v.double().abs().sum()) - z["abssums"][i]) < 1e-9 * max(1.0, z["abssums"][i]), k


def test_graph_parameters_are_views_of_the_flat_buffer():
    m = HAGCN_model(patch_size=25, num_patch=2, encoder_hidden_dim=60, hidden_dim=64, output_dim=32)
    assert _lib.load().rulgnn_hagcn_graph_param_count(C.byref(_lib.HagcnShape(10, 14, 60, 64))) == m._count == 43721
    sd = {k: torch.randn_like(v) for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    for name, (off, shp) in m._layout.items():
        assert torch.equal(m.flat_params[off:off + int(np.prod(shp))].view(shp), sd[name])


def test_abi_shape_rules_hparams_and_loud_rejections():
    lib = _lib.load()
    S = _lib.HagcnShape
    assert lib.rulgnn_hagcn_workspace_bytes(C.byref(S(200, 14, 60, 64))) > 0
    assert lib.rulgnn_hagcn_graph_param_count(C.byref(S(200, 9, 60, 64))) == -1     # SAGPool keeps 10 nodes
    assert lib.rulgnn_hagcn_graph_param_count(C.byref(S(200, 21, 60, 64))) == -1
    assert lib.rulgnn_hagcn_graph_param_count(C.byref(S(200, 14, 65, 64))) == -1
    assert lib.rulgnn_hagcn_workspace_bytes(C.byref(S(200, 14, 60, 63))) == 0
    assert lib.rulgnn_hagcn_graph_forward_f32(None, None, None) == -1
    for fd, (ps, npatch) in {"FD001": (10, 5), "FD002": (25, 2), "FD003": (25, 2), "FD004": (50, 1)}.items():
        h = get_hparams_class("CMAPSS")(fd)
        assert h.alg_hparams["HAGCN"] == {"patch_size": ps, "num_patch": npatch, "hidden_dim": 64, "encoder_hidden_dim": 60, "output_dim": 32}
        assert h.train_params["HAGCN"]["alpha"] == 100
    m = HAGCN_model(patch_size=25, num_patch=2, encoder_hidden_dim=60, hidden_dim=64, output_dim=32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(2, 14, 50))
    algo = get_algorithm_class("HAGCN")(get_hparams_class("NCMAPSS")(None).alg_hparams["HAGCN"],
                                        get_hparams_class("NCMAPSS")(None).train_params["HAGCN"], "cpu")
    with pytest.raises(RuntimeError, match="replicas"):
        algo.attach_data_parallel(object())
