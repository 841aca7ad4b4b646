"""ASTGCNN host side without a GPU: state_dict surface, init parity with the reference, C-ABI shape rules, hparams."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from gnn_rul_benchmarking_amd import _lib
from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
from gnn_rul_benchmarking_amd.astgcnn import ASTGCNN_model, live_layout

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(num_nodes=14, time_length=50, encoder_out_dim=50, output_dim=64, K=3)


def test_state_dict_and_initial_weights_equal_the_reference_for_the_same_seed():
    z = np.load(os.path.join(GOLD, "astgcnn_train_curve_14x50_bs20.npz"))
    torch.manual_seed(int(z["seed"]))
    algo = get_algorithm_class("ASTGCNN")(CFG, {"learning_rate": 1e-3, "weight_decay": 1e-4}, "cpu")
    sd = algo.state_dict()
    ref = {k[4:]: z[k] for k in z.files if k.startswith("sd0:")}
    assert list(sd.keys()) == list(ref.keys()) and len(sd) == 29           # ASTGCNN 29 entries
    for k, v in ref.items():
        assert np.array_equal(sd[k].numpy(), v), k


def test_parameter_count_and_flat_views():
    m = ASTGCNN_model(**dict(CFG, num_nodes=20))
    assert m.num_live == 19645 == live_layout(20, 50, 64, 3)[1]                 # 19,645 live
    assert sum(p.numel() for p in m.parameters()) == 29365                      # 29,365 params
    assert _lib.load().rulgnn_astgcnn_param_count(C.byref(_lib.AstgcnnShape(8, 20, 50, 64, 3))) == 19645
    sd = {k: torch.randn_like(v) if v.is_floating_point() else v for k, v in m.state_dict().items()}
    m.load_state_dict(sd)
    for name, (off, shp) in m._layout.items():
        assert torch.equal(m.flat_params[off:off + int(np.prod(shp))].view(shp), sd[name])
    assert torch.equal(m._bn[:20], sd["tcn.conv_block1.2.running_mean"])
    assert m.bucket.numel() == m.num_live + 1 + 80


def test_abi_shape_rules_and_loud_cpu_rejection():
    lib = _lib.load()
    S = _lib.AstgcnnShape
    assert lib.rulgnn_astgcnn_workspace_bytes(C.byref(S(100, 14, 50, 64, 3))) > 0
    assert lib.rulgnn_astgcnn_param_count(C.byref(S(100, 26, 50, 64, 3))) == -1     # torch.cdist switches algorithm above 25 rows
    assert lib.rulgnn_astgcnn_param_count(C.byref(S(100, 14, 65, 64, 3))) == -1
    assert lib.rulgnn_astgcnn_param_count(C.byref(S(100, 14, 50, 64, 4))) == -1
    assert lib.rulgnn_astgcnn_workspace_bytes(C.byref(S(100, 14, 50, 0, 3))) == 0
    assert lib.rulgnn_astgcnn_forward_f32(None, None, None) == -1
    m = ASTGCNN_model(**CFG)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(2, 14, 50))


def test_hparams_rows_are_the_reference_tables():
    """configs/hparams.py:19,38,... (C-MAPSS FD001-4) and :184,202 (N-CMAPSS) of the reference, restated."""
    from gnn_rul_benchmarking_amd.hparams import get_hparams_class
    train = {"num_epochs": 81, "batch_size": 100, "weight_decay": 1e-4, "learning_rate": 1e-3}
    for fd in ("FD001", "FD002", "FD003", "FD004"):
        h = get_hparams_class("CMAPSS")(fd)
        assert h.alg_hparams["ASTGCNN"] == CFG and h.train_params["ASTGCNN"] == train
    h = get_hparams_class("NCMAPSS")(None)
    assert h.alg_hparams["ASTGCNN"] == dict(CFG, num_nodes=20) and h.train_params["ASTGCNN"] == train
    with pytest.raises(KeyError):
        get_hparams_class("PHM2012")("Condition_1").alg_hparams["ASTGCNN"]
