"""STGNN host side without a GPU: state_dict surface, init parity with the reference, C-ABI shape rules, hparams."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from gnn_rul_benchmarking_amd import _lib
from gnn_rul_benchmarking_amd.algorithms import get_algorithm_class
from gnn_rul_benchmarking_amd.hparams import get_hparams_class
from gnn_rul_benchmarking_amd.stgnn import STGNN_model

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CFG = dict(patch_size=50, num_patch=1, num_nodes=14, hidden_dim=64, K=3, top_k=10)


def test_state_dict_and_initial_weights_equal_the_reference_for_the_same_seed():
    z = np.load(os.path.join(GOLD, "stgnn_init_cmapss_seed5.npz"))
    torch.manual_seed(int(z["seed"]))
    m = STGNN_model(**CFG)
    sd = m.state_dict()
    assert list(sd.keys()) == [k[3:] for k in z.files if k.startswith("sd:")]
    for k in sd:
        assert np.array_equal(sd[k].numpy(), z["sd:" + k]), k
    zc = np.load(os.path.join(GOLD, "stgnn_train_curve_1x50_bs16.npz"))
    torch.manual_seed(int(zc["seed"]))
    algo = get_algorithm_class("STGNN")(CFG, {"learning_rate": 1e-3, "weight_decay": 1e-4}, "cpu")
    for k, v in algo.state_dict().items():
        assert np.array_equal(v.numpy(), zc["sd0:" + k]), k


def test_abi_rules_hparams_and_loud_cpu_rejection():
    lib, S = _lib.load(), _lib.StgnnShape
    assert lib.rulgnn_stgnn_workspace_bytes(C.byref(S(100, 14, 1, 50, 64, 3, 10))) > 0
    assert lib.rulgnn_stgnn_workspace_bytes(C.byref(S(100, 20, 5, 10, 64, 3, 10))) > 0
    assert lib.rulgnn_stgnn_workspace_bytes(C.byref(S(100, 33, 1, 50, 64, 3, 10))) == 0       # more nodes than a graph tile holds
    assert lib.rulgnn_stgnn_workspace_bytes(C.byref(S(100, 14, 1, 129, 64, 3, 10))) == 0
    assert lib.rulgnn_stgnn_workspace_bytes(C.byref(S(100, 14, 1, 50, 64, 5, 10))) == 0
    assert lib.rulgnn_stgnn_workspace_bytes(C.byref(S(100, 14, 1, 50, 64, 3, 15))) == 0       # top_k > nodes: torch.topk raises
    assert lib.rulgnn_stgnn_terms_f32(None, None, None, None, None) == -1
    assert lib.rulgnn_stgnn_cheb_forward_f32(C.byref(S(4, 14, 1, 50, 64, 3, 10)), None, None, None, None) == -1
    for fd in ("FD001", "FD002", "FD003", "FD004"):
        h = get_hparams_class("CMAPSS")(fd)
        assert h.alg_hparams["STGNN"] == CFG
        assert h.train_params["STGNN"] == {"num_epochs": 81, "batch_size": 100, "weight_decay": 1e-4, "learning_rate": 1e-3}
    assert get_hparams_class("NCMAPSS")(None).alg_hparams["STGNN"] == dict(CFG, patch_size=10, num_patch=5, num_nodes=20)
    m = STGNN_model(**CFG)
    with pytest.raises(RuntimeError, match="HIP path only"):
        m(torch.rand(2, 14, 50))
    with pytest.raises(RuntimeError, match="expects"):
        m(torch.rand(2, 14, 51))
