"""Hyper-parameter tables, looked up by dataset name then dataset id, like the reference's
configs/hparams.py:3-7 (``get_hparams_class(name)(dataset_id)`` -> object with ``train_params`` and
``alg_hparams`` dicts keyed by ``--GNN_method``; unknown dataset -> NotImplementedError, unknown id ->
ValueError).  Only the ST_GCN, STMSGCN and ASTGCNN rows are restated (the methods this package implements).

PHM2012 / XJTU_SY rows are the reference's (configs/hparams.py:223,238,... and :334,349,...; STMSGCN
:226,242,275,311,355,390,424).
The CMAPSS / NCMAPSS rows are a BUILD EXTENSION: the reference never pairs ST_GCN with the aero-engine
datasets although the model only needs numel/bs == num_patch*patch_size.  Here each
sensor's window is one patch: num_patch = sensors (14 / 20), patch_size = window length (30 per
BASELINE.json; the reference's preprocessed C-MAPSS windows are 50 long -- pass ``window=50``), with the
reference's ST_GCN training parameters (lr 1e-4, wd 1e-4, 81 epochs, batch 100) and dropout 0.2."""
from __future__ import annotations

_ST_GCN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-4}
_STMSGCN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 0, 'learning_rate': 1e-2}
_MSG = {'gcn_dims': [16, 64, 16, 1], 'gru_hidden_dim': 8}
_ASTGCNN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-3}


def get_hparams_class(dataset_name):
    """Return the hparams class with the given name."""
    if dataset_name not in _DATASETS:
        raise NotImplementedError("Dataset not found: {}".format(dataset_name))
    return _DATASETS[dataset_name]


class _Table:
    _rows: dict = {}
    _stmsgcn_rows: dict = {}          # the reference wires STMSGCN to the bearing datasets only
    _astgcnn_nodes = None             # ... and ASTGCNN to the aero-engine datasets only (configs/hparams.py:38,202)

    def __init__(self, dataset_id=None, **overrides):
        if dataset_id not in self._rows:
            raise ValueError(f"No hparams found for dataset: {dataset_id}")
        self.train_params = {'ST_GCN': dict(_ST_GCN_TRAIN)}
        self.alg_hparams = {'ST_GCN': dict(self._rows[dataset_id])}
        self.alg_hparams['ST_GCN'].update(overrides)
        if self._astgcnn_nodes:
            self.train_params['ASTGCNN'] = dict(_ASTGCNN_TRAIN)
            self.alg_hparams['ASTGCNN'] = {'num_nodes': self._astgcnn_nodes, 'time_length': 50, 'encoder_out_dim': 50,
                                           'output_dim': 64, 'K': 3}
        if dataset_id in self._stmsgcn_rows:
            self.train_params['STMSGCN'] = dict(_STMSGCN_TRAIN)
            self.alg_hparams['STMSGCN'] = dict(self._stmsgcn_rows[dataset_id], gcn_dims=list(_MSG['gcn_dims']),
                                               gru_hidden_dim=_MSG['gru_hidden_dim'])


class CMAPSS(_Table):
    _astgcnn_nodes = 14

    def __init__(self, dataset_id, window=30):
        self._rows = {fd: {'num_patch': 14, 'patch_size': int(window), 'dropout': 0.2}
                      for fd in ('FD001', 'FD002', 'FD003', 'FD004')}
        super().__init__(dataset_id)


class NCMAPSS(_Table):
    _astgcnn_nodes = 20

    def __init__(self, dataset_id=None, window=50):
        self._rows = {None: {'num_patch': 20, 'patch_size': int(window), 'dropout': 0.2}}
        super().__init__(None)


class PHM2012(_Table):
    _rows = {'Condition_1': {'num_patch': 40, 'patch_size': 64, 'dropout': 0.2},
             'Condition_2': {'num_patch': 160, 'patch_size': 16, 'dropout': 0.2},
             'Condition_3': {'num_patch': 40, 'patch_size': 64, 'dropout': 0.2}}
    _stmsgcn_rows = {'Condition_1': {'num_patch': 160, 'patch_size': 16, 'interval': 6, 'band_width': 5},
                     'Condition_2': {'num_patch': 128, 'patch_size': 20, 'interval': 2, 'band_width': 3},
                     'Condition_3': {'num_patch': 160, 'patch_size': 16, 'interval': 6, 'band_width': 5}}


class XJTU_SY(_Table):
    _rows = {'Condition_1': {'num_patch': 1024, 'patch_size': 32, 'dropout': 0.3},
             'Condition_2': {'num_patch': 2048, 'patch_size': 16, 'dropout': 0.2},
             'Condition_3': {'num_patch': 2048, 'patch_size': 16, 'dropout': 0.2}}
    _stmsgcn_rows = {'Condition_1': {'num_patch': 256, 'patch_size': 128, 'interval': 3, 'band_width': 5},
                     'Condition_2': {'num_patch': 128, 'patch_size': 256, 'interval': 6, 'band_width': 10},
                     'Condition_3': {'num_patch': 256, 'patch_size': 128, 'interval': 3, 'band_width': 5}}


_DATASETS = {'CMAPSS': CMAPSS, 'NCMAPSS': NCMAPSS, 'PHM2012': PHM2012, 'XJTU_SY': XJTU_SY}
