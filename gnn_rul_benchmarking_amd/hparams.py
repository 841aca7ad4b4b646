"""Hyper-parameter tables, looked up by dataset name then dataset id, like the reference's
configs/hparams.py:3-7 (``get_hparams_class(name)(dataset_id)`` -> object with ``train_params`` and
``alg_hparams`` dicts keyed by ``--GNN_method``; unknown dataset -> NotImplementedError, unknown id ->
ValueError).  Only the ST_GCN, STMSGCN, ASTGCNN, FC_STGNN, HAGCN, ST_Conv, STGNN, RGCNU, STNet, SAGCN and STAGNN rows are restated (the methods this package
implements).

PHM2012 / XJTU_SY rows are the reference's (configs/hparams.py:223,238,... and :334,349,...; STMSGCN
:226,242,275,311,355,390,424).
The CMAPSS / NCMAPSS rows are a BUILD EXTENSION: the reference never pairs ST_GCN with the aero-engine
datasets although the model only needs numel/bs == num_patch*patch_size.  Here each
sensor's window is one patch: num_patch = sensors (14 / 20), patch_size = window length (default 50, the
reference's preprocessed C-MAPSS windows; BASELINE.json's config 0 and bench.py use 30 -- pass ``window=30``), with the
reference's ST_GCN training parameters (lr 1e-4, wd 1e-4, 81 epochs, batch 100) and dropout 0.2."""
from __future__ import annotations

_ST_GCN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-4}
_STMSGCN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 0, 'learning_rate': 1e-2}
_MSG = {'gcn_dims': [16, 64, 16, 1], 'gru_hidden_dim': 8}
_ASTGCNN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-3}
_HAGCN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-3, 'alpha': 100}
# configs/hparams.py:41,79,119,159 (C-MAPSS FD001-4) and :204 (N-CMAPSS)
_HAGCN_PATCH = {'FD001': (10, 5), 'FD002': (25, 2), 'FD003': (25, 2), 'FD004': (50, 1), None: (25, 2)}
_FC_STGNN_TRAIN = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-3}
# configs/hparams.py:32,69,109,149 (C-MAPSS FD001-4) and :196 (N-CMAPSS)
_FC_STGNN_ROWS = {
    'FD001': {'patch_size': 25, 'num_patch': 2, 'encoder_time_out': 27, 'encoder_hidden_dim': 8, 'encoder_out_dim': 32,
              'encoder_conv_kernel': 2, 'hidden_dim': 8, 'num_sequential': 6, 'num_node': 14, 'num_windows': 2},
    'FD002': {'patch_size': 1, 'num_patch': 50, 'encoder_time_out': 3, 'encoder_hidden_dim': 8, 'encoder_out_dim': 12,
              'encoder_conv_kernel': 2, 'hidden_dim': 8, 'num_sequential': 10, 'num_node': 14, 'num_windows': 74},
    'FD003': {'patch_size': 1, 'num_patch': 50, 'encoder_time_out': 3, 'encoder_hidden_dim': 8, 'encoder_out_dim': 6,
              'encoder_conv_kernel': 2, 'hidden_dim': 24, 'num_sequential': 25, 'num_node': 14, 'num_windows': 74},
    'FD004': {'patch_size': 2, 'num_patch': 25, 'encoder_time_out': 4, 'encoder_hidden_dim': 8, 'encoder_out_dim': 6,
              'encoder_conv_kernel': 2, 'hidden_dim': 8, 'num_sequential': 10, 'num_node': 14, 'num_windows': 36},
    None: {'patch_size': 2, 'num_patch': 25, 'encoder_time_out': 4, 'encoder_hidden_dim': 8, 'encoder_out_dim': 32,
           'encoder_conv_kernel': 2, 'hidden_dim': 8, 'num_sequential': 6, 'num_node': 20, 'num_windows': 36},
}


def get_hparams_class(dataset_name):
    """Return the hparams class with the given name."""
    if dataset_name not in _DATASETS:
        raise NotImplementedError("Dataset not found: {}".format(dataset_name))
    return _DATASETS[dataset_name]


class _Table:
    _rows: dict = {}
    _stmsgcn_rows: dict = {}          # the reference wires STMSGCN to the bearing datasets only
    _stnet_rows: dict = {}            # ... and STNet (configs/hparams.py:222,236,267,303 and :333,347,382,416)
    _sagcn_rows: dict = {}            # ... and SAGCN (configs/hparams.py:221,235,266,302 and :332,346,381,415)
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
            self.train_params['ST_Conv'] = dict(_ASTGCNN_TRAIN)            # configs/hparams.py:21,40,... : same values
            self.alg_hparams['ST_Conv'] = {'num_nodes': self._astgcnn_nodes, 'time_length': 50, 'kernel_size': 6}
            ps, npatch = _HAGCN_PATCH[dataset_id]
            self.train_params['HAGCN'] = dict(_HAGCN_TRAIN)
            self.alg_hparams['HAGCN'] = {'patch_size': ps, 'num_patch': npatch, 'hidden_dim': 64, 'encoder_hidden_dim': 60,
                                         'output_dim': 32}
            # configs/hparams.py:23,42,80,120,160 (C-MAPSS FD001-4) and :187,205 (N-CMAPSS): one row everywhere
            self.train_params['RGCNU'] = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-3, 'lambda': 0.1}
            self.alg_hparams['RGCNU'] = {'num_nodes': self._astgcnn_nodes, 'time_length': 50, 'hidden_dim': 32, 'encoder_hidden_dim': 32,
                                         'kernel_size': 3, 'alpha': 1}
            # configs/hparams.py:27,47,88,128,168 (C-MAPSS: one patch of 50) and :191,211 (N-CMAPSS: 5 patches of 10)
            self.train_params['STGNN'] = dict(_ASTGCNN_TRAIN)
            self.alg_hparams['STGNN'] = {'patch_size': 50 if self._astgcnn_nodes == 14 else 10,
                                         'num_patch': 1 if self._astgcnn_nodes == 14 else 5, 'num_nodes': self._astgcnn_nodes,
                                         'hidden_dim': 64, 'K': 3, 'top_k': 10}
            # configs/hparams.py:24,43,82,122,162 (C-MAPSS FD001-4: hidden 64 / 16 / 32 / 32) and :188,206 (N-CMAPSS: hidden 32)
            self.train_params['STAGNN'] = dict(_ASTGCNN_TRAIN)
            self.alg_hparams['STAGNN'] = {'num_nodes': self._astgcnn_nodes, 'time_length': 50,
                                          'hidden_dim': {'FD001': 64, 'FD002': 16, 'FD003': 32, 'FD004': 32, None: 32}[dataset_id],
                                          'output_dim': 10, 'num_heads': 3, 'threshold': 0}
            self.train_params['FC_STGNN'] = dict(_FC_STGNN_TRAIN)
            self.alg_hparams['FC_STGNN'] = dict(_FC_STGNN_ROWS[dataset_id])
        if dataset_id in self._stnet_rows:
            self.train_params['STNet'] = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-2, 'learning_rate': 1e-2}
            num_patch, patch_size, num_nodes, nperseg, input_dim = self._stnet_rows[dataset_id]
            self.alg_hparams['STNet'] = {'num_patch': num_patch, 'patch_size': patch_size, 'num_nodes': num_nodes, 'nperseg': nperseg,
                                         'input_dim': input_dim, 'Cheb_layers': [300, 200, 100], 'lstm_hidden_dim': 10,
                                         'autoencoder_hidden_dim': 50}
        if dataset_id in self._sagcn_rows:
            self.train_params['SAGCN'] = {'num_epochs': 81, 'batch_size': 100, 'weight_decay': 1e-4, 'learning_rate': 1e-4}
            num_patch, patch_size, gcn_hidden, attention_hidden = self._sagcn_rows[dataset_id]
            self.alg_hparams['SAGCN'] = {'num_patch': num_patch, 'patch_size': patch_size, 'gcn_hidden_dim': gcn_hidden,
                                         'attention_hidden_dim': attention_hidden}
        if dataset_id in self._stmsgcn_rows:
            self.train_params['STMSGCN'] = dict(_STMSGCN_TRAIN)
            self.alg_hparams['STMSGCN'] = dict(self._stmsgcn_rows[dataset_id], gcn_dims=list(_MSG['gcn_dims']),
                                               gru_hidden_dim=_MSG['gru_hidden_dim'])


class CMAPSS(_Table):
    _astgcnn_nodes = 14

    def __init__(self, dataset_id, window=50):
        # default window = the reference's preprocessed C-MAPSS windows (data_model_configs.CMAPSS.sequence_len = 50); bench.py and
        # BASELINE.json config 0 use window 30 and pass it explicitly (--window 30 / patch_size)
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
    _stnet_rows = {'Condition_1': (20, 128, 9, 16, 9), 'Condition_2': (20, 128, 9, 16, 9), 'Condition_3': (80, 32, 5, 8, 5)}
    _sagcn_rows = {'Condition_1': (160, 16, 100, 100), 'Condition_2': (128, 20, 1000, 200), 'Condition_3': (128, 20, 1000, 200)}


class XJTU_SY(_Table):
    _rows = {'Condition_1': {'num_patch': 1024, 'patch_size': 32, 'dropout': 0.3},
             'Condition_2': {'num_patch': 2048, 'patch_size': 16, 'dropout': 0.2},
             'Condition_3': {'num_patch': 2048, 'patch_size': 16, 'dropout': 0.2}}
    _stmsgcn_rows = {'Condition_1': {'num_patch': 256, 'patch_size': 128, 'interval': 3, 'band_width': 5},
                     'Condition_2': {'num_patch': 128, 'patch_size': 256, 'interval': 6, 'band_width': 10},
                     'Condition_3': {'num_patch': 256, 'patch_size': 128, 'interval': 3, 'band_width': 5}}
    _stnet_rows = {'Condition_1': (128, 256, 9, 16, 17), 'Condition_2': (32, 1024, 17, 32, 33), 'Condition_3': (64, 512, 17, 32, 17)}
    _sagcn_rows = {'Condition_1': (32, 1024, 1000, 100), 'Condition_2': (32, 1024, 1000, 200), 'Condition_3': (32, 1024, 1000, 200)}


_DATASETS = {'CMAPSS': CMAPSS, 'NCMAPSS': NCMAPSS, 'PHM2012': PHM2012, 'XJTU_SY': XJTU_SY}
