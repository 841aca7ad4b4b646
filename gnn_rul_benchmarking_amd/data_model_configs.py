"""Dataset constants, as the reference's configs/data_model_configs.py:7-47 states them."""


def get_dataset_class(dataset_name):
    """Return the dataset-config class with the given name."""
    if dataset_name not in _DATASETS:
        raise NotImplementedError("Dataset not found: {}".format(dataset_name))
    return _DATASETS[dataset_name]


class _Cfg:
    sequence_len = 0
    input_channels = 0
    shuffle = True

    def __init__(self):
        self.sequence_len = type(self).sequence_len
        self.input_channels = type(self).input_channels
        self.shuffle = type(self).shuffle
        self.drop_last = False
        self.normalize = False


class CMAPSS(_Cfg):
    sequence_len, input_channels, shuffle = 50, 14, True


class NCMAPSS(_Cfg):
    sequence_len, input_channels, shuffle = 50, 20, True


class PHM2012(_Cfg):
    sequence_len, input_channels, shuffle = 2560, 1, False


class XJTU_SY(_Cfg):
    # the reference says 30768 (data_model_configs.py:43), a typo for 32768; kept as the reference has it
    sequence_len, input_channels, shuffle = 30768, 1, False


_DATASETS = {'CMAPSS': CMAPSS, 'NCMAPSS': NCMAPSS, 'PHM2012': PHM2012, 'XJTU_SY': XJTU_SY}
