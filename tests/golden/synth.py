"""Deterministic synthetic datasets shared by make_golden.py (reference side) and the tests."""
import numpy as np


def synthetic_phm2012(seed, n_train, n_test):
    """Deterministic PHM2012-shaped vibration snapshots [n, 2560] with a learnable degradation target.
    The tests regenerate exactly this from the seed instead of storing megabytes of noise."""
    rng = np.random.default_rng(seed)
    def make(n):
        life = rng.uniform(0.05, 1.0, n)                               # label: RUL / max_rul
        t = np.arange(2560)[None, :] / 2560.0
        x = (1.2 - life)[:, None] * np.sin(2 * np.pi * (40 + 25 * (1 - life))[:, None] * t) \
            + 0.4 * rng.standard_normal((n, 2560)) * (1.3 - life)[:, None]
        return x.astype(np.float32), life.astype(np.float32)
    return make(n_train), make(n_test)




def synthetic_cmapss(seed, n_train, n_test, window=50, sensors=14):
    """Deterministic C-MAPSS-shaped windows in the reference's on-disk layout [n, window, sensors], values in [0, 1]
    (min-max scaled sensors), label RUL / max_rul in [0, 1]: sensors drift with degradation plus noise."""
    rng = np.random.default_rng(seed)
    slope = rng.uniform(-0.4, 0.4, sensors)

    def make(n):
        life = rng.uniform(0.0, 1.0, n)
        t = np.linspace(0.0, 1.0, window)[None, :, None]
        base = 0.5 + slope[None, None, :] * (1.0 - life)[:, None, None] * (0.7 + 0.3 * t)
        x = np.clip(base + 0.05 * rng.standard_normal((n, window, sensors)), 0.0, 1.0)
        return x.astype(np.float32), life.astype(np.float32)
    return make(n_train), make(n_test)
