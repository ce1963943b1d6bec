"""Documented synthetic inputs of SURVEY.md section 8(d): float32 stand-ins for the data sets the
reference's configurations name (no data set ships with the reference, and there is no network).
numpy only; bit-identical to the copies in oracle/vgan_oracle.py (tests/test_host_logic.py checks)."""
import numpy as np


def latent_size(d):
    """src/vgan.py:559: max(int(d/16), 1)."""
    return max(int(d / 16), 1)


def synthetic_dataset(config, rows=None, seed=0):
    """c1: 2-Gaussian mixture d=20; c2: ADBench 'musk' stand-in d=166 (standardised, block-correlated);
    c3: MNIST-pixel stand-in d=784 (values in [0,1], ~80% exact zeros, rank-32 structure + noise);
    c4 / c5: synthetic tabular d=2048 / 4096, N(0,1) with 64 planted correlated feature blocks."""
    rng = np.random.default_rng(seed)
    if config == "c1":
        d, n = 20, 128
        rows = rows or 16 * n
        sign = np.where(rng.random(rows) < 0.5, -2.0, 2.0)[:, None]
        X = rng.normal(size=(rows, d)) + sign
    elif config == "c2":
        d, n = 166, 512
        rows = rows or 3062
        blocks = rng.normal(size=(rows, 16)) @ rng.normal(size=(16, d))
        X = blocks + 0.5 * rng.normal(size=(rows, d))
        X = (X - X.mean(0)) / X.std(0)
    elif config == "c3":
        d, n = 784, 1024
        rows = rows or 16 * n
        low = rng.random(size=(rows, 32)) @ rng.random(size=(32, d)) / 16.0
        X = np.clip(low + 0.05 * rng.normal(size=(rows, d)), 0.0, 1.0)
        X = X * (rng.random(size=(rows, d)) < 0.2)
    elif config in ("c4", "c5"):
        # synthetic tabular: N(0,1) features with 64 planted correlated blocks (each block shares one latent factor)
        d, n = (2048, 4096) if config == "c4" else (4096, 8192)
        rows = rows or 4 * n
        rng32 = np.random.default_rng(seed + (4 if config == "c4" else 5))
        X = rng32.standard_normal(size=(rows, d), dtype=np.float32)
        factors = rng32.standard_normal(size=(rows, 64), dtype=np.float32)
        width = d // 128  # 64 blocks covering half of the features
        for b in range(64):
            X[:, 2 * b * width:(2 * b + 1) * width] = 0.6 * X[:, 2 * b * width:(2 * b + 1) * width] + 0.8 * factors[:, b:b + 1]
    else:
        raise ValueError(config)
    return np.ascontiguousarray(X, dtype=np.float32)


def synthetic_generator_params(d, seed=0, dtype=np.float32):
    """PyTorch-default-like Linear init (U(-1/sqrt(in), 1/sqrt(in))) from a numpy stream."""
    rng = np.random.default_rng(seed + 1000)
    L = latent_size(d)
    params = []
    for out, inp in [(2 * L, L), (4 * L, 2 * L), (8 * L, 4 * L), (d, 8 * L)]:
        k = 1.0 / np.sqrt(inp)
        params.append(rng.uniform(-k, k, size=(out, inp)).astype(dtype))
        params.append(rng.uniform(-k, k, size=(out,)).astype(dtype))
    return params
