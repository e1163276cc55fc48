"""TEST INFRASTRUCTURE — numpy building blocks of the CPU oracle.

Plain-numpy restatement of the handful of tensor ops the reference's hot path
uses (torch semantics), so the oracle runs without torch on the GPU box.
Nothing in `larynx_amd/` may import this package; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg do.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def fold_weight_norm(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """w = g * v / ||v||, norm over every dim but 0 (torch.nn.utils.weight_norm
    dim=0; folded by `remove_weight_norm`, hifi_gan/models.py:204-211,
    glow_tts/layers.py:164-170)."""
    v64 = v.astype(np.float64)
    norm = np.sqrt((v64 ** 2).reshape(v.shape[0], -1).sum(axis=1)).reshape(g.shape)
    return (v64 * (g.astype(np.float64) / norm)).astype(F32)


def conv1d(x: np.ndarray, w: np.ndarray, b=None, dilation: int = 1, padding: int = 0) -> np.ndarray:
    """torch.nn.functional.conv1d for x [Cin, L], w [Cout, Cin, K] (stride 1)."""
    cout, cin, k = w.shape
    assert x.shape[0] == cin
    L = x.shape[1]
    xp = np.zeros((cin, L + 2 * padding), F32)
    xp[:, padding : padding + L] = x
    lout = L + 2 * padding - dilation * (k - 1)
    wt = np.ascontiguousarray(w.transpose(2, 0, 1))  # [K][Cout][Cin]: contiguous GEMM operands
    y = np.zeros((cout, lout), F32)
    tmp = np.empty((cout, lout), F32)
    for j in range(k):
        np.matmul(wt[j], xp[:, j * dilation : j * dilation + lout], out=tmp)
        y += tmp
    if b is not None:
        y += b.reshape(-1, 1)
    return y


def conv_transpose1d(x: np.ndarray, w: np.ndarray, b, stride: int, padding: int) -> np.ndarray:
    """torch ConvTranspose1d for x [Cin, L], w [Cin, Cout, K]."""
    cin, cout, k = w.shape
    L = x.shape[1]
    full = np.zeros((cout, (L - 1) * stride + k), F32)
    wt = np.ascontiguousarray(w.transpose(2, 1, 0))  # [K][Cout][Cin]
    for j in range(k):
        full[:, j : j + (L - 1) * stride + 1 : stride] += wt[j] @ x
    y = full[:, padding : full.shape[1] - padding]
    if b is not None:
        y = y + b.reshape(-1, 1)
    return np.ascontiguousarray(y, dtype=F32)


def layer_norm_channels(x: np.ndarray, gamma: np.ndarray, beta: np.ndarray, eps: float = 1e-4) -> np.ndarray:
    """glow_tts/layers.py:19-28 — normalise over the channel dim, biased variance."""
    mean = x.mean(axis=0, keepdims=True, dtype=F32)
    var = ((x - mean) ** 2).mean(axis=0, keepdims=True, dtype=F32)
    y = (x - mean) * (F32(1.0) / np.sqrt(var + F32(eps)))
    return (y * gamma.reshape(-1, 1) + beta.reshape(-1, 1)).astype(F32)


def leaky_relu(x: np.ndarray, slope: float) -> np.ndarray:
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


def sigmoid(x: np.ndarray) -> np.ndarray:
    return (F32(1.0) / (F32(1.0) + np.exp(-x))).astype(F32)


def softmax_last(x: np.ndarray) -> np.ndarray:
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(axis=-1, keepdims=True)).astype(F32)
