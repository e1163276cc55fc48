"""Randomised shape coverage of the conv kernels on the emulator build: hypothesis
draws channel counts, taps, dilations, lengths (down to a single column), batch rows
with ragged lengths and every tile shape; each result is checked against the numpy
oracle.  Derandomised (fixed example set) so CI is reproducible."""
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from oracle import nn_np

SETTINGS = dict(max_examples=30, deadline=None, derandomize=True,
                suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])

# (K, max dilation) pairs the reference configs need and the staged halo covers
# (hifi_gan ResBlock1/2: k 3,5,7,11 with d <= 5 / 12; GlowTTS: k 1,3,5 with d = 1)
TAPS = st.sampled_from([(1, 1), (3, 1), (3, 5), (5, 1), (5, 6), (7, 1), (7, 5), (7, 12), (11, 1), (11, 5)])


def check_conv1d(engine, taps, cin, cout, L, B, shape, slope, act, seed):
    K, dil = taps
    if shape >= 0:
        os.environ["MI355TTS_FORCE_TILE_DYNAMIC"] = str(shape)
    else:
        os.environ.pop("MI355TTS_FORCE_TILE_DYNAMIC", None)
    try:
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((B, cin, L)).astype(np.float32)
        w = (rng.standard_normal((cout, cin, K)) / np.sqrt(cin * K)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32) if seed % 3 else None
        lens = np.array([L] + [int(rng.integers(1, L + 1)) for _ in range(B - 1)], np.int32)
        y = engine.conv1d(x, w, b, dilation=dil, in_slope=slope, out_act=act, lens=lens)
    finally:
        os.environ.pop("MI355TTS_FORCE_TILE_DYNAMIC", None)
    for i in range(B):
        n = int(lens[i])
        ref = nn_np.conv1d(nn_np.leaky_relu(x[i, :, :n], slope), w, b, dilation=dil, padding=(K * dil - dil) // 2)
        if act == 1:
            ref = np.maximum(ref, 0)
        elif act == 2:
            ref = np.tanh(ref)
        np.testing.assert_allclose(y[i, :, :n], ref, rtol=1e-5, atol=3e-5)
        assert np.all(y[i, :, n:] == 0)  # writers skip the padded tail


@settings(**{**SETTINGS, "max_examples": 80})
@given(taps=TAPS, cin=st.integers(1, 72), cout=st.integers(1, 80), L=st.integers(1, 330), B=st.integers(1, 3),
       shape=st.sampled_from([-1, 0, 1, 2, 3]), slope=st.sampled_from([1.0, 0.1, 0.01]), act=st.sampled_from([0, 1, 2]),
       seed=st.integers(0, 2 ** 16))
def test_conv1d_random_shapes(emu_engine, taps, cin, cout, L, B, shape, slope, act, seed):
    check_conv1d(emu_engine, taps, cin, cout, L, B, shape, slope, act, seed)


@settings(**{**SETTINGS, "max_examples": 24})
@given(u=st.sampled_from([2, 4, 8]), cin=st.integers(2, 48), cout=st.integers(1, 40), L=st.integers(1, 150),
       shape=st.sampled_from([-1, 0, 1, 3]), seed=st.integers(0, 2 ** 16))
def test_conv_transpose1d_random_shapes(emu_engine, u, cin, cout, L, shape, seed):
    """HiFi-GAN upsamplers: kernel = 2 * stride, padding = stride / 2 (hifi_gan/models.py:159-172)."""
    K = 2 * u
    if shape >= 0:
        os.environ["MI355TTS_FORCE_TILE_DYNAMIC"] = str(shape)
    try:
        rng = np.random.default_rng(seed)
        x = rng.standard_normal((1, cin, L)).astype(np.float32)
        w = (rng.standard_normal((cin, cout, K)) / np.sqrt(2 * cin)).astype(np.float32)
        b = rng.standard_normal(cout).astype(np.float32)
        y = emu_engine.conv_transpose1d(x, w, b, stride=u, in_slope=0.1)
    finally:
        os.environ.pop("MI355TTS_FORCE_TILE_DYNAMIC", None)
    ref = nn_np.conv_transpose1d(nn_np.leaky_relu(x[0], 0.1), w, b, stride=u, padding=(K - u) // 2)
    assert y.shape[2] == ref.shape[1] == L * u
    np.testing.assert_allclose(y[0], ref, rtol=1e-5, atol=3e-5)
