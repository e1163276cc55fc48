"""CPU check of the conv kernel SOURCES (larynx_amd/csrc/conv_mfma.h) through the
C ABI, compiled against the fiber emulator in tests/hipemu — index arithmetic,
halo/mask handling, weight fragment packing and the MFMA C/D map.  The same
assertions run on the real GPU in tests/test_gpu_ops.py."""
import numpy as np
import pytest

from oracle import nn_np

CASES = [
    # Cin, Cout, K, dil, L, B, slope, act
    (16, 32, 3, 1, 100, 1, 1.0, 0),
    (24, 40, 3, 3, 300, 2, 0.1, 0),
    (32, 64, 7, 5, 260, 1, 0.1, 0),
    (8, 1, 7, 1, 513, 1, 0.01, 2),
    (80, 96, 1, 1, 77, 2, 1.0, 1),
    (20, 33, 5, 2, 129, 1, 1.0, 0),
    (16, 16, 11, 5, 400, 1, 0.1, 0),
    (40, 70, 7, 12, 300, 1, 0.1, 0),
]


@pytest.mark.parametrize("Cin,Cout,K,dil,L,B,slope,act", CASES)
def test_conv1d_matches_oracle(emu_engine, Cin, Cout, K, dil, L, B, slope, act):
    rng = np.random.default_rng(Cin * 1000 + Cout + K)
    x = rng.standard_normal((B, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    lens = np.array([L] + [L - 37] * (B - 1), np.int32)
    y = emu_engine.conv1d(x, w, b, dilation=dil, in_slope=slope, out_act=act, lens=lens)
    for i in range(B):
        n = lens[i]
        xi = nn_np.leaky_relu(x[i, :, :n], slope)
        ref = nn_np.conv1d(xi, w, b, dilation=dil, padding=(K * dil - dil) // 2)
        if act == 1:
            ref = np.maximum(ref, 0)
        elif act == 2:
            ref = np.tanh(ref)
        np.testing.assert_allclose(y[i, :, :n], ref, rtol=1e-5, atol=2e-5)
        assert np.all(y[i, :, n:] == 0)


@pytest.mark.parametrize("Cin,Cout,K,u,L", [(16, 8, 16, 8, 50), (32, 16, 4, 2, 131), (24, 12, 8, 4, 70), (64, 32, 16, 8, 130)])
def test_conv_transpose1d_matches_oracle(emu_engine, Cin, Cout, K, u, L):
    rng = np.random.default_rng(K * 100 + u)
    x = rng.standard_normal((1, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * 2)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    y = emu_engine.conv_transpose1d(x, w, b, stride=u, in_slope=0.1)
    ref = nn_np.conv_transpose1d(nn_np.leaky_relu(x[0], 0.1), w, b, stride=u, padding=(K - u) // 2)
    assert y.shape[2] == ref.shape[1] == L * u
    np.testing.assert_allclose(y[0], ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [0, 1, 2, 3])
@pytest.mark.parametrize("Cin,Cout,K,dil,L", [(24, 40, 3, 3, 300), (16, 16, 11, 5, 520), (40, 33, 1, 1, 290), (16, 72, 7, 1, 300),
                                              (72, 70, 11, 5, 300)])  # the last one: 64-row k=11 tiles (LDS weight ring at 64 columns)
def test_conv1d_every_tile_shape(emu_engine, monkeypatch, shape, Cin, Cout, K, dil, L):
    """The launcher picks the tile shape from the problem size; pin each one."""
    monkeypatch.setenv("MI355TTS_FORCE_TILE_DYNAMIC", str(shape))
    rng = np.random.default_rng(shape * 7 + K)
    x = rng.standard_normal((1, Cin, L)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, K)) / np.sqrt(Cin * K)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    y = emu_engine.conv1d(x, w, b, dilation=dil, in_slope=0.1)
    ref = nn_np.conv1d(nn_np.leaky_relu(x[0], 0.1), w, b, dilation=dil, padding=(K * dil - dil) // 2)
    np.testing.assert_allclose(y[0], ref, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("shape", [0, 1, 2, 3])
def test_conv_transpose1d_every_tile_shape(emu_engine, monkeypatch, shape):
    monkeypatch.setenv("MI355TTS_FORCE_TILE_DYNAMIC", str(shape))
    rng = np.random.default_rng(shape)
    x = rng.standard_normal((1, 24, 300)).astype(np.float32)
    w = (rng.standard_normal((24, 12, 16)) / 7).astype(np.float32)
    b = rng.standard_normal(12).astype(np.float32)
    y = emu_engine.conv_transpose1d(x, w, b, stride=8, in_slope=0.1)
    ref = nn_np.conv_transpose1d(nn_np.leaky_relu(x[0], 0.1), w, b, stride=8, padding=4)
    np.testing.assert_allclose(y[0], ref, rtol=1e-5, atol=2e-5)
