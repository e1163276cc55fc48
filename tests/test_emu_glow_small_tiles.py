"""The small-launch kernels of the GlowTTS path on the CPU emulator build, against the numpy oracle:

* `gate16_kernel` (csrc/gate16.h): the WaveNet gate conv (glow_tts/layers.py:138-162) on 16-row x 32-column tiles of
  v_mfma_f32_16x16x4_f32 — several channel-group counts, ragged batches, tile seams, and the 32-row tile of the same
  library (option `gate16` off) as a second opinion;
* `attention_mfma_kernel` (csrc/small_kernels.h): exact (dk = 2 NK) and clamped head sizes, one to many key blocks.
"""
import numpy as np
import pytest

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from oracle import glow_tts_np


def _run(engine, hp, seed, lens, noise_scale=0.0):
    sd = synthetic.make_glow_state_dict(hp, seed=seed)
    g = engine.load_glow(hp, sd)
    rng = np.random.default_rng(seed + 1)
    out = []
    try:
        for n in lens:
            ids = synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols)
            ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 1.0)
            mel = engine.glow_infer(g, ids, 0.0, 1.0)
            assert mel.frames[0] == ref.shape[1]
            out.append((ids, ref, mel.numpy("raw")[0]))
    finally:
        engine.unload(g)
    return out


@pytest.mark.parametrize("hidden,ksz", [(32, 5), (64, 5), (96, 3), (128, 5)])
def test_gate16_matches_the_oracle_and_the_32_row_tile(emu_engine, hidden, ksz):
    """hidden = 32 / 64 / 96 / 128: one to four 4-channel groups per k-group; 3 and 5 taps; decoder lengths on both sides
    of a 32-column tile seam."""
    hp = HP.GlowHParams(num_symbols=30, hidden_channels=hidden, filter_channels=32, filter_channels_dp=32, n_blocks_dec=2,
                        n_layers_enc=1, n_block_layers=2, mel_channels=8, kernel_size_dec=ksz)
    on = _run(emu_engine, hp, 41, (9, 40))
    emu_engine.set_option("gate16", 0)
    try:
        off = _run(emu_engine, hp, 41, (9, 40))
    finally:
        emu_engine.set_option("gate16", 1)
    for (_, ref, a), (_, _, b) in zip(on, off):
        np.testing.assert_allclose(a, ref, atol=5e-5, rtol=1e-4)
        np.testing.assert_allclose(b, ref, atol=5e-5, rtol=1e-4)
        assert np.abs(a - b).max() < 2e-5  # the same arithmetic up to summation order


def test_gate16_ragged_batch(emu_engine):
    """A padded batch of three rows: every row deals only its own tiles; padding columns stay zero."""
    hp = HP.GlowHParams(num_symbols=30, hidden_channels=64, filter_channels=32, filter_channels_dp=32, n_blocks_dec=2,
                        n_layers_enc=1, n_block_layers=2, mel_channels=8)
    sd = synthetic.make_glow_state_dict(hp, seed=43)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(44)
    lens = [37, 5, 18]
    ids = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in lens]
    mel = emu_engine.glow_infer(g, ids, 0.0, 1.0)
    raw = mel.numpy("raw")
    for b, n in enumerate(lens):
        ref = glow_tts_np.glow_tts_infer(sd, hp, ids[b], None, 0.0, 1.0)
        assert mel.frames[b] == ref.shape[1]
        np.testing.assert_allclose(raw[b][:, : ref.shape[1]], ref, atol=5e-5, rtol=1e-4)
    emu_engine.unload(g)


@pytest.mark.parametrize("hidden,heads", [(64, 2), (96, 2), (96, 4), (128, 1), (160, 2)])
def test_attention_head_sizes(emu_engine, hidden, heads):
    """dk = 32 (exact, NK = 16), 48, 24 and 80 (clamped channels under NK = 32 / 16 / 48) and 128 (four channel blocks, two key
    slices each); P = one key block, a ragged second one, and five."""
    hp = HP.GlowHParams(num_symbols=30, hidden_channels=hidden, n_heads=heads, filter_channels=32, filter_channels_dp=32,
                        n_blocks_dec=1, n_layers_enc=2, n_block_layers=1, mel_channels=8)
    for _, ref, got in _run(emu_engine, hp, 47, (7, 45, 150)):
        np.testing.assert_allclose(got, ref, atol=5e-5, rtol=1e-4)
