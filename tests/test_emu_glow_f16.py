"""The acoustic model's share of the `half` switch (MI355TTS_PRECISION_F16 on a GlowTTS model: csrc/wn_f16.h) on the CPU
emulator, against the numpy oracle in f32 and against the library's own f32 chain.

The reference's `half` is `.half()` on the whole FlowGenerator (larynx/glow_tts.py:90-91); here the decoder's WaveNets — every
gate conv and res_skip of a coupling block (glow_tts/layers.py:138-162) — run as ONE fp16 launch per block, everything else of
the acoustic model stays f32 (so the frame counts are the f32 model's).  The bar is a half-precision band: an index slip (a
wrong tap, octet, tile margin, res / skip half) is an O(1) error, two to three orders above it."""
import numpy as np
import pytest

from larynx_amd import ffi
from larynx_amd import hparams as HP
from larynx_amd import synthetic
from oracle import glow_tts_np


def _hp(hidden, layers, blocks=3, mel=8):
    return HP.GlowHParams(num_symbols=30, hidden_channels=hidden, filter_channels=32, filter_channels_dp=32, n_blocks_dec=blocks,
                          n_layers_enc=2, n_block_layers=layers, mel_channels=mel)


def _mels(engine, g, ids, noise_scale=0.0, seed=0):
    out = []
    for i in ids:
        mel = engine.glow_infer(g, i, noise_scale, 1.0, seed=seed)
        out.append(mel.numpy("raw")[0][:, : mel.frames[0]])
    return out


@pytest.mark.parametrize(
    "hidden,layers,lens",
    [
        (32, 2, (9, 23, 70)),    # one 32-row tile pair; 56 exact columns per tile: one, one and two column tiles
        (32, 1, (5, 40)),        # a single layer: no res_skip inside the launch, no skip plane
        (192, 4, (11, 60)),      # the released voices' shape: twelve row tiles on four waves, 48 exact columns per tile
        (32, 4, (130,)),         # four layers, many column tiles: every seam's recomputed margin
    ],
)
def test_f16_wavenets_match_the_oracle_within_half_precision(emu_engine, hidden, layers, lens):
    hp = _hp(hidden, layers)
    sd = synthetic.make_glow_state_dict(hp, seed=71)
    g = emu_engine.load_glow(hp, sd)
    try:
        rng = np.random.default_rng(72)
        ids = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in lens]
        exact = _mels(emu_engine, g, ids)
        assert emu_engine.set_precision(g, ffi.PRECISION_F16) == 0
        emu_engine.profile_reset()
        half = _mels(emu_engine, g, ids)
        counts = emu_engine.kernel_counts()
        # one launch per block and call; no gate conv, no separate res_skip of the earlier layers
        assert counts.get("wn_f16_kernel", 0) == hp.n_blocks_dec * len(lens), counts
        assert counts.get("gate16_kernel", 0) == 0 and counts.get("gate16_kernel.wide", 0) == 0
        assert counts.get("glow_tail_kernel", 0) == hp.n_blocks_dec * len(lens)
        for i, e, hmel in zip(ids, exact, half):
            ref = glow_tts_np.glow_tts_infer(sd, hp, i, None, 0.0, 1.0)
            assert hmel.shape == ref.shape == e.shape  # the encoder and the durations are the f32 model's
            scale = float(np.abs(ref).max())
            err = float(np.abs(hmel - ref).max())
            assert 0 < err < 2e-2 * max(1.0, scale), (err, scale)  # fp16 really ran, and inside its band
            assert float(np.sqrt(np.mean((hmel - ref) ** 2))) < 4e-3 * max(1.0, scale)
        # back to f32: the exact chain again, bit for bit
        assert emu_engine.set_precision(g, ffi.PRECISION_F32) == 0
        again = _mels(emu_engine, g, ids)
        for e, a in zip(exact, again):
            np.testing.assert_array_equal(e, a)
    finally:
        emu_engine.unload(g)


def test_f16_ragged_batch_rows_equal_their_solitary_calls(emu_engine):
    """A padded batch: every row deals its own column tiles (len[b]), so a row inside the batch equals its batch-1 call bit
    for bit, and the padded tails stay exactly 0."""
    hp = _hp(32, 2, blocks=2)
    sd = synthetic.make_glow_state_dict(hp, seed=73)
    g = emu_engine.load_glow(hp, sd)
    try:
        assert emu_engine.set_precision(g, ffi.PRECISION_F16) == 0
        rng = np.random.default_rng(74)
        lens = (37, 5, 61)
        ids = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in lens]
        mel = emu_engine.glow_infer(g, ids, 0.0, 1.0)
        raw = mel.numpy("raw")
        solo = _mels(emu_engine, g, ids)
        for b in range(len(lens)):
            np.testing.assert_array_equal(raw[b][:, : mel.frames[b]], solo[b])
            assert np.all(raw[b][:, mel.frames[b]:] == 0)
    finally:
        emu_engine.unload(g)


def test_f16_request_is_a_reported_noop_where_the_kernel_does_not_cover_the_geometry(emu_engine):
    """hidden_channels other than 192 / 32 (and the split-bf16 requests on any GlowTTS model): a distinct positive status, the model
    keeps computing in f32."""
    hp = _hp(64, 2, blocks=2)
    sd = synthetic.make_glow_state_dict(hp, seed=75)
    g = emu_engine.load_glow(hp, sd)
    try:
        assert emu_engine.set_precision(g, ffi.PRECISION_F16) == ffi.PRECISION_NOOP
        assert emu_engine.set_precision(g, ffi.PRECISION_BF16X3) == ffi.PRECISION_NOOP
        assert emu_engine.set_precision(g, ffi.PRECISION_F32) == 0
        ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(1), 12, hp.num_symbols)
        emu_engine.profile_reset()
        emu_engine.glow_infer(g, ids, 0.0, 1.0)
        assert emu_engine.kernel_counts().get("wn_f16_kernel", 0) == 0
    finally:
        emu_engine.unload(g)
    # speaker-conditioned WaveNets (cond_layer on every gate conv: glow_tts/layers.py:144-154) are not covered either
    import dataclasses

    hpm = dataclasses.replace(HP.TINY_GLOW, n_speakers=3, gin_channels=20)
    gm = emu_engine.load_glow(hpm, synthetic.make_glow_state_dict(hpm, seed=75))
    try:
        assert emu_engine.set_precision(gm, ffi.PRECISION_F16) == ffi.PRECISION_NOOP
    finally:
        emu_engine.unload(gm)
    hp2 = _hp(32, 2, blocks=2)
    g2 = emu_engine.load_glow(hp2, synthetic.make_glow_state_dict(hp2, seed=75))
    try:
        assert emu_engine.set_precision(g2, ffi.PRECISION_BF16X3) == ffi.PRECISION_NOOP
        assert emu_engine.set_precision(g2, ffi.PRECISION_F16) == 0
    finally:
        emu_engine.unload(g2)
