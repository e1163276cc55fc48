"""The throughput form of the GlowTTS decoder's WaveNet layers (csrc/wn_layer.h: gate conv + gate + res_skip conv of
glow_tts/layers.py:138-162 as ONE column-owner launch per layer) on the CPU emulator build.

The kernel promises the SAME BITS as the launches it replaces (gate16_kernel + lin16_kernel: same packed fragments, same MFMA
chains, partial sums folded in the same order) — that is what lets the host pick the form by load.  So the bar here is
`np.array_equal` between option `wn_layer` = 2 (always) and 0 (never), next to the oracle check, for: lengths on both sides of
the 16-column tile seams, a ragged batch, a multi-speaker voice (the `cond` offsets), one-layer blocks (gate-only form only),
and the automatic rule of option 1 (wide passes, calls in flight).  The option is OFF by default: on the device the column owners
lose to the launches they replace under every load measured (profiles/r05_wn_layer_ab.txt)."""
import dataclasses
import threading

import numpy as np
import pytest

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from oracle import glow_tts_np


def _hp(**kw):
    base = dict(num_symbols=30, hidden_channels=192, filter_channels=32, filter_channels_dp=32, n_blocks_dec=2, n_layers_enc=1,
                n_block_layers=3, mel_channels=8)
    base.update(kw)
    return HP.GlowHParams(**base)


def _mels(engine, g, rows, batch, **kw):
    out = []
    if batch:
        mel = engine.glow_infer(g, rows, kw.pop("noise_scale", 0.0), kw.pop("length_scale", 1.0), **kw)
        raw = mel.numpy("raw")
        out = [raw[b][:, : mel.frames[b]].copy() for b in range(len(rows))]
    else:
        for i, ids in enumerate(rows):
            k = dict(kw)
            if "speaker_ids" in k:
                k["speaker_ids"] = k["speaker_ids"][i]
            mel = engine.glow_infer(g, ids, k.pop("noise_scale", 0.0), k.pop("length_scale", 1.0), **k)
            out.append(mel.numpy("raw")[0][:, : mel.frames[0]].copy())
    return out


def _counts(engine, fn):
    engine.profile_reset()
    r = fn()
    return r, engine.kernel_counts()


def _forms(engine, g, rows, batch, **kw):
    """(bits with the column-owner launches, bits with the 16-row tiles), each with its kernel counts."""
    engine.set_option("wn_layer", 2)
    try:
        on, c_on = _counts(engine, lambda: _mels(engine, g, rows, batch, **dict(kw)))
        engine.set_option("wn_layer", 0)
        off, c_off = _counts(engine, lambda: _mels(engine, g, rows, batch, **dict(kw)))
    finally:
        engine.set_option("wn_layer", 0)
    return on, c_on, off, c_off


def test_same_bits_as_the_launches_it_replaces_and_the_oracle(emu_engine):
    hp = _hp()
    sd = synthetic.make_glow_state_dict(hp, seed=101)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(102)
    # decoder columns = frames / 2: one tile, a tile seam (16), a partial last tile
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (3, 7, 11)]
    try:
        on, c_on, off, c_off = _forms(emu_engine, g, rows, batch=False)
        calls, blocks, layers = len(rows), hp.n_blocks_dec, hp.n_block_layers
        assert c_on["wn_layer_kernel"] == calls * blocks * (layers - 1) and c_on["wn_layer_kernel.gate_only"] == calls * blocks
        assert c_on["gate16_kernel"] == 0 and c_on["glow_tail_kernel"] == calls * blocks
        assert c_off["wn_layer_kernel"] == 0 and c_off["wn_layer_kernel.gate_only"] == 0
        assert c_off["gate16_kernel"] == calls * blocks * layers
        for ids, a, b in zip(rows, on, off):
            ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 1.0)
            assert a.shape == ref.shape
            np.testing.assert_allclose(a, ref, atol=5e-5, rtol=1e-4)
            assert np.array_equal(a, b)
    finally:
        emu_engine.unload(g)


def test_ragged_batch_and_noise(emu_engine):
    """A padded batch (every row owns only its own column tiles; the device RNG's noise field) in both forms; a row of the
    batch equals its own batch-1 call in the column-owner form too."""
    hp = _hp(n_blocks_dec=1, n_block_layers=2)
    sd = synthetic.make_glow_state_dict(hp, seed=111)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(112)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (9, 2, 5)]
    try:
        on, c_on, off, _ = _forms(emu_engine, g, rows, batch=True, noise_scale=0.667, length_scale=1.3, seed=7)
        assert c_on["wn_layer_kernel"] == 1 and c_on["wn_layer_kernel.gate_only"] == 1
        for a, b in zip(on, off):
            assert np.array_equal(a, b)
        emu_engine.set_option("wn_layer", 2)
        try:
            for b, ids in enumerate(rows):
                mel = emu_engine.glow_infer(g, ids, 0.667, 1.3, seed=7 + b)
                assert np.array_equal(mel.numpy("raw")[0][:, : mel.frames[0]], on[b])
        finally:
            emu_engine.set_option("wn_layer", 0)
    finally:
        emu_engine.unload(g)


def test_speaker_conditioning(emu_engine):
    hp = _hp(n_blocks_dec=1, n_block_layers=2, n_speakers=3, gin_channels=8)
    sd = synthetic.make_glow_state_dict(hp, seed=121)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(122)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (6, 10)]
    try:
        on, c_on, off, _ = _forms(emu_engine, g, rows, batch=True, speaker_ids=[2, 0])
        assert c_on["wn_layer_kernel"] == 1
        for ids, spk, a, b in zip(rows, (2, 0), on, off):
            ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 1.0, speaker_id=spk)
            np.testing.assert_allclose(a, ref, atol=5e-5, rtol=1e-4)
            assert np.array_equal(a, b)
    finally:
        emu_engine.unload(g)


def test_shapes_outside_the_kernel_keep_the_16_row_tiles(emu_engine):
    """Only the released voices' width (192 channels, k = 5 or 3) has the column-owner form; other widths run the launches
    of gate16.h whatever the option says."""
    hp = _hp(hidden_channels=64, n_blocks_dec=1, n_block_layers=2)
    sd = synthetic.make_glow_state_dict(hp, seed=131)
    g = emu_engine.load_glow(hp, sd)
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(132), 7, hp.num_symbols)
    try:
        on, c_on, off, _ = _forms(emu_engine, g, [ids], batch=False)
        assert c_on["wn_layer_kernel"] == 0 and c_on["gate16_kernel"] == 2
        assert np.array_equal(on[0], off[0])
    finally:
        emu_engine.unload(g)


def test_automatic_rule(emu_engine):
    """Option 1: a lone short call keeps the latency form, a pass with at least `wn_layer_min_tiles` 16-column
    tiles takes the column owners, and so does a call that starts while another one holds a worker."""
    hp = _hp(n_blocks_dec=1, n_block_layers=2)
    sd = synthetic.make_glow_state_dict(hp, seed=141)
    g = emu_engine.load_glow(hp, sd)
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(142), 9, hp.num_symbols)
    emu_engine.set_option("wn_layer", 1)
    try:
        lone, c = _counts(emu_engine, lambda: _mels(emu_engine, g, [ids], False))
        assert c["wn_layer_kernel"] == 0 and c["gate16_kernel"] == 2
        emu_engine.set_option("wn_layer_min_tiles", 2)
        try:
            wide, c = _counts(emu_engine, lambda: _mels(emu_engine, g, [ids], False))
        finally:
            emu_engine.set_option("wn_layer_min_tiles", 48)
        assert c["wn_layer_kernel"] == 1 and c["gate16_kernel"] == 0
        assert np.array_equal(lone[0], wide[0])
        # two threads: whichever call starts second sees the first one's worker checked out
        emu_engine.profile_reset()
        out = [None, None]
        gate = threading.Barrier(2)

        def run(i):
            gate.wait()
            for _ in range(3):
                out[i] = _mels(emu_engine, g, [ids], False)[0]

        ts = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        c = emu_engine.kernel_counts()
        assert c["wn_layer_kernel"] + c["gate16_kernel"] // 2 == 6  # every call took one form or the other
        assert c["wn_layer_kernel"] >= 1  # and the overlapped ones the column owners
        assert np.array_equal(out[0], lone[0]) and np.array_equal(out[1], lone[0])
    finally:
        emu_engine.set_option("wn_layer", 0)
        emu_engine.unload(g)


def test_wide_gate_tile_computes_the_same_bits(emu_engine):
    """`gate16_kernel<K, J, 2>` / `lin16_kernel<1, 6, 2, false, 4>` (option `gate16_wide` = the pass size from which they are
    used): two / four row tiles per workgroup from one staged input tile — a row tile's arithmetic does not depend on it, so
    the bits are the 16-row launch's; a ragged batch and a multi-speaker voice (the `cond` offsets)."""
    hp = _hp(n_blocks_dec=1, n_block_layers=2, n_speakers=3, gin_channels=8)
    sd = synthetic.make_glow_state_dict(hp, seed=151)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(152)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (11, 4, 8)]
    try:
        got = {}
        for wide in (1, 0):
            emu_engine.set_option("gate16_wide", wide)
            try:
                got[wide] = _counts(emu_engine, lambda: _mels(emu_engine, g, rows, True, speaker_ids=[1, 2, 0]))
            finally:
                emu_engine.set_option("gate16_wide", 512)
        assert got[1][1]["gate16_kernel.wide"] == 2 and got[1][1]["gate16_kernel"] == 0
        assert got[0][1]["gate16_kernel.wide"] == 0 and got[0][1]["gate16_kernel"] == 2
        # ... and the 1 x 1 convs with whole groups of four row tiles (res_skip: 2H rows, qkv: 3H) four row tiles per workgroup
        assert got[1][1]["lin16_kernel.wide"] >= 2 and got[0][1]["lin16_kernel.wide"] == 0
        for a, b in zip(got[1][0], got[0][0]):
            assert np.array_equal(a, b)
    finally:
        emu_engine.unload(g)
