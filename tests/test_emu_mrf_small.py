"""The one-launch MRF stage kernel (csrc/mrf_small.h: all three ResBlock1 chains of a 16- / 8-channel stage and their
average on an LDS-resident tile, v_mfma_f32_16x16x4_f32) on the CPU emulator build, against the numpy oracle and
against the un-fused schedule of the same library.  Reference: hifi_gan/models.py:91-98,186-202."""
import numpy as np
import pytest

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from oracle import audio_np, hifi_gan_np


@pytest.fixture(scope="module")
def narrow(emu_engine):
    hp = HP.TINY_HIFIGAN_NARROW
    sd = synthetic.make_hifigan_state_dict(hp, seed=11)
    return hp, sd, emu_engine.load_hifigan(hp, sd)


def test_narrow_stages_match_the_oracle_across_tile_edges(emu_engine, narrow):
    """Rows of 70 and 150 frames: the 16-channel stage is 140 / 300 columns (one and two 256-column tiles), the
    8-channel stage 280 / 600 (two and three) — tile seams, a ragged batch and sequence ends inside a tile."""
    hp, sd, v = narrow
    rng = np.random.default_rng(21)
    frames = np.array([150, 70], np.int32)
    melin = (rng.standard_normal((2, hp.num_mels, 150)) * 2).astype(np.float32)
    mb = emu_engine.mel_from_numpy(melin, frames)
    f32, i16 = emu_engine.hifigan_infer(v, mb)
    emu_engine.set_option("mrf_small", 0)
    try:
        g32, _ = emu_engine.hifigan_infer(v, mb)
    finally:
        emu_engine.set_option("mrf_small", 1)
    for b in range(2):
        ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : frames[b]])
        n = frames[b] * hp.hop
        assert ref.shape[0] == n
        err = f32[b, :n] - ref
        assert np.sqrt(np.mean(err ** 2)) < 1e-5, np.abs(err).max()
        assert np.abs(err).max() < 1e-4
        assert np.all(f32[b, n:] == 0) and np.all(i16[b, n:] == 0)
        ref16 = audio_np.audio_float_to_int16(ref)
        assert np.abs(i16[b, :n].astype(np.int32) - ref16.astype(np.int32)).max() <= 1
        # the fused stage and the conv-by-conv schedule are the same arithmetic up to summation order
        assert np.abs(f32[b, :n] - g32[b, :n]).max() < 2e-5


def test_single_short_utterance(emu_engine, narrow):
    hp, sd, v = narrow
    rng = np.random.default_rng(22)
    melin = (rng.standard_normal((1, hp.num_mels, 9)) * 2).astype(np.float32)
    f32, _ = emu_engine.hifigan_infer(v, emu_engine.mel_from_numpy(melin))
    ref = hifi_gan_np.hifigan_infer(sd, hp, melin[0])
    assert np.sqrt(np.mean((f32[0, : ref.shape[0]] - ref) ** 2)) < 1e-5


def test_two_dilation_steps_and_the_fallback_for_other_taps(emu_engine):
    """Chains of two dilation steps run on the fused kernels (`nsteps` is a run-time value); tap sets other than (3, 7, 11) —
    and receptive fields wider than the staged halo — fall back to the conv-by-conv schedule.  Both against the oracle."""
    rng = np.random.default_rng(31)
    for hp in (
        HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=32, num_mels=16,
                          resblock_dilation_sizes=((1, 5), (3, 1), (2, 4))),
        HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=32, num_mels=16,
                          resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12))),
        HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=32, num_mels=16,
                          resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 5, 5))),  # k = 11 chain: half-width 70 > 64
    ):
        sd = synthetic.make_hifigan_state_dict(hp, seed=12)
        v = emu_engine.load_hifigan(hp, sd)
        melin = (rng.standard_normal((1, hp.num_mels, 40)) * 2).astype(np.float32)
        f32, _ = emu_engine.hifigan_infer(v, emu_engine.mel_from_numpy(melin))
        ref = hifi_gan_np.hifigan_infer(sd, hp, melin[0])
        assert np.sqrt(np.mean((f32[0, : ref.shape[0]] - ref) ** 2)) < 1e-5
        emu_engine.unload(v)
