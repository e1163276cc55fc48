"""The native fp16 vocoder (MI355TTS_PRECISION_F16: csrc/conv_f16.h + csrc/hifigan_f16.h) on the CPU emulator, against the numpy
oracle in f32.  The mode trades accuracy for speed as the reference's `.half()` does (larynx/hifi_gan.py:96-97), so the bar is
a half-precision error band — an index slip (a wrong tap, octet, phase or chain) is an O(1) error, three orders above it.
Covers: both ResBlock types, 2 and 3 chains (single and grouped launches, both grouped tap sets), the fused conv1 + conv2
launches (pair_f16.h) at 128 / 64 / 16 / 8 channels and their two-launch form, the three tile shapes (128 / 64 / <= 32 rows),
ragged batches with per-row tails, the int16 tail, and the mode's error reporting."""
import numpy as np
import pytest

from larynx_amd import ffi
from larynx_amd import hparams as HP
from larynx_amd import synthetic
from oracle import audio_np, hifi_gan_np

RB2_THREE = HP.HifiGanHParams(
    resblock="2",
    upsample_rates=(4, 2),
    upsample_kernel_sizes=(8, 4),
    upsample_initial_channel=32,
    resblock_kernel_sizes=(3, 5, 7),
    resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
    num_mels=16,
)
# stages of 128 and 64 channels with the shipped (3, 7, 11) x (1, 3, 5) chains: the fused-pair kernel's WIDE and MID tiles
PAIR_WIDE = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256, num_mels=16)
CASES = {
    "pairs_128_64": PAIR_WIDE,
    "rb1_two_chains": HP.TINY_HIFIGAN,
    "rb2_two_chains": HP.TINY_HIFIGAN_RB2,
    "wide_and_mid_tiles": HP.TINY_HIFIGAN_PAIR,  # 64- and 32-channel stages, conv_pre 128 rows, upsamplers of 128 / 64 rows
    "grouped_11_7_3": HP.TINY_HIFIGAN_NARROW,    # 16- and 8-channel stages, the shipped (3, 7, 11) x (1, 3, 5) chains
    "grouped_7_5_3": RB2_THREE,
}


def _rel_rms(got, ref):
    return float(np.sqrt(np.mean((got - ref) ** 2)) / max(1e-9, np.sqrt(np.mean(ref ** 2))))


@pytest.mark.parametrize("case", sorted(CASES))
def test_f16_vocoder_matches_the_oracle_within_half_precision(emu_engine, case):
    hp = CASES[case]
    sd = synthetic.make_hifigan_state_dict(hp, seed=21)
    v = emu_engine.load_hifigan(hp, sd)
    try:
        assert emu_engine.set_precision(v, ffi.PRECISION_F16) == 0
        rng = np.random.default_rng(11)
        frames = np.array([41, 9, 300 if case == "grouped_11_7_3" else 70 if case == "pairs_128_64" else 23], np.int32)
        Fm = int(frames.max())
        melin = (rng.standard_normal((3, hp.num_mels, Fm)) * 2).astype(np.float32)
        mb = emu_engine.mel_from_numpy(melin, frames)
        emu_engine.profile_reset()
        f32, i16 = emu_engine.hifigan_infer(v, mb)
        counts = emu_engine.kernel_counts()
        assert counts.get("conv_f16_kernel", 0) > 0 and counts.get("post_f16_kernel", 0) == 1 and counts.get("pack_octets_kernel", 0) == 1
        fused = case in ("grouped_11_7_3", "pairs_128_64")  # ResBlock1 chains with the (3, 7, 11) taps: one launch per dilation step
        assert (counts.get("conv_f16_group_kernel", 0) > 0) == (case == "grouped_7_5_3")
        assert counts.get("pair_f16_group_kernel", 0) == (2 * 3 if fused else 0)
        for name in ("conv_mfma_kernel", "rb_conv_kernel", "conv_bf16_kernel", "mrf_small_kernel", "mrf8_kernel", "post_conv_kernel"):
            assert counts.get(name, 0) == 0, name  # every layer honours the switch
        hop = hp.hop
        for b in range(3):
            ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : frames[b]])
            n = frames[b] * hop
            assert _rel_rms(f32[b, :n], ref) < 1e-2, (b, _rel_rms(f32[b, :n], ref))
            assert np.all(f32[b, n:] == 0) and np.all(i16[b, n:] == 0)
            # the int16 tail normalises by the row's own peak: compare with the int16 of the row's own float output
            own16 = audio_np.audio_float_to_int16(f32[b, :n])
            assert np.abs(i16[b, :n].astype(np.int32) - own16.astype(np.int32)).max() <= 1
        # a row of the batch equals its solitary call (same tiles per row: the ragged grid deals a row its own tiles)
        solo = emu_engine.mel_from_numpy(melin[1:2, :, : frames[1]], frames[1:2])
        f1, _ = emu_engine.hifigan_infer(v, solo)
        n1 = frames[1] * hop
        np.testing.assert_array_equal(f1[0, :n1], f32[1, :n1])
        # back to f32: the exact mode again
        emu_engine.set_precision(v, ffi.PRECISION_F32)
        f32x, _ = emu_engine.hifigan_infer(v, solo)
        ref1 = hifi_gan_np.hifigan_infer(sd, hp, melin[1, :, : frames[1]])
        assert np.sqrt(np.mean((f32x[0, :n1] - ref1) ** 2)) < 1e-5
    finally:
        emu_engine.unload(v)


@pytest.mark.parametrize("which", ["narrow", "wide"])
def test_f16_fused_pairs_against_their_two_launch_form(emu_engine, which):
    """pair_f16.h keeps conv1's tile in LDS: same products, same f32 accumulation order, same roundings as conv1 -> plane ->
    conv2 — the fused launch must give the two-launch form's bits."""
    hp = HP.TINY_HIFIGAN_NARROW if which == "narrow" else PAIR_WIDE
    sd = synthetic.make_hifigan_state_dict(hp, seed=9)
    v = emu_engine.load_hifigan(hp, sd)
    try:
        emu_engine.set_precision(v, ffi.PRECISION_F16)
        rng = np.random.default_rng(4)
        frames = np.array([150 if which == "narrow" else 45, 7], np.int32)
        melin = (rng.standard_normal((2, hp.num_mels, int(frames.max()))) * 2).astype(np.float32)
        mb = emu_engine.mel_from_numpy(melin, frames)
        a, _ = emu_engine.hifigan_infer(v, mb)
        emu_engine.set_option("rb_pair", 0)
        try:
            emu_engine.profile_reset()
            b, _ = emu_engine.hifigan_infer(v, mb)
            counts = emu_engine.kernel_counts()
            assert counts.get("pair_f16_group_kernel", 0) == 0 and counts.get("conv_f16_group_kernel", 0) == 2 * 3 * 2
        finally:
            emu_engine.set_option("rb_pair", 1)
        np.testing.assert_array_equal(a, b)
    finally:
        emu_engine.unload(v)


def test_f16_ungrouped_schedule_gives_the_same_bits(emu_engine):
    hp = HP.TINY_HIFIGAN_NARROW
    sd = synthetic.make_hifigan_state_dict(hp, seed=5)
    v = emu_engine.load_hifigan(hp, sd)
    try:
        emu_engine.set_precision(v, ffi.PRECISION_F16)
        rng = np.random.default_rng(2)
        melin = (rng.standard_normal((1, hp.num_mels, 70)) * 2).astype(np.float32)
        mb = emu_engine.mel_from_numpy(melin)
        a, _ = emu_engine.hifigan_infer(v, mb)
        emu_engine.set_option("mrf_group", 0)
        try:
            emu_engine.profile_reset()
            b, _ = emu_engine.hifigan_infer(v, mb)
            assert emu_engine.kernel_counts().get("conv_f16_group_kernel", 0) == 0
        finally:
            emu_engine.set_option("mrf_group", 1)
        np.testing.assert_array_equal(a, b)
    finally:
        emu_engine.unload(v)


def test_f16_is_refused_with_a_reason_where_it_does_not_apply(emu_engine):
    # channel counts that are not whole octets
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=16, resblock_kernel_sizes=(3, 5),
                           resblock_dilation_sizes=((1, 3), (1, 2)), num_mels=16)
    v = emu_engine.load_hifigan(hp, synthetic.make_hifigan_state_dict(hp, seed=1))
    try:
        with pytest.raises(ffi.Mi355ttsError, match="multiples of 8"):
            emu_engine.set_precision(v, ffi.PRECISION_F16)
        emu_engine.set_precision(v, ffi.PRECISION_BF16X3)  # the accurate reduced mode still applies
    finally:
        emu_engine.unload(v)
    # GlowTTS: fp16 = the decoder's WaveNets (csrc/wn_f16.h; tests/test_emu_glow_f16.py); the split-bf16 requests are accepted and
    # reported as a no-op
    g = emu_engine.load_glow(HP.TINY_GLOW, synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=1))
    try:
        assert emu_engine.set_precision(g, ffi.PRECISION_F16) == 0
        assert emu_engine.set_precision(g, ffi.PRECISION_BF16X3) == ffi.PRECISION_NOOP
        assert emu_engine.set_precision(g, ffi.PRECISION_F32) == 0
    finally:
        emu_engine.unload(g)
