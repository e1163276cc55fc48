"""Host-runtime features of the C ABI on the CPU emulator build: the fused
`mi355tts_synthesize` call, device-side pause padding, `mi355tts_reserve`, the
load-adaptive vocoder schedule, unusual-but-legal vocoder configurations and
hyper-parameter validation (round-1 advisor findings)."""
import threading

import numpy as np
import pytest

from larynx_amd import ffi
from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from oracle import audio_np, glow_tts_np, hifi_gan_np


@pytest.fixture(scope="module")
def tiny(emu_engine):
    gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7)
    vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7)
    return dict(gsd=gsd, vsd=vsd, g=emu_engine.load_glow(HP.TINY_GLOW, gsd), v=emu_engine.load_hifigan(HP.TINY_HIFIGAN, vsd))


def check_synthesize_equals_two_calls(eng, g, v, num_symbols, hop, lens=(13, 7, 21)):
    rng = np.random.default_rng(41)
    s = ljspeech_audio_settings()
    rows = [synthetic.synthetic_phoneme_ids(rng, n, num_symbols) for n in lens]
    for ids in (rows[0], rows):
        mel = eng.glow_infer(g, ids, 0.667, 1.0, seed=99, audio_settings=s)
        f2, i2 = eng.hifigan_infer(v, mel, pad_before=5, pad_after=9)
        frames, f1, i1 = eng.synthesize(g, v, ids, 0.667, 1.0, seed=99, audio_settings=s, pad_before=5, pad_after=9, want_float=True)
        assert np.array_equal(frames, mel.frames)
        assert np.array_equal(i1, i2) and np.array_equal(f1, f2)
        for b in range(len(frames)):
            n = int(frames[b]) * hop
            assert np.all(i1[b, :5] == 0) and np.all(i1[b, 5 + n :] == 0) and np.all(f1[b, :5] == 0) and np.all(f1[b, 5 + n :] == 0)
        # the padded row is the un-padded audio shifted by pad_before
        _, i0 = eng.hifigan_infer(v, mel)
        for b in range(len(frames)):
            n = int(frames[b]) * hop
            assert np.array_equal(i1[b, 5 : 5 + n], i0[b, :n])
    # a guess that is too small is retried with a bigger buffer
    frames, _, i3 = eng.synthesize(g, v, rows[0], 0.667, 1.0, seed=99, audio_settings=s, pad_before=5, pad_after=9, frames_per_id_guess=0.2)
    assert np.array_equal(i3, eng.synthesize(g, v, rows[0], 0.667, 1.0, seed=99, audio_settings=s, pad_before=5, pad_after=9)[2])


def check_output_tail_forms(eng, g, v, num_symbols, hop, lens=(13, 7, 21), device_buffers=None):
    """csrc/voc_out.h (option `voc_out`, default 1): conv_post + tanh + the rows' peaks as ONE dedicated launch and the
    delivery of the rows (pause before | samples | zeros) as one more, against round 4's ten launches: float rows within
    f32 round-off (conv_post's fmaf chains vs the padded MFMA tile), int16 within 1 LSB, the zero regions exactly zero — host
    and device destinations, a ragged batch, with and without the float output, and behind the denoiser."""
    rng = np.random.default_rng(43)
    s = ljspeech_audio_settings()
    rows = [synthetic.synthetic_phoneme_ids(rng, n, num_symbols) for n in lens]
    for ids in (rows[0], rows):
        mel = eng.glow_infer(g, ids, 0.667, 1.0, seed=17, audio_settings=s)
        B = len(mel.frames)
        got = {}
        for form in (1, 0):
            eng.set_option("voc_out", form)
            try:
                eng.profile_reset()
                f, i = eng.hifigan_infer(v, mel, pad_before=3, pad_after=6)
                names = eng.kernel_counts()
                assert names["post_conv_kernel"] == form and names["wave_out_kernel"] == form
                _, ionly = eng.hifigan_infer(v, mel, want_float=False, pad_before=3, pad_after=6)
                fonly, _ = eng.hifigan_infer(v, mel, want_int16=False)
                n = mel.max_frames * hop + 16
                df, di = np.full((B, n), 7.0, np.float32), np.full((B, n), 7, np.int16)
                if device_buffers is None:
                    eng.hifigan_infer_raw(v, mel, df.ctypes.data, di.ctypes.data, n, flags=ffi.OUT_DEVICE, pad_before=3, pad_after=6)
                else:
                    df, di = device_buffers(eng, v, mel, n, 3, 6)
                got[form] = (f, i, ionly, fonly, df, di)
            finally:
                eng.set_option("voc_out", 1)
        a, b = got[1], got[0]
        assert np.abs(a[0] - b[0]).max() <= 2e-6 and np.abs(a[3] - b[3]).max() <= 2e-6 and np.abs(a[4] - b[4]).max() <= 2e-6
        for k in (1, 2, 5):
            assert np.abs(a[k].astype(np.int32) - b[k].astype(np.int32)).max() <= 1
        assert np.array_equal(a[1], a[2])  # with or without the float output
        for r in range(B):
            nr = int(mel.frames[r]) * hop
            for arr in (a[0], a[1], a[4], a[5]):
                assert np.all(arr[r, :3] == 0) and np.all(arr[r, 3 + nr :] == 0)
            assert np.array_equal(a[4][r, : 3 + nr], a[0][r, : 3 + nr]) and np.array_equal(a[5][r, : 3 + nr], a[1][r, : 3 + nr])
            assert np.all(a[3][r, nr:] == 0) and np.array_equal(a[3][r, :nr], a[0][r, 3 : 3 + nr])
        mel.free()


def test_output_tail_forms(emu_engine, tiny):
    check_output_tail_forms(emu_engine, tiny["g"], tiny["v"], HP.TINY_GLOW.num_symbols, HP.TINY_HIFIGAN.hop)


def test_synthesize_equals_two_calls(emu_engine, tiny):
    check_synthesize_equals_two_calls(emu_engine, tiny["g"], tiny["v"], HP.TINY_GLOW.num_symbols, HP.TINY_HIFIGAN.hop)


def test_synthesize_matches_oracle_with_denoiser_free_path(emu_engine, tiny):
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(5), 15, HP.TINY_GLOW.num_symbols)
    s = ljspeech_audio_settings()
    frames, f32, i16 = emu_engine.synthesize(tiny["g"], tiny["v"], ids, 0.0, 1.0, audio_settings=s, want_float=True)
    ref_mel = glow_tts_np.glow_tts_infer(tiny["gsd"], HP.TINY_GLOW, ids, None, 0.0, 1.0)
    ref = hifi_gan_np.hifigan_infer(tiny["vsd"], HP.TINY_HIFIGAN, audio_np.mel_to_vocoder_input(ref_mel, s))
    assert int(frames[0]) == ref_mel.shape[1]
    assert np.sqrt(np.mean((f32[0] - ref) ** 2)) < 1e-4
    assert np.abs(i16[0].astype(np.int32) - audio_np.audio_float_to_int16(ref).astype(np.int32)).max() <= 1


def test_reserve_then_calls_do_not_grow(emu_engine, tiny):
    emu_engine.reserve(3, tiny["g"], tiny["v"], max_batch=2, max_ids=40, max_frames=400, denoiser=False, max_pad_samples=64)
    with pytest.raises(ffi.Mi355ttsError):
        emu_engine.reserve(0, tiny["g"], tiny["v"])
    with pytest.raises(ffi.Mi355ttsError):
        emu_engine.reserve(1, 12345, tiny["v"])
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(2), 30, HP.TINY_GLOW.num_symbols)
    a = emu_engine.synthesize(tiny["g"], tiny["v"], ids, 0.0, 1.0)[2]
    b = emu_engine.synthesize(tiny["g"], tiny["v"], ids, 0.0, 1.0)[2]
    assert np.array_equal(a, b)
    # mi355tts_reserve also measured which worker streams share a hardware queue (mi355tts_worker_queue_groups); on the emulator a
    # launch completes inside the launch call, so no stream ever waits behind another: every probed worker is a group of its own,
    # the groups are numbered 0 .. n - 1, and calls keep working (the free worker of the least busy group is taken)
    groups = emu_engine.worker_queue_groups()
    probed = [g for g in groups if g >= 0]
    assert len(groups) >= 3 and len(probed) >= 3 and sorted(probed) == list(range(len(probed))), groups
    c = emu_engine.synthesize(tiny["g"], tiny["v"], ids, 0.0, 1.0)[2]
    assert np.array_equal(a, c)


def check_schedule_invariance(eng, v, num_mels, frames=33, threads=4):
    """Grouped launches of the MRF chains (default), the members one by one while other calls are in flight
    (`adaptive_schedule`), the round-1 fork onto side streams (`mrf_group` = 0): results must not depend on
    which form ran, nor on how many calls are in flight."""
    rng = np.random.default_rng(17)
    melin = (rng.standard_normal((1, num_mels, frames)) * 2).astype(np.float32)
    outs = {}
    try:
        for adaptive, group in ((0, 1), (1, 1), (0, 0), (1, 0)):
            eng.set_option("adaptive_schedule", adaptive)
            eng.set_option("mrf_group", group)
            outs[(adaptive, group)] = eng.hifigan_infer(v, eng.mel_from_numpy(melin))[0]
    finally:
        eng.set_option("adaptive_schedule", 0)  # the defaults
        eng.set_option("mrf_group", 1)
    outs[0] = outs[(0, 1)]
    for k, o in outs.items():
        assert np.array_equal(o, outs[0]), k
    got = [None] * threads
    gate = threading.Barrier(threads)

    def work(i):
        gate.wait()
        for _ in range(3):
            got[i] = eng.hifigan_infer(v, eng.mel_from_numpy(melin))[0]

    ts = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for g in got:
        assert np.array_equal(g, outs[0])


def test_adaptive_schedule_results_do_not_depend_on_load(emu_engine, tiny):
    check_schedule_invariance(emu_engine, tiny["v"], HP.TINY_HIFIGAN.num_mels)


def test_vocoder_with_four_resblock_kernels_and_odd_strides(emu_engine):
    """num_kernels = 4 (> the 3 side-stream chains: runs the serial MRF form; round-1 advisor:
    `outs[3]` overflow) and upsample rates (2, 2) with an odd frame count (stage lengths that
    are not multiples of 4: row strides are padded, round-1 advisor)."""
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=32,
                           resblock_kernel_sizes=(3, 5, 7, 3), resblock_dilation_sizes=((1, 2), (1, 3), (1, 1), (2, 1)), num_mels=16)
    sd = synthetic.make_hifigan_state_dict(hp, seed=51)
    v = emu_engine.load_hifigan(hp, sd)
    rng = np.random.default_rng(52)
    for F in (7, 13):
        melin = (rng.standard_normal((2, hp.num_mels, F)) * 2).astype(np.float32)
        frames = np.array([F, F - 2], np.int32)
        f32, _ = emu_engine.hifigan_infer(v, emu_engine.mel_from_numpy(melin, frames))
        for b in range(2):
            ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : frames[b]])
            n = frames[b] * hp.hop
            assert np.sqrt(np.mean((f32[b, :n] - ref) ** 2)) < 1e-5
            assert np.all(f32[b, n:] == 0)
    emu_engine.unload(v)
    hp3 = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=32,
                            resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (1, 3), (1, 1)), num_mels=16)
    sd3 = synthetic.make_hifigan_state_dict(hp3, seed=53)
    v3 = emu_engine.load_hifigan(hp3, sd3)
    melin = (rng.standard_normal((1, 16, 9)) * 2).astype(np.float32)
    f32, _ = emu_engine.hifigan_infer(v3, emu_engine.mel_from_numpy(melin))
    assert np.sqrt(np.mean((f32[0] - hifi_gan_np.hifigan_infer(sd3, hp3, melin[0])) ** 2)) < 1e-5
    emu_engine.unload(v3)


def test_bad_hparams_return_errors(emu_engine):
    """`n_split = 0` used to divide by zero inside the validator (round-1 advisor)."""
    import dataclasses

    for bad in (dict(n_split=0), dict(n_sqz=0), dict(n_split=3), dict(n_block_layers=0)):
        hp = dataclasses.replace(HP.TINY_GLOW, **bad)
        with pytest.raises(ffi.Mi355ttsError):
            ffi.manifest(emu_engine.lib, ffi.glow_hparams_c(hp))


def test_long_input_durations_scan(emu_engine):
    """More ids than the 64 lanes of the duration kernel's scan, ragged batch."""
    hp = HP.GlowHParams(num_symbols=30, hidden_channels=32, filter_channels=32, filter_channels_dp=32, n_blocks_dec=1,
                        n_layers_enc=1, n_block_layers=1, mel_channels=8)
    sd = synthetic.make_glow_state_dict(hp, seed=61)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(62)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (131, 64, 65, 1)]
    mel = emu_engine.glow_infer(g, rows, 0.0, 0.7)
    for b, ids in enumerate(rows):
        ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 0.7)
        assert int(mel.frames[b]) == ref.shape[1]
        np.testing.assert_allclose(mel.numpy("raw")[b, :, : ref.shape[1]], ref, atol=3e-5, rtol=1e-4)
    emu_engine.unload(g)


def test_fresh_seed_per_call_by_default(emu_library, tmp_path):
    """Without an explicit `seed` every `phonemes_to_mels` call draws new noise, as the
    reference's `torch.randn_like` does (round-1 advisor); an explicit seed repeats."""
    import json

    import larynx_amd
    from larynx_amd.constants import TextToSpeechType

    gdir = tmp_path / "tiny-glow_tts"
    gdir.mkdir()
    (gdir / "config.json").write_text(json.dumps(HP.TINY_GLOW.to_config()))
    np.savez(gdir / "generator.npz", **synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=3))
    tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, library_path=emu_library)
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(2), 10, HP.TINY_GLOW.num_symbols)
    a, b = np.asarray(tts.phonemes_to_mels(ids)), np.asarray(tts.phonemes_to_mels(ids))
    assert a.shape == b.shape and not np.array_equal(a, b)
    c, d = np.asarray(tts.phonemes_to_mels(ids, {"seed": 5})), np.asarray(tts.phonemes_to_mels(ids, {"seed": 5}))
    assert np.array_equal(c, d)


def check_seeded_path_equals_injected_noise(eng, g, num_symbols, channels, lens=(120,), seed=4321, length_scale=0.65):
    """The TIMED noise mode against the PARITY-CHECKED one (both stand in for glow_tts/models.py:348, `torch.randn_like`):
    `glow_infer(seed = s)` — the generator fused into expand_noise_squeeze_kernel, row b on the stream s + b — must be, bit for
    bit, `glow_infer(noise = gauss_noise(s, B, M, T))`, the same field generated by itself and injected like a golden's recorded
    noise.  An indexing slip between the generator and its fused consumer (channel / frame / row keys) would pass the
    distribution test and fail here."""
    rng = np.random.default_rng(seed)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, num_symbols) for n in lens]
    x = rows[0] if len(rows) == 1 else rows
    seeded = eng.glow_infer(g, x, 0.667, length_scale, seed=seed)
    frames = [int(f) for f in seeded.frames]
    T = max(frames) + 7  # (a wider field than needed: the row stride of the injected noise is not the frame count)
    field = eng.gauss_noise(seed, len(rows), channels, T)
    injected = eng.glow_infer(g, x, 0.667, length_scale, noise=field)
    try:
        assert [int(f) for f in injected.frames] == frames and min(frames) > 0
        a, b = seeded.numpy("raw"), injected.numpy("raw")
        assert np.array_equal(a, b)
        # and the noise is really there: another seed gives another mel
        other = eng.glow_infer(g, x, 0.667, length_scale, seed=seed + 1000)
        assert [int(f) for f in other.frames] == frames and not np.array_equal(other.numpy("raw"), a)
        other.free()
    finally:
        seeded.free()
        injected.free()


def test_seeded_path_equals_injected_noise(emu_engine, tiny):
    hp = HP.TINY_GLOW
    check_seeded_path_equals_injected_noise(emu_engine, tiny["g"], hp.num_symbols, hp.mel_channels, lens=(23,), length_scale=1.0)
    check_seeded_path_equals_injected_noise(emu_engine, tiny["g"], hp.num_symbols, hp.mel_channels, lens=(17, 5, 30), length_scale=1.0)


def test_device_noise_is_standard_normal(emu_engine):
    from tests.noise_check import check_gauss_noise

    check_gauss_noise(emu_engine, 2, 16, 4000, ks_bound=1.95 / np.sqrt(2 * 16 * 4000))


def check_grouped_schedule(eng, hp, seed, frames):
    """Three MRF chains: the default schedule issues the same-geometry launches of a step as ONE
    grouped launch (conv_group_kernel / pair_group_kernel); the forked / one-by-one schedules must
    give the same bits, and all of them the oracle's waveform."""
    sd = synthetic.make_hifigan_state_dict(hp, seed=seed)
    v = eng.load_hifigan(hp, sd)
    rng = np.random.default_rng(seed + 1)
    fr = np.asarray(frames, np.int32)
    melin = (rng.standard_normal((len(fr), hp.num_mels, int(fr.max()))) * 2).astype(np.float32)
    mb = eng.mel_from_numpy(melin, fr)
    eng.set_profiling(True)
    eng.profile_reset()
    grouped, _ = eng.hifigan_infer(v, mb)
    n_grouped = eng.profile()["conv_mfma.hifigan_resblock"]["launches"]
    eng.set_option("mrf_group", 0)
    eng.profile_reset()
    try:
        forked, _ = eng.hifigan_infer(v, mb)
        n_forked = eng.profile()["conv_mfma.hifigan_resblock"]["launches"]
        eng.set_option("adaptive_schedule", 1)
        forked2, _ = eng.hifigan_infer(v, mb)
    finally:
        eng.set_option("mrf_group", 1)
        eng.set_option("adaptive_schedule", 0)
        eng.set_profiling(False)
    assert np.array_equal(grouped, forked) and np.array_equal(grouped, forked2)
    assert n_forked == 3 * n_grouped, (n_forked, n_grouped)  # every step of the three chains became one launch
    for b in range(len(fr)):
        ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : fr[b]])
        n = fr[b] * hp.hop
        assert np.sqrt(np.mean((grouped[b, :n] - ref) ** 2)) < 1e-5
        assert np.all(grouped[b, n:] == 0)
    eng.unload(v)


def test_grouped_mrf_launches_resblock1(emu_engine):
    # stages of 128 channels (un-fused convs -> conv_group_kernel) and 64 channels (fused pairs -> pair_group_kernel)
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3), (1, 3), (1, 5)), num_mels=16)
    check_grouped_schedule(emu_engine, hp, 71, [23, 17])


def test_grouped_mrf_launches_resblock2(emu_engine):
    hp = HP.HifiGanHParams(resblock="2", upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), upsample_initial_channel=64,
                           resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)), num_mels=16)
    check_grouped_schedule(emu_engine, hp, 73, [19])


def check_bf16x3_mode(eng, hp, seed, frames, rms_tol, precision=ffi.PRECISION_BF16X3):
    """`half`-style reduced precision: the wide ResBlock convs on the bf16 matrix cores with split
    operands (three bf16 MFMAs per product, f32 accumulate).  Close to — not equal to — the exact
    mode, within the documented tolerance of the oracle; switching back restores the exact bits."""
    sd = synthetic.make_hifigan_state_dict(hp, seed=seed)
    v = eng.load_hifigan(hp, sd)
    rng = np.random.default_rng(seed + 1)
    fr = np.asarray(frames, np.int32)
    melin = (rng.standard_normal((len(fr), hp.num_mels, int(fr.max()))) * 2).astype(np.float32)
    mb = eng.mel_from_numpy(melin, fr)
    exact, _ = eng.hifigan_infer(v, mb)
    eng.set_precision(v, precision)
    try:
        split, _ = eng.hifigan_infer(v, mb)
        eng.set_option("mrf_group", 0)
        split_ungrouped, _ = eng.hifigan_infer(v, mb)
    finally:
        eng.set_option("mrf_group", 1)
        eng.set_precision(v, ffi.PRECISION_F32)
    again, _ = eng.hifigan_infer(v, mb)
    assert np.array_equal(exact, again) and np.array_equal(split, split_ungrouped)
    assert not np.array_equal(exact, split)  # the mode really ran
    errs = []
    for b in range(len(fr)):
        ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : fr[b]])
        n = fr[b] * hp.hop
        errs.append(float(np.sqrt(np.mean((split[b, :n] - ref) ** 2))))
        assert errs[-1] < rms_tol, errs
        assert np.all(split[b, n:] == 0)
    with pytest.raises(ffi.Mi355ttsError):
        eng.set_precision(v, 7)
    eng.unload(v)
    return max(errs)


def test_bf16x3_mode_resblock1(emu_engine):
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3), (1, 3), (1, 5)), num_mels=16)
    check_bf16x3_mode(emu_engine, hp, 81, [23, 9], 1e-4)


def test_bf16x3_mode_resblock2_and_256_channels(emu_engine):
    hp = HP.HifiGanHParams(resblock="2", upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=512,
                           resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)), num_mels=16)
    check_bf16x3_mode(emu_engine, hp, 83, [11], 1e-4)


def test_128_row_tile_shape(emu_engine, monkeypatch):
    """TILE_M128 (four row groups of waves share one staged input tile, no k-split) is chosen for ResBlock convs with
    >= 256 such tiles — here forced at emulator sizes; also inside the grouped launch and against the one-conv form."""
    monkeypatch.setenv("MI355TTS_M128_MIN_TILES", "1")
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3), (1, 3), (1, 5)), num_mels=16)
    check_grouped_schedule(emu_engine, hp, 91, [70, 33])
    x = np.random.default_rng(3).standard_normal((2, 128, 300)).astype(np.float32)
    w = (np.random.default_rng(4).standard_normal((128, 128, 7)) / 30).astype(np.float32)
    bias = np.random.default_rng(5).standard_normal(128).astype(np.float32)
    lens = np.array([300, 201], np.int32)
    y = emu_engine.conv1d(x, w, bias, dilation=3, in_slope=0.1, lens=lens)  # KC_RESBLOCK class op: takes the M128 shape
    from oracle import nn_np

    for b in range(2):
        n = lens[b]
        ref = nn_np.conv1d(nn_np.leaky_relu(x[b, :, :n], 0.1), w, bias, dilation=3, padding=9)
        np.testing.assert_allclose(y[b, :, :n], ref, rtol=1e-4, atol=5e-5)
        assert np.all(y[b, :, n:] == 0)


def test_bf16x3_mode_64_and_32_channel_stages(emu_engine):
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=128,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3), (1, 3), (1, 5)), num_mels=16)
    check_bf16x3_mode(emu_engine, hp, 85, [150, 37], 1e-4)
    # these stages run conv1 -> lrelu -> conv2 -> + x as ONE kernel in this mode too (csrc/resblock_pair_bf16.h):
    # 2 stages x 2 dilations grouped launches of fused pairs (un-fused it would be 8)
    sd = synthetic.make_hifigan_state_dict(hp, seed=85)
    v = emu_engine.load_hifigan(hp, sd)
    mb = emu_engine.mel_from_numpy(np.zeros((1, hp.num_mels, 150), np.float32), np.array([150], np.int32))
    emu_engine.set_precision(v, ffi.PRECISION_BF16X3)
    emu_engine.set_profiling(True)
    emu_engine.profile_reset()
    try:
        emu_engine.hifigan_infer(v, mb)
        launches = emu_engine.profile()["conv_mfma.hifigan_resblock"]["launches"]
    finally:
        emu_engine.set_profiling(False)
        emu_engine.unload(v)
    assert launches == 4


def test_plain_bf16_mode(emu_engine):
    """MI355TTS_PRECISION_BF16: one bf16 MFMA per product (operands rounded to 8 mantissa bits, f32 accumulate) —
    the plain reduced precision of the reference's `.half()`; tolerance 2e-2 RMS on these O(0.2) waveforms, and
    clearly coarser than the split mode."""
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3), (1, 3), (1, 5)), num_mels=16)
    plain = check_bf16x3_mode(emu_engine, hp, 81, [23], 2e-2, precision=ffi.PRECISION_BF16)
    split = check_bf16x3_mode(emu_engine, hp, 81, [23], 1e-4)
    assert plain > 20 * split


def test_128_row_tile_for_the_upsampler(emu_engine, monkeypatch):
    """The polyphase ConvTranspose1d with whole 128-row groups of virtual rows (C_out * stride) also takes the 128-row
    tile — the 32-row shapes stage the same input once per m-tile (32x at stage 1 of 'high')."""
    from oracle import nn_np

    monkeypatch.setenv("MI355TTS_M128_MIN_TILES", "1")
    rng = np.random.default_rng(77)
    for Cin, Cout, K, u, L in ((40, 16, 16, 8, 150), (24, 64, 4, 2, 301)):
        x = rng.standard_normal((1, Cin, L)).astype(np.float32)
        w = (rng.standard_normal((Cin, Cout, K)) / np.sqrt(Cin * 2)).astype(np.float32)
        b = rng.standard_normal(Cout).astype(np.float32)
        y = emu_engine.conv_transpose1d(x, w, b, stride=u, in_slope=0.1)
        ref = nn_np.conv_transpose1d(nn_np.leaky_relu(x[0], 0.1), w, b, stride=u, padding=(K - u) // 2)
        np.testing.assert_allclose(y[0], ref, rtol=1e-4, atol=5e-5)


def test_continuous_stream_tile_equals_the_chunked_tile(emu_engine, monkeypatch):
    """`rb_group_kernel` (rb_conv.h: four-buffer LDS ring, mid-chunk barrier, no chunk seam) runs the grouped 128-row ResBlock
    launches by default; option "rb_conv" = 0 sends the same launches to the chunked tile of conv_mfma.h.  Same fragment
    streams and accumulation order: the waveforms are the same bits, rows of different lengths and a tile-edge length included."""
    monkeypatch.setenv("MI355TTS_M128_MIN_TILES", "1")
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), num_mels=16)
    sd = synthetic.make_hifigan_state_dict(hp, seed=93)
    v = emu_engine.load_hifigan(hp, sd)
    rng = np.random.default_rng(12)
    try:
        for frames in ([64], [70, 33], [16]):
            F = max(frames)
            mel = (0.5 + 0.1 * rng.standard_normal((len(frames), hp.num_mels, F))).astype(np.float32)
            mb = emu_engine.mel_from_numpy(mel, np.array(frames, np.int32))
            new, _ = emu_engine.hifigan_infer(v, mb)
            emu_engine.set_option("rb_conv", 0)
            try:
                old, _ = emu_engine.hifigan_infer(v, mb)
            finally:
                emu_engine.set_option("rb_conv", 1)
            assert np.array_equal(new, old)
            assert np.isfinite(new).all() and np.abs(new).max() > 1e-3
    finally:
        emu_engine.unload(v)


def test_grouped_launch_promotion_and_snake_order(emu_engine, monkeypatch):
    """At batch 1 the same-geometry ResBlock convs of a step that plan_conv left on the small tiles move to the 128-row tile
    when together they give every CU more than one workgroup (`promote_group_plans`, decided on the plans: every schedule then
    runs the same tile arithmetic), and a grouped launch whose workgroups are all resident at once is dispatched as a snake
    (`group_snake_order`: round 0 longest first, round 1 shortest first).  `MI355TTS_GROUP_NCU` = 24 makes the 490-frame shape
    such a launch (16 + 16 + 16 workgroups in rounds of 24: segments k11 x16, k7 x8, k3 x16, k7 x8).  The order is a
    permutation of the tiles: same bits as the chunked 128-row kernel in its plain order (option rb_conv = 0), as the forked
    schedule, and as the plain order of the same kernel."""
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 5), (1, 3), (1, 5)), num_mels=16)
    sd = synthetic.make_hifigan_state_dict(hp, seed=95)
    v = emu_engine.load_hifigan(hp, sd)
    rng = np.random.default_rng(14)
    try:
        mel = (0.5 + 0.1 * rng.standard_normal((1, hp.num_mels, 490))).astype(np.float32)  # 980 columns: 16 tiles per member
        mb = emu_engine.mel_from_numpy(mel)
        monkeypatch.setenv("MI355TTS_GROUP_NCU", "24")
        emu_engine.set_profiling(True)
        emu_engine.profile_reset()
        promoted, _ = emu_engine.hifigan_infer(v, mb)  # small tiles in the plan -> promoted, snake order
        n_grouped = emu_engine.profile()["conv_mfma.hifigan_resblock"]["launches"]
        emu_engine.set_profiling(False)
        emu_engine.set_option("rb_conv", 0)
        try:
            chunked, _ = emu_engine.hifigan_infer(v, mb)  # promoted too; the chunked 128-row kernel, plain order
        finally:
            emu_engine.set_option("rb_conv", 1)
        emu_engine.set_option("mrf_group", 0)
        try:
            forked, _ = emu_engine.hifigan_infer(v, mb)   # promoted; one launch per conv
        finally:
            emu_engine.set_option("mrf_group", 1)
        emu_engine.set_option("group_promote", 0)
        try:
            unpromoted, _ = emu_engine.hifigan_infer(v, mb)  # option off: the small tiles of the plan
        finally:
            emu_engine.set_option("group_promote", 1)
        monkeypatch.setenv("MI355TTS_M128_MIN_TILES", "1")
        monkeypatch.setenv("MI355TTS_GROUP_NCU", "8")     # 48 workgroups > 4 x 8: the plain order of the same kernel
        plain, _ = emu_engine.hifigan_infer(v, mb)
        monkeypatch.delenv("MI355TTS_M128_MIN_TILES")
        monkeypatch.setenv("MI355TTS_GROUP_NCU", "1024")  # no step has more workgroups than CUs: no promotion, the 64 x 32 k-split tile
        small, _ = emu_engine.hifigan_infer(v, mb)
        assert np.isfinite(promoted).all() and np.abs(promoted).max() > 1e-3
        assert n_grouped == 2 * 2 + 2  # the 128-channel stage: two grouped launches per dilation step; the 64-channel stage: fused pairs
        assert np.array_equal(promoted, chunked) and np.array_equal(promoted, forked)
        # (MI355TTS_M128_MIN_TILES also moves the upsamplers to their 128-row tile — another summation order: round-off here)
        assert np.abs(plain - promoted).max() <= 1e-6
        assert not np.array_equal(small, promoted) and np.abs(small - promoted).max() <= 1e-6  # the k-split tile sums in another order
        assert np.array_equal(unpromoted, small)
        # the rule: a step just above one workgroup per CU stays on the small tiles — 6 tiles per member on "16 CUs" would put
        # 11 + 3 tap-units on six CUs where the mean is 7.9 (group_order_imbalance 1.78 > 1.35)
        mel2 = (0.5 + 0.1 * rng.standard_normal((1, hp.num_mels, 170))).astype(np.float32)
        mb2 = emu_engine.mel_from_numpy(mel2)
        small2, _ = emu_engine.hifigan_infer(v, mb2)           # MI355TTS_GROUP_NCU = 1024: nothing promoted
        monkeypatch.setenv("MI355TTS_GROUP_NCU", "16")
        kept, _ = emu_engine.hifigan_infer(v, mb2)
        monkeypatch.setenv("MI355TTS_PROMOTE_MAX_IMBALANCE", "2")
        forced, _ = emu_engine.hifigan_infer(v, mb2)
        assert np.array_equal(kept, small2) and not np.array_equal(forced, small2) and np.abs(forced - small2).max() <= 1e-6
    finally:
        emu_engine.set_profiling(False)
        emu_engine.unload(v)


def test_128_column_tiles_of_the_one_row_tile_stage_give_the_same_bits(emu_engine, monkeypatch):
    """A grouped launch of the 128-channel stage with more 128-column tiles than the chip holds at once (3 workgroups per CU), issued
    while another call is in flight (here: forced through `MI355TTS_RB_NB4_MIN_TILES`), runs `rb_group_kernel<11, 7, 3, 4>` — four column blocks per wave, the interior tiles' 16-byte epilogue in two passes of 64
    columns: the same chain per output element, so the same bits as the 64-column tile.  `MI355TTS_GROUP_NCU` = 8 puts the
    threshold at 24 tiles; 1010 columns = 8 tiles per member, the last one an edge tile (114 of 128 columns)."""
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=256,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 5), (1, 3), (1, 5)), num_mels=16)
    sd = synthetic.make_hifigan_state_dict(hp, seed=96)
    v = emu_engine.load_hifigan(hp, sd)
    try:
        mel = (0.5 + 0.1 * np.random.default_rng(15).standard_normal((1, hp.num_mels, 505))).astype(np.float32)
        mb = emu_engine.mel_from_numpy(mel)
        monkeypatch.setenv("MI355TTS_M128_MIN_TILES", "1")
        monkeypatch.setenv("MI355TTS_GROUP_NCU", "8")
        monkeypatch.setenv("MI355TTS_RB_NB4_MIN_TILES", "0")
        narrow, _ = emu_engine.hifigan_infer(v, mb)
        monkeypatch.setenv("MI355TTS_RB_NB4_MIN_TILES", "24")  # the default threshold of a busy context: 3 x 8 = 24 tiles
        emu_engine.profile_reset()
        wide, _ = emu_engine.hifigan_infer(v, mb)
        counts = emu_engine.kernel_counts()
        assert counts.get("rb_group_kernel.nb4", 0) == 2 * 2 and counts.get("rb_group_kernel", 0) + counts.get("rb_group_kernel.snake", 0) == 0, counts
        monkeypatch.setenv("MI355TTS_RB_NB4_MIN_TILES", "25")  # one more than the launch has: the 64-column tile again
        again, _ = emu_engine.hifigan_infer(v, mb)
        assert np.isfinite(wide).all() and np.abs(wide).max() > 1e-3
        assert np.array_equal(wide, narrow) and np.array_equal(again, narrow)
        # without the variable the rule looks at the load: a lone call keeps the 64-column tiles (the launch has the chip to itself)
        monkeypatch.delenv("MI355TTS_RB_NB4_MIN_TILES")
        emu_engine.profile_reset()
        lone, _ = emu_engine.hifigan_infer(v, mb)
        assert emu_engine.kernel_counts().get("rb_group_kernel.nb4", 0) == 0 and np.array_equal(lone, narrow)
        ref = hifi_gan_np.hifigan_infer(sd, hp, mel[0])
        assert np.sqrt(np.mean((wide[0, : ref.shape[0]] - ref) ** 2)) < 1e-5
    finally:
        emu_engine.unload(v)


def test_four_wave_pair_kernel_against_the_k_split_one(emu_engine, monkeypatch):
    """`rb_pair_kernel` / `rb_pair_group_kernel` (rb_pair.h: 4 waves, no k-split, the parked conv1 tile over the x tile) run the
    fused ResBlock steps of the 64- / 32-channel stages by default; option "rb_pair" = 0 sends them to the 8-wave k-split
    kernel of resblock_pair.h.  Same tiles and arithmetic up to the summation order: the waveforms agree to f32 round-off, in
    the grouped and in the one-launch-per-step schedule, rows of different lengths and tile-edge lengths included.  (Launches
    with fewer than 512 tiles per member keep the k-split kernel: `MI355TTS_RB_PAIR_MIN_TILES` lowers that for emulator sizes.)"""
    monkeypatch.setenv("MI355TTS_RB_PAIR_MIN_TILES", "1")
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=128,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5), (1, 3, 5)), num_mels=16)
    sd = synthetic.make_hifigan_state_dict(hp, seed=94)
    v = emu_engine.load_hifigan(hp, sd)
    rng = np.random.default_rng(13)
    try:
        for frames in ([61], [70, 33], [30]):
            F = max(frames)
            mel = (0.5 + 0.1 * rng.standard_normal((len(frames), hp.num_mels, F))).astype(np.float32)
            mb = emu_engine.mel_from_numpy(mel, np.array(frames, np.int32))
            new, _ = emu_engine.hifigan_infer(v, mb)
            emu_engine.set_option("serial_branches", 1)
            try:
                new_serial, _ = emu_engine.hifigan_infer(v, mb)
            finally:
                emu_engine.set_option("serial_branches", 0)
            emu_engine.set_option("rb_pair", 0)
            try:
                old, _ = emu_engine.hifigan_infer(v, mb)
            finally:
                emu_engine.set_option("rb_pair", 1)
            assert np.isfinite(new).all() and np.abs(new).max() > 1e-3
            assert np.sqrt(np.mean((new - old) ** 2)) <= 2e-6 and np.abs(new - old).max() <= 2e-5
            assert np.sqrt(np.mean((new - new_serial) ** 2)) <= 2e-6  # (the serial form folds the MRF average elsewhere)
            for b, f in enumerate(frames):
                assert np.all(new[b, f * hp.hop:] == 0)
    finally:
        emu_engine.unload(v)
