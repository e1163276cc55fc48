"""Whole hot path (GlowTTS -> mel transform -> HiFi-GAN -> int16) through the C
ABI on the CPU emulator build, at shrunk hyper-parameters, against the numpy
oracle.  Exercises the host schedule, weight folding/packing, every kernel's
index logic and the batch/padding path without a GPU."""
import numpy as np
import pytest

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from oracle import audio_np, glow_tts_np, hifi_gan_np


@pytest.fixture(scope="module")
def tiny_models(emu_engine):
    gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7)
    vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7)
    vsd2 = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN_RB2, seed=8)
    return dict(
        gsd=gsd, vsd=vsd, vsd2=vsd2,
        g=emu_engine.load_glow(HP.TINY_GLOW, gsd),
        v=emu_engine.load_hifigan(HP.TINY_HIFIGAN, vsd),
        v2=emu_engine.load_hifigan(HP.TINY_HIFIGAN_RB2, vsd2),
    )


def _ids(rng, n):
    return synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols)


def test_glow_single_utterance(emu_engine, tiny_models):
    rng = np.random.default_rng(3)
    ids = _ids(rng, 23)
    noise = rng.standard_normal((HP.TINY_GLOW.mel_channels, 400)).astype(np.float32)
    taps = {}
    ref = glow_tts_np.glow_tts_infer(tiny_models["gsd"], HP.TINY_GLOW, ids, noise, 0.667, 1.0, taps)
    mel = emu_engine.glow_infer(tiny_models["g"], ids, 0.667, 1.0, noise=noise, audio_settings=ljspeech_audio_settings())
    assert mel.frames[0] == ref.shape[1]
    got = mel.numpy("raw")[0]
    np.testing.assert_allclose(got, ref, atol=2e-5, rtol=1e-4)
    voc = audio_np.mel_to_vocoder_input(ref, ljspeech_audio_settings())
    np.testing.assert_allclose(mel.numpy("vocoder")[0], voc, atol=2e-3, rtol=1e-4)


def test_glow_variable_length_batch_equals_rowwise(emu_engine, tiny_models):
    """the reference never batches; each row of a padded batch must
    equal its own B=1 result and the padded tail must be exactly zero."""
    rng = np.random.default_rng(4)
    rows = [_ids(rng, n) for n in (9, 30, 17)]
    M = HP.TINY_GLOW.mel_channels
    noise = rng.standard_normal((3, M, 400)).astype(np.float32)
    mel = emu_engine.glow_infer(tiny_models["g"], rows, 0.5, 1.1, noise=noise)
    got = mel.numpy("raw")
    for b, ids in enumerate(rows):
        ref = glow_tts_np.glow_tts_infer(tiny_models["gsd"], HP.TINY_GLOW, ids, noise[b], 0.5, 1.1)
        F = ref.shape[1]
        assert mel.frames[b] == F
        np.testing.assert_allclose(got[b, :, :F], ref, atol=2e-5, rtol=1e-4)
        assert np.all(got[b, :, F:] == 0)


@pytest.mark.parametrize("which", ["v", "v2"])
def test_hifigan_batch(emu_engine, tiny_models, which):
    hp = HP.TINY_HIFIGAN if which == "v" else HP.TINY_HIFIGAN_RB2
    sd = tiny_models["vsd" if which == "v" else "vsd2"]
    rng = np.random.default_rng(5)
    frames = np.array([37, 12], np.int32)
    melin = (rng.standard_normal((2, hp.num_mels, 37)) * 2).astype(np.float32)
    mb = emu_engine.mel_from_numpy(melin, frames)
    f32, i16 = emu_engine.hifigan_infer(tiny_models[which], mb)
    hop = hp.hop
    for b in range(2):
        ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : frames[b]])
        n = frames[b] * hop
        assert ref.shape[0] == n
        err = f32[b, :n] - ref
        assert np.sqrt(np.mean(err ** 2)) < 1e-5, np.abs(err).max()
        assert np.all(f32[b, n:] == 0) and np.all(i16[b, n:] == 0)
        ref16 = audio_np.audio_float_to_int16(ref)
        assert np.abs(i16[b, :n].astype(np.int32) - ref16.astype(np.int32)).max() <= 1


def test_end_to_end_tiny(emu_engine, tiny_models):
    rng = np.random.default_rng(6)
    ids = _ids(rng, 12)
    s = ljspeech_audio_settings()
    mel = emu_engine.glow_infer(tiny_models["g"], ids, 0.0, 1.0, audio_settings=s)
    f32, i16 = emu_engine.hifigan_infer(tiny_models["v"], mel)
    ref_mel = glow_tts_np.glow_tts_infer(tiny_models["gsd"], HP.TINY_GLOW, ids, None, 0.0, 1.0)
    ref_wav = hifi_gan_np.hifigan_infer(tiny_models["vsd"], HP.TINY_HIFIGAN, audio_np.mel_to_vocoder_input(ref_mel, s))
    assert f32.shape[1] == ref_wav.shape[0]
    assert np.sqrt(np.mean((f32[0] - ref_wav) ** 2)) < 1e-4


def test_errors_are_reported_not_crashed(emu_engine, tiny_models):
    from larynx_amd.ffi import Mi355ttsError

    with pytest.raises(Mi355ttsError):
        emu_engine.glow_infer(999, np.array([3, 4, 2]))
    with pytest.raises(Mi355ttsError):  # id outside the voice's symbol table (the reference's embedding raises)
        emu_engine.glow_infer(tiny_models["g"], np.array([3, HP.TINY_GLOW.num_symbols, 2]))
    with pytest.raises(Mi355ttsError):  # noise too short for the utterance
        emu_engine.glow_infer(tiny_models["g"], _ids(np.random.default_rng(1), 20), 0.667, 1.0,
                              noise=np.zeros((HP.TINY_GLOW.mel_channels, 4), np.float32))


def test_serial_and_concurrent_branches_agree(emu_engine, tiny_models):
    """The MRF chains run on side streams by default; the single-stream schedule
    (used when timing single kernels) must give the same waveform."""
    rng = np.random.default_rng(9)
    melin = (rng.standard_normal((1, HP.TINY_HIFIGAN.num_mels, 21)) * 2).astype(np.float32)
    mb = emu_engine.mel_from_numpy(melin)
    a, _ = emu_engine.hifigan_infer(tiny_models["v"], mb)
    emu_engine.set_option("serial_branches", 1)
    try:
        b, _ = emu_engine.hifigan_infer(tiny_models["v"], mb)
    finally:
        emu_engine.set_option("serial_branches", 0)
    ref = hifi_gan_np.hifigan_infer(tiny_models["vsd"], HP.TINY_HIFIGAN, melin[0])
    assert np.sqrt(np.mean((a[0] - ref) ** 2)) < 1e-5 and np.sqrt(np.mean((b[0] - ref) ** 2)) < 1e-5


def test_glow_multi_block_attention(emu_engine):
    """P = 75 ids: three 32-wide key blocks with a ragged last one and dk = 32 per
    head — exercises the MFMA attention kernel's blocking, masking and band terms."""
    hp = HP.GlowHParams(num_symbols=30, hidden_channels=64, filter_channels=64, filter_channels_dp=32, n_blocks_dec=1,
                        n_layers_enc=2, n_block_layers=1, mel_channels=8)
    sd = synthetic.make_glow_state_dict(hp, seed=11)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(12)
    for n in (75, 33):
        ids = synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols)
        taps = {}
        ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 1.0, taps)
        mel = emu_engine.glow_infer(g, ids, 0.0, 1.0)
        assert mel.frames[0] == ref.shape[1]
        np.testing.assert_allclose(mel.numpy("raw")[0], ref, atol=3e-5, rtol=1e-4)
    emu_engine.unload(g)


def check_fallback_kernels(engine):
    """Hyper-parameters no shipped voice uses still take correct (slower) paths:
    `n_split = 8` runs InvConvNear + ActNorm as the standalone kernel instead of inside the
    coupling epilogue, a 264-channel duration predictor uses the generic LayerNorm, and an
    input longer than the MFMA attention's score tile (768 keys) the VALU attention."""
    hp = HP.GlowHParams(num_symbols=30, hidden_channels=32, filter_channels=32, filter_channels_dp=264, n_blocks_dec=2,
                        n_layers_enc=1, n_block_layers=1, mel_channels=8, n_split=8)
    sd = synthetic.make_glow_state_dict(hp, seed=13)
    g = engine.load_glow(hp, sd)
    rng = np.random.default_rng(14)
    for n in (21, 790):
        ids = synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols)
        ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 0.4)
        mel = engine.glow_infer(g, ids, 0.0, 0.4)
        assert mel.frames[0] == ref.shape[1]
        np.testing.assert_allclose(mel.numpy("raw")[0], ref, atol=5e-5, rtol=1e-4)
    engine.unload(g)


def test_fallback_kernels_for_unusual_hparams(emu_engine):
    check_fallback_kernels(emu_engine)


def test_denoise_kernels_match_oracle(emu_engine):
    """STFT -> spectral subtraction -> iSTFT (larynx/hifi_gan.py:171-179,
    larynx/audio.py:232-289) against the float64 numpy restatement."""
    from oracle import denoise_np

    rng = np.random.default_rng(21)
    wav = (rng.standard_normal((2, 256 * 14)) * 0.3).astype(np.float32)
    bias = np.abs(rng.standard_normal(513)).astype(np.float32)
    for strength in (0.0, 0.4):
        got = emu_engine.denoise(wav, bias, strength)
        for b in range(2):
            ref = denoise_np.denoise(wav[b], bias, strength)
            assert ref.shape[0] == wav.shape[1]
            assert np.abs(got[b] - ref).max() < 2e-5, np.abs(got[b] - ref).max()


@pytest.mark.parametrize("serial", [0, 1])
def test_fused_resblock_pair_kernel(emu_engine, serial):
    """Stages of 64 and 32 channels run conv1 -> lrelu -> conv2 -> +x as ONE kernel
    (csrc/resblock_pair.h): several tiles per row, ragged batch, both MRF schedules."""
    hp = HP.TINY_HIFIGAN_PAIR
    sd = synthetic.make_hifigan_state_dict(hp, seed=31)
    v = emu_engine.load_hifigan(hp, sd)
    rng = np.random.default_rng(32)
    frames = np.array([150, 67], np.int32)
    melin = (rng.standard_normal((2, hp.num_mels, 150)) * 2).astype(np.float32)
    mb = emu_engine.mel_from_numpy(melin, frames)
    emu_engine.set_option("serial_branches", serial)
    emu_engine.set_profiling(True)
    emu_engine.profile_reset()
    try:
        f32, _ = emu_engine.hifigan_infer(v, mb)
        launches = emu_engine.profile()["conv_mfma.hifigan_resblock"]["launches"]
    finally:
        emu_engine.set_option("serial_branches", 0)
        emu_engine.set_profiling(False)
    assert launches == 2 * 2 * 2  # stages x kernels x dilations: one fused launch per conv PAIR
    for b in range(2):
        ref = hifi_gan_np.hifigan_infer(sd, hp, melin[b, :, : frames[b]])
        n = frames[b] * hp.hop
        assert np.sqrt(np.mean((f32[b, :n] - ref) ** 2)) < 1e-5
        assert np.all(f32[b, n:] == 0)
    emu_engine.unload(v)


def test_mel_outliving_its_engine_is_harmless(emu_library):
    """Closing an engine frees the mels it still owns; a later MelBatch.free()/__del__ is a no-op."""
    from larynx_amd.engine import Engine

    eng = Engine(0, library_path=emu_library)
    g = eng.load_glow(HP.TINY_GLOW, synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7))
    mel = eng.glow_infer(g, np.array([3, 5, 6, 2]), 0.0, 1.0)
    eng.close()
    mel.free()
    del mel


@pytest.mark.parametrize("n_ids", [1, 2, 3])
def test_shortest_utterances(emu_engine, tiny_models, n_ids):
    """One-, two- and three-id inputs (shorter than the attention window and than any
    tile), and a one-frame mel through the vocoder."""
    ids = np.array([3, 7, 2][:n_ids], np.int64)
    ref = glow_tts_np.glow_tts_infer(tiny_models["gsd"], HP.TINY_GLOW, ids, None, 0.0, 1.0)
    mel = emu_engine.glow_infer(tiny_models["g"], ids, 0.0, 1.0)
    assert mel.frames[0] == ref.shape[1]
    if ref.shape[1]:
        np.testing.assert_allclose(mel.numpy("raw")[0], ref, atol=2e-5, rtol=1e-4)
    one = (np.random.default_rng(n_ids).standard_normal((1, HP.TINY_HIFIGAN.num_mels, n_ids))).astype(np.float32)
    f32, i16 = emu_engine.hifigan_infer(tiny_models["v"], emu_engine.mel_from_numpy(one))
    refw = hifi_gan_np.hifigan_infer(tiny_models["vsd"], HP.TINY_HIFIGAN, one[0])
    assert f32.shape[1] == refw.shape[0] and np.sqrt(np.mean((f32[0] - refw) ** 2)) < 1e-5
