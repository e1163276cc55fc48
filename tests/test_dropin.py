"""The reference-shaped host interface (`larynx_amd.load_tts_model`,
`load_vocoder_model`, `sentence_task`, `phonemes_to_speech`) over a voice
directory on disk — every test runs twice: on the emulator build (CPU CI) and,
under `-m gpu`, on the real `libmi355tts.so`."""
import json

import numpy as np
import pytest

import larynx_amd
from larynx_amd import ffi
from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from larynx_amd.constants import InferenceBackend, TextToSpeechType, VocoderType
from oracle import audio_np, glow_tts_np, hifi_gan_np


@pytest.fixture(scope="module", params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def emu_library(request):
    """Library path handed to the host classes: the emulator build, or None = the hipcc-built
    in-tree library (the name is kept so the tests read the same for both)."""
    if request.param == "emu":
        return request.getfixturevalue("emu_library_path")
    return None


@pytest.fixture(scope="module")
def voice_dirs(tmp_path_factory):
    root = tmp_path_factory.mktemp("voices")
    gdir, vdir = root / "tiny-glow_tts", root / "tiny_hifi_gan"
    gdir.mkdir()
    vdir.mkdir()
    gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=3)
    vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=3)
    cfg = HP.TINY_GLOW.to_config()
    cfg["audio"].update({k: v for k, v in vars(ljspeech_audio_settings()).items() if k != "mel_channels"})
    (gdir / "config.json").write_text(json.dumps(cfg))
    vcfg = HP.TINY_HIFIGAN.to_config()
    vcfg["audio"] = {"num_mels": HP.TINY_HIFIGAN.num_mels}
    (vdir / "config.json").write_text(json.dumps(vcfg))
    np.savez(gdir / "generator.npz", **gsd)
    np.savez(vdir / "generator.npz", **vsd)
    return gdir, vdir, gsd, vsd


def test_models_behave_like_the_reference_classes(emu_library, voice_dirs):
    gdir, vdir, gsd, vsd = voice_dirs
    tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, backend=InferenceBackend.HIP, library_path=emu_library)
    voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, backend=InferenceBackend.HIP, library_path=emu_library)
    # the registry attaches these after construction (larynx/__init__.py:362-363)
    setattr(tts, "phoneme_to_id", {"_": 0})
    setattr(tts, "audio_settings", ljspeech_audio_settings())
    exposed = getattr(tts, "audio_settings")
    assert not (exposed.signal_norm or exposed.convert_db_to_amp or exposed.do_dynamic_range_compression)
    assert exposed.sample_rate == 22050
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(2), 13, HP.TINY_GLOW.num_symbols)
    audio = larynx_amd.sentence_task("hello", ids, exposed, tts, {"noise_scale": 0.0}, voc, None, pause_before_ms=10, pause_after_ms=20)
    assert audio.dtype == np.int16 and audio.ndim == 1
    s = ljspeech_audio_settings()
    ref_mel = glow_tts_np.glow_tts_infer(gsd, HP.TINY_GLOW, ids, None, 0.0, 1.0)
    ref = audio_np.audio_float_to_int16(hifi_gan_np.hifigan_infer(vsd, HP.TINY_HIFIGAN, audio_np.mel_to_vocoder_input(ref_mel, s)))
    before, after = 220, 441
    assert audio.shape[0] == before + ref.shape[0] + after
    assert np.all(audio[:before] == 0) and np.all(audio[-after:] == 0)
    assert np.abs(audio[before:-after].astype(np.int32) - ref.astype(np.int32)).max() <= 1
    # reference-style array in, int16 out (`mels_to_audio` with an already transformed ndarray)
    a2 = voc.mels_to_audio(audio_np.mel_to_vocoder_input(ref_mel, s)[None])
    assert np.abs(a2.astype(np.int32) - ref.astype(np.int32)).max() <= 1
    # np.asarray on the returned mel gives the reference's [1, M, F] array
    mel = tts.phonemes_to_mels(ids, {"noise_scale": 0.0})
    assert np.asarray(mel).shape == (1, HP.TINY_GLOW.mel_channels, ref_mel.shape[1])
    np.testing.assert_allclose(np.asarray(mel)[0], ref_mel, atol=2e-5)


def test_reference_pth_checkpoints_load(emu_library, voice_dirs, tmp_path):
    """The reference's own checkpoint files: `generator.pth` = torch-pickled
    `{"model": state_dict}` (glow_tts/checkpoint.py:41) / `{"generator": state_dict}`
    (hifi_gan/checkpoint.py:49), read with `weights_only=True`."""
    import shutil

    import torch

    gdir, vdir, gsd, vsd = voice_dirs
    g2, v2 = tmp_path / "pth-glow_tts", tmp_path / "pth_hifi_gan"
    g2.mkdir()
    v2.mkdir()
    shutil.copy(gdir / "config.json", g2 / "config.json")
    shutil.copy(vdir / "config.json", v2 / "config.json")
    torch.save({"model": {k: torch.from_numpy(np.asarray(v)) for k, v in gsd.items()}, "global_step": 1}, g2 / "generator.pth")
    torch.save({"generator": {k: torch.from_numpy(np.asarray(v)) for k, v in vsd.items()}}, v2 / "generator.pth")
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(4), 11, HP.TINY_GLOW.num_symbols)
    out = []
    for gd, vd in ((gdir, vdir), (g2, v2)):
        tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gd, library_path=emu_library)
        voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vd, library_path=emu_library)
        out.append(voc.mels_to_audio(tts.phonemes_to_mels(ids, {"noise_scale": 0.0})))
    assert out[0].size > 0 and np.array_equal(out[0], out[1])


def test_reference_style_tts_model_feeds_the_hip_vocoder(emu_library, voice_dirs):
    """Mixed deployment: a TextToSpeechModel that returns a plain `[1, M, F]` array with the
    un-fused `audio_settings` (what the reference's own GlowTextToSpeech does) in front of
    the HIP vocoder.  `sentence_task` then applies the three mel transforms in-kernel while
    wrapping the array; the result equals the fully fused path."""
    from larynx_amd.interfaces import TextToSpeechModel

    gdir, vdir, *_ = voice_dirs
    tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, library_path=emu_library)
    voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, library_path=emu_library)
    setattr(tts, "audio_settings", ljspeech_audio_settings())

    class ArrayTTS(TextToSpeechModel):
        def __init__(self):
            pass

        def phonemes_to_mels(self, phoneme_ids, settings=None):
            return np.asarray(tts.phonemes_to_mels(phoneme_ids, settings))  # raw GlowTTS output, host array

    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(6), 12, HP.TINY_GLOW.num_symbols)
    fused = larynx_amd.sentence_task("x", ids, tts.audio_settings, tts, {"noise_scale": 0.0}, voc, None)
    mixed = larynx_amd.sentence_task("x", ids, ljspeech_audio_settings(), ArrayTTS(), {"noise_scale": 0.0}, voc, None)
    assert fused.shape == mixed.shape and np.array_equal(fused, mixed)


def test_phonemes_to_speech_keeps_submission_order(emu_library, voice_dirs):
    gdir, vdir, gsd, vsd = voice_dirs
    tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, library_path=emu_library)
    voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, library_path=emu_library)
    rng = np.random.default_rng(8)
    sents = [(f"s{i}", synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols)) for i, n in enumerate((12, 6, 9))]
    res = list(larynx_amd.phonemes_to_speech(sents, tts, voc, tts_settings={"noise_scale": 0.0}))
    assert [r.text for r in res] == ["s0", "s1", "s2"] and all(r.sample_rate == 22050 for r in res)
    for (text, ids), r in zip(sents, res):
        one = voc.mels_to_audio(tts.phonemes_to_mels(ids, {"noise_scale": 0.0}))
        assert np.array_equal(one, r.audio)
    # the pool host asked the engine for a worker per pool thread (+ a spare) before the first sentence (Engine.ensure_workers:
    # mi355tts_reserve without models), so the engine knows its worker streams' hardware-queue groups; an explicit pool of three:
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=3) as pool:
        res3 = list(larynx_amd.phonemes_to_speech(sents, tts, voc, tts_settings={"noise_scale": 0.0}, executor=pool))
    groups = tts.engine.worker_queue_groups()
    assert len(groups) >= 4 and sum(1 for g in groups if g >= 0) >= 4, groups
    assert all(np.array_equal(a.audio, b.audio) for a, b in zip(res, res3))


def test_raw_stream_writes_sentences_in_order(emu_library, voice_dirs):
    import io

    from larynx_amd.streaming import stream_raw_pcm

    gdir, vdir, *_ = voice_dirs
    tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, library_path=emu_library)
    voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, library_path=emu_library)
    setattr(tts, "audio_settings", ljspeech_audio_settings())
    rng = np.random.default_rng(11)
    sents = [(f"s{i}", synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols), 0, 5 * (i % 2))
             for i, n in enumerate((14, 5, 9, 7, 11, 6, 8))]
    sink = io.BytesIO()
    stats = stream_raw_pcm(iter(sents), tts, voc, sink, tts_settings={"noise_scale": 0.0}, max_thread_workers=2,
                           raw_stream_queue_size=2, max_pending=3)
    want = []
    for _, ids, before, after in sents:
        a = voc.mels_to_audio(tts.phonemes_to_mels(ids, {"noise_scale": 0.0}))
        want.append(np.pad(a, (0, (after * 22050) // 1000)))
    want = np.concatenate(want)
    got = np.frombuffer(sink.getvalue(), np.int16)
    assert stats.sentences == len(sents) and stats.samples == want.shape[0] == got.shape[0]
    assert np.array_equal(got, want)
    assert 0.0 < stats.seconds_to_first_audio <= stats.seconds_total

    class Broken(io.BytesIO):
        def write(self, b):
            raise BrokenPipeError("sink closed")

    with pytest.raises(BrokenPipeError):
        stream_raw_pcm(iter(sents), tts, voc, Broken(), tts_settings={"noise_scale": 0.0})

    def bad_source():
        yield sents[0]
        yield ("bad", [10 ** 6])  # id outside the symbol table: the task fails, the error surfaces here

    from larynx_amd.ffi import Mi355ttsError

    with pytest.raises(Mi355ttsError):
        stream_raw_pcm(bad_source(), tts, voc, io.BytesIO(), tts_settings={"noise_scale": 0.0})


def test_unsupported_requests_raise(emu_library, voice_dirs):
    gdir, vdir, *_ = voice_dirs
    with pytest.raises(ValueError):
        larynx_amd.load_tts_model("tacotron2", gdir, library_path=emu_library)
    with pytest.raises(ValueError):
        larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, backend=InferenceBackend.ONNX, library_path=emu_library)
    voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, library_path=emu_library)
    from larynx_amd.ffi import Mi355ttsError

    with pytest.raises(Mi355ttsError):  # 4 frames x hop 8 = 32 samples: shorter than one STFT frame (the reference raises too)
        voc.mels_to_audio(np.zeros((1, HP.TINY_HIFIGAN.num_mels, 4), np.float32), {"denoiser_strength": 0.01})


def test_half_switch_is_accepted_like_the_reference(emu_library, tmp_path):
    """`half=True` (the reference's registry default for voices, larynx/__init__.py:297; `.half()` at
    larynx/glow_tts.py:90-91, larynx/hifi_gan.py:96-97): the acoustic model's decoder WaveNets run in fp16 (csrc/wn_f16.h), the
    vocoder runs its native fp16 mode (csrc/conv_f16.h) — same interface, audio within a half-precision band of the exact mode
    (the reference's own models under .half() move the int16 samples by 340 - 470 LSB on the golden set:
    tests/golden/glow_half_reference.json, both_half_i16; its generator alone by 40 - 85: tests/golden/*.npz, ref_half_i16)."""
    hp = HP.HifiGanHParams(upsample_rates=(2, 2), upsample_kernel_sizes=(4, 4), upsample_initial_channel=128,
                           resblock_kernel_sizes=(3, 7, 11), resblock_dilation_sizes=((1, 3), (1, 3), (1, 5)), num_mels=16)
    gdir, vdir = tmp_path / "half-glow_tts", tmp_path / "half_hifi_gan"
    gdir.mkdir()
    vdir.mkdir()
    (gdir / "config.json").write_text(json.dumps(HP.TINY_GLOW.to_config()))
    (vdir / "config.json").write_text(json.dumps(hp.to_config()))
    np.savez(gdir / "generator.npz", **synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=3))
    np.savez(vdir / "generator.npz", **synthetic.make_hifigan_state_dict(hp, seed=5))
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(9), 14, HP.TINY_GLOW.num_symbols)
    out = {}
    for half in (False, True):
        tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, half=half, library_path=emu_library)
        voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, half=half, library_path=emu_library)
        assert voc.half is half
        assert voc.precision == (ffi.PRECISION_F16 if half else ffi.PRECISION_F32)
        out[half] = voc.mels_to_audio(tts.phonemes_to_mels(ids, {"noise_scale": 0.0}))
    assert out[True].shape == out[False].shape and out[True].dtype == np.int16
    d = np.abs(out[True].astype(np.int32) - out[False].astype(np.int32))
    assert d.max() <= 512 and not np.array_equal(out[True], out[False])


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
def test_full_size_voice_through_sentence_task_on_the_device(tmp_path, half):
    """BASELINE config 2 through the reference-shaped classes at FULL size: a voice directory holding the ljspeech GlowTTS +
    hifi_gan 'high' checkpoints (seeded synthetic weights — the goldens' own), `HipGlowTextToSpeech` / `HipHiFiGanVocoder` built
    by the registry functions, `larynx_amd.sentence_task` as `/root/reference/larynx/__init__.py:229-257` runs it, against the
    reference-made golden of the bench utterance (`ljspeech_high_S120`): int16 within 1 LSB in the exact mode; with `half=True`
    (the registry's default for voices, larynx/__init__.py:297: BOTH models take it) within the deviation of the reference's OWN
    models under .half() on that case (tests/golden/glow_half_reference.json: both_half_i16)."""
    from tests.golden_util import load_case, load_glow_half_reference

    c = load_case("ljspeech_high_S120")
    gdir, vdir = tmp_path / "ljspeech-glow_tts", tmp_path / "hifi_gan_universal_large"
    gdir.mkdir()
    vdir.mkdir()
    cfg = c["glow_hp"].to_config()
    cfg["audio"].update({k: v for k, v in vars(ljspeech_audio_settings()).items() if k != "mel_channels"})
    (gdir / "config.json").write_text(json.dumps(cfg))
    vcfg = c["voc_hp"].to_config()
    vcfg["audio"] = {"num_mels": c["voc_hp"].num_mels}
    (vdir / "config.json").write_text(json.dumps(vcfg))
    np.savez(gdir / "generator.npz", **synthetic.make_glow_state_dict(c["glow_hp"], seed=1234))
    np.savez(vdir / "generator.npz", **synthetic.make_hifigan_state_dict(c["voc_hp"], seed=1234))
    tts = larynx_amd.load_tts_model(TextToSpeechType.GLOW_TTS, gdir, backend=InferenceBackend.HIP, half=half)
    voc = larynx_amd.load_vocoder_model(VocoderType.HIFI_GAN, vdir, backend=InferenceBackend.HIP, half=half)
    assert voc.precision == (ffi.PRECISION_F16 if half else ffi.PRECISION_F32)
    setattr(tts, "phoneme_to_id", {"_": 0})
    setattr(tts, "audio_settings", ljspeech_audio_settings())
    settings = {"noise_scale": float(c["noise_scale"]), "length_scale": float(c["length_scale"]), "noise": c["noise"]}
    audio = larynx_amd.sentence_task("golden", [int(i) for i in c["ids"]], getattr(tts, "audio_settings"), tts, settings, voc, None)
    assert audio.dtype == np.int16 and audio.shape == c["wav_i16"].shape
    d = int(np.abs(audio.astype(np.int32) - c["wav_i16"].astype(np.int32)).max())
    assert d <= (int(load_glow_half_reference()["ljspeech_high_S120"]["both_half_i16"]) if half else 1), d
    if half:
        assert d > 1  # the fp16 modes really ran
