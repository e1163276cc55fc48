"""Multi-speaker GlowTTS voices: the reference's `speaker_id` setting (larynx/glow_tts.py:116-130 -> `g` of
`FlowGenerator.forward`, glow_tts/models.py:318-319; WaveNet conditioning layers.py:109-113, 141-154; the duration
predictor's concatenated input models.py:114-116, 128-132).

The golden vectors (tests/golden/multispeaker/ljspeech_4speakers.npz, made by oracle/make_golden_speakers.py from the reference's own
torch modules) pin the oracle on the CPU; the HIP path is compared with them through the C ABI on the GPU, and — at
shrunk hyper-parameters — with the oracle on the CPU emulator build."""
import dataclasses
import json
from pathlib import Path

import numpy as np
import pytest

from larynx_amd import ffi
from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from oracle import audio_np, glow_tts_np

GOLDEN = Path(__file__).parent / "golden" / "multispeaker" / "ljspeech_4speakers.npz"
MEL_TOL = 1e-3  # north_star: mel +-1e-3 max-abs, identical frame counts


def golden():
    z = np.load(GOLDEN)
    hp = HP.GlowHParams.from_config(json.loads(str(z["glow"])))
    names = json.loads(str(z["names"]))
    noise = lambda ids: np.random.default_rng(int(z["noise_seed"])).standard_normal((hp.mel_channels, 16 * len(ids) + 64)).astype(np.float32)
    return z, hp, names, noise


def test_config_round_trip_and_validation():
    z, hp, _, _ = golden()
    assert hp.n_speakers == 4 and hp.gin_channels == 48
    assert HP.GlowHParams.from_config(hp.to_config()) == hp
    assert HP.GlowHParams.from_config(HP.LJSPEECH.to_config()) == HP.LJSPEECH  # single speaker: n_speakers 1, gin 0
    cfg = hp.to_config()
    cfg["model"]["gin_channels"] = 0
    with pytest.raises(ValueError):
        HP.GlowHParams.from_config(cfg)
    cfg = HP.LJSPEECH.to_config()
    cfg["model"]["gin_channels"] = 16  # conditioning layers nothing can feed (no emb_g): rejected
    with pytest.raises(ValueError):
        HP.GlowHParams.from_config(cfg)


def test_oracle_against_the_references_multispeaker_output():
    """The numpy restatement reproduces the reference's FlowGenerator with g = speaker (committed golden vectors)."""
    z, hp, names, noise = golden()
    sd = synthetic.make_glow_state_dict(hp, seed=1234)
    # the speaker tensors ride on top of the single-speaker weights: same numbers for everything else
    base = synthetic.make_glow_state_dict(HP.LJSPEECH, seed=1234)
    assert all(np.array_equal(sd[k], v) for k, v in base.items() if k != "encoder.proj_w.conv_1.weight")
    for name in names:
        ids = z[f"{name}.ids"]
        taps = {}
        mel = glow_tts_np.glow_tts_infer(sd, hp, ids, noise(ids), float(z[f"{name}.noise_scale"]), float(z[f"{name}.length_scale"]), taps,
                                         speaker_id=int(z[f"{name}.speaker"]))
        assert mel.shape == z[f"{name}.mel"].shape
        assert np.abs(mel - z[f"{name}.mel"]).max() < 2e-4 and np.abs(taps["logw"] - z[f"{name}.logw"]).max() < 1e-4
        voc = audio_np.mel_to_vocoder_input(mel, ljspeech_audio_settings())
        assert np.abs(voc - z[f"{name}.mel_voc"]).max() < 5e-4
    # another speaker is another voice (durations included), and the oracle refuses what the reference cannot run
    assert z["echo_s0.mel"].shape != z["echo_s2.mel"].shape
    with pytest.raises(ValueError):
        glow_tts_np.glow_tts_infer(sd, hp, z["echo_s0.ids"], None, 0.0, 1.0)
    with pytest.raises(ValueError):
        glow_tts_np.glow_tts_infer(base, HP.LJSPEECH, z["echo_s0.ids"], None, 0.0, 1.0, speaker_id=1)


TINY_MULTI = dataclasses.replace(HP.TINY_GLOW, n_speakers=3, gin_channels=20)


def check_against_oracle(eng, hp, sd, g, rows, speakers, noise_scale=0.667, length_scale=1.0, tol=MEL_TOL):
    rng = np.random.default_rng(5)
    noise = rng.standard_normal((len(rows), hp.mel_channels, 16 * max(len(r) for r in rows) + 64)).astype(np.float32)
    mb = eng.glow_infer(g, rows, noise_scale, length_scale, noise=noise, speaker_ids=speakers)
    got = mb.numpy("raw")
    for b, (ids, spk) in enumerate(zip(rows, speakers)):
        ref = glow_tts_np.glow_tts_infer(sd, hp, ids, noise[b], noise_scale, length_scale, speaker_id=spk)
        assert int(mb.frames[b]) == ref.shape[1]
        assert np.abs(got[b, :, : ref.shape[1]] - ref).max() < tol
        assert np.all(got[b, :, ref.shape[1]:] == 0)
    return mb


def test_emulator_build_against_the_oracle(emu_engine):
    """The C-ABI path (CPU emulator build of the same kernels) with speakers: batch 1, a ragged batch with a different speaker
    per row, the fused synthesize call, the fallback schedule (gate16 / glow_fuse off), and the argument checks."""
    hp = TINY_MULTI
    sd = synthetic.make_glow_state_dict(hp, seed=21)
    g = emu_engine.load_glow(hp, sd)
    rng = np.random.default_rng(3)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (17, 9, 26)]
    try:
        check_against_oracle(emu_engine, hp, sd, g, rows[:1], [2])
        check_against_oracle(emu_engine, hp, sd, g, rows, [0, 2, 1], length_scale=0.8)
        for opt in ("gate16", "glow_fuse"):
            emu_engine.set_option(opt, 0)
            try:
                check_against_oracle(emu_engine, hp, sd, g, rows, [1, 1, 0])
            finally:
                emu_engine.set_option(opt, 1)
        # one int = every row; the speakers matter
        a = emu_engine.glow_infer(g, rows[0], 0.0, 1.0, speaker_ids=0)
        b = emu_engine.glow_infer(g, rows[0], 0.0, 1.0, speaker_ids=1)
        assert a.numpy("raw").shape != b.numpy("raw").shape or np.abs(a.numpy("raw") - b.numpy("raw")).max() > 1e-2
        # the fused call = the two calls
        vhp = HP.TINY_HIFIGAN
        v = emu_engine.load_hifigan(vhp, synthetic.make_hifigan_state_dict(vhp, seed=21))
        s = ljspeech_audio_settings()
        mel = emu_engine.glow_infer(g, rows, 0.667, 1.0, seed=5, audio_settings=s, speaker_ids=[2, 0, 1])
        _, i2 = emu_engine.hifigan_infer(v, mel)
        frames, _, i1 = emu_engine.synthesize(g, v, rows, 0.667, 1.0, seed=5, audio_settings=s, speaker_ids=[2, 0, 1])
        assert np.array_equal(frames, mel.frames) and np.array_equal(i1, i2)
        emu_engine.unload(v)
        # what the reference cannot run is an error here too
        with pytest.raises(ffi.Mi355ttsError, match="speaker"):
            emu_engine.glow_infer(g, rows[0], 0.0, 1.0)
        with pytest.raises(ffi.Mi355ttsError, match="outside"):
            emu_engine.glow_infer(g, rows[0], 0.0, 1.0, speaker_ids=3)
        with pytest.raises(ValueError):
            emu_engine.glow_infer(g, rows, 0.0, 1.0, speaker_ids=[0, 1])
        single = emu_engine.load_glow(HP.TINY_GLOW, synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=21))
        with pytest.raises(ffi.Mi355ttsError, match="single-speaker"):
            emu_engine.glow_infer(single, rows[0], 0.0, 1.0, speaker_ids=0)
        emu_engine.unload(single)
        bad = dataclasses.replace(HP.TINY_GLOW, n_speakers=1, gin_channels=8)
        with pytest.raises(ffi.Mi355ttsError, match="gin_channels"):
            ffi.manifest(emu_engine.lib, ffi.glow_hparams_c(bad))
    finally:
        emu_engine.unload(g)


def test_reference_style_model_object_takes_the_speaker_setting(emu_engine, emu_library):
    """`HipGlowTextToSpeech.phonemes_to_mels(ids, {"speaker_id": n})` — the reference's settings key."""
    from larynx_amd.constants import TextToSpeechModelConfig
    from larynx_amd.glow_tts import HipGlowTextToSpeech

    hp = TINY_MULTI
    sd = synthetic.make_glow_state_dict(hp, seed=21)
    cfg = TextToSpeechModelConfig(model_path=Path("unused"))
    tts = HipGlowTextToSpeech(cfg, library_path=emu_library, state_dict=sd, model_config=hp.to_config())
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(8), 12, hp.num_symbols)
    mel = tts.phonemes_to_mels(ids, {"speaker_id": 2, "noise_scale": 0.0})
    ref = glow_tts_np.glow_tts_infer(sd, hp, ids, None, 0.0, 1.0, speaker_id=2)
    assert np.abs(mel.numpy("raw")[0] - ref).max() < MEL_TOL
    with pytest.raises(ffi.Mi355ttsError):
        tts.phonemes_to_mels(ids, {"noise_scale": 0.0})  # larynx/glow_tts.py:148 with g = None fails in the reference too


@pytest.mark.gpu
def test_multispeaker_golden_on_the_device(gpu_engine):
    """The reference's own multi-speaker output (LJSpeech architecture, 4 speakers, gin 48) through the C ABI on the GPU:
    mel within 1e-3, identical frame counts; batch 1 per case, then all cases as ONE ragged batch with a speaker per row."""
    z, hp, names, noise = golden()
    sd = synthetic.make_glow_state_dict(hp, seed=1234)
    g = gpu_engine.load_glow(hp, sd)
    s = ljspeech_audio_settings()
    try:
        for name in names:
            ids = z[f"{name}.ids"]
            mb = gpu_engine.glow_infer(g, ids, float(z[f"{name}.noise_scale"]), float(z[f"{name}.length_scale"]), noise=noise(ids),
                                       audio_settings=s, speaker_ids=int(z[f"{name}.speaker"]))
            ref = z[f"{name}.mel"]
            assert int(mb.frames[0]) == ref.shape[1]
            assert np.abs(mb.numpy("raw")[0, :, : ref.shape[1]] - ref).max() < MEL_TOL
            assert np.abs(mb.numpy("voc")[0, :, : ref.shape[1]] - z[f"{name}.mel_voc"]).max() < 5e-4 + 1e-3 * np.abs(z[f"{name}.mel_voc"]).max()
        # one padded batch: the two cases that share their scales, another speaker per row
        pair = ["echo_s0", "echo_s2"]
        rows = [z[f"{n}.ids"] for n in pair]
        nz = np.stack([noise(r) for r in rows])
        mb = gpu_engine.glow_infer(g, rows, 0.667, 1.0, noise=nz, speaker_ids=[int(z[f"{n}.speaker"]) for n in pair])
        for b, n in enumerate(pair):
            ref = z[f"{n}.mel"]
            assert int(mb.frames[b]) == ref.shape[1]
            assert np.abs(mb.numpy("raw")[b, :, : ref.shape[1]] - ref).max() < MEL_TOL
        # the fallback schedule takes the same speaker offsets
        gpu_engine.set_option("gate16", 0)
        try:
            ids = z["dave_s3.ids"]
            mb = gpu_engine.glow_infer(g, ids, float(z["dave_s3.noise_scale"]), float(z["dave_s3.length_scale"]), noise=noise(ids), speaker_ids=3)
            assert np.abs(mb.numpy("raw")[0, :, : z["dave_s3.mel"].shape[1]] - z["dave_s3.mel"]).max() < MEL_TOL
        finally:
            gpu_engine.set_option("gate16", 1)
        with pytest.raises(ffi.Mi355ttsError, match="speaker"):
            gpu_engine.glow_infer(g, z["echo_s0.ids"], 0.0, 1.0)
    finally:
        gpu_engine.unload(g)
