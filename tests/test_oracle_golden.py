"""The CPU oracle against the committed golden vectors, which are outputs of the
REFERENCE's own torch modules (`oracle/make_golden.py`, run where /root/reference
exists).  Runs everywhere, no GPU, no torch."""
import json

import numpy as np
import pytest

from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from oracle import audio_np, glow_tts_np, hifi_gan_np
from tests.golden_util import CASES, GOLDEN, load_case

_cache = {}


def _glow_sd(hp):
    if hp not in _cache:
        _cache[hp] = synthetic.make_glow_state_dict(hp, seed=1234)
    return _cache[hp]


def test_golden_report_shows_oracle_pinned():
    rep = json.loads((GOLDEN / "oracle_vs_reference.json").read_text())
    assert {k for k in rep if "/" not in k} == set(CASES)
    assert sum(k.startswith("batch8/") for k in rep) == 8
    for name, e in rep.items():
        assert e["mel"] < 2e-5 and e["wav_rms"] < 5e-6 and e["i16"] <= 1, (name, e)


@pytest.mark.parametrize("name", CASES)
def test_oracle_glow_reproduces_reference(name):
    c = load_case(name)
    taps = {}
    mel = glow_tts_np.glow_tts_infer(_glow_sd(c["glow_hp"]), c["glow_hp"], c["ids"], c["noise"], float(c["noise_scale"]), float(c["length_scale"]), taps)
    assert mel.shape == c["mel"].shape  # frame count is an integer parity item
    np.testing.assert_allclose(taps["logw"], c["logw"], atol=2e-5)
    np.testing.assert_allclose(mel, c["mel"], atol=2e-5)
    voc = audio_np.mel_to_vocoder_input(mel, ljspeech_audio_settings())
    np.testing.assert_allclose(voc, c["mel_voc"], atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("name", [n for n in CASES if "high" not in n or "short" in n])
def test_oracle_hifigan_reproduces_reference(name):
    c = load_case(name)
    vsd = synthetic.make_hifigan_state_dict(c["voc_hp"], seed=1234)
    wav = hifi_gan_np.hifigan_infer(vsd, c["voc_hp"], c["mel_voc"])
    st = int(c["wav_stride"])
    assert np.sqrt(np.mean((wav[::st] - c["wav"]) ** 2)) < 5e-6
    i16 = audio_np.audio_float_to_int16(wav)[::st]
    assert np.abs(i16.astype(np.int32) - c["wav_i16"].astype(np.int32)).max() <= 1


def test_duration_expansion_edge_cases():
    # frame -> id map of `generate_path` (glow_tts/utils.py:99-115), odd total truncated
    logw = np.log(np.array([1.0, 2.5, 0.2, 3.0], np.float32))
    w_ceil, F, idx = glow_tts_np.durations_to_frames(logw, 1.0, 2)
    assert list(w_ceil) == [1, 3, 1, 3] and F == 8
    assert list(idx) == [0, 1, 1, 1, 2, 3, 3, 3]
    _, F, idx = glow_tts_np.durations_to_frames(np.log(np.array([1.0, 2.0], np.float32)), 1.0, 2)
    assert F == 2 and list(idx) == [0, 1]  # 3 frames -> truncated to 2


def test_torch_operator_port_matches_numpy_oracle():
    """bench.py times this variant as the CPU baseline; it must be the same function."""
    pytest.importorskip("torch")
    from larynx_amd import hparams as HP
    from oracle import hifi_gan_torch

    for hp in (HP.HIFIGAN_MEDIUM, HP.HIFIGAN_LOW):
        vsd = synthetic.make_hifigan_state_dict(hp, seed=1234)
        mel = (np.random.default_rng(0).standard_normal((80, 40)) * 3).astype(np.float32)
        a = hifi_gan_torch.hifigan_infer_torch(vsd, hp, mel)
        b = hifi_gan_np.hifigan_infer(vsd, hp, mel)
        assert np.sqrt(np.mean((a - b) ** 2)) < 2e-6


def test_torch_glow_port_matches_numpy_oracle_and_the_reference_golden():
    """The GlowTTS half of bench.py's cpu_baseline (oracle/glow_tts_torch.py: torch CPU operators, the arithmetic of the
    reference's `--backend pytorch`) is the same function as the numpy oracle — and reproduces a golden mel the
    reference's own modules produced."""
    pytest.importorskip("torch")
    from larynx_amd import hparams as HP
    from oracle import glow_tts_torch
    from tests.golden_util import load_case

    hp = HP.TINY_GLOW
    sd = synthetic.make_glow_state_dict(hp, seed=5)
    ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(3), 23, hp.num_symbols)
    noise = np.random.default_rng(1).standard_normal((hp.mel_channels, 16 * 23 + 64)).astype(np.float32)
    for ns, ls in ((0.667, 1.0), (0.0, 1.3)):
        a = glow_tts_np.glow_tts_infer(sd, hp, ids, noise, ns, ls)
        b = glow_tts_torch.glow_tts_infer_torch(sd, hp, ids, noise, ns, ls)
        assert a.shape == b.shape and np.abs(a - b).max() < 5e-6
    c = load_case("ljspeech_high_short5")
    gsd = synthetic.make_glow_state_dict(c["glow_hp"], seed=1234)
    mel = glow_tts_torch.glow_tts_infer_torch(gsd, c["glow_hp"], c["ids"], c["noise"], float(c["noise_scale"]), float(c["length_scale"]))
    assert mel.shape == c["mel"].shape
    np.testing.assert_allclose(mel, c["mel"], atol=2e-5)


def test_oracle_reproduces_config4_batch_rows():
    """BASELINE config 4 golden (thorsten + 'medium', 8 rows, each through the reference at
    B = 1): the oracle on the three shortest rows, GlowTTS and vocoder."""
    from tests.golden_util import load_batch8

    c = load_batch8()
    gsd = synthetic.make_glow_state_dict(c["glow_hp"], seed=1234)
    vsd = synthetic.make_hifigan_state_dict(c["voc_hp"], seed=1234)
    s = ljspeech_audio_settings()
    for b in (0, 1, 2):
        mel = glow_tts_np.glow_tts_infer(gsd, c["glow_hp"], c["ids"][b], c["noise"][b], c["noise_scale"], c["length_scale"])
        assert mel.shape == c["mel"][b].shape
        np.testing.assert_allclose(mel, c["mel"][b], atol=2e-5)
        wav = hifi_gan_np.hifigan_infer(vsd, c["voc_hp"], audio_np.mel_to_vocoder_input(c["mel"][b], s))
        assert np.sqrt(np.mean((wav - c["wav"][b]) ** 2)) < 5e-6


def test_weight_norm_folding_matches_torch_to_float32_roundoff():
    """`weights.fold_weight_norm` folds in float64 and rounds once; the reference's
    `remove_weight_norm` is `torch._weight_norm(v, g, 0)` in float32.  On tensors of real-checkpoint shape and
    magnitude the two agree to float32 round-off (a few ulp: torch accumulates the norm in float32), so the choice cannot
    move parity."""
    torch = pytest.importorskip("torch")
    from larynx_amd.weights import fold_weight_norm

    rng = np.random.default_rng(7)
    for shape, scale in (((512, 80, 7), 0.02), ((256, 256, 11), 0.5), ((384, 192, 5), 3.0), ((192, 80, 1), 1e-3)):
        v = (rng.standard_normal(shape) * scale).astype(np.float32)
        g = np.abs(rng.standard_normal((shape[0], 1, 1)) * scale * 10 + 0.1).astype(np.float32)
        ours = fold_weight_norm(g, v)
        ref = torch._weight_norm(torch.from_numpy(v), torch.from_numpy(g), 0).numpy()
        ulp = np.abs(ours - ref) / np.maximum(np.spacing(np.abs(ref)), 1e-45)
        assert ulp.max() <= 4.0, (shape, float(ulp.max()))
