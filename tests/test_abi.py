"""The C-ABI library builds for gfx950, loads, and exports exactly what
include/mi355tts.h declares (no compute calls — there is no GPU here)."""
import ctypes
import re
from pathlib import Path

import pytest

from larynx_amd import ffi
from larynx_amd.build import build

REPO = Path(__file__).resolve().parent.parent


def header_symbols():
    text = (REPO / "include" / "mi355tts.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi355tts_[a-z0-9_]+)\s*\(", text)))


def test_ffi_table_matches_header():
    assert header_symbols() == list(ffi.EXPORTED_SYMBOLS)


def test_library_builds_and_exports_every_symbol():
    lib_path = build()
    lib = ctypes.CDLL(str(lib_path))
    for name in header_symbols():
        assert hasattr(lib, name), name
    lib.mi355tts_abi_version.restype = ctypes.c_int
    assert lib.mi355tts_abi_version() == 2


def test_manifest_matches_reference_checkpoint_keys():
    """Every manifest entry resolves against a reference-format state-dict
    (weight_g/weight_v folded, 4x4 inverted) and covers every checkpoint tensor."""
    from larynx_amd import hparams as HP
    from larynx_amd import synthetic, weights

    lib = ffi.load_library(build())
    for hp, sd, conv in (
        (HP.LJSPEECH, synthetic.make_glow_state_dict(HP.LJSPEECH), ffi.glow_hparams_c),
        (HP.HIFIGAN_LOW, synthetic.make_hifigan_state_dict(HP.HIFIGAN_LOW), ffi.hifigan_hparams_c),
        (HP.HIFIGAN_HIGH, synthetic.make_hifigan_state_dict(HP.HIFIGAN_HIGH), ffi.hifigan_hparams_c),
    ):
        man = ffi.manifest(lib, conv(hp))
        blob = weights.build_blob(man, sd)
        assert blob.size == sum(n for _, n in man)
        used = set()
        for name, _ in man:
            base = name[:-4] if name.endswith("_inv") else name
            if base in sd:
                used.add(base)
            else:
                used.update({base[: -len(".weight")] + ".weight_g", base[: -len(".weight")] + ".weight_v"})
        assert used == set(sd), set(sd) ^ used


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(FileNotFoundError):
        ffi.load_library(tmp_path / "libmi355tts.so")


def test_every_shipped_voice_and_vocoder_config_is_accepted():
    """All 51 voice configs and the three vocoder configs of the reference
    (tests/golden/reference_configs.json, made by oracle/make_config_fixture.py) parse into
    hyper-parameters the library accepts: its manifest enumerates tensors for each."""
    import json

    from larynx_amd import hparams as HP
    from larynx_amd.audio import AudioSettings

    lib = ffi.load_library(build())
    cfgs = json.loads((REPO / "tests" / "golden" / "reference_configs.json").read_text())
    assert len(cfgs["voices"]) == 51 and set(cfgs["vocoders"]) == {"universal_large", "vctk_medium", "vctk_small"}
    symbols = set()
    for name, cfg in cfgs["voices"].items():
        hp = HP.GlowHParams.from_config(cfg)
        man = ffi.manifest(lib, ffi.glow_hparams_c(hp))
        assert sum(n for _, n in man) > 20_000_000, name  # ~28.5 M parameters each
        symbols.add(hp.num_symbols)
        # the audio block feeds the fused mel transforms
        known = {k: v for k, v in cfg["audio"].items() if k in AudioSettings.__dataclass_fields__}
        ffi.audio_settings_c(AudioSettings(**known))
    assert symbols == {38, 41, 42, 44, 46, 52, 54, 58}  # the vocabularies the voices use
    params = {}
    for name, cfg in cfgs["vocoders"].items():
        hp = HP.HifiGanHParams.from_config(cfg)
        params[name] = sum(n for _, n in ffi.manifest(lib, ffi.hifigan_hparams_c(hp)))
    # parameter counts after weight-norm folding
    assert params == {"universal_large": 13_926_017, "vctk_medium": 925_985, "vctk_small": 1_462_273}
