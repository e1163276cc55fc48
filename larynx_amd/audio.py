"""Audio settings of a voice, as the HIP path needs them.

`AudioSettings` accepts exactly the keys a voice's `config.json["audio"]` carries, so
`AudioSettings(**config["audio"])` (rhasspy/larynx `larynx/__init__.py:352-356`) keeps
working, and defaults match the reference's record (`larynx/audio.py:26-50`).  Unlike the
reference class it has no numpy methods: the three mel transforms `_sentence_task`
applies between the two networks (`larynx/__init__.py:242-249`) run inside the HIP path
(kernel `mel_finalize`, csrc/small_kernels.h); this record only carries their parameters
across the C ABI (`mi355tts_audio_settings`).
"""
from __future__ import annotations

import dataclasses
import typing

# (name, type, default) — STFT geometry first, then the normalisation switches the
# kernel reads.  Only the second group influences the arithmetic.
_STFT_FIELDS = (
    ("filter_length", int, 1024), ("hop_length", int, 256), ("win_length", int, 256),
    ("mel_channels", int, 80), ("sample_rate", int, 22050), ("sample_bytes", int, 2), ("channels", int, 1),
    ("mel_fmin", float, 0.0), ("mel_fmax", typing.Optional[float], 8000.0),
)
_NORM_FIELDS = (
    ("ref_level_db", float, 20.0), ("spec_gain", float, 1.0), ("signal_norm", bool, False),
    ("min_level_db", float, -100.0), ("max_norm", float, 4.0), ("clip_norm", bool, True),
    ("symmetric_norm", bool, True), ("do_dynamic_range_compression", bool, True), ("convert_db_to_amp", bool, True),
)

AudioSettings = dataclasses.make_dataclass(
    "AudioSettings",
    [(n, t, dataclasses.field(default=d)) for n, t, d in _STFT_FIELDS + _NORM_FIELDS],
    namespace={"__doc__": "Mel (de)normalisation parameters of a voice; see the module docstring."},
)
AudioSettings.__module__ = __name__


def ljspeech_audio_settings() -> "AudioSettings":
    """The `audio` block of `local/en-us/ljspeech-glow_tts/config.json` (thorsten's is identical)."""
    return AudioSettings(win_length=1024, signal_norm=True, max_norm=1.0)
