"""Audio settings of a voice — the host-side mirror of the reference's
`AudioSettings` dataclass (`larynx/audio.py:26-50`), fields and defaults kept so
`AudioSettings(**config["audio"])` (`larynx/__init__.py:352-356`) keeps working.

The three mel transforms `_sentence_task` applies between the two networks
(`larynx/__init__.py:242-249`) are NOT computed here: they are fused into the
HIP path (kernel `mel_finalize`, csrc/glow_kernels.h) and this class only
carries their parameters across the C ABI.
"""
from __future__ import annotations

import typing
from dataclasses import dataclass


@dataclass
class AudioSettings:
    # STFT settings
    filter_length: int = 1024
    hop_length: int = 256
    win_length: int = 256
    mel_channels: int = 80
    sample_rate: int = 22050
    sample_bytes: int = 2
    channels: int = 1
    mel_fmin: float = 0.0
    mel_fmax: typing.Optional[float] = 8000.0
    ref_level_db: float = 20.0
    spec_gain: float = 1.0

    # Normalization
    signal_norm: bool = False
    min_level_db: float = -100.0
    max_norm: float = 4.0
    clip_norm: bool = True
    symmetric_norm: bool = True
    do_dynamic_range_compression: bool = True
    convert_db_to_amp: bool = True


def ljspeech_audio_settings() -> AudioSettings:
    """`local/en-us/ljspeech-glow_tts/config.json:17-36`."""
    return AudioSettings(
        win_length=1024,
        signal_norm=True,
        min_level_db=-100.0,
        max_norm=1.0,
        ref_level_db=20.0,
        spec_gain=1.0,
    )
