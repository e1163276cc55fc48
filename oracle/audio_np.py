"""TEST INFRASTRUCTURE — CPU oracle for the glue between the two networks.

Restates `AudioSettings.denormalize` / `db_to_amp` / `dynamic_range_compression`
(`/root/reference/larynx/audio.py:83-108`) as `_sentence_task` applies them
(`larynx/__init__.py:242-249`), and `audio_float_to_int16` (audio.py:118-125).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def mel_to_vocoder_input(mel: np.ndarray, s) -> np.ndarray:
    """`s` is any object with the AudioSettings fields (larynx_amd.audio.AudioSettings)."""
    x = np.asarray(mel, F32)
    if s.signal_norm:  # denormalize, audio.py:83-104
        if s.symmetric_norm:
            if s.clip_norm:
                x = np.clip(x, -s.max_norm, s.max_norm)
            x = ((x + s.max_norm) * -s.min_level_db / (2 * s.max_norm)) + s.min_level_db
        else:
            if s.clip_norm:
                x = np.clip(x, 0, s.max_norm)
            x = (x * -s.min_level_db / s.max_norm) + s.min_level_db
        x = x + s.ref_level_db
    if s.convert_db_to_amp:  # audio.py:58-59
        x = np.power(10.0, x / s.spec_gain)
    if s.do_dynamic_range_compression:  # audio.py:106-108
        x = np.log(np.clip(x, a_min=1e-5, a_max=None))
    return np.asarray(x, F32)


def audio_float_to_int16(audio: np.ndarray, max_wav_value: float = 32767.0) -> np.ndarray:
    """audio.py:118-125 (astype truncates toward zero)."""
    audio = np.asarray(audio, F32)
    peak = max(0.01, float(np.max(np.abs(audio)))) if audio.size else 0.01
    a = audio * F32(max_wav_value / peak)
    a = np.clip(a, -max_wav_value, max_wav_value)
    return a.astype("int16")
