"""TEST INFRASTRUCTURE — CPU oracle for the HiFi-GAN output denoiser.

Restates `HiFiGanVocoder.denoise` / `maybe_init_denoiser`
(`/root/reference/larynx/hifi_gan.py:171-203`) and the STFT helpers it uses
(`larynx/audio.py:232-306`: `stft`, `istft`, `transform`, `inverse`), including
their quirks: symmetric `np.hanning` window on both sides with NO window-sum
normalisation, frames `range(0, N - 1024, 256)`, float64 arithmetic, output
length `frames*256 + 1024`.
"""
from __future__ import annotations

import numpy as np

FFT = 1024
HOP = 256


def transform(audio: np.ndarray):
    """audio float [N] -> magnitude, phase  [513, T]  (audio.py:292-306, 232-249)."""
    window = np.hanning(FFT)
    frames = np.array([np.fft.rfft(window * audio[i : i + FFT]) for i in range(0, len(audio) - FFT, HOP)])
    if frames.ndim != 2:
        raise ValueError("audio too short for the denoiser's STFT")
    spec = frames.T
    return np.abs(spec), np.arctan2(spec.imag, spec.real)


def inverse(magnitude: np.ndarray, phase: np.ndarray) -> np.ndarray:
    """audio.py:272-289, 252-269 (the reference goes through complex64)."""
    x = np.empty(magnitude.shape, np.complex64)
    x.real = magnitude * np.cos(phase)
    x.imag = magnitude * np.sin(phase)
    window = np.hanning(FFT)
    T = x.shape[1]
    out = np.zeros(T * HOP + FFT)
    for n in range(T):
        out[n * HOP : n * HOP + FFT] += window * np.real(np.fft.irfft(x[:, n]))
    return out


def bias_spectrum(generator, num_mels: int = 80) -> np.ndarray:
    """`maybe_init_denoiser` (hifi_gan.py:181-203): first STFT column of the
    generator's response to an all-zero mel of 88 frames.  `generator(mel)` is any
    callable mel [num_mels, F] -> waveform [F*hop]."""
    wav = generator(np.zeros((num_mels, 88), np.float32))
    mag, _ = transform(np.asarray(wav, np.float32))
    return mag[:, 0].copy()


def denoise(audio: np.ndarray, bias_spec: np.ndarray, strength: float) -> np.ndarray:
    """hifi_gan.py:171-179."""
    mag, phase = transform(np.asarray(audio, np.float32))
    mag = np.clip(mag - bias_spec[:, None] * strength, a_min=0.0, a_max=None)
    return inverse(mag, phase)
