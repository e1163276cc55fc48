"""Interface types of the hot path — the host-side mirror of `larynx/constants.py`.

Names, signatures and meanings follow the reference so a Larynx maintainer can
swap imports: the model ABCs (`constants.py:62-72`, `:90-100`), their config
records (`:51-59`, `:78-87`), the enums (`:19-45`) and the result record
(`:106-114`).  The one addition is `InferenceBackend.HIP`.
"""
from __future__ import annotations

import enum
import typing
from abc import ABC
from dataclasses import dataclass
from pathlib import Path

import numpy as np

ARRAY_OR_TENSOR = typing.Union[np.ndarray, typing.Any]
SettingsType = typing.Dict[str, typing.Any]


class TextToSpeechType(str, enum.Enum):
    TACOTRON2 = "tacotron2"
    GLOW_TTS = "glow_tts"


class VocoderType(str, enum.Enum):
    GRIFFIN_LIM = "griffin_lim"
    HIFI_GAN = "hifi_gan"
    WAVEGLOW = "waveglow"


class VocoderQuality(str, enum.Enum):
    HIGH = "high"
    MEDIUM = "medium"
    LOW = "low"


class InferenceBackend(str, enum.Enum):
    ONNX = "onnx"
    PYTORCH = "pytorch"
    HIP = "hip"  # this package: hand-written gfx950 kernels behind the C ABI


@dataclass
class TextToSpeechModelConfig:
    """`session_options` is accepted (and ignored) so reference call sites that
    always pass an onnxruntime.SessionOptions keep working."""

    model_path: Path
    session_options: typing.Any = None
    use_cuda: bool = True
    half: bool = False
    backend: typing.Optional[InferenceBackend] = None


class TextToSpeechModel(ABC):
    def __init__(self, config: TextToSpeechModelConfig):
        pass

    def phonemes_to_mels(self, phoneme_ids: np.ndarray, settings: typing.Optional[SettingsType] = None) -> ARRAY_OR_TENSOR:
        """Convert phoneme ids to mel spectrograms"""
        raise NotImplementedError


@dataclass
class VocoderModelConfig:
    model_path: Path
    session_options: typing.Any = None
    use_cuda: bool = True
    half: bool = False
    denoiser_strength: float = 0.0
    backend: typing.Optional[InferenceBackend] = None


class VocoderModel(ABC):
    def __init__(self, config: VocoderModelConfig):
        pass

    def mels_to_audio(self, mels: ARRAY_OR_TENSOR, settings: typing.Optional[SettingsType] = None) -> np.ndarray:
        """Convert mel spectrograms to WAV audio"""
        raise NotImplementedError


@dataclass
class TextToSpeechResult:
    text: str
    audio: typing.Optional[np.ndarray]
    sample_rate: int
    marks_before: typing.Optional[typing.Sequence[str]] = None
    marks_after: typing.Optional[typing.Sequence[str]] = None
