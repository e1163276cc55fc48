"""The reference's model-facing types, re-stated for the HIP backend.

A Larynx maintainer swaps `from larynx.constants import ...` for this module and
nothing else changes at the call sites: same class names, same field names, same
`phonemes_to_mels(phoneme_ids, settings)` / `mels_to_audio(mels, settings)` calling
convention (rhasspy/larynx `larynx/constants.py:19-114`).  Additions: the `HIP`
backend member, and `session_options` is optional (there is no ONNX Runtime here).
"""
from __future__ import annotations

import abc
import dataclasses
import enum
import pathlib
import typing

import numpy as np

ARRAY_OR_TENSOR = typing.Union[np.ndarray, typing.Any]
SettingsType = typing.Dict[str, typing.Any]


def _str_enum(name: str, **members: str):
    """String-valued enum (`Enum(str)` semantics: members compare equal to their value)."""
    return enum.Enum(name, members, type=str, module=__name__)


TextToSpeechType = _str_enum("TextToSpeechType", TACOTRON2="tacotron2", GLOW_TTS="glow_tts")
VocoderType = _str_enum("VocoderType", GRIFFIN_LIM="griffin_lim", HIFI_GAN="hifi_gan", WAVEGLOW="waveglow")
VocoderQuality = _str_enum("VocoderQuality", HIGH="high", MEDIUM="medium", LOW="low")
InferenceBackend = _str_enum("InferenceBackend", ONNX="onnx", PYTORCH="pytorch", HIP="hip")


@dataclasses.dataclass
class _ModelLocation:
    """What both model kinds are constructed from (the reference keeps two copies of
    these fields, `constants.py:51-59` and `:78-87`)."""

    model_path: pathlib.Path
    session_options: typing.Any = None  # accepted and ignored: reference call sites always pass one
    use_cuda: bool = True
    half: bool = False  # the HIP backend is fp32; True is rejected by the model classes
    backend: typing.Optional[InferenceBackend] = None


@dataclasses.dataclass
class TextToSpeechModelConfig(_ModelLocation):
    pass


@dataclasses.dataclass
class VocoderModelConfig(_ModelLocation):
    denoiser_strength: float = 0.0


class TextToSpeechModel(abc.ABC):
    """ids -> mel.  Implemented by `larynx_amd.glow_tts.HipGlowTextToSpeech`."""

    def __init__(self, config: TextToSpeechModelConfig):
        self.config = config

    @abc.abstractmethod
    def phonemes_to_mels(self, phoneme_ids: np.ndarray, settings: typing.Optional[SettingsType] = None) -> ARRAY_OR_TENSOR:
        ...


class VocoderModel(abc.ABC):
    """mel -> int16 audio.  Implemented by `larynx_amd.hifi_gan.HipHiFiGanVocoder`."""

    def __init__(self, config: VocoderModelConfig):
        self.config = config

    @abc.abstractmethod
    def mels_to_audio(self, mels: ARRAY_OR_TENSOR, settings: typing.Optional[SettingsType] = None) -> np.ndarray:
        ...


@dataclasses.dataclass
class TextToSpeechResult:
    """One sentence of `text_to_speech` output (`constants.py:106-114`)."""

    text: str
    audio: typing.Optional[np.ndarray]
    sample_rate: int
    marks_before: typing.Optional[typing.Sequence[str]] = None
    marks_after: typing.Optional[typing.Sequence[str]] = None
