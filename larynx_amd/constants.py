"""Import-compatibility alias: `larynx.constants` -> `larynx_amd.constants`."""
from .interfaces import *  # noqa: F401,F403
from .interfaces import (  # noqa: F401
    ARRAY_OR_TENSOR,
    InferenceBackend,
    SettingsType,
    TextToSpeechModel,
    TextToSpeechModelConfig,
    TextToSpeechResult,
    TextToSpeechType,
    VocoderModel,
    VocoderModelConfig,
    VocoderQuality,
    VocoderType,
)
