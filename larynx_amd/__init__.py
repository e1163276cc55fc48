"""larynx_amd — the MI355X-native forward path of Larynx TTS.

phoneme ids -> GlowTTS -> mel transform -> HiFi-GAN -> waveform as hand-written
gfx950 HIP kernels behind a C ABI (`include/mi355tts.h`), wrapped in the
reference's own model interface.  Text handling (gruut, SSML), the CLI and the
HTTP server stay in Larynx; see INTEGRATION.md for the two edit points.
"""
from __future__ import annotations

import logging
import time
import typing
from concurrent.futures import Executor, ThreadPoolExecutor
from pathlib import Path

import numpy as np

from .audio import AudioSettings
from .constants import (
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

_LOGGER = logging.getLogger("larynx_amd")

__all__ = [
    "AudioSettings", "InferenceBackend", "TextToSpeechResult", "load_tts_model", "load_vocoder_model",
    "sentence_task", "phonemes_to_speech",
]


def load_tts_model(model_type, model_path, backend=InferenceBackend.HIP, no_optimizations: bool = False,
                   use_cuda: bool = True, half: bool = False, **kwargs) -> TextToSpeechModel:
    """Same signature as `larynx.load_tts_model` (`larynx/__init__.py:379-407`)."""
    config = TextToSpeechModelConfig(model_path=Path(model_path), use_cuda=use_cuda, half=half, backend=backend)
    if model_type == TextToSpeechType.GLOW_TTS:
        from .glow_tts import HipGlowTextToSpeech

        return HipGlowTextToSpeech(config, **kwargs)
    raise ValueError(f"Unknown text to speech model type: {model_type}")


def load_vocoder_model(model_type, model_path, backend=InferenceBackend.HIP, no_optimizations: bool = False,
                       use_cuda: bool = True, half: bool = False, denoiser_strength: float = 0.0,
                       executor: typing.Optional[Executor] = None, **kwargs) -> VocoderModel:
    """Same signature as `larynx.load_vocoder_model` (`larynx/__init__.py:472-508`)."""
    config = VocoderModelConfig(model_path=Path(model_path), use_cuda=use_cuda, half=half,
                                denoiser_strength=denoiser_strength, backend=backend)
    if model_type == VocoderType.HIFI_GAN:
        from .hifi_gan import HipHiFiGanVocoder

        return HipHiFiGanVocoder(config, executor=executor, **kwargs)
    raise ValueError(f"Unknown vocoder model type: {model_type}")


def sentence_task(text: str, phoneme_ids, audio_settings, tts_model, tts_settings, vocoder_model, vocoder_settings,
                  pause_before_ms: int = 0, pause_after_ms: int = 0) -> np.ndarray:
    """The per-sentence hot loop, argument for argument the reference's
    `_sentence_task` (`larynx/__init__.py:214-285`): GlowTTS, the mel transforms
    (numpy here only if the model did not already fuse them), the vocoder, the
    three debug log lines and the SSML pause padding."""
    t0 = time.perf_counter()
    mels = tts_model.phonemes_to_mels(phoneme_ids, settings=tts_settings)
    t1 = time.perf_counter()
    _LOGGER.debug("Got mels in %s second(s) (shape=%s, text='%s')", t1 - t0, getattr(mels, "shape", None), text)
    if audio_settings is not None and (audio_settings.signal_norm or audio_settings.convert_db_to_amp
                                       or audio_settings.do_dynamic_range_compression):
        # a non-fused TextToSpeechModel (e.g. the reference's own GlowTextToSpeech feeding the HIP
        # vocoder): the three transforms still run in-kernel, applied while wrapping the array
        from .glow_tts import mels_as_numpy

        mels = vocoder_model.engine.mel_from_numpy(mels_as_numpy(mels), audio_settings=audio_settings)
    sample_rate = audio_settings.sample_rate if audio_settings is not None else 22050
    before = max(0, (pause_before_ms * sample_rate) // 1000)
    after = max(0, (pause_after_ms * sample_rate) // 1000)
    t2 = time.perf_counter()
    padded = getattr(vocoder_model, "mels_to_audio_padded", None)
    if padded is not None and (before or after):
        # SSML pauses written by the device's int16 kernel instead of np.pad (same samples)
        audio = padded(mels, vocoder_settings, before, after)
        before = after = 0
    else:
        audio = vocoder_model.mels_to_audio(mels, settings=vocoder_settings)
    t3 = time.perf_counter()
    _LOGGER.debug("Got audio in %s second(s) (shape=%s, text='%s')", t3 - t2, audio.shape, text)
    dur = audio.shape[-1] / sample_rate
    _LOGGER.debug("Real-time factor: %0.2f (infer=%0.2f sec, audio=%0.2f sec)", (t3 - t0) / dur if dur > 0 else 0.0, t3 - t0, dur)
    if before or after:
        audio = np.pad(audio, pad_width=(before, after), constant_values=0)
    return audio


def _ensure_pool_workers(executor, *models):
    """One worker per pool thread (+ a spare) on the models' engine before the first sentence is submitted: the engine then knows
    which worker streams share a hardware queue and spreads the sentences' calls evenly (Engine.ensure_workers)."""
    n = getattr(executor, "_max_workers", None)
    if not isinstance(n, int) or n < 2:
        return
    for m in models:
        eng = getattr(m, "engine", None)
        if eng is not None and hasattr(eng, "ensure_workers"):
            eng.ensure_workers(min(n, 32) + 1)


def phonemes_to_speech(sentences: typing.Iterable[typing.Tuple[str, typing.Sequence[int]]], tts_model, vocoder_model,
                       tts_settings: typing.Optional[SettingsType] = None,
                       vocoder_settings: typing.Optional[SettingsType] = None,
                       executor: typing.Optional[Executor] = None) -> typing.Iterable[TextToSpeechResult]:
    """`text_to_speech` (`larynx/__init__.py:47-190`) from the point where gruut /
    phonemes2ids have produced ids: one task per sentence on an executor, results
    yielded in submission order."""
    own = executor is None
    executor = executor or ThreadPoolExecutor()
    _ensure_pool_workers(executor, tts_model, vocoder_model)
    try:
        audio_settings = getattr(tts_model, "audio_settings", None)
        futures = []
        for text, ids in sentences:
            fut = executor.submit(sentence_task, text, np.asarray(ids, np.int64), audio_settings, tts_model, tts_settings,
                                  vocoder_model, vocoder_settings)
            futures.append((text, fut))
        sr = audio_settings.sample_rate if audio_settings is not None else 22050
        for text, fut in futures:
            yield TextToSpeechResult(text=text, audio=fut.result(), sample_rate=sr)
    finally:
        if own:
            executor.shutdown(wait=True)
