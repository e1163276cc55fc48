"""GlowTTS text-to-speech model on the HIP backend — drop-in for
`larynx.glow_tts.GlowTextToSpeech` (`larynx/glow_tts.py:29-170`)."""
from __future__ import annotations

import copy
import itertools
import logging
import os
import typing

import numpy as np

from . import ffi
from .audio import AudioSettings
from .constants import ARRAY_OR_TENSOR, InferenceBackend, SettingsType, TextToSpeechModel, TextToSpeechModelConfig
from .engine import MelBatch
from .hparams import GlowHParams
from .runtime import find_checkpoint, get_engine, read_config
from .weights import load_state_dict

_LOGGER = logging.getLogger("glow_tts")


class HipGlowTextToSpeech(TextToSpeechModel):
    """Same constructor record, same `phonemes_to_mels(phoneme_ids, settings)` and
    the same settings keys (`noise_scale`, `length_scale`, `speaker_id`; `larynx/glow_tts.py:118-121`)
    as the reference class.  Differences a caller can see:

    * the returned object is a device-resident `MelBatch` (shape `[1, 80, F]`, like
      the reference's array) that `HipHiFiGanVocoder.mels_to_audio` consumes
      without a host round trip; `np.asarray(mel)` gives the reference's ndarray;
    * the three numpy mel transforms `_sentence_task` applies between the models
      (`larynx/__init__.py:242-249`) are fused into the HIP path, so the
      `audio_settings` attribute this object exposes to `_sentence_task` has those
      three switches off (the real ones are kept for the kernel);
    * extra settings `noise` (explicit N(0,1) tensor `[80, >=F]`, the parity mode)
      and `seed` (device RNG; default: a fresh seed per call, as the reference draws
      fresh noise per call) — the reference's noise is not reproducible from the
      host (`torch.randn_like`, `glow_tts/models.py:348`).
    """

    def __init__(self, config: TextToSpeechModelConfig, device: int = 0, library_path=None, state_dict=None,
                 model_config: typing.Optional[dict] = None):
        super().__init__(config)
        if config.backend not in (None, InferenceBackend.HIP):
            raise ValueError(f"Unknown backend: {config.backend}")
        # `half` (larynx/glow_tts.py:90-91 calls `.half()` on the model): the decoder's WaveNets run in fp16 (csrc/wn_f16.h) where
        # the library's kernel covers the geometry (the released voices'); the encoder and the durations stay f32
        self.engine = get_engine(device, library_path)
        cfg = model_config if model_config is not None else read_config(config.model_path)
        self.hparams = GlowHParams.from_config(cfg)
        if state_dict is None:
            ckpt = find_checkpoint(config.model_path)
            _LOGGER.debug("Loading GlowTTS checkpoint from %s", ckpt)
            names = [n for n, _ in ffi.manifest(self.engine.lib, ffi.glow_hparams_c(self.hparams))]
            state_dict = load_state_dict(ckpt, "model", manifest_names=names, n_split=self.hparams.n_split)
        self.model_id = self.engine.load_glow(self.hparams, state_dict)
        if config.half and self.engine.set_precision(self.model_id, ffi.PRECISION_F16) == ffi.PRECISION_NOOP:
            _LOGGER.debug("half: the acoustic model computes in f32 (the library reports the switch as a no-op for this geometry)")
        self.noise_scale = 0.667
        self.length_scale = 1.0
        self._audio_settings: typing.Optional[AudioSettings] = None
        if "audio" in cfg and "sample_rate" in cfg["audio"]:
            known = {k: v for k, v in cfg["audio"].items() if k in AudioSettings.__dataclass_fields__}
            self._audio_settings = AudioSettings(**known)
        self.phoneme_to_id: typing.Optional[typing.Dict[str, int]] = None
        # The reference draws fresh noise on every call (`torch.randn_like`, glow_tts/models.py:348):
        # without an explicit `seed` setting each call takes the next value of a per-model
        # counter that starts at a random 63-bit number (thread-safe: itertools.count is atomic).
        self._seeds = itertools.count(int.from_bytes(os.urandom(8), "little") >> 1)

    # -- `get_tts_model` does setattr(model, "audio_settings", ...) and `text_to_speech`
    #    reads it back with getattr (larynx/__init__.py:117-120, 362-363)
    @property
    def audio_settings(self) -> typing.Optional[AudioSettings]:
        if self._audio_settings is None:
            return None
        view = copy.copy(self._audio_settings)
        view.signal_norm = False
        view.convert_db_to_amp = False
        view.do_dynamic_range_compression = False
        return view

    @audio_settings.setter
    def audio_settings(self, value):
        self._audio_settings = value

    @property
    def kernel_audio_settings(self) -> typing.Optional[AudioSettings]:
        return self._audio_settings

    def phonemes_to_mels(self, phoneme_ids: np.ndarray, settings: typing.Optional[SettingsType] = None) -> ARRAY_OR_TENSOR:
        noise_scale, length_scale = self.noise_scale, self.length_scale
        noise, seed = None, None
        speaker_idx: typing.Optional[int] = None
        if settings:
            noise_scale = float(settings.get("noise_scale", noise_scale))
            length_scale = float(settings.get("length_scale", length_scale))
            speaker_idx = settings.get("speaker_id")  # larynx/glow_tts.py:121
            noise = settings.get("noise")
            seed = settings.get("seed")
        ids = np.asarray(phoneme_ids, dtype=np.int64).reshape(-1)
        if ids.size == 0:
            raise ValueError("empty phoneme id sequence")
        seed = next(self._seeds) if seed is None else int(seed)
        # multi-speaker voices (larynx/glow_tts.py:125-130, 148: `g = speaker_id`): the library rejects a missing speaker for a
        # multi-speaker voice and a speaker for a single-speaker one — both fail in the reference too
        return self.engine.glow_infer(
            self.model_id, ids, noise_scale, length_scale, noise=noise, seed=seed, audio_settings=self._audio_settings,
            speaker_ids=None if speaker_idx is None else int(speaker_idx),
        )


def mels_as_numpy(mels: typing.Union[MelBatch, np.ndarray]) -> np.ndarray:
    return mels.numpy("raw") if isinstance(mels, MelBatch) else np.asarray(mels)
