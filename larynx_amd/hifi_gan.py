"""HiFi-GAN vocoder on the HIP backend — drop-in for
`larynx.hifi_gan.HiFiGanVocoder` (`larynx/hifi_gan.py:31-203`)."""
from __future__ import annotations

import logging
import typing
from concurrent.futures import Executor

import numpy as np

from .constants import ARRAY_OR_TENSOR, InferenceBackend, SettingsType, VocoderModel, VocoderModelConfig
from . import ffi
from .engine import MelBatch
from .hparams import HifiGanHParams
from .runtime import find_checkpoint, get_engine, read_config
from .weights import load_state_dict

_LOGGER = logging.getLogger("hifi_gan")


class HipHiFiGanVocoder(VocoderModel):
    """`mels_to_audio(mels, settings)` accepts what the reference accepts — a
    float32 `[1, 80, F]` array that already went through the AudioSettings
    transforms (`larynx/hifi_gan.py:145-150`) — or the device-resident `MelBatch`
    a `HipGlowTextToSpeech` returned, and gives back the same `int16 [N]`
    (`audio_float_to_int16(...).squeeze()`, :168-169).

    `denoiser_strength > 0` (config or per-call `settings`, :152-156) runs the
    reference's spectral-subtraction denoiser on the device (bias spectrum from an
    all-zero mel, :181-203), before the int16 conversion."""

    def __init__(self, config: VocoderModelConfig, executor: typing.Optional[Executor] = None, device: int = 0,
                 library_path=None, state_dict=None, model_config: typing.Optional[dict] = None):
        super().__init__(config)
        if config.backend not in (None, InferenceBackend.HIP):
            raise ValueError(f"Unknown backend: {config.backend}")
        self.engine = get_engine(device, library_path)
        cfg = model_config if model_config is not None else read_config(config.model_path)
        self.hparams = HifiGanHParams.from_config(cfg)
        self.mel_channels = self.hparams.num_mels
        if state_dict is None:
            ckpt = find_checkpoint(config.model_path)
            _LOGGER.debug("Loading HiFi-GAN checkpoint from %s", ckpt)
            names = [n for n, _ in ffi.manifest(self.engine.lib, ffi.hifigan_hparams_c(self.hparams))]
            state_dict = load_state_dict(ckpt, "generator", manifest_names=names)
        self.model_id = self.engine.load_hifigan(self.hparams, state_dict)
        # `half` (larynx/hifi_gan.py:96-97 calls `.half()` on the generator): the native fp16 vocoder — fp16 weights and
        # activation planes, one fp16 MFMA per product, f32 accumulate (csrc/conv_f16.h).  A vocoder whose geometry the fp16
        # tiles do not cover runs the split-bf16 mode instead (f32 planes, 3 x bf16 MFMA per product), and says so.
        self.half = bool(config.half)
        self.precision = ffi.PRECISION_F32
        if self.half:
            try:
                self.engine.set_precision(self.model_id, ffi.PRECISION_F16)
                self.precision = ffi.PRECISION_F16
            except ffi.Mi355ttsError as e:
                _LOGGER.warning("half: %s; using the split-bf16 mode", e)
                self.engine.set_precision(self.model_id, ffi.PRECISION_BF16X3)
                self.precision = ffi.PRECISION_BF16X3
        self.denoiser_strength = float(config.denoiser_strength)

    def mels_to_audio(self, mels: ARRAY_OR_TENSOR, settings: typing.Optional[SettingsType] = None) -> np.ndarray:
        return self.mels_to_audio_padded(mels, settings, 0, 0)

    def mels_to_audio_padded(self, mels: ARRAY_OR_TENSOR, settings: typing.Optional[SettingsType] = None,
                             pad_before: int = 0, pad_after: int = 0) -> np.ndarray:
        """`mels_to_audio` with the SSML pauses of `_sentence_task` (`larynx/__init__.py:277-283`,
        `np.pad` on the host there) written by the device's int16 kernel: `pad_before` zero
        samples, the audio, `pad_after` zero samples."""
        strength = self.denoiser_strength
        if settings:
            strength = float(settings.get("denoiser_strength", strength))
        batch = mels if isinstance(mels, MelBatch) else self.engine.mel_from_numpy(np.asarray(mels, np.float32))
        _, i16 = self.engine.hifigan_infer(self.model_id, batch, want_float=False, want_int16=True,
                                           denoiser_strength=max(strength, 0.0), pad_before=pad_before, pad_after=pad_after)
        n = int(batch.frames[0]) * self.engine.hop(self.model_id) + int(pad_before) + int(pad_after)
        return i16[0, :n] if batch.batch == 1 else i16

    def mels_to_float(self, mels: ARRAY_OR_TENSOR, denoiser_strength: float = 0.0) -> np.ndarray:
        """Generator output before `audio_float_to_int16` (what waveform parity is defined on)."""
        batch = mels if isinstance(mels, MelBatch) else self.engine.mel_from_numpy(np.asarray(mels, np.float32))
        f32, _ = self.engine.hifigan_infer(self.model_id, batch, want_float=True, want_int16=False,
                                           denoiser_strength=denoiser_strength)
        return f32
