"""Hyper-parameters of the two networks on the Larynx hot path.

The values are read off the reference's checked-in voice/vocoder configs
(`local/en-us/ljspeech-glow_tts/config.json:37-63`,
`local/hifi_gan/{universal_large,vctk_medium,vctk_small}/config.json`) and the
dataclasses that parse them (`glow_tts/config.py:36-62`,
`hifi_gan/config.py:29-41`).  Only the fields the inference path reads are kept.
"""
from __future__ import annotations

import json
import typing
from dataclasses import dataclass, field, asdict
from pathlib import Path


@dataclass(frozen=True)
class GlowHParams:
    """GlowTTS `ModelConfig` (glow_tts/config.py:36-62) + mel channel count."""

    num_symbols: int = 46
    hidden_channels: int = 192
    filter_channels: int = 768
    filter_channels_dp: int = 256
    kernel_size: int = 3
    n_blocks_dec: int = 12
    n_layers_enc: int = 6
    n_heads: int = 2
    dilation_rate: int = 1
    kernel_size_dec: int = 5
    n_block_layers: int = 4
    n_sqz: int = 2
    prenet: bool = True
    mean_only: bool = True
    window_size: int = 4
    n_split: int = 4
    mel_channels: int = 80
    prenet_kernel_size: int = 5  # glow_tts/models.py:96 (hard-coded)
    prenet_layers: int = 3  # glow_tts/models.py:97 (hard-coded)
    # multi-speaker voices (glow_tts/models.py:304-306, 318-319): n_speakers > 1 adds the speaker embedding `emb_g`
    # [n_speakers, gin_channels]; its L2-normalised row conditions every WaveNet layer of the decoder (layers.py:109-113,
    # 144-154) and is concatenated to the duration predictor's input (models.py:128-132)
    n_speakers: int = 1
    gin_channels: int = 0

    @staticmethod
    def from_config(cfg: typing.Mapping[str, typing.Any]) -> "GlowHParams":
        """Build from a voice `config.json` dict (keys `model` and `audio`)."""
        m = dict(cfg.get("model", {}))
        a = dict(cfg.get("audio", {}))
        hid = int(m.get("hidden_channels", 192))
        if int(m.get("hidden_channels_enc", hid) or hid) != hid or int(
            m.get("hidden_channels_dec", hid) or hid
        ) != hid:
            raise ValueError("hidden_channels_enc/dec must equal hidden_channels")
        # `"n_speakers": null` (some exported configs) and 0 both mean what the reference's default of 1 means: no embedding
        n_spk = int(m.get("n_speakers", 1) or 0)
        gin = int(m.get("gin_channels", 0) or 0)
        if n_spk > 1 and gin <= 0:
            raise ValueError("a multi-speaker voice needs gin_channels > 0")
        if n_spk <= 1 and gin != 0:
            # (the reference builds the conditioning layers from gin_channels alone but has no embedding to feed them)
            raise ValueError(f"unsupported GlowTTS option gin_channels={gin} with n_speakers={n_spk}")
        unsupported = {
            "sigmoid_scale": (False,),
            "block_length": (None,),
        }
        for key, ok in unsupported.items():
            if key in m and m[key] not in ok:
                raise ValueError(f"unsupported GlowTTS option {key}={m[key]!r}")
        if not bool(m.get("mean_only", True)):
            raise ValueError("mean_only=False voices are not supported")
        return GlowHParams(
            num_symbols=int(m["num_symbols"]),
            hidden_channels=hid,
            filter_channels=int(m.get("filter_channels", 768)),
            filter_channels_dp=int(m.get("filter_channels_dp", 256)),
            kernel_size=int(m.get("kernel_size", 3)),
            n_blocks_dec=int(m.get("n_blocks_dec", 12)),
            n_layers_enc=int(m.get("n_layers_enc", 6)),
            n_heads=int(m.get("n_heads", 2)),
            dilation_rate=int(m.get("dilation_rate", 1)),
            kernel_size_dec=int(m.get("kernel_size_dec", 5)),
            n_block_layers=int(m.get("n_block_layers", 4)),
            n_sqz=int(m.get("n_sqz", 2)),
            prenet=bool(m.get("prenet", True)),
            mean_only=True,
            window_size=int(m.get("window_size", 4)),
            n_split=int(m.get("n_split", 4)),
            mel_channels=int(a.get("mel_channels", 80)),
            n_speakers=n_spk if n_spk > 1 else 1,
            gin_channels=gin if n_spk > 1 else 0,
        )

    def to_config(self) -> typing.Dict[str, typing.Any]:
        d = asdict(self)
        mel = d.pop("mel_channels")
        d.pop("prenet_kernel_size")
        d.pop("prenet_layers")
        d.update(
            hidden_channels_enc=self.hidden_channels,
            hidden_channels_dec=self.hidden_channels,
            sigmoid_scale=False,
            block_length=None,
            p_dropout=0.1,
            p_dropout_dec=0.05,
        )
        return {"model": d, "audio": {"mel_channels": mel}}


@dataclass(frozen=True)
class HifiGanHParams:
    """HiFi-GAN `ModelConfig` (hifi_gan/config.py:29-41)."""

    resblock: str = "1"
    upsample_rates: typing.Tuple[int, ...] = (8, 8, 2, 2)
    upsample_kernel_sizes: typing.Tuple[int, ...] = (16, 16, 4, 4)
    upsample_initial_channel: int = 512
    resblock_kernel_sizes: typing.Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: typing.Tuple[typing.Tuple[int, ...], ...] = (
        (1, 3, 5),
        (1, 3, 5),
        (1, 3, 5),
    )
    num_mels: int = 80  # hifi_gan/models.py:153 hard-codes 80 input channels

    @staticmethod
    def from_config(cfg: typing.Mapping[str, typing.Any]) -> "HifiGanHParams":
        """Accepts both config layouts the reference ships: the `TrainingConfig`
        layout (`{"model": {...}}`, vctk_medium/vctk_small) and the upstream flat
        layout (universal_large/config.json:2-15)."""
        m = cfg["model"] if "model" in cfg else cfg
        num_mels = int(cfg.get("audio", {}).get("num_mels", cfg.get("num_mels", 80))) if isinstance(cfg.get("audio", {}), dict) else 80
        return HifiGanHParams(
            num_mels=num_mels,
            resblock=str(m.get("resblock", "1")),
            upsample_rates=tuple(int(v) for v in m["upsample_rates"]),
            upsample_kernel_sizes=tuple(int(v) for v in m["upsample_kernel_sizes"]),
            upsample_initial_channel=int(m["upsample_initial_channel"]),
            resblock_kernel_sizes=tuple(int(v) for v in m["resblock_kernel_sizes"]),
            resblock_dilation_sizes=tuple(
                tuple(int(d) for d in ds) for ds in m["resblock_dilation_sizes"]
            ),
        )

    def to_config(self) -> typing.Dict[str, typing.Any]:
        return {
            "model": {
                "resblock": self.resblock,
                "upsample_rates": list(self.upsample_rates),
                "upsample_kernel_sizes": list(self.upsample_kernel_sizes),
                "upsample_initial_channel": self.upsample_initial_channel,
                "resblock_kernel_sizes": list(self.resblock_kernel_sizes),
                "resblock_dilation_sizes": [list(d) for d in self.resblock_dilation_sizes],
            },
            "audio": {"num_mels": self.num_mels},
        }

    @property
    def hop(self) -> int:
        h = 1
        for u in self.upsample_rates:
            h *= u
        return h

    def stage_channels(self, i: int) -> int:
        """Channels after upsample stage i (hifi_gan/models.py:166-167)."""
        return self.upsample_initial_channel // (2 ** (i + 1))


# --- presets (values from the reference's checked-in configs) -----------------

LJSPEECH = GlowHParams(num_symbols=46)  # local/en-us/ljspeech-glow_tts/config.json
THORSTEN = GlowHParams(num_symbols=54)  # local/de-de/thorsten-glow_tts/config.json
SIWIS = GlowHParams(num_symbols=42)  # local/fr-fr/siwis-glow_tts/config.json

HIFIGAN_HIGH = HifiGanHParams()  # universal_large
HIFIGAN_MEDIUM = HifiGanHParams(upsample_initial_channel=128)  # vctk_medium
HIFIGAN_LOW = HifiGanHParams(  # vctk_small
    resblock="2",
    upsample_rates=(8, 8, 4),
    upsample_kernel_sizes=(16, 16, 8),
    upsample_initial_channel=256,
    resblock_kernel_sizes=(3, 5, 7),
    resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)),
)

VOCODER_QUALITY = {  # larynx/utils.py:27-31
    "high": HIFIGAN_HIGH,
    "medium": HIFIGAN_MEDIUM,
    "low": HIFIGAN_LOW,
}

# Shrunk variants with the same topology, for CPU-side tests of the kernel
# sources (tests/hipemu) where the full sizes would take minutes.
TINY_GLOW = GlowHParams(
    num_symbols=20,
    hidden_channels=32,
    filter_channels=64,
    filter_channels_dp=40,
    n_blocks_dec=2,
    n_layers_enc=2,
    n_block_layers=2,
    mel_channels=16,
)
TINY_HIFIGAN = HifiGanHParams(
    upsample_rates=(4, 2),
    upsample_kernel_sizes=(8, 4),
    upsample_initial_channel=32,
    resblock_kernel_sizes=(3, 5),
    resblock_dilation_sizes=((1, 3), (1, 2)),
    num_mels=16,
)
TINY_HIFIGAN_RB2 = HifiGanHParams(
    resblock="2",
    upsample_rates=(4, 4),
    upsample_kernel_sizes=(8, 8),
    upsample_initial_channel=32,
    resblock_kernel_sizes=(3, 5),
    resblock_dilation_sizes=((1, 2), (2, 6)),
    num_mels=16,
)


# stages of 64 and 32 channels: exercises the fused ResBlock-pair kernel on the emulator
TINY_HIFIGAN_PAIR = HifiGanHParams(
    upsample_rates=(2, 2),
    upsample_kernel_sizes=(4, 4),
    upsample_initial_channel=128,
    resblock_kernel_sizes=(3, 11),
    resblock_dilation_sizes=((1, 3), (5, 1)),
    num_mels=16,
)


# stages of 16 and 8 channels with the shipped (3, 7, 11) x (1, 3, 5) ResBlock1 chains: the one-launch MRF kernel (mrf_small.h)
TINY_HIFIGAN_NARROW = HifiGanHParams(
    upsample_rates=(2, 2),
    upsample_kernel_sizes=(4, 4),
    upsample_initial_channel=32,
    num_mels=16,
)


def load_config_json(path: typing.Union[str, Path]) -> typing.Dict[str, typing.Any]:
    with open(path, "r", encoding="utf-8") as f:
        return json.load(f)
