"""TEST INFRASTRUCTURE — CPU oracle for the HiFi-GAN generator.

numpy restatement of `hifi_gan.models.Generator.forward`
(`/root/reference/hifi_gan/models.py:186-202`) with `ResBlock1` (:91-98) and
`ResBlock2` (:136-141), for ONE utterance, consuming the reference's
checkpoint format (`weight_g`/`weight_v`, folded here exactly as
`remove_weight_norm` does).  Pinned by `oracle/make_golden.py` against the
reference's torch module.
"""
from __future__ import annotations

import typing

import numpy as np

from . import nn_np as nn

F32 = np.float32
LRELU_SLOPE = 0.1  # hifi_gan/models.py:13


def _w(sd, prefix):
    if prefix + ".weight" in sd:
        return np.asarray(sd[prefix + ".weight"], F32)
    return nn.fold_weight_norm(np.asarray(sd[prefix + ".weight_g"]), np.asarray(sd[prefix + ".weight_v"]))


def _b(sd, prefix):
    return np.asarray(sd[prefix + ".bias"], F32)


def get_padding(k: int, d: int = 1) -> int:
    return int((k * d - d) / 2)  # hifi_gan/utils.py:17-18


def resblock1(sd, prefix: str, x: np.ndarray, k: int, dil) -> np.ndarray:
    for m, d in enumerate(dil):
        xt = nn.leaky_relu(x, LRELU_SLOPE)
        xt = nn.conv1d(xt, _w(sd, f"{prefix}.convs1.{m}"), _b(sd, f"{prefix}.convs1.{m}"), dilation=d, padding=get_padding(k, d))
        xt = nn.leaky_relu(xt, LRELU_SLOPE)
        xt = nn.conv1d(xt, _w(sd, f"{prefix}.convs2.{m}"), _b(sd, f"{prefix}.convs2.{m}"), dilation=1, padding=get_padding(k, 1))
        x = xt + x
    return x


def resblock2(sd, prefix: str, x: np.ndarray, k: int, dil) -> np.ndarray:
    for m, d in enumerate(dil):
        xt = nn.leaky_relu(x, LRELU_SLOPE)
        xt = nn.conv1d(xt, _w(sd, f"{prefix}.convs.{m}"), _b(sd, f"{prefix}.convs.{m}"), dilation=d, padding=get_padding(k, d))
        x = xt + x
    return x


def hifigan_infer(sd, hp, mel: np.ndarray, taps: typing.Optional[dict] = None) -> np.ndarray:
    """mel float32 [num_mels, F] -> waveform float32 [F * hop] in (-1, 1)."""
    x = nn.conv1d(np.asarray(mel, F32), _w(sd, "conv_pre"), _b(sd, "conv_pre"), padding=3)
    if taps is not None:
        taps["conv_pre"] = x.copy()
    nk = len(hp.resblock_kernel_sizes)
    rb = resblock1 if hp.resblock == "1" else resblock2
    for i, (u, ku) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        x = nn.leaky_relu(x, LRELU_SLOPE)
        x = nn.conv_transpose1d(x, _w(sd, f"ups.{i}"), _b(sd, f"ups.{i}"), stride=u, padding=(ku - u) // 2)
        if taps is not None:
            taps[f"up{i}"] = x.copy()
        xs = None
        for j, (k, dil) in enumerate(zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)):
            r = rb(sd, f"resblocks.{i * nk + j}", x, k, dil)
            xs = r if xs is None else xs + r
        x = (xs / F32(nk)).astype(F32)
        if taps is not None:
            taps[f"stage{i}"] = x.copy()
    x = nn.leaky_relu(x, 0.01)  # F.leaky_relu default slope, models.py:198
    x = nn.conv1d(x, _w(sd, "conv_post"), _b(sd, "conv_post"), padding=3)
    return np.tanh(x)[0].astype(F32)
