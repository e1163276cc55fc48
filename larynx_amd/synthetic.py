"""Seeded synthetic voices in the reference's checkpoint format.

There are no released voices offline (the reference downloads them at run time,
`larynx/utils.py:104-167`), so parity tests and the benchmark run on random
weights.  This module produces state-dicts with EXACTLY the key names, shapes
and weight-norm parametrisation (`weight_g`/`weight_v`) a trained checkpoint
has (`glow_tts/checkpoint.py:41` -> `checkpoint["model"]`,
`hifi_gan/checkpoint.py:49,63` -> `dict["generator"]`), from a numpy PCG64 stream,
so the same weights can be regenerated bit-identically on any box without torch.

Layers the reference initialises to zero / identity (`CouplingBlock.end`,
attentions.py:104-106; `ConvReluNorm.proj`, layers.py:70-71; `ActNorm`,
layers.py:179-180) are given non-trivial values, otherwise half of the decoder
would never be exercised.  Scales are chosen so activations stay O(1), the
waveform is neither silent nor saturated, and mel values land where a real
voice's do (about 0.57 +- 0.06 in normalised units).
"""
from __future__ import annotations

import math
import typing

import numpy as np

from .hparams import GlowHParams, HifiGanHParams

StateDict = typing.Dict[str, np.ndarray]


def _normal(rng: np.random.Generator, shape, std: float) -> np.ndarray:
    return (rng.standard_normal(size=shape) * std).astype(np.float32)


def _weight_norm_pair(
    rng: np.random.Generator, shape, std: float, prefix: str, out: StateDict
) -> None:
    """Emit `weight_v` ~ N(0, std) and `weight_g` = ||v|| * U(0.8, 1.25) (norm over
    all dims but 0, as `torch.nn.utils.weight_norm(dim=0)` defines it)."""
    v = _normal(rng, shape, std)
    norm = np.sqrt((v.astype(np.float64) ** 2).reshape(shape[0], -1).sum(axis=1))
    g = norm * rng.uniform(0.8, 1.25, size=shape[0])
    out[prefix + ".weight_g"] = g.astype(np.float32).reshape((shape[0],) + (1,) * (len(shape) - 1))
    out[prefix + ".weight_v"] = v


def make_glow_state_dict(
    hp: GlowHParams,
    seed: int = 1234,
    frames_per_id: float = 5.2,
    mel_mean: float = 0.57,
    mel_std: float = 0.06,
) -> StateDict:
    """State-dict of `glow_tts.models.FlowGenerator` (key list: SURVEY.md §8(b))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: StateDict = {}
    H, Fc, Fdp = hp.hidden_channels, hp.filter_channels, hp.filter_channels_dp
    M = hp.mel_channels
    k = hp.kernel_size
    dk = H // hp.n_heads

    sd["encoder.emb.weight"] = _normal(rng, (hp.num_symbols, H), H ** -0.5)
    if hp.prenet:
        for i in range(hp.prenet_layers):
            pk = hp.prenet_kernel_size
            sd[f"encoder.pre.conv_layers.{i}.weight"] = _normal(rng, (H, H, pk), 1.0 / math.sqrt(H * pk))
            sd[f"encoder.pre.conv_layers.{i}.bias"] = _normal(rng, (H,), 0.1)
            sd[f"encoder.pre.norm_layers.{i}.gamma"] = 1.0 + _normal(rng, (H,), 0.1)
            sd[f"encoder.pre.norm_layers.{i}.beta"] = _normal(rng, (H,), 0.1)
        sd["encoder.pre.proj.weight"] = _normal(rng, (H, H, 1), 0.5 / math.sqrt(H))
        sd["encoder.pre.proj.bias"] = _normal(rng, (H,), 0.1)
    for l in range(hp.n_layers_enc):
        p = f"encoder.encoder.attn_layers.{l}"
        sd[p + ".emb_rel_k"] = _normal(rng, (1, 2 * hp.window_size + 1, dk), dk ** -0.5)
        sd[p + ".emb_rel_v"] = _normal(rng, (1, 2 * hp.window_size + 1, dk), dk ** -0.5)
        for name in ("conv_q", "conv_k", "conv_v", "conv_o"):
            # q/k a little hot so the softmax is not uniform
            std = (2.0 if name in ("conv_q", "conv_k") else 1.0) / math.sqrt(H)
            sd[f"{p}.{name}.weight"] = _normal(rng, (H, H, 1), std)
            sd[f"{p}.{name}.bias"] = _normal(rng, (H,), 0.1)
        for n in ("norm_layers_1", "norm_layers_2"):
            sd[f"encoder.encoder.{n}.{l}.gamma"] = 1.0 + _normal(rng, (H,), 0.1)
            sd[f"encoder.encoder.{n}.{l}.beta"] = _normal(rng, (H,), 0.1)
        f = f"encoder.encoder.ffn_layers.{l}"
        sd[f + ".conv_1.weight"] = _normal(rng, (Fc, H, k), 1.0 / math.sqrt(H * k))
        sd[f + ".conv_1.bias"] = _normal(rng, (Fc,), 0.1)
        sd[f + ".conv_2.weight"] = _normal(rng, (H, Fc, k), 1.0 / math.sqrt(Fc * k))
        sd[f + ".conv_2.bias"] = _normal(rng, (H,), 0.1)
    sd["encoder.proj_m.weight"] = _normal(rng, (M, H, 1), 1.0 / math.sqrt(H))
    sd["encoder.proj_m.bias"] = _normal(rng, (M,), 0.1)
    w = "encoder.proj_w"
    gin = hp.gin_channels if hp.n_speakers > 1 else 0
    sd[w + ".conv_1.weight"] = _normal(rng, (Fdp, H, k), 1.0 / math.sqrt(H * k))
    sd[w + ".conv_1.bias"] = _normal(rng, (Fdp,), 0.1)
    sd[w + ".norm_1.gamma"] = 1.0 + _normal(rng, (Fdp,), 0.1)
    sd[w + ".norm_1.beta"] = _normal(rng, (Fdp,), 0.1)
    sd[w + ".conv_2.weight"] = _normal(rng, (Fdp, Fdp, k), 1.0 / math.sqrt(Fdp * k))
    sd[w + ".conv_2.bias"] = _normal(rng, (Fdp,), 0.1)
    sd[w + ".norm_2.gamma"] = 1.0 + _normal(rng, (Fdp,), 0.1)
    sd[w + ".norm_2.beta"] = _normal(rng, (Fdp,), 0.1)
    # log-durations: tuned so ceil(exp(logw)) averages about frames_per_id
    sd[w + ".proj.weight"] = _normal(rng, (1, Fdp, 1), 0.35 / math.sqrt(Fdp))
    sd[w + ".proj.bias"] = np.array([math.log(frames_per_id) + 0.1], np.float32)

    C = M * hp.n_sqz
    for b in range(hp.n_blocks_dec):
        an, ic, cp = (f"decoder.flows.{3 * b + j}" for j in range(3))
        sd[an + ".logs"] = _normal(rng, (1, C, 1), 0.05)
        sd[an + ".bias"] = _normal(rng, (1, C, 1), 0.1)
        q, r = np.linalg.qr(rng.standard_normal((hp.n_split, hp.n_split)))
        q = q * rng.uniform(0.8, 1.25, size=(1, hp.n_split))  # not exactly orthogonal
        sd[ic + ".weight"] = q.astype(np.float32)
        sd[cp + ".start.bias"] = _normal(rng, (H,), 0.1)
        _weight_norm_pair(rng, (H, C // 2, 1), 1.0 / math.sqrt(C // 2), cp + ".start", sd)
        end_w = _normal(rng, (C, H, 1), 1.0 / math.sqrt(H))
        end_w[: C // 2] *= 0.25  # m
        end_w[C // 2 :] *= 0.05  # logs
        sd[cp + ".end.weight"] = end_w
        end_b = _normal(rng, (C,), 0.05)
        sd[cp + ".end.bias"] = end_b
        for j in range(hp.n_block_layers):
            kd = hp.kernel_size_dec
            sd[f"{cp}.wn.in_layers.{j}.bias"] = _normal(rng, (2 * H,), 0.1)
            _weight_norm_pair(rng, (2 * H, H, kd), 1.0 / math.sqrt(H * kd), f"{cp}.wn.in_layers.{j}", sd)
            rs = 2 * H if j < hp.n_block_layers - 1 else H
            sd[f"{cp}.wn.res_skip_layers.{j}.bias"] = _normal(rng, (rs,), 0.1)
            _weight_norm_pair(rng, (rs, H, 1), 1.0 / math.sqrt(H), f"{cp}.wn.res_skip_layers.{j}", sd)
    if gin:
        # Multi-speaker voice (models.py:304-306; layers.py:109-113; models.py:114-116).  Drawn from a generator of their
        # own AFTER everything else, so the single-speaker tensors above are the same numbers with or without speakers.
        # The speaker part is made strong enough to matter: a unit-norm g moves durations and gate pre-activations by a
        # few tenths.
        rg = np.random.Generator(np.random.PCG64(seed + 7919))
        sd["emb_g.weight"] = rg.uniform(-0.1, 0.1, size=(hp.n_speakers, gin)).astype(np.float32)
        wg = _normal(rg, (Fdp, gin, k), 1.5 / math.sqrt(gin * k))
        sd[w + ".conv_1.weight"] = np.concatenate([sd[w + ".conv_1.weight"], wg], axis=1)
        for b in range(hp.n_blocks_dec):
            cp = f"decoder.flows.{3 * b + 2}"
            sd[cp + ".wn.cond_layer.bias"] = _normal(rg, (2 * H * hp.n_block_layers,), 0.05)
            _weight_norm_pair(rg, (2 * H * hp.n_block_layers, gin, 1), 0.5 / math.sqrt(gin), cp + ".wn.cond_layer", sd)
    # flows[0] (ActNorm) is applied LAST in the reverse pass (models.py:195-206):
    # use it to put the mel where a real voice's lives.
    sd["decoder.flows.0.logs"] = np.full((1, C, 1), -math.log(mel_std), np.float32) + _normal(rng, (1, C, 1), 0.05)
    sd["decoder.flows.0.bias"] = np.full((1, C, 1), -mel_mean / mel_std, np.float32) + _normal(rng, (1, C, 1), 0.1)
    return sd


def make_hifigan_state_dict(hp: HifiGanHParams, seed: int = 1234) -> StateDict:
    """State-dict of `hifi_gan.models.Generator` before `remove_weight_norm`."""
    rng = np.random.Generator(np.random.PCG64(seed + 1))
    sd: StateDict = {}
    C0 = hp.upsample_initial_channel
    sd["conv_pre.bias"] = _normal(rng, (C0,), 0.05)
    _weight_norm_pair(rng, (C0, hp.num_mels, 7), 0.25 / math.sqrt(hp.num_mels * 7), "conv_pre", sd)
    ch = C0
    for i, (u, ku) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        cin, cout = C0 // (2 ** i), C0 // (2 ** (i + 1))
        sd[f"ups.{i}.bias"] = _normal(rng, (cout,), 0.05)
        taps = max(1, ku // u)
        _weight_norm_pair(rng, (cin, cout, ku), 1.25 / math.sqrt(cin * taps), f"ups.{i}", sd)
        ch = cout
        for j, (k, dil) in enumerate(zip(hp.resblock_kernel_sizes, hp.resblock_dilation_sizes)):
            n = i * len(hp.resblock_kernel_sizes) + j
            std = 0.9 / math.sqrt(ch * k)
            if hp.resblock == "1":
                for m in range(len(dil)):
                    for cs in ("convs1", "convs2"):
                        sd[f"resblocks.{n}.{cs}.{m}.bias"] = _normal(rng, (ch,), 0.05)
                        _weight_norm_pair(rng, (ch, ch, k), std, f"resblocks.{n}.{cs}.{m}", sd)
            else:
                for m in range(len(dil)):
                    sd[f"resblocks.{n}.convs.{m}.bias"] = _normal(rng, (ch,), 0.05)
                    _weight_norm_pair(rng, (ch, ch, k), std, f"resblocks.{n}.convs.{m}", sd)
    sd["conv_post.bias"] = _normal(rng, (1,), 0.02)
    _weight_norm_pair(rng, (1, ch, 7), 0.16 / math.sqrt(ch * 7), "conv_post", sd)
    return sd


def synthetic_phoneme_ids(rng: np.random.Generator, n_ids: int, num_symbols: int = 46) -> np.ndarray:
    """Ids laid out like the reference's fixtures (`test_phonemes.csv`):
    `3 (#) w o r d 3 w o r d ... 3 2(‖)`, words of 2-7 symbols drawn from [4, V)."""
    ids = [3]
    while len(ids) < n_ids - 2:
        wl = int(rng.integers(2, 8))
        wl = min(wl, n_ids - 2 - len(ids))
        ids.extend(int(v) for v in rng.integers(4, num_symbols, size=wl))
        if len(ids) < n_ids - 2:
            ids.append(3)
    ids = ids[: n_ids - 2] + [3, 2]
    return np.asarray(ids[:n_ids], dtype=np.int64)


def synthetic_batch(seed: int, count: int, num_symbols: int = 46) -> typing.List[np.ndarray]:
    """BASELINE.json config 3: `P_i = clip(round(N(120,15)), 60, 200)` utterances."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        p = int(np.clip(round(rng.normal(120.0, 15.0)), 60, 200))
        out.append(synthetic_phoneme_ids(rng, p, num_symbols))
    return out
