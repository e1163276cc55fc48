"""TEST INFRASTRUCTURE — CPU oracle for the GlowTTS inference path.

A numpy restatement of `glow_tts.models.FlowGenerator.forward` at inference
(`/root/reference/glow_tts/models.py:308-354`) for ONE utterance (the reference
never batches: `larynx/__init__.py:146-157`, `larynx/glow_tts.py:156-157`).
It consumes the reference's own checkpoint format (state-dict with
`weight_g`/`weight_v`), so it doubles as a check of the product's weight
converter.  Pinned against the reference's torch modules by
`oracle/make_golden.py` (run in the build container, where /root/reference
exists) and against the committed vectors in `tests/golden/` everywhere else.

Parity status: the reference's own tests hold no tensor-level vectors for this
path; the oracle is pinned to outputs of the reference's torch
implementation run on seeded synthetic weights.
"""
from __future__ import annotations

import math
import typing

import numpy as np

from . import nn_np as nn

F32 = np.float32


def _w(sd, prefix: str) -> np.ndarray:
    """Conv weight, folding weight-norm if the checkpoint still carries it."""
    if prefix + ".weight" in sd:
        return np.asarray(sd[prefix + ".weight"], F32)
    return nn.fold_weight_norm(np.asarray(sd[prefix + ".weight_g"]), np.asarray(sd[prefix + ".weight_v"]))


def _b(sd, prefix: str) -> np.ndarray:
    return np.asarray(sd[prefix + ".bias"], F32)


def _ln(sd, prefix: str, x: np.ndarray) -> np.ndarray:
    return nn.layer_norm_channels(x, np.asarray(sd[prefix + ".gamma"], F32), np.asarray(sd[prefix + ".beta"], F32))


def attention(sd, prefix: str, x: np.ndarray, hp) -> np.ndarray:
    """`MultiHeadAttention.forward` (attentions.py:205-264) for self-attention
    with a full mask (B=1 => no padded keys).  The relative-position terms are
    evaluated directly on the |i-j| <= window band: scores[i,j] += q_i . Ek[j-i+w],
    out[i] += sum_j p[i,j] Ev[j-i+w]  — what the reference's pad/reshape helpers
    (attentions.py:284-335) compute."""
    H = hp.hidden_channels
    nh = hp.n_heads
    dk = H // nh
    P = x.shape[1]
    w = hp.window_size
    q = nn.conv1d(x, _w(sd, prefix + ".conv_q"), _b(sd, prefix + ".conv_q"))
    k = nn.conv1d(x, _w(sd, prefix + ".conv_k"), _b(sd, prefix + ".conv_k"))
    v = nn.conv1d(x, _w(sd, prefix + ".conv_v"), _b(sd, prefix + ".conv_v"))
    ek = np.asarray(sd[prefix + ".emb_rel_k"], F32)[0]  # [2w+1, dk] (heads_share)
    ev = np.asarray(sd[prefix + ".emb_rel_v"], F32)[0]
    out = np.zeros((H, P), F32)
    scale = F32(1.0 / math.sqrt(dk))
    for h in range(nh):
        qh = q[h * dk : (h + 1) * dk].T  # [P, dk]
        kh = k[h * dk : (h + 1) * dk].T
        vh = v[h * dk : (h + 1) * dk].T
        scores = (qh @ kh.T) * scale
        rel = (qh @ ek.T) * scale  # [P, 2w+1]
        for r in range(2 * w + 1):
            off = r - w  # j - i
            i0, i1 = max(0, -off), min(P, P - off)
            if i0 < i1:
                idx = np.arange(i0, i1)
                scores[idx, idx + off] += rel[idx, r]
        p = nn.softmax_last(scores)
        oh = p @ vh
        for r in range(2 * w + 1):
            off = r - w
            i0, i1 = max(0, -off), min(P, P - off)
            if i0 < i1:
                idx = np.arange(i0, i1)
                oh[idx] += p[idx, idx + off][:, None] * ev[r][None, :]
        out[h * dk : (h + 1) * dk] = oh.T
    return nn.conv1d(out, _w(sd, prefix + ".conv_o"), _b(sd, prefix + ".conv_o"))


def speaker_vector(sd, hp, speaker_id) -> typing.Optional[np.ndarray]:
    """`g = F.normalize(self.emb_g(g))` (models.py:318-319): the speaker's embedding row over max(its L2 norm, 1e-12);
    None for a single-speaker voice."""
    if getattr(hp, "n_speakers", 1) <= 1:
        if speaker_id is not None:
            raise ValueError("speaker_id given to a single-speaker voice (the reference has no emb_g there)")
        return None
    if speaker_id is None:
        raise ValueError("a multi-speaker voice needs a speaker_id")
    e = np.asarray(sd["emb_g.weight"], F32)[int(speaker_id)]
    return (e / max(float(np.sqrt(np.sum(e.astype(np.float64) ** 2))), 1e-12)).astype(F32)


def text_encoder(sd, ids: np.ndarray, hp, taps=None, g: typing.Optional[np.ndarray] = None):
    """`TextEncoder.forward` (models.py:118-140): returns x_m [M,P], logw [P].  `g` [gin]: the speaker vector, repeated
    along time and concatenated to the duration predictor's input (models.py:128-132)."""
    H = hp.hidden_channels
    k = hp.kernel_size
    x = (np.asarray(sd["encoder.emb.weight"], F32)[ids] * F32(math.sqrt(H))).T.copy()  # [H,P]
    if taps is not None:
        taps["emb"] = x.copy()
    if hp.prenet:
        # ConvReluNorm: conv -> LayerNorm -> ReLU, x3; then x + proj(.)  (layers.py:73-80)
        x_org = x
        for i in range(hp.prenet_layers):
            pk = hp.prenet_kernel_size
            x = nn.conv1d(x, _w(sd, f"encoder.pre.conv_layers.{i}"), _b(sd, f"encoder.pre.conv_layers.{i}"), padding=pk // 2)
            x = _ln(sd, f"encoder.pre.norm_layers.{i}", x)
            x = np.maximum(x, 0)
        x = x_org + nn.conv1d(x, _w(sd, "encoder.pre.proj"), _b(sd, "encoder.pre.proj"))
        if taps is not None:
            taps["prenet"] = x.copy()
    for l in range(hp.n_layers_enc):  # Encoder.forward, attentions.py:62-74
        y = attention(sd, f"encoder.encoder.attn_layers.{l}", x, hp)
        x = _ln(sd, f"encoder.encoder.norm_layers_1.{l}", x + y)
        f = f"encoder.encoder.ffn_layers.{l}"  # FFN.forward, attentions.py:375-383
        y = nn.conv1d(x, _w(sd, f + ".conv_1"), _b(sd, f + ".conv_1"), padding=k // 2)
        y = np.maximum(y, 0)
        y = nn.conv1d(y, _w(sd, f + ".conv_2"), _b(sd, f + ".conv_2"), padding=k // 2)
        x = _ln(sd, f"encoder.encoder.norm_layers_2.{l}", x + y)
        if taps is not None:
            taps[f"enc{l}"] = x.copy()
    x_m = nn.conv1d(x, _w(sd, "encoder.proj_m"), _b(sd, "encoder.proj_m"))
    # DurationPredictor: conv -> ReLU -> LayerNorm (models.py:39-49)
    w = "encoder.proj_w"
    x_dp = x if g is None else np.concatenate([x, np.repeat(g[:, None], x.shape[1], axis=1)], axis=0).astype(F32)
    d = nn.conv1d(x_dp, _w(sd, w + ".conv_1"), _b(sd, w + ".conv_1"), padding=k // 2)
    d = _ln(sd, w + ".norm_1", np.maximum(d, 0))
    d = nn.conv1d(d, _w(sd, w + ".conv_2"), _b(sd, w + ".conv_2"), padding=k // 2)
    d = _ln(sd, w + ".norm_2", np.maximum(d, 0))
    logw = nn.conv1d(d, _w(sd, w + ".proj"), _b(sd, w + ".proj"))[0]
    if taps is not None:
        taps["x_m"] = x_m.copy()
        taps["logw"] = logw.copy()
    return x_m, logw


def durations_to_frames(logw: np.ndarray, length_scale: float, n_sqz: int):
    """models.py:323-336 + utils.py:99-115: ceil'd durations, frame count
    (truncated to a multiple of n_sqz) and, per frame, the id it repeats —
    `generate_path` puts frame j on id t iff cum[t-1] <= j < cum[t]."""
    w = np.exp(logw.astype(F32)) * F32(length_scale)
    w_ceil = np.ceil(w).astype(F32)
    y_len = int(max(float(w_ceil.sum(dtype=F32)), 1.0))
    y_len = (y_len // n_sqz) * n_sqz
    cum = np.cumsum(w_ceil, dtype=F32)
    frames = np.arange(y_len, dtype=F32)
    idx = np.searchsorted(cum, frames, side="right")  # #{t : cum[t] <= j}
    return w_ceil, y_len, idx.astype(np.int64)


def squeeze(x: np.ndarray, n: int) -> np.ndarray:
    """utils.py:135-147 — [C, T] -> [C*n, T/n], channel c' = s*C + c."""
    C, T = x.shape
    return x.reshape(C, T // n, n).transpose(2, 0, 1).reshape(C * n, T // n).copy()


def unsqueeze(x: np.ndarray, n: int) -> np.ndarray:
    """utils.py:150-160."""
    C, T = x.shape
    return x.reshape(n, C // n, T).transpose(1, 2, 0).reshape(C // n, T * n).copy()


def wavenet(sd, prefix: str, x: np.ndarray, hp, g: typing.Optional[np.ndarray] = None) -> np.ndarray:
    """`WN.forward` (layers.py:138-162).  `g` [gin]: `cond_layer(g)` (a 1x1 conv of a length-1 tensor = a matrix-vector
    product) gives every layer a [2H] offset added to its gate pre-activation (layers.py:141-154)."""
    H = hp.hidden_channels
    kd = hp.kernel_size_dec
    out = np.zeros_like(x)
    cond = None
    if g is not None:
        cond = (_w(sd, f"{prefix}.cond_layer")[:, :, 0] @ g + _b(sd, f"{prefix}.cond_layer")).astype(F32)
    for i in range(hp.n_block_layers):
        d = hp.dilation_rate ** i
        pad = (kd * d - d) // 2
        x_in = nn.conv1d(x, _w(sd, f"{prefix}.in_layers.{i}"), _b(sd, f"{prefix}.in_layers.{i}"), dilation=d, padding=pad)
        if cond is not None:
            x_in = (x_in + cond[i * 2 * H : (i + 1) * 2 * H, None]).astype(F32)
        acts = np.tanh(x_in[:H]) * nn.sigmoid(x_in[H:])  # utils.py:31-38
        rs = nn.conv1d(acts, _w(sd, f"{prefix}.res_skip_layers.{i}"), _b(sd, f"{prefix}.res_skip_layers.{i}"))
        if i < hp.n_block_layers - 1:
            x = x + rs[:H]
            out = out + rs[H:]
        else:
            out = out + rs
    return out


def flow_decoder_reverse(sd, z: np.ndarray, hp, taps=None, g: typing.Optional[np.ndarray] = None) -> np.ndarray:
    """`FlowSpecDecoder.forward(reverse=True)` (models.py:191-209): squeeze,
    then for blocks 11..0: CouplingBlock -> InvConvNear -> ActNorm, all reversed."""
    x = squeeze(z, hp.n_sqz)
    C = x.shape[0]
    half = C // 2
    ns = hp.n_split
    for b in reversed(range(hp.n_blocks_dec)):
        an, ic, cp = (f"decoder.flows.{3 * b + j}" for j in range(3))
        # CouplingBlock reverse (attentions.py:119-142)
        x0, x1 = x[:half], x[half:]
        h = nn.conv1d(x0, _w(sd, cp + ".start"), _b(sd, cp + ".start"))
        h = wavenet(sd, cp + ".wn", h, hp, g)
        o = nn.conv1d(h, _w(sd, cp + ".end"), _b(sd, cp + ".end"))
        m, logs = o[:half], o[half:]
        x = np.concatenate([x0, (x1 - m) * np.exp(-logs)], axis=0).astype(F32)
        if taps is not None:
            taps[f"coupling{b}"] = x.copy()
        # InvConvNear reverse (layers.py:238-272)
        T = x.shape[1]
        w_inv = np.linalg.inv(np.asarray(sd[ic + ".weight"], F32).astype(F32)).astype(F32)
        xs = x.reshape(2, C // ns, ns // 2, T).transpose(0, 2, 1, 3).reshape(ns, C // ns, T)
        zs = np.einsum("on,nkt->okt", w_inv, xs).astype(F32)
        x = zs.reshape(2, ns // 2, C // ns, T).transpose(0, 2, 1, 3).reshape(C, T).copy()
        # ActNorm reverse (layers.py:192-194)
        x = ((x - np.asarray(sd[an + ".bias"], F32)[0]) * np.exp(-np.asarray(sd[an + ".logs"], F32)[0])).astype(F32)
        if taps is not None:
            taps[f"flow{b}"] = x.copy()
    return unsqueeze(x, hp.n_sqz)


def glow_tts_infer(
    sd,
    hp,
    ids: np.ndarray,
    noise: typing.Optional[np.ndarray] = None,
    noise_scale: float = 0.667,
    length_scale: float = 1.0,
    taps: typing.Optional[dict] = None,
    speaker_id: typing.Optional[int] = None,
) -> np.ndarray:
    """ids int64 [P] -> mel float32 [M, F].  `noise` is the N(0,1) tensor the
    reference draws with `torch.randn_like(z_m)` (models.py:348), laid out
    [M, >=F]; None means zeros (equivalent to noise_scale=0)."""
    ids = np.asarray(ids, np.int64)
    g = speaker_vector(sd, hp, speaker_id)  # models.py:318-319
    x_m, logw = text_encoder(sd, ids, hp, taps, g)
    w_ceil, F, idx = durations_to_frames(logw, length_scale, hp.n_sqz)
    if taps is not None:
        taps["w_ceil"] = w_ceil
        taps["frame_to_id"] = idx
    M = hp.mel_channels
    if F == 0:
        return np.zeros((M, 0), F32)
    z = x_m[:, idx]
    if noise is not None and noise_scale != 0.0:
        z = z + np.asarray(noise, F32)[:, :F] * F32(noise_scale)
    z = z.astype(F32)
    if taps is not None:
        taps["z"] = z.copy()
    return flow_decoder_reverse(sd, z, hp, taps, g)
