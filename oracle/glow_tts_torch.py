"""TEST INFRASTRUCTURE — the GlowTTS oracle restated on torch CPU operators.

Same algorithm as `oracle/glow_tts_np.py` (a restatement of
`glow_tts/models.py:308-354` at inference, NOT the reference's module classes —
/root/reference does not exist on the GPU box), with `torch.nn.functional.conv1d`,
`torch.matmul` and `torch.softmax` doing the arithmetic: the operators (and oneDNN /
MKL kernels) the reference's own `--backend pytorch` path runs on.  Used only as the
GlowTTS half of bench.py's timed `cpu_baseline` ("the in-tree torch
path ... wrapped like `_sentence_task`"); pinned against the numpy oracle — itself pinned
to the reference's modules — in tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
import typing

import numpy as np

from . import glow_tts_np as ref_np


def glow_tts_infer_torch(sd, hp, ids: np.ndarray, noise: typing.Optional[np.ndarray] = None, noise_scale: float = 0.667,
                         length_scale: float = 1.0, threads: int = 0) -> np.ndarray:
    """ids int64 [P] -> mel float32 [M, F] (see `glow_tts_np.glow_tts_infer`)."""
    import torch
    import torch.nn.functional as F

    if threads:
        torch.set_num_threads(threads)

    def w(prefix):
        return torch.from_numpy(np.ascontiguousarray(ref_np._w(sd, prefix)))

    def b(prefix):
        return torch.from_numpy(np.ascontiguousarray(ref_np._b(sd, prefix)))

    def t(name):
        return torch.from_numpy(np.ascontiguousarray(np.asarray(sd[name], np.float32)))

    def ln(prefix, x):  # layers.py:19-28: over channels, biased variance, eps 1e-4
        mean = x.mean(1, keepdim=True)
        var = ((x - mean) ** 2).mean(1, keepdim=True)
        return (x - mean) * torch.rsqrt(var + 1e-4) * t(prefix + ".gamma").view(1, -1, 1) + t(prefix + ".beta").view(1, -1, 1)

    H, nh, k, win = hp.hidden_channels, hp.n_heads, hp.kernel_size, hp.window_size
    dk = H // nh
    with torch.no_grad():
        ids_t = torch.from_numpy(np.asarray(ids, np.int64))
        P = ids_t.shape[0]
        x = (t("encoder.emb.weight")[ids_t] * math.sqrt(H)).t()[None]  # [1, H, P]
        if hp.prenet:  # ConvReluNorm, layers.py:73-80
            x_org = x
            for i in range(hp.prenet_layers):
                x = F.conv1d(x, w(f"encoder.pre.conv_layers.{i}"), b(f"encoder.pre.conv_layers.{i}"), padding=hp.prenet_kernel_size // 2)
                x = torch.relu(ln(f"encoder.pre.norm_layers.{i}", x))
            x = x_org + F.conv1d(x, w("encoder.pre.proj"), b("encoder.pre.proj"))
        # band index of the relative-position terms: r = j - i + win, valid for |j - i| <= win
        ii = torch.arange(P)[:, None]
        jj = torch.arange(P)[None, :]
        rel_idx = (jj - ii + win).clamp(0, 2 * win)
        in_band = ((jj - ii).abs() <= win).to(torch.float32)
        band_j = (ii + torch.arange(2 * win + 1)[None, :] - win)  # [P, 2w+1]: key index of band slot r for query i
        band_ok = ((band_j >= 0) & (band_j < P)).to(torch.float32)
        band_j = band_j.clamp(0, P - 1)
        for l in range(hp.n_layers_enc):  # Encoder.forward, attentions.py:62-74
            a = f"encoder.encoder.attn_layers.{l}"
            q = F.conv1d(x, w(a + ".conv_q"), b(a + ".conv_q")).view(nh, dk, P).transpose(1, 2)  # [nh, P, dk]
            kk = F.conv1d(x, w(a + ".conv_k"), b(a + ".conv_k")).view(nh, dk, P).transpose(1, 2)
            v = F.conv1d(x, w(a + ".conv_v"), b(a + ".conv_v")).view(nh, dk, P).transpose(1, 2)
            ek, ev = t(a + ".emb_rel_k")[0], t(a + ".emb_rel_v")[0]  # [2w+1, dk]
            scale = 1.0 / math.sqrt(dk)
            scores = torch.matmul(q, kk.transpose(1, 2)) * scale  # attentions.py:214-222
            rel = torch.matmul(q, ek.t()) * scale  # [nh, P, 2w+1]
            scores = scores + torch.gather(rel, 2, rel_idx[None].expand(nh, P, P)) * in_band[None]
            p = torch.softmax(scores, dim=-1)
            o = torch.matmul(p, v)
            p_band = torch.gather(p, 2, band_j[None].expand(nh, P, 2 * win + 1)) * band_ok[None]
            o = o + torch.matmul(p_band, ev)  # attentions.py:240-250
            y = F.conv1d(o.transpose(1, 2).reshape(1, H, P), w(a + ".conv_o"), b(a + ".conv_o"))
            x = ln(f"encoder.encoder.norm_layers_1.{l}", x + y)
            f = f"encoder.encoder.ffn_layers.{l}"  # FFN.forward, attentions.py:375-383
            y = torch.relu(F.conv1d(x, w(f + ".conv_1"), b(f + ".conv_1"), padding=k // 2))
            y = F.conv1d(y, w(f + ".conv_2"), b(f + ".conv_2"), padding=k // 2)
            x = ln(f"encoder.encoder.norm_layers_2.{l}", x + y)
        x_m = F.conv1d(x, w("encoder.proj_m"), b("encoder.proj_m"))
        pw = "encoder.proj_w"  # DurationPredictor, models.py:39-49
        d = ln(pw + ".norm_1", torch.relu(F.conv1d(x, w(pw + ".conv_1"), b(pw + ".conv_1"), padding=k // 2)))
        d = ln(pw + ".norm_2", torch.relu(F.conv1d(d, w(pw + ".conv_2"), b(pw + ".conv_2"), padding=k // 2)))
        logw = F.conv1d(d, w(pw + ".proj"), b(pw + ".proj"))[0, 0]
        # durations -> frames (models.py:323-346): integer bookkeeping shared with the numpy oracle
        _, n_frames, idx = ref_np.durations_to_frames(logw.numpy(), length_scale, hp.n_sqz)
        M = hp.mel_channels
        if n_frames == 0:
            return np.zeros((M, 0), np.float32)
        z = x_m[0][:, torch.from_numpy(idx)]
        if noise is not None and noise_scale != 0.0:
            z = z + torch.from_numpy(np.ascontiguousarray(np.asarray(noise, np.float32)[:, :n_frames])) * float(noise_scale)
        # FlowSpecDecoder.forward(reverse=True), models.py:191-209
        n = hp.n_sqz
        C, T = M * n, n_frames // n
        x = z.view(M, T, n).permute(2, 0, 1).reshape(1, C, T)  # squeeze, utils.py:135-147
        half, ns = C // 2, hp.n_split
        for blk in reversed(range(hp.n_blocks_dec)):
            an, ic, cp = (f"decoder.flows.{3 * blk + j}" for j in range(3))
            x0, x1 = x[:, :half], x[:, half:]
            h = F.conv1d(x0, w(cp + ".start"), b(cp + ".start"))
            out = torch.zeros_like(h)
            for i in range(hp.n_block_layers):  # WN.forward, layers.py:138-162
                dil = hp.dilation_rate ** i
                x_in = F.conv1d(h, w(f"{cp}.wn.in_layers.{i}"), b(f"{cp}.wn.in_layers.{i}"), dilation=dil,
                                padding=(hp.kernel_size_dec * dil - dil) // 2)
                acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
                rs = F.conv1d(acts, w(f"{cp}.wn.res_skip_layers.{i}"), b(f"{cp}.wn.res_skip_layers.{i}"))
                if i < hp.n_block_layers - 1:
                    h = h + rs[:, :H]
                    out = out + rs[:, H:]
                else:
                    out = out + rs
            o = F.conv1d(out, w(cp + ".end"), b(cp + ".end"))
            x = torch.cat([x0, (x1 - o[:, :half]) * torch.exp(-o[:, half:])], 1)
            w_inv = torch.from_numpy(np.linalg.inv(np.asarray(sd[ic + ".weight"], np.float32)).astype(np.float32))
            xs = x.view(2, C // ns, ns // 2, T).permute(0, 2, 1, 3).reshape(ns, C // ns, T)  # InvConvNear, layers.py:238-272
            zs = torch.einsum("on,nkt->okt", w_inv, xs)
            x = zs.view(2, ns // 2, C // ns, T).permute(0, 2, 1, 3).reshape(1, C, T)
            x = (x - t(an + ".bias")) * torch.exp(-t(an + ".logs"))  # ActNorm reverse, layers.py:192-194
        mel = x[0].view(n, M, T).permute(1, 2, 0).reshape(M, T * n)  # unsqueeze, utils.py:150-160
        return mel.numpy().copy()
