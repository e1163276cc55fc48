"""Utterance-level data parallelism over the GPUs of one node.

Utterances are independent units (the reference already treats every sentence
as an isolated task, `larynx/__init__.py:146-157`), so the path shards with NO
collective inside an utterance.  One process per GPU; the only collective is a
one-time broadcast of the folded weight blobs from rank 0 (RCCL over xGMI —
`torch.distributed` backend "nccl"; "gloo" in the CPU tests), and an optional
gather of the finished audio to rank 0 in sentence order
(the reference yields results in submission order, `larynx/__init__.py:187-190`).
"""
from __future__ import annotations

import typing

import numpy as np


def lpt_assign(costs: typing.Sequence[float], world: int) -> typing.List[typing.List[int]]:
    """Longest-processing-time-first: sort by cost (id count ~ frames ~ FLOPs),
    give each item to the least-loaded rank.  Deterministic on every rank."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    shards: typing.List[typing.List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += float(costs[i])
    for s in shards:
        s.sort()
    return shards


def broadcast_blob(blob: typing.Optional[np.ndarray], numel: int, device, src: int = 0):
    """Rank `src` passes the host blob; every rank gets a tensor on `device` holding it.
    Returns the tensor (keep it alive until the model is loaded)."""
    import torch
    import torch.distributed as dist

    t = torch.empty(numel, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        if blob is None or blob.size != numel:
            raise ValueError("source rank must provide the full blob")
        t.copy_(torch.from_numpy(np.ascontiguousarray(blob, np.float32)))
    dist.broadcast(t, src=src)
    return t


def load_models_broadcast(engine, glow_hp, voc_hp, glow_sd=None, voc_sd=None, device="cpu"):
    """Fold on rank 0, broadcast, load from the receive buffer on every rank.
    With `device="cuda:N"` the process group and torch's GPU state must have been set up
    BEFORE `engine` was created (torch wheels carry their own HIP runtime; see
    INTEGRATION.md "Sharing a process with PyTorch")."""
    import torch.distributed as dist

    from . import ffi

    man_g = ffi.manifest(engine.lib, ffi.glow_hparams_c(glow_hp))
    man_v = ffi.manifest(engine.lib, ffi.hifigan_hparams_c(voc_hp))
    n_g, n_v = sum(n for _, n in man_g), sum(n for _, n in man_v)
    bg = bv = None
    if dist.get_rank() == 0:
        from .weights import build_blob

        bg, bv = build_blob(man_g, glow_sd), build_blob(man_v, voc_sd)
    tg = broadcast_blob(bg, n_g, device)
    tv = broadcast_blob(bv, n_v, device)
    g = engine.load_glow(glow_hp, device_ptr=tg.data_ptr())
    v = engine.load_hifigan(voc_hp, device_ptr=tv.data_ptr())
    return g, v


def micro_batches(indices: typing.Sequence[int], lengths: typing.Sequence[int], batch: int) -> typing.List[typing.List[int]]:
    """Length-bucketed micro-batches: sort this rank's utterances by
    id count and cut runs of `batch`, so the rows of one padded batch have similar
    lengths and little of the launch is padding."""
    order = sorted(indices, key=lambda i: (lengths[i], i))
    return [order[k : k + batch] for k in range(0, len(order), max(1, batch))]


def synthesize_shard(engine, glow: int, vocoder: int, id_rows: typing.Sequence[np.ndarray], rank: int, world: int,
                     noise_scale: float = 0.667, length_scale: float = 1.0, seed: int = 0, audio_settings=None,
                     batch: int = 1, speaker_ids: typing.Optional[typing.Sequence[int]] = None) -> typing.Dict[int, np.ndarray]:
    """This rank's share of the work list -> {utterance index: int16 audio}.
    `speaker_ids`: a multi-speaker voice's speaker per UTTERANCE (indexed like `id_rows` and the seeds; the reference's
    `speaker_id` setting, larynx/glow_tts.py:116-130) — required for such a voice, an error for a single-speaker one.
    `batch` > 1 runs length-bucketed micro-batches through one pair of calls each.  The
    kernels mask by row length and the device RNG stream of a row is keyed by the UTTERANCE
    (`seed + utterance index`, `mi355tts_glow_infer_rows`), so every utterance draws the noise
    field a call of its own would draw, whatever rank it lands on and whatever batch it rides
    in.  The audio is then equal to the single call's UP TO f32 SUMMATION ORDER (a padded batch
    picks other tile shapes than a batch-1 launch: +-1 int16 LSB in the tests), not bit for bit."""
    lengths = [len(r) for r in id_rows]
    if speaker_ids is not None and len(speaker_ids) != len(id_rows):
        raise ValueError("speaker_ids: one speaker per utterance of the work list")
    mine = lpt_assign(lengths, world)[rank]
    out: typing.Dict[int, np.ndarray] = {}
    hop = engine.hop(vocoder)
    for group in micro_batches(mine, lengths, batch):
        rows = [np.asarray(id_rows[i], np.int64) for i in group]
        mel = engine.glow_infer(glow, rows if len(rows) > 1 else rows[0], noise_scale, length_scale,
                                row_seeds=[seed + i for i in group], audio_settings=audio_settings,
                                speaker_ids=None if speaker_ids is None else [int(speaker_ids[i]) for i in group])
        _, i16 = engine.hifigan_infer(vocoder, mel, want_float=False)
        for b, i in enumerate(group):
            out[i] = i16[b, : int(mel.frames[b]) * hop].copy()
        mel.free()
    return out


def gather_in_order(local: typing.Dict[int, np.ndarray], total: int, dst: int = 0):
    """Collect every rank's results on `dst`, restored to sentence order."""
    import torch.distributed as dist

    world = dist.get_world_size()
    bucket = [None] * world if dist.get_rank() == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    merged: typing.Dict[int, np.ndarray] = {}
    for part in bucket:
        merged.update(part)
    return [merged[i] for i in range(total)]
