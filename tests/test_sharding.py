"""Multi-GPU path on CPU: world_size-2 gloo, weights broadcast from rank 0, LPT
sharding, ordered gather — each rank drives the emulator build of the C ABI."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

from larynx_amd.sharding import lpt_assign

REPO = Path(__file__).resolve().parent.parent


def test_lpt_assign_balances_and_is_deterministic():
    costs = [120, 60, 200, 90, 90, 150, 61, 75]
    shards = lpt_assign(costs, 3)
    assert sorted(i for s in shards for i in s) == list(range(8))
    loads = [sum(costs[i] for i in s) for s in shards]
    assert max(loads) - min(loads) <= max(costs)
    assert shards == lpt_assign(costs, 3)
    assert lpt_assign(costs, 1) == [list(range(8))]


def test_micro_batches_bucket_by_length():
    from larynx_amd.sharding import micro_batches

    lengths = [50, 10, 30, 12, 48, 31]
    groups = micro_batches([0, 1, 2, 3, 4, 5], lengths, 2)
    assert groups == [[1, 3], [2, 5], [4, 0]]
    assert micro_batches([4, 2], lengths, 8) == [[2, 4]]


def test_micro_batched_shard_equals_single_calls(emu_engine):
    from larynx_amd import hparams as HP
    from larynx_amd import sharding, synthetic

    g = emu_engine.load_glow(HP.TINY_GLOW, synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7))
    v = emu_engine.load_hifigan(HP.TINY_HIFIGAN, synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7))
    rng = np.random.default_rng(1)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols) for n in (7, 15, 9, 12, 6)]
    one = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.0)
    many = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.0, batch=3)
    assert sorted(one) == sorted(many) == list(range(5))
    for i in range(5):
        assert one[i].shape == many[i].shape
        assert np.abs(one[i].astype(np.int32) - many[i].astype(np.int32)).max() <= 1
    # with the device noise generator ON: a row's stream is keyed by the utterance (seed + index), so the micro-batched
    # shard still reproduces the single calls — and a direct batch-1 call with that seed gives the same audio
    one = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.667, seed=40)
    many = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.667, seed=40, batch=3)
    quiet = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.0)
    for i in range(5):
        assert np.abs(one[i].astype(np.int32) - many[i].astype(np.int32)).max() <= 1
        assert not np.array_equal(one[i], quiet[i])  # the noise really is on
    mel = emu_engine.glow_infer(g, rows[3], 0.667, 1.0, seed=43)
    _, direct = emu_engine.hifigan_infer(v, mel, want_float=False)
    assert np.abs(direct[0, : one[3].shape[0]].astype(np.int32) - one[3].astype(np.int32)).max() <= 1


def test_shard_of_a_multi_speaker_voice(emu_engine):
    """`speaker_ids` ride with the utterances (indexed like the seeds): the micro-batched shard of a multi-speaker voice equals
    the single calls with each utterance's own speaker, a missing list is the library's error, a short one a ValueError."""
    import dataclasses

    import pytest

    from larynx_amd import ffi
    from larynx_amd import hparams as HP
    from larynx_amd import sharding, synthetic

    hp = dataclasses.replace(HP.TINY_GLOW, n_speakers=3, gin_channels=20)
    g = emu_engine.load_glow(hp, synthetic.make_glow_state_dict(hp, seed=9))
    v = emu_engine.load_hifigan(HP.TINY_HIFIGAN, synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=9))
    rng = np.random.default_rng(2)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, hp.num_symbols) for n in (8, 13, 6, 10)]
    spk = [2, 0, 1, 2]
    try:
        one = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.667, seed=70, speaker_ids=spk)
        many = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.667, seed=70, batch=3, speaker_ids=spk)
        other = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.667, seed=70, speaker_ids=[0, 0, 0, 0])
        for i in range(4):
            assert one[i].shape == many[i].shape
            assert np.abs(one[i].astype(np.int32) - many[i].astype(np.int32)).max() <= 1
            mel = emu_engine.glow_infer(g, rows[i], 0.667, 1.0, seed=70 + i, speaker_ids=spk[i])
            _, direct = emu_engine.hifigan_infer(v, mel, want_float=False)
            assert np.array_equal(direct[0, : one[i].shape[0]], one[i])
        assert any(one[i].shape != other[i].shape or not np.array_equal(one[i], other[i]) for i in (0, 2, 3))  # the speakers matter
        with pytest.raises(ffi.Mi355ttsError, match="speaker"):
            sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1)
        with pytest.raises(ValueError):
            sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, speaker_ids=[0, 1])
    finally:
        emu_engine.unload(g)
        emu_engine.unload(v)


def _worker(rank, world, port, lib, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    from larynx_amd import hparams as HP
    from larynx_amd import sharding, synthetic
    from larynx_amd.engine import Engine

    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = Engine(0, library_path=lib)
    gsd = vsd = None
    if rank == 0:  # only rank 0 has the checkpoint
        gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7)
        vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7)
    g, v = sharding.load_models_broadcast(eng, HP.TINY_GLOW, HP.TINY_HIFIGAN, gsd, vsd, device="cpu")
    rng = np.random.default_rng(0)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols) for n in (9, 14, 6, 11, 8)]
    local = sharding.synthesize_shard(eng, g, v, rows, rank, world, noise_scale=0.0)
    merged = sharding.gather_in_order(local, len(rows))
    if rank == 0:
        np.savez(Path(out_dir) / "merged.npz", *merged)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def test_two_rank_gloo_matches_single_process(emu_library, emu_engine, tmp_path):
    import torch.multiprocessing as mp

    from larynx_amd import hparams as HP
    from larynx_amd import synthetic

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(emu_library), str(tmp_path)), nprocs=2, join=True)
    z = np.load(tmp_path / "merged.npz")
    merged = [z[f"arr_{i}"] for i in range(5)]
    gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7)
    vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7)
    g = emu_engine.load_glow(HP.TINY_GLOW, gsd)
    v = emu_engine.load_hifigan(HP.TINY_HIFIGAN, vsd)
    rng = np.random.default_rng(0)
    rows = [synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols) for n in (9, 14, 6, 11, 8)]
    for i, ids in enumerate(rows):
        mel = emu_engine.glow_infer(g, ids, 0.0, 1.0)
        _, i16 = emu_engine.hifigan_infer(v, mel, want_float=False)
        n = int(mel.frames[0]) * HP.TINY_HIFIGAN.hop
        assert np.array_equal(merged[i], i16[0, :n])


def test_lpt_over_eight_ranks_balances_config3():
    """BASELINE config 3 as `bench.py` shards it on an 8-GPU node: 256 utterances, P ~ clip(N(120, 15), 60, 200), LPT over 8
    ranks — every utterance exactly once, 32 per rank, and the heaviest rank within 2 % of the mean load."""
    rng = np.random.default_rng(1234)  # the recipe of bench.config3_ids
    lengths = [int(p) for p in np.clip(np.rint(rng.normal(120.0, 15.0, 256)), 60, 200)]
    shards = lpt_assign(lengths, 8)
    assert sorted(i for s in shards for i in s) == list(range(256))
    loads = [sum(lengths[i] for i in s) for s in shards]
    assert max(loads) * 8 / sum(loads) <= 1.02, loads
    assert all(len(s) == 32 for s in shards), [len(s) for s in shards]
    for w in (2, 4):  # the driver's other scaling points
        loads = [sum(lengths[i] for i in s) for s in lpt_assign(lengths, w)]
        assert max(loads) * w / sum(loads) <= 1.02


def _worker8(rank, world, port, lib, out_dir):
    sys.path.insert(0, str(REPO))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "1"
    import torch.distributed as dist

    from larynx_amd import hparams as HP
    from larynx_amd import sharding, synthetic
    from larynx_amd.engine import Engine

    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = Engine(0, library_path=lib)
    gsd = vsd = None
    if rank == 0:  # only rank 0 has the checkpoint
        gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7)
        vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7)
    g, v = sharding.load_models_broadcast(eng, HP.TINY_GLOW, HP.TINY_HIFIGAN, gsd, vsd, device="cpu")
    rows = _rows8()
    local = sharding.synthesize_shard(eng, g, v, rows, rank, world, noise_scale=0.667, seed=900)
    assert sorted(local) == sharding.lpt_assign([len(r) for r in rows], world)[rank]
    merged = sharding.gather_in_order(local, len(rows))
    if rank == 0:
        np.savez(Path(out_dir) / "merged8.npz", *merged)
    dist.barrier()
    dist.destroy_process_group()
    eng.close()


def _rows8():
    from larynx_amd import hparams as HP
    from larynx_amd import synthetic

    # config 3's length distribution scaled to the emulator's sizes: 20 utterances over 8 ranks (2-3 per rank)
    rng = np.random.default_rng(1234)
    lens = np.clip(np.rint(rng.normal(12.0, 1.5, 20)), 6, 20).astype(int)
    return [synthetic.synthetic_phoneme_ids(rng, int(p), HP.TINY_GLOW.num_symbols) for p in lens]


def test_eight_rank_gloo_matches_single_process(emu_library, emu_engine, tmp_path):
    """The 8-GPU job of BASELINE config 3 on CPU: world size 8 (gloo), weight broadcast from rank 0, LPT shards over 8
    bins, device noise ON with utterance-keyed streams, ordered gather from 8 ranks — equal to one process, bit for bit."""
    import torch.multiprocessing as mp

    from larynx_amd import hparams as HP
    from larynx_amd import sharding, synthetic

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker8, args=(8, port, str(emu_library), str(tmp_path)), nprocs=8, join=True)
    rows = _rows8()
    z = np.load(tmp_path / "merged8.npz")
    merged = [z[f"arr_{i}"] for i in range(len(rows))]
    g = emu_engine.load_glow(HP.TINY_GLOW, synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7))
    v = emu_engine.load_hifigan(HP.TINY_HIFIGAN, synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7))
    solo = sharding.synthesize_shard(emu_engine, g, v, rows, 0, 1, noise_scale=0.667, seed=900)
    shards = sharding.lpt_assign([len(r) for r in rows], 8)
    assert all(2 <= len(s) <= 3 for s in shards)
    for i in range(len(rows)):
        assert np.array_equal(merged[i], solo[i]), i
