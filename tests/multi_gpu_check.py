"""TEST HELPER — one rank of the two-GPU check (tests/test_multi_gpu.py starts two of these under torch.distributed.run).

The path's ONE collective on real hardware with more than one rank: rank 0 holds the folded weight blob,
`mi355tts_broadcast_weights` sends it over a caller-owned RCCL communicator (made here with ctypes, its unique id passed
through a gloo group), every rank loads the model from its device buffer and must synthesise exactly what a model loaded
from host memory gives.  Then the utterance shards: every rank runs its LPT share and rank 0 gathers them in order."""
import ctypes
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    import torch
    import torch.distributed as dist

    from larynx_amd import hparams as HP
    from larynx_amd import sharding, synthetic
    from larynx_amd.engine import Engine

    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    torch.cuda.init()
    dist.init_process_group("gloo")
    path = "/opt/rocm/lib/librccl.so"
    rccl = ctypes.CDLL(path)

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    rccl.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    uid = UniqueId()
    if rank == 0:
        assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    box = [bytes(uid.internal) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    ctypes.memmove(ctypes.byref(uid), box[0], 128)
    comm = ctypes.c_void_p()
    assert rccl.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0

    eng = Engine(device=local)
    hp = HP.HIFIGAN_LOW
    sd = synthetic.make_hifigan_state_dict(hp, seed=1234)  # seeded: every rank can check what it received
    blob = eng.hifigan_blob(hp, sd)
    t = torch.from_numpy(blob).cuda() if rank == 0 else torch.zeros(blob.size, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    eng.broadcast_weights(comm.value, 0, t.data_ptr(), blob.size, rccl_library=path)
    assert np.array_equal(t.cpu().numpy(), blob), f"rank {rank}: broadcast blob differs"
    v_dev = eng.load_hifigan(hp, device_ptr=t.data_ptr())
    v_host = eng.load_hifigan(hp, sd)
    melin = (np.random.default_rng(3).standard_normal((1, 80, 40)) * 2).astype(np.float32)
    a, _ = eng.hifigan_infer(v_dev, eng.mel_from_numpy(melin))
    b, _ = eng.hifigan_infer(v_host, eng.mel_from_numpy(melin))
    assert np.array_equal(a, b), f"rank {rank}: model from the broadcast buffer differs"
    rccl.ncclCommDestroy(comm)

    # utterance shards across the ranks, ordered gather (no collective inside an utterance)
    ghp = HP.LJSPEECH
    g = eng.load_glow(ghp, synthetic.make_glow_state_dict(ghp, seed=1234))
    rng = np.random.default_rng(0)
    rows = [synthetic.synthetic_phoneme_ids(rng, int(n), ghp.num_symbols) for n in rng.integers(20, 60, 10)]
    local_out = sharding.synthesize_shard(eng, g, v_host, rows, rank, world, noise_scale=0.667, seed=7)
    merged = sharding.gather_in_order(local_out, len(rows))
    if rank == 0:
        assert len(merged) == len(rows) and all(m.dtype == np.int16 and m.size > 0 for m in merged)
        mine = sharding.synthesize_shard(eng, g, v_host, rows, 0, 1, noise_scale=0.667, seed=7)  # the whole list on one GPU
        for i in range(len(rows)):
            assert np.array_equal(mine[i], merged[i]), f"utterance {i} differs between the 1-rank and the {world}-rank run"
    dist.barrier()
    print(f"MULTI_GPU_CHECK rank {rank}/{world} OK", flush=True)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
