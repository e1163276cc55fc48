"""A/B of library builds inside ONE process, alternating (A B A B ...): box-to-box and warm-up drift cancel.
`python tools/ab_inproc.py [--quality high] [--frames 800] [--rounds 8] [--calls 20] [--streams 1] libA.so libB.so ...`
Per build: wall-clock ms per vocoder call (device-resident mel in, device waveform out; `--streams` caller threads), and
the ResBlock class's event-timed ms per call from a profiled single-stream pass (what bench.py's roofline uses)."""
import argparse
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from larynx_amd import ffi, synthetic
from larynx_amd import hparams as HP
from larynx_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--quality", default="high")
ap.add_argument("--frames", type=int, default=800)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--calls", type=int, default=20)
ap.add_argument("--streams", type=int, default=1)
args = ap.parse_args()

hp = HP.VOCODER_QUALITY[args.quality]
sd = synthetic.make_hifigan_state_dict(hp, seed=1234)
rng = np.random.default_rng(5)
mel = (rng.standard_normal((1, hp.num_mels, args.frames)) * 1.5 - 4).astype(np.float32)
S = args.frames * hp.hop
envs = []
for lib in args.libs:
    eng = Engine(device=0, library_path=lib)
    v = eng.load_hifigan(hp, sd)
    eng.reserve(args.streams + 1, 0, v, max_batch=1, max_frames=args.frames + 8)
    mbs = [eng.mel_from_numpy(mel) for _ in range(args.streams)]
    wav = [torch.empty(S, dtype=torch.float32, device="cuda:0") for _ in range(args.streams)]
    i16 = [torch.empty(S, dtype=torch.int16, device="cuda:0") for _ in range(args.streams)]
    envs.append((lib, eng, v, mbs, wav, i16))

pool = ThreadPoolExecutor(args.streams)


def calls(env, n):
    _, eng, v, mbs, wav, i16 = env

    def work(s):
        for _ in range(n):
            eng.hifigan_infer_raw(v, mbs[s], wav[s].data_ptr(), i16[s].data_ptr(), S, flags=ffi.OUT_DEVICE)

    list(pool.map(work, range(args.streams)))


for env in envs:
    calls(env, 5)
wall = {e[0]: [] for e in envs}
cls = {e[0]: [] for e in envs}
for r in range(args.rounds):
    for env in envs:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        calls(env, args.calls)
        wall[env[0]].append((time.perf_counter() - t0) * 1e3 / (args.calls * args.streams))
    for env in envs:
        eng = env[1]
        eng.set_profiling(True)
        eng.profile_reset()
        for _ in range(args.calls):
            eng.hifigan_infer_raw(env[2], env[3][0], env[4][0].data_ptr(), env[5][0].data_ptr(), S, flags=ffi.OUT_DEVICE)
        p = eng.profile()["conv_mfma.hifigan_resblock"]
        eng.set_profiling(False)
        cls[env[0]].append(p["ms"] / args.calls)
for lib in wall:
    w, c = np.array(wall[lib]), np.array(cls[lib])
    print("%-40s wall ms/call median %.4f (min %.4f, max %.4f) | ResBlock class, events, ms/call median %.4f (min %.4f max %.4f)"
          % (lib.split("/")[-1], np.median(w), w.min(), w.max(), np.median(c), c.min(), c.max()))
