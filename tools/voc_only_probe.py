#!/usr/bin/env python
"""What GlowTTS costs under load: throughput of the vocoder alone (mi355tts_hifigan_infer on ONE resident mel of the
standard utterance, N calls in flight) next to the full fused call, same threads, same box.
Usage: python tools/voc_only_probe.py [threads=8] [calls_per_thread=40] [name=value context options ...]"""
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd import hparams as HP, synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from larynx_amd.engine import Engine  # noqa: E402

nthr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ncall = int(sys.argv[2]) if len(sys.argv) > 2 else 40
eng = Engine(0)
for kv in sys.argv[3:]:
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
g = eng.load_glow(HP.LJSPEECH, synthetic.make_glow_state_dict(HP.LJSPEECH, seed=1234))
v = eng.load_hifigan(HP.HIFIGAN_HIGH, synthetic.make_hifigan_state_dict(HP.HIFIGAN_HIGH, seed=1234))
s = ljspeech_audio_settings()
ids = synthetic.synthetic_phoneme_ids(np.random.default_rng(7), 120, HP.LJSPEECH.num_symbols)
eng.reserve(nthr, g, v, max_batch=1, max_ids=120, max_frames=1024)
mels = [eng.glow_infer(g, ids, 0.667, 0.65, seed=i, audio_settings=s) for i in range(nthr)]
print("frames", [int(m.frames[0]) for m in mels][:3])


def run(fn):
    def worker(i):
        for _ in range(3):
            fn(i)
        bar.wait()
        for _ in range(ncall):
            fn(i)

    bar = threading.Barrier(nthr + 1)
    th = [threading.Thread(target=worker, args=(i,)) for i in range(nthr)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return nthr * ncall / (time.perf_counter() - t0)


for r in range(2):
    voc = run(lambda i: eng.hifigan_infer(v, mels[i], want_float=False, want_int16=True))
    glow = run(lambda i: eng.glow_infer(g, ids, 0.667, 0.65, seed=i, audio_settings=s))
    full = run(lambda i: eng.synthesize(g, v, ids, 0.667, 0.65, seed=i, audio_settings=s, frames_per_id_guess=12.0 / 0.65))
    # the same vocoder load with ONE extra thread running batched GlowTTS passes (B = nthr rows per call) back to back: what
    # a pass costs the vocoder calls = the price of GlowTTS if concurrent callers' passes were coalesced
    stop = threading.Event()
    passes = [0]

    def glow_bg():
        rows = [ids] * nthr
        while not stop.is_set():
            eng.glow_infer(g, rows, 0.667, 0.65, seed=3, audio_settings=s).free()
            passes[0] += 1

    bg = threading.Thread(target=glow_bg)
    bg.start()
    t_bg = time.perf_counter()
    voc_bg = run(lambda i: eng.hifigan_infer(v, mels[i], want_float=False, want_int16=True))
    dt_bg = time.perf_counter() - t_bg
    stop.set()
    bg.join()
    n_voc = nthr * (ncall + 3)
    print(f"   vocoder with batched GlowTTS passes in the background: {voc_bg:.1f} /s ({1e3 / voc_bg:.3f} ms); {passes[0]} passes of {nthr} rows in "
          f"{dt_bg * 1e3:.0f} ms = {passes[0] * nthr / n_voc:.2f} rows per vocoder call -> extra time per vocoder call "
          f"{1e3 / voc_bg - 1e3 / voc:.3f} ms, per GlowTTS row {(1e3 / voc_bg - 1e3 / voc) * n_voc / max(passes[0] * nthr, 1):.3f} ms")
    print(f"{nthr} calls in flight: vocoder alone {voc:.1f} /s ({1e3 / voc:.3f} ms), GlowTTS alone {glow:.1f} /s ({1e3 / glow:.3f} ms), "
          f"full call {full:.1f} /s ({1e3 / full:.3f} ms); sum of the parts {1e3 / voc + 1e3 / glow:.3f} ms")
eng.close()
