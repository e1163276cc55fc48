#!/usr/bin/env python
"""Where a call's time goes UNDER LOAD: N threads run the two-call form (GlowTTS pass, then the vocoder) on the standard utterance
and stamp the phases; then the same utterances with the acoustic passes produced by G dedicated threads (a two-stage host
pipeline: the vocoder threads only ever run vocoder calls).  python tools/phase_probe.py [threads=8] [calls=40] [name=value options]"""
import queue
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
rng = np.random.default_rng(7)
ids = [synthetic.synthetic_phoneme_ids(rng, 120, HP.LJSPEECH.num_symbols) for _ in range(16)]
eng.reserve(nthr + 4, g, v, max_batch=1, max_ids=120, max_frames=1024)


def two_call(n_threads):
    tg, tv = [], []
    lock = threading.Lock()
    bar = threading.Barrier(n_threads + 1)

    def worker(i):
        lg, lv = [], []
        for k in range(ncall + 3):
            if k == 3:
                bar.wait()
            t0 = time.perf_counter()
            mel = eng.glow_infer(g, ids[(i + k) % 16], 0.667, 0.65, seed=i * 1000 + k, audio_settings=s)
            t1 = time.perf_counter()
            eng.hifigan_infer(v, mel, want_float=False, want_int16=True)
            t2 = time.perf_counter()
            mel.free()
            if k >= 3:
                lg.append(t1 - t0)
                lv.append(t2 - t1)
        with lock:
            tg.extend(lg)
            tv.extend(lv)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    print(f"{n_threads} threads, GlowTTS then vocoder per call: {n_threads * ncall / dt:.1f} utterances/s; per call GlowTTS phase {1e3 * np.median(tg):.2f} ms "
          f"(p90 {1e3 * np.percentile(tg, 90):.2f}), vocoder phase {1e3 * np.median(tv):.2f} ms (p90 {1e3 * np.percentile(tv, 90):.2f})")


def sleep_then_vocoder(n_threads, sleep_ms):
    """The acoustic pass replaced by a HOST sleep of its under-load duration: is it the GPU work of GlowTTS that costs the
    vocoder calls, or only that a caller does not feed the GPU while its pass crawls?"""
    mels = [eng.glow_infer(g, ids[i % 16], 0.667, 0.65, seed=i, audio_settings=s) for i in range(n_threads)]
    bar = threading.Barrier(n_threads + 1)

    def worker(i):
        for k in range(ncall + 3):
            if k == 3:
                bar.wait()
            time.sleep(sleep_ms * 1e-3)
            eng.hifigan_infer(v, mels[i], want_float=False, want_int16=True)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(n_threads)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    for m in mels:
        m.free()
    print(f"{n_threads} threads, host sleep {sleep_ms} ms then vocoder per call: {n_threads * ncall / dt:.1f} utterances/s")


def staged(n_voc, n_glow, depth=16):
    total = n_voc * ncall
    q = queue.Queue(maxsize=depth)
    todo = queue.SimpleQueue()
    for k in range(total + 3 * n_voc):
        todo.put(k)
    tg, tv = [], []

    def producer():
        while True:
            try:
                k = todo.get_nowait()
            except queue.Empty:
                return
            t0 = time.perf_counter()
            mel = eng.glow_infer(g, ids[k % 16], 0.667, 0.65, seed=k, audio_settings=s)
            tg.append(time.perf_counter() - t0)
            q.put(mel)

    done = [0]
    lock = threading.Lock()
    t_start = [None]

    def consumer():
        while True:
            mel = q.get()
            if mel is None:
                return
            t0 = time.perf_counter()
            eng.hifigan_infer(v, mel, want_float=False, want_int16=True)
            tv.append(time.perf_counter() - t0)
            mel.free()
            with lock:
                done[0] += 1
                if done[0] == 3 * n_voc:
                    t_start[0] = time.perf_counter()

    pr = [threading.Thread(target=producer) for _ in range(n_glow)]
    co = [threading.Thread(target=consumer) for _ in range(n_voc)]
    for t in pr + co:
        t.start()
    for t in pr:
        t.join()
    for _ in co:
        q.put(None)
    for t in co:
        t.join()
    dt = time.perf_counter() - t_start[0]
    print(f"{n_glow} GlowTTS thread(s) -> queue -> {n_voc} vocoder threads: {total / dt:.1f} utterances/s; GlowTTS call {1e3 * np.median(tg):.2f} ms, "
          f"vocoder call {1e3 * np.median(tv):.2f} ms")


eng.reserve(34, g, v, max_batch=1, max_ids=120, max_frames=1024)
for r in range(2):
    two_call(nthr)
    two_call(16)
    for n, ms in ((8, 0.0), (8, 4.0), (8, 8.0), (8, 12.0), (12, 8.0), (16, 8.0), (16, 12.0), (24, 12.0)):
        sleep_then_vocoder(n, ms)
    staged(nthr, 1)
    staged(nthr, 2)
eng.close()
