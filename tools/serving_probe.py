#!/usr/bin/env python
"""Serving-shaped measurements on one GPU.

config 3: 256 synthetic utterances with the fixture layout, P_i = clip(round(N(120,15)), 60, 200),
          this rank's LPT share, run (a) one call per utterance with C calls in flight and
          (b) as length-bucketed micro-batches; reports utterances/s and x real time.
config 5: 210 sentences cycling three resident voices (en V=46 / de V=54 / fr V=42) against one
          'high' vocoder, C host threads, results delivered in submission order; reports the
          time to the first sentence's audio and the sustained x real time.

Run on the GPU box:  python tools/serving_probe.py [--length-scale 0.65] [--threads 3] [--batch 8]
Prints one JSON object.
"""
import argparse
import json
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd import hparams as HP  # noqa: E402
from larynx_amd import sharding, synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from larynx_amd.engine import Engine  # noqa: E402

SR = 22050


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--length-scale", type=float, default=0.65)
    ap.add_argument("--threads", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--utterances", type=int, default=256)
    args = ap.parse_args()
    eng = Engine(0)
    s = ljspeech_audio_settings()
    vhp = HP.HIFIGAN_HIGH
    v = eng.load_hifigan(vhp, synthetic.make_hifigan_state_dict(vhp, seed=1234))
    voices = []
    for ghp in (HP.LJSPEECH, HP.THORSTEN, HP.SIWIS):
        voices.append((ghp, eng.load_glow(ghp, synthetic.make_glow_state_dict(ghp, seed=1234))))
    hop = eng.hop(v)
    out = {}

    # ---------------------------------------------------------------- config 3
    rng = np.random.default_rng(1234)
    P = np.clip(np.round(rng.normal(120, 15, args.utterances)), 60, 200).astype(int)
    rows = [synthetic.synthetic_phoneme_ids(rng, int(p), HP.LJSPEECH.num_symbols) for p in P]
    g = voices[0][1]

    def one(i):
        mel = eng.glow_infer(g, rows[i], 0.667, args.length_scale, seed=1234 + i, audio_settings=s)
        _, i16 = eng.hifigan_infer(v, mel, want_float=False)
        n = int(mel.frames[0]) * hop
        mel.free()
        return n

    for i in range(3):
        one(i)  # warm-up: arenas, streams
    with ThreadPoolExecutor(args.threads) as pool:
        list(pool.map(one, range(args.threads)))
        t0 = time.perf_counter()
        samples = sum(pool.map(one, range(len(rows))))
        dt = time.perf_counter() - t0
    out["config3_single_calls"] = {"utterances": len(rows), "calls_in_flight": args.threads, "seconds": dt,
                                   "utterances_per_s": len(rows) / dt, "x_realtime": samples / SR / dt,
                                   "mean_ids": float(P.mean()), "audio_s": samples / SR}
    sharding.synthesize_shard(eng, g, v, rows[: args.batch], 0, 1, length_scale=args.length_scale, audio_settings=s, batch=args.batch)
    t0 = time.perf_counter()
    res = sharding.synthesize_shard(eng, g, v, rows, 0, 1, length_scale=args.length_scale, audio_settings=s, batch=args.batch)
    dt = time.perf_counter() - t0
    samples = sum(a.shape[0] for a in res.values())
    out["config3_micro_batches"] = {"utterances": len(rows), "batch": args.batch, "seconds": dt,
                                    "utterances_per_s": len(rows) / dt, "x_realtime": samples / SR / dt}

    # ---------------------------------------------------------------- config 5
    n_sent = 210
    rng = np.random.default_rng(5)
    sents = []
    for i in range(n_sent):
        ghp, gm = voices[i % 3]
        sents.append((gm, synthetic.synthetic_phoneme_ids(rng, int(rng.integers(40, 160)), ghp.num_symbols)))

    def sentence(i):
        gm, ids = sents[i]
        mel = eng.glow_infer(gm, ids, 0.667, args.length_scale, seed=i, audio_settings=s)
        _, i16 = eng.hifigan_infer(v, mel, want_float=False)
        n = int(mel.frames[0]) * hop
        mel.free()
        return i16[0, :n]

    with ThreadPoolExecutor(args.threads) as pool:
        list(pool.map(sentence, range(3)))
        t0 = time.perf_counter()
        futs = [pool.submit(sentence, i) for i in range(n_sent)]
        first = None
        total = 0
        for f in futs:  # in-order delivery
            a = f.result()
            if first is None:
                first = time.perf_counter() - t0
            total += a.shape[0]
        dt = time.perf_counter() - t0
    out["config5_three_voices_in_order"] = {"sentences": n_sent, "threads": args.threads, "seconds": dt,
                                            "ms_to_first_audio": 1e3 * first, "x_realtime": total / SR / dt,
                                            "sentences_per_s": n_sent / dt, "audio_s": total / SR}
    print(json.dumps(out))
    eng.close()


if __name__ == "__main__":
    main()
