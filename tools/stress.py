#!/usr/bin/env python
"""Soak test on one GPU: N host threads hammer one engine with random-length sentences on
three resident voices and five vocoders ('high' exact, 'low' exact, 'high' in the split-bf16 mode, 'high' and 'medium' in the
native fp16 mode), a share of
the calls through the fused one-call entry with pause padding, a share with the denoiser on; every
result must be finite, of the expected length, identical when the whole job list is run a
second time, and equal to a single-threaded recomputation for a sample.  VRAM use after
pass 1 and pass 2 must match (leak check: workspaces are grow-only per worker).
Run on the GPU box:  python tools/stress.py [--calls 1500] [--threads 4]"""
import argparse
import json
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd import hparams as HP  # noqa: E402
from larynx_amd import synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from larynx_amd.engine import Engine  # noqa: E402


def vram_used():
    try:
        out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=30).stdout
        d = json.loads(out)
        return int(next(iter(d.values()))["VRAM Total Used Memory (B)"])
    except Exception:  # noqa: BLE001 - diagnostics only
        return -1


def unload_leg(eng, threads: int, seconds: float):
    """A voice is unloaded and reloaded (and the schedule options are flipped) WHILE `threads` host threads synthesise with
    it: calls that already hold the model finish with the right audio (models are pinned per call), calls that arrive
    after the unload fail cleanly with MI355TTS_ERR_NO_MODEL, nothing crashes, and VRAM use returns to where it was."""
    import threading

    from larynx_amd import ffi

    s = ljspeech_audio_settings()
    ghp, vhp = HP.LJSPEECH, HP.HIFIGAN_MEDIUM
    gsd = synthetic.make_glow_state_dict(ghp, seed=1234)
    vsd = synthetic.make_hifigan_state_dict(vhp, seed=1234)
    rng = np.random.default_rng(5)
    rows = [synthetic.synthetic_phoneme_ids(rng, int(n), ghp.num_symbols) for n in rng.integers(10, 120, 12)]
    g0, v0 = eng.load_glow(ghp, gsd), eng.load_hifigan(vhp, vsd)
    want = [eng.synthesize(g0, v0, r, 0.667, 0.8, seed=k, audio_settings=s)[2] for k, r in enumerate(rows)]
    ids = {"g": g0, "v": v0}
    stop = threading.Event()
    stats = {"ok": 0, "no_model": 0, "bad": 0}
    lock = threading.Lock()

    def worker(t):
        k = t
        while not stop.is_set():
            k = (k + 1) % len(rows)
            try:
                i16 = eng.synthesize(ids["g"], ids["v"], rows[k], 0.667, 0.8, seed=k, audio_settings=s)[2]
                good = np.array_equal(i16, want[k])
                with lock:
                    stats["ok" if good else "bad"] += 1
                    if not good:
                        stats.setdefault("errors", []).append(f"job {k}: int16 differs by {int(np.abs(i16.astype(np.int32) - want[k].astype(np.int32)).max())}")
            except ffi.Mi355ttsError as e:
                with lock:
                    stats["no_model" if e.code == -5 else "bad"] += 1  # MI355TTS_ERR_NO_MODEL: looked the id up after the unload
                    if e.code != -5:
                        stats.setdefault("errors", []).append(str(e)[:120])
                time.sleep(0.001)

    eng.reserve(threads + 2, g0, v0, max_batch=1, max_ids=128, max_frames=128 * 12)  # workspaces sized up front: VRAM must come back flat
    used0 = vram_used()
    th = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for t in th:
        t.start()
    t_end = time.perf_counter() + seconds
    cycles = 0
    while time.perf_counter() < t_end:
        time.sleep(0.05)
        old_g, old_v = ids["g"], ids["v"]
        eng.unload(old_g)  # in-flight calls keep their pin; new ones fail with NO_MODEL until the reload below
        eng.set_option("mrf_group", cycles % 2)  # (both schedules give the same bits; mrf_small changes the summation order)
        eng.set_option("adaptive_schedule", (cycles // 2) % 2)
        time.sleep(0.005)
        ids["g"] = eng.load_glow(ghp, gsd)
        eng.unload(old_v)
        ids["v"] = eng.load_hifigan(vhp, vsd)
        cycles += 1
    stop.set()
    for t in th:
        t.join()
    eng.set_option("mrf_group", 1)
    eng.set_option("adaptive_schedule", 0)
    eng.unload(ids["g"])
    eng.unload(ids["v"])
    used1 = vram_used()
    assert stats["bad"] == 0 and stats["ok"] > 0, stats
    return {"unload_reload_cycles": cycles, "threads": threads, **stats, "vram_delta_bytes": used1 - used0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unload-leg", type=float, default=0.0, metavar="SECONDS",
                    help="only the unload/reload-under-load leg, for this long (6 threads unless --threads is given)")
    ap.add_argument("--calls", type=int, default=1500)
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--no-reserve", action="store_true", help="leave the workspaces grow-only (no mi355tts_reserve up front)")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE",
                    help="mi355tts_set_option before anything runs, e.g. glow_coalesce=1 (the fused calls then share GlowTTS passes)")
    args = ap.parse_args()
    eng = Engine(0)
    for kv in args.set_option:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if args.unload_leg > 0:
        print(json.dumps(unload_leg(eng, args.threads if args.threads != 4 else 6, args.unload_leg)))
        eng.close()
        return
    s = ljspeech_audio_settings()
    from larynx_amd import ffi

    vocs = [(hp, eng.load_hifigan(hp, synthetic.make_hifigan_state_dict(hp, seed=1234)))
            for hp in (HP.HIFIGAN_HIGH, HP.HIFIGAN_LOW, HP.HIFIGAN_HIGH, HP.HIFIGAN_HIGH, HP.HIFIGAN_MEDIUM)]
    eng.set_precision(vocs[2][1], ffi.PRECISION_BF16X3)
    eng.set_precision(vocs[3][1], ffi.PRECISION_F16)
    eng.set_precision(vocs[4][1], ffi.PRECISION_F16)
    voices = [(hp, eng.load_glow(hp, synthetic.make_glow_state_dict(hp, seed=1234))) for hp in (HP.LJSPEECH, HP.THORSTEN, HP.SIWIS, HP.LJSPEECH)]
    assert eng.set_precision(voices[3][1], ffi.PRECISION_F16) == 0  # the fourth voice: the decoder's WaveNets in fp16 (csrc/wn_f16.h)
    rng = np.random.default_rng(99)
    jobs = []
    for i in range(args.calls):
        ghp, g = voices[int(rng.integers(len(voices)))]
        vhp, v = vocs[int(rng.integers(len(vocs)))]
        B = 1 if rng.random() < 0.8 else int(rng.integers(2, 5))
        rows = [synthetic.synthetic_phoneme_ids(rng, int(rng.integers(1, 220)), ghp.num_symbols) for _ in range(B)]
        jobs.append((i, g, v, vhp, rows, 0.01 if rng.random() < 0.2 else 0.0, bool(rng.random() < 0.4)))

    # every worker a call can land on exists and is sized for the largest job before the first pass (mi355tts_reserve):
    # VRAM use must then be FLAT from pass 1 on.  (Without it the per-worker workspaces are grow-only and keep
    # rising until every worker has met the largest job.)
    if not args.no_reserve:
        eng.reserve(args.threads + 1, voices[0][1], vocs[0][1], max_batch=4, max_ids=220, max_frames=220 * 12, denoiser=True,
                    max_pad_samples=2 * 37 + 11)

    def run(job):
        i, g, v, vhp, rows, dn, fused = job
        if fused:  # the one-call entry with SSML pause padding; frame counts come back with the audio
            pb, pa = 37 * (i % 3), 11 * (i % 2)
            frames, wav, i16 = eng.synthesize(g, v, rows if len(rows) > 1 else rows[0], 0.667, 0.8, seed=i, audio_settings=s,
                                              pad_before=pb, pad_after=pa, want_float=True)
            frames = [int(f) for f in frames]
            assert np.isfinite(wav).all()
            for b, f in enumerate(frames):
                n = f * vhp.hop
                assert np.all(i16[b, :pb] == 0) and np.all(i16[b, pb + n :] == 0)
                assert n == 0 or np.abs(wav[b, pb : pb + n]).max() > 0
            return i, frames, i16
        mel = eng.glow_infer(g, rows if len(rows) > 1 else rows[0], 0.667, 0.8, seed=i, audio_settings=s)
        frames = [int(f) for f in mel.frames]
        if dn > 0 and min(frames) * vhp.hop <= 1024:
            dn = 0.0  # shorter than one STFT frame: the reference raises, so does the library
        wav, i16 = eng.hifigan_infer(v, mel, denoiser_strength=dn)
        mel.free()
        assert np.isfinite(wav).all()
        for b, f in enumerate(frames):
            n = f * vhp.hop
            assert np.all(i16[b, n:] == 0)
            assert dn > 0 or np.abs(wav[b, :n]).max() > 0
        return i, frames, i16

    used0 = vram_used()
    t0 = time.perf_counter()
    with ThreadPoolExecutor(args.threads) as pool:
        results = list(pool.map(run, jobs))
        dt = time.perf_counter() - t0
        used1 = vram_used()  # models + grow-only per-worker workspaces sized for the largest call seen
        # same work again, several times: results identical; the workspaces are grow-only per
        # worker, so VRAM use may still rise while a worker meets its first largest call, and
        # must then stay flat
        trail = [used1]
        for _ in range(4):
            again = list(pool.map(run, jobs))
            trail.append(vram_used())
            for a, b in zip(results, again):
                assert a[1] == b[1] and np.array_equal(a[2], b[2])
    used2 = trail[-1]
    # determinism under concurrency: recompute a sample single-threaded
    for k in range(0, len(jobs), max(1, len(jobs) // 25)):
        i, frames, i16 = run(jobs[k])
        assert frames == results[k][1] and np.array_equal(i16, results[k][2]), f"job {k} differs when recomputed"
    for _, g in voices:
        eng.unload(g)
    for _, v in vocs:
        eng.unload(v)
    eng.close()
    print(json.dumps({"calls": args.calls, "threads": args.threads, "seconds": dt, "calls_per_s": args.calls / dt,
                      "reserved_up_front": not args.no_reserve, "vram_used_before": used0, "vram_used_after_each_pass": trail,
                      "growth_last_pass_bytes": trail[-1] - trail[-2]}))


if __name__ == "__main__":
    main()
