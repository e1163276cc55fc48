#!/usr/bin/env python
"""Benchmark of the Larynx hot path on MI355X (contract: see the task brief).

One "step" = one utterance through the whole path — phoneme ids -> GlowTTS ->
mel transform -> HiFi-GAN 'high' -> float + int16 waveform — at batch 1
(BASELINE.json configs[1]; standard utterance S of SURVEY.md §8: P = 120 ids,
ljspeech hyper-parameters, about 620 frames = 7.2 s of audio).  Inputs (ids) and
outputs (waveforms) are device resident; the only host traffic inside a step is
the frame-count read-back the data-dependent length needs.

`--gpus N` (N > 1): one process per GPU.  Started by the driver under
`python -m torch.distributed.run` the ranks are already there; started by hand
without that environment, this script re-executes itself under
torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1).  Rank 0
builds the folded weight blobs and broadcasts them over RCCL/xGMI once;
utterances are independent, so the timed region has no collective ("weak"
scaling: every rank synthesises its own K utterances; max over ranks).  After
the headline region every run also times BASELINE config 3 — 256 utterances,
P ~ clip(N(120,15),60,200), LPT-sharded over the ranks, audio gathered to rank 0
in sentence order — and reports it as `config3` (strong scaling), and BASELINE
config 5 — 210 sentences cycling three resident voices, in-order delivery — as
`config5` (time to first audio, sustained x real time).

`--device cpu --library <emulator .so> --tiny` runs the same code path on the
CPU emulator build with gloo (tests/test_bench_path.py, world size 2).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

FP32_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, f32 MFMA = f32 vector peak
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide; AMD's 5 PF headline includes 2:1 sparsity)
SAMPLE_RATE = 22050
ROUND = "r06"


def algorithmic_flop(P: int, F: float, quality: str = "high") -> float:
    """FLOP(P,F,q) = 2*[7142656 P + 6(384 P^2 + 3456 P) + 10675968 F + H_q F]."""
    H = {"high": 307052544, "medium": 19255296, "low": 22482944}.get(quality, 0)
    return 2.0 * (7142656.0 * P + 6.0 * (384.0 * P * P + 3456.0 * P) + 10675968.0 * F + H * F)


def cpu_baseline(ids: np.ndarray, length_scale: float, max_seconds: float = 45.0):
    """The CPU oracle timed on this box's host cores on the SAME utterance the GPU leg times
    first (BASELINE config 2 at S: 120 ids -> ~600 frames), end to end on torch CPU operators —
    GlowTTS (oracle/glow_tts_torch.py) + mel transforms + HiFi-GAN 'high' (oracle/hifi_gan_torch.py):
    the conv / matmul / softmax kernels (oneDNN, MKL) the reference's own `--backend pytorch`
    path runs on, wrapped like `_sentence_task` — + int16 conversion.  Both
    ports are pinned to the numpy oracle, which is pinned to the reference's modules.  Thread
    count: best of a short sweep; then >= 5 timed runs at that count (min and median) and one
    single-thread run."""
    import torch

    from larynx_amd import hparams as HP
    from larynx_amd import synthetic
    from larynx_amd.audio import ljspeech_audio_settings
    from oracle import audio_np, glow_tts_torch, hifi_gan_torch

    gsd = synthetic.make_glow_state_dict(HP.LJSPEECH, seed=1234)
    vsd = synthetic.make_hifigan_state_dict(HP.HIFIGAN_HIGH, seed=1234)
    noise = np.random.default_rng(1234).standard_normal((80, 16 * len(ids) + 64)).astype(np.float32)
    s = ljspeech_audio_settings()
    ncpu = os.cpu_count() or 1

    def once(threads):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        mel = glow_tts_torch.glow_tts_infer_torch(gsd, HP.LJSPEECH, ids, noise, 0.667, length_scale, threads=threads)
        wav = hifi_gan_torch.hifigan_infer_torch(vsd, HP.HIFIGAN_HIGH, audio_np.mel_to_vocoder_input(mel, s), threads=threads)
        audio_np.audio_float_to_int16(wav)
        return time.perf_counter() - t0, mel.shape[1]

    t_all = time.perf_counter()
    once(min(ncpu, 16))  # warm-up (oneDNN primitive creation)
    sweep = {}
    for threads in sorted({min(ncpu, t) for t in (8, 16, 32, 64)}):
        if time.perf_counter() - t_all > max_seconds * 0.4:
            break
        sweep[threads], F = once(threads)
    best_threads = min(sweep, key=sweep.get)
    runs = [sweep[best_threads]]
    while len(runs) < 6 and time.perf_counter() - t_all < max_seconds * 0.8:
        dt, F = once(best_threads)
        runs.append(dt)
    one_thread = None
    if time.perf_counter() - t_all < max_seconds:
        one_thread, F = once(1)
    torch.set_num_threads(min(ncpu, 32))
    audio_s = F * 256 / SAMPLE_RATE
    best, med = min(runs), float(np.median(runs))
    return {
        "value": 1.0 / best,
        "unit": "utterances/s",
        "cores": best_threads,
        "host_cpus": ncpu,
        "kind": "port",
        "seconds_min": best,
        "seconds_median": med,
        "runs": len(runs),
        "rtf": best / audio_s,
        "rtf_median": med / audio_s,
        "rtf_1_thread": (one_thread / audio_s) if one_thread else None,
        "x_realtime": audio_s / best,
        "thread_sweep_seconds": {str(k): v for k, v in sweep.items()},
        "sample": f"CPU oracle on torch CPU operators end to end (GlowTTS + mel transforms + HiFi-GAN 'high' + int16: the arithmetic of the "
                  f"reference's --backend pytorch; reported at world size 1 only) on the benchmark's own first "
                  f"utterance: {len(ids)} ids -> {F} frames = {audio_s:.2f} s audio, whole utterance per run; {best_threads} threads "
                  f"(best of a {sorted(sweep)} sweep on {ncpu} host CPUs), min / median of {len(runs)} runs = {best:.3f} / {med:.3f} s",
    }


def by_kernel_table(prof_kernels: dict, glow_top: int = 5) -> dict:
    """`roofline.by_kernel`: launches and average RAW event microseconds per kernel name / sub-key (output rows of a conv launch,
    channels of a fused ResBlock step) of the profiled single-stream pass — every kernel of the ResBlock and upsampler classes,
    the vocoder's pre / post convs and the `glow_top` GlowTTS kernels with the largest total time.  A driver record on an unknown
    box can be compared with the builder's kernel by kernel (round 5's table: profiles/r05_by_kernel.json)."""
    out = {}

    def rows(cls):
        return {k: {"launches": int(v["launches"]), "avg_us": 1e3 * v["ms"] / max(1, v["launches"]), "total_ms": v["ms"]}
                for k, v in prof_kernels.get(cls, {}).items()}

    for cls in ("conv_mfma.hifigan_resblock", "conv_mfma.hifigan_upsample", "conv_mfma.hifigan_pre_post", "mrf_small.hifigan_narrow_stage"):
        r = rows(cls)
        if r:
            out[cls] = r
    glow = {}
    for cls in ("conv_mfma.glow_encoder", "conv_mfma.glow_decoder", "elementwise"):
        for k, v in rows(cls).items():
            glow[f"{cls}:{k}"] = v
    top = sorted(glow.items(), key=lambda kv: -kv[1]["total_ms"])[:glow_top]
    out["glow_top"] = dict(top)
    return out


def pin_rank_to_gpu_numa(local: int) -> dict:
    """Keep this rank's host threads (the ThreadPoolExecutor feeding its GPU, torch's intra-op pool) on the NUMA node its GPU
    hangs off: /sys/bus/pci/devices/<bdf>/numa_node -> /sys/devices/system/node/node<N>/cpulist -> sched_setaffinity.
    Best effort: a box without that information (or a single-node one) is left alone; what was done goes into the line."""
    try:
        import torch

        pr = torch.cuda.get_device_properties(local)
        bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(Path(f"/sys/bus/pci/devices/{bdf}/numa_node").read_text().strip())
        if node < 0:
            return {"pci": bdf, "numa_node": node, "pinned": False, "why": "the platform reports no NUMA affinity for the device"}
        cpus = set()
        for part in Path(f"/sys/devices/system/node/node{node}/cpulist").read_text().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return {"pci": bdf, "numa_node": node, "pinned": False, "why": "no allowed CPU on that node"}
        os.sched_setaffinity(0, cpus)
        return {"pci": bdf, "numa_node": node, "pinned": True, "cpus": len(cpus)}
    except Exception as e:  # noqa: BLE001 - placement is an optimisation, never a reason to fail the run
        return {"pinned": False, "why": f"{type(e).__name__}: {e}"[:160]}


def respawn_under_torchrun(n: int) -> int:
    """`python bench.py --gpus N` by hand: become N ranks (one per GPU) under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def config3_ids(num_symbols: int, n: int = 256, mean: float = 120.0, std: float = 15.0, lo: int = 60, hi: int = 200):
    """rng = default_rng(1234); P_i = clip(round(N(120, 15)), 60, 200)."""
    from larynx_amd import synthetic

    rng = np.random.default_rng(1234)
    lens = np.clip(np.rint(rng.normal(mean, std, n)), lo, hi).astype(int)
    return [synthetic.synthetic_phoneme_ids(rng, int(p), num_symbols) for p in lens]


CONFIG4_P = (19, 26, 31, 33, 64, 47, 90, 120)  # SURVEY.md §8(d) config 4 (first five = the thorsten fixture sentences)


def config4_rows(num_symbols: int):
    """The rows of the golden batch (tests/golden/batch8, every row pinned to the reference) when the fixture is in the
    tree, else seeded synthetic ids of the same lengths."""
    gp = REPO / "tests" / "golden" / "batch8" / "thorsten_medium_batch8.npz"
    if gp.is_file():
        z = np.load(gp)
        rows = [np.asarray(z[f"ids{b}"], np.int64) for b in range(8)]
        if tuple(len(r) for r in rows) == CONFIG4_P:
            return rows, float(z["length_scale"]), "tests/golden/batch8/thorsten_medium_batch8.npz"
    from larynx_amd import synthetic

    rng = np.random.default_rng(4)
    return [synthetic.synthetic_phoneme_ids(rng, n, num_symbols) for n in CONFIG4_P], 0.5, "seeded synthetic ids"


def micro_batch_leg(eng, g, v, ghp, vhp, args, dev, barrier, timed, world, red_dev, use_dist, audio, rank):
    """The headline's utterances as MICRO-BATCHES: 8 standard utterances per fused call (a padded batch; every row with its own
    noise stream), 2 calls in flight — what a serving front-end that batches gets (sharding.synthesize_shard(batch = 8)).  Same
    models, same arithmetic; rows equal their batch-1 results up to f32 summation order (a padded batch picks other tiles)."""
    import queue
    from concurrent.futures import ThreadPoolExecutor

    import torch
    import torch.distributed as dist

    from larynx_amd import ffi
    from larynx_amd import synthetic

    MB, threads = 8, 2
    K = 2 * max(4, args.steps // 4)  # calls per region: whole rounds of the two callers
    rng = np.random.default_rng(777 + rank)
    ids_host = np.stack([synthetic.synthetic_phoneme_ids(rng, args.ids, ghp.num_symbols) for _ in range(MB * (K + 1))])
    ids_dev = torch.from_numpy(ids_host).to(dev)
    lens = np.full(MB, args.ids, np.int32)
    hop = vhp.hop
    max_frames = args.ids * 12
    max_samples = max_frames * hop
    wav_f32 = [torch.empty(MB * max_samples, dtype=torch.float32, device=dev) for _ in range(threads)]
    wav_i16 = [torch.empty(MB * max_samples, dtype=torch.int16, device=dev) for _ in range(threads)]
    flags = ffi.IN_DEVICE | ffi.OUT_DEVICE
    eng.reserve(threads + 1, g, v, max_batch=MB, max_ids=max(args.ids, 200), max_frames=max_frames)
    frames_seen = []

    def call(i, slot=0):
        fr = eng.synthesize_raw(g, v, ids_dev[i * MB].data_ptr(), lens, args.ids, 0.667, args.length_scale, wav_f32[slot].data_ptr(),
                                wav_i16[slot].data_ptr(), max_samples, seed=9000 + MB * i, audio_settings=audio, flags=flags)
        return fr

    pool = ThreadPoolExecutor(threads)

    def run(n):
        q = queue.SimpleQueue()
        for i in range(n):
            q.put(i)

        def work(slot):
            while True:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return
                call(i, slot)

        list(pool.map(work, range(threads)))

    frames_seen = call(0)
    run(threads * 2)
    reps = max(5, min(20, int(np.ceil(1.0 / max(1e-4, min(timed(lambda: run(K), 1)))))))
    t = timed(lambda: run(K), reps)
    pool.shutdown()
    st = torch.tensor([float(np.median(t)), min(t), max(t)], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
    dt, dt_min, dt_max = (float(x) for x in st)
    return {
        "what": f"the headline's standard utterances as padded micro-batches: {MB} utterances per fused call, {threads} calls in flight "
                "(a front-end that batches: sharding.synthesize_shard(batch = 8)); same models and arithmetic, every row its own noise stream",
        "batch": MB,
        "calls_in_flight_per_gpu": threads,
        "steps": K,
        "repeats": reps,
        "utterances_per_sec": world * K * MB / dt,
        "ms_per_utterance": 1e3 * dt / (K * MB),
        "ms_per_utterance_min": 1e3 * dt_min / (K * MB),
        "ms_per_utterance_max": 1e3 * dt_max / (K * MB),
        "frames_per_utterance": float(np.mean(frames_seen)),
    }


def config4_leg(eng, args, dev, barrier, timed, pool, conc, world, red_dev, use_dist, audio):
    """BASELINE config 4 on this rank; returns the `config4` object (timings MAX-reduced over ranks)."""
    import torch
    import torch.distributed as dist

    from larynx_amd import ffi
    from larynx_amd import hparams as HP
    from larynx_amd import synthetic

    ghp, vhp = HP.THORSTEN, HP.HIFIGAN_MEDIUM
    g = eng.load_glow(ghp, synthetic.make_glow_state_dict(ghp, seed=1234))  # seeded: identical on every rank, = the golden's
    v = eng.load_hifigan(vhp, synthetic.make_hifigan_state_dict(vhp, seed=1234))
    rows, length_scale, src = config4_rows(ghp.num_symbols)
    B, ld = len(rows), max(len(r) for r in rows)
    lens = np.array([len(r) for r in rows], np.int32)
    packed = np.zeros((B, ld), np.int64)
    for b, r in enumerate(rows):
        packed[b, : len(r)] = r
    ids_dev = torch.from_numpy(packed).to(dev)
    hop = vhp.hop
    max_frames = ld * 12
    max_samples = max_frames * hop
    wav_f32 = [torch.empty(B * max_samples, dtype=torch.float32, device=dev) for _ in range(conc)]
    wav_i16 = [torch.empty(B * max_samples, dtype=torch.int16, device=dev) for _ in range(conc)]
    flags = ffi.IN_DEVICE | ffi.OUT_DEVICE
    eng.reserve(conc + 1, g, v, max_batch=B, max_ids=ld, max_frames=max_frames)
    K = max(4, args.steps // 2)

    def call(i, slot=0):
        fr = eng.synthesize_raw(g, v, ids_dev.data_ptr(), lens, ld, 0.667, length_scale, wav_f32[slot].data_ptr(),
                                wav_i16[slot].data_ptr(), max_samples, seed=4000 + 8 * i, audio_settings=audio, flags=flags)  # a batch of 8 consumes the streams seed .. seed + 7
        return fr

    def run(n, threads):
        if pool is None or threads <= 1:
            for i in range(n):
                call(i)
            return
        import queue

        q = queue.SimpleQueue()
        for i in range(n):
            q.put(i)

        def work(slot):
            while True:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return
                call(i, slot)

        list(pool.map(work, range(threads)))

    frames = call(0)
    run(max(2, conc), conc)  # every in-flight slot at this shape
    eng.set_profiling(True)
    call(0)
    eng.profile_reset()
    barrier()
    for i in range(K):
        call(i)
    barrier()
    prof = eng.profile()
    eng.set_profiling(False)
    call(0)
    reps = max(5, min(20, int(np.ceil(1.0 / max(1e-4, min(timed(lambda: run(K, 1), 1)))))))
    t_single = timed(lambda: run(K, 1), reps)
    t_flight = timed(lambda: run(K, conc), reps) if conc > 1 else t_single
    st = torch.tensor([float(np.median(t_flight)), float(np.median(t_single)), min(t_flight), max(t_flight)], dtype=torch.float64, device=red_dev)
    if use_dist:
        dist.all_reduce(st, op=dist.ReduceOp.MAX)
    dt_f, dt_s, dt_min, dt_max = (float(x) for x in st)
    eng.unload(g)
    eng.unload(v)
    F = int(frames.sum())
    audio_s = F * hop / SAMPLE_RATE
    flop = sum(algorithmic_flop(int(p), float(f), "medium") for p, f in zip(lens, frames))
    wide, narrow = prof["conv_mfma.hifigan_resblock"], prof["mrf_small.hifigan_narrow_stage"]

    def tf(c):
        return c["flop"] / (c["ms"] * 1e-3) / 1e12 if c["ms"] > 0 else None

    # algorithmic bytes of a narrow-stage launch: one read + one write of the stage's [C x L] plane per row
    C0 = vhp.upsample_initial_channel
    ups = np.cumprod(vhp.upsample_rates)
    narrow_bytes = sum(2 * 4 * (C0 >> (i + 1)) * int(ups[i]) * F for i in range(len(ups)) if (C0 >> (i + 1)) <= 16)
    return {
        "workload": f"de-de thorsten GlowTTS (V=54) + hifi_gan 'medium', ONE padded batch of 8 rows per call, P = {list(map(int, lens))} "
                    f"-> frames {list(map(int, frames))} ({F} frames = {audio_s:.2f} s audio per call; ids: {src}), length_scale "
                    f"{length_scale}, fused entry (mi355tts_synthesize), ids and waveforms device resident, {conc} calls in flight",
        "batch": B,
        "frames_per_call": F,
        "steps": K,
        "repeats": reps,
        "ms_per_call": 1e3 * dt_f / K,
        "ms_per_call_min": 1e3 * dt_min / K,
        "ms_per_call_max": 1e3 * dt_max / K,
        "utterances_per_sec": world * K * B / dt_f,
        "x_realtime_per_gpu": audio_s * K / dt_f,
        "latency_ms_single_stream": 1e3 * dt_s / K,
        "end_to_end_tflops_per_gpu": flop * K / dt_f / 1e12,
        "algorithmic_gflop_per_call": flop / 1e9,
        "scaling": "weak",
        "roofline": {
            "narrow_stages": {
                "kernel": "mrf_small_kernel (16 channels: v_mfma_f32_16x16x4_f32) + mrf8_kernel (8 channels: v_mfma_f32_4x4x1_16B_f32, no "
                          "padding rows): a whole stage — all three ResBlock1 chains on LDS-resident tiles — per launch",
                "bound": "mfma",
                "why": "fused, a stage is 252 (C = 8) / 504 (C = 16) FLOP per byte of its one read + one write — far above the ~20 FLOP/B ridge",
                "achieved": tf(narrow), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (tf(narrow) or 0.0) / FP32_PEAK_TFLOPS,
                "launches": narrow["launches"], "avg_launch_us": 1e3 * narrow["ms"] / max(1, narrow["launches"]),
                "ms_per_call": narrow["ms"] / K,
                "algorithmic_bytes_per_call": narrow_bytes,
                "algorithmic_gbytes_per_s": narrow_bytes * K / (narrow["ms"] * 1e-3) / 1e9 if narrow["ms"] > 0 else None,
            },
            "wide_stages": {
                "kernel": "pair_group_kernel / rb_pair_group_kernel (fused conv pairs, 64- and 32-channel stages: the 8-wave k-split tile below 512 tiles per member, the four-wave tile above)",
                "bound": "mfma", "achieved": tf(wide), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": (tf(wide) or 0.0) / FP32_PEAK_TFLOPS,
                "launches": wide["launches"], "avg_launch_us": 1e3 * wide["ms"] / max(1, wide["launches"]),
                "ms_per_call": wide["ms"] / K,
            },
            "timing": "HIP events on the launch stream around every launch, profiled single-stream pass of the same K calls",
        },
        "profile_ms_per_call": {k_: v_["ms"] / K for k_, v_ in prof.items()},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--ids", type=int, default=120, help="phoneme ids per utterance")
    ap.add_argument("--quality", default="high")
    ap.add_argument("--length-scale", type=float, default=0.65,
                    help="GlowTTS length_scale; 0.65 puts the synthetic 120-id utterances at SURVEY's standard ~620 frames")
    ap.add_argument("--concurrency", type=int, default=8,
                    help="utterances in flight per GPU in the timed region (host threads, each batch-1 call on its own "
                         "HIP stream — the reference's ThreadPoolExecutor pattern); the single-stream latency is "
                         "measured with the same method and reported next to it.  A multiple of the runtime's 4 hardware "
                         "queues: 5-7 streams load the queues unevenly (measured 4 / 8: 249 / 248 utterances/s, 5 / 6 / 7: "
                         "242-244; split-bf16 mode 529 / 525 vs 494-502)")
    ap.add_argument("--batch", type=int, default=1,
                    help="utterances per call (rows of one padded batch). Default 1 = BASELINE.json's quoted configuration")
    ap.add_argument("--repeats", type=int, default=0,
                    help="how many times the K-step region is timed (each bracketed by barrier + synchronize); the line "
                         "reports the median.  0 = as many as fit ~2 s, at least 5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady-state", action="store_true", help="skip the 10 x K-step regions reported as `steady_state`")
    ap.add_argument("--voc-only-sleep-ms", type=float, default=0.0,
                    help="probe: in the vocoder-only region every call first sleeps this long on the host (stands in for the latency "
                         "of an acoustic pass that uses no GPU at all)")
    ap.add_argument("--no-config3", action="store_true")
    ap.add_argument("--no-half-mode", action="store_true", help="skip the secondary bf16x3 (`half` switch) leg")
    ap.add_argument("--config3-utterances", type=int, default=256)
    ap.add_argument("--no-config4", action="store_true", help="skip BASELINE config 4 (thorsten + 'medium', one padded batch of 8)")
    ap.add_argument("--no-config5", action="store_true")
    ap.add_argument("--no-micro-batch", action="store_true", help="skip the micro-batched leg (8 utterances per call, 2 calls in flight)")
    ap.add_argument("--call-coalesce-lanes", type=int, default=2,
                    help="lanes of the call_coalesce A/B leg when the library's default is 0 (option call_coalesce; csrc/host_join.h)")
    ap.add_argument("--config5-sentences", type=int, default=210)
    ap.add_argument("--config5-threads", type=int, default=3,
                    help="host threads per GPU in the streaming config (the reference's raw-stream default is 2: fewer workers "
                         "for latency to first audio, more for throughput)")
    ap.add_argument("--serial-branches", action="store_true",
                    help="run the headline pass with the MRF chains on one stream too (for rocprofv3 kernel traces)")
    ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"], help="cpu = the emulator build (tests only)")
    ap.add_argument("--set-option", action="append", default=[], metavar="NAME=VALUE", help="mi355tts_set_option before anything runs (A/B of schedules)")
    ap.add_argument("--library", default=os.environ.get("MI355TTS_LIB"), help="alternative libmi355tts build (A/B runs, emulator)")
    ap.add_argument("--tiny", action="store_true", help="shrunk hyper-parameters (emulator runs)")
    ap.add_argument("--tiny-half", action="store_true", help="with --tiny: run the half-mode (fp16) and split-bf16 legs too (they are skipped on the emulator by default)")
    ap.add_argument("--acoustic-f16", action="store_true",
                    help="probe: with --precision f32, the acoustic model's decoder WaveNets in fp16 (csrc/wn_f16.h) in front of the exact f32 "
                         "vocoder; implies --no-half-mode.  NOT the headline's configuration: the line's dtype says so")
    ap.add_argument("--half-acoustic", default="f16", choices=["f16", "f32"],
                    help="the acoustic model in the fp16 mode (half_mode leg, --precision f16): f16 = the decoder's WaveNets in fp16 "
                         "(csrc/wn_f16.h), f32 = only the vocoder takes the switch")
    ap.add_argument("--precision", default="f32", choices=["f32", "f16", "bf16x3"],
                    help="f32 = exact f32 MFMA everywhere (the graded parity mode); f16 = the `half` switch: the native fp16 vocoder "
                         "(fp16 planes, one fp16 MFMA per product); bf16x3 = split-bf16 ResBlock convs (3 x bf16 MFMA per product, f32 planes)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args.gpus))

    import torch
    import torch.distributed as dist

    from larynx_amd import ffi, sharding
    from larynx_amd import hparams as HP
    from larynx_amd import synthetic
    from larynx_amd.audio import ljspeech_audio_settings
    from larynx_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    on_gpu = args.device == "cuda"
    if world != max(1, args.gpus):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU")
    affinity = None
    if on_gpu:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        affinity = pin_rank_to_gpu_numa(local)
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0)) // max(1, world)))))
    else:
        dev = torch.device("cpu")
    # under torch.distributed.run (the driver's launch form) the process group is always brought up — also at
    # world size 1, so that the RCCL code path (init, broadcast, all_reduce, barrier, gather_object) is the one
    # that runs whenever the script is launched that way
    use_dist = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints a version banner on fd 1 when its first communicator comes up; stdout carries the ONE JSON
        # line of the contract, so fd 1 points at stderr until the group has done its first collective
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        pg_note = None
        try:
            if on_gpu:
                try:
                    if os.environ.get("BENCH_SIMULATE_RCCL_FAILURE"):  # exercises the fallback below on a healthy node
                        raise RuntimeError("simulated (BENCH_SIMULATE_RCCL_FAILURE)")
                    dist.init_process_group("nccl", device_id=dev)  # "nccl" IS RCCL on ROCm
                    dist.barrier()
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001 - the data path has no collective: a host-side group is enough to time it
                    # RCCL could not come up on this node: keep the job alive on gloo (barriers and the max-over-ranks
                    # reductions go over the host), every rank folds its own copy of the seeded weights
                    pg_note = f"{type(e).__name__}: {str(e).splitlines()[0][:160] if str(e) else ''}"
                    print(f"[bench] RCCL process group failed ({pg_note}); falling back to gloo", file=sys.stderr)
                    if dist.is_initialized():
                        dist.destroy_process_group()
                    dist.init_process_group("gloo")
                    dist.barrier()
            else:
                dist.init_process_group("gloo")
                dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    rccl = use_dist and on_gpu and dist.get_backend() == "nccl"
    red_dev = dev if (rccl or not use_dist) else torch.device("cpu")  # where the timing reductions live

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        sync()

    _dummy = []
    if on_gpu and os.environ.get("BENCH_DUMMY_STREAMS"):  # probe: streams created (and used once) before the library's worker streams
        for _ in range(int(os.environ["BENCH_DUMMY_STREAMS"])):
            st = torch.cuda.Stream(device=local)
            with torch.cuda.stream(st):
                torch.zeros(64, device=f"cuda:{local}").add_(1.0)
            _dummy.append(st)
        torch.cuda.synchronize()
    eng = Engine(device=local if on_gpu else 0, library_path=args.library)
    for kv in args.set_option:
        eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    if args.tiny:
        ghp, vhp, quality = HP.TINY_GLOW, HP.TINY_HIFIGAN, "tiny"
    else:
        ghp, vhp, quality = HP.LJSPEECH, HP.VOCODER_QUALITY[args.quality], args.quality
    # ---- weights: rank 0 folds, everyone receives over RCCL/xGMI
    man_g = ffi.manifest(eng.lib, ffi.glow_hparams_c(ghp))
    man_v = ffi.manifest(eng.lib, ffi.hifigan_hparams_c(vhp))
    n_g, n_v = sum(n for _, n in man_g), sum(n for _, n in man_v)
    blob = torch.empty(n_g + n_v, dtype=torch.float32, device=dev)
    if rank == 0 or (use_dist and on_gpu and not rccl):
        from larynx_amd.weights import build_blob

        bg = build_blob(man_g, synthetic.make_glow_state_dict(ghp, seed=1234))
        bv = build_blob(man_v, synthetic.make_hifigan_state_dict(vhp, seed=1234))
        blob.copy_(torch.from_numpy(np.concatenate([bg, bv])))
    t_b = time.perf_counter()
    if use_dist and (rccl or not on_gpu):
        dist.broadcast(blob, src=0)
    sync()
    broadcast_s = time.perf_counter() - t_b
    g = eng.load_glow(ghp, device_ptr=blob.data_ptr())
    v = eng.load_hifigan(vhp, device_ptr=blob.data_ptr() + 4 * n_g)
    del blob
    glow_f16_line = False  # --precision f16: the acoustic model's decoder WaveNets in fp16 too
    if args.precision == "f32" and args.acoustic_f16:  # probe: exact f32 vocoder behind the fp16 acoustic mode (NOT the headline's configuration)
        glow_f16_line = eng.set_precision(g, ffi.PRECISION_F16) == 0
    if args.precision == "bf16x3":
        eng.set_precision(v, ffi.PRECISION_BF16X3)
    elif args.precision == "f16":
        eng.set_precision(v, ffi.PRECISION_F16)
        if args.half_acoustic == "f16":
            glow_f16_line = eng.set_precision(g, ffi.PRECISION_F16) == 0

    # ---- synthetic utterances (one per step per rank), resident in HBM
    rng = np.random.default_rng(1234 + rank)
    K, W = args.steps, args.warmup
    n_utts = K + W  # steps; each step is one call over `batch` utterances
    n_rows = max(n_utts, max(1, args.concurrency))  # the warm-up runs one utterance on every in-flight slot
    B = max(1, args.batch)
    ids_host = np.stack([synthetic.synthetic_phoneme_ids(rng, args.ids, ghp.num_symbols) for _ in range(n_rows * B)])
    ids_dev = torch.from_numpy(ids_host).to(dev)
    lens = np.full(B, args.ids, np.int32)
    hop = vhp.hop
    max_frames = args.ids * 12
    max_samples = max_frames * hop
    conc = max(1, args.concurrency)
    wav_f32 = [torch.empty(B * max_samples, dtype=torch.float32, device=dev) for _ in range(conc)]
    wav_i16 = [torch.empty(B * max_samples, dtype=torch.int16, device=dev) for _ in range(conc)]
    s = ljspeech_audio_settings()
    io_flags = ffi.IN_DEVICE | ffi.OUT_DEVICE

    # every worker a call can land on exists and is sized before anything is timed: no
    # hipMalloc / hipFree / stream creation can happen inside a timed region
    dn_ok = 88 * hop > 1024  # the denoiser's 1024-point STFT needs a real vocoder hop (not the emulator's tiny one)
    eng.reserve(conc + 1, g, v, max_batch=B, max_ids=max(args.ids, 200), max_frames=max(max_frames, 2400), denoiser=dn_ok)
    qgroups_headline = eng.worker_queue_groups()  # as the headline's calls find them (later legs reserve other worker counts)
    if B == 1 and conc > 1:
        # concurrent batch-1 calls may ride fused padded calls (csrc/host_join.h): any worker may lead a pass of up to `conc` rows
        eng.reserve(conc + 1, g, v, max_batch=conc, max_ids=max(args.ids, 200), max_frames=max_frames)

    def step(i, slot=0, denoiser=0.0):
        """One utterance (one batch of B) through the fused call; returns its frame count."""
        fr = eng.synthesize_raw(g, v, ids_dev[i * B].data_ptr(), lens, args.ids, 0.667, args.length_scale,
                                wav_f32[slot].data_ptr(), wav_i16[slot].data_ptr(), max_samples, seed=1234 + i,
                                audio_settings=s, denoiser_strength=denoiser, flags=io_flags)
        return int(fr.sum())

    from concurrent.futures import ThreadPoolExecutor

    pool = ThreadPoolExecutor(conc) if conc > 1 else None

    def run_steps(lo, hi, denoiser=0.0, threads=conc):
        """Steps lo..hi-1; `threads` host threads pull utterances (the reference's own
        ThreadPoolExecutor pattern, larynx/__init__.py:146), each call on its own stream."""
        if pool is None or threads <= 1:
            return sum(step(i, 0, denoiser) for i in range(lo, hi))
        import queue

        q = queue.SimpleQueue()
        for i in range(lo, hi):
            q.put(i)

        def work(slot):
            tot = 0
            while True:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return tot
                tot += step(i, slot, denoiser)

        return sum(pool.map(work, range(threads)))

    def timed(fn, repeats):
        """`repeats` timings of fn(), each bracketed by barrier + synchronize; seconds list."""
        out = []
        for _ in range(repeats):
            barrier()
            t0 = time.perf_counter()
            fn()
            barrier()
            out.append(time.perf_counter() - t0)
        return out

    # ---- warm-up: W steps single-stream, then every in-flight slot on the longest shape
    for i in range(W):
        step(i)
    if conc > 1:
        run_steps(0, max(W, conc))
        if dn_ok:
            run_steps(0, max(W, conc), denoiser=0.005)
    if dn_ok:
        step(0, 0, 0.005)
    barrier()

    # ---- profiled pass: HIP events around every launch.  The product schedule keeps a call on ONE
    # stream (the three MRF chains' same-geometry convs go out as one grouped launch), so every
    # launch is timed alone on the GPU and its duration is what the roofline uses.
    # --serial-branches times the un-grouped, one-conv-per-launch form instead.
    eng.set_profiling(True)
    eng.set_option("serial_branches", 1 if args.serial_branches else 0)
    step(W)  # that schedule's workspace shape, outside the measured pass
    eng.profile_reset()
    barrier()
    t0 = time.perf_counter()
    frames = sum(step(i) for i in range(W, n_utts))
    barrier()
    dt_prof = time.perf_counter() - t0
    prof = eng.profile()
    prof_kernels = eng.profile_kernels()
    eng.set_profiling(False)
    eng.set_option("serial_branches", 1 if args.serial_branches else 0)
    step(W)

    # ---- the timed regions (event-free): exactly K steps each, repeated; medians reported
    est = max(1e-4, min(timed(lambda: run_steps(W, n_utts, threads=1), 1)))
    repeats = args.repeats if args.repeats > 0 else int(min(40, max(5, np.ceil(2.0 / est))))
    t_single = timed(lambda: run_steps(W, n_utts, threads=1), repeats)
    t_flight = timed(lambda: run_steps(W, n_utts), repeats) if conc > 1 else t_single
    # the same region with the other setting of option "call_coalesce" (csrc/host_join.h: concurrent batch-1 callers become the
    # rows of fused padded calls), reported next to the headline — which runs the library's DEFAULT
    t_flight_nc = t_flight
    t_steady_nc = None
    cs0 = cs1 = (0, 0)
    cc_default = eng.get_call_coalesce_default()
    cc_other = 0 if cc_default else args.call_coalesce_lanes
    if conc > 1 and B == 1:
        eng.set_option("call_coalesce", cc_other)
        run_steps(0, max(W, conc))
        cs0 = eng.coalesce_stats()
        t_flight_nc = timed(lambda: run_steps(W, n_utts), max(3, repeats // 3))
        cs1 = eng.coalesce_stats()
        if not args.tiny and not args.no_steady_state:  # and one steady-state region (10 x K steps) with that setting
            import queue as _q

            def run_long_cc():
                q = _q.SimpleQueue()
                for j in range(10 * K):
                    q.put(W + (j % K))

                def work(slot):
                    while True:
                        try:
                            i = q.get_nowait()
                        except _q.Empty:
                            return
                        step(i, slot)

                list(pool.map(work, range(conc)))

            t_steady_nc = timed(run_long_cc, 2)
        eng.set_option("call_coalesce", cc_default)
        run_steps(0, max(W, conc))
    t_dn = timed(lambda: run_steps(W, n_utts, denoiser=0.005), max(3, repeats // 3)) if dn_ok else [float("nan")]
    # ---- steady state: the SAME calls in regions ten times as long.  A K-step region starts on an idle GPU (every caller begins
    # with its acoustic pass: ~2 ms before the first vocoder launch) and drains at its end (the last calls run with fewer and
    # fewer others to fill their launches' tails); at K = 20 that is ~4 % of a 74 ms region.  Reported next to the headline,
    # which keeps the K-step definition.
    t_steady, K_steady = None, 0
    if conc > 1 and not args.tiny and not args.no_steady_state:
        K_steady = 10 * K

        def run_long():
            if pool is None:
                return sum(step(W + (j % K)) for j in range(K_steady))
            import itertools
            import queue

            q = queue.SimpleQueue()
            for j in range(K_steady):
                q.put(W + (j % K))

            def work(slot):
                while True:
                    try:
                        i = q.get_nowait()
                    except queue.Empty:
                        return
                    step(i, slot)

            list(pool.map(work, range(conc)))

        t_steady = timed(run_long, 3)
    # host CPU seconds (all threads of this process) over one more pass of the headline region: what a rank costs the host
    # per utterance — the one term of the 1 -> 8 GPU curve that the ranks share (8 ranks x `conc` threads on one host)
    barrier()
    c0, w0 = time.process_time(), time.perf_counter()
    run_steps(W, n_utts)
    barrier()
    host_cpu_s, host_wall_s = time.process_time() - c0, time.perf_counter() - w0
    # what GlowTTS costs UNDER LOAD: the same K steps with the acoustic model taken out — every call is the vocoder alone on
    # the device-resident mel GlowTTS produced for that utterance (same in-flight count, same buffers); the difference to the
    # headline region is GlowTTS's price per utterance when its launches compete with other calls' vocoder launches
    t_voc = t_voc_steady = None
    if B == 1 and not args.tiny:
        mels = {i: eng.glow_infer_raw(g, ids_dev[i].data_ptr(), lens, args.ids, 0.667, args.length_scale, seed=1234 + i, audio_settings=s,
                                      flags=ffi.IN_DEVICE) for i in range(W, n_utts)}
        if mels:
            def voc_step(i, slot):
                if args.voc_only_sleep_ms > 0:
                    time.sleep(args.voc_only_sleep_ms * 1e-3)
                eng.hifigan_infer_raw(v, mels[i], wav_f32[slot].data_ptr(), wav_i16[slot].data_ptr(), max_samples, flags=ffi.OUT_DEVICE)

            def run_voc():
                if pool is None:
                    for i in range(W, n_utts):
                        voc_step(i, 0)
                    return
                import queue

                q = queue.SimpleQueue()
                for i in range(W, n_utts):
                    q.put(i)

                def work(slot):
                    while True:
                        try:
                            i = q.get_nowait()
                        except queue.Empty:
                            return
                        voc_step(i, slot)

                list(pool.map(work, range(conc)))

            run_voc()
            t_voc = timed(run_voc, max(3, repeats // 3))
            if t_steady:  # the vocoder-only region at the steady-state length too
                def run_voc_long():
                    import queue

                    q = queue.SimpleQueue()
                    for j in range(K_steady):
                        q.put(W + (j % K))

                    def work(slot):
                        while True:
                            try:
                                i = q.get_nowait()
                            except queue.Empty:
                                return
                            voc_step(i, slot)

                    list(pool.map(work, range(conc)))

                t_voc_steady = timed(run_voc_long, 3)
            for m_ in mels.values():
                m_.free()

    def med(x):
        return float(np.median(x))

    # ---- the reference's `half` switch on this backend: the native fp16 vocoder (secondary figure; the headline above is the
    # exact f32 mode).  Same steps, same method, fewer repeats.  Next to it the split-bf16 mode (f32-class accuracy), in-flight only.
    half = None
    x3_flight = 0.0
    glow_f16 = False
    if args.precision == "f32" and (not args.tiny or args.tiny_half) and not args.no_half_mode and not args.acoustic_f16:
        eng.set_precision(v, ffi.PRECISION_F16)
        # `half` on the acoustic model: the decoder's WaveNets in fp16 (0), or a reported no-op on a geometry wn_f16.h does not cover
        glow_f16 = args.half_acoustic == "f16" and eng.set_precision(g, ffi.PRECISION_F16) == 0
        run_steps(0, max(W, conc))
        step(W)
        eng.set_profiling(True)
        eng.profile_reset()
        barrier()
        for i in range(W, n_utts):
            step(i)
        barrier()
        hprof = eng.profile()
        hprof_kn = eng.profile_kernels()
        eng.set_profiling(False)
        h_single = timed(lambda: run_steps(W, n_utts, threads=1), max(3, repeats // 3))
        h_flight = timed(lambda: run_steps(W, n_utts), max(3, repeats // 3)) if conc > 1 else h_single
        eng.set_precision(g, ffi.PRECISION_F32)
        eng.set_precision(v, ffi.PRECISION_BF16X3)
        run_steps(0, max(W, conc))
        x3_flight = med(timed(lambda: run_steps(W, n_utts), max(3, repeats // 3)))
        eng.set_precision(v, ffi.PRECISION_F32)
        step(W)
        half = (med(h_flight), med(h_single), hprof["conv_mfma.hifigan_resblock"], hprof, hprof_kn)

    stats = torch.tensor([med(t_flight), med(t_single), float(frames), min(t_flight), max(t_flight), min(t_single), med(t_dn), dt_prof,
                          half[0] if half else 0.0, half[1] if half else 0.0, med(t_flight_nc), med(t_voc) if t_voc else 0.0,
                          host_cpu_s, host_wall_s, med(t_steady) if t_steady else 0.0, med(t_voc_steady) if t_voc_steady else 0.0, x3_flight],
                         dtype=torch.float64, device=red_dev)
    per_rank = [[float(stats[0]), float(stats[1])]]  # this rank's (in-flight, single-stream) seconds per K-step region
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        total_frames = float(sm[2])
        stats = mx
    else:
        total_frames = float(frames)
    (dt_flight, dt_single, _, dt_flight_min, dt_flight_max, dt_single_min, dt_dn, dt_prof, dt_half_flight, dt_half_single, dt_flight_nc, dt_voc,
     host_cpu_max, host_wall_max, dt_steady, dt_voc_steady, dt_x3_flight) = (float(x) for x in stats)

    # ---- BASELINE config 3: 256 utterances, LPT-sharded over the ranks, ordered gather (strong scaling)
    c3 = None
    if not args.no_config3:
        rows = config3_ids(ghp.num_symbols, n=args.config3_utterances) if not args.tiny else \
            config3_ids(ghp.num_symbols, n=args.config3_utterances, mean=12, std=3, lo=5, hi=20)
        lengths = [len(r) for r in rows]
        mine = sharding.lpt_assign(lengths, world)[rank]
        cap = (max(lengths) * 12) * hop

        def shard_job():
            out = {}
            import queue

            q = queue.SimpleQueue()
            for i in mine:
                q.put(i)

            def work(_):
                while True:
                    try:
                        i = q.get_nowait()
                    except queue.Empty:
                        return
                    fr, _, i16 = eng.synthesize(g, v, rows[i], 0.667, args.length_scale, seed=1234 + i, audio_settings=s,
                                                frames_per_id_guess=12.0 / max(args.length_scale, 0.05))
                    out[i] = i16[0, : int(fr[0]) * hop].copy()

            if pool is None:
                work(0)
            else:
                list(pool.map(work, range(conc)))
            return out

        eng.reserve(conc + 1, g, v, max_batch=1, max_ids=max(lengths), max_frames=max(lengths) * 12)
        if conc > 1:
            eng.reserve(conc + 1, g, 0, max_batch=conc, max_ids=max(lengths), max_frames=max(lengths) * 12)
        shard_job()  # warm (host staging buffers at the largest shape)
        barrier()
        t0 = time.perf_counter()
        local_out = shard_job()
        barrier()
        t_job = time.perf_counter() - t0
        t0 = time.perf_counter()
        merged = sharding.gather_in_order(local_out, len(rows)) if use_dist else [local_out[i] for i in range(len(rows))]
        t_gather = time.perf_counter() - t0
        tj = torch.tensor([t_job], dtype=torch.float64, device=red_dev)
        if use_dist:
            dist.all_reduce(tj, op=dist.ReduceOp.MAX)
        if rank == 0:
            assert len(merged) == len(rows) and all(m.dtype == np.int16 and m.size > 0 for m in merged)
            c3 = {
                "workload": f"{len(rows)} synthetic utterances, P ~ clip(N(120,15),60,200) (mean {np.mean(lengths):.1f} ids), "
                            f"LPT-sharded by id count over {world} rank(s), batch-1 calls, {conc} in flight per GPU, host int16 "
                            f"output, audio gathered to rank 0 in sentence order after the timed region",
                "utterances": len(rows),
                "seconds": float(tj[0]),
                "utterances_per_sec": len(rows) / float(tj[0]),
                "audio_seconds": float(sum(m.size for m in merged)) / SAMPLE_RATE,
                "x_realtime": float(sum(m.size for m in merged)) / SAMPLE_RATE / float(tj[0]),
                "scaling": "strong",
                "ordered_gather_seconds": t_gather,
                "shard_sizes": [len(x) for x in sharding.lpt_assign(lengths, world)],
                "shard_ids": [int(sum(lengths[i] for i in x)) for x in sharding.lpt_assign(lengths, world)],
                "shard_imbalance": float(max(sum(lengths[i] for i in x) for x in sharding.lpt_assign(lengths, world)) * world / max(1, sum(lengths))),
            }

    # ---- BASELINE config 5: long multi-voice text as a stream — sentences cycling three resident voices (en V=46 /
    # de V=54 / fr V=42), dealt round-robin to the ranks, each rank keeps `--config5-threads` batch-1 calls in flight
    # and delivers its sentences in order (larynx/__main__.py:229-268 raw-stream semantics)
    c5 = None
    if not args.no_config5 and not args.tiny and on_gpu:
        voices = [(ghp, g)]
        for vhp2, seed2 in ((HP.THORSTEN, 2001), (HP.SIWIS, 2002)):
            voices.append((vhp2, eng.load_glow(vhp2, synthetic.make_glow_state_dict(vhp2, seed=seed2))))  # seeded: identical on every rank
        rng5 = np.random.default_rng(5)
        sents = []
        for i in range(args.config5_sentences):
            hp_i, g_i = voices[i % 3]
            sents.append((g_i, synthetic.synthetic_phoneme_ids(rng5, int(rng5.integers(40, 160)), hp_i.num_symbols)))
        mine5 = list(range(rank, len(sents), world))
        eng.reserve(args.config5_threads + 1, g, v, max_batch=1, max_ids=160, max_frames=160 * 12)

        def sentence(i):
            g_i, ids_i = sents[i]
            fr, _, i16 = eng.synthesize(g_i, v, ids_i, 0.667, args.length_scale, seed=i, audio_settings=s,
                                        frames_per_id_guess=12.0 / max(args.length_scale, 0.05))
            return i16[0, : int(fr[0]) * hop]

        with ThreadPoolExecutor(args.config5_threads) as pool5:
            list(pool5.map(sentence, mine5[: 2 * args.config5_threads]))  # warm every worker / voice
            barrier()
            t0 = time.perf_counter()
            futs = [pool5.submit(sentence, i) for i in mine5]
            first, total5 = None, 0
            for f in futs:  # in-order delivery
                a5 = f.result()
                if first is None:
                    first = time.perf_counter() - t0
                total5 += a5.shape[0]
            barrier()
            t5 = time.perf_counter() - t0
        st5 = torch.tensor([t5, first, float(total5)], dtype=torch.float64, device=red_dev)
        if use_dist:
            mx5 = st5.clone()
            dist.all_reduce(mx5, op=dist.ReduceOp.MAX)
            sm5 = st5.clone()
            dist.all_reduce(sm5, op=dist.ReduceOp.SUM)
            st5 = torch.stack([mx5[0], mx5[1], sm5[2]])
        if rank == 0:
            c5 = {
                "workload": f"{len(sents)} sentences of 40-160 ids cycling three resident GlowTTS voices (en-us V=46, de-de V=54, fr-fr V=42) "
                            f"against one 'high' vocoder, dealt round-robin to {world} rank(s), {args.config5_threads} batch-1 calls in "
                            f"flight per GPU, int16 to the host, delivered in sentence order per rank",
                "sentences": len(sents),
                "threads_per_gpu": args.config5_threads,
                "seconds": float(st5[0]),
                "ms_to_first_audio": 1e3 * float(st5[1]),
                "sentences_per_sec": len(sents) / float(st5[0]),
                "x_realtime": float(st5[2]) / SAMPLE_RATE / float(st5[0]),
            }
        for _, g_i in voices[1:]:
            eng.unload(g_i)

    # ---- BASELINE config 4: de-de thorsten GlowTTS + hifi_gan 'medium', ONE padded batch of 8 variable-length rows
    # (P = 19 ... 120: the golden batch of tests/golden/batch8, whose every row the parity suite checks against the
    # reference) per call, fused entry, device-resident ids and waveforms; every rank runs the same batch (weak scaling)
    c4 = None
    if not args.no_config4 and not args.tiny and on_gpu:
        c4 = config4_leg(eng, args, dev, barrier, timed, pool, conc, world, red_dev, use_dist, s)

    # ---- the same utterances as micro-batches of 8 (2 calls in flight): what batching buys on top of the batch-1 headline
    mb = None
    if B == 1 and on_gpu and not args.tiny and not args.no_micro_batch and args.precision == "f32":
        mb = micro_batch_leg(eng, g, v, ghp, vhp, args, dev, barrier, timed, world, red_dev, use_dist, s, rank)

    if rank == 0:
        audio_s = total_frames * hop / SAMPLE_RATE  # audio produced by all ranks in one K-step region
        utt_s = world * K * B / dt_flight
        fpu = total_frames / (world * K * B)
        traffic = None
        tpath = REPO / "profiles" / f"{ROUND}_roofline_traffic.json"
        if not tpath.is_file():
            tpath = REPO / "profiles" / "r01_roofline_traffic.json"
        if tpath.is_file():  # PMC counters cannot be read from inside this process: committed rocprofv3 passes
            tj = json.loads(tpath.read_text())
            traffic = tj.get("hbm_bytes_per_launch", tj.get("hbm_bytes_per_launch_raw"))
        dom = prof["conv_mfma.hifigan_resblock"]
        # The roofline uses the RAW event time of every launch (HIP events on the launch stream): on the boxes where both exist it
        # agrees with rocprofv3's kernel durations of the same command (profiles/r05_events_vs_rocprof.txt), which is what the
        # committed profiles/ summary holds.  The same figures minus the cost of an EMPTY event pair (measured here on an idle
        # stream: ~4.5 us) are kept under `*_minus_event_overhead`: that subtraction over-corrects (part of the records' cost
        # overlaps the kernel) and is NOT the headline.
        ev_us = eng.profile_event_overhead_us() if on_gpu else 0.0
        dom_ms_raw = dom["ms"]
        dom_ms = max(dom_ms_raw - 1e-3 * ev_us * dom["launches"], 0.5 * dom_ms_raw)
        dom_tf = dom["flop"] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        dom_tf_raw = dom["flop"] / (dom_ms_raw * 1e-3) / 1e12 if dom_ms_raw > 0 else 0.0
        all_conv_ms = sum(v_["ms"] for k_, v_ in prof.items() if k_.startswith("conv_mfma"))
        flop_utt = algorithmic_flop(args.ids, fpu, quality)
        out = {
            "metric": "utterances_per_sec",
            "value": utt_s,
            "unit": "utterances/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": 1e3 * dt_flight / K,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f32": "f32" if not glow_f16_line else "MIXED (probe, --acoustic-f16): f32 vocoder, fp16 WaveNets in GlowTTS' decoder (wn_f16.h), f32 elsewhere", "f16": "f16 (fp16 planes and weights, f32 accumulate) in the whole vocoder"
                                            + (" and in the WaveNets of GlowTTS' decoder (wn_f16.h), f32 in the rest of GlowTTS" if glow_f16_line else ", f32 in GlowTTS"),
                      "bf16x3": "bf16x3 (split-bf16 MFMA, f32 accumulate) in the ResBlock convs and upsamplers, f32 elsewhere"}[args.precision],
            "data": "synthetic",
            "config": {
                "workload": f"en-us ljspeech GlowTTS + hifi_gan '{quality}', batch={B}, {args.ids} phoneme ids per utterance "
                            f"(~{fpu:.0f} frames = {fpu * hop / SAMPLE_RATE:.2f} s audio), seeded random weights, device RNG noise, "
                            f"fused one-call entry (mi355tts_synthesize), denoiser off (the Python API default)",
                "ids_per_utterance": args.ids,
                "length_scale": args.length_scale,
                "frames_per_utterance": fpu,
                "parallelism": f"utterance-dp{world}",
                "batch": B,
                "calls_in_flight_per_gpu": conc,
            },
            "timing": {
                "method": f"K={K} steps bracketed by barrier + synchronize, repeated {repeats}x after {W} warm-up steps and a full "
                          f"warm-up of every in-flight slot; value / ms_per_step = the median repeat (max over ranks)",
                "repeats": repeats,
                "ms_per_step_min": 1e3 * dt_flight_min / K,
                "ms_per_step_max": 1e3 * dt_flight_max / K,
            },
            "rtf": dt_flight * world / audio_s,
            "x_realtime_per_gpu": audio_s / (dt_flight * world),
            "latency_ms_single_stream": 1e3 * dt_single / K,  # per call (= per utterance at batch 1), same method
            "latency_ms_single_stream_min": 1e3 * dt_single_min / K,
            "utterances_per_sec_single_stream": world * K * B / dt_single,
            "rtf_single_stream": dt_single * world / audio_s,
            "x_realtime_single_stream": audio_s / (dt_single * world),
            "end_to_end_tflops_per_gpu": flop_utt * K * B / dt_flight / 1e12,
            # (max over ranks) host CPU time of one pass of the headline region / its utterances: all threads of the rank's
            # process — the Python callers, ctypes, the HIP runtime's launch path (~170 launches per utterance)
            "host_cpu_ms_per_utterance": 1e3 * host_cpu_max / (K * B),
            "host_cpu_cores_busy_per_rank": host_cpu_max / host_wall_max if host_wall_max > 0 else None,
            "steady_state": None if dt_steady <= 0 else {
                "what": f"the same calls, the same {conc} in flight, in timed regions of {K_steady} steps instead of {K} (3 repeats, median; same "
                        "bracketing).  NOT the headline: `value` keeps the K-step region, which starts on an idle GPU (every caller begins "
                        "with its acoustic pass) and drains at its end — about 4 % of a 74 ms region at K = 20",
                "steps": K_steady,
                "utterances_per_sec": world * K_steady * B / dt_steady,
                "ms_per_step": 1e3 * dt_steady / K_steady,
                "vocoder_only_utterances_per_sec": None if dt_voc_steady <= 0 else world * K_steady * B / dt_voc_steady,
                "glow_under_load_ms": None if dt_voc_steady <= 0 else 1e3 * (dt_steady - dt_voc_steady) / K_steady,
            },
            "glow_under_load_ms": None if dt_voc <= 0 else 1e3 * (dt_flight - dt_voc) / K,
            "vocoder_only_under_load": None if dt_voc <= 0 else {
                "what": "the headline region with GlowTTS taken out: the same K utterances, the same calls in flight, each call the vocoder "
                        "alone (mi355tts_hifigan_infer_padded) on the device-resident mel GlowTTS produced for it; "
                        "glow_under_load_ms = ms_per_step - this = what the acoustic model costs per utterance when its ~140 small launches "
                        "compete with other calls' vocoder launches (alone it takes latency_ms_single_stream minus the vocoder's share)",
                "ms_per_step": 1e3 * dt_voc / K,
                "utterances_per_sec": world * K * B / dt_voc,
            },
            "call_coalescing": None if not (conc > 1 and B == 1) else {
                "what": f"option call_coalesce = {cc_other} (the library default, which the headline runs, is {cc_default}): concurrent batch-1 "
                        "mi355tts_synthesize callers become the rows of fused padded calls — acoustic pass and vocoder — at most that many "
                        "in flight (csrc/host_join.h); rows equal their solitary calls to f32 round-off (tests/test_emu_coalesce.py, "
                        "tests/test_gpu_parity.py::test_coalesced_calls_equal_their_solitary_results).  NOT the headline: the same timed "
                        "region with the option at its other setting",
                "call_coalesce": cc_other,
                "rows_per_pass": (cs1[1] - cs0[1]) / max(1, cs1[0] - cs0[0]),
                "utterances_per_sec": world * K * B / dt_flight_nc,
                "ms_per_step": 1e3 * dt_flight_nc / K,
                "steady_state_utterances_per_sec_rank0": None if not t_steady_nc else 10 * K * B / float(np.median(t_steady_nc)),
            },
            "denoiser_on": None if not dn_ok else {
                "denoiser_strength": 0.005,
                "note": "the reference CLI/server default (larynx/__main__.py:512-516); STFT denoiser on the device",
                "ms_per_step": 1e3 * dt_dn / K,
                "utterances_per_sec": world * K * B / dt_dn,
            },
            "half_mode": None if not half else {
                "dtype": "f16: the native fp16 vocoder — fp16 weights, fp16 activation planes in HBM between ALL layers (conv_pre, upsamplers, every "
                         "ResBlock conv, conv_post), one v_mfma_f32_32x32x16_f16 per product, f32 accumulate (csrc/conv_f16.h, hifigan_f16.h); "
                         + ("GlowTTS: the decoder's WaveNets in fp16, one launch per coupling block (csrc/wn_f16.h); encoder, durations, "
                            "start / end convs, coupling and InvConvNear in f32" if glow_f16 else "GlowTTS in f32"),
                "acoustic_model": "f16 decoder WaveNets (wn_f16_kernel)" if glow_f16 else "f32",
                "what": "the reference's `half` switch (`.half()` on both models, larynx/glow_tts.py:90-91, larynx/hifi_gan.py:96-97) on this backend; "
                        "NOT the headline — reported next to it",
                "parity": "vocoder: waveform error vs the reference's f32 output no larger than the reference's OWN generator under .half() on the same "
                          "input (tests/golden/*.npz: ref_half_rms, made by oracle/make_golden.py); acoustic model: mel error no larger than the "
                          "reference's OWN decoder under .half() (tests/golden/glow_half_reference.json, oracle/make_golden_glow_half.py); both: the fused "
                          "call within the reference's two models under .half().  Asserted per golden case in tests/test_gpu_parity.py::"
                          "test_f16_mode_against_the_reference, ::test_f16_acoustic_mode_against_the_reference, ::test_f16_fused_call_against_the_reference",
                "utterances_per_sec": world * K * B / dt_half_flight,
                "ms_per_step": 1e3 * dt_half_flight / K,
                "latency_ms_single_stream": 1e3 * dt_half_single / K,
                "x_realtime_per_gpu": audio_s / (dt_half_flight * world),
                "resblock_class_ms_per_step": half[2]["ms"] / K,
                "profile_ms_per_step": {k: c["ms"] / K for k, c in half[3].items() if c["launches"]},
                # ONE fp16 MFMA per product: executed matrix work = the algorithmic FLOPs
                "roofline": None if half[2]["ms"] <= 0 else {
                    "kernel": "HiFi-GAN ResBlock launches in the fp16 mode: conv_f16_group_kernel (the three MRF chains' same-geometry convs per launch), "
                              "HIP events per launch, single stream",
                    "bound": "mfma",
                    "achieved": half[2]["flop"] / (half[2]["ms"] * 1e-3) / 1e12,
                    "peak": BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": half[2]["flop"] / (half[2]["ms"] * 1e-3) / 1e12 / BF16_PEAK_TFLOPS,
                    "note": "algorithmic FLOPs (one MFMA per product) / dense fp16 peak (= the bf16 figure); launches measured one call at a time",
                    "launches": half[2]["launches"],
                    "avg_launch_us": 1e3 * half[2]["ms"] / max(1, half[2]["launches"]),
                    "frac_minus_event_overhead": half[2]["flop"] / (max(half[2]["ms"] - 1e-3 * ev_us * half[2]["launches"], 0.5 * half[2]["ms"]) * 1e-3) / 1e12 / BF16_PEAK_TFLOPS,
                    "by_kernel": by_kernel_table(half[4]),
                },
            },
            "bf16x3_mode": None if not half else {
                "dtype": "bf16x3: ResBlock convs and upsamplers on the bf16 matrix cores with split operands (three MFMAs per product, f32 planes; "
                         "conv_bf16.h) — the ACCURATE reduced mode: waveform RMS <= 2.9e-6 vs the reference's f32 output on the golden set",
                "utterances_per_sec": world * K * B / dt_x3_flight if dt_x3_flight > 0 else None,
                "ms_per_step": 1e3 * dt_x3_flight / K,
            },
            "weight_broadcast_seconds": broadcast_s if use_dist else None,
            "per_rank": {
                "utterances_per_sec": [K * B / t[0] for t in per_rank],
                "min": min(K * B / t[0] for t in per_rank), "max": max(K * B / t[0] for t in per_rank),
                "latency_ms_single_stream": [1e3 * t[1] / K for t in per_rank],
                "note": "each rank's own median over the repeats; `value` = all ranks' utterances / the slowest rank's time",
            },
            "host_affinity_rank0": affinity,
            # hardware-queue group of every worker stream as mi355tts_reserve measured it (creation order; a call takes the free
            # worker whose queue carries the fewest calls: include/mi355tts.h, mi355tts_worker_queue_groups)
            "worker_queue_groups_rank0": qgroups_headline,
            "process_group": (("nccl (RCCL)" if rccl else ("gloo" if not on_gpu else f"gloo (RCCL init failed: {pg_note}); every rank folded its own seeded weights")) if use_dist else None),
            "roofline": {
                "kernel": "HiFi-GAN ResBlock launches: rb_group_kernel (256- and 128-channel stages: the continuous-stream tile of rb_conv.h; conv_group_kernel = the k-split tile at the utterance lengths the promotion rule leaves alone), rb_pair_group_kernel (fused conv pairs of the 64/32-channel stages on four waves, rb_pair.h)",
                "bound": "mfma",
                "achieved": dom_tf_raw,
                "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": dom_tf_raw / FP32_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": f"profiles/{tpath.name if tpath.is_file() else '-'} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same "
                                  f"command, bytes per launch, FETCH_SIZE doubled per the guide's gfx950 correction for 16-B/lane reads)",
                "algorithmic_bytes_per_launch": "in + out planes of the launch's three members: e.g. stage 1 of 'high' 3 x (20.4 + 20.4) MB = 123 MB; "
                                                "class average 92 MB per launch (1656 MB per utterance over 18 launches)",
                "algorithmic_flop_per_launch": dom["flop"] / max(1, dom["launches"]),
                "launches": dom["launches"],
                "avg_launch_us": 1e3 * dom_ms_raw / max(1, dom["launches"]),
                "event_pair_overhead_us": ev_us,
                "avg_launch_us_minus_event_overhead": 1e3 * dom_ms / max(1, dom["launches"]),
                "achieved_minus_event_overhead": dom_tf,
                "frac_minus_event_overhead": dom_tf / FP32_PEAK_TFLOPS,
                # context, not the kernel figure: all algorithmic FLOPs of the utterances / the timed regions' wall time (GlowTTS, the
                # upsamplers and the output tail included) — what the chip sustains with the headline's calls in flight
                "end_to_end_under_load": {"achieved": flop_utt * K * B / dt_flight / 1e12, "frac": flop_utt * K * B / dt_flight / 1e12 / FP32_PEAK_TFLOPS,
                                          "steady_state_achieved": None if dt_steady <= 0 else flop_utt * K_steady * B / dt_steady / 1e12,
                                          "steady_state_frac": None if dt_steady <= 0 else flop_utt * K_steady * B / dt_steady / 1e12 / FP32_PEAK_TFLOPS,
                                          "unit": "TFLOP/s per GPU"},
                "share_of_step_time": dom["ms"] / (1e3 * dt_prof),
                "all_conv_mfma_ms_per_step": all_conv_ms / K,
                "schedule": ("serial_branches=1: one conv (or fused conv pair) per launch, the three MRF chains one after another "
                             "on one stream" if args.serial_branches else
                             "product schedule: one stream per call; the same-geometry convs (or fused conv pairs) of the three MRF "
                             "chains of a step are ONE grouped launch (conv_group_kernel / rb_group_kernel / rb_pair_group_kernel), each launch timed alone"),
                "timing": "HIP events on the launch stream around every launch, profiled pass of the same K steps; `avg_launch_us` / `achieved` / "
                          "`frac` = the RAW event times (they agree with rocprofv3's kernel durations of the same command: "
                          "profiles/r05_events_vs_rocprof.txt); `*_minus_event_overhead` = the same minus the cost of an EMPTY event pair "
                          "measured in this run (`event_pair_overhead_us`) — an over-correction, kept for comparison with round 4's line",
            },
            "profile_ms_per_step": {k_: v_["ms"] / K for k_, v_ in prof.items()},
        }
        out["roofline"]["by_kernel"] = by_kernel_table(prof_kernels)
        if args.precision != "f32":
            # the reduced modes as the line's own subject: price the class against the 16-bit matrix peak — per USEFUL product (fp16: one
            # MFMA per product; bf16x3 executes three), and no committed PMC traffic for these kernels
            rf = out["roofline"]
            rf["kernel"] = ("HiFi-GAN ResBlock launches in the fp16 mode: pair_f16_group_kernel (fused conv1 + conv2 of a step, 128 / 64 / 32 channels), "
                            "conv_f16_group_kernel (256 channels)" if args.precision == "f16" else
                            "HiFi-GAN ResBlock launches in the split-bf16 mode: conv_bf16_group_kernel / pair_bf16_group_kernel (three bf16 MFMAs per product)")
            rf["peak"] = BF16_PEAK_TFLOPS
            rf["frac"] = dom_tf_raw / BF16_PEAK_TFLOPS
            rf["frac_minus_event_overhead"] = dom_tf / BF16_PEAK_TFLOPS
            rf["frac_note"] = "achieved = algorithmic f32-path FLOPs / time, priced against the dense 16-bit MFMA peak (2.5 PFLOP/s): per useful product"
            e2e = rf["end_to_end_under_load"]
            e2e["frac"] = e2e["achieved"] / BF16_PEAK_TFLOPS
            e2e["steady_state_frac"] = None if e2e["steady_state_achieved"] is None else e2e["steady_state_achieved"] / BF16_PEAK_TFLOPS
            rf["traffic"] = None
            rf["traffic_source"] = "no committed PMC passes for this mode's kernels"
            rf["algorithmic_bytes_per_launch"] = "half the f32 figure in the fp16 mode (fp16 planes), conv1's plane never leaves LDS in the fused launches"
        if c3 is not None:
            out["config3"] = c3
        if c4 is not None:
            out["config4"] = c4
        if mb is not None:
            out["micro_batched"] = mb
        if c5 is not None:
            out["config5"] = c5
        if not args.no_cpu_baseline and world == 1 and on_gpu and not args.tiny:
            out["cpu_baseline"] = cpu_baseline(ids_host[0], args.length_scale)  # = golden case ljspeech_high_S120
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
