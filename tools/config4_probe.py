#!/usr/bin/env python
"""BASELINE config 4 alone (thorsten GlowTTS + hifi_gan 'medium', the golden batch of 8 rows per call, fused entry),
single stream — the command the round's rocprofv3 kernel trace / PMC passes of config 4 wrap.
Usage: python tools/config4_probe.py [calls] [name=value context options ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
from larynx_amd import hparams as HP, synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from larynx_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = Engine(0)
for kv in sys.argv[2:]:
    eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
g = eng.load_glow(HP.THORSTEN, synthetic.make_glow_state_dict(HP.THORSTEN, seed=1234))
v = eng.load_hifigan(HP.HIFIGAN_MEDIUM, synthetic.make_hifigan_state_dict(HP.HIFIGAN_MEDIUM, seed=1234))
rows, ls, src = bench.config4_rows(HP.THORSTEN.num_symbols)
s = ljspeech_audio_settings()
eng.reserve(2, g, v, max_batch=8, max_ids=120, max_frames=1440)
for i in range(3):
    fr, _, _ = eng.synthesize(g, v, rows, 0.667, ls, seed=i, audio_settings=s, frames_per_id_guess=12.0 / ls)
eng.set_profiling(True)
eng.profile_reset()
t0 = time.perf_counter()
for i in range(n):
    eng.synthesize(g, v, rows, 0.667, ls, seed=i, audio_settings=s, frames_per_id_guess=12.0 / ls)
dt = time.perf_counter() - t0
prof = eng.profile()
print(f"config4: {n} calls, {1e3 * dt / n:.3f} ms per call (host int16 out), frames {fr.tolist()} = {int(fr.sum())}")
for k, c in prof.items():
    if c["launches"]:
        print(f"  {k:36s} {c['launches'] // n:4d} launches/call {c['ms'] / n:8.3f} ms/call {c['flop'] / max(c['ms'], 1e-9) / 1e9:8.2f} TFLOP/s")
eng.close()
