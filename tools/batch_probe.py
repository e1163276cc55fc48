#!/usr/bin/env python
"""Throughput of ONE batched call (B utterances per glow_infer / hifigan_infer) vs B."""
import faulthandler
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd import hparams as HP, synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from larynx_amd.engine import Engine  # noqa: E402

faulthandler.dump_traceback_later(40, exit=True)
eng = Engine(0)
g = eng.load_glow(HP.LJSPEECH, synthetic.make_glow_state_dict(HP.LJSPEECH))
v = eng.load_hifigan(HP.HIFIGAN_HIGH, synthetic.make_hifigan_state_dict(HP.HIFIGAN_HIGH))
s = ljspeech_audio_settings()
rng = np.random.default_rng(0)
for B in [int(x) for x in (sys.argv[1:] or ["1", "2", "4", "8"])]:
    rows = [synthetic.synthetic_phoneme_ids(rng, 120, 46) for _ in range(B)]
    for it in range(3):
        t0 = time.perf_counter()
        mel = eng.glow_infer(g, rows, 0.667, 0.65, seed=it, audio_settings=s)
        t1 = time.perf_counter()
        eng.hifigan_infer(v, mel, want_float=False)
        t2 = time.perf_counter()
    fr = mel.frames.sum()
    print(f"B={B:2d}: glow {1e3*(t1-t0):7.2f} ms  hifigan(+D2H) {1e3*(t2-t1):7.2f} ms  -> {1e3*(t2-t0)/B:6.2f} ms/utt, {B/(t2-t0):6.1f} utt/s, frames/utt {fr/B:.0f}", flush=True)
eng.close()
