"""The two forms of the GlowTTS decoder's WaveNet layers on the device: bits, launch counts, lone-call latency and calls per
second with N caller threads (GlowTTS only).  `python tools/wn_probe.py [--ids 120] [--threads 8] [--calls 40]`"""
import argparse
import json
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--ids", type=int, default=120)
ap.add_argument("--threads", type=int, default=8)
ap.add_argument("--calls", type=int, default=40)
ap.add_argument("--batch-lens", default="19,26,31,33,64,47,90,120")
args = ap.parse_args()

eng = Engine(device=0)
hp = HP.LJSPEECH
g = eng.load_glow(hp, synthetic.make_glow_state_dict(hp, seed=1234))
rng = np.random.default_rng(7)
ids = synthetic.synthetic_phoneme_ids(rng, args.ids, hp.num_symbols)
rows = [synthetic.synthetic_phoneme_ids(rng, int(n), hp.num_symbols) for n in args.batch_lens.split(",")]
eng.reserve(args.threads + 1, g, 0, max_batch=len(rows), max_frames=2048)
out = {}


def mel_of(x, **kw):
    m = eng.glow_infer(g, x, 0.667, 0.65, **kw)
    r = m.numpy("raw")
    fr = [int(f) for f in m.frames]
    m.free()
    return r, fr


for name, x, kw in (("batch1", ids, dict(seed=3)), ("batch8", rows, dict(seed=11))):
    res = {}
    for form in (2, 0):
        eng.set_option("wn_layer", form)
        eng.profile_reset()
        r, fr = mel_of(x, **kw)
        c = eng.kernel_counts()
        res[form] = (r, fr, {k: v for k, v in c.items() if v})
        # lone-call latency
        for _ in range(5):
            eng.glow_infer(g, x, 0.667, 0.65, **kw).free()
        t0 = time.perf_counter()
        for _ in range(args.calls):
            eng.glow_infer(g, x, 0.667, 0.65, **kw).free()
        lat = (time.perf_counter() - t0) / args.calls * 1e3
        # calls per second with N caller threads
        pool = ThreadPoolExecutor(args.threads)

        def work(_):
            for _ in range(args.calls):
                eng.glow_infer(g, x, 0.667, 0.65, **kw).free()

        list(pool.map(work, range(args.threads)))
        t0 = time.perf_counter()
        list(pool.map(work, range(args.threads)))
        thr = args.threads * args.calls / (time.perf_counter() - t0)
        pool.shutdown()
        out[f"{name}.form{form}"] = {"frames": fr, "kernels": res[form][2], "lone_ms": round(lat, 4), "calls_per_s_%dthr" % args.threads: round(thr, 1)}
    out[f"{name}.same_bits"] = bool(np.array_equal(res[2][0], res[0][0]))
    out[f"{name}.max_abs_diff"] = float(np.abs(res[2][0] - res[0][0]).max())
eng.set_option("wn_layer", 1)
print(json.dumps(out, indent=1))
