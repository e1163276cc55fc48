import json
from pathlib import Path

import numpy as np

from larynx_amd import hparams as HP

GOLDEN = Path(__file__).resolve().parent / "golden"
CASES = sorted(p.stem for p in GOLDEN.glob("*.npz"))


def load_case(name):
    z = np.load(GOLDEN / f"{name}.npz")
    d = {k: z[k] for k in z.files}
    d["glow_hp"] = HP.GlowHParams.from_config(json.loads(str(d["glow"])))
    d["voc_hp"] = HP.HifiGanHParams.from_config(json.loads(str(d["vocoder"])))
    d["noise"] = np.random.default_rng(1234).standard_normal((d["glow_hp"].mel_channels, 16 * len(d["ids"]) + 64)).astype(np.float32)
    return d
