"""Do two builds of the library give the same waveform bits?  `python tools/ab_bits.py libA.so libB.so [quality]`
Runs the synthetic vocoder of bench.py on the same ragged batch of mels through each build and compares the samples.
Used for build switches that must not change the arithmetic (RB_WIDE_STORES, RBP_WIDE_STORES: profiles/NOTES.md)."""
import hashlib
import sys

import numpy as np

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.engine import Engine


def run(lib, quality):
    eng = Engine(device=0, library_path=lib)
    hp = HP.VOCODER_QUALITY[quality]
    v = eng.load_hifigan(hp, synthetic.make_hifigan_state_dict(hp, seed=1234))
    rng = np.random.default_rng(5)
    frames = np.array([823, 640, 97], np.int32)
    mel = (rng.standard_normal((len(frames), hp.num_mels, int(frames.max()))) * 1.5 - 4).astype(np.float32)
    wav, _ = eng.hifigan_infer(v, eng.mel_from_numpy(mel, frames))
    eng.unload(v)
    return np.ascontiguousarray(wav)


if __name__ == "__main__":
    quality = sys.argv[3] if len(sys.argv) > 3 else "high"
    a, b = run(sys.argv[1], quality), run(sys.argv[2], quality)
    print("%s: %s %s" % (quality, hashlib.sha256(a.tobytes()).hexdigest()[:16], hashlib.sha256(b.tobytes()).hexdigest()[:16]),
          "IDENTICAL" if np.array_equal(a, b) else "DIFFER: %d of %d samples, max |d| %.3g" % ((a != b).sum(), a.size, np.abs(a - b).max()))
    assert np.isfinite(a).all() and np.abs(a).max() > 1e-4
