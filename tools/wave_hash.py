"""The waveform bits of one batch-1 vocoder call (synthetic 'high' vocoder, a 617-frame mel): `python tools/wave_hash.py [quality] [frames]`.
Run under two settings of an environment switch that must not change the arithmetic and compare the printed hashes
(tools/ab_bits.py does the same for two BUILDS on a ragged batch)."""
import hashlib
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from larynx_amd import hparams as HP  # noqa: E402
from larynx_amd import synthetic
from larynx_amd.engine import Engine

if __name__ == "__main__":
    quality = sys.argv[1] if len(sys.argv) > 1 else "high"
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 617
    eng = Engine(device=0)
    hp = HP.VOCODER_QUALITY[quality]
    v = eng.load_hifigan(hp, synthetic.make_hifigan_state_dict(hp, seed=1234))
    mel = (np.random.default_rng(5).standard_normal((1, hp.num_mels, frames)) * 1.5 - 4).astype(np.float32)
    eng.profile_reset()
    wav, _ = eng.hifigan_infer(v, eng.mel_from_numpy(mel, np.array([frames], np.int32)))
    counts = {k: n for k, n in eng.kernel_counts().items() if n}
    assert np.isfinite(wav).all() and np.abs(wav).max() > 1e-4
    print(quality, frames, hashlib.sha256(np.ascontiguousarray(wav).tobytes()).hexdigest()[:16], counts)
