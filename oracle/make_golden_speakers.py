"""Golden vectors for MULTI-SPEAKER GlowTTS voices (glow_tts/models.py:304-306, 318-319; layers.py:109-113, 141-154;
models.py:114-116, 128-132), made by running the reference's own `FlowGenerator` from /root/reference — test
infrastructure, like make_golden.py (which it borrows its helpers from).  None of the reference's shipped voices has more
than one speaker, so the model is the LJSpeech architecture with `n_speakers = 4`, `gin_channels = 48` and seeded random
weights (larynx_amd/synthetic.py); the same fixture sentence goes through three speakers.

  python -m oracle.make_golden_speakers     ->  tests/golden/multispeaker/ljspeech_4speakers.npz  (+ an oracle-vs-reference report)
"""
from __future__ import annotations

import dataclasses
import json

import numpy as np

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from oracle import audio_np, glow_tts_np
from oracle.make_golden import GOLDEN, build_ref_glow, fixture_ids, import_reference, ref_sentence

MULTI = dataclasses.replace(HP.LJSPEECH, n_speakers=4, gin_channels=48)


def main():
    gm, hm, hc, ra = import_reference()
    fx = fixture_ids()
    audio_cfg = dict(vars(ljspeech_audio_settings()))
    sd = synthetic.make_glow_state_dict(MULTI, seed=1234)
    model = build_ref_glow(gm, MULTI, sd)
    cases = [  # name, ids, speaker, noise_scale, length_scale
        ("echo_s0", fx["ljspeech:be_a_voice_not_an_echo"], 0, 0.667, 1.0),
        ("echo_s2", fx["ljspeech:be_a_voice_not_an_echo"], 2, 0.667, 1.0),
        ("dave_s3", fx["ljspeech:im_sorry_dave"], 3, 0.333, 0.8),
        ("short_s1", [3, 8, 4, 14, 2, 9, 30], 1, 0.0, 1.0),
    ]
    out, report = {}, {}
    for name, ids, spk, ns, ls in cases:
        ids = np.asarray(ids, np.int64)
        noise = np.random.default_rng(77).standard_normal((MULTI.mel_channels, 16 * len(ids) + 64)).astype(np.float32)
        mel, mel_voc, _, _, logw = ref_sentence(model, None, ra, ids, noise, ns, ls, audio_cfg, speaker_id=spk)
        taps = {}
        o_mel = glow_tts_np.glow_tts_infer(sd, MULTI, ids, noise, ns, ls, taps, speaker_id=spk)
        assert o_mel.shape == mel.shape, (o_mel.shape, mel.shape)
        o_voc = audio_np.mel_to_vocoder_input(o_mel, ljspeech_audio_settings())
        e = dict(F=int(mel.shape[1]), P=len(ids), speaker=spk, logw=float(np.abs(taps["logw"] - logw).max()),
                 mel=float(np.abs(o_mel - mel).max()), mel_voc=float(np.abs(o_voc - mel_voc).max()))
        report[name] = e
        print(name, json.dumps(e))
        assert e["mel"] < 2e-4 and e["logw"] < 1e-4, e
        out.update({f"{name}.ids": ids, f"{name}.speaker": np.int32(spk), f"{name}.noise_scale": np.float32(ns),
                    f"{name}.length_scale": np.float32(ls), f"{name}.mel": mel.astype(np.float32),
                    f"{name}.mel_voc": mel_voc.astype(np.float32), f"{name}.logw": logw.astype(np.float32)})
    # the speakers must actually matter in this fixture: same sentence, another voice
    assert out["echo_s0.mel"].shape != out["echo_s2.mel"].shape or np.abs(out["echo_s0.mel"] - out["echo_s2.mel"]).max() > 0.05
    (GOLDEN / "multispeaker").mkdir(exist_ok=True)
    np.savez_compressed(GOLDEN / "multispeaker" / "ljspeech_4speakers.npz", glow=json.dumps(MULTI.to_config()), noise_seed=np.int32(77),
                        names=json.dumps([c[0] for c in cases]), report=json.dumps(report, sort_keys=True), **out)


if __name__ == "__main__":
    main()
