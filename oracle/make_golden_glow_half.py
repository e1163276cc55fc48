"""TEST INFRASTRUCTURE — what the reference's own `half` switch costs on the ACOUSTIC model, per golden case.

Runs ONLY in the build container (needs /root/reference and torch).  The reference's `half` is `.half()` on the whole
FlowGenerator (larynx/glow_tts.py:90-91).  For every single-utterance case of tests/golden/*.npz (same seeded checkpoint, same ids,
same recorded noise as oracle/make_golden.py) this script runs the reference's OWN modules twice more and records, against the
reference's f32 mel of the case:

* `dec_half_*`  — the FlowGenerator with its `decoder` (FlowSpecDecoder, glow_tts/models.py:143-209) under `.half()` and the
  encoder / duration path in f32: the frame count is the f32 model's, so every mel value is comparable.  This is the figure
  the HIP library's fp16 acoustic mode (csrc/wn_f16.h: the decoder's WaveNets in fp16, everything else f32) is held to;
* `both_half_wav_*` — that mel through the reference's AudioSettings transforms (larynx/audio.py) and ITS generator under `.half()`
  (larynx/hifi_gan.py:96-97), against the case's f32 waveform: what `half` costs end to end when both models take it at the
  f32 model's frame count — the bar for the fused call with both HIP models in fp16; `dec_half_wav_rms`: the same mel through
  the f32 generator (the acoustic model's share of it);
* `full_half_*` — the whole model under `.half()`, as the reference runs it: recorded with its frame count; the mel error
  only when the durations (ceil of exp(logw)) happen to come out equal.

Output: tests/golden/glow_half_reference.json.   Usage:  python -m oracle.make_golden_glow_half
"""
from __future__ import annotations

import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from larynx_amd import hparams as HP  # noqa: E402
from larynx_amd import synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from oracle.make_golden import GOLDEN, build_ref_glow, build_ref_hifigan, import_reference  # noqa: E402


def run_ref(model, ids, noise, noise_scale, length_scale):
    import torch

    dt = next(model.encoder.parameters()).dtype
    text = torch.LongTensor(np.asarray(ids)).unsqueeze(0)
    lengths = torch.LongTensor([text.shape[1]])
    orig = torch.randn_like

    def fixed_randn_like(t, *a, **k):
        return torch.from_numpy(np.ascontiguousarray(noise[None, :, : t.shape[2]])).to(t.dtype)

    torch.randn_like = fixed_randn_like
    try:
        with torch.no_grad():
            (mel, *_), _, _ = model(text, lengths, noise_scale=noise_scale, length_scale=length_scale)
    finally:
        torch.randn_like = orig
    assert dt in (torch.float32, torch.float16)
    return mel.float().cpu().numpy()[0]


def decoder_half(m):
    """`m`'s decoder (already under .half()): inputs cast on the way in, the mel cast back on the way out."""
    inner = m.decoder.forward

    def fwd(z, z_mask, g=None, reverse=False):
        y, logdet = inner(z.half(), z_mask.half(), g=None if g is None else g.half(), reverse=reverse)
        return y.float(), logdet

    m.decoder.forward = fwd
    return m


def main():
    import torch

    gm, hm, hc, ra = import_reference()
    settings = ra.AudioSettings(**dict(vars(ljspeech_audio_settings())))
    out = {}
    models = {}
    vocs = {}
    for p in sorted(GOLDEN.glob("*.npz")):
        z = np.load(p)
        ghp = HP.GlowHParams.from_config(json.loads(str(z["glow"])))
        if ghp not in models:
            sd = synthetic.make_glow_state_dict(ghp, seed=1234)
            # three instances from the same state dict (store_inverse() leaves non-leaf tensors: no deepcopy)
            f32 = build_ref_glow(gm, ghp, sd)
            # .half() BEFORE store_inverse(), as larynx/glow_tts.py:90-94 orders them (the stored inverse is a half tensor)
            full = build_ref_glow(gm, ghp, sd, prepare=lambda m: m.half())
            models[ghp] = (f32, decoder_half(build_ref_glow(gm, ghp, sd, prepare=lambda m: m.decoder.half())), full)
        f32, dech, full = models[ghp]
        ids = z["ids"]
        ns, ls = float(z["noise_scale"]), float(z["length_scale"])
        noise = np.random.default_rng(1234).standard_normal((ghp.mel_channels, 16 * len(ids) + 64)).astype(np.float32)
        mel = run_ref(f32, ids, noise, ns, ls)
        assert np.array_equal(mel, z["mel"]), p.stem  # the golden's own mel: the same model, ids and noise
        md = run_ref(dech, ids, noise, ns, ls)
        assert md.shape == mel.shape
        e = dict(
            F=int(mel.shape[1]),
            mel_absmax=float(np.abs(mel).max()),
            dec_half_max=float(np.abs(md - mel).max()),
            dec_half_rms=float(np.sqrt(np.mean((md - mel) ** 2))),
        )
        vhp = HP.HifiGanHParams.from_config(json.loads(str(z["vocoder"])))
        if vhp not in vocs:
            vsd = synthetic.make_hifigan_state_dict(vhp, seed=1234)
            vocs[vhp] = (build_ref_hifigan(hm, hc, vhp, vsd), build_ref_hifigan(hm, hc, vhp, vsd).half())
        gen, gen_h = vocs[vhp]
        mv = md[None]  # larynx/__init__.py:229-243: denormalize -> db_to_amp -> dynamic_range_compression
        if settings.signal_norm:
            mv = settings.denormalize(mv)
        if settings.convert_db_to_amp:
            mv = settings.db_to_amp(mv)
        if settings.do_dynamic_range_compression:
            mv = settings.dynamic_range_compression(mv)
        mv = torch.from_numpy(np.asarray(mv, np.float32))
        wav = z["wav"]
        assert int(z["wav_stride"]) == 1
        with torch.no_grad():
            w32 = gen(mv).squeeze(0).numpy()[0]
            w16 = gen_h(mv.half()).squeeze(0).float().numpy()[0]
        e["dec_half_wav_rms"] = float(np.sqrt(np.mean((w32 - wav) ** 2)))
        e["both_half_wav_rms"] = float(np.sqrt(np.mean((w16 - wav) ** 2)))
        e["both_half_wav_max"] = float(np.abs(w16 - wav).max())
        e["wav_rms"] = float(np.sqrt(np.mean(wav ** 2)))
        h16 = ra.audio_float_to_int16(w16[None]).squeeze()  # larynx/audio.py: peak-normalised int16, as _sentence_task returns it
        e["both_half_i16"] = int(np.abs(h16.astype(np.int32) - z["wav_i16"].astype(np.int32)).max())
        try:
            mf = run_ref(full, ids, noise, ns, ls)
            e["full_half_F"] = int(mf.shape[1])
            if mf.shape == mel.shape:
                e["full_half_max"] = float(np.abs(mf - mel).max())
                e["full_half_rms"] = float(np.sqrt(np.mean((mf - mel) ** 2)))
        except RuntimeError as ex:  # an operator without a CPU half kernel in this torch build
            e["full_half_error"] = str(ex).splitlines()[0][:200]
        out[p.stem] = e
        print(p.stem, json.dumps(e))
    (GOLDEN / "glow_half_reference.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")


if __name__ == "__main__":
    main()
