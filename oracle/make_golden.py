"""TEST INFRASTRUCTURE — pin the oracle to the reference and write tests/golden/.

Runs ONLY in the build container (needs /root/reference and torch).  It
  1. imports the reference's own torch modules (`glow_tts.models.FlowGenerator`,
     `hifi_gan.models.Generator`) with a 2-line stub for the one missing
     third-party import (`dataclasses_json.DataClassJsonMixin`),
  2. loads the seeded synthetic checkpoints (larynx_amd.synthetic) through the
     modules' own `load_state_dict`, applies `store_inverse()` /
     `remove_weight_norm()` / `.eval()` exactly as `larynx/glow_tts.py:94-95`
     and `larynx/hifi_gan.py:99-100` do,
  3. runs the reference path of `_sentence_task` (`larynx/__init__.py:214-285`):
     FlowGenerator -> AudioSettings transforms (the reference's own
     `larynx/audio.py`) -> Generator -> `audio_float_to_int16`,
     with `torch.randn_like` replaced by a recorded noise tensor,
  4. asserts the numpy oracle reproduces every output, and
  5. writes the vectors as `tests/golden/*.npz`.

Usage:  python -m oracle.make_golden            (from the repo root)
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import types
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("LARYNX_REFERENCE", "/root/reference"))
GOLDEN = REPO / "tests" / "golden"

sys.path.insert(0, str(REPO))

from larynx_amd import hparams as HP  # noqa: E402
from larynx_amd import synthetic  # noqa: E402
from larynx_amd.audio import ljspeech_audio_settings  # noqa: E402
from oracle import audio_np, denoise_np, glow_tts_np, hifi_gan_np  # noqa: E402


def import_reference():
    import torch  # noqa: F401

    stub = types.ModuleType("dataclasses_json")

    class DataClassJsonMixin:  # the only missing third-party symbol
        pass

    stub.DataClassJsonMixin = DataClassJsonMixin
    sys.modules.setdefault("dataclasses_json", stub)
    sys.path.insert(0, str(REF))
    import glow_tts.models as gm
    import hifi_gan.config as hc
    import hifi_gan.models as hm

    # larynx/audio.py is importable on its own (numpy only); larynx/__init__ is
    # not (needs gruut/onnxruntime), so load the file directly.
    import importlib.util

    spec = importlib.util.spec_from_file_location("_ref_larynx_audio", REF / "larynx" / "audio.py")
    ra = importlib.util.module_from_spec(spec)
    sys.modules["_ref_larynx_audio"] = ra
    spec.loader.exec_module(ra)
    return gm, hm, hc, ra


def build_ref_glow(gm, hp: HP.GlowHParams, sd, prepare=None):
    import torch

    model = gm.FlowGenerator(
        n_vocab=hp.num_symbols,
        hidden_channels=hp.hidden_channels,
        filter_channels=hp.filter_channels,
        filter_channels_dp=hp.filter_channels_dp,
        out_channels=hp.mel_channels,
        kernel_size=hp.kernel_size,
        n_heads=hp.n_heads,
        n_layers_enc=hp.n_layers_enc,
        p_dropout=0.1,
        n_blocks_dec=hp.n_blocks_dec,
        kernel_size_dec=hp.kernel_size_dec,
        dilation_rate=hp.dilation_rate,
        n_block_layers=hp.n_block_layers,
        p_dropout_dec=0.05,
        n_speakers=hp.n_speakers,
        gin_channels=hp.gin_channels,
        n_split=hp.n_split,
        n_sqz=hp.n_sqz,
        sigmoid_scale=False,
        window_size=hp.window_size,
        block_length=None,
        mean_only=hp.mean_only,
        hidden_channels_enc=hp.hidden_channels,
        hidden_channels_dec=hp.hidden_channels,
        prenet=hp.prenet,
    )
    tsd = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(tsd, strict=True), None
    if prepare is not None:  # e.g. `.half()`, which larynx/glow_tts.py:90-94 applies BEFORE store_inverse()
        prepare(model)
    model.decoder.store_inverse()  # larynx/glow_tts.py:94
    model.eval()
    return model


def build_ref_hifigan(hm, hc, hp: HP.HifiGanHParams, sd):
    import torch

    cfg = hc.TrainingConfig()
    cfg.model = hc.ModelConfig(
        resblock=hp.resblock,
        upsample_rates=tuple(hp.upsample_rates),
        upsample_kernel_sizes=tuple(hp.upsample_kernel_sizes),
        upsample_initial_channel=hp.upsample_initial_channel,
        resblock_kernel_sizes=tuple(hp.resblock_kernel_sizes),
        resblock_dilation_sizes=tuple(tuple(d) for d in hp.resblock_dilation_sizes),
    )
    gen = hm.Generator(cfg)
    gen.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    gen.eval()
    gen.remove_weight_norm()  # larynx/hifi_gan.py:99-100
    return gen


def ref_sentence(gm_model, gen, ra, ids, noise, noise_scale, length_scale, audio_cfg, speaker_id=None):
    """The reference's `_sentence_task` data path (larynx/__init__.py:229-257) on
    the torch backend (larynx/glow_tts.py:123-151, larynx/hifi_gan.py:134-169)."""
    import torch

    text = torch.LongTensor(np.asarray(ids)).unsqueeze(0)
    lengths = torch.LongTensor([text.shape[1]])
    orig = torch.randn_like

    def fixed_randn_like(t, *a, **k):
        assert t.shape[0] == 1 and t.shape[1] == noise.shape[0]
        return torch.from_numpy(np.ascontiguousarray(noise[None, :, : t.shape[2]]))

    torch.randn_like = fixed_randn_like
    try:
        with torch.no_grad():
            g = None if speaker_id is None else torch.LongTensor([int(speaker_id)])  # larynx/glow_tts.py:125-130
            (mel, *_), _, (attn, logw, _) = gm_model(text, lengths, noise_scale=noise_scale, length_scale=length_scale, g=g)
    finally:
        torch.randn_like = orig
    mel = mel.cpu()
    settings = ra.AudioSettings(**audio_cfg)
    mels = mel.numpy()
    if settings.signal_norm:
        mels = settings.denormalize(mels)
    if settings.convert_db_to_amp:
        mels = settings.db_to_amp(mels)
    if settings.do_dynamic_range_compression:
        mels = settings.dynamic_range_compression(mels)
    mels = np.asarray(mels, np.float32)
    if gen is None:  # acoustic model only (make_golden_speakers.py)
        return mel.numpy()[0], mels[0], None, None, logw.numpy()[0, 0]
    with torch.no_grad():
        audio = gen(torch.from_numpy(mels)).squeeze(0).cpu().numpy()
    audio_i16 = ra.audio_float_to_int16(audio).squeeze()
    return mel.numpy()[0], mels[0], audio[0], audio_i16, logw.numpy()[0, 0]


def fixture_ids():
    """Real gruut+phonemes2ids output shipped with the reference
    (`local/*/samples/test_phonemes.csv`)."""
    out = {}
    for voice in ("en-us/ljspeech-glow_tts", "de-de/thorsten-glow_tts"):
        p = REF / "local" / voice / "samples" / "test_phonemes.csv"
        for line in p.read_text(encoding="utf-8").splitlines():
            if "|" in line:
                name, ids = line.split("|", 1)
                out[f"{voice.split('/')[1].split('-')[0]}:{name}"] = [int(v) for v in ids.split()]
    return out


def main():
    gm, hm, hc, ra = import_reference()
    GOLDEN.mkdir(parents=True, exist_ok=True)
    fx = fixture_ids()
    (GOLDEN / "fixture_phoneme_ids.json").write_text(json.dumps(fx, indent=0, sort_keys=True))
    audio_cfg = dict(vars(ljspeech_audio_settings()))

    cases = [
        # name, glow hp, vocoder hp, ids, noise_scale, length_scale
        ("ljspeech_high_echo", HP.LJSPEECH, HP.HIFIGAN_HIGH, fx["ljspeech:be_a_voice_not_an_echo"], 0.667, 1.0),
        ("ljspeech_medium_dave_ls12", HP.LJSPEECH, HP.HIFIGAN_MEDIUM, fx["ljspeech:im_sorry_dave"], 0.667, 1.2),
        ("ljspeech_low_echo", HP.LJSPEECH, HP.HIFIGAN_LOW, fx["ljspeech:be_a_voice_not_an_echo"], 0.333, 0.9),
        ("thorsten_medium_veg", HP.THORSTEN, HP.HIFIGAN_MEDIUM, fx["thorsten:haben_sie_ein_vegetarisches"], 0.667, 1.0),
        ("ljspeech_high_short5", HP.LJSPEECH, HP.HIFIGAN_HIGH, [3, 8, 4, 14, 2], 0.0, 1.0),
        ("ljspeech_high_long", HP.LJSPEECH, HP.HIFIGAN_HIGH, fx["ljspeech:it_took_me_quite_a_long_time_to_develop_a_voice"], 0.667, 1.0),
        # BASELINE config 2 at S: the very utterance bench.py times first (P = 120 synthetic ids, length_scale 0.65
        # -> ~620 frames), with recorded noise instead of the device RNG
        ("ljspeech_high_S120", HP.LJSPEECH, HP.HIFIGAN_HIGH,
         synthetic.synthetic_phoneme_ids(np.random.default_rng(1234), 120, HP.LJSPEECH.num_symbols), 0.667, 0.65),
        # the upper end of BASELINE config 3's length range (P = 200 ids -> ~1000 frames = 11.7 s of audio) on 'high': a value
        # check well past the standard utterance (multi-tile launches in every stage, the attention's 200 x 200 scores)
        ("ljspeech_high_P200", HP.LJSPEECH, HP.HIFIGAN_HIGH,
         synthetic.synthetic_phoneme_ids(np.random.default_rng(200), 200, HP.LJSPEECH.num_symbols), 0.667, 0.65),
    ]
    # BASELINE config 4: thorsten + 'medium', B = 8 variable length: the five thorsten fixture
    # sentences (19, 26, 31, 33, 64 ids) + synthetic rows of 47, 90, 120 ids.  The reference never batches
    # (SURVEY F7), so every row goes through it at B = 1; the HIP path must reproduce each row inside ONE padded batch.
    rng4 = np.random.default_rng(11)
    rows4 = [fx["thorsten:ich_bin_allergisch"], fx["thorsten:mir_geht_es_gut"], fx["thorsten:konnen_sie_bitte"],
             fx["thorsten:haben_sie_ein_vegetarisches"], fx["thorsten:fischers_fritze_fischt"]]
    rows4 += [list(synthetic.synthetic_phoneme_ids(rng4, n, HP.THORSTEN.num_symbols)) for n in (47, 90, 120)]
    assert [len(r) for r in rows4] == [19, 26, 31, 33, 64, 47, 90, 120]
    for b, r in enumerate(rows4):
        cases.append((f"batch8/thorsten_medium_row{b}", HP.THORSTEN, HP.HIFIGAN_MEDIUM, r, 0.667, 0.5))
    models = {}
    report = {}
    batch_rows = []
    for name, ghp, vhp, ids, ns, ls in cases:
        gkey, vkey = ("g", ghp), ("v", vhp)
        if gkey not in models:
            sd = synthetic.make_glow_state_dict(ghp, seed=1234)
            models[gkey] = (sd, build_ref_glow(gm, ghp, sd))
        if vkey not in models:
            sd = synthetic.make_hifigan_state_dict(vhp, seed=1234)
            models[vkey] = (sd, build_ref_hifigan(hm, hc, vhp, sd))
        gsd, gmodel = models[gkey]
        vsd, vmodel = models[vkey]
        ids = np.asarray(ids, np.int64)
        # batch rows draw their noise from one [8, M, 2200] tensor (row b of the batch the HIP path will run)
        if name.startswith("batch8/"):
            noise = np.random.default_rng(4).standard_normal((8, ghp.mel_channels, 2200)).astype(np.float32)[int(name[-1])]
        else:
            noise = np.random.default_rng(1234).standard_normal((ghp.mel_channels, 16 * len(ids) + 64)).astype(np.float32)
        mel, mel_voc, wav, wav_i16, logw = ref_sentence(gmodel, vmodel, ra, ids, noise, ns, ls, audio_cfg)
        F = mel.shape[1]
        # --- oracle must reproduce the reference ---
        taps = {}
        o_mel = glow_tts_np.glow_tts_infer(gsd, ghp, ids, noise, ns, ls, taps)
        assert o_mel.shape == mel.shape, (o_mel.shape, mel.shape)
        o_voc = audio_np.mel_to_vocoder_input(o_mel, ljspeech_audio_settings())
        o_wav = hifi_gan_np.hifigan_infer(vsd, vhp, mel_voc)
        o_i16 = audio_np.audio_float_to_int16(o_wav)
        e = dict(
            F=F,
            P=len(ids),
            logw=float(np.abs(taps["logw"] - logw).max()),
            mel=float(np.abs(o_mel - mel).max()),
            mel_voc=float(np.abs(o_voc - mel_voc).max()),
            wav_rms=float(np.sqrt(np.mean((o_wav - wav) ** 2))),
            wav_max=float(np.abs(o_wav - wav).max()),
            i16=int(np.abs(o_i16.astype(np.int32) - wav_i16.astype(np.int32)).max()),
            mel_mean=float(mel.mean()),
            mel_std=float(mel.std()),
            wav_absmean=float(np.abs(wav).mean()),
            wav_absmax=float(np.abs(wav).max()),
        )
        # --- what the reference's own `half` switch costs on this case: ITS generator under .half() (larynx/hifi_gan.py:96-97;
        # the caller hands it half mels, larynx/__init__.py:244-247) and under .bfloat16(), on the same vocoder input, against its
        # own f32 waveform.  The HIP library's reduced modes are held to these figures (tests/test_gpu_parity.py).
        import copy

        import torch

        half_extra = {}
        for tag, dt in (("half", torch.float16), ("bf16", torch.bfloat16)):
            hmodel = copy.deepcopy(vmodel).to(dt)
            with torch.no_grad():
                hw = hmodel(torch.from_numpy(mel_voc[None]).to(dt)).squeeze(0).float().cpu().numpy()[0]
            h16 = ra.audio_float_to_int16(hw[None]).squeeze()
            e[f"ref_{tag}_rms"] = float(np.sqrt(np.mean((hw - wav) ** 2)))
            e[f"ref_{tag}_max"] = float(np.abs(hw - wav).max())
            e[f"ref_{tag}_i16"] = int(np.abs(h16.astype(np.int32) - wav_i16.astype(np.int32)).max())
            half_extra[f"ref_{tag}_rms"] = np.float32(e[f"ref_{tag}_rms"])
            half_extra[f"ref_{tag}_max"] = np.float32(e[f"ref_{tag}_max"])
            half_extra[f"ref_{tag}_i16"] = np.int32(e[f"ref_{tag}_i16"])
            del hmodel
        report[name] = e
        print(name, json.dumps(e))
        assert e["mel"] < 2e-4 and e["wav_rms"] < 2e-5 and e["i16"] <= 1, e
        extra = {}
        if name == "ljspeech_medium_dave_ls12":
            # denoiser (larynx/hifi_gan.py:152-203) through the reference's own STFT helpers
            import torch

            strength = 0.1
            with torch.no_grad():
                bias_audio = vmodel(torch.zeros(1, 80, 88)).squeeze(0).cpu().numpy()
            bias_spec, _ = ra.transform(bias_audio)
            bias_spec = bias_spec[:, :, 0][:, :, None]
            spec, ang = ra.transform(wav[None])
            den = ra.inverse(np.clip(spec - bias_spec * strength, a_min=0.0, a_max=None), ang)
            den_i16 = ra.audio_float_to_int16(den).squeeze()
            o_bias = denoise_np.bias_spectrum(lambda m: hifi_gan_np.hifigan_infer(vsd, vhp, m))
            o_den = denoise_np.denoise(o_wav, o_bias, strength)
            assert np.abs(o_bias - bias_spec[0, :, 0]).max() < 1e-4 * max(1.0, float(bias_spec.max()))
            assert np.sqrt(np.mean((o_den - den[0]) ** 2)) < 2e-5
            report[name]["denoise_rms"] = float(np.sqrt(np.mean((o_den - den[0]) ** 2)))
            extra = dict(denoiser_strength=np.float32(strength), bias_spec=bias_spec[0, :, 0].astype(np.float32),
                         wav_denoised=den[0].astype(np.float32), wav_denoised_i16=den_i16, wav_denoised_stride=np.int32(1))
        if name.startswith("batch8/"):
            batch_rows.append(dict(ids=ids, mel=mel.astype(np.float32), wav=wav.astype(np.float32), wav_i16=wav_i16,
                                   ref_half_rms=half_extra["ref_half_rms"], ref_bf16_rms=half_extra["ref_bf16_rms"]))
            continue
        keep_wav = wav  # full waveforms (round 1 stored 1 sample in 7 of the long ones)
        np.savez_compressed(
            GOLDEN / f"{name}.npz",
            ids=ids,
            noise_scale=np.float32(ns),
            length_scale=np.float32(ls),
            mel=mel.astype(np.float32),
            mel_voc=mel_voc.astype(np.float32),
            wav=(keep_wav if keep_wav is not None else wav[::7]).astype(np.float32),
            wav_stride=np.int32(1 if keep_wav is not None else 7),
            wav_i16=(wav_i16 if keep_wav is not None else wav_i16[::7]),
            logw=logw.astype(np.float32),
            glow=json.dumps(ghp.to_config()),
            vocoder=json.dumps(vhp.to_config()),
            **extra,
            **half_extra,
        )
    (GOLDEN / "batch8").mkdir(exist_ok=True)
    np.savez_compressed(
        GOLDEN / "batch8" / "thorsten_medium_batch8.npz",
        noise_seed=np.int32(4), noise_scale=np.float32(0.667), length_scale=np.float32(0.5),
        glow=json.dumps(HP.THORSTEN.to_config()), vocoder=json.dumps(HP.HIFIGAN_MEDIUM.to_config()),
        **{f"{k}{b}": r[k] for b, r in enumerate(batch_rows) for k in ("ids", "mel", "wav", "ref_half_rms", "ref_bf16_rms")},  # int16 = audio_float_to_int16(wav)
    )
    (GOLDEN / "oracle_vs_reference.json").write_text(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
