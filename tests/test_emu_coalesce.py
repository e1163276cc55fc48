"""Coalesced GlowTTS passes (csrc/host_join.h, option `glow_coalesce`) on the CPU emulator build: concurrent batch-1
`mi355tts_synthesize` calls — the reference's per-sentence thread pool (larynx/__init__.py:146-157, 187-190) — share GlowTTS
passes, and every caller still gets exactly the waveform of its own solitary call (same noise stream, same launches)."""
import threading

import numpy as np
import pytest

from larynx_amd import hparams as HP
from larynx_amd import synthetic
from larynx_amd.audio import ljspeech_audio_settings
from larynx_amd.ffi import Mi355ttsError


@pytest.fixture(scope="module")
def tiny(emu_engine):
    gsd = synthetic.make_glow_state_dict(HP.TINY_GLOW, seed=7)
    vsd = synthetic.make_hifigan_state_dict(HP.TINY_HIFIGAN, seed=7)
    emu_engine.set_option("glow_coalesce", 1)  # off by default
    yield dict(g=emu_engine.load_glow(HP.TINY_GLOW, gsd), v=emu_engine.load_hifigan(HP.TINY_HIFIGAN, vsd))
    emu_engine.set_option("glow_coalesce", 0)


def _concurrent(eng, fn, n):
    out, err = [None] * n, [None] * n
    bar = threading.Barrier(n)

    def work(i):
        try:
            bar.wait()
            out[i] = fn(i)
        except Exception as e:  # noqa: BLE001 - reported by the caller
            err[i] = e

    th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    return out, err


def test_concurrent_calls_share_passes_and_equal_their_solitary_results(emu_engine, tiny):
    eng, g, v = emu_engine, tiny["g"], tiny["v"]
    s = ljspeech_audio_settings()
    rng = np.random.default_rng(5)
    lens = (13, 7, 21, 9, 30, 17)
    ids = [synthetic.synthetic_phoneme_ids(rng, n, HP.TINY_GLOW.num_symbols) for n in lens]
    eng.set_option("glow_coalesce", 0)
    try:
        solo = [eng.synthesize(g, v, ids[i], 0.667, 1.0, seed=100 + i, audio_settings=s, want_float=True) for i in range(len(ids))]
    finally:
        eng.set_option("glow_coalesce", 1)
    p0, r0 = eng.coalesce_stats()
    out, err = _concurrent(eng, lambda i: eng.synthesize(g, v, ids[i], 0.667, 1.0, seed=100 + i, audio_settings=s, want_float=True), len(ids))
    assert not any(err), err
    p1, r1 = eng.coalesce_stats()
    assert r1 - r0 == len(ids)
    assert p1 - p0 < len(ids)  # the callers that arrived while the first pass ran shared the next one(s)
    for (fa, wa, ia), (fb, wb, ib) in zip(solo, out):
        assert np.array_equal(fa, fb)
        assert np.array_equal(ia, ib) and np.array_equal(wa, wb)  # noise on: each row drew from its OWN seed's stream


def test_incompatible_requests_do_not_share_a_pass(emu_engine, tiny):
    """Different length scales cannot be rows of one pass; each still gets its own result."""
    eng, g, v = emu_engine, tiny["g"], tiny["v"]
    s = ljspeech_audio_settings()
    rng = np.random.default_rng(6)
    ids = [synthetic.synthetic_phoneme_ids(rng, 12, HP.TINY_GLOW.num_symbols) for _ in range(4)]
    scales = (1.0, 1.3, 1.0, 1.3)
    eng.set_option("glow_coalesce", 0)
    try:
        solo = [eng.synthesize(g, v, ids[i], 0.5, scales[i], seed=7 + i, audio_settings=s) for i in range(4)]
    finally:
        eng.set_option("glow_coalesce", 1)
    out, err = _concurrent(eng, lambda i: eng.synthesize(g, v, ids[i], 0.5, scales[i], seed=7 + i, audio_settings=s), 4)
    assert not any(err), err
    for (fa, _, ia), (fb, _, ib) in zip(solo, out):
        assert np.array_equal(fa, fb) and np.array_equal(ia, ib)


def test_an_invalid_request_fails_alone_when_it_leads_and_with_its_pass_otherwise(emu_engine, tiny):
    """An out-of-range phoneme id is caught by the caller's own pre-check, before it can join a pass."""
    eng, g, v = emu_engine, tiny["g"], tiny["v"]
    s = ljspeech_audio_settings()
    rng = np.random.default_rng(8)
    good = synthetic.synthetic_phoneme_ids(rng, 10, HP.TINY_GLOW.num_symbols)
    bad = good.copy()
    bad[3] = HP.TINY_GLOW.num_symbols + 5
    out, err = _concurrent(eng, lambda i: eng.synthesize(g, v, bad if i == 1 else good, 0.667, 1.0, seed=3, audio_settings=s), 3)
    assert isinstance(err[1], Mi355ttsError) and err[0] is None and err[2] is None
    assert np.array_equal(out[0][2], out[2][2])
